"""fp8 (f8f6f4 MFMA) vs bf16 GEMM throughput at the QKV / alignment shapes of BASELINE cfg 3/5."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for name, M, N, K in [("qkv 7B", 4608, 12288, 4096), ("qkv 13B", 4608, 15360, 5120), ("align K/V", 32007, 8192, 4096),
                      ("gate|up", 4608, 22016, 4096)]:
    x = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t16 = bench(lambda: ops.linear_fwd(x, W, out=y))
    xq, sx = ops.quantize_fp8(x); wq, sw = ops.quantize_fp8(W)
    t8 = bench(lambda: ops.linear_fp8(xq, sx, wq, sw, out=y))
    tq = bench(lambda: ops.quantize_fp8(x))
    fl = 2.0 * M * N * K
    print(f"{name:10s} M={M} N={N} K={K}: bf16 {fl / t16 / 1e12:6.0f} TF ({t16 * 1e6:7.1f} us) | fp8 {fl / t8 / 1e12:6.0f} TF "
          f"({t8 * 1e6:7.1f} us) | quantise activations {tq * 1e6:6.1f} us")
