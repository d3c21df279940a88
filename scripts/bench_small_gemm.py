"""Encoder-sized GEMMs (CLIP-L: M = 32*257, K = 1024/4096; Whisper: M = 32*1500, K = 512/2048)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
def bench(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for M, N, K in [(8224, 1024, 1024), (8224, 3072, 1024), (8224, 4096, 1024), (8224, 1024, 4096), (48000, 512, 512),
                (48000, 1536, 512), (48000, 2048, 512), (48000, 512, 2048), (8192, 1024, 1024), (4608, 4096, 4096)]:
    x = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    b = torch.randn(N, device=dev).bfloat16(); r = torch.randn(M, N, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = bench(lambda: ops.linear_fwd(x, W, out=y))
    t1 = bench(lambda: ops.linear_fwd(x, W, bias=b, residual=r, out=y))
    t2 = bench(lambda: ops.linear_fwd(x, W, bias=b, act=2, out=y))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: plain {fl/t0/1e12:5.0f} TF ({t0*1e6:6.1f} us) | +bias+res {fl/t1/1e12:5.0f} TF | +bias+quick_gelu {fl/t2/1e12:5.0f} TF")
