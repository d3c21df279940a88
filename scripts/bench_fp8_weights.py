"""the once-per-optimizer-step weight quantisation of BASELINE cfg 5 (q|k|v of LLaMA-13B: [15360, 5120] x 40): row-scaled e4m3 copy
for the forward, column-scaled TRANSPOSED copy for the grad-input GEMM -- microseconds per matrix over 8 distinct (cold) weights"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
Ws = [(torch.randn(15360, 5120, device=dev) * 0.02).bfloat16() for _ in range(8)]
def bench(fn, iters=3):
    for W in Ws[:2]: fn(W)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for W in Ws: fn(W)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * len(Ws)) * 1e3
mb = Ws[0].numel() * 2 / 1e6
tr = bench(lambda W: ops.quantize_fp8_rows(W))
tc = bench(lambda W: ops.quantize_fp8_cols_t(W))
print(f"MK_FP8_COLAMAX_RPB={os.environ.get('MK_FP8_COLAMAX_RPB', 'default')}: rows {tr:7.1f} us ({1.5 * mb / tr:.2f} TB/s of 1.5 x {mb:.0f} MB)   "
      f"cols_t {tc:7.1f} us ({2.5 * mb / tc:.2f} TB/s of 2.5 x {mb:.0f} MB)   x 40 layers = {(tr + tc) * 40 / 1e3:.2f} ms per step")
