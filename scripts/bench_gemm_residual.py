"""o_proj / down_proj forward (x W^T + residual) at the cfg-3 step's shapes on v7 (cfg 11) and v9 (cfg 15), rotating operand
copies so that every launch reads cold weights and residuals (as inside the step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import lib as _L, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _L.load()
NC = 6


def bench(M, N, K, cfg, residual, iters=30):
    A = [(torch.randn(M, K, device=dev)).bfloat16() for _ in range(NC)]
    B = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(NC)]
    R = [torch.randn(M, N, device=dev).bfloat16() for _ in range(NC)]
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lib.mk_gemm_set_cfg(cfg)
    try:
        def run(i):
            ops.gemm_raw(A[i % NC], B[i % NC], C, M, N, K, K, K, N, R=R[i % NC] if residual else None, ldr=N if residual else 0)
        for i in range(NC):
            run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
    finally:
        lib.mk_gemm_set_cfg(-1)
    return e0.elapsed_time(e1) / iters


for rd in range(2):
    for M, N, K in ((4608, 4096, 4096), (4608, 4096, 11008), (4096, 4096, 4096)):
        for residual in (False, True):
            t = {c: bench(M, N, K, c, residual) for c in (11, 15)}
            fl = 2.0 * M * N * K
            print(f"{M}x{N}x{K} residual={int(residual)}: v7 {t[11] * 1e3:7.1f} us ({fl / t[11] / 1e9:6.0f} TF)   "
                  f"v9 {t[15] * 1e3:7.1f} us ({fl / t[15] / 1e9:6.0f} TF)")
