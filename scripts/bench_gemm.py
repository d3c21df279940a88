"""Micro-benchmark of the bf16 MFMA GEMM at the LLaMA-7B shapes of BASELINE cfg 3
(M = 32*144 = 4608 tokens).  Writes one JSON line per shape to stdout / gpurun_out."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from macaw_llm_amd import ops

dev = torch.device("cuda:0")


def bench(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    M = 4608
    shapes = [("qkvo", 4096, 4096), ("gate/up", 11008, 4096), ("down", 4096, 11008),
              ("lm_head", 32007, 4096), ("align_kv", 8192, 4096)]
    out = []
    for name, N, K in shapes:
        Mx = 32007 if name == "align_kv" else M
        x = torch.randn(Mx, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        ldc = (N + 63) // 64 * 64
        y = torch.empty(Mx, ldc, device=dev, dtype=torch.bfloat16)
        dy = torch.randn(Mx, N, device=dev).bfloat16()
        t_f = bench(lambda: ops.gemm_raw(x, W, y, Mx, N, K, K, K, ldc))
        dx = torch.empty(Mx, K, device=dev, dtype=torch.bfloat16)
        t_dx = bench(lambda: ops.linear_dx(dy, W, out=dx))
        dW = torch.empty(N, K, device=dev, dtype=torch.bfloat16)
        t_dw = bench(lambda: ops.linear_dw(dy, x, out=dW))
        fl = 2.0 * Mx * N * K
        rec = dict(name=name, M=Mx, N=N, K=K, fwd_ms=t_f * 1e3, dx_ms=t_dx * 1e3, dw_ms=t_dw * 1e3,
                   fwd_tf=fl / t_f / 1e12, dx_tf=fl / t_dx / 1e12, dw_tf=fl / t_dw / 1e12)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, W, y, dy, dx, dW
    # correctness spot check of the big shapes against torch.matmul (on-device cross-check only)
    x = torch.randn(512, 4096, device=dev).bfloat16()
    W = (torch.randn(1024, 4096, device=dev) * 0.02).bfloat16()
    y = ops.linear_fwd(x, W)
    ref = (x.float() @ W.float().t())
    print(json.dumps(dict(check_max_err=(y.float() - ref).abs().max().item(),
                          ref_max=ref.abs().max().item())))


if __name__ == "__main__":
    main()
