"""The alignment attention's batched products (modeling.py:882-910 as engine._align_fwd / _align_bwd run them: batch folded
into the query axis, Lq = 6 x 32 = 192 queries, 16 heads of 256, keys = the 32,007-row table + bias_k + zero row) under each
GEMM kernel configuration forced (mk_gemm_set_cfg): is the default (128x128, batched) the right kernel for them?"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import engine as eng, lib as _L, ops  # noqa: E402

dev = torch.device("cuda:0")
D, H, Lq, V = 4096, 16, 192, 32007
hd, Lk = D // H, V + 2
Lkp = (Lk + 63) // 64 * 64
g = torch.Generator(device=dev).manual_seed(0)
q = (torch.randn(Lq, D, device=dev, generator=g) * 0.5).bfloat16()
kv = (torch.randn(Lkp, 2 * D, device=dev, generator=g) * 0.5).bfloat16()
kv[Lk:].zero_()
do = (torch.randn(Lq, D, device=dev, generator=g) * 0.1).bfloat16()
o = torch.empty(Lq, D, device=dev, dtype=torch.bfloat16)
dq, dkv = torch.empty_like(q), torch.empty_like(kv)
T = eng.TDesc
kd, vd = T(kv, 2 * D, 0, 0), T(kv, 2 * D, 0, D)
lib = _L.load()


def fwd():
    return eng.attention_fwd(T(q, D, 0), kd, vd, T(o, D, 0), 1, H, Lq, Lk, hd, math.sqrt(1.0 / hd), p=0.1, seed=7, Lk_pad=Lkp)


def bwd(probs, pd):
    eng.attention_bwd(T(do, D, 0), T(q, D, 0), kd, vd, probs, pd, T(dq, D, 0), T(dkv, 2 * D, 0, 0), T(dkv, 2 * D, 0, D), 1, H,
                      Lq, Lk, hd, math.sqrt(1.0 / hd), p=0.1, seed=7, Lk_pad=Lkp)


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ref = None
for rd in range(2):
    for cfg in (-1, 5, 11, 0):
        lib.mk_gemm_set_cfg(cfg)
        try:
            tf = timed(lambda: fwd())
            probs, pd = fwd()
            p0 = probs.clone()

            def b():
                probs.copy_(p0)
                bwd(probs, pd)
            tb = timed(b)
            chk = (o.float().abs().mean().item(), dq.float().abs().mean().item(), dkv.float().abs().mean().item())
            print(f"round {rd} cfg {cfg:3d}: forward {tf:.3f} ms, backward (+ one probs copy) {tb:.3f} ms; |o| |dq| |dkv| means {chk}")
        except Exception as e:      # noqa: BLE001
            print(f"round {rd} cfg {cfg}: {e!r}"[:200])
lib.mk_gemm_set_cfg(-1)
