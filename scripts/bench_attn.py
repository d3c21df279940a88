"""Fused-attention micro-benchmark: forward and backward of the step's attention shapes through the C ABI.
    python scripts/bench_attn.py            # TFLOP/s per shape (algorithmic FLOPs: causal = lower triangle)
MK_ATTN_DQ_ASYNC=1 selects the asynchronous-staging dq kernel (one workgroup per CU; A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
SHAPES = [("llama cfg3 (B32 H32 S144 hd128 causal)", 32, 32, 144, 128, True),
          ("llama cfg4 (B4 H32 S2048 hd128 causal)", 4, 32, 2048, 128, True),
          ("llama S4096 (B2 H32 S4096 hd128 causal)", 2, 32, 4096, 128, True),
          ("noncausal (B4 H32 S2048 hd128)", 4, 32, 2048, 128, False),
          ("whisper (B32 H8 S1500 hd64)", 32, 8, 1500, 64, False),
          ("clip (B32 H16 S257 hd64)", 32, 16, 257, 64, False)]
for name, B, H, S, hd, causal in SHAPES:
    D = H * hd
    g = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda: (torch.randn(B, S, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    q, k, v, do = mk(), mk(), mk(), mk()
    o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(q); dv = torch.empty_like(q)
    lse = torch.empty(B * H * S, dtype=torch.float32, device=dev)
    sc = hd ** -0.5
    a = (B, H, S, S, hd, D, S * D, D, S * D, D, S * D, D, S * D, sc)
    fwd = lambda: ops.flash_attn_fwd(q, k, v, o, *a, causal=causal, lse=lse)
    bwd = lambda: ops.flash_attn_bwd(q, k, v, o, do, lse, dq, dk, dv, *a, causal=causal)
    pairs = S * S - (S * (S - 1) // 2 if causal else 0)
    fl = 4.0 * pairs * hd * B * H
    variants = [("fwd", fwd, 1.0, {}), ("bwd", bwd, 2.0, {})]
    if S >= 1024:       # A/B of the forward kernels in one process (the launcher reads the switches per call)
        e4, e8 = {"MK_ATTN_FWD8_MIN": "0"}, {"MK_ATTN_FWD8_MIN": "1024", "MK_ATTN_FWD8_HD64": "1", "MK_ATTN_FWD4X64": "0"}
        e464 = {"MK_ATTN_FWD8_MIN": "1024", "MK_ATTN_FWD4X64": "1"}
        variants = [("fwd 4-wave", fwd, 1.0, e4), ("fwd 8-wave", fwd, 1.0, e8)] + ([("fwd 4x64", fwd, 1.0, e464)] if (hd == 128 and os.environ.get("MK_EXPERIMENTS")) else [])
        variants = variants + variants + [("bwd", bwd, 2.0, {})]
    for tag, fn, mult, env in variants:
        for k_ in ("MK_ATTN_FWD8_MIN", "MK_ATTN_FWD8_HD64", "MK_ATTN_FWD4X64"):
            os.environ.pop(k_, None)
        os.environ.update(env)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:44s} {tag:10s}: {ms * 1e3:9.1f} us  {fl * mult / (ms * 1e-3) / 1e12:7.1f} TFLOP/s")
