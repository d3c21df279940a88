"""Torch restatement of the e4m3 quantisation scheme of macaw_llm_amd's fp8 path (engine.FP8), used
two ways by the tests:
  * byte-exact references for the quantisation kernels (same arithmetic: sc = 448 / amax in fp32,
    x * sc, clamp, round-to-nearest-even to OCP e4m3fn);
  * `fake_quant_oracle()`: a context manager that makes oracle.restate run its fp32 reference
    arithmetic with ONLY the fp8 path's operand quantisation added (forward: per-token x, per-
    output-channel W; grad-input: per-token dy, per-input-channel W; grad-weight untouched) -- the
    error of that run against the plain fp32 oracle is what the e4m3 FORMAT costs, independent of
    any kernel, and is the yardstick the HIP fp8 path is held to."""
import contextlib

import torch

from oracle import restate


def quant_rows_bytes(x):
    """(uint8 e4m3 bytes, scales) exactly as mk_fp8_quantize_rows computes them"""
    xf = x.float()
    amax = xf.abs().amax(dim=1, keepdim=True)
    sc = torch.where(amax > 0, torch.tensor(448.0, device=x.device) / amax, torch.ones_like(amax))
    q = (xf * sc).clamp(-448, 448).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).view(-1)


def fq_rows(t):
    """fake quantisation (quantise + de-quantise in fp32), one scale per row of the last dimension"""
    shp = t.shape
    t2 = t.reshape(-1, shp[-1]).float()
    q, s = quant_rows_bytes(t2)
    return (q.view(torch.float8_e4m3fn).float() * s[:, None]).reshape(shp).to(t.dtype)


def fq_blocks32(t):
    """fake quantisation with the block scales v_mfma_scale_f32_32x32x64_f8f6f4 takes natively (OCP MX): one E8M0
    (power-of-two) scale per 32 consecutive elements of the reduction (last) dimension, elements e4m3.  The scale is
    the smallest power of two that keeps the block's largest magnitude <= 448 (2^ceil(log2(amax / 448)): no clipping --
    kinder than the OCP MX rule floor(log2 amax) - 8, which saturates magnitudes in (448, 512) x scale)."""
    shp = t.shape
    K = shp[-1]
    pad = (-K) % 32
    x = t.reshape(-1, K).float()
    if pad:
        x = torch.nn.functional.pad(x, (0, pad))
    xb = x.view(x.shape[0], -1, 32)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    e = torch.ceil(torch.log2(amax.clamp_min(2.0 ** -120) / 448.0))
    sc = torch.exp2(e)
    q = (xb / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sc
    q = torch.where(amax > 0, q, torch.zeros_like(q)).view(x.shape[0], -1)
    return q[:, :K].reshape(shp).to(t.dtype)


FQ = fq_rows          # the operand quantiser of the yardstick; fake_quant_oracle(scheme="blocks32") swaps fq_blocks32 in


class _FakeQuantLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        y = FQ(x) @ FQ(W).t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = FQ(dy) @ FQ(W.t()).t()                           # W^T with one scale per input channel (rows: along the reduction)
        dW = dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1])     # grad-weight stays unquantised
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dW, db


@contextlib.contextmanager
def fake_quant_oracle(sites=("qkv", "align"), scheme="rows"):
    global FQ
    old = restate.FP8_LINEAR, restate.FP8_SITES, FQ
    restate.FP8_LINEAR, restate.FP8_SITES = (lambda x, W, b=None: _FakeQuantLinear.apply(x, W, b)), tuple(sites)
    FQ = {"rows": fq_rows, "blocks32": fq_blocks32}[scheme]
    try:
        yield
    finally:
        restate.FP8_LINEAR, restate.FP8_SITES, FQ = old
