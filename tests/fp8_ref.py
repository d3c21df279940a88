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


class _FakeQuantLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        ctx.save_for_backward(x, W)
        ctx.has_b = b is not None
        y = fq_rows(x) @ fq_rows(W).t()
        return y + b if b is not None else y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dx = fq_rows(dy) @ fq_rows(W.t()).t()                 # W^T with one scale per input channel
        dW = dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1])     # grad-weight stays unquantised
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_b else None
        return dx, dW, db


@contextlib.contextmanager
def fake_quant_oracle(sites=("qkv", "align")):
    old = restate.FP8_LINEAR, restate.FP8_SITES
    restate.FP8_LINEAR, restate.FP8_SITES = (lambda x, W, b=None: _FakeQuantLinear.apply(x, W, b)), tuple(sites)
    try:
        yield
    finally:
        restate.FP8_LINEAR, restate.FP8_SITES = old
