"""GPU: fp8 (OCP e4m3, per-tensor scale) quantisation and the f8f6f4-MFMA GEMM of BASELINE cfg 5
("fp8 MFMA for alignment-attn and QKV GEMMs").
  * quantisation is byte-exact against torch's float8_e4m3fn cast of the same scaled values;
  * the GEMM is checked against an fp32 matmul of the DE-QUANTISED operands (fp8 products are
    exact in fp32, so only the accumulation order and the bf16 output rounding differ: rtol 8e-3
    of the largest output), i.e. the kernel itself adds no error beyond its stated formats;
  * against the un-quantised bf16 product the error is the fp8 format's: e4m3 has 3 mantissa bits
    (2^-4 relative per operand), a K-term dot product averages it down -- bound 4 % of max|y|.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from macaw_llm_amd import ops  # noqa: E402


def _deq(q, s):
    return q.cpu().view(torch.float8_e4m3fn).float() * s.cpu().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fp8_quantize_matches_torch_cast(dev, dtype):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 136, generator=g) * 3).to(dtype)
    x[5, 7] = 1000.0                                  # defines amax; must map to +448 exactly
    q, s = ops.quantize_fp8(x.to(dev))
    amax = x.float().abs().max()
    assert abs(s.item() - (amax / 448).item()) <= 1e-7 * amax.item()
    ref = (x.float() * (torch.tensor(448.0) / amax)).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q.cpu(), ref.view(torch.uint8))
    assert _deq(q, s)[5, 7].item() == pytest.approx(1000.0, rel=1e-6)
    z, sz = ops.quantize_fp8(torch.zeros(4, 64, dtype=dtype, device=dev))   # all-zero tensor: scale 1
    assert not z.any() and sz.item() == 1.0


@pytest.mark.parametrize("M,N,K", [(256, 384, 512), (200, 136, 256), (4608, 12288, 4096), (1000, 4096, 1024)])
def test_fp8_gemm_vs_dequantised_fp32_and_bf16(dev, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16)
    xq, sx = ops.quantize_fp8(x.to(dev))
    wq, sw = ops.quantize_fp8(W.to(dev))
    y = ops.linear_fp8(xq, sx, wq, sw).float().cpu()
    ref = _deq(xq, sx) @ _deq(wq, sw).t()
    assert torch.isfinite(y).all()
    assert (y - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    full = x.float() @ W.float().t()
    e = (y - full).abs()
    assert e.max().item() <= 4e-2 * full.abs().max().item(), e.max().item() / full.abs().max().item()
    # fused epilogue: bias + residual
    y2 = ops.linear_fp8(xq, sx, wq, sw, bias=b.to(dev), residual=r.to(dev)).float().cpu()
    ref2 = ref + b.float()[None] + r.float()
    assert (y2 - ref2).abs().max().item() <= 8e-3 * ref2.abs().max().item() + 1e-2


def test_fp8_gemm_rejects_what_it_cannot_do(dev):
    x = torch.zeros(128, 192, dtype=torch.uint8, device=dev)          # K % 128 != 0
    s = torch.ones(1, device=dev)
    with pytest.raises(ops.MacawHipError):
        ops.linear_fp8(x, s, x, s)


def test_model_with_fp8_qkv_and_alignment(dev):
    """MM_LLMs.set_fp8 (BASELINE cfg 5): same weights and inputs through the bf16 engine and through
    the fp8 forward of q|k|v and of the alignment K/V projection.  e4m3 carries 3 mantissa bits, so
    the logits move by a few percent of their range (bound 8 %), the loss by < 5 %; the backward is
    the bf16 straight-through one and must stay finite and close in norm."""
    from golden_util import load_case
    from oracle import configs
    from test_model_gpu import build_model, to_dev
    from macaw_llm_amd import engine as E
    from macaw_llm_amd.modeling import MM_LLMs
    fx = load_case("micro_all")
    cfg = configs.get(fx["config_name"])
    model = build_model(cfg, fx["state"], torch.bfloat16, dev, fuse=True).eval()
    inp = to_dev(fx["inputs"], dev)

    def run():
        model.zero_grad(set_to_none=True)
        out = model(inputs=inp)
        out.loss.backward()
        gn = {n: p.grad.float().norm().item() for n, p in model.named_parameters() if p.grad is not None}
        return out.logits.float().cpu(), out.loss.item(), gn

    base_logits, base_loss, base_g = run()
    calls = {"n": 0}
    real = E.ops.linear_fp8

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    try:
        MM_LLMs.set_fp8(qkv=True, align=True)
        E.ops.linear_fp8 = counting
        logits, loss, g = run()
    finally:
        E.ops.linear_fp8 = real
        MM_LLMs.set_fp8(qkv=False, align=False)
    assert calls["n"] == cfg["llama"]["num_hidden_layers"] + 3      # every layer's q|k|v + 3 modalities
    assert torch.isfinite(logits).all()
    span = base_logits.abs().max().item()
    assert (logits - base_logits).abs().max().item() <= 8e-2 * span
    assert abs(loss - base_loss) <= 5e-2 * abs(base_loss)
    assert g.keys() == base_g.keys()
    for n in g:
        assert math.isfinite(g[n]) and abs(g[n] - base_g[n]) <= 0.25 * base_g[n] + 1e-4, (n, g[n], base_g[n])
    # switched off again: bit-identical to the first run
    again, _, _ = run()
    assert torch.equal(again, base_logits)
