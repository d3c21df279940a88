"""GPU parity tests of every C-ABI kernel against the CPU oracle (plain fp32 torch ops /
oracle.restate functions) on the same seeded inputs.  Tolerances are written per test:
  f32 kernels  : rtol 2e-5 (exact-f32 MFMA / fp32 VALU, only summation order differs)
  bf16 kernels : inputs are bf16-rounded on both sides, fp32 accumulation; the only error
                 is the final bf16 rounding (rel 2^-8 = 3.9e-3) -> rtol 8e-3 + small atol.
"""
import math

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from macaw_llm_amd import ops  # noqa: E402
from oracle import restate  # noqa: E402


def _tol(dtype):
    # (fp16 keeps 10 mantissa bits: it passes the bf16 bound with room to spare; one bound for both)
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=8e-3, atol=8e-3)


def _rand(shape, dtype, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _close(got, ref, dtype, scale=1.0, what=""):
    tol = _tol(dtype)
    got = got.float().cpu()
    ref = ref.float()
    err = (got - ref).abs().max().item()
    lim = tol["atol"] * scale + tol["rtol"] * ref.abs().max().item()
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e}"


DTYPES = [torch.float32, torch.bfloat16, torch.float16]
H16 = [torch.bfloat16, torch.float16]      # the two 16-bit element types of the MFMA kernels (csrc E16<>)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 107, 72), (37, 250, 1000),
                                   (6, 9, 4), (256, 384, 1000), (136, 264, 2007)])
def test_gemm_layouts(dev, dtype, a_red, b_red, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = _rand((K, M) if a_red else (M, K), dtype, g)
    B = _rand((K, N) if b_red else (N, K), dtype, g)
    Al = A.float().t() if a_red else A.float()
    Bl = B.float().t() if b_red else B.float()
    ref = Al @ Bl.t()
    C = torch.empty((M, N), dtype=dtype, device=dev)
    Ad, Bd = A.to(dev), B.to(dev)
    ops.gemm_raw(Ad, Bd, C, M, N, K, Ad.stride(0), Bd.stride(0), N, a_red=a_red, b_red=b_red)
    _close(C, ref, dtype, scale=math.sqrt(K), what=f"gemm {M}x{N}x{K} {a_red}{b_red}")


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 520, 128), (513, 1000, 192), (1031, 776, 320), (256, 256, 64 * 7),
                                   (2176, 1096, 1024)])
def test_gemm_v7_256_tile_kernel(dev, dtype, a_red, b_red, M, N, K):
    """the 256x256 quadrant-phase kernel (csrc/gemm_v7.hip) forced on ragged / short-K problems:
    whole tiles, the 128x128 sub-tile tail, nk = 2, 3, 5, 7 (prologue / penultimate / last tile
    paths), every operand layout -- against torch fp32 matmul on the same bf16 inputs, and within
    one bf16 ulp of the 128x128 kernel (whose K-split tail may re-associate the fp32 sum)."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = _rand((K, M) if a_red else (M, K), dtype, g)
    B = _rand((K, N) if b_red else (N, K), dtype, g, 0.1)
    ref = (A.float().t() if a_red else A.float()) @ (B.float() if b_red else B.float().t())
    Ad, Bd = A.to(dev), B.to(dev)
    ldc = (N + 7) // 8 * 8
    outs = {}
    try:
        for cfg in (11, 5):
            lib.mk_gemm_set_cfg(cfg)
            C = torch.full((M, ldc), float("nan"), dtype=dtype, device=dev)
            ops.gemm_raw(Ad, Bd, C, M, N, K, Ad.stride(0), Bd.stride(0), ldc, a_red=a_red, b_red=b_red)
            outs[cfg] = C
    finally:
        lib.mk_gemm_set_cfg(-1)
    _close(outs[11][:, :N], ref, dtype, scale=0.1 * math.sqrt(K), what=f"v7 {M}x{N}x{K} {a_red}{b_red}")
    assert torch.isnan(outs[11][:, N:].float()).all()      # pad columns of C untouched
    # agreement with the 128x128 kernel up to one bf16 ulp of the result (its K-split tail may
    # re-associate the fp32 sum)
    d = (outs[11][:, :N].float() - outs[5][:, :N].float()).abs().max().item()
    assert d <= 2 ** -7 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("a_red,b_red,M,N,K", [
    (False, False, 4608, 4096, 192),    # 288 tiles: 256 + a 32-tile tail as 64 x 128 eighths
    (False, True, 4608, 4096, 448),     # ... reduction-major B (transpose reads), nk = 7
    (False, False, 4600, 4090, 320),    # ... M and N edges inside the tail, unaligned C rows
    (False, False, 4385, 4096, 64),     # ... tail row group with ONE valid row, nk = 1
    (False, False, 4096, 8192, 192),    # 512 tiles = two whole rounds: walking workgroups, early prologue
    (True, True, 8192, 4096, 128),      # ... reduction-major A and B
    (False, True, 4096, 12288, 320),    # ... three tiles per walker
])
def test_gemm_v7_walkers_and_eighth_tail(dev, dtype, a_red, b_red, M, N, K):
    """more tiles than CUs on the 256 x 256 kernel (round 4): walking workgroups that request the next tile's
    first K-tiles before the epilogue of the current one (staging moved out of those ring slots), and the
    eighth-tile tail (not reached by the small shapes of the test above) -- with the full epilogue (alpha,
    column bias, residual, accumulate) against torch fp32 on the same 16-bit inputs, within one ulp of the
    128 x 128 kernel, and bit-identical run to run."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + 3 * N + K)
    A = _rand((K, M) if a_red else (M, K), dtype, g)
    B = _rand((K, N) if b_red else (N, K), dtype, g, 0.1)
    bias, R, C0 = _rand((N,), dtype, g), _rand((M, N), dtype, g), _rand((M, N), dtype, g)
    ref = 0.5 * ((A.float().t() if a_red else A.float()) @ (B.float() if b_red else B.float().t())) \
        + bias.float() + R.float() + C0.float()
    Ad, Bd, bd, Rd = A.to(dev), B.to(dev), bias.to(dev), R.to(dev)
    outs = {}
    try:
        for cfg in (11, 5, 11):
            lib.mk_gemm_set_cfg(cfg)
            C = C0.to(dev).clone()
            ops.gemm_raw(Ad, Bd, C, M, N, K, Ad.stride(0), Bd.stride(0), N, a_red=a_red, b_red=b_red, R=Rd, ldr=N,
                         bias=bd, bias_mode=1, accumulate=True, alpha=0.5)
            if cfg == 11 and 11 in outs:
                assert torch.equal(C, outs[11]), "256 x 256 kernel not reproducible run to run"
            outs[cfg] = C
    finally:
        lib.mk_gemm_set_cfg(-1)
    _close(outs[11], ref, dtype, scale=0.1 * math.sqrt(K) + 3.0, what=f"v7 walk/tail {M}x{N}x{K} {a_red}{b_red}")
    d = (outs[11].float() - outs[5].float()).abs().max().item()
    assert d <= 2 ** -7 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 520, 128), (513, 1000, 192), (1031, 776, 320), (256, 256, 64 * 7),
                                   (2176, 1096, 1024), (512, 768, 4096)])
def test_gemm_v8_one_wave_per_simd_kernel(dev, dtype, a_red, b_red, M, N, K):
    """the 256x256 one-wave-per-SIMD kernel (csrc/gemm_v8.hip) forced on ragged / short-K problems:
    edge tiles in M and N, nk = 2, 3, 5, 7, 16, 64 (prologue / steady state / penultimate / last tile
    paths, the three-slot A ring and two-slot B ring wrapping several times), every operand layout --
    against torch fp32 matmul on the same 16-bit inputs, and BIT-IDENTICAL to the v7 kernel (same
    MFMA, same k order per output element)."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    if not lib.mk_gemm_has_cfg(14):
        pytest.skip("gemm_v8 is an experiment kernel: build with MK_EXPERIMENTS=1 (superseded by gemm_v9)")
    g = torch.Generator().manual_seed(M + 3 * N + K + 1)
    A = _rand((K, M) if a_red else (M, K), dtype, g)
    B = _rand((K, N) if b_red else (N, K), dtype, g, 0.1)
    ref = (A.float().t() if a_red else A.float()) @ (B.float() if b_red else B.float().t())
    Ad, Bd = A.to(dev), B.to(dev)
    ldc = (N + 7) // 8 * 8
    outs = {}
    try:
        for cfg in (14, 11):
            lib.mk_gemm_set_cfg(cfg)
            C = torch.full((M, ldc), float("nan"), dtype=dtype, device=dev)
            ops.gemm_raw(Ad, Bd, C, M, N, K, Ad.stride(0), Bd.stride(0), ldc, a_red=a_red, b_red=b_red)
            outs[cfg] = C
    finally:
        lib.mk_gemm_set_cfg(-1)
    _close(outs[14][:, :N], ref, dtype, scale=0.1 * math.sqrt(K), what=f"v8 {M}x{N}x{K} {a_red}{b_red}")
    assert torch.isnan(outs[14][:, N:].float()).all()      # pad columns of C untouched
    assert torch.equal(outs[14][:, :N], outs[11][:, :N])


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(4096, 4096, 128), (4096, 4096, 192), (4096, 4096, 256), (4096, 4096, 64 * 7),
                                   (4608, 4096, 1024), (4096, 4352, 64 * 9),
                                   (8192, 4096, 256), (6144, 8192, 192)])   # 2 and 3 whole rounds: walking workgroups
def test_gemm_v9_hand_placed_k_loop(dev, dtype, a_red, b_red, M, N, K):
    """gemm_v9.hip: the 4-wave 256 x 256 kernel whose K loop is one generated inline-asm statement
    (scripts/gen_v9_loop.py; slot / wait protocol simulated on CPU in tests/test_v9_gen_cpu.py).  Whole tiles only,
    at least one full round of 256: nk = 2, 3, 4, 7, 9, 16 (peeled first tile / zero, one, many trips of the
    steady-state body / penultimate / last tile; the three-slot A ring and the two-slot B ring wrap), 288 and 272
    tiles (the spatial tail goes to v7's sub-tile kernels in a second launch), every operand layout, bf16 and fp16
    -- against torch fp32 matmul on the same 16-bit inputs and BIT-IDENTICAL to v7 (same MFMA, same k order per
    output element), twice in a row (no state left behind in LDS / registers by the previous launch)."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + 3 * N + K + 9)
    A = _rand((K, M) if a_red else (M, K), dtype, g)
    B = _rand((K, N) if b_red else (N, K), dtype, g, 0.1)
    Ad, Bd = A.to(dev), B.to(dev)
    ref = ((Ad.float().t() if a_red else Ad.float()) @ (Bd.float() if b_red else Bd.float().t())).cpu()
    outs = {}
    try:
        for cfg in (15, 11, 15):
            lib.mk_gemm_set_cfg(cfg)
            C = torch.full((M, N), float("nan"), dtype=dtype, device=dev)
            ops.gemm_raw(Ad, Bd, C, M, N, K, Ad.stride(0), Bd.stride(0), N, a_red=a_red, b_red=b_red)
            if cfg == 15 and 15 in outs:
                assert torch.equal(C, outs[15]), "v9 not reproducible run to run"
            outs[cfg] = C
    finally:
        lib.mk_gemm_set_cfg(-1)
    _close(outs[15], ref, dtype, scale=0.1 * math.sqrt(K), what=f"v9 {M}x{N}x{K} {a_red}{b_red}")
    assert torch.equal(outs[15], outs[11])


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("M,N,K,alpha", [(4096, 4096, 256, 1.0), (4608, 4096, 704, 0.5), (8192, 4096, 128, 1.0)])
def test_gemm_v9_register_epilogue_with_residual(dev, dtype, M, N, K, alpha):
    """round 6: v9's loops run on 16 x 16 x 32 MFMAs and its epilogue is all-in-registers (a lane owns 4 consecutive columns
    of one row: 8-byte accesses) in two fixed forms -- alpha, and alpha + RESIDUAL (o_proj / down_proj forward: modeling.py
    :215,:140 + the residual add of :281,:289 folded in).  The residual form against v7's LDS-transposed epilogue: bit for
    bit, whole rounds (walking) and 288 tiles (tail on v7's sub-tile kernels), C pitch 8- but not 16-byte aligned included."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = _rand((M, K), dtype, g).to(dev)
    B = _rand((N, K), dtype, g, 0.1).to(dev)
    for ldc in (N, N + 4):
        R = _rand((M, ldc), dtype, g).to(dev)
        outs = {}
        try:
            for cfg in (15, 11):
                lib.mk_gemm_set_cfg(cfg)
                C = torch.full((M, ldc), float("nan"), dtype=dtype, device=dev)
                ops.gemm_raw(A, B, C, M, N, K, K, K, ldc, R=R, ldr=ldc, alpha=alpha)
                outs[cfg] = C
        finally:
            lib.mk_gemm_set_cfg(-1)
        ref = (alpha * (A.float() @ B.float().t()) + R[:, :N].float()).cpu()
        _close(outs[15][:, :N], ref, dtype, scale=0.1 * math.sqrt(K) + 1.0, what=f"v9 residual epilogue {M}x{N}x{K} ldc {ldc}")
        assert torch.equal(outs[15][:, :N], outs[11][:, :N])


def test_gemm_v9_epilogue_and_fallback(dev):
    """a problem with bias + GELU + residual + accumulate forced to cfg 15 is routed to v7 (round 6: v9's register epilogue
    has the alpha and alpha + residual forms only), and so is a ragged one (whole tiles only); both stay correct."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    for M, N, K in ((4096, 4096, 256), (4000, 4100, 200)):
        g = torch.Generator().manual_seed(7)
        ld = (K + 7) // 8 * 8
        A = torch.zeros((M, ld), dtype=torch.bfloat16)
        A[:, :K] = _rand((M, K), torch.bfloat16, g)
        B = torch.zeros((N, ld), dtype=torch.bfloat16)
        B[:, :K] = _rand((N, K), torch.bfloat16, g, 0.1)
        A, B = A.to(dev), B.to(dev)
        bias = _rand((N,), torch.bfloat16, g).to(dev)
        ldc = (N + 7) // 8 * 8
        R = _rand((M, ldc), torch.bfloat16, g).to(dev)
        outs = {}
        try:
            for cfg in (15, 5):
                lib.mk_gemm_set_cfg(cfg)
                C = torch.ones((M, ldc), dtype=torch.bfloat16, device=dev)
                ops.gemm_raw(A, B, C, M, N, K, ld, ld, ldc, bias=bias, bias_mode=1, act=1, R=R, ldr=ldc, accumulate=True,
                             alpha=0.5)
                outs[cfg] = C
        finally:
            lib.mk_gemm_set_cfg(-1)
        ref = torch.nn.functional.gelu(0.5 * (A[:, :K].float() @ B[:, :K].float().t()) + bias.float()) \
            + R[:, :N].float() + 1.0
        _close(outs[15][:, :N], ref.cpu(), torch.bfloat16, scale=2.0, what=f"v9 epilogue {M}x{N}x{K}")
        d = (outs[15][:, :N].float() - outs[5][:, :N].float()).abs().max().item()
        assert d <= 2 ** -6 * ref.abs().max().item() + 1e-6


def test_gemm_v8_epilogue(dev):
    """bias + GELU + residual + accumulate through the 4 x 4-fragment LDS-transposed epilogue of v8."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    if not lib.mk_gemm_has_cfg(14):
        pytest.skip("gemm_v8 is an experiment kernel: build with MK_EXPERIMENTS=1 (superseded by gemm_v9)")
    M, N, K = 520, 776, 256
    g = torch.Generator().manual_seed(5)
    A = _rand((M, K), torch.bfloat16, g).to(dev)
    B = _rand((N, K), torch.bfloat16, g, 0.1).to(dev)
    bias = _rand((N,), torch.bfloat16, g).to(dev)
    R = _rand((M, N), torch.bfloat16, g).to(dev)
    outs = {}
    try:
        for cfg in (14, 11):
            lib.mk_gemm_set_cfg(cfg)
            C = torch.ones((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm_raw(A, B, C, M, N, K, K, K, N, bias=bias, bias_mode=1, act=1, R=R, ldr=N, accumulate=True, alpha=0.5)
            outs[cfg] = C
    finally:
        lib.mk_gemm_set_cfg(-1)
    ref = torch.nn.functional.gelu(0.5 * (A.float() @ B.float().t()) + bias.float()) + R.float() + 1.0
    _close(outs[14], ref.cpu(), torch.bfloat16, scale=2.0, what="v8 epilogue")
    assert torch.equal(outs[14], outs[11])


def test_gemm_odd_rows_reduction_major_last_element(dev):
    """regression: the hardware buffer range check is per DWORD -- with an odd row count the last
    valid element of the last k-row of a reduction-major operand shares its dword with the first
    out-of-range one and used to load as zero (the k = K-1 product of output row M-1 was lost;
    lm_head dW, V = 32007).  Make that single product dominate the result."""
    M, N, K = 775, 320, 256
    ldm = 776
    A = torch.zeros((K, ldm), dtype=torch.bfloat16)
    A[K - 1, M - 1] = 64.0
    B = torch.zeros((K, N), dtype=torch.bfloat16)
    B[K - 1] = torch.arange(N, dtype=torch.float32).to(torch.bfloat16)
    from macaw_llm_amd import lib as L
    lib = L.load()
    try:
        for cfg in (5, 11):
            lib.mk_gemm_set_cfg(cfg)
            C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            ops.gemm_raw(A.to(dev), B.to(dev), C, M, N, K, ldm, N, N, a_red=True, b_red=True)
            want = 64.0 * B[K - 1].float()
            assert torch.equal(C[M - 1].float().cpu(), want.to(torch.bfloat16).float()), cfg
            assert not C[: M - 1].any()
    finally:
        lib.mk_gemm_set_cfg(-1)


@pytest.mark.parametrize("K", [1031, 2007])
def test_gemm_k_tail_with_zero_padded_pitch(dev, K):
    """MK_GEMM_A_KPAD_ZERO: a K that is not a multiple of 64 on the MFMA tile kernels when the
    K-major operand is a pitched buffer with zero pad columns (d(logits) [tokens, 32064] x W for
    V = 32007) -- same result as the generic kernel / fp32 matmul, incl. odd K."""
    M, N = 520, 392
    g = torch.Generator().manual_seed(K)
    kp = (K + 63) // 64 * 64
    dy = torch.zeros((M, kp), dtype=torch.bfloat16)
    dy[:, :K] = _rand((M, K), torch.bfloat16, g)
    W = _rand((K, N), torch.bfloat16, g, 0.1)
    ref = dy[:, :K].float() @ W.float()
    dyd, Wd = dy.to(dev), W.to(dev)
    got = ops.linear_dx(dyd[:, :K], Wd, dy_pad_zero=True)
    _close(got, ref, torch.bfloat16, scale=0.1 * math.sqrt(K), what="k-tail padded")
    plain = ops.linear_dx(dyd[:, :K], Wd)          # generic edge kernel
    assert (got.float() - plain.float()).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(dev, dtype):
    g = torch.Generator().manual_seed(5)
    M, N, K = 192, 200, 136
    x, W = _rand((M, K), dtype, g), _rand((N, K), dtype, g, 0.1)
    bias, R, C0 = _rand((N,), dtype, g), _rand((M, N), dtype, g), _rand((M, N), dtype, g)
    xd, Wd, bd, Rd = x.to(dev), W.to(dev), bias.to(dev), R.to(dev)
    base = x.float() @ W.float().t()
    for act, fn in [(0, lambda t: t), (1, F.gelu), (2, restate.quick_gelu)]:
        out = ops.linear_fwd(xd, Wd, bias=bd, act=act, residual=Rd, alpha=0.5)
        ref = fn(0.5 * base + bias.float()) + R.float()
        _close(out, ref, dtype, scale=4.0, what=f"epilogue act={act}")
    # accumulate into existing C
    Cd = C0.to(dev).clone()
    ops.gemm_raw(xd, Wd, Cd, M, N, K, K, K, N, accumulate=True)
    _close(Cd, base + C0.float(), dtype, scale=4.0, what="accumulate")
    # per-row bias (bias_mode 2)
    brow = _rand((M,), dtype, g).to(dev)
    Cd = torch.empty((M, N), dtype=dtype, device=dev)
    ops.gemm_raw(xd, Wd, Cd, M, N, K, K, K, N, bias=brow, bias_mode=2)
    _close(Cd, base + brow.float().cpu()[:, None], dtype, scale=4.0, what="row bias")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_strided_pitched(dev, dtype):
    """attention-shaped batched GEMM: scores[b,h] = q[b,h] k[b,h]^T read straight out of the
    [B*S, H*hd] projection buffers, written into a pitched [B,H,S,Sp] buffer."""
    g = torch.Generator().manual_seed(9)
    Bn, H, S, hd = 2, 3, 50, 16
    D = H * hd
    q, k = _rand((Bn * S, D), dtype, g), _rand((Bn * S, D), dtype, g)
    qd, kd = q.to(dev), k.to(dev)
    Sp = 56
    sc = torch.zeros((Bn, H, S, Sp), dtype=dtype, device=dev)
    ops.gemm_raw(qd, kd, sc, S, S, hd, D, D, Sp, nb1=Bn, nb2=H, sA=(S * D, hd), sB=(S * D, hd),
                 sC=(H * S * Sp, S * Sp), alpha=0.25)
    qf = q.float().view(Bn, S, H, hd).transpose(1, 2)
    kf = k.float().view(Bn, S, H, hd).transpose(1, 2)
    ref = 0.25 * qf @ kf.transpose(-1, -2)
    _close(sc[..., :S], ref, dtype, scale=4.0, what="batched qk")
    assert (sc[..., S:] == 0).all()  # pad columns untouched
    # P @ V with V consumed red-major from the same projection layout, output into [B*S, D]
    p = torch.softmax(ref, -1).to(dtype)
    pd = torch.zeros((Bn, H, S, Sp), dtype=dtype, device=dev)
    pd[..., :S] = p.to(dev)
    v = _rand((Bn * S, D), dtype, g)
    vd = v.to(dev)
    ctx = torch.empty((Bn * S, D), dtype=dtype, device=dev)
    ops.gemm_raw(pd, vd, ctx, S, hd, S, Sp, D, D, b_red=True, nb1=Bn, nb2=H,
                 sA=(H * S * Sp, S * Sp), sB=(S * D, hd), sC=(S * D, hd))
    vf = v.float().view(Bn, S, H, hd).transpose(1, 2)
    refc = (p.float() @ vf).transpose(1, 2).reshape(Bn * S, D)
    _close(ctx, refc, dtype, scale=1.0, what="batched pv")


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_grads(dev, dtype):
    g = torch.Generator().manual_seed(11)
    M, N, K = 320, 264, 200
    dy, W, x = _rand((M, N), dtype, g), _rand((N, K), dtype, g, 0.1), _rand((M, K), dtype, g)
    dx = ops.linear_dx(dy.to(dev), W.to(dev))
    _close(dx, dy.float() @ W.float(), dtype, scale=2.0, what="dx")
    dw = ops.linear_dw(dy.to(dev), x.to(dev))
    _close(dw, dy.float().t() @ x.float(), dtype, scale=math.sqrt(M), what="dw")


@pytest.mark.parametrize("es", [2, 4])
def test_transpose(dev, es):
    dtype = torch.bfloat16 if es == 2 else torch.float32
    x = torch.randn(3, 70, 130).to(dtype)
    out = ops.transpose(x.to(dev))
    assert torch.equal(out.cpu(), x.transpose(1, 2).contiguous())
    x2 = x[0]
    assert torch.equal(ops.transpose(x2.to(dev)).cpu(), x2.t().contiguous())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(37, 128), (16, 4096), (5, 8)])
def test_rmsnorm(dev, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x, res, w = _rand((rows, cols), dtype, g), _rand((rows, cols), dtype, g), (1 + 0.1 * torch.randn(cols, generator=g)).to(dtype)
    h, y, rstd = ops.rmsnorm_fwd(x.to(dev), w.to(dev), 1e-6, res=res.to(dev))
    href = x + res  # in dtype, as the eager reference does
    assert torch.equal(h.cpu(), href)
    yref = restate.rms_norm(href, w, 1e-6)
    _close(y, yref, dtype, what="rmsnorm y")
    # no-residual form
    h2, y2, _ = ops.rmsnorm_fwd(x.to(dev), w.to(dev), 1e-6)
    _close(y2, restate.rms_norm(x, w, 1e-6), dtype, what="rmsnorm y (no res)")
    # backward vs autograd of the fp32 restatement
    hf = href.float().requires_grad_(True)
    wf = w.float().requires_grad_(True)
    dy, dres = _rand((rows, cols), dtype, g), _rand((rows, cols), dtype, g)
    restate.rms_norm(hf, wf, 1e-6).backward(dy.float())
    dx, dw = ops.rmsnorm_bwd(dy.to(dev), h, w.to(dev), rstd, dres=dres.to(dev))
    _close(dx, hf.grad + dres.float(), dtype, scale=2.0, what="rmsnorm dx")
    _close(dw, wf.grad, dtype, scale=math.sqrt(rows), what="rmsnorm dw")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(50, 64), (9, 1024), (3, 20), (7, 512), (6, 520), (5, 1032)])
def test_layernorm(dev, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows * cols)
    x = _rand((rows, cols), dtype, g)
    w, b = (1 + 0.1 * torch.randn(cols, generator=g)).to(dtype), (0.1 * torch.randn(cols, generator=g)).to(dtype)
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), 1e-5)
    xf, wf, bf = x.float().requires_grad_(True), w.float().requires_grad_(True), b.float().requires_grad_(True)
    yref = F.layer_norm(xf, (cols,), wf, bf, 1e-5)
    _close(y, yref.detach(), dtype, what="layernorm y")
    dy, dres = _rand((rows, cols), dtype, g), _rand((rows, cols), dtype, g)
    yref.backward(dy.float())
    dx, dw, db = ops.layernorm_bwd(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, dres=dres.to(dev))
    _close(dx, xf.grad + dres.float(), dtype, scale=2.0, what="layernorm dx")
    _close(dw, wf.grad, dtype, scale=math.sqrt(rows), what="layernorm dw")
    _close(db, bf.grad, dtype, scale=math.sqrt(rows), what="layernorm db")
    _close(ops.colsum(dy.to(dev)), dy.float().sum(0), dtype, scale=math.sqrt(rows), what="colsum")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd", [128, 32, 8])
def test_rope(dev, dtype, hd):
    g = torch.Generator().manual_seed(hd)
    Bn, S, H = 2, 7, 3
    cos, sin = restate.rotary_tables(hd, 64)
    cos, sin = cos.to(dtype), sin.to(dtype)
    q, k = _rand((Bn, S, H, hd), dtype, g), _rand((Bn, S, H, hd), dtype, g)
    pos = torch.arange(S).repeat(Bn, 1)
    qr, kr = restate.apply_rope(q.transpose(1, 2), k.transpose(1, 2), cos, sin, pos)
    qd = q.reshape(Bn * S, H * hd).to(dev).clone()
    posd = pos.reshape(-1).to(torch.int32).to(dev)
    ops.rope_(qd, cos.to(dev), sin.to(dev), posd, H, hd)
    got = qd.view(Bn, S, H, hd).transpose(1, 2).cpu()
    if dtype == torch.bfloat16:  # same rounding points as the eager bf16 reference: bit-exact
        nbad = (got != qr).sum().item()
        assert nbad == 0, (nbad, got.numel(), (got.float() - qr.float()).abs().max().item())
    else:
        _close(got, qr, dtype, what="rope")
    # inverse rotation undoes it (orthogonal): fp32 only, to rounding
    if dtype == torch.float32:
        ops.rope_(qd, cos.to(dev), sin.to(dev), posd, H, hd, inverse=True)
        _close(qd.view(Bn, S, H, hd), q, dtype, what="rope inverse")


@pytest.mark.parametrize("dtype", DTYPES)
def test_swiglu_act_add_cast(dev, dtype):
    g = torch.Generator().manual_seed(3)
    n = (33, 352)
    gt, u, da = _rand(n, dtype, g), _rand(n, dtype, g), _rand(n, dtype, g)
    a = ops.swiglu_fwd(gt.to(dev), u.to(dev))
    gf, uf = gt.float().clone().requires_grad_(True), u.float().clone().requires_grad_(True)
    ref = F.silu(gf) * uf
    _close(a, ref.detach(), dtype, what="swiglu")
    ref.backward(da.float())
    dg, du = ops.swiglu_bwd(gt.to(dev), u.to(dev), da.to(dev))
    _close(dg, gf.grad, dtype, scale=2.0, what="swiglu dg")
    _close(du, uf.grad, dtype, scale=2.0, what="swiglu du")
    for act, fn in [(1, F.gelu), (2, restate.quick_gelu)]:
        xf = gt.float().clone().requires_grad_(True)
        yref = fn(xf)
        _close(ops.act_fwd(gt.to(dev), act), yref.detach(), dtype, what=f"act{act}")
        yref.backward(da.float())
        _close(ops.act_bwd(gt.to(dev), da.to(dev), act), xf.grad, dtype, scale=2.0, what=f"act{act} bwd")
    _close(ops.add(gt.to(dev), u.to(dev)), gt.float() + u.float(), dtype, what="add")
    row = _rand((352,), dtype, g)
    _close(ops.add(gt.to(dev), row.to(dev), period=352), gt.float() + row.float(), dtype, what="add bcast")
    h16 = torch.randn(1000).half()
    assert torch.equal(ops.cast(h16.to(dev), dtype).cpu(), h16.to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding(dev, dtype):
    g = torch.Generator().manual_seed(17)
    V, D = 107, 128
    table = _rand((V, D), dtype, g)
    ids = torch.randint(0, V, (40,), generator=g)
    ids[5] = ids[9] = ids[33] = 7  # duplicates
    ids[2] = 0                      # padding_idx row
    out = ops.embedding_fwd(table.to(dev), ids.to(dev))
    assert torch.equal(out.cpu(), table[ids])
    dout = _rand((40, D), dtype, g)
    dt0 = _rand((V, D), dtype, g)
    dtab = dt0.to(dev).clone()
    ops.embedding_bwd_(dtab, dout.to(dev), ids.to(dev), padding_idx=0)
    ref = dt0.float().clone()
    for t in range(40):
        if ids[t] != 0:
            ref[ids[t]] += dout[t].float()
    _close(dtab, ref, dtype, scale=2.0, what="embedding bwd")


@pytest.mark.parametrize("dtype", DTYPES)
def test_im2col_conv1d(dev, dtype):
    """Conv1d == im2col + GEMM for the three geometries on the path: Whisper conv1 (k3,p1,
    channels-first input), conv2 (k3,s2,p1, channels-last input) and project_* (k>stride)."""
    g = torch.Generator().manual_seed(23)
    Bn = 2
    # channels-first input [B, C, T]
    C_, T, O = 10, 40, 24
    x = _rand((Bn, C_, T), dtype, g)
    W, b = _rand((O, C_, 3), dtype, g, 0.2), _rand((O,), dtype, g)
    cols, Lout = ops.im2col1d(x.to(dev), Bn, C_, T, 3, 1, 1, C_ * T, T, 1)
    Wp = torch.zeros((O, cols.shape[1]), dtype=dtype)
    Wp[:, : C_ * 3] = W.reshape(O, -1)
    y = ops.linear_fwd(cols, Wp.to(dev), bias=b.to(dev))
    ref = F.conv1d(x.float(), W.float(), b.float(), padding=1).transpose(1, 2).reshape(Bn * Lout, O)
    _close(y, ref, dtype, scale=2.0, what="conv k3 p1")
    # channels-last input [B, T, C], stride 2
    xl = _rand((Bn, T, C_), dtype, g)
    cols, Lout = ops.im2col1d(xl.to(dev), Bn, C_, T, 3, 2, 1, T * C_, 1, C_)
    y = ops.linear_fwd(cols, Wp.to(dev), bias=b.to(dev))
    ref = F.conv1d(xl.float().transpose(1, 2), W.float(), b.float(), stride=2, padding=1).transpose(1, 2).reshape(Bn * Lout, O)
    _close(y, ref, dtype, scale=2.0, what="conv k3 s2 p1")
    # overlapping windows k=6 s=5 (project_image-like) + adjoint
    W6 = _rand((O, C_, 6), dtype, g, 0.2)
    cols, Lout = ops.im2col1d(xl.to(dev), Bn, C_, T, 6, 5, 0, T * C_, 1, C_)
    assert Lout == (T - 6) // 5 + 1
    Wp6 = torch.zeros((O, cols.shape[1]), dtype=dtype)
    Wp6[:, : C_ * 6] = W6.reshape(O, -1)
    y = ops.linear_fwd(cols, Wp6.to(dev))
    xr = xl.float().requires_grad_(True)
    ref = F.conv1d(xr.transpose(1, 2), W6.float(), stride=5).transpose(1, 2).reshape(Bn * Lout, O)
    _close(y, ref.detach(), dtype, scale=2.0, what="conv k6 s5")
    dy = _rand((Bn * Lout, O), dtype, g)
    ref.backward(dy.float())
    dcols = ops.linear_dx(dy.to(dev), Wp6.to(dev))
    dx = ops.col2im1d(dcols, Bn, C_, T, 6, 5, 0, Lout, T * C_, 1, C_, (Bn, T, C_))
    _close(dx, xr.grad, dtype, scale=2.0, what="col2im")


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify(dev, dtype):
    g = torch.Generator().manual_seed(29)
    Bn, P, Hh, O = 2, 14, 56, 32
    img = _rand((Bn, 3, Hh, Hh), dtype, g)
    W = _rand((O, 3, P, P), dtype, g, 0.05)
    cols = ops.patchify(img.to(dev), P)
    assert cols.shape == (Bn * 16, 592) and (cols[:, 588:] == 0).all()
    Wp = torch.zeros((O, 592), dtype=dtype)
    Wp[:, :588] = W.reshape(O, -1)
    y = ops.linear_fwd(cols, Wp.to(dev))
    ref = F.conv2d(img.float(), W.float(), stride=P).flatten(2).transpose(1, 2).reshape(Bn * 16, O)
    _close(y, ref, dtype, scale=2.0, what="patch embed")
    back = ops.unpatchify(cols, Bn, 3, Hh, Hh, P)
    assert torch.equal(back.cpu(), img)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Lq,Lk,causal", [(29, 29, True), (6, 109, False), (5, 1500, False), (3, 2100, False)])
def test_softmax_fwd_bwd(dev, dtype, Lq, Lk, causal):
    g = torch.Generator().manual_seed(Lq * Lk)
    Bn, H = 2, 3
    ld = (Lk + 7) // 8 * 8
    s = _rand((Bn, H, Lq, Lk), dtype, g, 2.0)
    kmask = torch.ones(Bn, Lk, dtype=torch.int32)
    if causal:
        kmask[1, -4:] = 0  # right padding on sample 1
    buf = torch.zeros((Bn, H, Lq, ld), dtype=dtype, device=dev)
    buf[..., :Lk] = s.to(dev)
    probs, _ = ops.softmax_fwd(buf, Bn * H, H, Lq, Lk, ld, kmask=kmask.to(dev), causal=causal)
    # reference masking semantics (modeling.py:205-214)
    sf = s.clone()
    if causal:
        m = restate.decoder_mask(kmask, Bn, Lq, dtype, torch.device("cpu"))
        sf = torch.max(sf + m, torch.tensor(torch.finfo(dtype).min, dtype=dtype))
    pref = torch.softmax(sf.float(), -1)
    _close(probs[..., :Lk], pref, dtype, what="softmax")
    # backward
    dP = _rand((Bn, H, Lq, Lk), dtype, g)
    pf = probs[..., :Lk].float().cpu()
    dS_ref = pf * (dP.float() - (dP.float() * pf).sum(-1, keepdim=True)) * 0.125
    dbuf = torch.zeros_like(buf)
    dbuf[..., :Lk] = dP.to(dev)
    ops.softmax_bwd_(probs, dbuf, Bn * H, Lq, Lk, ld, scale=0.125)
    _close(dbuf[..., :Lk], dS_ref, dtype, what="softmax bwd")


def test_softmax_dropout_statistics_and_consistency(dev):
    """dropout: kept fraction ~ 1-p, kept values scaled by 1/(1-p), and the backward kernel
    regenerates exactly the forward mask."""
    dtype = torch.float32
    Bn, H, Lq, Lk = 1, 4, 64, 2000
    g = torch.Generator().manual_seed(1)
    s = _rand((Bn, H, Lq, Lk), dtype, g).to(dev)
    p = 0.1
    probs, pd = ops.softmax_fwd(s, Bn * H, H, Lq, Lk, Lk, dropout_p=p, seed=1234, want_dropped=True)
    keep = pd != 0
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 5e-3, frac
    torch.testing.assert_close(pd[keep], probs[keep] / (1 - p), rtol=1e-6, atol=0)
    dP = _rand((Bn, H, Lq, Lk), dtype, g).to(dev)
    gref = torch.where(keep, dP / (1 - p), torch.zeros_like(dP))
    dS_ref = probs * (gref - (gref * probs).sum(-1, keepdim=True))
    dS = ops.softmax_bwd_(probs, dP.clone(), Bn * H, Lq, Lk, Lk, dropout_p=p, seed=1234)
    torch.testing.assert_close(dS, dS_ref, rtol=1e-4, atol=1e-7)
    # a different seed gives a different mask
    _, pd2 = ops.softmax_fwd(s, Bn * H, H, Lq, Lk, Lk, dropout_p=p, seed=99, want_dropped=True)
    assert ((pd2 != 0) != keep).any()


@pytest.mark.parametrize("dtype", DTYPES)
def test_cross_entropy(dev, dtype):
    g = torch.Generator().manual_seed(31)
    rows, V, ld = 58, 107, 112
    logits = _rand((rows, V), dtype, g, 2.0)
    labels = torch.randint(0, V, (rows,), generator=g)
    labels[::5] = -100
    buf = torch.full((rows, ld), 7.0, dtype=dtype, device=dev)  # garbage in the pad columns
    buf[:, :V] = logits.to(dev)
    row_loss, row_lse, sc = ops.cross_entropy(buf, labels.to(dev), V)
    lf = logits.float().requires_grad_(True)
    ref = F.cross_entropy(lf, labels)
    loss = (sc[0] / sc[1]).item()
    assert abs(loss - ref.item()) <= 2e-5 * abs(ref.item()) + 1e-6, (loss, ref.item())
    ref.backward()
    dl = ops.cross_entropy_bwd(buf, labels.to(dev), row_lse, sc, V, grad_scale=1.0)
    tol = 1e-6 if dtype == torch.float32 else 2e-4
    assert (dl[:, :V].float().cpu() - lf.grad).abs().max().item() <= tol
    assert (dl[:, V:] == 0).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_adamw(dev, dtype):
    g = torch.Generator().manual_seed(37)
    n = 5000
    w0 = torch.randn(n, generator=g)
    p_ref = torch.nn.Parameter(w0.clone())
    opt = torch.optim.AdamW([p_ref], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    param = w0.to(dtype).to(dev)
    master = w0.to(dev).clone()
    m = torch.zeros(n, device=dev)
    v = torch.zeros(n, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g).to(dtype)
        p_ref.grad = grad.float()
        opt.step()
        ops.adamw_(param, master, m, v, grad.to(dev), 3e-3, 0.9, 0.95, 1e-8, 0.1, step)
    torch.testing.assert_close(master.cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(param.cpu(), master.cpu().to(dtype))


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("Lq,Lk,causal,masked", [(144, 144, True, True), (257, 257, False, False),
                                                 (70, 1500, False, False), (300, 300, True, False),
                                                 (5, 5, True, False)])
@pytest.mark.parametrize("dtype", H16)
def test_flash_attention_fwd(dev, dtype, hd, Lq, Lk, causal, masked):
    """fused attention == softmax(scale QK^T + mask) V in fp32 on the same bf16 inputs; bound =
    bf16 rounding of P and of the output (rtol 8e-3 + 8e-3 abs at |o| <= ~1)."""
    g = torch.Generator().manual_seed(Lq * 31 + Lk + hd)
    Bn, H = 2, 3
    D = H * hd
    q, k, v = (_rand((Bn * L, D), dtype, g) for L in (Lq, Lk, Lk))
    kmask = torch.ones(Bn, Lk, dtype=torch.int32)
    if masked:
        kmask[1, -9:] = 0
    o = torch.zeros((Bn * Lq, D), dtype=dtype, device=dev)
    lse = torch.empty((Bn, H, Lq), dtype=torch.float32, device=dev)
    scale = hd ** -0.5
    ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), o, Bn, H, Lq, Lk, hd, D, Lq * D, D, Lk * D,
                       D, Lk * D, D, Lq * D, scale, kmask=kmask.to(dev) if masked else None,
                       causal=causal, lse=lse)
    qf = q.float().view(Bn, Lq, H, hd).transpose(1, 2)
    kf = k.float().view(Bn, Lk, H, hd).transpose(1, 2)
    vf = v.float().view(Bn, Lk, H, hd).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
    if masked:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(Bn * Lq, D)
    _close(o, ref, dtype, what="flash fwd")
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(s, -1), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("hd,Lq,Lk,causal,masked", [
    (128, 1024, 1024, True, True),       # whole 256-row blocks, padded tail
    (128, 1100, 1100, True, False),      # ragged last block: waves without rows still stage and count barriers
    (128, 1030, 1500, False, True),      # non-causal, Lk not a multiple of the 64-key tile
    (128, 1200, 1030, True, False),      # Lk < Lq: the first rows see no key at all (exact zeros, lse = -inf)
    (128, 1030, 1300, True, False),      # Lk > Lq: every row sees a prefix of Lk - Lq extra keys
    (64, 1500, 1500, False, False),      # head_dim 64 instantiation (Whisper's shape)
    (64, 1100, 1100, True, True),
])
@pytest.mark.parametrize("dtype", H16)
def test_flash_attention_fwd_eight_wave_kernel(dev, dtype, hd, Lq, Lk, causal, masked, monkeypatch):
    """flash_fwd8_kernel (256 query rows per workgroup, waves 4-7 one segment behind waves 0-3, P V of tile t - 1 in
    the matrix segment of tile t, 2-deep K / V rings) against fp32 softmax(scale QK^T + mask) V on the same inputs.
    B H = 6 (b, h) pairs x 5-6 blocks; the 4-wave kernel on the same inputs must agree to output rounding."""
    monkeypatch.setenv("MK_ATTN_FWD8_MIN", "1024")
    monkeypatch.setenv("MK_ATTN_FWD8_HD64", "1")
    g = torch.Generator().manual_seed(Lq * 7 + Lk + hd)
    Bn, H = 2, 3
    D = H * hd
    q, k, v = (_rand((Bn * L, D), dtype, g) for L in (Lq, Lk, Lk))
    kmask = torch.ones(Bn, Lk, dtype=torch.int32)
    if masked:
        kmask[1, -77:] = 0
        kmask[0, 100:130] = 0
    scale = hd ** -0.5

    def run():
        o = torch.full((Bn * Lq, D), 3.0, dtype=dtype, device=dev)
        lse = torch.full((Bn, H, Lq), 5.0, dtype=torch.float32, device=dev)
        ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), o, Bn, H, Lq, Lk, hd, D, Lq * D, D, Lk * D,
                           D, Lk * D, D, Lq * D, scale, kmask=kmask.to(dev) if masked else None,
                           causal=causal, lse=lse)
        torch.cuda.synchronize()
        return o, lse

    monkeypatch.setenv("MK_ATTN_FWD4X64", "0")
    o8, lse8 = run()
    exp464 = hd == 128 and bool(os.environ.get("MK_EXPERIMENTS"))      # (experiment builds only: build.py)
    if exp464:                   # the 4-wave x 64-row form (one wave per SIMD) of the same walk
        monkeypatch.setenv("MK_ATTN_FWD4X64", "1")
        o464, lse464 = run()
        monkeypatch.setenv("MK_ATTN_FWD4X64", "0")
    monkeypatch.setenv("MK_ATTN_FWD8_MIN", "0")
    o4, lse4 = run()
    qf = q.float().view(Bn, Lq, H, hd).transpose(1, 2)
    kf = k.float().view(Bn, Lk, H, hd).transpose(1, 2)
    vf = v.float().view(Bn, Lk, H, hd).transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
    if masked:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.where(torch.isnan(p), torch.zeros_like(p), p)        # rows without a visible key: zeros
    ref = (p @ vf).transpose(1, 2).reshape(Bn * Lq, D)
    _close(o8, ref, dtype, what="flash fwd8")
    _close(o4, ref, dtype, what="flash fwd (4-wave)")
    if exp464:
        _close(o464, ref, dtype, what="flash fwd 4x64")
        torch.testing.assert_close(lse464.cpu(), torch.logsumexp(s, -1), rtol=1e-4, atol=2e-4)
    ref_lse = torch.logsumexp(s, -1)
    torch.testing.assert_close(lse8.cpu(), ref_lse, rtol=1e-4, atol=2e-4)
    assert (o8.float() - o4.float()).abs().max().item() <= (2e-2 if dtype == torch.bfloat16 else 4e-3)


def test_flash_attention_fwd_lazy_rescale_branches_at_seq_2048(dev):
    """cdna_hip_programming.md rule 26: the lazy rescale (attention forward keeps a row's running
    maximum until it grows by more than 2^DEFER) is a data-dependent branch that bounded random
    data rarely takes late in a row -- so FORCE it: S = 2048, causal, hd = 128, and per row a key
    whose score spikes at a chosen tile (early, middle, the diagonal tile, the very last key),
    rows whose maximum creeps up by less than the threshold per tile, fully padded tails.  Every
    row of the full tensor is compared with the fp32 reference."""
    g = torch.Generator().manual_seed(2048)
    Bn, H, hd, S = 1, 2, 128, 2048
    D = H * hd
    q = torch.randn(Bn * S, D, generator=g) * 0.5
    k = torch.randn(Bn * S, D, generator=g) * 0.5
    v = torch.randn(Bn * S, D, generator=g)
    # row r (head 0) meets a huge key at position spike(r) <= r: aligned query / key directions
    for r, kp, gain in ((100, 3, 6.0), (700, 650, 8.0), (1300, 1299, 10.0), (2047, 2047, 12.0), (1500, 64, 5.0)):
        d = torch.randn(hd, generator=g)
        d = d / d.norm()
        q[r, :hd] = d * gain * 3.0
        k[kp, :hd] = d * gain * 3.0
    # creeping maxima: keys along one direction with slowly growing length (growth < 2^6 per tile)
    d = torch.randn(hd, generator=g)
    d = d / d.norm()
    q[1800:1832, hd:] = d * 4.0
    k[::64, hd:] = d[None, :] * torch.linspace(0.5, 12.0, S // 64)[:, None]
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    kmask = torch.ones(Bn, S, dtype=torch.int32)
    kmask[0, -200:] = 0
    o = torch.zeros((Bn * S, D), dtype=torch.bfloat16, device=dev)
    lse = torch.empty((Bn, H, S), dtype=torch.float32, device=dev)
    scale = hd ** -0.5
    ops.flash_attn_fwd(q.to(dev), k.to(dev), v.to(dev), o, Bn, H, S, S, hd, D, S * D, D, S * D, D, S * D, D, S * D,
                       scale, kmask=kmask.to(dev), causal=True, lse=lse)
    qf = q.float().view(Bn, S, H, hd).transpose(1, 2)
    kf = k.float().view(Bn, S, H, hd).transpose(1, 2)
    vf = v.float().view(Bn, S, H, hd).transpose(1, 2)
    s_ = qf @ kf.transpose(-1, -2) * scale
    i = torch.arange(S)[:, None]
    j = torch.arange(S)[None, :]
    s_ = s_.masked_fill(j > i, float("-inf")).masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    ref = (torch.softmax(s_, -1) @ vf).transpose(1, 2).reshape(Bn * S, D)
    valid = (kmask[0] != 0)                     # (padded QUERY rows still attend to valid keys: all rows finite)
    _close(o, ref, torch.bfloat16, what="flash fwd, forced rescale branches")
    torch.testing.assert_close(lse.cpu(), torch.logsumexp(s_, -1), rtol=1e-4, atol=2e-4)
    assert valid.sum() == S - 200


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("Lq,Lk,causal,masked", [(144, 144, True, True), (257, 257, False, False),
                                                 (130, 300, False, True), (300, 300, True, False)])
@pytest.mark.parametrize("dtype", H16)
def test_flash_attention_bwd(dev, dtype, hd, Lq, Lk, causal, masked):
    """fused backward (recompute) == autograd of softmax(scale QK^T + mask) V in fp32 on the same
    bf16 inputs.  bf16 P / dS operands => rtol 2e-2 of the gradient scale."""
    g = torch.Generator().manual_seed(Lq * 17 + Lk + hd)
    Bn, H = 2, 2
    D = H * hd
    q, k, v = (_rand((Bn * L, D), dtype, g, 0.7) for L in (Lq, Lk, Lk))
    do = _rand((Bn * Lq, D), dtype, g)
    kmask = torch.ones(Bn, Lk, dtype=torch.int32)
    if masked:
        kmask[1, -11:] = 0
    km_d = kmask.to(dev) if masked else None
    scale = hd ** -0.5
    qd, kd_, vd, dod = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    o = torch.zeros((Bn * Lq, D), dtype=dtype, device=dev)
    lse = torch.empty((Bn, H, Lq), dtype=torch.float32, device=dev)
    geo = (D, Lq * D, D, Lk * D, D, Lk * D, D, Lq * D)
    ops.flash_attn_fwd(qd, kd_, vd, o, Bn, H, Lq, Lk, hd, *geo, scale, kmask=km_d, causal=causal, lse=lse)
    dq, dk, dv = torch.zeros_like(qd), torch.zeros_like(kd_), torch.zeros_like(vd)
    ops.flash_attn_bwd(qd, kd_, vd, o, dod, lse, dq, dk, dv, Bn, H, Lq, Lk, hd, *geo, scale,
                       kmask=km_d, causal=causal)
    qf = q.float().view(Bn, Lq, H, hd).transpose(1, 2).requires_grad_(True)
    kf = k.float().view(Bn, Lk, H, hd).transpose(1, 2).requires_grad_(True)
    vf = v.float().view(Bn, Lk, H, hd).transpose(1, 2).requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Lq)[:, None]
        j = torch.arange(Lk)[None, :]
        s = s.masked_fill(j > i + (Lk - Lq), float("-inf"))
    if masked:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    out = torch.softmax(s, -1) @ vf
    out.backward(do.float().view(Bn, Lq, H, hd).transpose(1, 2))
    for name, got, ref, L in (("dq", dq, qf.grad, Lq), ("dk", dk, kf.grad, Lk), ("dv", dv, vf.grad, Lk)):
        ref2 = ref.transpose(1, 2).reshape(Bn * L, D)
        err = (got.float().cpu() - ref2).abs().max().item()
        lim = 2e-2 * ref2.abs().max().item() + 2e-3
        assert torch.isfinite(got).all() and err <= lim, (name, err, lim)


_SHORT_FWD_SNIPPET = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from macaw_llm_amd import ops
dev = torch.device("cuda:0")
out = {}
for S, causal, masked in [(144, True, True), (160, False, True), (129, True, False), (33, True, True), (5, True, False)]:
    g = torch.Generator().manual_seed(S)
    Bn, H, hd = 3, 3, 128
    D = H * hd
    q, k, v = ((torch.randn((Bn * S, D), generator=g) * 0.7).to(torch.bfloat16).to(dev) for _ in range(3))
    kmask = torch.ones(Bn, S, dtype=torch.int32)
    if masked:
        kmask[1, -min(11, S - 1):] = 0
    o = torch.zeros((Bn * S, D), dtype=torch.bfloat16, device=dev)
    lse = torch.empty((Bn, H, S), dtype=torch.float32, device=dev)
    geo = (D, S * D, D, S * D, D, S * D, D, S * D)
    ops.flash_attn_fwd(q, k, v, o, Bn, H, S, S, hd, *geo, hd ** -0.5, kmask=kmask.to(dev) if masked else None,
                       causal=causal, lse=lse)
    out[(S, causal, masked)] = (o.cpu(), lse.cpu())
torch.save(out, sys.argv[2])
"""


def test_flash_attention_short_forward_is_bit_identical_to_the_tiled_kernel(dev, tmp_path):
    """Lq == Lk <= 160, head_dim 128: the one-workgroup-per-(b, h) forward (whole-sequence K / V images) keeps
    flash_fwd_kernel's per-row arithmetic and key-block order, so o and lse are bit-identical to it -- the tiled
    kernel is selected in a second process with MK_ATTN_NO_SHORT_FWD=1 (the switch is read once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, extra in (("short", {}), ("tiled", {"MK_ATTN_NO_SHORT_FWD": "1"})):
        f = str(tmp_path / f"{tag}.pt")
        env = dict(os.environ, **extra)
        env.pop("MK_ATTN_NO_SHORT_FWD", None) if not extra else None
        r = subprocess.run([sys.executable, "-c", _SHORT_FWD_SNIPPET, root, f], env=env, capture_output=True, text=True,
                           timeout=240)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = torch.load(f)
    for key, (o, lse) in res["short"].items():
        o2, lse2 = res["tiled"][key]
        assert torch.isfinite(o.float()).all()
        assert torch.equal(o, o2), key
        assert torch.equal(lse, lse2), key


@pytest.mark.parametrize("S,causal,masked", [(144, True, True), (160, True, False), (160, False, True), (129, True, False),
                                             (128, False, False), (96, True, True), (33, True, True), (32, False, False),
                                             (5, True, False), (1, True, False), (150, False, False)])
@pytest.mark.parametrize("dtype", H16)
def test_flash_attention_bwd_short_sequences(dev, dtype, S, causal, masked):
    """Lq == Lk <= 160 at head_dim 128: ONE kernel per (b, h) runs the D = rowsum(dO * O) pass, dQ and dK / dV
    (csrc flash_bwd_short_kernel: whole-sequence LDS images with one swizzle for row and transposed reads, 32-row
    blocks rotated over the waves by (b + h), rows beyond S clamped) -- against fp32 autograd on the same 16-bit
    inputs; 3 x 3 (b, h) so that every rotation occurs, key-padding mask on one sample, a fully visible and a
    one-token sequence among the lengths."""
    g = torch.Generator().manual_seed(S * 31 + causal + 2 * masked)
    Bn, H, hd = 3, 3, 128
    D = H * hd
    q, k, v = (_rand((Bn * S, D), dtype, g, 0.7) for _ in range(3))
    do = _rand((Bn * S, D), dtype, g)
    kmask = torch.ones(Bn, S, dtype=torch.int32)
    if masked:
        kmask[1, -min(11, S - 1):] = 0
    km_d = kmask.to(dev) if masked else None
    scale = hd ** -0.5
    qd, kd_, vd, dod = q.to(dev), k.to(dev), v.to(dev), do.to(dev)
    o = torch.zeros((Bn * S, D), dtype=dtype, device=dev)
    lse = torch.empty((Bn, H, S), dtype=torch.float32, device=dev)
    geo = (D, S * D, D, S * D, D, S * D, D, S * D)
    ops.flash_attn_fwd(qd, kd_, vd, o, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal, lse=lse)
    nan = float("nan")
    dq, dk, dv = (torch.full_like(qd, nan) for _ in range(3))
    ops.flash_attn_bwd(qd, kd_, vd, o, dod, lse, dq, dk, dv, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal)
    qf = q.float().view(Bn, S, H, hd).transpose(1, 2).requires_grad_(True)
    kf = k.float().view(Bn, S, H, hd).transpose(1, 2).requires_grad_(True)
    vf = v.float().view(Bn, S, H, hd).transpose(1, 2).requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(S)[:, None]
        j = torch.arange(S)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    if masked:
        s = s.masked_fill(kmask[:, None, None, :] == 0, float("-inf"))
    out = torch.softmax(s, -1) @ vf
    out.backward(do.float().view(Bn, S, H, hd).transpose(1, 2))
    for name, got, ref in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        ref2 = ref.transpose(1, 2).reshape(Bn * S, D)
        assert torch.isfinite(got).all(), name + ": rows not written or non-finite"
        err = (got.float().cpu() - ref2).abs().max().item()
        lim = 2e-2 * ref2.abs().max().item() + 2e-3
        assert err <= lim, (name, err, lim)
    # the same call again: bit-identical (no cross-workgroup state, no atomics)
    dq2, dk2, dv2 = (torch.full_like(qd, nan) for _ in range(3))
    ops.flash_attn_bwd(qd, kd_, vd, o, dod, lse, dq2, dk2, dv2, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("S,causal,masked,fused_qkv", [(144, True, True, True), (160, True, False, False), (160, False, True, True),
                                                       (129, True, False, True), (96, True, True, False), (33, True, True, True),
                                                       (5, True, False, True), (1, True, False, False), (150, False, False, True)])
@pytest.mark.parametrize("dtype", H16)
def test_flash_attention_with_rope_inside_is_bit_identical_to_the_three_launches(dev, dtype, S, causal, masked, fused_qkv):
    """mk_flash_attn_rope_fwd / _bwd (RoPE applied to q and k on their way into the short-sequence kernels, dq / dk rotated
    back at the store) against mk_rope -> mk_flash_attn_fwd / _bwd -> mk_rope(inverse): o, lse, dq, dk, dv BIT-IDENTICAL
    (same rounding points: modeling.py:76-91 in the element type).  Positions are NOT arange (shifted per sample, as
    a left-padded batch has them), q | k | v both as three tensors and as column blocks of one [M, 3D] buffer, untouched
    input buffers, 3 x 3 (b, h) so that every wave rotation occurs."""
    g = torch.Generator().manual_seed(S * 17 + causal + 2 * masked)
    Bn, H, hd = 3, 3, 128
    D = H * hd
    cos, sin = restate.rotary_tables(hd, 256)
    cos, sin = cos.to(dtype).to(dev).contiguous(), sin.to(dtype).to(dev).contiguous()
    pos = (torch.arange(S)[None, :] + torch.tensor([0, 7, 91])[:, None]).reshape(-1).to(torch.int32).to(dev)
    if fused_qkv:
        qkv = _rand((Bn * S, 3 * D), dtype, g, 0.7).to(dev)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        ld = 3 * D
    else:
        q, k, v = (_rand((Bn * S, D), dtype, g, 0.7).to(dev) for _ in range(3))
        ld = D
    do = _rand((Bn * S, D), dtype, g).to(dev)
    kmask = torch.ones(Bn, S, dtype=torch.int32)
    if masked:
        kmask[1, -min(11, S - 1):] = 0
    km_d = kmask.to(dev) if masked else None
    scale = hd ** -0.5
    geo = (ld, S * ld, ld, S * ld, ld, S * ld, D, S * D)
    assert ops.flash_rope_ok(hd, S, S, cos, q)
    # ---- three launches
    if fused_qkv:
        rot = qkv.clone()
        ops.rope_(rot[:, :2 * D], cos, sin, pos, 2 * H, hd)
        qr, kr, vr = rot[:, :D], rot[:, D:2 * D], rot[:, 2 * D:]
    else:
        qr, kr, vr = q.clone(), k.clone(), v
        ops.rope_(qr, cos, sin, pos, H, hd)
        ops.rope_(kr, cos, sin, pos, H, hd)
    o_ref = torch.zeros((Bn * S, D), dtype=dtype, device=dev)
    lse_ref = torch.empty((Bn, H, S), dtype=torch.float32, device=dev)
    ops.flash_attn_fwd(qr, kr, vr, o_ref, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal, lse=lse_ref)
    nan = float("nan")
    if fused_qkv:
        dref = torch.full((Bn * S, 3 * D), nan, dtype=dtype, device=dev)
        dq_r, dk_r, dv_r = dref[:, :D], dref[:, D:2 * D], dref[:, 2 * D:]
    else:
        dq_r, dk_r, dv_r = (torch.full((Bn * S, D), nan, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd(qr, kr, vr, o_ref, do, lse_ref, dq_r, dk_r, dv_r, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal)
    if fused_qkv:
        ops.rope_(dref[:, :2 * D], cos, sin, pos, 2 * H, hd, inverse=True)
    else:
        ops.rope_(dq_r, cos, sin, pos, H, hd, inverse=True)
        ops.rope_(dk_r, cos, sin, pos, H, hd, inverse=True)
    # ---- one launch per pass
    q0, k0 = q.clone(), k.clone()
    o = torch.zeros((Bn * S, D), dtype=dtype, device=dev)
    lse = torch.empty((Bn, H, S), dtype=torch.float32, device=dev)
    ops.flash_attn_fwd(q, k, v, o, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal, lse=lse, rope=(cos, sin, pos))
    assert torch.equal(o, o_ref), (o.float() - o_ref.float()).abs().max().item()
    assert torch.equal(lse, lse_ref)
    if fused_qkv:
        dgot = torch.full((Bn * S, 3 * D), nan, dtype=dtype, device=dev)
        dq, dk, dv = dgot[:, :D], dgot[:, D:2 * D], dgot[:, 2 * D:]
    else:
        dq, dk, dv = (torch.full((Bn * S, D), nan, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd(q, k, v, o, do, lse, dq, dk, dv, Bn, H, S, S, hd, *geo, scale, kmask=km_d, causal=causal,
                       rope=(cos, sin, pos))
    assert torch.equal(q, q0) and torch.equal(k, k0), "the fused kernels must leave q and k unrotated in HBM"
    for name, got, ref in (("dq", dq, dq_r), ("dk", dk, dk_r), ("dv", dv, dv_r)):
        assert torch.isfinite(got).all(), name
        assert torch.equal(got, ref), (name, (got.float() - ref.float()).abs().max().item())
    # ---- the training step's form: q, k rotated by mk_rope, only dq / dk rotated back inside the kernel
    if fused_qkv:
        dgot2 = torch.full((Bn * S, 3 * D), nan, dtype=dtype, device=dev)
        dq2, dk2, dv2 = dgot2[:, :D], dgot2[:, D:2 * D], dgot2[:, 2 * D:]
    else:
        dq2, dk2, dv2 = (torch.full((Bn * S, D), nan, dtype=dtype, device=dev) for _ in range(3))
    ops.flash_attn_bwd(qr, kr, vr, o_ref, do, lse_ref, dq2, dk2, dv2, Bn, H, S, S, hd, *geo, scale, kmask=km_d,
                       causal=causal, rope=(cos, sin, pos), qk_rotated=True)
    for name, got, ref in (("dq", dq2, dq_r), ("dk", dk2, dk_r), ("dv", dv2, dv_r)):
        assert torch.equal(got, ref), (name + " (q, k rotated)", (got.float() - ref.float()).abs().max().item())


def test_flash_attention_rope_entry_points_refuse_what_the_short_kernels_do_not_cover(dev):
    """outside hd == 128, Lq == Lk <= 160 the rope entry points return MK_ERR_UNSUPPORTED (the caller then launches
    mk_rope itself: engine.LlamaLayerFn does, ops.flash_rope_ok is its gate) -- never a silently unrotated result"""
    dtype = torch.bfloat16
    Bn, H, hd, S = 1, 2, 128, 192
    D = H * hd
    cos, sin = restate.rotary_tables(hd, 256)
    cos, sin = cos.to(dtype).to(dev).contiguous(), sin.to(dtype).to(dev).contiguous()
    pos = torch.arange(S, dtype=torch.int32, device=dev)
    q = torch.zeros((S, D), dtype=dtype, device=dev)
    o = torch.zeros_like(q)
    assert not ops.flash_rope_ok(hd, S, S, cos, q)
    with pytest.raises(ops.MacawHipError):
        ops.flash_attn_fwd(q, q, q, o, Bn, H, S, S, hd, D, S * D, D, S * D, D, S * D, D, S * D, 1.0, causal=True,
                           rope=(cos, sin, pos))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cfg", [19, 22])
@pytest.mark.parametrize("M,N,K", [(32, 4096, 4096), (17, 1000, 1024), (24, 12288, 4096), (32, 250, 192),
                                   (29, 4096, 11008), (32, 77, 64)])
def test_gemm_skinny_17_to_32_rows_every_kernel(dev, dtype, cfg, M, N, K):
    """17 ... 32 token rows: the two-tile 16-row kernel (19) and the pipelined 32 x 32 kernel (22) forced in
    turn (mk_gemm picks by N) -- ragged N (clamped weight rows), K blocks fewer than the waves (nkb = 1, 3), every epilogue
    option -- against fp32 math on the same inputs; the two must agree with each other within one
    rounding of the result (same products, different fp32 summation order)."""
    from macaw_llm_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = _rand((M, K), dtype, g)
    W = (_rand((N, K), dtype, g).float() * 0.05).to(dtype)
    b = _rand((N,), dtype, g)
    r = _rand((M, N), dtype, g)
    xd, Wd = x.to(dev), W.to(dev)
    ref = x.float() @ W.float().t()
    try:
        lib.mk_gemm_set_cfg(cfg)
        y = ops.linear_fwd(xd, Wd)
        y2 = ops.linear_fwd(xd, Wd, bias=b.to(dev), act=2, residual=r.to(dev))
        ldc = (N + 63) // 64 * 64
        buf = torch.full((M, ldc), float("nan"), dtype=dtype, device=dev)
        ops.gemm_raw(xd, Wd, buf, M, N, K, K, K, ldc)
        c0 = _rand((M, N), dtype, g)
        cd = c0.to(dev).clone()
        ops.gemm_raw(xd, Wd, cd, M, N, K, K, K, N, accumulate=True, alpha=0.5)
        lib.mk_gemm_set_cfg(19)
        y19 = ops.linear_fwd(xd, Wd)
    finally:
        lib.mk_gemm_set_cfg(-1)
    _close(y, ref, dtype, scale=math.sqrt(K) * 0.05, what=f"skinny32 cfg {cfg} plain")
    pre = ref + b.float()[None]
    want = pre * torch.sigmoid(1.702 * pre) + r.float()
    _close(y2, want, dtype, scale=math.sqrt(K) * 0.05 + 1.0, what=f"skinny32 cfg {cfg} bias+quick_gelu+residual")
    _close(cd, 0.5 * ref + c0.float(), dtype, scale=math.sqrt(K) * 0.05 + 1.0, what=f"skinny32 cfg {cfg} accumulate")
    assert torch.equal(buf[:, :N], y) and torch.isnan(buf[:, N:].float()).all()       # pitched C, pad untouched
    ulp = 2.0 ** (-7 if dtype == torch.bfloat16 else -10)
    assert (y.float() - y19.float()).abs().max().item() <= ulp * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (8, 22016, 4096), (32, 4096, 11008), (5, 107, 128),
                                   (2, 32007, 4096), (31, 250, 192), (16, 4096, 11008), (7, 4000, 1024),
                                   (13, 520, 704), (17, 4096, 4096), (24, 1000, 1024)])
def test_gemm_skinny_decode_rows(dev, M, N, K):
    """M <= 32 (one decode position per sample): the weight-streaming kernel (W rows as the MFMA M
    dimension, split-K across the waves of a workgroup) with every epilogue option, against fp32
    math on the same bf16 inputs; must agree with the tile kernel within bf16 rounding."""
    g = torch.Generator().manual_seed(M * 7 + N)
    x = _rand((M, K), torch.bfloat16, g)
    W = (_rand((N, K), torch.bfloat16, g).float() * 0.05).to(torch.bfloat16)
    b = _rand((N,), torch.bfloat16, g)
    r = _rand((M, N), torch.bfloat16, g)
    xd, Wd = x.to(dev), W.to(dev)
    ref = x.float() @ W.float().t()
    y = ops.linear_fwd(xd, Wd)
    _close(y, ref, torch.bfloat16, scale=math.sqrt(K) * 0.05, what="skinny plain")
    y = ops.linear_fwd(xd, Wd, bias=b.to(dev), act=2, residual=r.to(dev))
    pre = ref + b.float()[None]
    want = pre * torch.sigmoid(1.702 * pre) + r.float()
    _close(y, want, torch.bfloat16, scale=math.sqrt(K) * 0.05 + 1.0, what="skinny bias+quick_gelu+residual")
    c0 = _rand((M, N), torch.bfloat16, g)
    cd = c0.to(dev).clone()
    ops.gemm_raw(xd, Wd, cd, M, N, K, K, K, N, accumulate=True, alpha=0.5)
    _close(cd, 0.5 * ref + c0.float(), torch.bfloat16, scale=math.sqrt(K) * 0.05 + 1.0, what="skinny accumulate")
    # pitched output (the logits buffer is pitched) and agreement with the tile kernel
    ldc = (N + 63) // 64 * 64
    buf = torch.zeros((M, ldc), dtype=torch.bfloat16, device=dev)
    ops.gemm_raw(xd, Wd, buf, M, N, K, K, K, ldc)
    assert torch.equal(buf[:, :N], ops.linear_fwd(xd, Wd)) and not buf[:, N:].any()


def test_adamw_multi_tensor_is_bit_identical_to_per_tensor(dev):
    """one mk_adamw_multi launch over a mixed list (big / tiny / ragged sizes) == one mk_adamw launch
    per tensor, bit for bit, over several steps"""
    from macaw_llm_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(300, 1000), (4096,), (7,), (33, 129), (70000,), (64, 512)]
    base = [_rand(s, torch.bfloat16, g) for s in shapes]

    def make():
        return [torch.nn.Parameter(b.clone().to(dev)) for b in base]

    pa, pb = make(), make()
    oa = FusedAdamW(pa, lr=1e-2, weight_decay=0.1)
    ob = FusedAdamW(pb, lr=1e-2, weight_decay=0.1)
    for step in range(3):
        grads = [_rand(tuple(p.shape), torch.bfloat16, g).to(dev) for p in pa]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()                                   # multi-tensor
        ob.step_count += 1
        for q in pb:
            ob.step_param(q)                        # one launch per tensor
        for p, q in zip(pa, pb):
            assert torch.equal(p.data, q.data), (step, tuple(p.shape))
            assert torch.equal(oa.state[p][1], ob.state[q][1]) and torch.equal(oa.state[p][2], ob.state[q][2])


def test_adamw_grid_cap_changes_nothing_but_the_grid(dev):
    """mk_adamw_set_max_blocks (the persistent / confined form of the per-shard update, profiles/r06_local_overlap_confined.txt):
    every element is updated exactly once whatever the cap -- bit-identical parameters and moments for caps 1, 7, 64 and
    the default grid on a ragged length; the call returns the previous cap and 0 restores the default."""
    from macaw_llm_amd import lib as _L
    g = torch.Generator().manual_seed(91)
    n = 3 * 1024 * 1024 + 13 * 8
    w0 = _rand((n,), torch.bfloat16, g)
    grad = _rand((n,), torch.bfloat16, g).to(dev)
    lib = _L.load()

    def run(cap):
        prev = lib.mk_adamw_set_max_blocks(cap)
        assert prev == 0
        try:
            param = w0.clone().to(dev)
            master, m, v = param.float(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            for step in (1, 2):
                ops.adamw_(param, master, m, v, grad, 1e-2, 0.9, 0.999, 1e-8, 0.01, step)
            torch.cuda.synchronize()
        finally:
            assert lib.mk_adamw_set_max_blocks(0) == cap
        return param, master, m, v

    ref = run(0)
    for cap in (1, 7, 64):
        got = run(cap)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), cap


def test_adamw_unaligned_slice_and_param_groups(dev):
    """a parameter that is an odd-offset slice of a fused buffer (not 16-byte aligned)
    used to hit MK_ERR_UNSUPPORTED in the 'fallback'; it now goes through an aligned staging copy
    and must equal the aligned update bit for bit.  A strided view is rejected loudly."""
    from macaw_llm_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(21)
    vals, grads = _rand((1003,), torch.bfloat16, g), _rand((1003,), torch.bfloat16, g)
    buf = torch.zeros(1100, dtype=torch.bfloat16, device=dev)
    pa = torch.nn.Parameter(torch.empty(0, device=dev))
    pa.data = buf[3:1006]                      # byte offset 6: unaligned
    pa.data.copy_(vals.to(dev))
    pb = torch.nn.Parameter(vals.to(dev).clone())
    oa, ob = FusedAdamW([pa], lr=1e-2, weight_decay=0.1), FusedAdamW([pb], lr=1e-2, weight_decay=0.1)
    for _ in range(3):
        gbuf = torch.zeros(1100, dtype=torch.bfloat16, device=dev)
        gbuf[5:1008] = grads.to(dev)
        pa.grad, pb.grad = gbuf[5:1008], grads.to(dev).clone()
        oa.step()
        ob.step()
    assert torch.equal(pa.data, pb.data)
    assert buf[:3].abs().sum() == 0 and buf[1006:].abs().sum() == 0      # neighbours untouched
    ob.param_groups[0]["lr"] = 5e-3
    assert ob.lr == 5e-3
    ps = torch.nn.Parameter(torch.empty(0, device=dev))
    ps.data = torch.zeros(64, 8, dtype=torch.bfloat16, device=dev)[:, :4]
    ps.grad = torch.zeros(64, 4, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        FusedAdamW([ps]).step()


@pytest.mark.gpu
@pytest.mark.parametrize("hd,H,B,T", [(128, 32, 1, 173), (128, 4, 3, 1), (64, 8, 2, 64), (32, 4, 2, 333), (16, 4, 3, 40)])
def test_decode_attn_and_kv_append_with_device_position(hd, H, B, T):
    """mk_kv_append / mk_decode_attn read the position from device memory: one query row per
    (sample, head) against the first T cached keys, fp32 reference"""
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(hd + T)
    D, Tmax = H * hd, T + 7
    cache = torch.zeros((B, Tmax, 2 * D), dtype=torch.bfloat16, device=dev)
    cache[:, : T - 1] = torch.randn((B, T - 1, 2 * D), device=dev).to(torch.bfloat16)
    qkv = torch.randn((B, 3 * D), device=dev).to(torch.bfloat16)
    t_dev = torch.tensor([T - 1], dtype=torch.int32, device=dev)
    before = cache.clone()
    ops.kv_append(qkv, cache, 2 * D, B, 3 * D, Tmax * 2 * D, 2 * D, t_dev, Tmax, src_off=D)
    assert torch.equal(cache[:, T - 1], qkv[:, D:])
    before[:, T - 1] = qkv[:, D:]
    assert torch.equal(cache, before)                     # nothing else touched
    out = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
    scale = 1.0 / hd ** 0.5
    ops.decode_attn(qkv, cache, cache, out, t_dev, 1, Tmax, B, H, hd, 3 * D, 2 * D, Tmax * 2 * D, 2 * D,
                    Tmax * 2 * D, D, scale, v_off=D)
    q = qkv[:, :D].float().view(B, H, 1, hd)
    k = cache[:, :T, :D].float().view(B, T, H, hd).permute(0, 2, 1, 3)
    v = cache[:, :T, D:].float().view(B, T, H, hd).permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).reshape(B, D)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    # the position really is read at execution time
    t_dev.fill_(0)
    ops.decode_attn(qkv, cache, cache, out, t_dev, 1, Tmax, B, H, hd, 3 * D, 2 * D, Tmax * 2 * D, 2 * D,
                    Tmax * 2 * D, D, scale, v_off=D)
    assert torch.allclose(out.float(), cache[:, 0, D:].float(), atol=1e-6)   # one key: output = its value


@pytest.mark.gpu
@pytest.mark.parametrize("hd,H,B,T", [(128, 32, 1, 173), (128, 4, 2, 1), (64, 8, 2, 65), (32, 4, 2, 300), (16, 4, 3, 40),
                                      # B x H >= 512: the four-heads-per-workgroup kernel (first position, one
                                      # trip, several trips with a ragged last one)
                                      (128, 32, 16, 1), (128, 32, 32, 150), (128, 32, 17, 139), (128, 64, 8, 7)])
def test_decode_step_attn_equals_rope_append_attention(hd, H, B, T):
    """mk_decode_step_attn = mk_rope (q and k heads) + cache append + attention over keys 0 ... p with
    p read from device memory: rotated key / value rows bit-identical to the separate kernels, output
    against an fp32 reference"""
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3 * hd + T)
    D, Tmax, p = H * hd, T + 5, T - 1
    cache = torch.zeros((B, Tmax, 2 * D), dtype=torch.bfloat16, device=dev)
    cache[:, :p] = torch.randn((B, p, 2 * D), device=dev).to(torch.bfloat16)
    qkv = torch.randn((B, 3 * D), device=dev).to(torch.bfloat16)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.cat((torch.outer(torch.arange(Tmax).float(), inv),) * 2, dim=-1)
    cos, sin = ang.cos().to(dev).to(torch.bfloat16), ang.sin().to(dev).to(torch.bfloat16)
    t_dev = torch.tensor([p], dtype=torch.int32, device=dev)
    # separate kernels
    ref_qkv = qkv.clone()
    pos = torch.full((B,), p, dtype=torch.int32, device=dev)
    ops.rope_(ref_qkv[:, :2 * D], cos, sin, pos, 2 * H, hd)
    want_cache = cache.clone()
    want_cache[:, p] = ref_qkv[:, D:]
    out = torch.empty((B, D), dtype=torch.bfloat16, device=dev)
    scale = 1.0 / hd ** 0.5
    ops.decode_step_attn(qkv, qkv, qkv, 3 * D, cos, sin, cache, t_dev, Tmax, B, H, hd, out, scale,
                         k_off=D, v_off=2 * D)
    assert torch.equal(cache, want_cache)
    q = ref_qkv[:, :D].float().view(B, H, 1, hd)
    k = want_cache[:, :T, :D].float().view(B, T, H, hd).permute(0, 2, 1, 3)
    v = want_cache[:, :T, D:].float().view(B, T, H, hd).permute(0, 2, 1, 3)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).reshape(B, D)
    err = (out.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(1, 12288, 4096), (3, 4096, 4096), (16, 520, 704), (4, 32007, 4096), (1, 4096, 11008),
                                   (1, 1001, 704), (1, 32007, 5120), (1, 5120, 13824)])
def test_decode_linear_prologues_match_the_separate_kernels(M, N, K):
    """mk_decode_linear with the RMSNorm / SwiGLU prologue against rmsnorm_fwd / swiglu2d_fwd followed
    by the plain linear (same rounding points: the results may differ only through the order of the
    fp32 sum of squares), and against fp32 math"""
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + N + K)
    x = _rand((M, K), torch.bfloat16, g).to(dev)
    W = (_rand((N, K), torch.bfloat16, g).float() * 0.05).to(torch.bfloat16).to(dev)
    res = _rand((M, N), torch.bfloat16, g).to(dev)
    nw = (1.0 + 0.1 * _rand((K,), torch.float32, g)).to(torch.bfloat16).to(dev)
    # plain
    y0 = ops.decode_linear(x, W, residual=res)
    _close(y0, x.float().cpu() @ W.float().cpu().t() + res.float().cpu(), torch.bfloat16, scale=math.sqrt(K) * 0.05 + 1.0,
           what="decode_linear plain")
    # RMSNorm prologue
    _, yn, _ = ops.rmsnorm_fwd(x, nw, 1e-6)
    want = ops.linear_fwd(yn, W)
    got = ops.decode_linear(x, W, 1, nw, 1e-6)
    d = (got.float() - want.float()).abs().max().item()
    assert d <= 0.02 * want.float().abs().max().item() + 1e-3, d
    assert (got == want).float().mean().item() > 0.98           # a differing rstd ulp flips few roundings
    # SwiGLU prologue: x2 = [gate | up]
    gu = _rand((M, 2 * K), torch.bfloat16, g).to(dev)
    a = ops.swiglu2d_fwd(gu, K)
    want = ops.linear_fwd(a, W, residual=res)
    got = ops.decode_linear(gu, W, 2, residual=res)
    assert torch.equal(got, want)


@pytest.mark.gpu
def test_decode_emit_argmax_pad_eos_and_step_state():
    """mk_decode_emit: first argmax per row, pad for finished samples, eos marks a sample finished,
    output column and position advance exactly once per launch"""
    from macaw_llm_amd import ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    B, V, ld = 5, 32007, 32064
    logits = torch.randn((B, ld), device=dev).to(torch.bfloat16)
    logits[:, V:] = 100.0                                   # pad columns must be ignored
    logits[1, 777] = logits[1, 20001] = 50.0                # tie: lowest index
    logits[2, 9] = 60.0                                     # will be eos
    tok = torch.zeros(B, dtype=torch.long, device=dev)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    done[3] = True
    out = torch.full((B, 4), -1, dtype=torch.long, device=dev)
    state = torch.tensor([144, 1, 0, 0], dtype=torch.int32, device=dev)
    ops.decode_emit(logits, V, 106, 9, tok, done, out, state)
    want = logits[:, :V].float().argmax(1)
    want[1] = 777
    want[3] = 106
    assert torch.equal(out[:, 1], want) and torch.equal(tok, want)
    assert torch.equal(out[:, 0], torch.full((B,), -1, device=dev)) and torch.equal(out[:, 2:], torch.full((B, 2), -1, device=dev))
    assert done.tolist() == [False, False, True, True, False]
    assert state.tolist() == [145, 2, 0, 0]
    ops.decode_emit(logits, V, 106, 9, tok, done, out, state)
    assert out[2, 2].item() == 106 and state.tolist() == [146, 3, 0, 0]
