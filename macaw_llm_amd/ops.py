"""Tensor-level wrappers over the C ABI (ctypes).  PyTorch is used only for
device memory and the current HIP stream; every arithmetic op below runs in a
hand-written gfx950 kernel from csrc/.

All wrappers require CUDA(HIP) tensors and raise MacawHipError otherwise —
there is no eager / CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import lib as _L
from .lib import GemmDesc, MacawHipError, MK_BF16, MK_F16, MK_F32, MK_FP8

_DT = {torch.float32: MK_F32, torch.bfloat16: MK_BF16, torch.float16: MK_F16}
_null = None


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MacawHipError(f"unsupported dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise MacawHipError("macaw_llm_amd ops need device tensors (HIP); no CPU fallback exists")
    return t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


def _rowmajor(t: torch.Tensor) -> int:
    """leading dimension of a 2-D row-major (possibly pitched) tensor"""
    if t.dim() != 2 or (t.size(1) > 1 and t.stride(1) != 1):
        raise MacawHipError(f"expected row-major 2-D tensor, got {tuple(t.shape)} {t.stride()}")
    return t.stride(0) if t.size(0) > 1 else max(t.stride(0), t.size(1))


# ------------------------------------------------------------------- GEMM --
_WS = {}
WS_BYTES = 72 << 20


def _workspace(device):
    """per-(device, stream) scratch for the GEMM stream-K tail (partials + counters)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None and torch.cuda.is_current_stream_capturing():
        # a graph capture runs on its own stream: use the device's existing scratch (the captured
        # kernels are serialised) instead of allocating + zero-filling one inside the graph
        for (dev_i, _), w in _WS.items():
            if dev_i == device.index:
                return w
    if ws is None:
        ws = _WS[key] = torch.empty(WS_BYTES, dtype=torch.uint8, device=device)
        # the first 4 KiB are the stream-K arrival counters: zero once, the kernels leave them zero
        fill_(ws[:4096].view(torch.float32), 0.0)
    return ws


def gemm_raw(A, B, Cc, M, N, K, lda, ldb, ldc, *, a_red=False, b_red=False, R=None, ldr=0,
             bias=None, bias_mode=0, act=0, accumulate=False, alpha=1.0, nb1=1, nb2=1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), a_off=0, b_off=0, c_off=0, r_off=0,
             flags=0):
    """Direct struct fill. *_off are element offsets added to the base pointers.
    flags: MK_GEMM_A_KPAD_ZERO (1) / MK_GEMM_B_KPAD_ZERO (2), see include/macaw_hip.h."""
    lib = _L.load()
    es = A.element_size()
    d = GemmDesc()
    d.A = _p(A) + a_off * es
    d.B = _p(B) + b_off * es
    d.C = _p(Cc) + c_off * es
    d.R = (_p(R) + r_off * es) if R is not None else None
    d.bias = _p(bias) if bias is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.a_red_major, d.b_red_major = int(a_red), int(b_red)
    d.nb1, d.nb2 = nb1, nb2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.sR1, d.sR2 = sR
    d.alpha = alpha
    d.bias_mode = bias_mode if bias is not None else 0
    d.act = act
    d.accumulate = int(accumulate)
    d.dtype = dt(A)
    d.flags = flags
    ws = _workspace(A.device)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    if B.dtype != A.dtype or Cc.dtype != A.dtype:
        raise MacawHipError("gemm: mixed dtypes")
    _L.check(lib.mk_gemm(C.byref(d), _st()), "mk_gemm")
    return Cc


# --------------------------------------------------------------------- fp8 --
def quantize_fp8(x):
    """per-tensor scaled OCP e4m3: returns (q uint8 tensor of x's shape, dequant scale f32[1] on
    the device).  x bf16 / f32, contiguous, numel % 8 == 0."""
    lib = _L.load()
    xc = x if x.is_contiguous() else x.contiguous()
    q = torch.empty(xc.shape, dtype=torch.uint8, device=x.device)
    ws = torch.empty(2, dtype=torch.float32, device=x.device)
    _L.check(lib.mk_fp8_quantize(_p(xc), xc.numel(), dt(xc), _p(q), _p(ws), _p(ws) + 4, _st()),
             "mk_fp8_quantize")
    return q, ws[1:2]


def quantize_fp8_rows(x, out=None):
    """per-ROW scaled OCP e4m3 of a row-major bf16 matrix [rows, cols] (pitched rows allowed):
    returns (q uint8 [rows, cols], scales f32 [rows]); q[r] = e4m3(x[r] * 448 / amax_r), scales[r] =
    amax_r / 448.  One scale per token (activations, gradients) / per output channel (weights)."""
    lib = _L.load()
    rows, cols = x.shape
    q = out if out is not None else torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    _L.check(lib.mk_fp8_quantize_rows(_p(x), rows, cols, _rowmajor(x), dt(x), _p(q), _rowmajor(q), _p(sc), _st()),
             "mk_fp8_quantize_rows")
    return q, sc


def quantize_fp8_cols_t(W):
    """per-COLUMN scaled e4m3 of W [rows, cols], written TRANSPOSED: returns (qt uint8 [cols, rows],
    scales f32 [cols]) -- W^T K-major for the fp8 grad-input GEMM dx = dy W (one scale per input
    channel)."""
    lib = _L.load()
    rows, cols = W.shape
    qt = torch.empty((cols, rows), dtype=torch.uint8, device=W.device)
    sc = torch.empty(cols, dtype=torch.float32, device=W.device)
    ws = torch.empty(cols, dtype=torch.float32, device=W.device)
    _L.check(lib.mk_fp8_quantize_cols_t(_p(W), rows, cols, _rowmajor(W), dt(W), _p(qt), rows, _p(sc), _p(ws), _st()),
             "mk_fp8_quantize_cols_t")
    return qt, sc


# fp8 copies of weights are made ONCE per optimizer step, not per call: (data_ptr, shape, kind) ->
# (weight version, torch version counter, q, scales).  The optimizer runtimes bump WEIGHT_VERSION after
# every update (bucketed.BucketedStep.finish, optim.FusedAdamW.step*); torch-side in-place edits
# (load_state_dict, manual .copy_) move the tensor's own version counter.
WEIGHT_VERSION = [0]
_W8 = {}


def bump_weight_version():
    WEIGHT_VERSION[0] += 1


def fp8_weight(W, transposed=False):
    """cached e4m3 copy of weight W [N, K]: rows scaled (q [N, K], scales [N]) for y = x W^T, or
    transposed / column scaled (qt [K, N], scales [K]) for dx = dy W"""
    key = (W.data_ptr(), tuple(W.shape), W.stride(0), transposed)
    ver = (WEIGHT_VERSION[0], W._version)
    ent = _W8.get(key)
    if ent is None or ent[0] != ver:
        q, sc = quantize_fp8_cols_t(W) if transposed else quantize_fp8_rows(W)
        ent = _W8[key] = (ver, q, sc)
    return ent[1], ent[2]


def clear_fp8_cache():
    _W8.clear()


def linear_fp8(xq, sx, wq, sw, bias=None, act=0, residual=None, out=None, accumulate=False):
    """out[M, N] (bf16) = act((xq . wq^T) * sx * sw + bias) + residual (+ out) with fp8 operands xq
    [M, K], wq [N, K] (uint8 e4m3 bytes, K-major) and their device de-quantisation scales: f32[1]
    per-tensor scalars, or f32[M] / f32[N] per-row vectors (quantize_fp8_rows / fp8_weight)."""
    lib = _L.load()
    M, K = xq.shape
    N = wq.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device)
    vec = sx.numel() > 1 or sw.numel() > 1
    if vec and (sx.numel() != M or sw.numel() != N):
        raise MacawHipError(f"linear_fp8: scale vectors {sx.numel()} / {sw.numel()} for a {M} x {N} output")
    d = GemmDesc()
    d.A, d.B, d.C = _p(xq), _p(wq), _p(out)
    d.R = _p(residual) if residual is not None else None
    d.bias = _p(bias) if bias is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = _rowmajor(xq), _rowmajor(wq), _rowmajor(out)
    d.ldr = _rowmajor(residual) if residual is not None else 0
    d.nb1 = d.nb2 = 1
    d.alpha = 1.0
    d.bias_mode = 1 if bias is not None else 0
    d.act = act
    d.accumulate = int(accumulate)
    d.dtype = MK_FP8
    d.flags = 4 if vec else 0            # MK_GEMM_SCALE_VEC
    ws = _workspace(xq.device)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel()
    d.scale_a, d.scale_b = _p(sx), _p(sw)
    _L.check(lib.mk_gemm(C.byref(d), _st()), "mk_gemm(fp8)")
    return out


def linear_fwd(x, W, bias=None, act=0, residual=None, out=None, alpha=1.0):
    """y[M,N] = act(alpha * x[M,K] @ W[N,K]^T + bias) + residual"""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    gemm_raw(x, W, out, M, N, K, _rowmajor(x), _rowmajor(W), _rowmajor(out), bias=bias,
             bias_mode=1, act=act, R=residual, ldr=_rowmajor(residual) if residual is not None else 0,
             alpha=alpha)
    return out


def linear_dx(dy, W, out=None, accumulate=False, residual=None, dy_pad_zero=False):
    """dx[M,K] = dy[M,N] @ W[N,K]  (W consumed red-major: no transpose pass).
    dy_pad_zero: dy is a column view of a pitched buffer whose columns [N, pad64(N)) are zero
    (d(logits) for V = 32007): the reduction may then run over the padded width on the MFMA tile
    kernels instead of the generic edge kernel."""
    M, N = dy.shape
    K = W.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    gemm_raw(dy, W, out, M, K, N, _rowmajor(dy), _rowmajor(W), _rowmajor(out), b_red=True,
             accumulate=accumulate, R=residual,
             ldr=_rowmajor(residual) if residual is not None else 0,
             flags=1 if dy_pad_zero else 0)
    return out


# Weight-gradient destinations (bucketed.BucketedStep): (data_ptr, numel) of a weight (or of a fused
# q|k|v / gate|up view) -> the view of the flat gradient bucket its dW is written to, so that
# the grad-weight GEMM stores straight into the buffer the collective reads (no copy, no per-step
# gradient allocation).  Empty = every dW gets a fresh tensor (the default).
GRAD_DST = {}
GRAD_DST_OWNER = [None]     # the BucketedStep whose table is installed (begin() ... finish())
GRAD_DST_TAKEN = set()      # keys handed out in the current backward: a direct store OVERWRITES its slot, so a
                            # parameter whose gradient is produced twice (tied / shared weights, a module called
                            # twice) gets the slot once and a fresh tensor afterwards (autograd sums the two)

# data_ptr of every storage that a BucketedStep owns (parameter buckets).  Parameters living there
# must never be re-homed by the lazy q|k|v / gate|up fusion of modeling.py: the optimizer and the
# collectives keep working on the bucket, the model would compute with the moved copy.
PINNED_STORAGE = set()


def is_pinned(t) -> bool:
    """True if tensor `t` views a storage owned by a BucketedStep"""
    return bool(PINNED_STORAGE) and t.untyped_storage().data_ptr() in PINNED_STORAGE


def grad_dst(w):
    """registered destination for the gradient of parameter `w` (same shape), else None"""
    if not GRAD_DST or w is None:
        return None
    key = (w.data_ptr(), w.numel())
    d = GRAD_DST.get(key)
    if d is not None and key in GRAD_DST_TAKEN:
        return None
    if d is not None and d.dtype == w.dtype and d.numel() == w.numel() and (d.shape == w.shape or w.is_contiguous()):
        # a FRESH view object every time: autograd's AccumulateGrad takes a returned gradient as p.grad
        # without copying only if nobody else references that tensor object -- handing out the
        # dictionary's own tensor made it clone every such gradient (and the hook copy it back):
        # ~150 hidden device copies + ~150 copy-backs per step in round 2's "direct" path
        GRAD_DST_TAKEN.add(key)
        return d.view(w.shape)
    return None


def linear_dw(dy, x, out=None, accumulate=False, w=None):
    """dW[N,K] = dy[M,N]^T @ x[M,K]  (both operands red-major); `w` = the weight this is the
    gradient of (looked up in GRAD_DST when no explicit `out` is given)"""
    M, N = dy.shape
    K = x.shape[1]
    if out is None and not accumulate:
        out = grad_dst(w)
    if out is None:
        out = torch.empty((N, K), dtype=dy.dtype, device=dy.device)
    gemm_raw(dy, x, out, N, K, M, _rowmajor(dy), _rowmajor(x), _rowmajor(out), a_red=True,
             b_red=True, accumulate=accumulate)
    return out


def transpose(x, out=None):
    """out[c, r] = x[r, c] for a 2-D row-major tensor (batched form: 3-D contiguous)."""
    lib = _L.load()
    if x.dim() == 2:
        rows, cols = x.shape
        if out is None:
            out = torch.empty((cols, rows), dtype=x.dtype, device=x.device)
        _L.check(lib.mk_transpose(_p(x), _p(out), rows, cols, _rowmajor(x), _rowmajor(out), 1, 0, 0,
                                  x.element_size(), _st()), "mk_transpose")
        return out
    b, rows, cols = x.shape
    x = x.contiguous()
    if out is None:
        out = torch.empty((b, cols, rows), dtype=x.dtype, device=x.device)
    _L.check(lib.mk_transpose(_p(x), _p(out), rows, cols, cols, rows, b, rows * cols, rows * cols,
                              x.element_size(), _st()), "mk_transpose")
    return out


# ------------------------------------------------------------------ norms --
NORM_BLOCKS = 512  # row-slab blocks for the weight-gradient partial sums
# RMSNorm backward prefetches its next row (csrc/norm.hip): ONE block per CU keeps more rows in flight than two
# did without it, and halves the partial sums the reduction has to read (scripts/bench_norm.py, 4608 x 4096:
# 37.6 us at 256 blocks against 45.4 at 512; the round-2 kernel: 55.4 / 49.9)
RMSNORM_BLOCKS = 256


def rmsnorm_fwd(x, w, eps, res=None):
    """returns (h, y, rstd); h = x (+ res) — h is x itself when res is None"""
    lib = _L.load()
    rows, cols = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    h = torch.empty_like(x) if res is not None else x
    _L.check(lib.mk_rmsnorm_fwd(_p(x), _p(res), _p(w), _p(h) if res is not None else None, _p(y),
                                _p(rstd), rows, cols, eps, dt(x), _st()), "mk_rmsnorm_fwd")
    return h, y, rstd


def rmsnorm_fwd_fp8(x, w, eps, res=None):
    """rmsnorm_fwd + quantize_fp8_rows(y) in one pass over the row: returns (h, y, rstd, q uint8 [rows, cols],
    scales f32 [rows]) -- bit-identical to the two calls"""
    lib = _L.load()
    rows, cols = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    h = torch.empty_like(x) if res is not None else x
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    _L.check(lib.mk_rmsnorm_fwd_fp8(_p(x), _p(res), _p(w), _p(h) if res is not None else None, _p(y), _p(rstd),
                                    _p(q), cols, _p(sc), rows, cols, eps, dt(x), _st()), "mk_rmsnorm_fwd_fp8")
    return h, y, rstd, q, sc


def rmsnorm_bwd(dy, h, w, rstd, dres=None, dw_out=None, dw_accumulate=False):
    """returns (dx, dw); dx = dres + d/dh, dw in w.dtype"""
    lib = _L.load()
    rows, cols = h.shape
    nblk = min(RMSNORM_BLOCKS, rows)
    dx = torch.empty_like(h)
    part = torch.empty((nblk, cols), dtype=torch.float32, device=h.device)
    _L.check(lib.mk_rmsnorm_bwd(_p(dy), _p(h), _p(w), _p(rstd), _p(dres), _p(dx), _p(part), nblk,
                                rows, cols, dt(h), _st()), "mk_rmsnorm_bwd")
    if dw_out is None:
        dw_out = torch.empty_like(w)
        dw_accumulate = False
    _L.check(lib.mk_colsum_partials(_p(part), _p(dw_out), nblk, cols, int(dw_accumulate),
                                    dt(w), _st()), "mk_colsum_partials")
    return dx, dw_out


def layernorm_fwd(x, w, b, eps):
    lib = _L.load()
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    _L.check(lib.mk_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, cols, eps,
                                  dt(x), _st()), "mk_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dres=None):
    """returns (dx, dw, db)"""
    lib = _L.load()
    rows, cols = x.shape
    nblk = min(NORM_BLOCKS, rows)
    dx = torch.empty_like(x)
    pw = torch.empty((nblk, cols), dtype=torch.float32, device=x.device)
    pb = torch.empty((nblk, cols), dtype=torch.float32, device=x.device)
    _L.check(lib.mk_layernorm_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx),
                                  _p(pw), _p(pb), nblk, rows, cols, dt(x), _st()),
             "mk_layernorm_bwd")
    dw = torch.empty_like(w)
    db = torch.empty_like(w)
    _L.check(lib.mk_colsum_partials(_p(pw), _p(dw), nblk, cols, 0, dt(w), _st()), "colsum")
    _L.check(lib.mk_colsum_partials(_p(pb), _p(db), nblk, cols, 0, dt(w), _st()), "colsum")
    return dx, dw, db


def colsum(x, out=None, accumulate=False):
    """out[c] (+)= sum_r x[r, c]  (bias gradient)"""
    lib = _L.load()
    rows, cols = x.shape
    nblk = max(1, min(256, rows // 16))
    ws = torch.empty((nblk, cols), dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(cols, dtype=x.dtype, device=x.device)
        accumulate = False
    _L.check(lib.mk_colsum(_p(x), _rowmajor(x), _p(out), _p(ws), nblk, rows, cols, int(accumulate),
                           dt(x), _st()), "mk_colsum")
    return out


# -------------------------------------------------------------- pointwise --
def rope_(x, cos_t, sin_t, pos, heads, hd, inverse=False):
    """in-place RoPE on x [tokens, heads*hd] (pitched rows allowed)"""
    lib = _L.load()
    tokens = x.shape[0]
    _L.check(lib.mk_rope(_p(x), _p(cos_t), _p(sin_t), _p(pos), tokens, heads, hd, _rowmajor(x),
                         int(inverse), dt(x), _st()), "mk_rope")
    return x


def swiglu_fwd(g, u):
    lib = _L.load()
    a = torch.empty_like(g)
    _L.check(lib.mk_swiglu_fwd(_p(g), _p(u), _p(a), g.numel(), dt(g), _st()), "mk_swiglu_fwd")
    return a


def swiglu_bwd(g, u, da):
    lib = _L.load()
    dg, du = torch.empty_like(g), torch.empty_like(u)
    _L.check(lib.mk_swiglu_bwd(_p(g), _p(u), _p(da), _p(dg), _p(du), g.numel(), dt(g), _st()),
             "mk_swiglu_bwd")
    return dg, du


def swiglu2d_fwd(gu, cols):
    """gu [rows, 2*cols] = [gate | up]; returns a [rows, cols] = silu(gate) * up"""
    lib = _L.load()
    rows = gu.shape[0]
    a = torch.empty((rows, cols), dtype=gu.dtype, device=gu.device)
    es = gu.element_size()
    _L.check(lib.mk_swiglu2d_fwd(_p(gu), _p(gu) + cols * es, _p(a), rows, cols, gu.stride(0), cols,
                                 dt(gu), _st()), "mk_swiglu2d_fwd")
    return a


def swiglu2d_bwd(gu, da, cols):
    """returns dgu [rows, 2*cols] = [dgate | dup]"""
    lib = _L.load()
    rows = gu.shape[0]
    dgu = torch.empty_like(gu)
    es = gu.element_size()
    _L.check(lib.mk_swiglu2d_bwd(_p(gu), _p(gu) + cols * es, _p(da), _p(dgu), _p(dgu) + cols * es,
                                 rows, cols, gu.stride(0), da.stride(0), dt(gu), _st()),
             "mk_swiglu2d_bwd")
    return dgu


def act_fwd(x, act):
    lib = _L.load()
    y = torch.empty_like(x)
    _L.check(lib.mk_act_fwd(_p(x), _p(y), x.numel(), act, dt(x), _st()), "mk_act_fwd")
    return y


def act_bwd(x_pre, dy, act):
    lib = _L.load()
    dx = torch.empty_like(dy)
    _L.check(lib.mk_act_bwd(_p(x_pre), _p(dy), _p(dx), dy.numel(), act, dt(dy), _st()), "mk_act_bwd")
    return dx


def add(a, b, out=None, period=0):
    lib = _L.load()
    if out is None:
        out = torch.empty_like(a)
    _L.check(lib.mk_add(_p(a), _p(b), _p(out), a.numel(), period, dt(a), _st()), "mk_add")
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    lib = _L.load()
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _L.check(lib.mk_cast(_p(x), dt(x), _p(out), _DT[dtype], x.numel(), _st()), "mk_cast")
    return out


def fill_(x, v):
    lib = _L.load()
    _L.check(lib.mk_fill(_p(x), float(v), x.numel(), dt(x), _st()), "mk_fill")
    return x


_SUMSQ_WS = {}


def sumsq(x, out=None, accumulate=False):
    """out[0] (+)= sum x^2 (fp32, deterministic); x contiguous / 16-byte aligned"""
    lib = _L.load()
    ws = _SUMSQ_WS.get(x.device)
    if ws is None:
        ws = _SUMSQ_WS[x.device] = torch.empty(1024, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        accumulate = False
    _L.check(lib.mk_sumsq(_p(x), x.numel(), _p(ws), _p(out), int(accumulate), dt(x), _st()), "mk_sumsq")
    return out


def copy2d(src, dst, rows, cols, ld_src, ld_dst, batch=1, s_src=0, s_dst=0, src_off=0, dst_off=0):
    """strided 2-D (batched) copy; offsets/strides in elements"""
    lib = _L.load()
    es = src.element_size()
    _L.check(lib.mk_copy2d(_p(src) + src_off * es, _p(dst) + dst_off * es, rows, cols, ld_src, ld_dst,
                           batch, s_src, s_dst, es, _st()), "mk_copy2d")
    return dst


def set_dropout_seed_offset(t):
    """register (or, with None, clear) a device int64[1] whose value every dropout kernel adds to
    its seed at execution time (mk_set_dropout_seed_offset)"""
    lib = _L.load()
    _L.check(lib.mk_set_dropout_seed_offset(_p(t) if t is not None else None), "mk_set_dropout_seed_offset")


def kv_append(src, cache, cols, batch, s_src, s_cache, ld_cache, t_dev, t_max, src_off=0, dst_off=0):
    """cache[b, *t_dev, dst_off : dst_off + cols] = src[b, src_off : src_off + cols]; the row index is
    read from the DEVICE int32 t_dev at execution time (graph-capturable decode step)"""
    lib = _L.load()
    es = src.element_size()
    _L.check(lib.mk_kv_append(_p(src) + src_off * es, _p(cache) + dst_off * es, cols, batch, s_src, s_cache,
                              ld_cache, _p(t_dev), t_max, es, _st()), "mk_kv_append")
    return cache


def decode_attn(q, k, v, out, t_dev, t_add, t_max, B, H, hd, q_bs, k_ld, k_bs, v_ld, v_bs, o_bs, scale,
                k_off=0, v_off=0):
    """one query row per (sample, head) against the first *t_dev + t_add cached keys (device int32)"""
    lib = _L.load()
    es = q.element_size()
    _L.check(lib.mk_decode_attn(_p(q), _p(k) + k_off * es, _p(v) + v_off * es, _p(out), _p(t_dev), t_add,
                                t_max, B, H, hd, q_bs, k_ld, k_bs, v_ld, v_bs, o_bs, scale, dt(q), _st()),
             "mk_decode_attn")
    return out


def decode_linear(x, W, prologue=0, norm_w=None, eps=0.0, residual=None, out=None):
    """y[M <= 16, N] = prologue(x) W^T (+ residual) in one launch: prologue 1 = RMSNorm(x; norm_w, eps),
    2 = SwiGLU of x = [gate | up] ([M, 2K]); W [N, K] row-major"""
    lib = _L.load()
    M = x.shape[0]
    N, K = W.shape
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _L.check(lib.mk_decode_linear(_p(x), _rowmajor(x), _p(W), _rowmajor(W), _p(out), _rowmajor(out),
                                  _p(residual), _rowmajor(residual) if residual is not None else 0, M, N, K,
                                  prologue, _p(norm_w), eps, dt(x), _st()), "mk_decode_linear")
    return out


def decode_linear_ok(x, W, prologue=0):
    """mk_decode_linear's domain; with a prologue the prepared token rows must fit 40 KiB of LDS"""
    M, K = x.shape[0], W.shape[1]
    return (x.dtype in (torch.bfloat16, torch.float16) and M <= (16 if prologue else 32) and K % 64 == 0 and W.is_contiguous()
            and (prologue == 0 or M * (K + 8) * 2 <= 40 * 1024))


def decode_emit(logits, V, pad, eos, tok, done, out, state):
    """greedy selection + step bookkeeping in one launch (see mk_decode_emit): logits [B, >= V]
    row-major, tok int64 [B], done bool [B], out int64 [B, n], state int32 [>= 3] = (position, output
    column, 0)"""
    lib = _L.load()
    B = logits.shape[0]
    _L.check(lib.mk_decode_emit(_p(logits), _rowmajor(logits), V, B, pad, eos, _p(tok), _p(done), _p(out),
                                out.stride(0), _p(state), dt(logits), _st()), "mk_decode_emit")


def decode_step_attn(q, k_new, v_new, in_bs, cos_t, sin_t, cache, t_dev, t_max, B, H, hd, out, scale,
                     q_off=0, k_off=0, v_off=0):
    """RoPE(q, k_new) at position *t_dev + append [k_new | v_new] to cache [B, t_max, 2 * H * hd] row
    *t_dev + attention of q over keys 0 ... *t_dev, one launch (q / k_new / v_new may be slices of one
    fused [B, 3D] buffer: element offsets q_off / k_off / v_off)"""
    lib = _L.load()
    es = q.element_size()
    D = H * hd
    _L.check(lib.mk_decode_step_attn(_p(q) + q_off * es, _p(k_new) + k_off * es, _p(v_new) + v_off * es,
                                     in_bs, _p(cos_t), _p(sin_t), _p(cache), _p(cache) + D * es, 2 * D,
                                     t_max * 2 * D, _p(out), D, _p(t_dev), t_max, B, H, hd, scale, dt(q),
                                     _st()), "mk_decode_step_attn")
    return out


def decode_attn_ok(dtype, hd, t_max):
    return dtype in (torch.bfloat16, torch.float16) and hd in (16, 32, 64, 128) and t_max * 4 <= 60 * 1024


def embedding_fwd(table, ids, out=None):
    """out[t, :] = table[ids[t], :]; ids int64 1-D"""
    lib = _L.load()
    tokens = ids.numel()
    vocab, dim = table.shape
    if out is None:
        out = torch.empty((tokens, dim), dtype=table.dtype, device=table.device)
    _L.check(lib.mk_embedding_fwd(_p(table), _p(ids), _p(out), tokens, dim, _rowmajor(out), vocab,
                                  dt(table), _st()), "mk_embedding_fwd")
    return out


def embedding_bwd_(dtable, dout, ids, padding_idx=-1):
    """dtable[ids[t]] += dout[t] (deterministic)"""
    lib = _L.load()
    vocab, dim = dtable.shape
    _L.check(lib.mk_embedding_bwd(_p(dout), _rowmajor(dout), _p(ids), _p(dtable), ids.numel(), dim,
                                  vocab, -1 if padding_idx is None else padding_idx, dt(dtable),
                                  _st()), "mk_embedding_bwd")
    return dtable


def pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def im2col1d(x, B, C_, T, kw, stride, pad, sb, sc, st, ld_out=None):
    lib = _L.load()
    Lout = (T + 2 * pad - kw) // stride + 1
    K = C_ * kw
    if ld_out is None:
        ld_out = pad8(K)
    out = torch.empty((B * Lout, ld_out), dtype=x.dtype, device=x.device)
    _L.check(lib.mk_im2col1d(_p(x), _p(out), B, C_, T, kw, stride, pad, Lout, sb, sc, st, ld_out,
                             dt(x), _st()), "mk_im2col1d")
    return out, Lout


def col2im1d(dcols, B, C_, T, kw, stride, pad, Lout, sb, sc, st, out_shape):
    lib = _L.load()
    dx = torch.empty(out_shape, dtype=dcols.dtype, device=dcols.device)
    _L.check(lib.mk_col2im1d(_p(dcols), _p(dx), B, C_, T, kw, stride, pad, Lout, sb, sc, st,
                             _rowmajor(dcols), dt(dcols), _st()), "mk_col2im1d")
    return dx


def patchify(img, P):
    lib = _L.load()
    B, C_, H, W = img.shape
    img = img.contiguous()
    K = C_ * P * P
    ld = pad8(K)
    out = torch.empty((B * (H // P) * (W // P), ld), dtype=img.dtype, device=img.device)
    _L.check(lib.mk_patchify(_p(img), _p(out), B, C_, H, W, P, ld, dt(img), _st()), "mk_patchify")
    return out


def unpatchify(dcols, B, C_, H, W, P):
    lib = _L.load()
    dimg = torch.empty((B, C_, H, W), dtype=dcols.dtype, device=dcols.device)
    _L.check(lib.mk_unpatchify(_p(dcols), _p(dimg), B, C_, H, W, P, _rowmajor(dcols), dt(dcols),
                               _st()), "mk_unpatchify")
    return dimg


# ---------------------------------------------------------------- softmax --
def softmax_fwd(scores, nz, heads, Lq, Lk, ld, kmask=None, causal=False, dropout_p=0.0, seed=0,
                probs=None, want_dropped=False):
    """scores: buffer of nz*Lq rows with pitch ld.  Returns (probs, probs_dropped|None)."""
    lib = _L.load()
    if probs is None:
        probs = torch.empty_like(scores)
    pd = torch.empty_like(scores) if (dropout_p > 0.0 and want_dropped) else None
    _L.check(lib.mk_softmax_fwd(_p(scores), _p(probs), _p(pd), _p(kmask), nz, heads, Lq, Lk, ld,
                                int(causal), float(dropout_p), int(seed), dt(scores), _st()),
             "mk_softmax_fwd")
    return probs, pd


def softmax_bwd_(probs, dprobs, nz, Lq, Lk, ld, scale=1.0, dropout_p=0.0, seed=0):
    lib = _L.load()
    _L.check(lib.mk_softmax_bwd(_p(probs), _p(dprobs), nz, Lq, Lk, ld, float(scale),
                                float(dropout_p), int(seed), dt(probs), _st()), "mk_softmax_bwd")
    return dprobs


def flash_rope_ok(hd, Lq, Lk, cos_t, x) -> bool:
    """RoPE can ride inside the fused attention kernels (mk_flash_attn_rope_*): the one-workgroup-per-(b, h)
    short-sequence kernels, tables in the activations' dtype."""
    return (hd == 128 and Lq == Lk and Lk <= 160 and cos_t.dtype == x.dtype and cos_t.shape[-1] == hd
            and cos_t.is_contiguous())


def rope_fuse_mode() -> str:
    """how LlamaLayerFn folds RoPE into the short-sequence attention kernels (MACAW_ROPE_FUSE, read per layer call).
    "off" (default): mk_rope before the forward, mk_rope(inverse) behind the backward -- three launches;
    "bwd": the rotation of dq / dk back inside the backward kernel's stores; "full": q, k stay unrotated in HBM and
    are rotated on every load (three rotations per element in the backward).  All three are bit-identical
    (tests/test_kernels_gpu.py); measured in the cfg-3 step the fused forms save the 1.9 ms of the two in-place
    launches and spend 0.9 (bwd) / 1.8 ms (full) more inside the attention kernels, whose tails wait for the table
    rows and round six times per output pair: 215.1-215.6 / 214.6-215.2 / 215.0-215.5 ms -- no gain, so the plain
    form stays the default (profiles/r06_rope_fuse.txt)."""
    m = os.environ.get("MACAW_ROPE_FUSE", "off")
    if m not in ("bwd", "full", "off"):
        raise MacawHipError(f"MACAW_ROPE_FUSE={m!r}: expected bwd, full or off")
    return m


def flash_attn_fwd(q, k, v, o, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs,
                   scale, kmask=None, causal=False, lse=None, q_off=0, k_off=0, v_off=0, o_off=0,
                   rope=None):
    """fused attention forward on strided bf16 buffers (element offsets/strides).
    rope = (cos_t, sin_t, pos): q and k are the unrotated projections (see flash_rope_ok)"""
    lib = _L.load()
    es = q.element_size()
    if rope is not None:
        cos_t, sin_t, pos = rope
        _L.check(lib.mk_flash_attn_rope_fwd(_p(q) + q_off * es, _p(k) + k_off * es, _p(v) + v_off * es,
                                            _p(o) + o_off * es, _p(lse), _p(kmask), _p(cos_t), _p(sin_t),
                                            _p(pos), B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs,
                                            o_ld, o_bs, float(scale), int(causal), dt(q), _st()),
                 "mk_flash_attn_rope_fwd")
        return o
    _L.check(lib.mk_flash_attn_fwd(_p(q) + q_off * es, _p(k) + k_off * es, _p(v) + v_off * es,
                                   _p(o) + o_off * es, _p(lse), _p(kmask), B, H, Lq, Lk, hd, q_ld,
                                   q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, float(scale),
                                   int(causal), dt(q), _st()), "mk_flash_attn_fwd")
    return o


def flash_attn_bwd(q, k, v, o, dout, lse, dq, dk, dv, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld,
                   v_bs, o_ld, o_bs, scale, kmask=None, causal=False, rope=None, qk_rotated=False):
    """rope = (cos_t, sin_t, pos): dq / dk come back as gradients of the UNROTATED q, k; q and k themselves are
    the unrotated tensors, or with qk_rotated the rotated ones (mk_rope ran before the forward)"""
    lib = _L.load()
    dvec = torch.empty(B * H * Lq, dtype=torch.float32, device=q.device)
    if rope is not None:
        cos_t, sin_t, pos = rope
        _L.check(lib.mk_flash_attn_rope_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(dvec), _p(dq),
                                            _p(dk), _p(dv), _p(kmask), _p(cos_t), _p(sin_t), _p(pos), B, H,
                                            Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs,
                                            float(scale), int(causal), int(qk_rotated), dt(q), _st()),
                 "mk_flash_attn_rope_bwd")
        return
    _L.check(lib.mk_flash_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(dvec), _p(dq),
                                   _p(dk), _p(dv), _p(kmask), B, H, Lq, Lk, hd, q_ld, q_bs, k_ld,
                                   k_bs, v_ld, v_bs, o_ld, o_bs, float(scale), int(causal), dt(q),
                                   _st()), "mk_flash_attn_bwd")


def cross_entropy(logits, labels, V):
    """logits [rows, ld>=V] row-major; labels int64 [rows] already shifted.
    returns (row_loss, row_lse, sum_cnt[2])"""
    lib = _L.load()
    rows = logits.shape[0]
    row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    row_lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    sum_cnt = torch.empty(4, dtype=torch.float32, device=logits.device)
    _L.check(lib.mk_cross_entropy(_p(logits), _p(labels), _p(row_loss), _p(row_lse), _p(sum_cnt),
                                  rows, V, _rowmajor(logits), dt(logits), _st()),
             "mk_cross_entropy")
    return row_loss, row_lse, sum_cnt


def cross_entropy_bwd(logits, labels, row_lse, sum_cnt, V, grad_scale=1.0, out=None,
                      grad_scale_dev=None):
    lib = _L.load()
    rows = logits.shape[0]
    if out is None:
        out = torch.empty_like(logits)
    _L.check(lib.mk_cross_entropy_bwd(_p(logits), _p(out), _p(labels), _p(row_lse), _p(sum_cnt),
                                      float(grad_scale), _p(grad_scale_dev), rows, V,
                                      _rowmajor(logits), dt(logits),
                                      _st()), "mk_cross_entropy_bwd")
    return out


def argmax_rows(x, cols=None):
    """first-max index of every row of a 2-D row-major (pitched) tensor -> int64 [rows]"""
    lib = _L.load()
    rows = x.shape[0]
    cols = x.shape[1] if cols is None else cols
    out = torch.empty(rows, dtype=torch.int64, device=x.device)
    _L.check(lib.mk_argmax_rows(_p(x), _rowmajor(x), rows, cols, _p(out), dt(x), _st()),
             "mk_argmax_rows")
    return out


def adamw_(param, master, m, v, grad, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    lib = _L.load()
    _L.check(lib.mk_adamw(_p(param), _p(master), _p(m), _p(v), _p(grad), param.numel(), lr, beta1,
                          beta2, eps, wd, step, grad_scale, dt(param), _st()), "mk_adamw")


def prof_begin():
    _L.check(_L.load().mk_prof_begin(), "mk_prof_begin")


def prof_report(path: str):
    _L.check(_L.load().mk_prof_report(path.encode()), "mk_prof_report")


def prof_sum(kind: int):
    """(total_ms, total_flops, launches) of one launch kind since prof_begin: 0 GEMM, 1 fused
    attention forward, 2 fused attention backward.  Call before prof_end."""
    ms, fl, n = C.c_double(0), C.c_double(0), C.c_int64(0)
    _L.check(_L.load().mk_prof_sum(kind, C.byref(ms), C.byref(fl), C.byref(n)), "mk_prof_sum")
    return ms.value, fl.value, n.value


def prof_end():
    """returns (total_ms, total_flops, launches) of the mk_gemm launches since prof_begin"""
    ms, fl, n = C.c_double(0), C.c_double(0), C.c_int64(0)
    _L.check(_L.load().mk_prof_end(C.byref(ms), C.byref(fl), C.byref(n)), "mk_prof_end")
    return ms.value, fl.value, n.value
