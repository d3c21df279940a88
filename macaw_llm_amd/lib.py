"""ctypes binding of libmacaw_hip.so (the C ABI declared in include/macaw_hip.h).

The product path has NO fallback: if the library is missing or a symbol is
absent, importing/using the ops raises.  (The CPU oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# PyTorch must load ITS HIP runtime first: the library is linked against the system ROCm 7.2
# libamdhip64 while torch bundles its own (ROCm 7.0 build).  If ours were dlopen'ed first the
# process would end up with kernels registered in one runtime and torch's streams/allocations in
# the other, and every launch fails.  Importing torch here makes the order deterministic.
import torch  # noqa: F401

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libmacaw_hip.so"

MK_F32, MK_BF16, MK_F16, MK_FP8 = 0, 1, 2, 3
ABI_VERSION = 6

_ERR = {-1: "MK_ERR_BAD_ARG", -2: "MK_ERR_UNSUPPORTED", -3: "MK_ERR_LAUNCH"}


class MacawHipError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("R", C.c_void_p),
        ("bias", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("a_red_major", C.c_int32), ("b_red_major", C.c_int32),
        ("nb1", C.c_int32), ("nb2", C.c_int32),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64), ("sR1", C.c_int64), ("sR2", C.c_int64),
        ("alpha", C.c_float),
        ("bias_mode", C.c_int32), ("act", C.c_int32), ("accumulate", C.c_int32),
        ("dtype", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("scale_a", C.c_void_p), ("scale_b", C.c_void_p),
        ("flags", C.c_int32),
    ]


_vp, _i32, _i64, _f32, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64

# name -> argtypes; every symbol declared in include/macaw_hip.h
SIGNATURES = {
    "mk_abi_version": [],
    "mk_gemm": [C.POINTER(GemmDesc), _vp],
    "mk_gemm_set_cfg": [_i32],
    "mk_gemm_has_cfg": [_i32],
    "mk_gemm_set_cus": [_i32],
    "mk_prof_begin": [],
    "mk_prof_end": [_vp, _vp, _vp],
    "mk_prof_sum": [_i32, _vp, _vp, _vp],
    "mk_prof_report": [C.c_char_p],
    "mk_transpose": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _vp],
    "mk_rmsnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "mk_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "mk_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "mk_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "mk_colsum_partials": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "mk_colsum": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "mk_rope": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _i32, _vp],
    "mk_swiglu_fwd": [_vp, _vp, _vp, _i64, _i32, _vp],
    "mk_swiglu_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "mk_swiglu2d_fwd": [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _i32, _vp],
    "mk_swiglu2d_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i32, _vp],
    "mk_act_fwd": [_vp, _vp, _i64, _i32, _i32, _vp],
    "mk_act_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "mk_add": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "mk_cast": [_vp, _i32, _vp, _i32, _i64, _vp],
    "mk_fill": [_vp, _f32, _i64, _i32, _vp],
    "mk_sumsq": [_vp, _i64, _vp, _vp, _i32, _i32, _vp],
    "mk_copy2d": [_vp, _vp, _i32, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _vp],
    "mk_embedding_fwd": [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp],
    "mk_embedding_bwd": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp],
    "mk_im2col1d": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64,
                    _i32, _vp],
    "mk_col2im1d": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64,
                    _i32, _vp],
    "mk_patchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mk_unpatchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i32, _vp],
    "mk_softmax_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _i32, _f32, _u64, _i32,
                       _vp],
    "mk_softmax_bwd": [_vp, _vp, _i32, _i32, _i32, _i64, _f32, _f32, _u64, _i32, _vp],
    "mk_flash_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64,
                          _i64, _i64, _i64, _i64, _i64, _f32, _i32, _i32, _vp],
    "mk_flash_attn_bwd": [_vp] * 11 + [_i32] * 5 + [_i64] * 8 + [_f32, _i32, _i32, _vp],
    "mk_flash_attn_rope_fwd": [_vp] * 9 + [_i32] * 5 + [_i64] * 8 + [_f32, _i32, _i32, _vp],
    "mk_flash_attn_rope_bwd": [_vp] * 14 + [_i32] * 5 + [_i64] * 8 + [_f32, _i32, _i32, _i32, _vp],
    "mk_cross_entropy": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp],
    "mk_cross_entropy_bwd": [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i32, _i64, _i32, _vp],
    "mk_argmax_rows": [_vp, _i64, _i32, _i32, _vp, _i32, _vp],
    "mk_decode_linear": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _f32, _i32, _vp],
    "mk_decode_emit": [_vp, _i64, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _vp],
    "mk_decode_step_attn": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i32, _i32,
                            _i32, _i32, _f32, _i32, _vp],
    "mk_adamw_bias_correction": [_f32, _f32, _i32, _vp],
    "mk_adamw_multi_dev": [_vp, _vp, _i32, _i64, _f32, _f32, _f32, _f32, _vp, _i32, _vp],
    "mk_set_dropout_seed_offset": [_vp],
    "mk_kv_append": [_vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i32, _vp],
    "mk_decode_attn": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _i64,
                       _i64, _f32, _i32, _vp],
    "mk_adamw_set_max_blocks": [_i32],
    "mk_adamw_chunk": [],
    "mk_adamw": [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _i32,
                 _vp],
    "mk_adamw_multi": [_vp, _vp, _i32, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _i32, _vp],
    "mk_fp8_quantize": [_vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "mk_fp8_quantize_rows": [_vp, _i32, _i32, _i64, _i32, _vp, _i64, _vp, _vp],
    "mk_rmsnorm_fwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _f32, _i32, _vp],
    "mk_fp8_quantize_cols_t": [_vp, _i32, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _vp],
    "mk_image_transform": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp],
    "mk_log_mel": [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
}

_lib = None


def load() -> C.CDLL:
    """Load the library (once). Raises MacawHipError when it is absent/broken."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MacawHipError(
            f"{LIB_PATH} not found. Build it with `python -m macaw_llm_amd.build` "
            "(__graft_entry__.build()). There is no CPU fallback for the product path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MacawHipError(f"symbol {name} missing from {LIB_PATH}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    v = lib.mk_abi_version()
    if v != ABI_VERSION:
        raise MacawHipError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, op: str) -> None:
    if rc != 0:
        raise MacawHipError(f"{op} failed: {_ERR.get(rc, rc)}")
