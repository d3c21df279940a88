// Fused (flash-style) attention forward for gfx950: softmax(scale * Q K^T + mask) V without
// materialising the score matrix (at S = 2048 the fp32 score tensor is 537 MB per
// sample per layer).  bf16 in/out, fp32 softmax state, head_dim 64 (CLIP / Whisper) or 128
// (LLaMA).  Replaces modeling.py:197-215 and the HF encoder attention everywhere in the 16-bit engines: the
// frozen towers, inference AND training -- the fused backward (flash_bwd_prep / _dq / _dkv in
// attention_impl.inc) has been the LLaMA layers' training path since round 1; only the fp32 parity engine keeps
// the GEMM + softmax formulation.
//
// Long sequences at head_dim 128 (Lq >= 1024, LLaMA at S = 2048): flash_fwd8_kernel -- 8 waves = 256 query rows per
// workgroup, matrix and softmax segments of its two wave groups interleaved across barriers (attention_impl.inc).
// Work decomposition of the 4-wave kernels: block = 4 waves, each wave owns 32 query rows; the block walks the keys
// in tiles of 64 staged in LDS (K: [key][d] XOR-swizzled for ds_read_b128; V: [key][d] read
// back TRANSPOSED with ds_read_b64_tr_b16).  Both products use v_mfma_f32_32x32x16_bf16 with
// the operands swapped so that a lane owns ONE query column:
//   S^T[key][q] = K Q^T      lane (q = l&31, half = l>>5) holds 16 keys of its row
//   O^T[d][q]  += V^T P^T    P^T fragments are exactly the lane's own S registers (the MFMA
//                            k-slot <-> key mapping is chosen to match, no cross-lane shuffle)
// Row max / sum need one __shfl_xor(.., 32) with the partner half.
#include <utility>
#include "common.h"
#include "../../include/macaw_hip.h"

// written once over a 16-bit element type, instantiated for bf16 and f16 (common.h E16<>)
#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#include "attention_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define flash_fwd_kernel flash_fwd_f16_kernel
#define flash_fwd8_kernel flash_fwd8_f16_kernel
#define flash_fwd4x64_kernel flash_fwd4x64_f16_kernel
#define flash_bwd_prep_kernel flash_bwd_prep_f16_kernel
#define flash_bwd_dq_kernel flash_bwd_dq_f16_kernel
#define flash_bwd_dkv_kernel flash_bwd_dkv_f16_kernel
#define flash_bwd_short_kernel flash_bwd_short_f16_kernel
#define flash_fwd_short_kernel flash_fwd_short_f16_kernel
#include "attention_impl.inc"
#undef flash_fwd_kernel
#undef flash_fwd8_kernel
#undef flash_fwd4x64_kernel
#undef flash_bwd_prep_kernel
#undef flash_bwd_dq_kernel
#undef flash_bwd_dkv_kernel
#undef flash_bwd_short_kernel
#undef flash_fwd_short_kernel
#undef MK_E16_T
#undef MK_E16_NS

extern "C" int mk_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                 const int32_t* kmask, int32_t B, int32_t H, int32_t Lq,
                                 int32_t Lk, int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld,
                                 int64_t k_bs, int64_t v_ld, int64_t v_bs, int64_t o_ld,
                                 int64_t o_bs, float scale, int32_t causal, int32_t dtype,
                                 void* stream) {
  if (dtype == MK_F16) return e_f16::flash_attn_fwd_impl(q, k, v, o, lse, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream);
  return e_bf16::flash_attn_fwd_impl(q, k, v, o, lse, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream);
}

extern "C" int mk_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o,
                                 const void* dout, const float* lse, float* dvec, void* dq,
                                 void* dk, void* dv, const int32_t* kmask, int32_t B, int32_t H,
                                 int32_t Lq, int32_t Lk, int32_t hd, int64_t q_ld, int64_t q_bs,
                                 int64_t k_ld, int64_t k_bs, int64_t v_ld, int64_t v_bs,
                                 int64_t o_ld, int64_t o_bs, float scale, int32_t causal,
                                 int32_t dtype, void* stream) {
  if (dtype == MK_F16) return e_f16::flash_attn_bwd_impl(q, k, v, o, dout, lse, dvec, dq, dk, dv, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream);
  return e_bf16::flash_attn_bwd_impl(q, k, v, o, dout, lse, dvec, dq, dk, dv, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream);
}

// The same two kernels with RoPE (modeling.py:76-91, 167-170) folded in: q and k arrive UNROTATED, are rotated on
// their way into registers / LDS with mk_rope's arithmetic (bit-identical q, k), dq and dk leave as gradients of the
// unrotated tensors (mk_flash_attn_rope_bwd with qk_rotated = 1: q and k ARE the rotated tensors -- mk_rope ran before
// the forward -- and only the rotation of dq / dk back is folded in).  Short sequences only (head_dim 128, Lq == Lk <= 160: the one-workgroup-per-(b, h) kernels, where
// every q / k element is loaded once); anything else returns MK_ERR_UNSUPPORTED and the caller launches mk_rope.
extern "C" int mk_flash_attn_rope_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                      const int32_t* kmask, const void* cos_t, const void* sin_t,
                                      const int32_t* pos, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                                      int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                                      int64_t v_ld, int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale,
                                      int32_t causal, int32_t dtype, void* stream) {
  if (!cos_t || !sin_t || !pos) return MK_ERR_BAD_ARG;
  if (hd != 128 || Lq != Lk || Lk > 160) return MK_ERR_UNSUPPORTED;
  if (dtype == MK_F16) return e_f16::flash_attn_fwd_impl(q, k, v, o, lse, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream, cos_t, sin_t, pos);
  return e_bf16::flash_attn_fwd_impl(q, k, v, o, lse, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream, cos_t, sin_t, pos);
}

extern "C" int mk_flash_attn_rope_bwd(const void* q, const void* k, const void* v, const void* o,
                                      const void* dout, const float* lse, float* dvec, void* dq, void* dk,
                                      void* dv, const int32_t* kmask, const void* cos_t, const void* sin_t,
                                      const int32_t* pos, int32_t B, int32_t H, int32_t Lq, int32_t Lk,
                                      int32_t hd, int64_t q_ld, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                                      int64_t v_ld, int64_t v_bs, int64_t o_ld, int64_t o_bs, float scale,
                                      int32_t causal, int32_t qk_rotated, int32_t dtype, void* stream) {
  if (!cos_t || !sin_t || !pos) return MK_ERR_BAD_ARG;
  if (hd != 128 || Lq != Lk || Lk > 160) return MK_ERR_UNSUPPORTED;
  if (dtype == MK_F16) return e_f16::flash_attn_bwd_impl(q, k, v, o, dout, lse, dvec, dq, dk, dv, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream, cos_t, sin_t, pos, qk_rotated);
  return e_bf16::flash_attn_bwd_impl(q, k, v, o, dout, lse, dvec, dq, dk, dv, kmask, B, H, Lq, Lk, hd, q_ld, q_bs, k_ld, k_bs, v_ld, v_bs, o_ld, o_bs, scale, causal, dtype, stream, cos_t, sin_t, pos, qk_rotated);
}
