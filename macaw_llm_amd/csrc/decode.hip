// Greedy-decode kernels whose per-step position comes from DEVICE memory, so that one decode step
// is a fixed launch sequence and can be captured once in a hipGraph and replayed per token
// (modeling.py:190-195,954-960: the reference appends to the KV cache with torch.cat and re-launches
// the whole eager step from Python for every token).
//
//   mk_kv_append    cache[b][*t_dev][0:cols] = src[b][0:cols]   (post-RoPE keys | values of the new token)
//   mk_decode_attn  one query row per (sample, head) against the first *t_dev + t_add cached keys
//   mk_decode_step_attn  RoPE(q, k_new) + append(k_new, v_new) + that attention in one launch
//
// The fused training / prefill attention (attention.hip) works on 128-row query tiles: at Lq = 1
// it would spend 127 of 128 MFMA rows on padding and, more to the point here, takes its key count
// as a launch argument.
#include "common.h"
#include "../../include/macaw_hip.h"

namespace {

__global__ __launch_bounds__(256) void kv_append_kernel(const char* src, char* dst, long row_bytes,
                                                        long s_src, long s_dst, long ld_dst,
                                                        const int32_t* t_dev, int t_max) {
  const int t = min(max(*t_dev, 0), t_max - 1);
  src += (long)blockIdx.y * s_src;
  dst += (long)blockIdx.y * s_dst + (long)t * ld_dst;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 16; i < row_bytes; i += (long)gridDim.x * 256 * 16)
    *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
}

template <typename T>
__global__ __launch_bounds__(1024) void decode_emit_kernel(const T* logits, long ld, int V, long pad,
                                                           long eos, int64_t* tok, unsigned char* done,
                                                           int64_t* out, long out_ld, int32_t* state) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int b = blockIdx.x;
  const T* xr = logits + (long)b * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int c0 = threadIdx.x; c0 < V; c0 += 1024 * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j * 1024;
      v[j] = c < V ? (float)xr[c] : -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (v[j] > best) { best = v[j]; idx = c0 + j * 1024; }      // ascending columns: first maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    const int col = state[1];
    const long nxt = done[b] ? pad : (long)idx;
    out[(long)b * out_ld + col] = nxt;
    if (nxt == eos) done[b] = 1;
    tok[b] = nxt;
    __threadfence();
    const int old = __hip_atomic_fetch_add(state + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (int)gridDim.x - 1) {          // every sample has read state[1]: advance the step
      state[0] += 1;
      state[1] = col + 1;
      state[2] = 0;
    }
  }
}

}  // namespace

// the 16-bit kernels: written once over an element type, instantiated for bf16 and f16 (common.h E16<>)
#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#include "decode_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define decode_attn_kernel decode_attn_f16_kernel
#define decode_step_attn_kernel decode_step_attn_f16_kernel
#define decode_step_attn4_kernel decode_step_attn4_f16_kernel
#include "decode_impl.inc"
#undef decode_attn_kernel
#undef decode_step_attn_kernel
#undef decode_step_attn4_kernel
#undef MK_E16_T
#undef MK_E16_NS

extern "C" int mk_decode_step_attn(const void* q, const void* k_new, const void* v_new, int64_t in_bs,
                                   const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                   int64_t kv_ld, int64_t kv_bs, void* o, int64_t o_bs,
                                   const int32_t* t_dev, int32_t t_max, int32_t B, int32_t H,
                                   int32_t hd, float scale, int32_t dtype, void* stream) {
  if (dtype == MK_F16) return e_f16::decode_step_attn_impl(q, k_new, v_new, in_bs, cos_t, sin_t, k_cache, v_cache, kv_ld, kv_bs, o, o_bs, t_dev, t_max, B, H, hd, scale, dtype, stream);
  return e_bf16::decode_step_attn_impl(q, k_new, v_new, in_bs, cos_t, sin_t, k_cache, v_cache, kv_ld, kv_bs, o, o_bs, t_dev, t_max, B, H, hd, scale, dtype, stream);
}

extern "C" int mk_decode_attn(const void* q, const void* k, const void* v, void* o,
                              const int32_t* t_dev, int32_t t_add, int32_t t_max, int32_t B,
                              int32_t H, int32_t hd, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                              int64_t v_ld, int64_t v_bs, int64_t o_bs, float scale, int32_t dtype,
                              void* stream) {
  if (dtype == MK_F16) return e_f16::decode_attn_impl(q, k, v, o, t_dev, t_add, t_max, B, H, hd, q_bs, k_ld, k_bs, v_ld, v_bs, o_bs, scale, dtype, stream);
  return e_bf16::decode_attn_impl(q, k, v, o, t_dev, t_add, t_max, B, H, hd, q_bs, k_ld, k_bs, v_ld, v_bs, o_bs, scale, dtype, stream);
}

extern "C" int mk_decode_emit(const void* logits, int64_t ld, int32_t V, int32_t B, int64_t pad,
                              int64_t eos, int64_t* tok, void* done, int64_t* out, int64_t out_ld,
                              int32_t* state, int32_t dtype, void* stream) {
  if (!logits || !tok || !done || !out || !state || V <= 0 || B <= 0 || ld < V) return MK_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MK_BF16)
    MK_LAUNCH((decode_emit_kernel<bf16>), dim3(B), dim3(1024), 0, st, (const bf16*)logits, (long)ld, V,
              (long)pad, (long)eos, tok, (unsigned char*)done, out, (long)out_ld, state);
  else if (dtype == MK_F16)
    MK_LAUNCH((decode_emit_kernel<_Float16>), dim3(B), dim3(1024), 0, st, (const _Float16*)logits, (long)ld, V,
              (long)pad, (long)eos, tok, (unsigned char*)done, out, (long)out_ld, state);
  else if (dtype == MK_F32)
    MK_LAUNCH((decode_emit_kernel<float>), dim3(B), dim3(1024), 0, st, (const float*)logits, (long)ld, V,
              (long)pad, (long)eos, tok, (unsigned char*)done, out, (long)out_ld, state);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}


extern "C" int mk_kv_append(const void* src, void* cache, int32_t cols, int32_t batch, int64_t s_src,
                            int64_t s_cache, int64_t ld_cache, const int32_t* t_dev, int32_t t_max,
                            int32_t elem_size, void* stream) {
  if (!src || !cache || !t_dev || cols <= 0 || batch <= 0 || t_max <= 0) return MK_ERR_BAD_ARG;
  if (elem_size != 2 && elem_size != 4) return MK_ERR_UNSUPPORTED;
  const long es = elem_size, row_bytes = (long)cols * es;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(cache)) & 15) || (row_bytes % 16) ||
      ((s_src * es) % 16) || ((s_cache * es) % 16) || ((ld_cache * es) % 16))
    return MK_ERR_UNSUPPORTED;
  dim3 grid((unsigned)mk_cdiv(row_bytes, 256L * 16), batch), block(256);
  MK_LAUNCH(kv_append_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), (const char*)src,
            (char*)cache, row_bytes, s_src * es, s_cache * es, ld_cache * es, t_dev, t_max);
  return mk_check_launch();
}

