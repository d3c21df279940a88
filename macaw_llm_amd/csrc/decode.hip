// Greedy-decode kernels whose per-step position comes from DEVICE memory, so that one decode step
// is a fixed launch sequence and can be captured once in a hipGraph and replayed per token
// (modeling.py:190-195,954-960: the reference appends to the KV cache with torch.cat and re-launches
// the whole eager step from Python for every token).
//
//   mk_kv_append    cache[b][*t_dev][0:cols] = src[b][0:cols]   (post-RoPE keys | values of the new token)
//   mk_decode_attn  one query row per (sample, head) against the first *t_dev + t_add cached keys
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

struct DecodeArgs {
  const bf16* q; const bf16* k; const bf16* v; bf16* o;
  const int32_t* t_dev;
  int t_add, t_max;
  long q_bs, k_ld, k_bs, v_ld, v_bs, o_bs;
  float scale;
};

// One workgroup per (head, sample); LPK = HD / 8 lanes share a key (16 bytes each: a key row is
// one coalesced HD * 2-byte segment), 64 / LPK keys per wave and pass, 4 waves.
//   pass 1  scores s_t = scale * q . k_t  -> LDS, running maximum
//   pass 2  p_t = exp(s_t - max), sum
//   pass 3  o = sum_t p_t v_t / sum   (fp32 accumulation; one rounding to bf16)
template <int HD>
__global__ __launch_bounds__(256) void decode_attn_kernel(DecodeArgs a) {
  constexpr int LPK = HD / 8, KPW = 64 / LPK, KPP = 4 * KPW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sc = reinterpret_cast<float*>(smem_raw);          // [t_max] scores / probabilities
  __shared__ float red[4][HD];
  __shared__ float redw[8];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = min(max(*a.t_dev + a.t_add, 1), a.t_max);
  const int sub = lane % LPK, grp = lane / LPK;            // my 8 dims / my key inside the wave's pass
  const bf16* Q = a.q + (long)b * a.q_bs + (long)h * HD + sub * 8;
  const bf16* K = a.k + (long)b * a.k_bs + (long)h * HD + sub * 8;
  const bf16* V = a.v + (long)b * a.v_bs + (long)h * HD + sub * 8;
  float qf[8];
  {
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(Q);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e];
  }
  // ---- pass 1
  float mx = -INFINITY;
  for (int t0 = 0; t0 < T; t0 += KPP) {
    const int t = t0 + wave * KPW + grp;
    float s = 0.f;
    if (t < T) {
      const bf16x8 kv = *reinterpret_cast<const bf16x8*>(K + (long)t * a.k_ld);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += qf[e] * (float)kv[e];
    }
#pragma unroll
    for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
    s *= a.scale;
    if (t < T) {
      if (sub == 0) sc[t] = s;
      mx = fmaxf(mx, s);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) redw[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redw[0], redw[1]), fmaxf(redw[2], redw[3]));
  // ---- pass 2
  float sum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float p = __expf(sc[t] - mx);
    sc[t] = p;
    sum += p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (lane == 0) redw[4 + wave] = sum;
  __syncthreads();
  sum = redw[4] + redw[5] + redw[6] + redw[7];
  // ---- pass 3
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int t0 = 0; t0 < T; t0 += KPP) {
    const int t = t0 + wave * KPW + grp;
    if (t < T) {
      const float p = sc[t];
      const bf16x8 vv = *reinterpret_cast<const bf16x8*>(V + (long)t * a.v_ld);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += p * (float)vv[e];
    }
  }
  // keys of one wave: lanes with the same `sub`
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (grp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][sub * 8 + e] = acc[e];
  }
  __syncthreads();
  if (tid < HD) {
    const float v = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / sum;
    a.o[(long)b * a.o_bs + (long)h * HD + tid] = (bf16)v;
  }
}

}  // namespace

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

extern "C" int mk_decode_attn(const void* q, const void* k, const void* v, void* o,
                              const int32_t* t_dev, int32_t t_add, int32_t t_max, int32_t B,
                              int32_t H, int32_t hd, int64_t q_bs, int64_t k_ld, int64_t k_bs,
                              int64_t v_ld, int64_t v_bs, int64_t o_bs, float scale, int32_t dtype,
                              void* stream) {
  if (!q || !k || !v || !o || !t_dev || B <= 0 || H <= 0 || t_max <= 0) return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (hd != 16 && hd != 32 && hd != 64 && hd != 128)) return MK_ERR_UNSUPPORTED;
  if ((long)t_max * 4 > 60 * 1024) return MK_ERR_UNSUPPORTED;     // scores live in LDS
  const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                       reinterpret_cast<uintptr_t>(v);
  if ((al & 15) || (q_bs % 8) || (k_ld % 8) || (k_bs % 8) || (v_ld % 8) || (v_bs % 8)) return MK_ERR_UNSUPPORTED;
  DecodeArgs a;
  a.q = (const bf16*)q; a.k = (const bf16*)k; a.v = (const bf16*)v; a.o = (bf16*)o;
  a.t_dev = t_dev; a.t_add = t_add; a.t_max = t_max;
  a.q_bs = q_bs; a.k_ld = k_ld; a.k_bs = k_bs; a.v_ld = v_ld; a.v_bs = v_bs; a.o_bs = o_bs;
  a.scale = scale;
  dim3 grid(H, B), block(256);
  const size_t lds = (size_t)t_max * 4;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hd == 128) MK_LAUNCH((decode_attn_kernel<128>), grid, block, lds, st, a);
  else if (hd == 64) MK_LAUNCH((decode_attn_kernel<64>), grid, block, lds, st, a);
  else if (hd == 32) MK_LAUNCH((decode_attn_kernel<32>), grid, block, lds, st, a);
  else MK_LAUNCH((decode_attn_kernel<16>), grid, block, lds, st, a);
  return mk_check_launch();
}
