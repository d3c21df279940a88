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

// The whole attention block of a decode step in ONE launch: RoPE of the new query and key
// (modeling.py:76-91, same rounding points as rope_kernel: each product and the sum rounded to
// bf16), append of the rotated key and the value to cache row p = *t_dev, and the attention of the
// query over keys 0 ... p (the new key / value straight from registers).  NWV waves per (head, sample).
struct DecodeStepArgs {
  const bf16* q; const bf16* kn; const bf16* vn; long in_bs;   // new rows [H * hd] per sample
  const bf16* cos_t; const bf16* sin_t;                       // [positions][hd]
  bf16* kc; bf16* vc; long kv_ld, kv_bs;                      // caches [t_max][H * hd] per sample
  bf16* o; long o_bs;
  const int32_t* t_dev;
  int t_max;
  float scale;
};

MK_DEV float rnd_bf16(float x) { return rnd<bf16>(x); }

template <int HD, int NWV>
__global__ __launch_bounds__(NWV * 64) void decode_step_attn_kernel(DecodeStepArgs a) {
  constexpr int LPK = HD / 8, KPW = 64 / LPK, KPP = NWV * KPW;
  __shared__ float red[NWV][HD];
  __shared__ float redw[2 * NWV];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = min(max(*a.t_dev, 0), a.t_max - 1);
  const int T = p + 1;
  const int sub = lane % LPK, grp = lane / LPK;
  const int d0 = sub * 8;
  const bool first = d0 < HD / 2;
  const long in_off = (long)b * a.in_bs + (long)h * HD + d0;
  float qf[8], kf[8];
  bf16x8 vnew = *reinterpret_cast<const bf16x8*>(a.vn + in_off);
  {
    const bf16x8 qv = *reinterpret_cast<const bf16x8*>(a.q + in_off);
    const bf16x8 kv = *reinterpret_cast<const bf16x8*>(a.kn + in_off);
    const bf16x8 cv = *reinterpret_cast<const bf16x8*>(a.cos_t + (long)p * HD + d0);
    const bf16x8 sv = *reinterpret_cast<const bf16x8*>(a.sin_t + (long)p * HD + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cs = (float)cv[e], sn = (float)sv[e];
      const float qo = (float)qv[e], ko = (float)kv[e];
      const float qp = __shfl_xor(qo, LPK / 2, 64), kp = __shfl_xor(ko, LPK / 2, 64);
      // first half: x cos - partner sin; second half: x cos + partner sin
      const float sq = first ? rnd_bf16(-qp * sn) : rnd_bf16(qp * sn);
      const float sk = first ? rnd_bf16(-kp * sn) : rnd_bf16(kp * sn);
      qf[e] = rnd_bf16(rnd_bf16(qo * cs) + sq);
      kf[e] = rnd_bf16(rnd_bf16(ko * cs) + sk);
    }
  }
  if (wave == 0 && grp == 0) {     // append the new key / value (cache row p)
    bf16x8 kb;
#pragma unroll
    for (int e = 0; e < 8; ++e) kb[e] = (bf16)kf[e];
    const long off = (long)b * a.kv_bs + (long)p * a.kv_ld + (long)h * HD + d0;
    *reinterpret_cast<bf16x8*>(a.kc + off) = kb;
    *reinterpret_cast<bf16x8*>(a.vc + off) = vnew;
  }
  const bf16* K = a.kc + (long)b * a.kv_bs + (long)h * HD + d0;
  const bf16* V = a.vc + (long)b * a.kv_bs + (long)h * HD + d0;
  // Key t belongs to lane group (wave, grp), t = wave * KPW + grp (mod KPP); NB keys per group and
  // trip, their K AND V rows requested together (one memory round trip per KPP * NB keys: a 7B decode
  // step at a few hundred tokens is latency, not bandwidth), online softmax across trips.
  constexpr int NB = 8;
  float m_run = -INFINITY, lsum = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int t0 = wave * KPW + grp; t0 < T; t0 += KPP * NB) {
    bf16x8 kv[NB], vv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = t0 + j * KPP;
      if (t < p) {
        kv[j] = *reinterpret_cast<const bf16x8*>(K + (long)t * a.kv_ld);
        vv[j] = *reinterpret_cast<const bf16x8*>(V + (long)t * a.kv_ld);
      } else {
        vv[j] = vnew;
      }
    }
    float sj[NB], mt = m_run;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = t0 + j * KPP;
      float s = 0.f;
      if (t < p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qf[e] * (float)kv[j][e];
      } else if (t == p) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += qf[e] * kf[e];
      }
#pragma unroll
      for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
      sj[j] = t < T ? s * a.scale : -INFINITY;
      mt = fmaxf(mt, sj[j]);
    }
    const float corr = __expf(m_run - mt);       // (0 on the first trip: m_run = -inf, mt finite)
    lsum *= corr;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= corr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float pr = __expf(sj[j] - mt);       // exp(-inf) = 0 for keys past the end
      lsum += pr;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += pr * (float)vv[j][e];
    }
    m_run = mt;
  }
  // merge the lane groups: common maximum, then rescaled sums
  float mx = m_run;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) redw[wave] = mx;
  __syncthreads();
  mx = redw[0];
#pragma unroll
  for (int i = 1; i < NWV; ++i) mx = fmaxf(mx, redw[i]);
  {
    const float corr = __expf(m_run - mx);       // groups without a key: m_run = -inf -> 0
    lsum *= corr;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= corr;
  }
#pragma unroll
  for (int o = LPK; o < 64; o <<= 1) {
    lsum += __shfl_xor(lsum, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (grp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave][d0 + e] = acc[e];
  }
  if (lane == 0) redw[NWV + wave] = lsum;
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NWV; ++i) sum += redw[NWV + i];
  if (tid < HD) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < NWV; ++i) v += red[i][tid];
    a.o[(long)b * a.o_bs + (long)h * HD + tid] = (bf16)(v / sum);
  }
}

// Greedy token selection and the bookkeeping of a decode step in one launch (HF greedy_search as the
// reference calls it, modeling.py:959: argmax, pad for finished samples, eos marks a sample
// finished): one workgroup per sample; the last workgroup to finish advances the step state.
//   state[0] = position of the token being fed (t_dev of the other decode kernels), state[1] = output
//   column, state[2] = arrival counter (zero between launches).
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

extern "C" int mk_decode_emit(const void* logits, int64_t ld, int32_t V, int32_t B, int64_t pad,
                              int64_t eos, int64_t* tok, void* done, int64_t* out, int64_t out_ld,
                              int32_t* state, int32_t dtype, void* stream) {
  if (!logits || !tok || !done || !out || !state || V <= 0 || B <= 0 || ld < V) return MK_ERR_BAD_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == MK_BF16)
    MK_LAUNCH((decode_emit_kernel<bf16>), dim3(B), dim3(1024), 0, st, (const bf16*)logits, (long)ld, V,
              (long)pad, (long)eos, tok, (unsigned char*)done, out, (long)out_ld, state);
  else if (dtype == MK_F32)
    MK_LAUNCH((decode_emit_kernel<float>), dim3(B), dim3(1024), 0, st, (const float*)logits, (long)ld, V,
              (long)pad, (long)eos, tok, (unsigned char*)done, out, (long)out_ld, state);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_decode_step_attn(const void* q, const void* k_new, const void* v_new, int64_t in_bs,
                                   const void* cos_t, const void* sin_t, void* k_cache, void* v_cache,
                                   int64_t kv_ld, int64_t kv_bs, void* o, int64_t o_bs,
                                   const int32_t* t_dev, int32_t t_max, int32_t B, int32_t H,
                                   int32_t hd, float scale, int32_t dtype, void* stream) {
  if (!q || !k_new || !v_new || !cos_t || !sin_t || !k_cache || !v_cache || !o || !t_dev || B <= 0 ||
      H <= 0 || t_max <= 0)
    return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (hd != 16 && hd != 32 && hd != 64 && hd != 128)) return MK_ERR_UNSUPPORTED;
  const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k_new) |
                       reinterpret_cast<uintptr_t>(v_new) | reinterpret_cast<uintptr_t>(cos_t) |
                       reinterpret_cast<uintptr_t>(sin_t) | reinterpret_cast<uintptr_t>(k_cache) |
                       reinterpret_cast<uintptr_t>(v_cache);
  if ((al & 15) || (in_bs % 8) || (kv_ld % 8) || (kv_bs % 8)) return MK_ERR_UNSUPPORTED;
  DecodeStepArgs a;
  a.q = (const bf16*)q; a.kn = (const bf16*)k_new; a.vn = (const bf16*)v_new; a.in_bs = in_bs;
  a.cos_t = (const bf16*)cos_t; a.sin_t = (const bf16*)sin_t;
  a.kc = (bf16*)k_cache; a.vc = (bf16*)v_cache; a.kv_ld = kv_ld; a.kv_bs = kv_bs;
  a.o = (bf16*)o; a.o_bs = o_bs;
  a.t_dev = t_dev; a.t_max = t_max; a.scale = scale;
  dim3 grid(H, B), block(512);
  const size_t lds = 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hd == 128) MK_LAUNCH((decode_step_attn_kernel<128, 8>), grid, block, lds, st, a);
  else if (hd == 64) MK_LAUNCH((decode_step_attn_kernel<64, 8>), grid, block, lds, st, a);
  else if (hd == 32) MK_LAUNCH((decode_step_attn_kernel<32, 8>), grid, block, lds, st, a);
  else MK_LAUNCH((decode_step_attn_kernel<16, 8>), grid, block, lds, st, a);
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
