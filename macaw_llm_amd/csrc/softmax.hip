// Row softmax with the reference's masking semantics, its backward, attention
// dropout, shifted cross-entropy, and the fused AdamW update.  All HBM-bound.
//
// Masking (modeling.py:205-214, 44-73): masked logits take finfo(dtype).min —
// exactly what `scores + mask` followed by `max(., finfo.min)` produces — and
// the softmax itself runs in fp32.  A fully masked row therefore degenerates to
// a uniform distribution, not NaN.
#include "common.h"
#include "../../include/macaw_hip.h"

namespace {

template <typename T> MK_DEV float finfo_min();
template <> MK_DEV float finfo_min<float>() { return -3.4028234663852886e38f; }
template <> MK_DEV float finfo_min<bf16>() { return -3.3895313892515355e38f; }
template <> MK_DEV float finfo_min<_Float16>() { return -65504.f; }   // torch.finfo(torch.float16).min

MK_DEV uint32_t keep_thr(float p) {
  const double k = (1.0 - (double)p) * 4294967296.0;
  return k >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)k;
}

// masked logit of key k for query q (reference semantics, see header comment)
template <typename T>
MK_DEV float masked(float s, int k, int klim, const int32_t* km) {
  return (k > klim || (km && km[k] == 0)) ? finfo_min<T>() : s;
}

// One wave per row, 16-byte vector loads, the whole row held in registers
// (Lk <= 64 * N * MAXC: 2048 for bf16 at MAXC = 4).  Pad columns [Lk, ld) are written as 0.
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(
    const T* scores, T* probs, T* probs_drop, const int32_t* kmask, long nrows, int heads, int Lq,
    int Lk, long ld, int causal, float p, uint64_t seed, const uint64_t* seed_dev) {
  constexpr int N = VecIO<T>::N;
  if (seed_dev) seed += *seed_dev;       // step offset in device memory (mk_set_dropout_seed_offset)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int q = (int)(row % Lq);
  const long b = (row / Lq) / heads;
  const T* sr = scores + row * ld;
  const int32_t* km = kmask ? kmask + b * Lk : nullptr;
  const int klim = causal ? q + (Lk - Lq) : Lk - 1;
  const int nch = (int)(ld / N);
  float v[MAXC][N];
  float mx = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      VecIO<T>::load(sr + ch * N, v[c]);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int k = ch * N + i;
        v[c][i] = (k < Lk) ? masked<T>(v[c][i], k, klim, km) : -INFINITY;
        mx = fmaxf(mx, v[c][i]);
      }
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        v[c][i] = (ch * N + i < Lk) ? __expf(v[c][i] - mx) : 0.f;
        sum += v[c][i];
      }
    }
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const uint32_t thr = keep_thr(p);
  const float dscale = p > 0.f ? 1.f / (1.f - p) : 1.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      float o[N];
#pragma unroll
      for (int i = 0; i < N; ++i) o[i] = rnd<T>(v[c][i] * inv);
      VecIO<T>::store(probs + row * ld + ch * N, o);
      if (probs_drop) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int k = ch * N + i;
          o[i] = (k < Lk && mk_keep(seed, (uint64_t)row * (uint64_t)Lk + k, thr)) ? o[i] * dscale : 0.f;
        }
        VecIO<T>::store(probs_drop + row * ld + ch * N, o);
      }
    }
  }
}

// One 256-thread block per row (long rows, e.g. the 32,009-key alignment attention): three
// vectorised passes (max, sum, write); the row (<= 64 KiB) stays in L2 between passes.
// (Round 4, measured and removed: the row held in registers between the passes -- 16 raw 16-byte chunks per
// thread, read once -- is SLOWER, forward 231 vs 195 us and backward 276 vs 190 us at 3072 rows x 32,009: at
// 114 / 178 VGPRs fewer rows are in flight per CU, and the second and third pass hit L2 anyway.)
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_block_kernel(
    const T* scores, T* probs, T* probs_drop, const int32_t* kmask, int heads, int Lq, int Lk,
    long ld, int causal, float p, uint64_t seed, const uint64_t* seed_dev) {
  if (seed_dev) seed += *seed_dev;
  constexpr int N = VecIO<T>::N;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const int q = (int)(row % Lq);
  const long b = (row / Lq) / heads;
  const T* sr = scores + row * ld;
  const int32_t* km = kmask ? kmask + b * Lk : nullptr;
  const int klim = causal ? q + (Lk - Lq) : Lk - 1;
  const int nch = (int)(ld / N);
  float mx = -INFINITY;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    float v[N];
    VecIO<T>::load(sr + ch * N, v);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (ch * N + i < Lk) mx = fmaxf(mx, masked<T>(v[i], ch * N + i, klim, km));
  }
  mx = block_max<256>(mx, red);
  float sum = 0.f;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    float v[N];
    VecIO<T>::load(sr + ch * N, v);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (ch * N + i < Lk) sum += __expf(masked<T>(v[i], ch * N + i, klim, km) - mx);
  }
  sum = block_sum<256>(sum, red);
  const float inv = 1.f / sum;
  const uint32_t thr = keep_thr(p);
  const float dscale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int ch = threadIdx.x; ch < nch; ch += 256) {
    float v[N], o[N];
    VecIO<T>::load(sr + ch * N, v);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int k = ch * N + i;
      o[i] = (k < Lk) ? rnd<T>(__expf(masked<T>(v[i], k, klim, km) - mx) * inv) : 0.f;
    }
    VecIO<T>::store(probs + row * ld + ch * N, o);
    if (probs_drop) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int k = ch * N + i;
        o[i] = (k < Lk && mk_keep(seed, (uint64_t)row * (uint64_t)Lk + k, thr)) ? o[i] * dscale : 0.f;
      }
      VecIO<T>::store(probs_drop + row * ld + ch * N, o);
    }
  }
}

// dS = P .* (g - sum(g .* P)) * scale,  g = dP_drop .* keep/(1-p); in place on dprobs.
// WAVE = true: one wave per row, row in registers; else one block per row, two passes.
template <typename T, int MAXC, bool WAVE>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* probs, T* dprobs, int Lk,
                                                          long ld, float scale, float p,
                                                          uint64_t seed, long nrows, const uint64_t* seed_dev) {
  if (seed_dev) seed += *seed_dev;
  constexpr int N = VecIO<T>::N;
  __shared__ float red[16];
  const uint32_t thr = keep_thr(p);
  const float dscale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int nch = (int)(ld / N);
  if constexpr (WAVE) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    float pv[MAXC][N], gv[MAXC][N];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        VecIO<T>::load(probs + row * ld + ch * N, pv[c]);
        VecIO<T>::load(dprobs + row * ld + ch * N, gv[c]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int k = ch * N + i;
          float g = (k < Lk) ? gv[c][i] : 0.f;
          if (p > 0.f) g = (k < Lk && mk_keep(seed, (uint64_t)row * (uint64_t)Lk + k, thr)) ? g * dscale : 0.f;
          gv[c][i] = g;
          dot += (k < Lk) ? g * pv[c][i] : 0.f;
        }
      }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[N];
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = (ch * N + i < Lk) ? pv[c][i] * (gv[c][i] - dot) * scale : 0.f;
        VecIO<T>::store(dprobs + row * ld + ch * N, o);
      }
    }
  } else {
    const long row = blockIdx.x;
    float dot = 0.f;
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
      float pv[N], gv[N];
      VecIO<T>::load(probs + row * ld + ch * N, pv);
      VecIO<T>::load(dprobs + row * ld + ch * N, gv);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int k = ch * N + i;
        if (k < Lk) {
          float g = gv[i];
          if (p > 0.f) g = mk_keep(seed, (uint64_t)row * (uint64_t)Lk + k, thr) ? g * dscale : 0.f;
          dot += g * pv[i];
        }
      }
    }
    dot = block_sum<256>(dot, red);
    for (int ch = threadIdx.x; ch < nch; ch += 256) {
      float pv[N], gv[N], o[N];
      VecIO<T>::load(probs + row * ld + ch * N, pv);
      VecIO<T>::load(dprobs + row * ld + ch * N, gv);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int k = ch * N + i;
        float g = gv[i];
        if (p > 0.f) g = (k < Lk && mk_keep(seed, (uint64_t)row * (uint64_t)Lk + k, thr)) ? g * dscale : 0.f;
        o[i] = (k < Lk) ? pv[i] * (g - dot) * scale : 0.f;
      }
      VecIO<T>::store(dprobs + row * ld + ch * N, o);
    }
  }
}

// ---------------------------------------------------------- cross entropy --
// row_stat[r] = {lse} ; row_loss[r] = lse - logit[label] (0 when ignored)
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* logits, const int64_t* labels,
                                                     float* row_loss, float* row_lse, int V,
                                                     long ld) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const T* lr = logits + row * ld;
  const long lab = labels[row];
  if (lab < 0 || lab >= V) {  // ignore_index (-100) or out of range: no contribution
    if (threadIdx.x == 0) { row_loss[row] = 0.f; row_lse[row] = 0.f; }
    return;
  }
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, to_f32<T>(lr[c]));
  mx = block_max<256>(mx, red);
  float sum = 0.f;
  for (int c = threadIdx.x; c < V; c += 256) sum += __expf(to_f32<T>(lr[c]) - mx);
  sum = block_sum<256>(sum, red);
  if (threadIdx.x == 0) {
    const float lse = mx + __logf(sum);
    row_lse[row] = lse;
    row_loss[row] = lse - to_f32<T>(lr[lab]);
  }
}
// Deterministic single-block reduction: out = {sum loss, n_valid, mean loss, 0}
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss,
                                                        const int64_t* labels, float* out,
                                                        int rows, int V) {
  __shared__ float red[16];
  float s = 0.f, n = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const long lab = labels[r];
    if (lab >= 0 && lab < V) { s += row_loss[r]; n += 1.f; }
  }
  s = block_sum<256>(s, red);
  n = block_sum<256>(n, red);
  if (threadIdx.x == 0) { out[0] = s; out[1] = n; out[2] = n > 0.f ? s / n : 0.f; out[3] = 0.f; }
}
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* logits, T* dlogits,
                                                     const int64_t* labels, const float* row_lse,
                                                     const float* sum_cnt, float grad_scale,
                                                     const float* grad_scale_dev, int V, long ld) {
  const long row = blockIdx.x;
  const T* lr = logits + row * ld;
  T* dr = dlogits + row * ld;
  const long lab = labels[row];
  const bool valid = lab >= 0 && lab < V;
  const float n = sum_cnt[1];
  if (grad_scale_dev) grad_scale *= *grad_scale_dev;
  const float gs = (valid && n > 0.f) ? grad_scale / n : 0.f;
  const float lse = row_lse[row];
  for (int c = threadIdx.x; c < (int)ld; c += 256) {
    float g = 0.f;
    if (valid && c < V) g = (__expf(to_f32<T>(lr[c]) - lse) - (c == lab ? 1.f : 0.f)) * gs;
    dr[c] = from_f32<T>(g);
  }
}

// --------------------------------------------------------------- argmax ----
// Greedy token selection (HF greedy_search, modeling.py:959): index of the first maximum of
// every row; fp32 compare, ties -> lowest index (torch.argmax semantics).
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* x, long ld, int cols,
                                                          int64_t* out) {
  __shared__ float bv[4];
  __shared__ int bi[4];
  const long row = blockIdx.x;
  const T* xr = x + row * ld;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float v = to_f32<T>(xr[c]);
    if (v > best || (v == best && c < idx)) { best = v; idx = c; }
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
    for (int w = 1; w < 4; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    out[row] = idx;
  }
}

// ------------------------------------------------------------------ AdamW --
// 4-element load / store in the parameter dtype (8 B for bf16, 16 B for fp32)
MK_DEV void adam_load4(const bf16* p, float (&o)[4]) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)v[k];
}
MK_DEV void adam_load4(const _Float16* p, float (&o)[4]) {
  const f16x4 v = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)v[k];
}
MK_DEV void adam_load4(const float* p, float (&o)[4]) { VecIO<float>::load(p, o); }
MK_DEV void adam_store4(bf16* p, const float (&o)[4]) {
  bf16x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (bf16)o[k];
  *reinterpret_cast<bf16x4*>(p) = v;
}
MK_DEV void adam_store4(_Float16* p, const float (&o)[4]) {
  f16x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (_Float16)o[k];
  *reinterpret_cast<f16x4*>(p) = v;
}
MK_DEV void adam_store4(float* p, const float (&o)[4]) { VecIO<float>::store(p, o); }
// the same with the NON-TEMPORAL hint: used for the GRADIENT of the multi-tensor kernel only (read once per step, written by
// GEMM kernels of this device long before).  scripts/probe/adamw_stream.hip: the hinted loads are worth +1.2 ... +2.2 % of the
// stream; hinted stores of the 16-bit parameter copy added +0.3 % in the harness and are NOT used: with them in the per-slice
// kernel (whose output an all-gather stages through the copy engine right behind it) the retired runtime's world-2 test came out
// with one stale ZeRO-1 slice in 3 of 10 full-suite runs beside a busy second stream, and in 0 of 5 without them
// (profiles/r06_dw_side_stream.txt "World 2"; suggestive only -- an isolated probe does not reproduce it -- but 0.3 % is not
// worth the question).
// (All four fp32 streams non-temporal measured 4.0 TB/s in rounds 1 and 3.)
typedef unsigned int adam_u32x2 __attribute__((ext_vector_type(2)));
typedef float adam_f32x4 __attribute__((ext_vector_type(4)));
MK_DEV void adam_load4_nt(const bf16* p, float (&o)[4]) {
  const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_nontemporal_load(reinterpret_cast<const adam_u32x2*>(p)));
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)v[k];
}
MK_DEV void adam_load4_nt(const _Float16* p, float (&o)[4]) {
  const f16x4 v = __builtin_bit_cast(f16x4, __builtin_nontemporal_load(reinterpret_cast<const adam_u32x2*>(p)));
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)v[k];
}
MK_DEV void adam_load4_nt(const float* p, float (&o)[4]) {
  const adam_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const adam_f32x4*>(p));
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = v[k];
}

// 16-byte vector form: N = 8 (bf16) / 4 (fp32) parameters per thread and iteration
// (28 B/param of HBM traffic: the kernel is a pure stream, cdna_hip_programming.md G13).
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* param, float* master, float* m, float* v,
                                                    const T* grad, long n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1,
                                                    float bc2, float gscale) {
  // 4 consecutive elements per thread and iteration: every wave instruction then covers
  // CONTIGUOUS memory (fp32 arrays 16 B per lane = 1 KiB, bf16 arrays 8 B per lane = 512 B).  The
  // former 8-element form read / wrote the fp32 state as two float4 32 bytes apart per lane, i.e.
  // half-sector stores: 116 GB written per step for 99 GB of state (rocprofv3 WRITE_SIZE,
  // calibrated on a memset) at 5.2 TB/s.
  constexpr int N = 4;
  const long nch = n / N;
  const float ib1 = 1.f / bc1, ib2 = 1.f / bc2;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
    const long i0 = c * N;
    float g[N], w[N], mi[N], vi[N];
    adam_load4(grad + i0, g);      // (plain accesses here: this kernel's output feeds a collective, see adam_load4_nt)
    VecIO<float>::load(master + i0, w);
    VecIO<float>::load(m + i0, mi);
    VecIO<float>::load(v + i0, vi);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gk = g[k] * gscale;
      mi[k] = b1 * mi[k] + (1.f - b1) * gk;
      vi[k] = b2 * vi[k] + (1.f - b2) * gk * gk;
      float wk = w[k];
      wk -= lr * wd * wk;  // decoupled weight decay
      wk -= lr * (mi[k] * ib1) / (sqrtf(vi[k] * ib2) + eps);
      w[k] = wk;
    }
    VecIO<float>::store(master + i0, w);
    VecIO<float>::store(m + i0, mi);
    VecIO<float>::store(v + i0, vi);
    adam_store4(param + i0, w);
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float g = to_f32<T>(grad[i]) * gscale;
    float w = master[i];
    const float mi = b1 * m[i] + (1.f - b1) * g;
    const float vi = b2 * v[i] + (1.f - b2) * g * g;
    m[i] = mi; v[i] = vi;
    w -= lr * wd * w;
    w -= lr * (mi * ib1) / (sqrtf(vi * ib2) + eps);
    master[i] = w;
    param[i] = from_f32<T>(w);
  }
}


// Multi-tensor form: ONE launch updates every parameter tensor of the step (311 tensors at 7B:
// 311 launch ramps and ~130 sub-4-microsecond kernels for the norm weights otherwise).  A block
// owns one MK_ADAMW_CHUNK-element slice of one tensor, found by binary search in the chunk
// prefix; per-element arithmetic is exactly adamw_kernel's.
struct AdamItem { void* param; float* master; float* m; float* v; const void* grad; long n; };
// Slice size (round 6, scripts/probe/adamw_stream.hip): 32768-element slices left the ~2000 resident blocks spread over
// 2000 x 128 KiB of each of the eight streams; with 4096 the blocks in flight cover a window an eighth as wide (consecutive
// block ids = consecutive slices) and the stream gains 4-6 % on both a slow (5.54 -> 5.88 TB/s) and a fast box (6.08 -> 6.31),
// together with the hinted gradient loads above.  mk_adamw_chunk() tells the host which slice size its chunk table must use.
#ifndef MK_ADAMW_CHUNK_ELEMS
#define MK_ADAMW_CHUNK_ELEMS 4096
#endif
constexpr long MK_ADAMW_CHUNK = MK_ADAMW_CHUNK_ELEMS;

template <typename T>
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamItem* items, const long* chunk_start,
                                                          int n_items, float lr, float b1, float b2,
                                                          float eps, float wd, float bc1, float bc2,
                                                          float gscale, const float* hyper) {
  if (hyper) {                              // (lr, bc1, bc2, grad_scale) of THIS step, device memory
    lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; gscale = hyper[3];
  }
  int lo = 0, hi = n_items;                 // last item whose first chunk <= blockIdx.x
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= (long)blockIdx.x) lo = mid; else hi = mid;
  }
  const AdamItem it = items[lo];
  const long base = ((long)blockIdx.x - chunk_start[lo]) * MK_ADAMW_CHUNK;
  const long end = min(base + MK_ADAMW_CHUNK, it.n);
  T* param = reinterpret_cast<T*>(it.param);
  const T* grad = reinterpret_cast<const T*>(it.grad);
  constexpr int N = 4;                      // see adamw_kernel: wave-contiguous accesses
  const float ib1 = 1.f / bc1, ib2 = 1.f / bc2;
  const long vec_end = base + (end - base) / N * N;
  // two 4-element groups 256 * 4 elements apart per iteration: 8 independent 16-byte loads in
  // flight per lane before the first use
  for (long i0 = base + (long)threadIdx.x * N; i0 < vec_end; i0 += 512L * N) {
    const long i1 = i0 + 256L * N;
    const bool two = i1 < vec_end;             // (a chunk is 32768 elements: true except on a ragged tail)
    const long j1 = two ? i1 : i0;
    float g0[N], w0[N], m0[N], v0[N], g1[N], w1[N], m1[N], v1[N];
    adam_load4_nt(grad + i0, g0);
    adam_load4_nt(grad + j1, g1);
    VecIO<float>::load(it.master + i0, w0);
    VecIO<float>::load(it.master + j1, w1);
    VecIO<float>::load(it.m + i0, m0);
    VecIO<float>::load(it.m + j1, m1);
    VecIO<float>::load(it.v + i0, v0);
    VecIO<float>::load(it.v + j1, v1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gk = g0[k] * gscale;
      m0[k] = b1 * m0[k] + (1.f - b1) * gk;
      v0[k] = b2 * v0[k] + (1.f - b2) * gk * gk;
      float wk = w0[k];
      wk -= lr * wd * wk;
      wk -= lr * (m0[k] * ib1) / (sqrtf(v0[k] * ib2) + eps);
      w0[k] = wk;
    }
    VecIO<float>::store(it.master + i0, w0);
    VecIO<float>::store(it.m + i0, m0);
    VecIO<float>::store(it.v + i0, v0);
    adam_store4(param + i0, w0);
    if (two) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gk = g1[k] * gscale;
        m1[k] = b1 * m1[k] + (1.f - b1) * gk;
        v1[k] = b2 * v1[k] + (1.f - b2) * gk * gk;
        float wk = w1[k];
        wk -= lr * wd * wk;
        wk -= lr * (m1[k] * ib1) / (sqrtf(v1[k] * ib2) + eps);
        w1[k] = wk;
      }
      VecIO<float>::store(it.master + i1, w1);
      VecIO<float>::store(it.m + i1, m1);
      VecIO<float>::store(it.v + i1, v1);
      adam_store4(param + i1, w1);
    }
  }
  for (long i = vec_end + threadIdx.x; i < end; i += 256) {
    const float g = to_f32<T>(grad[i]) * gscale;
    float w = it.master[i];
    const float mi = b1 * it.m[i] + (1.f - b1) * g;
    const float vi = b2 * it.v[i] + (1.f - b2) * g * g;
    it.m[i] = mi; it.v[i] = vi;
    w -= lr * wd * w;
    w -= lr * (mi * ib1) / (sqrtf(vi * ib2) + eps);
    it.master[i] = w;
    param[i] = from_f32<T>(w);
  }
}

}  // namespace

#define MK_ST reinterpret_cast<hipStream_t>(stream)

namespace { const uint64_t* g_seed_dev = nullptr; }
// Training steps replayed from a hipGraph cannot change the seed ARGUMENT of the dropout kernels:
// with a device pointer registered here every mk_softmax_fwd / mk_softmax_bwd launch adds *ptr to
// its seed when it executes (the graph advances the value once per step).  NULL switches it off.
extern "C" int mk_set_dropout_seed_offset(const uint64_t* dev_ptr) {
  g_seed_dev = dev_ptr;
  return MK_OK;
}

namespace {
template <typename T>
int softmax_fwd_t(const void* scores, void* probs, void* probs_drop, const int32_t* kmask,
                  int nz, int heads, int Lq, int Lk, long ld, int causal, float p, uint64_t seed,
                  hipStream_t st) {
  constexpr int N = VecIO<T>::N;
  const long nrows = (long)nz * Lq;
  const long nch = ld / N;
  if (nch <= 64 * 8) {
    dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
#define MK_SW(E)                                                                               \
  MK_LAUNCH((softmax_fwd_wave_kernel<T, E>), grid, block, 0, st, (const T*)scores, (T*)probs,   \
            (T*)probs_drop, kmask, nrows, heads, Lq, Lk, ld, causal, p, seed, g_seed_dev)
    if (nch <= 64) MK_SW(1);
    else if (nch <= 128) MK_SW(2);
    else if (nch <= 256) MK_SW(4);
    else MK_SW(8);
#undef MK_SW
  } else {
    MK_LAUNCH((softmax_fwd_block_kernel<T>), dim3((unsigned)nrows), dim3(256), 0, st,
              (const T*)scores, (T*)probs, (T*)probs_drop, kmask, heads, Lq, Lk, ld, causal, p,
              seed, g_seed_dev);
  }
  return mk_check_launch();
}
template <typename T>
int softmax_bwd_t(const void* probs, void* dprobs, int nz, int Lq, int Lk, long ld, float scale,
                  float p, uint64_t seed, hipStream_t st) {
  constexpr int N = VecIO<T>::N;
  const long nrows = (long)nz * Lq;
  const long nch = ld / N;
  if (nch <= 64 * 4) {
    dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
#define MK_SB(E)                                                                               \
  MK_LAUNCH((softmax_bwd_kernel<T, E, true>), grid, block, 0, st, (const T*)probs, (T*)dprobs,  \
            Lk, ld, scale, p, seed, nrows, g_seed_dev)
    if (nch <= 64) MK_SB(1);
    else if (nch <= 128) MK_SB(2);
    else MK_SB(4);
#undef MK_SB
  } else {
    MK_LAUNCH((softmax_bwd_kernel<T, 1, false>), dim3((unsigned)nrows), dim3(256), 0, st,
              (const T*)probs, (T*)dprobs, Lk, ld, scale, p, seed, nrows, g_seed_dev);
  }
  return mk_check_launch();
}
}  // namespace

extern "C" int mk_softmax_fwd(const void* scores, void* probs, void* probs_drop,
                              const int32_t* kmask, int32_t nz, int32_t heads, int32_t Lq,
                              int32_t Lk, int64_t ld, int32_t causal, float dropout_p,
                              uint64_t seed, int32_t dtype, void* stream) {
  if (!scores || !probs || nz <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0 || ld < Lk)
    return MK_ERR_BAD_ARG;
  if (dropout_p < 0.f || dropout_p >= 1.f) return MK_ERR_BAD_ARG;
  if (dropout_p == 0.f) probs_drop = nullptr;
  const uintptr_t al = reinterpret_cast<uintptr_t>(scores) | reinterpret_cast<uintptr_t>(probs) |
                       reinterpret_cast<uintptr_t>(probs_drop);
  if ((al & 15) || ld % 8) return MK_ERR_UNSUPPORTED;  // rows must be 16-byte aligned, pitch % 8
  if (dtype == MK_BF16)
    return softmax_fwd_t<bf16>(scores, probs, probs_drop, kmask, nz, heads, Lq, Lk, ld, causal,
                               dropout_p, seed, MK_ST); else if (dtype == MK_F16)
    return softmax_fwd_t<_Float16>(scores, probs, probs_drop, kmask, nz, heads, Lq, Lk, ld, causal,
                               dropout_p, seed, MK_ST);
  if (dtype == MK_F32)
    return softmax_fwd_t<float>(scores, probs, probs_drop, kmask, nz, heads, Lq, Lk, ld, causal,
                                dropout_p, seed, MK_ST);
  return MK_ERR_UNSUPPORTED;
}

extern "C" int mk_softmax_bwd(const void* probs, void* dprobs, int32_t nz, int32_t Lq, int32_t Lk,
                              int64_t ld, float scale, float dropout_p, uint64_t seed,
                              int32_t dtype, void* stream) {
  if (!probs || !dprobs || nz <= 0 || Lq <= 0 || Lk <= 0 || ld < Lk) return MK_ERR_BAD_ARG;
  const uintptr_t al = reinterpret_cast<uintptr_t>(probs) | reinterpret_cast<uintptr_t>(dprobs);
  if ((al & 15) || ld % 8) return MK_ERR_UNSUPPORTED;
  if (dtype == MK_BF16)
    return softmax_bwd_t<bf16>(probs, dprobs, nz, Lq, Lk, ld, scale, dropout_p, seed, MK_ST); else if (dtype == MK_F16)
    return softmax_bwd_t<_Float16>(probs, dprobs, nz, Lq, Lk, ld, scale, dropout_p, seed, MK_ST);
  if (dtype == MK_F32)
    return softmax_bwd_t<float>(probs, dprobs, nz, Lq, Lk, ld, scale, dropout_p, seed, MK_ST);
  return MK_ERR_UNSUPPORTED;
}

extern "C" int mk_cross_entropy(const void* logits, const int64_t* labels, float* row_loss,
                                float* row_lse, float* loss_sum_cnt, int32_t rows, int32_t V,
                                int64_t ld, int32_t dtype, void* stream) {
  if (!logits || !labels || !row_loss || !row_lse || !loss_sum_cnt || rows <= 0 || V <= 0 ||
      ld < V)
    return MK_ERR_BAD_ARG;
  if (dtype == MK_BF16)
    MK_LAUNCH((ce_fwd_kernel<bf16>), dim3(rows), dim3(256), 0, MK_ST, (const bf16*)logits,
                       labels, row_loss, row_lse, V, (long)ld); else if (dtype == MK_F16)
    MK_LAUNCH((ce_fwd_kernel<_Float16>), dim3(rows), dim3(256), 0, MK_ST, (const _Float16*)logits,
                       labels, row_loss, row_lse, V, (long)ld);
  else if (dtype == MK_F32)
    MK_LAUNCH((ce_fwd_kernel<float>), dim3(rows), dim3(256), 0, MK_ST,
                       (const float*)logits, labels, row_loss, row_lse, V, (long)ld);
  else return MK_ERR_UNSUPPORTED;
  MK_LAUNCH(ce_reduce_kernel, dim3(1), dim3(256), 0, MK_ST, row_loss, labels,
                     loss_sum_cnt, rows, V);
  return mk_check_launch();
}

extern "C" int mk_cross_entropy_bwd(const void* logits, void* dlogits, const int64_t* labels,
                                    const float* row_lse, const float* loss_sum_cnt,
                                    float grad_scale, const float* grad_scale_dev, int32_t rows,
                                    int32_t V, int64_t ld, int32_t dtype, void* stream) {
  if (!logits || !dlogits || !labels || !row_lse || !loss_sum_cnt || rows <= 0 || V <= 0 ||
      ld < V)
    return MK_ERR_BAD_ARG;
  if (dtype == MK_BF16)
    MK_LAUNCH((ce_bwd_kernel<bf16>), dim3(rows), dim3(256), 0, MK_ST, (const bf16*)logits,
                       (bf16*)dlogits, labels, row_lse, loss_sum_cnt, grad_scale, grad_scale_dev, V,
                       (long)ld); else if (dtype == MK_F16)
    MK_LAUNCH((ce_bwd_kernel<_Float16>), dim3(rows), dim3(256), 0, MK_ST, (const _Float16*)logits,
                       (_Float16*)dlogits, labels, row_lse, loss_sum_cnt, grad_scale, grad_scale_dev, V,
                       (long)ld);
  else if (dtype == MK_F32)
    MK_LAUNCH((ce_bwd_kernel<float>), dim3(rows), dim3(256), 0, MK_ST,
                       (const float*)logits, (float*)dlogits, labels, row_lse, loss_sum_cnt,
                       grad_scale, grad_scale_dev, V, (long)ld);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

// Upper bound on mk_adamw's grid (0 = the default 2048 blocks of 256 threads).  The kernel is a grid-stride stream,
// so a small grid makes it a PERSISTENT update that holds only the CUs its blocks fit on (62 VGPRs: 8 blocks of four
// waves per CU) -- how BucketedStep(local_overlap) runs the per-bucket updates of ONE rank beside the backward's GEMMs
// on the CUs the GEMMs were told to leave (mk_gemm_set_cus), instead of taking every CU a finishing tile frees.
static int g_adamw_max_blocks = 0;
extern "C" int mk_adamw_set_max_blocks(int32_t n) {
  const int prev = g_adamw_max_blocks;
  g_adamw_max_blocks = n > 0 ? n : 0;
  return prev;
}

extern "C" int mk_adamw(void* param, float* master, float* m, float* v, const void* grad,
                        int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int32_t step, float grad_scale, int32_t dtype,
                        void* stream) {
  if (!param || !master || !m || !v || !grad || n <= 0 || step < 1) return MK_ERR_BAD_ARG;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  // the vector path needs 16-byte aligned bases (row-slice views of fused storage keep that)
  const uintptr_t al = reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(master) |
                       reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                       reinterpret_cast<uintptr_t>(grad);
  if (al & 15) return MK_ERR_UNSUPPORTED;
  long nb = (n / 8 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (g_adamw_max_blocks > 0 && nb > g_adamw_max_blocks) nb = g_adamw_max_blocks;
  if (nb < 1) nb = 1;
  dim3 grid((unsigned)nb), block(256);
  if (dtype == MK_BF16)
    MK_LAUNCH((adamw_kernel<bf16>), grid, block, 0, MK_ST, (bf16*)param, master, m, v,
                       (const bf16*)grad, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       grad_scale); else if (dtype == MK_F16)
    MK_LAUNCH((adamw_kernel<_Float16>), grid, block, 0, MK_ST, (_Float16*)param, master, m, v,
                       (const _Float16*)grad, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       grad_scale);
  else if (dtype == MK_F32)
    MK_LAUNCH((adamw_kernel<float>), grid, block, 0, MK_ST, (float*)param, master, m, v,
                       (const float*)grad, (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       grad_scale);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_argmax_rows(const void* x, int64_t ld, int32_t rows, int32_t cols, int64_t* out,
                              int32_t dtype, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return MK_ERR_BAD_ARG;
  if (dtype == MK_BF16)
    MK_LAUNCH((argmax_rows_kernel<bf16>), dim3(rows), dim3(256), 0, MK_ST, (const bf16*)x, (long)ld,
              cols, out); else if (dtype == MK_F16)
    MK_LAUNCH((argmax_rows_kernel<_Float16>), dim3(rows), dim3(256), 0, MK_ST, (const _Float16*)x, (long)ld,
              cols, out);
  else if (dtype == MK_F32)
    MK_LAUNCH((argmax_rows_kernel<float>), dim3(rows), dim3(256), 0, MK_ST, (const float*)x,
              (long)ld, cols, out);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_adamw_chunk(void) { return (int)MK_ADAMW_CHUNK; }

extern "C" int mk_adamw_multi(const void* items, const int64_t* chunk_start, int32_t n_items,
                              int64_t n_chunks, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int32_t step, float grad_scale, int32_t dtype,
                              void* stream) {
  if (!items || !chunk_start || n_items <= 0 || n_chunks <= 0 || step < 1) return MK_ERR_BAD_ARG;
  static_assert(sizeof(AdamItem) == 48 && sizeof(long) == sizeof(int64_t), "item = 6 x 8 bytes");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  const dim3 grid((unsigned)n_chunks), block(256);
  const AdamItem* it = reinterpret_cast<const AdamItem*>(items);
  const long* cs = reinterpret_cast<const long*>(chunk_start);
  if (dtype == MK_BF16)
    MK_LAUNCH((adamw_multi_kernel<bf16>), grid, block, 0, MK_ST, it, cs, n_items, lr, beta1, beta2, eps,
              weight_decay, bc1, bc2, grad_scale, (const float*)nullptr); else if (dtype == MK_F16)
    MK_LAUNCH((adamw_multi_kernel<_Float16>), grid, block, 0, MK_ST, it, cs, n_items, lr, beta1, beta2, eps,
              weight_decay, bc1, bc2, grad_scale, (const float*)nullptr);
  else if (dtype == MK_F32)
    MK_LAUNCH((adamw_multi_kernel<float>), grid, block, 0, MK_ST, it, cs, n_items, lr, beta1, beta2, eps,
              weight_decay, bc1, bc2, grad_scale, (const float*)nullptr);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

// mk_adamw_multi with the per-step scalars in DEVICE memory: hyper_dev = {lr, 1 - beta1^step,
// 1 - beta2^step, grad_scale}, so a captured launch (hipGraph replay of a training step) follows the
// learning-rate schedule and the bias correction.  mk_adamw_bias_correction computes the two
// corrections with the arithmetic mk_adamw_multi uses on the host (bit-identical updates).
extern "C" int mk_adamw_bias_correction(float beta1, float beta2, int32_t step, float* out2) {
  if (!out2 || step < 1) return MK_ERR_BAD_ARG;
  out2[0] = 1.f - powf(beta1, (float)step);
  out2[1] = 1.f - powf(beta2, (float)step);
  return MK_OK;
}
extern "C" int mk_adamw_multi_dev(const void* items, const int64_t* chunk_start, int32_t n_items,
                                  int64_t n_chunks, float beta1, float beta2, float eps,
                                  float weight_decay, const float* hyper_dev, int32_t dtype,
                                  void* stream) {
  if (!items || !chunk_start || !hyper_dev || n_items <= 0 || n_chunks <= 0) return MK_ERR_BAD_ARG;
  const dim3 grid((unsigned)n_chunks), block(256);
  const AdamItem* it = reinterpret_cast<const AdamItem*>(items);
  const long* cs = reinterpret_cast<const long*>(chunk_start);
  if (dtype == MK_BF16)
    MK_LAUNCH((adamw_multi_kernel<bf16>), grid, block, 0, MK_ST, it, cs, n_items, 0.f, beta1, beta2, eps,
              weight_decay, 1.f, 1.f, 1.f, hyper_dev); else if (dtype == MK_F16)
    MK_LAUNCH((adamw_multi_kernel<_Float16>), grid, block, 0, MK_ST, it, cs, n_items, 0.f, beta1, beta2, eps,
              weight_decay, 1.f, 1.f, 1.f, hyper_dev);
  else if (dtype == MK_F32)
    MK_LAUNCH((adamw_multi_kernel<float>), grid, block, 0, MK_ST, it, cs, n_items, 0.f, beta1, beta2, eps,
              weight_decay, 1.f, 1.f, 1.f, hyper_dev);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}
