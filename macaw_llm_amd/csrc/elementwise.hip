// HBM-bound pointwise / gather kernels: RoPE, SwiGLU, activations, add, cast,
// fill, embedding gather + deterministic scatter, im2col / col2im, patchify.
// All use 16-byte vector access on the fast path (cdna_hip_programming.md G13)
// and grid-stride loops capped at 256 CUs x 8 blocks.
#include "common.h"
#include <algorithm>
#include "../../include/macaw_hip.h"

namespace {


inline int ew_grid(long work_items) {
  long b = (work_items + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------- RoPE --
// x[t][h][hd], token pitch ld.  Half-split rotation (modeling.py:76-91):
//   out[j]      = x[j]*cos[j]     - x[j+half]*sin[j]
//   out[j+half] = x[j+half]*cos[j] + x[j]*sin[j]        (cos[j] == cos[j+half])
// with each product and the sum rounded to T, as the eager bf16 reference does.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void rope_kernel(T* x, const T* cos_t, const T* sin_t,
                                                   const int32_t* pos, int tokens, int heads,
                                                   int hd, long ld, float sgn) {
  constexpr int N = VEC ? VecIO<T>::N : 1;
  const int half = hd / 2;
  const int cph = half / N;  // chunks per head-half
  const long total = (long)tokens * heads * cph;
  // (32-bit index arithmetic with shifts where chunks-per-half and heads are powers of two -- every model here:
  // the 64-bit divisions by run-time values were ~4/5 of this kernel's instructions and held it at 4.1 TB/s)
  const bool pow2 = total < 0x7fffffffL && (cph & (cph - 1)) == 0 && (heads & (heads - 1)) == 0;
  const int sh_c = __builtin_ctz((unsigned)cph), sh_h = __builtin_ctz((unsigned)heads);
  // two chunk pairs per trip, all eight loads requested before the first use (round 4: one pair per trip kept
  // 4 x 16 bytes per thread in flight and the in-place stream at 4.4 TB/s)
  const long stride = (long)gridDim.x * 256;
  for (long idx0 = (long)blockIdx.x * 256 + threadIdx.x; idx0 < total; idx0 += 2 * stride) {
    T* xp[2];
    float a[2][N], b[2][N], cs[2][N], sn[2][N];
    bool live[2];
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      const long idx = idx0 + u2 * stride;
      live[u2] = idx < total;
      const long ic = live[u2] ? idx : idx0;       // a dead second slot re-reads the first (never stored)
      int c, h;
      long t;
      if (pow2) {
        const unsigned u = (unsigned)ic;
        c = (int)(u & (unsigned)(cph - 1));
        const unsigned th = u >> sh_c;
        h = (int)(th & (unsigned)(heads - 1));
        t = (long)(th >> sh_h);
      } else {
        c = (int)(ic % cph);
        const long th = ic / cph;
        h = (int)(th % heads);
        t = th / heads;
      }
      const int p = pos[t];
      xp[u2] = x + t * ld + (long)h * hd + c * N;
      const T* cp = cos_t + (long)p * hd + c * N;
      const T* sp = sin_t + (long)p * hd + c * N;
      if constexpr (VEC) {
        VecIO<T>::load(xp[u2], a[u2]); VecIO<T>::load(xp[u2] + half, b[u2]);
        VecIO<T>::load(cp, cs[u2]); VecIO<T>::load(sp, sn[u2]);
      } else {
        a[u2][0] = to_f32<T>(xp[u2][0]); b[u2][0] = to_f32<T>(xp[u2][half]);
        cs[u2][0] = to_f32<T>(cp[0]); sn[u2][0] = to_f32<T>(sp[0]);
      }
    }
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      float o1[N], o2[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const float s_ = sgn * sn[u2][i];
        o1[i] = rnd<T>(rnd<T>(a[u2][i] * cs[u2][i]) + rnd<T>(-b[u2][i] * s_));
        o2[i] = rnd<T>(rnd<T>(b[u2][i] * cs[u2][i]) + rnd<T>(a[u2][i] * s_));
      }
      if (live[u2]) {
        if constexpr (VEC) { VecIO<T>::store(xp[u2], o1); VecIO<T>::store(xp[u2] + half, o2); }
        else { xp[u2][0] = from_f32<T>(o1[0]); xp[u2][half] = from_f32<T>(o2[0]); }
      }
    }
  }
}

// ----------------------------------------------------------------- SwiGLU --
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* g, const T* u, T* a, long n) {
  constexpr int N = VecIO<T>::N;
  const long nch = n / N;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
    float gv[N], uv[N], o[N];
    VecIO<T>::load(g + c * N, gv); VecIO<T>::load(u + c * N, uv);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = rnd<T>(gv[i] / (1.f + __expf(-gv[i]))) * uv[i];
    VecIO<T>::store(a + c * N, o);
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gv = to_f32<T>(g[i]);
    a[i] = from_f32<T>(rnd<T>(gv / (1.f + __expf(-gv))) * to_f32<T>(u[i]));
  }
}
template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* g, const T* u, const T* da,
                                                         T* dg, T* du, long n) {
  constexpr int N = VecIO<T>::N;
  const long nch = n / N;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
    float gv[N], uv[N], dv[N], og[N], ou[N];
    VecIO<T>::load(g + c * N, gv); VecIO<T>::load(u + c * N, uv); VecIO<T>::load(da + c * N, dv);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float s = 1.f / (1.f + __expf(-gv[i]));
      ou[i] = dv[i] * gv[i] * s;
      og[i] = dv[i] * uv[i] * s * (1.f + gv[i] * (1.f - s));
    }
    VecIO<T>::store(dg + c * N, og); VecIO<T>::store(du + c * N, ou);
  }
  for (long i = nch * N + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gv = to_f32<T>(g[i]), uv = to_f32<T>(u[i]), dv = to_f32<T>(da[i]);
    const float s = 1.f / (1.f + __expf(-gv));
    du[i] = from_f32<T>(dv * gv * s);
    dg[i] = from_f32<T>(dv * uv * s * (1.f + gv * (1.f - s)));
  }
}

// 2-D (pitched) forms: g = gu[:, :cols], u = gu[:, cols:] inside one [rows, 2*cols] buffer
// produced by the fused gate|up projection.
template <typename T>
__global__ __launch_bounds__(256) void swiglu2d_fwd_kernel(const T* g, const T* u, T* a, long rows,
                                                           int cols, long ld_in, long ld_out) {
  constexpr int N = VecIO<T>::N;
  const int cpr = cols / N;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * N;
    float gv[N], uv[N], o[N];
    VecIO<T>::load(g + r * ld_in + c, gv); VecIO<T>::load(u + r * ld_in + c, uv);
#pragma unroll
    for (int k = 0; k < N; ++k) o[k] = rnd<T>(gv[k] / (1.f + __expf(-gv[k]))) * uv[k];
    VecIO<T>::store(a + r * ld_out + c, o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void swiglu2d_bwd_kernel(const T* g, const T* u, const T* da,
                                                           T* dg, T* du, long rows, int cols,
                                                           long ld_gu, long ld_a) {
  constexpr int N = VecIO<T>::N;
  const int cpr = cols / N;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / cpr;
    const int c = (int)(i - r * cpr) * N;
    float gv[N], uv[N], dv[N], og[N], ou[N];
    VecIO<T>::load(g + r * ld_gu + c, gv); VecIO<T>::load(u + r * ld_gu + c, uv);
    VecIO<T>::load(da + r * ld_a + c, dv);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float sg = 1.f / (1.f + __expf(-gv[k]));
      ou[k] = dv[k] * gv[k] * sg;
      og[k] = dv[k] * uv[k] * sg * (1.f + gv[k] * (1.f - sg));
    }
    VecIO<T>::store(dg + r * ld_gu + c, og); VecIO<T>::store(du + r * ld_gu + c, ou);
  }
}

// ------------------------------------------------------------ activations --
MK_DEV float act_fwd(float v, int act) {
  if (act == 1) return 0.5f * v * (1.0f + mk_erf(v * 0.70710678118654752440f));
  if (act == 2) return v / (1.0f + __expf(-1.702f * v));
  return v;
}
MK_DEV float act_grad(float v, int act) {
  if (act == 1) {
    const float cdf = 0.5f * (1.0f + mk_erf(v * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    return cdf + v * pdf;
  }
  if (act == 2) {
    const float s = 1.f / (1.f + __expf(-1.702f * v));
    return s * (1.f + 1.702f * v * (1.f - s));
  }
  return 1.f;
}
template <typename T>
__global__ __launch_bounds__(256) void act_fwd_kernel(const T* x, T* y, long n, int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = from_f32<T>(act_fwd(to_f32<T>(x[i]), act));
}
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* x, const T* dy, T* dx, long n,
                                                      int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    dx[i] = from_f32<T>(to_f32<T>(dy[i]) * act_grad(to_f32<T>(x[i]), act));
}

// -------------------------------------------------------- add / cast / fill --
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* a, const T* b, T* y, long n,
                                                  long period) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long j = period > 0 ? i % period : i;
    y[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[j]));
  }
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* in, TO* out, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    out[i] = from_f32<TO>(to_f32<TI>(in[i]));
}
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* p, float v, long n) {
  const T tv = from_f32<T>(v);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = tv;
}

// -------------------------------------------------------------- embedding --
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const T* table, const int64_t* ids,
                                                            T* out, int dim, long ld_out,
                                                            int vocab) {
  const long t = blockIdx.x;
  long id = ids[t];
  if (id < 0) return;                 // slot owned by a modality prefix: left untouched
  if (id >= vocab) id = 0;            // host validates ids; never fault
  const T* src = table + id * dim;
  T* dst = out + t * ld_out;
  constexpr int N = VecIO<T>::N;
  if ((dim % N) == 0 && (ld_out % N) == 0) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (int c = threadIdx.x; c < dim / N; c += 256) d4[c] = s4[c];
  } else {
    for (int c = threadIdx.x; c < dim; c += 256) dst[c] = src[c];
  }
}
// Deterministic scatter-add without atomics: the block of the FIRST occurrence of a token id
// collects the positions of every occurrence (parallel scan of the id list into LDS), sums
// them in position order in fp32 and updates the table row once.
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const T* dout, long ld,
                                                            const int64_t* ids, T* dtable,
                                                            int tokens, int dim, int vocab,
                                                            long padding_idx) {
  constexpr int MAXM = 1024;
  __shared__ int dup;
  __shared__ int wcnt[4];
  __shared__ int match[MAXM];
  const int t = blockIdx.x;
  const long id = ids[t];
  if (id < 0 || id >= vocab || id == padding_idx) return;
  if (threadIdx.x == 0) dup = 0;
  __syncthreads();
  for (int j = threadIdx.x; j < t; j += 256)
    if (ids[j] == id) dup = 1;
  __syncthreads();
  if (dup) return;
  // ordered compaction of the later occurrences (ballot + prefix): match[] is sorted by position
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int nm = 0;
  for (int base = t + 1; base < tokens; base += 256) {
    const int j = base + threadIdx.x;
    const bool hit = (j < tokens) && (ids[j] == id);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = nm;
    for (int i = 0; i < wv; ++i) off += wcnt[i];
    const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (hit) {
      const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (k < MAXM) match[k] = j;
    }
    nm += tot;
    __syncthreads();
  }
  for (int c = threadIdx.x; c < dim; c += 256) {
    float acc = to_f32<T>(dout[(long)t * ld + c]);
    if (nm <= MAXM) {
      for (int k = 0; k < nm; ++k) acc += to_f32<T>(dout[(long)match[k] * ld + c]);
    } else {  // pathological: > MAXM repeats of one id, fall back to the linear scan
      for (int j = t + 1; j < tokens; ++j)
        if (ids[j] == id) acc += to_f32<T>(dout[(long)j * ld + c]);
    }
    T* dp = dtable + id * dim + c;
    *dp = from_f32<T>(to_f32<T>(*dp) + acc);
  }
}

// --------------------------------------------------------- im2col / col2im --
template <typename T>
__global__ __launch_bounds__(256) void im2col1d_kernel(const T* x, T* out, int B, int C, int Tn,
                                                       int kw, int stride, int pad, int Lout,
                                                       long sb, long sc, long st, long ld_out) {
  const long total = (long)B * Lout * ld_out;
  const int K = C * kw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int col = (int)(i % ld_out);
    const long row = i / ld_out;
    T v = from_f32<T>(0.f);
    if (col < K) {
      const int c = col / kw, t = col - c * kw;
      const int j = (int)(row % Lout);
      const long b = row / Lout;
      const int tau = j * stride + t - pad;
      if (tau >= 0 && tau < Tn) v = x[b * sb + c * sc + tau * st];
    }
    out[i] = v;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void col2im1d_kernel(const T* dcols, T* dx, int B, int C, int Tn,
                                                       int kw, int stride, int pad, int Lout,
                                                       long sb, long sc, long st, long ld_cols) {
  const long total = (long)B * C * Tn;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    // enumerate in (b, tau, c) order when channels are contiguous (sc == 1), else (b, c, tau)
    int c, tau; long b;
    if (sc == 1) { c = (int)(i % C); const long r = i / C; tau = (int)(r % Tn); b = r / Tn; }
    else { tau = (int)(i % Tn); const long r = i / Tn; c = (int)(r % C); b = r / C; }
    const int p = tau + pad;
    int jlo = (p - kw + 1 + stride - 1) / stride;  // ceil((p-kw+1)/stride) for p-kw+1 >= 0
    if (p - kw + 1 < 0) jlo = 0;
    int jhi = p / stride;
    if (jhi > Lout - 1) jhi = Lout - 1;
    float acc = 0.f;
    for (int j = jlo; j <= jhi; ++j) {
      const int t = p - j * stride;
      if (t >= 0 && t < kw) acc += to_f32<T>(dcols[((long)b * Lout + j) * ld_cols + c * kw + t]);
    }
    dx[b * sb + c * sc + tau * st] = from_f32<T>(acc);
  }
}
template <typename T, bool FWD>
__global__ __launch_bounds__(256) void patchify_kernel(const T* img, T* cols, int B, int C, int H,
                                                       int W, int P, long ld) {
  const int gy = H / P, gx = W / P;
  const int K = C * P * P;
  const long total = (long)B * gy * gx * (FWD ? ld : K);
  const int rowlen = FWD ? (int)ld : K;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int col = (int)(i % rowlen);
    const long row = i / rowlen;
    if (col >= K) { if (FWD) cols[row * ld + col] = from_f32<T>(0.f); continue; }
    const int c = col / (P * P), rem = col - c * P * P, dy = rem / P, dx = rem - dy * P;
    const int px = (int)(row % gx);
    const long r2 = row / gx;
    const int py = (int)(r2 % gy);
    const long b = r2 / gy;
    const long ii = ((b * C + c) * H + py * P + dy) * W + px * P + dx;
    if (FWD) cols[row * ld + col] = img[ii];
    else const_cast<T*>(img)[ii] = cols[row * ld + col];
  }
}


// ------------------------------------------------------------------ copy --
// dst[z][r][0:cols] = src[z][r][0:cols] with independent pitches / batch strides
// (src batch stride 0 broadcasts).  16-byte vectors when everything is aligned.
template <int ES>
__global__ __launch_bounds__(256) void copy2d_kernel(const char* src, char* dst, int rows,
                                                     long row_bytes, long ld_src, long ld_dst,
                                                     long s_src, long s_dst, int vec) {
  src += (long)blockIdx.z * s_src;
  dst += (long)blockIdx.z * s_dst;
  if (vec) {
    const long nv = row_bytes / 16;
    const long total = (long)rows * nv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long r = i / nv, c = i - r * nv;
      *reinterpret_cast<uint4*>(dst + r * ld_dst + c * 16) =
          *reinterpret_cast<const uint4*>(src + r * ld_src + c * 16);
    }
  } else {
    const long ne = row_bytes / ES;
    const long total = (long)rows * ne;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long r = i / ne, c = i - r * ne;
      if (ES == 2)
        *reinterpret_cast<unsigned short*>(dst + r * ld_dst + c * 2) =
            *reinterpret_cast<const unsigned short*>(src + r * ld_src + c * 2);
      else
        *reinterpret_cast<unsigned*>(dst + r * ld_dst + c * 4) =
            *reinterpret_cast<const unsigned*>(src + r * ld_src + c * 4);
    }
  }
}

// ------------------------------------------------------------------- fp8 --
// Per-tensor scaled OCP e4m3 (v_cvt_pk_fp8_f32 on gfx950 produces e4m3fn and does NOT saturate:
// anything above 448 becomes the NaN code 0x7f, hence the explicit clamp).
__global__ void fp8_zero_kernel(float* amax) { amax[0] = 0.f; }

template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const T* x, long n, float* amax) {
  __shared__ float red[16];
  constexpr int N = VecIO<T>::N;
  float m = 0.f;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n / N; c += (long)gridDim.x * 256) {
    float v[N];
    VecIO<T>::load(x + c * N, v);
#pragma unroll
    for (int k = 0; k < N; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  m = block_max<256>(m, red);
  // non-negative floats order like their bit patterns
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
}

template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const T* x, long n, uint8_t* q,
                                                        const float* amax, float* dequant) {
  const float am = amax[0];
  const float sc = am > 0.f ? 448.f / am : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) dequant[0] = am > 0.f ? am / 448.f : 1.f;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n / 8; c += (long)gridDim.x * 256) {
    float v[8];
    if constexpr (VecIO<T>::N == 8) {
      VecIO<T>::load(x + c * 8, v);
    } else {
      float a[4], b[4];
      VecIO<T>::load(x + c * 8, a);
      VecIO<T>::load(x + c * 8 + 4, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = fminf(fmaxf(v[k] * sc, -448.f), 448.f);
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    *reinterpret_cast<int2*>(q + c * 8) = make_int2(lo, hi);
  }
}


// ---- per-ROW scaled e4m3 (activations, gradients, K-major weights): one workgroup per row.
// q[r, c] = e4m3(clamp(x[r, c] * 448 / amax_r)), scale[r] = amax_r / 448 (1 for an all-zero row).
// The row is read twice (the second pass hits L2): HBM traffic = 2 B in + 1 B out per element.
MK_DEV int2 fp8_pack8(const float (&v)[8], float sc) {
  float t[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) t[k] = fminf(fmaxf(v[k] * sc, -448.f), 448.f);
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], hi, true);
  return make_int2(lo, hi);
}

__global__ __launch_bounds__(256) void fp8_rowquant_kernel(const bf16* x, long ld, int cols, uint8_t* q,
                                                           long ldq, float* scale) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const bf16* xr = x + row * ld;
  const int nch = cols / 8;
  float m = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    float v[8];
    VecIO<bf16>::load(xr + c * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
  }
  m = block_max<256>(m, red);
  const float sc = m > 0.f ? 448.f / m : 1.f;
  if (threadIdx.x == 0) scale[row] = m > 0.f ? m / 448.f : 1.f;
  uint8_t* qr = q + row * ldq;
  for (int c = threadIdx.x; c < nch; c += 256) {
    float v[8];
    VecIO<bf16>::load(xr + c * 8, v);
    *reinterpret_cast<int2*>(qr + c * 8) = fp8_pack8(v, sc);
  }
}

// Round 6: the row stays in REGISTERS between the amax pass and the quantisation (CH chunks of 8 per thread: rows of up
// to 2048 * CH elements) -- one load per element instead of two dependent sweeps (load -> block reduce -> load -> store
// serialised per 4-wave block: 2.4 TB/s effective on the 13B step's 14 GB, profiles/r05_cfg5_last_step.txt).  Same bytes out.
template <int CH>
__global__ __launch_bounds__(256) void fp8_rowquant_reg_kernel(const bf16* x, long ld, int cols, uint8_t* q,
                                                               long ldq, float* scale) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const bf16* xr = x + row * ld;
  const int nch = cols / 8;
  float v[CH][8];
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    if (c < nch) VecIO<bf16>::load(xr + c * 8, v[k]);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[k][e]));
  }
  m = block_max<256>(m, red);
  const float sc = m > 0.f ? 448.f / m : 1.f;
  if (threadIdx.x == 0) scale[row] = m > 0.f ? m / 448.f : 1.f;
  uint8_t* qr = q + row * ldq;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) *reinterpret_cast<int2*>(qr + c * 8) = fp8_pack8(v[k], sc);
  }
}

// RMSNorm with the e4m3 row quantisation of its OUTPUT folded in (BASELINE cfg 5: y1 = RMSNorm(x) feeds the fp8 q|k|v
// GEMM): the row stays in registers from the sum of squares to the quantised store -- y (bf16, for the bf16 grad-weight
// GEMM), its e4m3 image and the row scale leave in one pass instead of rmsnorm_fwd + fp8_rowquant (which re-reads y).
// Same arithmetic in the same order as rmsnorm_fwd_kernel (norm.hip: fp32 sum of squares per thread over chunks
// tid, tid + 256, ..., block_sum, y = w * rnd(h * rstd)) and as fp8_rowquant_reg_kernel on the ROUNDED y: all four
// outputs are bit-identical to the two launches (tests/test_fp8_gpu.py).
template <int CH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_fp8_kernel(const bf16* x, const bf16* res, const bf16* w, bf16* h_out,
                                                              bf16* y, float* rstd_out, uint8_t* q, long ldq,
                                                              float* scale, int cols, float eps) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const bf16* xr = x + row * cols;
  const int nch = cols / 8;
  float v[CH][8];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    if (c < nch) {
      VecIO<bf16>::load(xr + c * 8, v[k]);
      if (res) {
        float r[8];
        VecIO<bf16>::load(res + row * cols + c * 8, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[k][e] = rnd<bf16>(v[k][e] + r[e]);
        VecIO<bf16>::store(h_out + row * cols + c * 8, v[k]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += v[k][e] * v[k][e];
    }
  }
  ss = block_sum<256>(ss, red);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (threadIdx.x == 0) rstd_out[row] = rstd;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) {
      float g[8];
      VecIO<bf16>::load(w + c * 8, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[k][e] = rnd<bf16>(g[e] * rnd<bf16>(v[k][e] * rstd));
        m = fmaxf(m, fabsf(v[k][e]));
      }
      VecIO<bf16>::store(y + row * cols + c * 8, v[k]);
    }
  }
  m = block_max<256>(m, red);
  const float sc = m > 0.f ? 448.f / m : 1.f;
  if (threadIdx.x == 0) scale[row] = m > 0.f ? m / 448.f : 1.f;
  uint8_t* qr = q + row * ldq;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) *reinterpret_cast<int2*>(qr + c * 8) = fp8_pack8(v[k], sc);
  }
}

// ---- per-COLUMN scaled e4m3, TRANSPOSED output (the weight of a grad-input GEMM: dx = dy W reduces
// over W's ROWS, so the fp8 MFMA wants W^T K-major with one scale per row of W^T = per column of W).
// pass 1: amax[c] = max_r |x[r, c]| (row slabs, atomicMax on the bit pattern of a non-negative float)
__global__ __launch_bounds__(256) void fp8_colamax_kernel(const bf16* x, long ld, int rows, int cols,
                                                          float* amax, int rows_per_block) {
  // a block = 256 columns (32 groups of 8, tx) x 8 row lanes (ty): a wave reads two 512-byte row segments per
  // instruction; the 8 row lanes meet in LDS, ONE atomicMax per column and block.  (Round 3 gave a thread 8 columns of a
  // 128-row slab: 360 blocks for a [15360, 5120] weight and 120 atomics per column -- 85 us = 1.8 TB/s, and smaller
  // slabs made it slower, not faster: the atomics were the limit.  scripts/bench_fp8_weights.py)
  __shared__ float red[8][256];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c8 = blockIdx.x * 32 + tx;                    // this thread's group of 8 columns
  const bool live = c8 * 8 < cols;
  const int cc = live ? c8 * 8 : 0;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int r = r0 + ty;
  for (; r + 24 < r1; r += 32) {                           // four rows of this lane requested before the first is used
    uint4 raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = VecIO<bf16>::load_raw(x + (long)(r + 8 * u) * ld + cc);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      VecIO<bf16>::unpack(raw[u], v);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], fabsf(v[k]));
    }
  }
  for (; r < r1; r += 8) {
    float v[8];
    VecIO<bf16>::load(x + (long)r * ld + cc, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], fabsf(v[k]));
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[ty][tx * 8 + k] = m[k];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    float mm = red[0][threadIdx.x];
#pragma unroll
    for (int j = 1; j < 8; ++j) mm = fmaxf(mm, red[j][threadIdx.x]);
    atomicMax(reinterpret_cast<unsigned int*>(amax) + c, __float_as_uint(mm));
  }
}
// pass 2: qt[c, r] = e4m3(x[r, c] * 448 / amax[c]) for a 64 x 64 tile through LDS; scale[c] = amax[c] / 448
__global__ __launch_bounds__(256) void fp8_quant_t_kernel(const bf16* x, long ld, int rows, int cols,
                                                          const float* amax, uint8_t* qt, long ldqt,
                                                          float* scale) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  // load: thread t covers row (t >> 3) + 32 * i, columns 8 * (t & 7) .. + 8
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (t >> 3) + 32 * i, c = 8 * (t & 7);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 + r < rows && c0 + c < cols) VecIO<bf16>::load(x + (long)(r0 + r) * ld + c0 + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[r][c + k] = v[k];
  }
  __syncthreads();
  // store: thread t covers output row (column of x) (t >> 3) + 32 * i, 8 consecutive r
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (t >> 3) + 32 * i, r = 8 * (t & 7);
    if (c0 + c >= cols || r0 + r >= rows) continue;
    const float am = amax[c0 + c];
    const float sc = am > 0.f ? 448.f / am : 1.f;
    if (blockIdx.y == 0 && r == 0) scale[c0 + c] = am > 0.f ? am / 448.f : 1.f;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tile[r + k][c];
    *reinterpret_cast<int2*>(qt + (long)(c0 + c) * ldqt + r0 + r) = fp8_pack8(v, sc);
  }
}
__global__ __launch_bounds__(256) void fp8_zero_n_kernel(float* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
}  // namespace

#define MK_ST reinterpret_cast<hipStream_t>(stream)
#define MK_DISPATCH_T(dtype, CALL)                     \
  do {                                                 \
    if ((dtype) == MK_BF16) { using T = bf16; CALL; }  \
    else if ((dtype) == MK_F16) { using T = _Float16; CALL; } \
    else if ((dtype) == MK_F32) { using T = float; CALL; } \
    else return MK_ERR_UNSUPPORTED;                    \
  } while (0)

extern "C" int mk_rope(void* x, const void* cos_t, const void* sin_t, const int32_t* pos,
                       int32_t tokens, int32_t heads, int32_t hd, int64_t ld, int32_t inverse,
                       int32_t dtype, void* stream) {
  if (!x || !cos_t || !sin_t || !pos || tokens <= 0 || heads <= 0 || hd <= 0 || (hd & 1))
    return MK_ERR_BAD_ARG;
  const float sgn = inverse ? -1.f : 1.f;
  const int half = hd / 2;
  MK_DISPATCH_T(dtype, {
    constexpr int N = VecIO<T>::N;
    const bool vec = (half % N == 0) && (ld % N == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(cos_t) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(sin_t) & 15) == 0);
    if (vec) {
      const long total = (long)tokens * heads * (half / N);
      MK_LAUNCH((rope_kernel<T, true>), dim3(ew_grid(total)), dim3(256), 0, MK_ST, (T*)x,
                         (const T*)cos_t, (const T*)sin_t, pos, tokens, heads, hd, (long)ld, sgn);
    } else {
      const long total = (long)tokens * heads * half;
      MK_LAUNCH((rope_kernel<T, false>), dim3(ew_grid(total)), dim3(256), 0, MK_ST, (T*)x,
                         (const T*)cos_t, (const T*)sin_t, pos, tokens, heads, hd, (long)ld, sgn);
    }
  });
  return mk_check_launch();
}

extern "C" int mk_swiglu_fwd(const void* g, const void* u, void* a, int64_t n, int32_t dtype,
                             void* stream) {
  if (!g || !u || !a || n <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((swiglu_fwd_kernel<T>),
                                          dim3(ew_grid(n / VecIO<T>::N + 1)), dim3(256), 0, MK_ST,
                                          (const T*)g, (const T*)u, (T*)a, (long)n));
  return mk_check_launch();
}
extern "C" int mk_swiglu_bwd(const void* g, const void* u, const void* da, void* dg, void* du,
                             int64_t n, int32_t dtype, void* stream) {
  if (!g || !u || !da || !dg || !du || n <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((swiglu_bwd_kernel<T>),
                                          dim3(ew_grid(n / VecIO<T>::N + 1)), dim3(256), 0, MK_ST,
                                          (const T*)g, (const T*)u, (const T*)da, (T*)dg, (T*)du,
                                          (long)n));
  return mk_check_launch();
}
extern "C" int mk_act_fwd(const void* x, void* y, int64_t n, int32_t act, int32_t dtype,
                          void* stream) {
  if (!x || !y || n <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((act_fwd_kernel<T>), dim3(ew_grid(n)), dim3(256), 0,
                                          MK_ST, (const T*)x, (T*)y, (long)n, act));
  return mk_check_launch();
}
extern "C" int mk_act_bwd(const void* x_pre, const void* dy, void* dx, int64_t n, int32_t act,
                          int32_t dtype, void* stream) {
  if (!x_pre || !dy || !dx || n <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((act_bwd_kernel<T>), dim3(ew_grid(n)), dim3(256), 0,
                                          MK_ST, (const T*)x_pre, (const T*)dy, (T*)dx, (long)n,
                                          act));
  return mk_check_launch();
}
extern "C" int mk_add(const void* a, const void* b, void* y, int64_t n, int64_t period,
                      int32_t dtype, void* stream) {
  if (!a || !b || !y || n <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((add_kernel<T>), dim3(ew_grid(n)), dim3(256), 0, MK_ST,
                                          (const T*)a, (const T*)b, (T*)y, (long)n, (long)period));
  return mk_check_launch();
}
// ---- sum of squares (global gradient norm for clipping: HF max_grad_norm / DeepSpeed
// gradient_clipping in the reference's recipe).  Deterministic: fixed partition into <= 1024
// block partials in fp32, then one block sums them in index order.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* x, long n, float* partial) {
  __shared__ float red[16];
  constexpr int N = VecIO<T>::N;
  const long nvec = n / N;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    float v[N];
    VecIO<T>::load(x + i * N, v);
#pragma unroll
    for (int e = 0; e < N; ++e) s += v[e] * v[e];
  }
  if (blockIdx.x == 0)
    for (long i = nvec * N + threadIdx.x; i < n; i += 256) { const float v = to_f32<T>(x[i]); s += v * v; }
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int nblk, float* out,
                                                          int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
  s = block_sum<256>(s, red);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}
}  // namespace
// out[0] (+)= sum x[i]^2 in fp32; ws: f32 [1024] scratch; x 16-byte aligned
extern "C" int mk_sumsq(const void* x, int64_t n, float* ws, float* out, int32_t accumulate,
                        int32_t dtype, void* stream) {
  if (!x || !ws || !out || n <= 0) return MK_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(x) & 15) return MK_ERR_UNSUPPORTED;
  const int nblk = (int)std::min<long>(1024, (n + 8191) / 8192);
  MK_DISPATCH_T(dtype, MK_LAUNCH((sumsq_partial_kernel<T>), dim3(nblk), dim3(256), 0, MK_ST, (const T*)x,
                                 (long)n, ws));
  MK_LAUNCH(sumsq_final_kernel, dim3(1), dim3(256), 0, MK_ST, (const float*)ws, nblk, out, (int)accumulate);
  return mk_check_launch();
}

extern "C" int mk_fill(void* p, float v, int64_t n, int32_t dtype, void* stream) {
  if (!p || n <= 0) return MK_ERR_BAD_ARG;
  if (dtype == MK_F16) {   // fp16 only carries the reference's `.half()` inputs (llm_trainer.py:366)
    MK_LAUNCH((fill_kernel<_Float16>), dim3(ew_grid(n)), dim3(256), 0, MK_ST, (_Float16*)p, v, (long)n);
    return mk_check_launch();
  }
  MK_DISPATCH_T(dtype, MK_LAUNCH((fill_kernel<T>), dim3(ew_grid(n)), dim3(256), 0, MK_ST,
                                          (T*)p, v, (long)n));
  return mk_check_launch();
}

namespace {
template <typename TI>
int cast_from(const void* in, void* out, int32_t out_dtype, long n, hipStream_t st) {
  dim3 grid(ew_grid(n)), block(256);
  if (out_dtype == MK_F32)
    MK_LAUNCH((cast_kernel<TI, float>), grid, block, 0, st, (const TI*)in, (float*)out, n);
  else if (out_dtype == MK_BF16)
    MK_LAUNCH((cast_kernel<TI, bf16>), grid, block, 0, st, (const TI*)in, (bf16*)out, n);
  else if (out_dtype == MK_F16)
    MK_LAUNCH((cast_kernel<TI, _Float16>), grid, block, 0, st, (const TI*)in,
                       (_Float16*)out, n);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}
}  // namespace
extern "C" int mk_cast(const void* in, int32_t in_dtype, void* out, int32_t out_dtype, int64_t n,
                       void* stream) {
  if (!in || !out || n <= 0) return MK_ERR_BAD_ARG;
  if (in_dtype == MK_F32) return cast_from<float>(in, out, out_dtype, n, MK_ST);
  if (in_dtype == MK_BF16) return cast_from<bf16>(in, out, out_dtype, n, MK_ST);
  if (in_dtype == MK_F16) return cast_from<_Float16>(in, out, out_dtype, n, MK_ST);
  return MK_ERR_UNSUPPORTED;
}

extern "C" int mk_embedding_fwd(const void* table, const int64_t* ids, void* out, int32_t tokens,
                                int32_t dim, int64_t ld_out, int32_t vocab, int32_t dtype,
                                void* stream) {
  if (!table || !ids || !out || tokens <= 0 || dim <= 0 || vocab <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((embedding_fwd_kernel<T>), dim3(tokens), dim3(256), 0,
                                          MK_ST, (const T*)table, ids, (T*)out, dim, (long)ld_out,
                                          vocab));
  return mk_check_launch();
}
extern "C" int mk_embedding_bwd(const void* dout, int64_t ld, const int64_t* ids, void* dtable,
                                int32_t tokens, int32_t dim, int32_t vocab, int64_t padding_idx,
                                int32_t dtype, void* stream) {
  if (!dout || !ids || !dtable || tokens <= 0 || dim <= 0 || vocab <= 0) return MK_ERR_BAD_ARG;
  MK_DISPATCH_T(dtype, MK_LAUNCH((embedding_bwd_kernel<T>), dim3(tokens), dim3(256), 0,
                                          MK_ST, (const T*)dout, (long)ld, ids, (T*)dtable, tokens,
                                          dim, vocab, (long)padding_idx));
  return mk_check_launch();
}

extern "C" int mk_im2col1d(const void* x, void* out, int32_t B, int32_t C, int32_t T_, int32_t kw,
                           int32_t stride, int32_t pad, int32_t Lout, int64_t sb, int64_t sc,
                           int64_t st, int64_t ld_out, int32_t dtype, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || T_ <= 0 || kw <= 0 || stride <= 0 || Lout <= 0 ||
      ld_out < (int64_t)C * kw)
    return MK_ERR_BAD_ARG;
  const long total = (long)B * Lout * ld_out;
  MK_DISPATCH_T(dtype, MK_LAUNCH((im2col1d_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                          MK_ST, (const T*)x, (T*)out, B, C, T_, kw, stride, pad,
                                          Lout, (long)sb, (long)sc, (long)st, (long)ld_out));
  return mk_check_launch();
}
extern "C" int mk_col2im1d(const void* dcols, void* dx, int32_t B, int32_t C, int32_t T_,
                           int32_t kw, int32_t stride, int32_t pad, int32_t Lout, int64_t sb,
                           int64_t sc, int64_t st, int64_t ld_cols, int32_t dtype, void* stream) {
  if (!dcols || !dx || B <= 0 || C <= 0 || T_ <= 0 || kw <= 0 || stride <= 0 || Lout <= 0)
    return MK_ERR_BAD_ARG;
  const long total = (long)B * C * T_;
  MK_DISPATCH_T(dtype, MK_LAUNCH((col2im1d_kernel<T>), dim3(ew_grid(total)), dim3(256), 0,
                                          MK_ST, (const T*)dcols, (T*)dx, B, C, T_, kw, stride, pad,
                                          Lout, (long)sb, (long)sc, (long)st, (long)ld_cols));
  return mk_check_launch();
}
extern "C" int mk_patchify(const void* img, void* out, int32_t B, int32_t C, int32_t H, int32_t W,
                           int32_t P, int64_t ld_out, int32_t dtype, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || P <= 0 || H % P || W % P || ld_out < (int64_t)C * P * P)
    return MK_ERR_BAD_ARG;
  const long total = (long)B * (H / P) * (W / P) * ld_out;
  MK_DISPATCH_T(dtype, MK_LAUNCH((patchify_kernel<T, true>), dim3(ew_grid(total)),
                                          dim3(256), 0, MK_ST, (const T*)img, (T*)out, B, C, H, W,
                                          P, (long)ld_out));
  return mk_check_launch();
}
extern "C" int mk_unpatchify(const void* dcols, void* dimg, int32_t B, int32_t C, int32_t H,
                             int32_t W, int32_t P, int64_t ld_cols, int32_t dtype, void* stream) {
  if (!dcols || !dimg || B <= 0 || C <= 0 || P <= 0 || H % P || W % P) return MK_ERR_BAD_ARG;
  const long total = (long)B * (H / P) * (W / P) * C * P * P;
  MK_DISPATCH_T(dtype, MK_LAUNCH((patchify_kernel<T, false>), dim3(ew_grid(total)),
                                          dim3(256), 0, MK_ST, (const T*)dimg, (T*)dcols, B, C, H,
                                          W, P, (long)ld_cols));
  return mk_check_launch();
}

extern "C" int mk_copy2d(const void* src, void* dst, int32_t rows, int32_t cols, int64_t ld_src,
                         int64_t ld_dst, int32_t batch, int64_t s_src, int64_t s_dst,
                         int32_t elem_size, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || batch <= 0) return MK_ERR_BAD_ARG;
  if (elem_size != 2 && elem_size != 4) return MK_ERR_UNSUPPORTED;
  const long es = elem_size;
  const long row_bytes = (long)cols * es;
  const int vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0 &&
                  (row_bytes % 16 == 0) && ((ld_src * es) % 16 == 0) && ((ld_dst * es) % 16 == 0) &&
                  ((s_src * es) % 16 == 0) && ((s_dst * es) % 16 == 0);
  const long work = (long)rows * (vec ? row_bytes / 16 : cols);
  dim3 grid(ew_grid(work), 1, batch), block(256);
  if (elem_size == 2)
    MK_LAUNCH((copy2d_kernel<2>), grid, block, 0, MK_ST, (const char*)src, (char*)dst, rows,
                       row_bytes, (long)ld_src * es, (long)ld_dst * es, (long)s_src * es,
                       (long)s_dst * es, vec);
  else
    MK_LAUNCH((copy2d_kernel<4>), grid, block, 0, MK_ST, (const char*)src, (char*)dst, rows,
                       row_bytes, (long)ld_src * es, (long)ld_dst * es, (long)s_src * es,
                       (long)s_dst * es, vec);
  return mk_check_launch();
}

extern "C" int mk_swiglu2d_fwd(const void* g, const void* u, void* a, int64_t rows, int32_t cols,
                               int64_t ld_in, int64_t ld_out, int32_t dtype, void* stream) {
  if (!g || !u || !a || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (cols % 8 || ld_in % 8 || ld_out % 8) return MK_ERR_UNSUPPORTED;
  MK_DISPATCH_T(dtype, MK_LAUNCH((swiglu2d_fwd_kernel<T>), dim3(ew_grid(rows * (cols / VecIO<T>::N))),
                                 dim3(256), 0, MK_ST, (const T*)g, (const T*)u, (T*)a, (long)rows,
                                 cols, (long)ld_in, (long)ld_out));
  return mk_check_launch();
}
extern "C" int mk_swiglu2d_bwd(const void* g, const void* u, const void* da, void* dg, void* du,
                               int64_t rows, int32_t cols, int64_t ld_gu, int64_t ld_a,
                               int32_t dtype, void* stream) {
  if (!g || !u || !da || !dg || !du || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (cols % 8 || ld_gu % 8 || ld_a % 8) return MK_ERR_UNSUPPORTED;
  MK_DISPATCH_T(dtype, MK_LAUNCH((swiglu2d_bwd_kernel<T>), dim3(ew_grid(rows * (cols / VecIO<T>::N))),
                                 dim3(256), 0, MK_ST, (const T*)g, (const T*)u, (const T*)da, (T*)dg,
                                 (T*)du, (long)rows, cols, (long)ld_gu, (long)ld_a));
  return mk_check_launch();
}

extern "C" int mk_fp8_quantize(const void* x, int64_t n, int32_t dtype, uint8_t* q, float* amax_ws,
                               float* dequant_scale, void* stream) {
  if (!x || !q || !amax_ws || !dequant_scale || n <= 0) return MK_ERR_BAD_ARG;
  if ((n % 8) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(q) & 7))
    return MK_ERR_UNSUPPORTED;
  MK_LAUNCH(fp8_zero_kernel, dim3(1), dim3(1), 0, MK_ST, amax_ws);
  long nb = (n / 8 + 255) / 256;
  if (nb > 4096) nb = 4096;
  const dim3 grid((unsigned)nb), block(256);
  if (dtype == MK_BF16) {
    MK_LAUNCH((fp8_amax_kernel<bf16>), grid, block, 0, MK_ST, (const bf16*)x, (long)n, amax_ws);
    MK_LAUNCH((fp8_quant_kernel<bf16>), grid, block, 0, MK_ST, (const bf16*)x, (long)n, q, amax_ws,
              dequant_scale);
  } else if (dtype == MK_F32) {
    MK_LAUNCH((fp8_amax_kernel<float>), grid, block, 0, MK_ST, (const float*)x, (long)n, amax_ws);
    MK_LAUNCH((fp8_quant_kernel<float>), grid, block, 0, MK_ST, (const float*)x, (long)n, q, amax_ws,
              dequant_scale);
  } else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_fp8_quantize_rows(const void* x, int32_t rows, int32_t cols, int64_t ld, int32_t dtype,
                                    uint8_t* q, int64_t ldq, float* scales, void* stream) {
  if (!x || !q || !scales || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (cols % 8) || (ld % 8) || (ldq % 8) || (reinterpret_cast<uintptr_t>(x) & 15) ||
      (reinterpret_cast<uintptr_t>(q) & 7))
    return MK_ERR_UNSUPPORTED;
  const int ch = mk_cdiv(cols / 8, 256);
  static const bool two_pass = getenv("MK_FP8_ROWQUANT_TWO_PASS") != nullptr;       // A/B: the round-3 kernel
#define MK_RQ(CHV) MK_LAUNCH(fp8_rowquant_reg_kernel<CHV>, dim3(rows), dim3(256), 0, MK_ST, (const bf16*)x, (long)ld, cols, q, \
                             (long)ldq, scales)
  if (two_pass || ch > 8)
    MK_LAUNCH(fp8_rowquant_kernel, dim3(rows), dim3(256), 0, MK_ST, (const bf16*)x, (long)ld, cols, q,
              (long)ldq, scales);
  else if (ch <= 1) MK_RQ(1);
  else if (ch <= 2) MK_RQ(2);
  else if (ch <= 3) MK_RQ(3);
  else if (ch <= 4) MK_RQ(4);
  else MK_RQ(8);
#undef MK_RQ
  return mk_check_launch();
}

extern "C" int mk_rmsnorm_fwd_fp8(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd,
                                  uint8_t* q, int64_t ldq, float* scales, int32_t rows, int32_t cols, float eps,
                                  int32_t dtype, void* stream) {
  if (!x || !w || !y || !rstd || !q || !scales || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (res && !h_out) return MK_ERR_BAD_ARG;
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                       reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(h_out);
  if (dtype != MK_BF16 || (cols % 8) || (ldq % 8) || cols > 8 * 2048 || (al & 15) || (reinterpret_cast<uintptr_t>(q) & 7))
    return MK_ERR_UNSUPPORTED;
  const int ch = mk_cdiv(cols / 8, 256);
#define MK_RF(CHV) MK_LAUNCH(rmsnorm_fwd_fp8_kernel<CHV>, dim3(rows), dim3(256), 0, MK_ST, (const bf16*)x, (const bf16*)res, \
                             (const bf16*)w, (bf16*)h_out, (bf16*)y, rstd, q, (long)ldq, scales, cols, eps)
  if (ch <= 1) MK_RF(1);
  else if (ch <= 2) MK_RF(2);
  else if (ch <= 3) MK_RF(3);
  else if (ch <= 4) MK_RF(4);
  else MK_RF(8);
#undef MK_RF
  return mk_check_launch();
}

extern "C" int mk_fp8_quantize_cols_t(const void* x, int32_t rows, int32_t cols, int64_t ld, int32_t dtype,
                                      uint8_t* qt, int64_t ldqt, float* scales, float* amax_ws, void* stream) {
  if (!x || !qt || !scales || !amax_ws || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (dtype != MK_BF16 || (cols % 8) || (rows % 8) || (ld % 8) || (ldqt % 8) ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(qt) & 7))
    return MK_ERR_UNSUPPORTED;
  MK_LAUNCH(fp8_zero_n_kernel, dim3(mk_cdiv(cols, 256)), dim3(256), 0, MK_ST, amax_ws, cols);
  // MK_FP8_COLAMAX_RPB: rows per block (sweep: scripts/bench_fp8_weights.py)
  static const int rpb = [] { const char* e = getenv("MK_FP8_COLAMAX_RPB"); const int v = e ? atoi(e) : 64; return v > 0 ? v : 64; }();
  MK_LAUNCH(fp8_colamax_kernel, dim3(mk_cdiv(cols, 256), mk_cdiv(rows, rpb)), dim3(256), 0, MK_ST,
            (const bf16*)x, (long)ld, rows, cols, amax_ws, rpb);
  MK_LAUNCH(fp8_quant_t_kernel, dim3(mk_cdiv(cols, 64), mk_cdiv(rows, 64)), dim3(256), 0, MK_ST,
            (const bf16*)x, (long)ld, rows, cols, amax_ws, qt, (long)ldqt, scales);
  return mk_check_launch();
}
