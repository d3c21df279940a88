// RMSNorm / LayerNorm forward+backward and column reductions (HBM-bound).
// One 256-thread block per row, 16-byte vector access, fp32 statistics.
// Rows are short enough (<= 16 KiB) that the second pass hits L1/L2, so the
// algorithmic HBM traffic is one read + one write of the activation.
//
// Reference numerics mirrored (modeling.py:311-319): variance in fp32, the
// normalised value is cast to the weight dtype BEFORE the weight multiply.
#include "common.h"
#include "../../include/macaw_hip.h"

namespace {


// ---------------------------------------------------------------- RMSNorm --
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* x, const T* res, const T* w,
                                                          T* h_out, T* y, float* rstd_out,
                                                          int cols, float eps) {
  constexpr int N = VecIO<T>::N;
  __shared__ float red[16];
  const long row = blockIdx.x;
  const T* xr = x + row * cols;
  const T* rr = res ? res + row * cols : nullptr;
  T* hr = h_out ? h_out + row * cols : nullptr;
  T* yr = y + row * cols;
  const int nch = cols / N;
  float ss = 0.f;
  for (int c = threadIdx.x; c < nch; c += 256) {
    float v[N];
    VecIO<T>::load(xr + c * N, v);
    if (rr) {
      float r[N];
      VecIO<T>::load(rr + c * N, r);
#pragma unroll
      for (int i = 0; i < N; ++i) v[i] = rnd<T>(v[i] + r[i]);
      VecIO<T>::store(hr + c * N, v);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) ss += v[i] * v[i];
  }
  for (int c = nch * N + threadIdx.x; c < cols; c += 256) {  // scalar tail
    float v = to_f32<T>(xr[c]);
    if (rr) { v = rnd<T>(v + to_f32<T>(rr[c])); hr[c] = from_f32<T>(v); }
    ss += v * v;
  }
  ss = block_sum<256>(ss, red);
  const float rstd = rsqrtf(ss / (float)cols + eps);
  if (threadIdx.x == 0) rstd_out[row] = rstd;
  const T* src = rr ? hr : xr;
  for (int c = threadIdx.x; c < nch; c += 256) {
    float v[N], g[N];
    VecIO<T>::load(src + c * N, v);
    VecIO<T>::load(w + c * N, g);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = g[i] * rnd<T>(v[i] * rstd);
    VecIO<T>::store(yr + c * N, v);
  }
  for (int c = nch * N + threadIdx.x; c < cols; c += 256)
    yr[c] = from_f32<T>(to_f32<T>(w[c]) * rnd<T>(to_f32<T>(src[c]) * rstd));
}

// dx = dres + rstd * (dn - n * mean(dn * n)), dn = dy*w, n = h*rstd.
// Block b owns rows b, b+nblk, ... and keeps its dw column sums in registers.
// PF: the three streams of the block's NEXT row are requested (raw 16-byte vectors, converted later) before
// the block-wide reduction of the current one -- a row is a serial chain load -> reduce (two barriers) ->
// store, and without the prefetch a block has nothing in flight during the last two links.
template <typename T, int CH, bool PF>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* dy, const T* h, const T* w,
                                                          const float* rstd, const T* dres, T* dx,
                                                          float* dw_partial, int rows, int cols) {
  constexpr int N = VecIO<T>::N;
  __shared__ float red[16];
  const int nch = cols / N;  // cols % N == 0 enforced by the host
  float dwacc[CH][N];
  float wv[CH][N];
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
#pragma unroll
    for (int i = 0; i < N; ++i) { dwacc[k][i] = 0.f; wv[k][i] = 0.f; }
    if (c < nch) VecIO<T>::load(w + c * N, wv[k]);
  }
  uint4 qdy[CH], qh[CH], qr[CH];
  float rs_q = 0.f;
  // Straight-line requests from CLAMPED addresses (row and chunk), selected to zero afterwards: a load under
  // `if` sits in its own basic block and the compiler then waits vmcnt(0) right behind it (see
  // attention_impl.inc ld16_or_zero) -- which is exactly the overlap this prefetch is for.
  auto fetch = [&](long row) {
    row = min(row, (long)rows - 1);
    const T* rsrc = dres ? dres : dy;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int cc = min((int)threadIdx.x + 256 * k, nch - 1);
      qdy[k] = VecIO<T>::load_raw(dy + row * cols + cc * N);
      qh[k] = VecIO<T>::load_raw(h + row * cols + cc * N);
      qr[k] = VecIO<T>::load_raw(rsrc + row * cols + cc * N);
    }
    rs_q = rstd[row];
  };
  if constexpr (PF) fetch(blockIdx.x);
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    float rs;
    float dyv[CH][N], nv[CH][N], rv[CH][N];
    float dot = 0.f;
    if constexpr (PF) {
      rs = rs_q;
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const bool ok = threadIdx.x + 256 * k < nch;
        VecIO<T>::unpack(qdy[k], dyv[k]);
        VecIO<T>::unpack(qh[k], nv[k]);
        VecIO<T>::unpack(qr[k], rv[k]);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          dyv[k][i] = ok ? dyv[k][i] : 0.f;
          nv[k][i] = ok ? nv[k][i] : 0.f;
          rv[k][i] = (ok && dres) ? rv[k][i] : 0.f;
        }
      }
      fetch(row + gridDim.x);          // (the last row of the block re-requests row rows - 1: harmless)
    } else {
      rs = rstd[row];
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        const int c = threadIdx.x + 256 * k;
#pragma unroll
        for (int i = 0; i < N; ++i) rv[k][i] = 0.f;
        if (c < nch) {
          VecIO<T>::load(dy + row * cols + c * N, dyv[k]);
          VecIO<T>::load(h + row * cols + c * N, nv[k]);
          if (dres) VecIO<T>::load(dres + row * cols + c * N, rv[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < nch) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          nv[k][i] *= rs;
          dwacc[k][i] += dyv[k][i] * nv[k][i];
          dyv[k][i] *= wv[k][i];
          dot += dyv[k][i] * nv[k][i];
        }
      }
    }
    dot = block_sum<256>(dot, red) / (float)cols;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < nch) {
        float o[N];
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = rv[k][i] + rs * (dyv[k][i] - nv[k][i] * dot);
        VecIO<T>::store(dx + row * cols + c * N, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < nch) {
#pragma unroll
      for (int i = 0; i < N; ++i) dw_partial[(long)blockIdx.x * cols + c * N + i] = dwacc[k][i];
    }
  }
}

// -------------------------------------------------------------- LayerNorm --
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* x, const T* w, const T* b,
                                                            T* y, float* mean_out, float* rstd_out,
                                                            int cols, float eps) {
  __shared__ float red[16];
  const long row = blockIdx.x;
  const T* xr = x + row * cols;
  T* yr = y + row * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s += to_f32<T>(xr[c]);
  const float mean = block_sum<256>(s, red) / (float)cols;
  float v = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) {
    const float d = to_f32<T>(xr[c]) - mean;
    v += d * d;
  }
  const float var = block_sum<256>(v, red) / (float)cols;
  const float rstd = rsqrtf(var + eps);
  if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  for (int c = threadIdx.x; c < cols; c += 256)
    yr[c] = from_f32<T>((to_f32<T>(xr[c]) - mean) * rstd * to_f32<T>(w[c]) + to_f32<T>(b[c]));
}

// Narrow rows (CLIP d = 1024, Whisper d = 512): one WAVE per row, the row lives in registers
// (CH x 16-byte loads per lane, read once), mean / variance by wave shuffles -- no LDS, no
// barriers, 4 rows per workgroup.  Same two-pass fp32 arithmetic as the block form.
template <int CH>
__global__ __launch_bounds__(256) void layernorm_fwd_wave_kernel(const bf16* x, const bf16* w,
                                                                 const bf16* b, bf16* y,
                                                                 float* mean_out, float* rstd_out,
                                                                 int rows, int cols, float eps) {
  const int l = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16* xr = x + row * cols;
  float v[CH][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (k * 64 + l) * 8;
    if (c < cols) {
      VecIO<bf16>::load(xr + c, v[k]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[k][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (k * 64 + l) * 8;
    if (c < cols) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
  if (l == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = (k * 64 + l) * 8;
    if (c < cols) {
      float wv[8], bv[8], o[8];
      VecIO<bf16>::load(w + c, wv);
      VecIO<bf16>::load(b + c, bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * wv[e] + bv[e];
      VecIO<bf16>::store(y + row * cols + c, o);
    }
  }
}

template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* dy, const T* x, const T* w,
                                                            const float* mean, const float* rstd,
                                                            const T* dres, T* dx, float* dw_partial,
                                                            float* db_partial, int rows, int cols) {
  __shared__ float red[16];
  float dwacc[CH], dbacc[CH], wv[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    dwacc[k] = 0.f; dbacc[k] = 0.f;
    wv[k] = (c < cols) ? to_f32<T>(w[c]) : 0.f;
  }
  for (long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float mu = mean[row], rs = rstd[row];
    float g[CH], xh[CH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      g[k] = 0.f; xh[k] = 0.f;
      if (c < cols) {
        const float d = to_f32<T>(dy[row * cols + c]);
        xh[k] = (to_f32<T>(x[row * cols + c]) - mu) * rs;
        dwacc[k] += d * xh[k];
        dbacc[k] += d;
        g[k] = d * wv[k];
        s1 += g[k];
        s2 += g[k] * xh[k];
      }
    }
    s1 = block_sum<256>(s1, red) / (float)cols;
    s2 = block_sum<256>(s2, red) / (float)cols;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const int c = threadIdx.x + 256 * k;
      if (c < cols) {
        float o = rs * (g[k] - s1 - xh[k] * s2);
        if (dres) o += to_f32<T>(dres[row * cols + c]);
        dx[row * cols + c] = from_f32<T>(o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < CH; ++k) {
    const int c = threadIdx.x + 256 * k;
    if (c < cols) {
      dw_partial[(long)blockIdx.x * cols + c] = dwacc[k];
      db_partial[(long)blockIdx.x * cols + c] = dbacc[k];
    }
  }
}

// out[c] (+)= sum_b partial[b][c].  Block = 32 columns x 8 row groups: every thread sums a
// strided subset of the partial rows (independent loads), then the 8 groups meet in LDS.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* partial, T* out,
                                                              int nblk, int cols, int accumulate) {
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, gy = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  float s = 0.f;
  if (c < cols)
    for (int b = gy; b < nblk; b += 8) s += partial[(long)b * cols + c];
  red[gy][cx] = s;
  __syncthreads();
  if (gy == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][cx];
    if (accumulate) t += to_f32<T>(out[c]);
    out[c] = from_f32<T>(t);
  }
}

// partial[b][c] = sum over the block's row slab of x[r][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* x, long ld, float* partial,
                                                          int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (long r = blockIdx.y; r < rows; r += gridDim.y) s += to_f32<T>(x[r * ld + c]);
  partial[(long)blockIdx.y * cols + c] = s;
}

template <typename T>
int rmsnorm_bwd_launch(const void* dy, const void* h, const void* w, const float* rstd,
                       const void* dres, void* dx, float* dwp, int nblk, int rows, int cols,
                       hipStream_t st) {
  constexpr int N = VecIO<T>::N;
  if (cols % N) return MK_ERR_UNSUPPORTED;
  const int ch = mk_cdiv(cols / N, 256);
  dim3 grid(nblk), block(256);
  // next-row prefetch for the narrow instantiations (ch <= 2: LLaMA widths); the wide ones have no registers
  // to spare.  MK_RMSNORM_BWD_NO_PREFETCH = the round-2 kernel, for A/B (scripts/bench_norm.py)
  static const bool pf = getenv("MK_RMSNORM_BWD_NO_PREFETCH") == nullptr;
#define MK_RB(CHV, PFV)                                                                        \
  MK_LAUNCH((rmsnorm_bwd_kernel<T, CHV, PFV>), grid, block, 0, st, (const T*)dy, (const T*)h, \
                     (const T*)w, rstd, (const T*)dres, (T*)dx, dwp, rows, cols)
  if (ch <= 1) { if (pf) MK_RB(1, true); else MK_RB(1, false); }
  else if (ch <= 2) { if (pf) MK_RB(2, true); else MK_RB(2, false); }
  else if (ch <= 4) MK_RB(4, false);
  else if (ch <= 8) MK_RB(8, false);
  else return MK_ERR_UNSUPPORTED;
#undef MK_RB
  return mk_check_launch();
}

template <typename T>
int layernorm_bwd_launch(const void* dy, const void* x, const void* w, const float* mean,
                         const float* rstd, const void* dres, void* dx, float* dwp, float* dbp,
                         int nblk, int rows, int cols, hipStream_t st) {
  const int ch = mk_cdiv(cols, 256);
  dim3 grid(nblk), block(256);
#define MK_LB(CHV)                                                                          \
  MK_LAUNCH((layernorm_bwd_kernel<T, CHV>), grid, block, 0, st, (const T*)dy,      \
                     (const T*)x, (const T*)w, mean, rstd, (const T*)dres, (T*)dx, dwp, dbp, \
                     rows, cols)
  if (ch <= 1) MK_LB(1);
  else if (ch <= 2) MK_LB(2);
  else if (ch <= 4) MK_LB(4);
  else if (ch <= 8) MK_LB(8);
  else if (ch <= 16) MK_LB(16);
  else if (ch <= 32) MK_LB(32);
  else return MK_ERR_UNSUPPORTED;
#undef MK_LB
  return mk_check_launch();
}

}  // namespace

#define MK_ST reinterpret_cast<hipStream_t>(stream)

extern "C" int mk_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y,
                              float* rstd, int32_t rows, int32_t cols, float eps, int32_t dtype,
                              void* stream) {
  if (!x || !w || !y || !rstd || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  if (res && !h_out) return MK_ERR_BAD_ARG;
  dim3 grid(rows), block(256);
  if (dtype == MK_BF16) {
    if (cols % 8) return MK_ERR_UNSUPPORTED;
    MK_LAUNCH((rmsnorm_fwd_kernel<bf16>), grid, block, 0, MK_ST, (const bf16*)x,
                       (const bf16*)res, (const bf16*)w, (bf16*)h_out, (bf16*)y, rstd, cols, eps);
  } else if (dtype == MK_F16) {
    if (cols % 8) return MK_ERR_UNSUPPORTED;
    MK_LAUNCH((rmsnorm_fwd_kernel<_Float16>), grid, block, 0, MK_ST, (const _Float16*)x,
                       (const _Float16*)res, (const _Float16*)w, (_Float16*)h_out, (_Float16*)y, rstd, cols, eps);
  } else if (dtype == MK_F32) {
    if (cols % 4) return MK_ERR_UNSUPPORTED;
    MK_LAUNCH((rmsnorm_fwd_kernel<float>), grid, block, 0, MK_ST, (const float*)x,
                       (const float*)res, (const float*)w, (float*)h_out, (float*)y, rstd, cols,
                       eps);
  } else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd,
                              const void* dres_in, void* dx, float* dw_partial, int32_t nblk,
                              int32_t rows, int32_t cols, int32_t dtype, void* stream) {
  if (!dy || !h || !w || !rstd || !dx || !dw_partial || nblk <= 0 || rows <= 0) return MK_ERR_BAD_ARG;
  if (dtype == MK_BF16)
    return rmsnorm_bwd_launch<bf16>(dy, h, w, rstd, dres_in, dx, dw_partial, nblk, rows, cols, MK_ST); else if (dtype == MK_F16)
    return rmsnorm_bwd_launch<_Float16>(dy, h, w, rstd, dres_in, dx, dw_partial, nblk, rows, cols, MK_ST);
  if (dtype == MK_F32)
    return rmsnorm_bwd_launch<float>(dy, h, w, rstd, dres_in, dx, dw_partial, nblk, rows, cols, MK_ST);
  return MK_ERR_UNSUPPORTED;
}

extern "C" int mk_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean,
                                float* rstd, int32_t rows, int32_t cols, float eps, int32_t dtype,
                                void* stream) {
  if (!x || !w || !b || !y || !mean || !rstd || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  dim3 grid(rows), block(256);
  const uintptr_t al16 = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) |
                         reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y);
  if (dtype == MK_BF16 && cols % 8 == 0 && cols <= 1024 && !(al16 & 15)) {
    const dim3 gw((rows + 3) / 4);
    if (cols <= 512)
      MK_LAUNCH((layernorm_fwd_wave_kernel<1>), gw, block, 0, MK_ST, (const bf16*)x, (const bf16*)w,
                (const bf16*)b, (bf16*)y, mean, rstd, rows, cols, eps);
    else
      MK_LAUNCH((layernorm_fwd_wave_kernel<2>), gw, block, 0, MK_ST, (const bf16*)x, (const bf16*)w,
                (const bf16*)b, (bf16*)y, mean, rstd, rows, cols, eps);
    return mk_check_launch();
  }
  if (dtype == MK_BF16)
    MK_LAUNCH((layernorm_fwd_kernel<bf16>), grid, block, 0, MK_ST, (const bf16*)x,
                       (const bf16*)w, (const bf16*)b, (bf16*)y, mean, rstd, cols, eps); else if (dtype == MK_F16)
    MK_LAUNCH((layernorm_fwd_kernel<_Float16>), grid, block, 0, MK_ST, (const _Float16*)x,
                       (const _Float16*)w, (const _Float16*)b, (_Float16*)y, mean, rstd, cols, eps);
  else if (dtype == MK_F32)
    MK_LAUNCH((layernorm_fwd_kernel<float>), grid, block, 0, MK_ST, (const float*)x,
                       (const float*)w, (const float*)b, (float*)y, mean, rstd, cols, eps);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean,
                                const float* rstd, const void* dres_in, void* dx,
                                float* dw_partial, float* db_partial, int32_t nblk, int32_t rows,
                                int32_t cols, int32_t dtype, void* stream) {
  if (!dy || !x || !w || !mean || !rstd || !dx || !dw_partial || !db_partial || nblk <= 0)
    return MK_ERR_BAD_ARG;
  if (dtype == MK_BF16)
    return layernorm_bwd_launch<bf16>(dy, x, w, mean, rstd, dres_in, dx, dw_partial, db_partial,
                                      nblk, rows, cols, MK_ST); else if (dtype == MK_F16)
    return layernorm_bwd_launch<_Float16>(dy, x, w, mean, rstd, dres_in, dx, dw_partial, db_partial,
                                      nblk, rows, cols, MK_ST);
  if (dtype == MK_F32)
    return layernorm_bwd_launch<float>(dy, x, w, mean, rstd, dres_in, dx, dw_partial, db_partial,
                                       nblk, rows, cols, MK_ST);
  return MK_ERR_UNSUPPORTED;
}

extern "C" int mk_colsum_partials(const float* partial, void* out, int32_t nblk, int32_t cols,
                                  int32_t accumulate, int32_t dtype, void* stream) {
  if (!partial || !out || nblk <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  dim3 grid(mk_cdiv(cols, 32)), block(256);
  if (dtype == MK_BF16)
    MK_LAUNCH((colsum_partials_kernel<bf16>), grid, block, 0, MK_ST, partial, (bf16*)out,
                       nblk, cols, accumulate); else if (dtype == MK_F16)
    MK_LAUNCH((colsum_partials_kernel<_Float16>), grid, block, 0, MK_ST, partial, (_Float16*)out,
                       nblk, cols, accumulate);
  else if (dtype == MK_F32)
    MK_LAUNCH((colsum_partials_kernel<float>), grid, block, 0, MK_ST, partial,
                       (float*)out, nblk, cols, accumulate);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_colsum(const void* x, int64_t ld, void* out, float* ws, int32_t nblk,
                         int32_t rows, int32_t cols, int32_t accumulate, int32_t dtype,
                         void* stream) {
  if (!x || !out || !ws || nblk <= 0 || rows <= 0 || cols <= 0) return MK_ERR_BAD_ARG;
  dim3 grid(mk_cdiv(cols, 256), nblk), block(256);
  if (dtype == MK_BF16)
    MK_LAUNCH((colsum_rows_kernel<bf16>), grid, block, 0, MK_ST, (const bf16*)x, (long)ld,
                       ws, rows, cols); else if (dtype == MK_F16)
    MK_LAUNCH((colsum_rows_kernel<_Float16>), grid, block, 0, MK_ST, (const _Float16*)x, (long)ld,
                       ws, rows, cols);
  else if (dtype == MK_F32)
    MK_LAUNCH((colsum_rows_kernel<float>), grid, block, 0, MK_ST, (const float*)x,
                       (long)ld, ws, rows, cols);
  else return MK_ERR_UNSUPPORTED;
  int rc = mk_check_launch();
  if (rc) return rc;
  return mk_colsum_partials(ws, out, nblk, cols, accumulate, dtype, stream);
}
