// Host-input pipeline on the GPU (llm_trainer.py:150-157,306-381): the
// per-step work the reference does synchronously on the CPU main thread --
//   * CLIP `_transform(224)`: PIL bicubic Resize (antialiased, 8-bit fixed point) ->
//     CenterCrop -> ToTensor -> Normalize,
//   * `whisper.log_mel_spectrogram`: reflect-padded 400/160 Hann STFT -> |.|^2 -> 80 Slaney
//     mel bands -> log10 / dynamic-range clamp / (x + 4) / 4
// as gfx950 kernels.  Image resampling is integer work and is BIT-EXACT with Pillow's
// ImagingResample (libImaging/Resample.c, Pillow 12.2): same 22-bit fixed-point coefficients
// (computed on the host in double, as Pillow does), same horizontal-then-vertical order with an
// 8-bit intermediate, same rounding (+2^21, arithmetic shift, clip).  The log-mel DFT
// accumulates in fp64 (v_fma_f64 is full rate on CDNA4), so it is closer to the exact spectrum
// than any fp32 FFT; parity with the fp32 reference is bounded by the reference's own FFT error.
#include "common.h"
#include "../../include/macaw_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow Resample.c

// One image of a batch; filled by the host (macaw_llm_amd/preprocess.py), read from HBM.
struct ImageDesc {
  int64_t src_off;   // byte offset of the HWC uint8 RGB image in the packed source buffer
  int64_t H, W;      // source size
  int64_t tmp_off;   // byte offset of this image's [nrows][OUT][3] intermediate
  int64_t row0;      // first source row the vertical pass needs
  int64_t nrows;     // number of source rows it needs
  int64_t hk_off, hb_off, hks;  // int32 offsets: horizontal coeffs [OUT][hks], bounds [OUT][2]
  int64_t vk_off, vb_off, vks;  // vertical coeffs [OUT][vks], bounds [OUT][2] (rows rel. to 0)
};

MK_DEV int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[r][c][ch] for the nrows x OUT window the crop needs
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src,
                                                         uint8_t* __restrict__ tmp,
                                                         const ImageDesc* __restrict__ descs,
                                                         const int32_t* __restrict__ coef, int OUT) {
  const ImageDesc d = descs[blockIdx.y];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= d.nrows * OUT) return;
  const int r = (int)(i / OUT), c = (int)(i % OUT);
  const int xmin = coef[d.hb_off + 2 * c], n = coef[d.hb_off + 2 * c + 1];
  const int32_t* k = coef + d.hk_off + (long)c * d.hks;
  const uint8_t* p = src + d.src_off + ((d.row0 + r) * d.W + xmin) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
    s0 += p[3 * x] * kv;
    s1 += p[3 * x + 1] * kv;
    s2 += p[3 * x + 2] * kv;
  }
  uint8_t* o = tmp + d.tmp_off + ((long)r * OUT + c) * 3;
  o[0] = (uint8_t)clip8(s0);
  o[1] = (uint8_t)clip8(s1);
  o[2] = (uint8_t)clip8(s2);
}

// vertical pass fused with CenterCrop + ToTensor + Normalize: out[img][ch][y][x] =
// lut[ch][resampled byte]  (the 3 x 256 table holds ((v / 255) - mean) / std evaluated on the
// host with the reference's own float ops, so the float result is bit-identical too)
template <typename T>
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp,
                                                              T* __restrict__ out,
                                                              const ImageDesc* __restrict__ descs,
                                                              const int32_t* __restrict__ coef,
                                                              const float* __restrict__ lut, int OUT) {
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const ImageDesc d = descs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= OUT * OUT) return;
  const int y = i / OUT, c = i % OUT;
  const int ymin = coef[d.vb_off + 2 * y] - (int)d.row0, n = coef[d.vb_off + 2 * y + 1];
  const int32_t* k = coef + d.vk_off + (long)y * d.vks;
  const uint8_t* p = tmp + d.tmp_off + ((long)ymin * OUT + c) * 3;
  int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
    const uint8_t* q = p + (long)x * OUT * 3;
    s0 += q[0] * kv;
    s1 += q[1] * kv;
    s2 += q[2] * kv;
  }
  T* o = out + (long)blockIdx.y * 3 * OUT * OUT + (long)y * OUT + c;
  o[0] = from_f32<T>(s_lut[clip8(s0)]);
  o[(long)OUT * OUT] = from_f32<T>(s_lut[256 + clip8(s1)]);
  o[2L * OUT * OUT] = from_f32<T>(s_lut[512 + clip8(s2)]);
}

// --------------------------------------------------------------------- log-mel
constexpr int NFFT = 400, HOP = 160, NBIN = NFFT / 2 + 1;  // whisper.audio: N_FFT, HOP_LENGTH
constexpr int FT = 32;                                      // frames per workgroup
constexpr int FG = 16;                                      // frames per register pass
constexpr int SPAN = (FT - 1) * HOP + NFFT;                 // samples a workgroup touches
constexpr int PW_LD = NBIN + 2;

MK_DEV int f2key(float v) {  // order-preserving float -> int map for atomicMax
  const int b = __float_as_int(v);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
MK_DEV float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void logmel_init_kernel(int32_t* maxkey, int B) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < B) maxkey[i] = f2key(-INFINITY);
}

// One workgroup = FT consecutive frames of one clip.  Thread k < 201 owns DFT bin k: the
// windowed twiddle (cos, sin)(2 pi k n / 400) * hann[n] is formed once per n and applied to FG
// frames held in registers (the frame samples are wave-uniform LDS broadcasts).
__global__ __launch_bounds__(256) void logmel_frames_kernel(
    const float* __restrict__ audio, long ld, int n_samples, int n_frames,
    const float* __restrict__ window, const double* __restrict__ twiddle,
    const float* __restrict__ melw, const int32_t* __restrict__ mel_lo,
    const int32_t* __restrict__ mel_hi, int n_mels, float* __restrict__ logspec,
    int32_t* __restrict__ maxkey) {
  extern __shared__ double s_mem[];
  double* s_x = s_mem;                         // SPAN
  double* s_tw = s_x + SPAN;                   // 2 * NFFT  (cos, sin)
  double* s_win = s_tw + 2 * NFFT;             // NFFT
  float* s_pw = reinterpret_cast<float*>(s_win + NFFT);  // FT * PW_LD
  __shared__ float s_red[4];
  const int b = blockIdx.y, f_base = blockIdx.x * FT, t = threadIdx.x;
  const float* x = audio + (long)b * ld;
  // torch.stft(center=True, pad_mode="reflect"): padded[i] = x[reflect(i - NFFT/2)]
  const long p0 = (long)f_base * HOP - NFFT / 2;
  for (int i = t; i < SPAN; i += 256) {
    long j = p0 + i;
    if (j < 0) j = -j;
    if (j >= n_samples) j = 2L * (n_samples - 1) - j;
    s_x[i] = (j >= 0 && j < n_samples) ? (double)x[j] : 0.0;
  }
  for (int i = t; i < 2 * NFFT; i += 256) s_tw[i] = twiddle[i];
  for (int i = t; i < NFFT; i += 256) s_win[i] = (double)window[i];
  __syncthreads();
  if (t < NBIN) {
#pragma unroll 1
    for (int f0 = 0; f0 < FT; f0 += FG) {
      double re[FG], im[FG];
#pragma unroll
      for (int f = 0; f < FG; ++f) re[f] = im[f] = 0.0;
      int idx = 0;
      const double* xs = s_x + f0 * HOP;
#pragma unroll 2
      for (int n = 0; n < NFFT; ++n) {
        const double w = s_win[n];
        const double cw = s_tw[2 * idx] * w, sw = s_tw[2 * idx + 1] * w;
#pragma unroll
        for (int f = 0; f < FG; ++f) {
          const double xv = xs[f * HOP + n];
          re[f] = fma(xv, cw, re[f]);
          im[f] = fma(xv, sw, im[f]);
        }
        idx += t;
        if (idx >= NFFT) idx -= NFFT;
      }
#pragma unroll
      for (int f = 0; f < FG; ++f)
        s_pw[(f0 + f) * PW_LD + t] = (float)(re[f] * re[f] + im[f] * im[f]);
    }
  }
  __syncthreads();
  // mel bands (each triangular filter touches few bins) + log10, frames innermost so the
  // [n_mels][n_frames] rows are written as 128-byte runs
  float vmax = -INFINITY;
  for (int o = t; o < n_mels * FT; o += 256) {
    const int m = o / FT, f = o % FT;
    if (f_base + f >= n_frames) continue;
    const int lo = mel_lo[m], hi = mel_hi[m];
    double acc = 0.0;
    for (int k = lo; k < hi; ++k) acc = fma((double)melw[m * NBIN + k], (double)s_pw[f * PW_LD + k], acc);
    // log10 evaluated in fp64 and rounded once: the correctly rounded fp32 result (ocml's
    // log10f is 1 ulp off at exactly 1e-10, the silence floor)
    const float v = (float)log10((double)fmaxf((float)acc, 1e-10f));
    logspec[((long)b * n_mels + m) * n_frames + f_base + f] = v;
    vmax = fmaxf(vmax, v);
  }
  vmax = block_max<256>(vmax, s_red);
  if (t == 0) atomicMax(&maxkey[b], f2key(vmax));
}

// log_spec = max(log_spec, max - 8); (log_spec + 4) / 4    (whisper/audio.py)
template <typename T>
__global__ __launch_bounds__(256) void logmel_finish_kernel(const float* __restrict__ logspec,
                                                            const int32_t* __restrict__ maxkey,
                                                            T* __restrict__ out, long per_clip,
                                                            long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float floor_v = key2f(maxkey[i / per_clip]) - 8.0f;
  out[i] = from_f32<T>((fmaxf(logspec[i], floor_v) + 4.0f) / 4.0f);
}

}  // namespace

#define MK_ST reinterpret_cast<hipStream_t>(stream)

extern "C" int mk_image_transform(const uint8_t* src, uint8_t* tmp, const int64_t* descs,
                                  const int32_t* coef, const float* lut, void* out,
                                  int32_t n_images, int32_t out_px, int64_t max_tmp_rows,
                                  int32_t dtype, void* stream) {
  if (!src || !tmp || !descs || !coef || !lut || !out || n_images <= 0 || out_px <= 0 ||
      max_tmp_rows <= 0)
    return MK_ERR_BAD_ARG;
  static_assert(sizeof(ImageDesc) == 12 * sizeof(int64_t), "descriptor = 12 x int64");
  const ImageDesc* d = reinterpret_cast<const ImageDesc*>(descs);
  const long hthreads = max_tmp_rows * out_px;
  MK_LAUNCH(resample_h_kernel, dim3((unsigned)((hthreads + 255) / 256), n_images), dim3(256), 0,
            MK_ST, src, tmp, d, coef, out_px);
  int rc = mk_check_launch();
  if (rc) return rc;
  const dim3 grid((out_px * out_px + 255) / 256, n_images);
  if (dtype == MK_F32)
    MK_LAUNCH((resample_v_norm_kernel<float>), grid, dim3(256), 0, MK_ST, tmp, (float*)out, d, coef,
              lut, out_px);
  else if (dtype == MK_BF16)
    MK_LAUNCH((resample_v_norm_kernel<bf16>), grid, dim3(256), 0, MK_ST, tmp, (bf16*)out, d, coef,
              lut, out_px);
  else if (dtype == MK_F16)
    MK_LAUNCH((resample_v_norm_kernel<_Float16>), grid, dim3(256), 0, MK_ST, tmp, (_Float16*)out, d,
              coef, lut, out_px);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}

extern "C" int mk_log_mel(const float* audio, int64_t ld, int32_t n_clips, int32_t n_samples,
                          const float* window, const double* twiddle, const float* mel_filters,
                          const int32_t* mel_lo, const int32_t* mel_hi, int32_t n_mels,
                          float* ws_logspec, int32_t* ws_max, void* out, int32_t dtype,
                          void* stream) {
  if (!audio || !window || !twiddle || !mel_filters || !mel_lo || !mel_hi || !ws_logspec ||
      !ws_max || !out || n_clips <= 0 || n_samples < NFFT || n_mels <= 0 || ld < n_samples)
    return MK_ERR_BAD_ARG;
  if (n_samples % HOP) return MK_ERR_UNSUPPORTED;   // whisper clips are whole hops (480000)
  const int n_frames = n_samples / HOP;             // stft gives n/HOP + 1 frames, the last is dropped
  MK_LAUNCH(logmel_init_kernel, dim3((n_clips + 63) / 64), dim3(64), 0, MK_ST, ws_max, n_clips);
  int rc = mk_check_launch();
  if (rc) return rc;
  const size_t lds = (SPAN + 3 * NFFT) * sizeof(double) + FT * PW_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(logmel_frames_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return MK_ERR_LAUNCH;
    attr_set = true;
  }
  MK_LAUNCH(logmel_frames_kernel, dim3((n_frames + FT - 1) / FT, n_clips), dim3(256), lds, MK_ST,
            audio, (long)ld, n_samples, n_frames, window, twiddle, mel_filters, mel_lo, mel_hi,
            n_mels, ws_logspec, ws_max);
  rc = mk_check_launch();
  if (rc) return rc;
  const long per = (long)n_mels * n_frames, total = per * n_clips;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == MK_F32)
    MK_LAUNCH((logmel_finish_kernel<float>), grid, dim3(256), 0, MK_ST, ws_logspec, ws_max,
              (float*)out, per, total);
  else if (dtype == MK_BF16)
    MK_LAUNCH((logmel_finish_kernel<bf16>), grid, dim3(256), 0, MK_ST, ws_logspec, ws_max,
              (bf16*)out, per, total);
  else if (dtype == MK_F16)
    MK_LAUNCH((logmel_finish_kernel<_Float16>), grid, dim3(256), 0, MK_ST, ws_logspec, ws_max,
              (_Float16*)out, per, total);
  else return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}
