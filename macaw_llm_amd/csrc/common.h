// Shared device helpers for the macaw_hip kernels (gfx950 / CDNA4 only).
// Wave size is hard-coded to 64 (cdna_hip_programming.md §1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MK_WAVE 64

// Error codes returned through the C ABI (include/macaw_hip.h).
#define MK_OK 0
#define MK_ERR_BAD_ARG (-1)
#define MK_ERR_UNSUPPORTED (-2)
#define MK_ERR_LAUNCH (-3)

// dtype codes of the C ABI
#define MK_F32 0
#define MK_BF16 1
#define MK_F16 2
#define MK_FP8 3   /* OCP e4m3fn bytes (GEMM operands only) */

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define MK_DEV __device__ __forceinline__

// hipGetLastError() is sticky across *any* earlier runtime call of the process (e.g. a
// hipEventQuery of PyTorch's allocator returning hipErrorNotReady), so every launch first
// clears it and mk_check_launch() then reports only this launch's status.
#define MK_LAUNCH(...)          \
  do {                          \
    (void)hipGetLastError();    \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)
static inline int mk_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MK_OK : MK_ERR_LAUNCH;
}

// ---- scalar load/store as float regardless of storage type -------------
template <typename T> MK_DEV float to_f32(T v);
template <> MK_DEV float to_f32<float>(float v) { return v; }
template <> MK_DEV float to_f32<bf16>(bf16 v) { return (float)v; }
template <> MK_DEV float to_f32<_Float16>(_Float16 v) { return (float)v; }
template <typename T> MK_DEV T from_f32(float v);
template <> MK_DEV float from_f32<float>(float v) { return v; }
template <> MK_DEV bf16 from_f32<bf16>(float v) { return (bf16)v; }  // RNE (v_cvt_pk_bf16_f32)
template <> MK_DEV _Float16 from_f32<_Float16>(float v) { return (_Float16)v; }

// Round-trip through the storage type: the value the eager reference would hold after an
// op in that dtype.  The bf16 form is done on the bits (RNE) because hipcc folds
// (float)(__bf16)x back to x under its excess-precision rules.
// erf(x) to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26: one exp, one reciprocal, five FMAs) for the
// exact GELU of the Whisper MLPs: libm's erff costs ~4x as many instructions, and the GELU epilogue of
// 48000 x 2048 outputs was VALU-bound on it.  The result feeds a value that is rounded to bf16
// (2^-9): the difference to erff is three orders of magnitude below that.
MK_DEV float mk_erf(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}

template <typename T> MK_DEV float rnd(float v);
template <> MK_DEV float rnd<float>(float v) { return v; }
// Round 4: the hardware conversion (v_cvt_pk_bf16_f32: RNE, quiet NaN) + one shift, in asm for the same reason as
// the fp16 form below.  Rounds 1-3 rounded on the bits (compare, shift, and, add, and: 6 VALU instructions); the
// in-place RoPE kernel rounds six times per output pair and was VALU-bound on it at 4.4 TB/s
// (profiles/r04_rope_rnd_ab.txt).  Same values for every finite input; a NaN comes back quiet with a truncated
// payload instead of unchanged.
template <> MK_DEV float rnd<bf16>(float v) {
  uint32_t r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(v));
  return __uint_as_float(r << 16);
}

// fp16 (the reference's `--fp16 True` / `.to(torch.float16)`, train.sh:36, llm_trainer.py:411-412):
// RNE through v_cvt_f16_f32 / v_cvt_f32_f16; written in asm so that fast-math cannot fold the
// round trip away (same reason as the bit trick of the bf16 form).
template <> MK_DEV float rnd<_Float16>(float v) {
  float r;
  asm volatile("v_cvt_f16_f32 %0, %1\n\tv_cvt_f32_f16 %0, %0" : "=v"(r) : "v"(v));
  return r;
}

// Vector-of-VEC access: VEC elements of T moved as one 16-byte (bf16x8) or
// 16-byte (float4) transaction.  VecIO<T>::N elements per 16 B.
template <typename T> struct VecIO;
template <> struct VecIO<float> {
  static constexpr int N = 4;
  MK_DEV static void load(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  MK_DEV static void store(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  }
  // the 16 bytes as they are (a load kept in flight across other work) and their conversion later
  MK_DEV static uint4 load_raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
  MK_DEV static void unpack(uint4 r, float (&o)[4]) {
    o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
  }
};
template <> struct VecIO<bf16> {
  static constexpr int N = 8;
  MK_DEV static void load(const bf16* p, float (&o)[8]) {
    bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  MK_DEV static void store(bf16* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)o[i];
    *reinterpret_cast<bf16x8*>(p) = v;
  }
  MK_DEV static uint4 load_raw(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }
  MK_DEV static void unpack(uint4 r, float (&o)[8]) {
    const bf16x8 v = __builtin_bit_cast(bf16x8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};

template <> struct VecIO<_Float16> {
  static constexpr int N = 8;
  MK_DEV static void load(const _Float16* p, float (&o)[8]) {
    f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
  MK_DEV static void store(_Float16* p, const float (&o)[8]) {
    f16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)o[i];
    *reinterpret_cast<f16x8*>(p) = v;
  }
  MK_DEV static uint4 load_raw(const _Float16* p) { return *reinterpret_cast<const uint4*>(p); }
  MK_DEV static void unpack(uint4 r, float (&o)[8]) {
    const f16x8 v = __builtin_bit_cast(f16x8, r);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  }
};

// ---- 16-bit MFMA element traits: the kernels of gemm*.hip / attention.hip / decode.hip are written
// once over an element type e16 (bf16 or f16, `*_impl.inc` included twice); everything that depends on
// the element type beyond plain conversions goes through here.  Same instruction rates for both.
template <typename T> struct E16;
template <> struct E16<bf16> {
  typedef bf16x8 x8; typedef bf16x4 x4;
  static constexpr int dtype = MK_BF16;
  static constexpr bool narrow_exponent = false;
  MK_DEV static f32x16 mma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  MK_DEV static f32x4 mma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  MK_DEV static x4 tr_read(const char* lds) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds));
  }
  // The same read for kernels that order their LDS-DMA by hand (counted vmcnt + barrier: gemm_v7 / gemm_v8).
  // hipcc (ROCm 7.2, SIInsertWaitcnts) orders an LDS read whose memory operand has NO alias info behind EVERY
  // outstanding LDS-DMA: it put an `s_waitcnt vmcnt(0)` in front of the first transpose read of each K-tile
  // and drained the `buffer_load ... lds` queue those loops keep full on purpose (plain C++ LDS loads carry
  // TBAA and never had it).  `__restrict__` gives the inlined intrinsic call !alias.scope metadata; the
  // hand-placed wait is then the only one, so use this form ONLY behind such a wait.
  MK_DEV static x4 tr_read_nw(const char* __restrict__ lds) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds));
  }
};
template <> struct E16<_Float16> {
  typedef f16x8 x8; typedef f16x4 x4;
  static constexpr int dtype = MK_F16;
  // 5 exponent bits: a value rounded to this type inside a kernel keeps its natural scaling (the attention
  // backward multiplies dS by the softmax scale BEFORE rounding it, as the reference's fp16 graph does: without it
  // dS has 1/scale = 11x less headroom under the 2^16 dynamic loss scale -- ADVICE r4)
  static constexpr bool narrow_exponent = true;
  MK_DEV static f32x16 mma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  MK_DEV static f32x4 mma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  MK_DEV static x4 tr_read(const char* lds) {
    // (ds_read_b64_tr_b16 moves 16-bit lanes, the element format is irrelevant to it)
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                         (__attribute__((address_space(3))) bf16x4*)(lds)));
  }
  MK_DEV static x4 tr_read_nw(const char* __restrict__ lds) {   // see E16<bf16>::tr_read_nw
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                         (__attribute__((address_space(3))) bf16x4*)(lds)));
  }
};

// ---- wave / block reductions -------------------------------------------
MK_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
MK_DEV float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Block-wide sum; `red` is LDS scratch of >= 16 floats. All threads get the result.
template <int NT> MK_DEV float block_sum(float v, float* red) {
  constexpr int NW = NT / 64;
  v = wave_sum(v);
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) r += red[i];
  return r;
}
template <int NT> MK_DEV float block_max(float v, float* red) {
  constexpr int NW = NT / 64;
  v = wave_max(v);
  if constexpr (NW == 1) return v;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, red[i]);
  return r;
}

// Counter-based RNG for attention dropout: one 32-bit hash per (seed, index).
// Keep/drop is a pure function of (seed, linear element index) so forward and
// backward kernels regenerate the identical mask without storing it.
MK_DEV uint32_t mk_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 16);
}
MK_DEV bool mk_keep(uint64_t seed, uint64_t idx, uint32_t keep_threshold) {
  // keep iff hash < keep_threshold, keep_threshold = (1-p) * 2^32 (saturated)
  return mk_hash32(seed, idx) < keep_threshold;
}

static inline int mk_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// live launch profiler (implemented in gemm.hip): see mk_prof_begin / mk_prof_sum
namespace mkp {
int begin(hipStream_t st, int kind, double flops, int M, int N, int K, int nb, int layout, int cfg);
void set_cfg(int idx, int cfg);
void end(int idx, hipStream_t st);
}
