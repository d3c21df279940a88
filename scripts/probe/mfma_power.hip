// Round 6 probe: is gemm_v9's power-bound K loop (profiles/r05_gemm_v9.txt: 1.50 GHz against the vendor kernel's 1.70)
// a property of the MFMA SHAPE?  The vendor's 256x256x64 kernel (profiles/r06_vendor_isa.txt) issues
// 128 x v_mfma_f32_16x16x32_bf16 per wave and K-tile where v9 issues 64 x v_mfma_f32_32x32x16_bf16 -- same FLOPs, same
// LDS bytes (32 ds_read_b128 per wave and K-tile), but half the accumulator traffic per FLOP and twice the operand reads.
//
// One workgroup of 4 waves per CU (one wave per SIMD, as v9), wave tile 128 x 128 in 256 accumulator registers, random
// bf16 operands (power depends on toggling: never zeros).  Variants, timed for ~100 ms each, throughput = clock:
//   mode 0 / 1: 32x32x16 / 16x16x32, operands resident in registers (pure matrix pipe)
//   mode 2 / 3: the same + the K-tile's 32 ds_read_b128 per wave from LDS (v9's / the vendor's LDS -> register traffic)
//   mode 4    : mode 3 + 128 v_perm_b32 per K-tile (the vendor's register transposition of reduction-major operands)
//
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/mfma_power.hip -o scripts/probe/_probe_mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void mfma_power(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, l = t & 63, w = t >> 6;
  // 64 KiB of random bf16 in LDS (A and B images of one K-tile: 2 x 256 x 64 x 2 B)
  u32x4* s4 = reinterpret_cast<u32x4*>(smem);
  for (int i = t; i < 4096; i += 256) s4[i] = src[i];
  __syncthreads();
  constexpr bool SMALL = (MODE & 1) != 0;          // 16x16x32
  constexpr bool LDS = MODE >= 2;
  constexpr bool PERM = MODE == 4;
  // fragments of one K-tile: 16 of A + 16 of B, 4 VGPRs each
  bf16x8 fa[16], fb[16];
  const char* base = smem + ((w * 4096 + l * 16) & 0x3fff);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    fa[i] = *reinterpret_cast<const bf16x8*>(base + (i & 7) * 1024 + (i >> 3) * 8192);
    fb[i] = *reinterpret_cast<const bf16x8*>(base + 32768 + (i & 7) * 1024 + (i >> 3) * 8192);
  }
  f32x16 cb[SMALL ? 1 : 16];
  f32x4 cs[SMALL ? 64 : 1];
#pragma unroll
  for (int i = 0; i < (SMALL ? 1 : 16); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) cb[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < (SMALL ? 64 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[i][e] = 0.f;
  int off = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (LDS) {
      // the K-tile's fragment reads; the address moves so that the loads cannot be hoisted
      const char* p = smem + ((w * 4096 + l * 16 + off) & 0x3fff);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(p + (i & 7) * 1024 + (i >> 3) * 8192);
        fb[i] = *reinterpret_cast<const bf16x8*>(p + 32768 + (i & 7) * 1024 + (i >> 3) * 8192);
      }
      off = 2048 - off;                    // toggles between two images
    }
    if constexpr (PERM) {
      // 128 v_perm_b32 per K-tile: every fragment dword rebuilt from two dwords of a neighbour fragment
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        u32x4 x = __builtin_bit_cast(u32x4, fa[i]), y = __builtin_bit_cast(u32x4, fb[i]);
        u32x4 r, q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          r[e] = __builtin_amdgcn_perm(x[e], x[(e + 1) & 3], 0x05040100u);
          q[e] = __builtin_amdgcn_perm(y[e], y[(e + 1) & 3], 0x07060302u);
        }
        fa[i] = __builtin_bit_cast(bf16x8, r);
        fb[i] = __builtin_bit_cast(bf16x8, q);
      }
    }
    if constexpr (!SMALL) {
      // 4 k-steps of 16: A fragments 4k .. 4k+3 (rows), B fragments 4k .. 4k+3 (columns) -> 64 MFMAs
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            cb[i * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[k * 4 + i], fb[k * 4 + j], cb[i * 4 + j], 0, 0, 0);
    } else {
      // 2 k-steps of 32: A fragments 8k .. 8k+7, B fragments 8k .. 8k+7 -> 128 MFMAs
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            cs[i * 8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[k * 8 + i], fb[k * 8 + j], cs[i * 8 + j], 0, 0, 0);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < (SMALL ? 1 : 16); ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc += cb[i][e];
#pragma unroll
  for (int i = 0; i < (SMALL ? 64 : 1); ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += cs[i][e];
  if (acc == 12345.678f) out[blockIdx.x * 256 + t] = acc;      // never true: keeps the work alive
}

template <int MODE>
static double run(const u32x4* src, float* out, int grid, int iters, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_power<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(mfma_power<MODE>, dim3(grid), dim3(256), 65536, 0, src, out, iters / 8);     // warm
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_power<MODE>, dim3(grid), dim3(256), 65536, 0, src, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // one K-tile per wave and iteration: 128 x 128 x 64 x 2 FLOP
  const double flops = (double)reps * grid * 4.0 * iters * 128.0 * 128.0 * 64.0 * 2.0;
  return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;      // ~17 ms per launch at full rate
  const int reps = argc > 3 ? atoi(argv[3]) : 8;
  std::vector<unsigned> h(4096 * 4);
  unsigned s = 12345u;
  const bool zeros = getenv("MP_ZEROS") != nullptr;      // all-zero operands: no data toggling -> how much of the limit is POWER
  for (auto& v : h) {
    // two random bf16 in [-2, 2): sign + exponent 0x3f / 0x3e / 0x40 + random mantissa
    unsigned a[2];
    for (int k = 0; k < 2; ++k) {
      s = s * 1664525u + 1013904223u;
      const unsigned e = 0x3e + ((s >> 9) % 3);
      a[k] = ((s >> 31) << 15) | (e << 7) | ((s >> 12) & 0x7f);
    }
    v = zeros ? 0u : (a[0] | (a[1] << 16));
  }
  u32x4* src;
  float* out;
  CK(hipMalloc(&src, 65536));
  CK(hipMalloc(&out, (size_t)grid * 256 * 4));
  CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
  printf("operands: %s\nmode,what,tflops\n", zeros ? "all zeros" : "random bf16 in [-4, 4)");
  for (int round = 0; round < 2; ++round) {
    printf("0,32x32x16 registers only,%.1f\n", run<0>(src, out, grid, iters, reps));
    printf("1,16x16x32 registers only,%.1f\n", run<1>(src, out, grid, iters, reps));
    printf("2,32x32x16 + 32 ds_read_b128 per K-tile,%.1f\n", run<2>(src, out, grid, iters, reps));
    printf("3,16x16x32 + 32 ds_read_b128 per K-tile,%.1f\n", run<3>(src, out, grid, iters, reps));
    printf("4,16x16x32 + 32 ds_read_b128 + 128 v_perm_b32 per K-tile,%.1f\n", run<4>(src, out, grid, iters, reps));
  }
  return 0;
}
