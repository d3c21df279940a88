// Round 6 probe: is the decode step's weight stream (4.2-4.8 TB/s per skinny GEMM, 3.1 ms per token at B = 1) held back by
// the ACCESS PATTERN of a row-major [N][K] weight?  gemm_skinny16_kernel gives a workgroup 16 weight rows: one load
// instruction of a wave touches 16 rows x 64 B, K * 2 bytes apart.  PACKED = the same kernel over a pre-tiled image
// [N / 16][K / 64][16 rows][64 k]: a workgroup's whole stream is ONE contiguous 32 K-byte... (16 x K x 2 B) range and a
// load pair of a wave covers 2 KiB of it.  Same grid, same waves, same loads in flight, same MFMAs; 32 distinct weight
// matrices per timing (as the 32 layers of a step: every launch streams cold bytes).
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/decode_weight_layout.hip -o scripts/probe/_probe_decode_weight_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool PACKED, int NBUF>
__global__ __launch_bounds__(512) void stream_kernel(const bf16* __restrict__ W, const bf16* __restrict__ x, float* out, int N, int K) {
  __shared__ float red[8][4][64];
  constexpr int NW = 8, U = 2, STEP = NW * U;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = l & 15, kq = l >> 4;
  const int n0 = blockIdx.x * 16, nkb = K / 64;
  const bf16* wp = PACKED ? W + (long)blockIdx.x * 16 * K + r16 * 64 + 8 * kq : W + (long)(n0 + r16) * K + 8 * kq;
  const long kstep = PACKED ? 1024 : 64;
  const bf16* xp = x + 8 * kq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  bf16x8 wbuf[NBUF][2 * U], xbuf[NBUF][2 * U];
  auto load = [&](bf16x8 (&wf)[2 * U], bf16x8 (&xf)[2 * U], int kb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = min(kb + u * NW, nkb - 1);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        wf[2 * u + hh] = *reinterpret_cast<const bf16x8*>(wp + kk * kstep + 32 * hh);
        xf[2 * u + hh] = *reinterpret_cast<const bf16x8*>(xp + kk * 64 + 32 * hh);
      }
    }
  };
  auto mma = [&](const bf16x8 (&wf)[2 * U], const bf16x8 (&xf)[2 * U], int kb) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kb + u * NW < nkb) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u], xf[2 * u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u + 1], xf[2 * u + 1], acc, 0, 0, 0);
      }
  };
  int kb = w;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (kb + i * STEP < nkb) load(wbuf[i], xbuf[i], kb + i * STEP);
  while (kb < nkb) {
#pragma unroll
    for (int i = 0; i < NBUF; ++i) {
      if (kb + (NBUF - 1) * STEP < nkb) load(wbuf[(i + NBUF - 1) % NBUF], xbuf[(i + NBUF - 1) % NBUF], kb + (NBUF - 1) * STEP);
      mma(wbuf[i], xbuf[i], kb);
      kb += STEP;
      if (kb >= nkb) break;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[w][e][l] = acc[e];
  __syncthreads();
  for (int t = threadIdx.x; t < 256; t += 512) {
    const int e = (t >> 6) & 3, ll = t & 63;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww][e][ll];
    if ((ll & 15) == 0) out[n0 + 4 * (ll >> 4) + e] = v;
  }
}

// reads `bytes` of W with `gridDim.x` workgroups of 256 threads and throws them away (the Infinity Cache keeps them)
__global__ __launch_bounds__(256) void touch_kernel(const uint4* __restrict__ W, long n16, unsigned* sink) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
    const uint4 v = W[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// the chain with the first `frac` of matrix i + 1 touched on a second stream while matrix i is consumed
float run_prefetch(const std::vector<bf16*>& Ws, const bf16* x, float* out, int N, int K, int reps, double frac, int pf_blocks) {
  hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(Ws.size());
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  unsigned* sink = reinterpret_cast<unsigned*>(out) + 100000;
  const long n16 = (long)((double)N * K * 2 * frac) / 16;
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    for (size_t i = 0; i < Ws.size(); ++i) {
      CK(hipEventRecord(ev[i], sa));                       // kernel i is about to start
      if (i + 1 < Ws.size() && n16 > 0) {
        CK(hipStreamWaitEvent(sb, ev[i], 0));
        hipLaunchKernelGGL(touch_kernel, dim3(pf_blocks), dim3(256), 0, sb, reinterpret_cast<const uint4*>(Ws[i + 1]), n16, sink);
      }
      hipLaunchKernelGGL((stream_kernel<false, 3>), dim3(N / 16), dim3(512), 0, sa, Ws[i], x, out, N, K);
    }
    CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best / (float)Ws.size();
}

template <bool PACKED, int NBUF>
float run(const std::vector<bf16*>& Ws, const bf16* x, float* out, int N, int K, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    for (bf16* W : Ws) hipLaunchKernelGGL((stream_kernel<PACKED, NBUF>), dim3(N / 16), dim3(512), 0, 0, W, x, out, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best / (float)Ws.size();
}

int main() {
  const int shapes[][2] = {{4096, 4096}, {4096, 11008}, {12288, 4096}, {22016, 4096}, {32000, 4096}};
  bf16* x; float* out;
  CK(hipMalloc(&x, 65536)); CK(hipMemset(x, 0x3c, 65536)); CK(hipMalloc(&out, 1 << 20));
  printf("N,K,MB,rowmajor_us,rowmajor_TBs,packed_us,packed_TBs,rowmajor4_us,packed4_us\n");
  for (auto& s : shapes) {
    const int N = s[0], K = s[1];
    const size_t bytes = (size_t)N * K * 2;
    std::vector<bf16*> Ws(32);
    for (auto& W : Ws) { CK(hipMalloc(&W, bytes)); CK(hipMemset(W, 0x3c, bytes)); }
    CK(hipDeviceSynchronize());
    const float a = run<false, 3>(Ws, x, out, N, K, 5), b = run<true, 3>(Ws, x, out, N, K, 5);
    const float a4 = run<false, 4>(Ws, x, out, N, K, 5), b4 = run<true, 4>(Ws, x, out, N, K, 5);
    printf("%d,%d,%.1f,%.2f,%.2f,%.2f,%.2f,%.2f,%.2f\n", N, K, bytes / 1e6, a * 1e3, bytes / (a * 1e-3) / 1e12, b * 1e3,
           bytes / (b * 1e-3) / 1e12, a4 * 1e3, b4 * 1e3);
    // prefetch of the next matrix on a second stream: fraction x blocks -> us per launch of the chain
    for (double frac : {0.0, 0.1, 0.25, 0.5})
      for (int blocks : {64, 256}) {
        if (frac == 0.0 && blocks != 64) continue;
        const float c = run_prefetch(Ws, x, out, N, K, 5, frac, blocks);
        printf("  prefetch frac %.2f blocks %d: %.2f us per launch (%.2f TB/s)\n", frac, blocks, c * 1e3, bytes / (c * 1e-3) / 1e12);
      }
    for (auto& W : Ws) CK(hipFree(W));
  }
  return 0;
}
