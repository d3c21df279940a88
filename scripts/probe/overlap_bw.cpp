// probe: how much does a throttled AdamW-like stream (N workgroups on a low-priority stream) slow the
// 256x256 GEMM it runs beside, and what bandwidth does it get?  (can the optimizer hide in the backward?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/macaw_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void adam_like(bf16* p, float* w, float* m, float* v, const bf16* g, long n, int reps) {
  __builtin_amdgcn_s_setprio(0);
  for (int r = 0; r < reps; ++r)
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
      const bf16x4 gv = *reinterpret_cast<const bf16x4*>(g + i);
      float4 wv = *reinterpret_cast<float4*>(w + i), mv = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      float gg[4] = {(float)gv[0], (float)gv[1], (float)gv[2], (float)gv[3]};
      float* wp = &wv.x; float* mp = &mv.x; float* vp = &vv.x;
      bf16x4 o;
      for (int k = 0; k < 4; ++k) {
        mp[k] = 0.9f * mp[k] + 0.1f * gg[k];
        vp[k] = 0.999f * vp[k] + 0.001f * gg[k] * gg[k];
        wp[k] -= 1e-5f * mp[k] / (sqrtf(vp[k]) + 1e-8f);
        o[k] = (bf16)wp[k];
      }
      *reinterpret_cast<float4*>(w + i) = wv; *reinterpret_cast<float4*>(m + i) = mv; *reinterpret_cast<float4*>(v + i) = vv;
      *reinterpret_cast<bf16x4*>(p + i) = o;
    }
}
__global__ void fillk(bf16* p, long n, float s) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = (bf16)(s * (float)((i * 2654435761u >> 8) & 255) / 255.f - s / 2);
}

int main() {
  const int M = 4608, N = 12288, K = 4096;
  bf16 *A, *B, *C;
  CK(hipMalloc(&A, (long)M * K * 2)); CK(hipMalloc(&B, (long)N * K * 2)); CK(hipMalloc(&C, (long)M * N * 2));
  fillk<<<2048, 256>>>(A, (long)M * K, 1.f); fillk<<<2048, 256>>>(B, (long)N * K, 0.04f);
  const long n = 256L << 20;            // 256 M elements: 7.2 GB of optimizer traffic per pass
  bf16 *p, *g; float *w, *m, *v;
  CK(hipMalloc(&p, n * 2)); CK(hipMalloc(&g, n * 2)); CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
  CK(hipMemset(w, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4)); fillk<<<2048, 256>>>(g, n, 0.01f);
  void* ws; CK(hipMalloc(&ws, 72L << 20)); CK(hipMemset(ws, 0, 4096));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s1, s2; CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  mk_gemm_desc d{};
  d.A = A; d.B = B; d.C = C; d.M = M; d.N = N; d.K = K; d.lda = K; d.ldb = K; d.ldc = N; d.nb1 = d.nb2 = 1; d.alpha = 1.f;
  d.dtype = MK_BF16; d.ws = ws; d.ws_bytes = 72L << 20;
  const int NG = 40;
  auto gemms = [&]() { for (int i = 0; i < NG; ++i) mk_gemm(&d, s1); };
  gemms(); CK(hipStreamSynchronize(s1));
  CK(hipEventRecord(a0, s1)); gemms(); CK(hipEventRecord(a1, s1)); CK(hipEventSynchronize(a1));
  float t_alone; CK(hipEventElapsedTime(&t_alone, a0, a1));
  printf("GEMM alone: %.3f ms per launch (%.0f TF)\n", t_alone / NG, 2.0 * M * N * K / (t_alone / NG * 1e-3) / 1e12);
  struct Cfg { int nwg, tpb; };
  for (Cfg c : {Cfg{64, 256}, Cfg{128, 256}, Cfg{256, 64}, Cfg{256, 128}, Cfg{512, 64}, Cfg{256, 256}}) {
    const int nwg = c.nwg, tpb = c.tpb;
    CK(hipEventRecord(b0, s2)); adam_like<<<nwg, tpb, 0, s2>>>(p, w, m, v, g, n, 1); CK(hipEventRecord(b1, s2)); CK(hipEventSynchronize(b1));
    float tb; CK(hipEventElapsedTime(&tb, b0, b1));
    const double bw_alone = 28.0 * n / (tb * 1e-3) / 1e12;
    const int reps = (int)(t_alone * 1.5 / tb) + 1;
    CK(hipEventRecord(b0, s2)); adam_like<<<nwg, tpb, 0, s2>>>(p, w, m, v, g, n, reps); CK(hipEventRecord(b1, s2));
    CK(hipEventRecord(a0, s1)); gemms(); CK(hipEventRecord(a1, s1));
    CK(hipEventSynchronize(a1)); CK(hipEventSynchronize(b1));
    float tg, tbo; CK(hipEventElapsedTime(&tg, a0, a1)); CK(hipEventElapsedTime(&tbo, b0, b1));
    const double slow = tg / t_alone - 1, bw = 28.0 * n * reps / (tbo * 1e-3) / 1e12;
    // a 250 ms step with 188 ms of GEMM and 195 GB of optimizer traffic: overlap window T = 195 GB / bw
    const double T = 195e9 / (bw * 1e12) * 1e3, extra = T - T / (1 + slow);
    printf("bg %4d WGs x %3d thr: alone %.2f TB/s | beside GEMM: GEMM %.3f ms (%+.1f %%), bg %.2f TB/s -> window %.0f ms, GEMM loss %.1f ms vs 34 ms saved\n",
           nwg, tpb, bw_alone, tg / NG, slow * 100, bw, T, extra);
  }
  return 0;
}
