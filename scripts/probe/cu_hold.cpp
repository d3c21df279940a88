// CU-theft probe: what does the 256 x 256 tile GEMM lose when another resident kernel
// HOLDS k compute units -- no bandwidth, no matrix work, just occupancy, which is what an RCCL channel does to
// a kernel that needs a whole CU (one workgroup per CU, all 160 KiB of LDS)?  And how much of it comes back when
// the launcher plans its rounds for the CUs that are actually free (mk_gemm_set_cus)?
//
//   hipcc -O2 --offload-arch=gfx950 scripts/probe/cu_hold.cpp -o scripts/probe/_probe_cu_hold \
//         -Lmacaw_llm_amd -lmacaw_hip -Wl,-rpath,'$ORIGIN/../../macaw_llm_amd'
//   scripts/probe/_probe_cu_hold > gpurun_out/cu_hold.csv
//
// The holder: k workgroups of one wave with 1 KiB of LDS (any LDS excludes a 160-KiB workgroup from the CU) that
// sleep until a host flag flips; launched on its own stream before the timed GEMMs.  The GEMMs: the step's big
// shapes, timed with events on a second stream while the holder is resident.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../include/macaw_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef __bf16 bf16;

__global__ void hold_kernel(volatile int* stop, int* where) {
  __shared__ int pad[256];
  pad[threadIdx.x & 255] = 0;
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    where[blockIdx.x] = (int)((xcc & 0xf) << 16 | (hw & 0xffff));      // XCD and CU/SE bits of this workgroup
    while (!*stop) __builtin_amdgcn_s_sleep(32);
  }
  __syncthreads();
}
__global__ void fill(bf16* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761U + seed;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    p[i] = (bf16)(scale * ((int)(x & 0xffff) - 32768) / 32768.0f);
  }
}

int main() {
  struct Sh { int M, N, K, layout; };
  const std::vector<Sh> shapes = {{4096, 4096, 4096, 0}, {4608, 12288, 4096, 0}, {4608, 4096, 4096, 0},
                                  {4608, 4096, 12288, 1}, {12288, 4096, 4608, 3}, {8192, 8192, 4096, 0}};
  hipStream_t sh, sg;
  CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
  int* stop;
  CK(hipHostMalloc((void**)&stop, sizeof(int), hipHostMallocMapped));
  int* where;
  CK(hipMalloc(&where, 64 * sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("M,N,K,layout,held_cus,planned_cus,ms,tflops\n");
  for (auto& s : shapes) {
    const int a_red = (s.layout >> 1) & 1, b_red = s.layout & 1;
    const long lda = a_red ? s.M : s.K, ldb = b_red ? s.N : s.K;
    const long na = (long)s.M * s.K, nb = (long)s.N * s.K, nc = (long)s.M * s.N;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, na * 2)); CK(hipMalloc(&B, nb * 2)); CK(hipMalloc(&C, nc * 2));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, sg, A, na, 1u, 1.0f);
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, sg, B, nb, 2u, 0.02f);
    CK(hipStreamSynchronize(sg));
    mk_gemm_desc d{};
    d.A = A; d.B = B; d.C = C; d.M = s.M; d.N = s.N; d.K = s.K; d.lda = lda; d.ldb = ldb; d.ldc = s.N;
    d.a_red_major = a_red; d.b_red_major = b_red; d.nb1 = d.nb2 = 1; d.alpha = 1.0f; d.dtype = MK_BF16;
    for (int held : {0, 8, 16, 32}) {
      for (int planned : {256, 256 - held}) {
        if (held == 0 && planned != 256) continue;
        if (held != 0 && planned == 256 - held && held == 0) continue;
        mk_gemm_set_cus(planned);
        *stop = 0;
        if (held) {
          hipLaunchKernelGGL(hold_kernel, dim3(held), dim3(64), 0, sh, stop, where);
          std::this_thread::sleep_for(std::chrono::microseconds(300));      // the holder is resident
        }
        double best = 1e30;
        for (int rd = 0; rd < 3; ++rd) {
          for (int i = 0; i < 2; ++i) mk_gemm(&d, sg);
          CK(hipEventRecord(e0, sg));
          for (int i = 0; i < 8; ++i) mk_gemm(&d, sg);
          CK(hipEventRecord(e1, sg));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms / 8 < best) best = ms / 8;
        }
        *stop = 1;
        CK(hipStreamSynchronize(sh));
        printf("%d,%d,%d,%d,%d,%d,%.4f,%.1f\n", s.M, s.N, s.K, s.layout, held, planned, best,
               2.0 * s.M * s.N * s.K / (best * 1e-3) / 1e12);
        fflush(stdout);
      }
      if (held == 32 && &s == &shapes[0]) {      // where did the holders land? (XCD ids of the 32 workgroups)
        std::vector<int> w(32);
        CK(hipMemcpy(w.data(), where, 32 * sizeof(int), hipMemcpyDeviceToHost));
        fprintf(stderr, "holder placement (xcd:hw_id):");
        for (int i = 0; i < 32; ++i) fprintf(stderr, " %d:%04x", w[i] >> 16, w[i] & 0xffff);
        fprintf(stderr, "\n");
      }
    }
    mk_gemm_set_cus(0);
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  return 0;
}
