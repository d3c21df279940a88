// Round 6 probe: how much HBM bandwidth do c CUs pull on their own?  (Can a 197.8 GB AdamW update hide on a few CUs
// beside the backward's GEMMs?  profiles/r06_local_overlap_confined.txt)  One 1024-thread block per CU (96 KiB of LDS
// requested so that a second block cannot join it), grid = c blocks, AdamW-like stream: per 4 elements read 8 B + 3 x 16 B,
// write 3 x 16 B + 8 B (28 B per element), 2 groups in flight per lane per iteration.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/cu_stream_bw.hip -o scripts/probe/_probe_cu_stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(1024) void stream_kernel(unsigned short* w, float* a, float* b, float* c, const unsigned short* g, long n) {
  extern __shared__ char smem[];
  if (n < 0) smem[threadIdx.x] = 0;
  const long nch = n / 4;
  for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < nch; i += (long)gridDim.x * 2048) {
    const long j = i + (long)gridDim.x * 1024 < nch ? i + (long)gridDim.x * 1024 : i;
    const uint2 g0 = *reinterpret_cast<const uint2*>(g + 4 * i), g1 = *reinterpret_cast<const uint2*>(g + 4 * j);
    float4 a0 = *reinterpret_cast<float4*>(a + 4 * i), a1 = *reinterpret_cast<float4*>(a + 4 * j);
    float4 b0 = *reinterpret_cast<float4*>(b + 4 * i), b1 = *reinterpret_cast<float4*>(b + 4 * j);
    float4 c0 = *reinterpret_cast<float4*>(c + 4 * i), c1 = *reinterpret_cast<float4*>(c + 4 * j);
    const float s0 = __uint_as_float(g0.x << 16), s1 = __uint_as_float(g1.x << 16);
    a0.x += s0; b0.x = b0.x * 0.9f + s0; c0.x = c0.x * 0.99f + s0 * s0;
    a1.x += s1; b1.x = b1.x * 0.9f + s1; c1.x = c1.x * 0.99f + s1 * s1;
    *reinterpret_cast<float4*>(a + 4 * i) = a0; *reinterpret_cast<float4*>(b + 4 * i) = b0; *reinterpret_cast<float4*>(c + 4 * i) = c0;
    *reinterpret_cast<uint2*>(w + 4 * i) = make_uint2(__float_as_uint(a0.x) >> 16, g0.y);
    if (j != i) {
      *reinterpret_cast<float4*>(a + 4 * j) = a1; *reinterpret_cast<float4*>(b + 4 * j) = b1; *reinterpret_cast<float4*>(c + 4 * j) = c1;
      *reinterpret_cast<uint2*>(w + 4 * j) = make_uint2(__float_as_uint(a1.x) >> 16, g1.y);
    }
  }
}

int main() {
  const long n = 256L << 20;       // 256 Mi elements: 7.5 GB of traffic per pass
  unsigned short *w, *g; float *a, *b, *c;
  CK(hipMalloc(&w, n * 2)); CK(hipMalloc(&g, n * 2)); CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
  CK(hipMemset(w, 0, n * 2)); CK(hipMemset(g, 0x3c, n * 2)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0, n * 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("blocks_of_1024_threads(=CUs),ms,TB_per_s,GB_per_s_per_CU\n");
  for (int cus : {256, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512}) {
    const long nn = cus >= 64 ? n : n / (64 / cus);          // keep the small grids short
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream_kernel, dim3(cus), dim3(1024), 98304, 0, w, a, b, c, g, nn);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 28.0 * nn;
    printf("%d,%.3f,%.3f,%.1f\n", cus, ms, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / (cus > 256 ? 256 : cus));
  }
  return 0;
}
