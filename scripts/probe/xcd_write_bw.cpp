// probe: store bandwidth when only `nx` of the 8 XCDs write (workgroup id & 7 = XCD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void fill(float4* p, long per_wg_f4, int nx, int reps) {
  const int xcd = blockIdx.x & 7;
  if (xcd >= nx) return;
  const long slot = (long)(blockIdx.x >> 3) * nx + xcd;
  float4* q = p + slot * per_wg_f4;
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int r = 0; r < reps; ++r)
    for (long i = threadIdx.x; i < per_wg_f4; i += 512) q[i + (long)r * 0] = v;
}
__global__ __launch_bounds__(512) void rd(const float4* p, float* out, long per_wg_f4, int nx) {
  const int xcd = blockIdx.x & 7;
  if (xcd >= nx) return;
  const long slot = (long)(blockIdx.x >> 3) * nx + xcd;
  const float4* q = p + slot * per_wg_f4;
  float s = 0.f;
  for (long i = threadIdx.x; i < per_wg_f4; i += 512) { float4 v = q[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) out[0] = s;
}
int main() {
  const long per_wg = 128 << 10;  // bytes per workgroup (one 256x256 bf16 tile)
  float4* buf; float* out;
  hipMalloc(&buf, 1L << 30); hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rounds : {1, 4}) {
    for (int nx : {1, 2, 4, 8}) {
      const int wgs = 256 * rounds;    // 32 * rounds workgroups on each active XCD
      float best = 1e9, bestr = 1e9;
      for (int it = 0; it < 20; ++it) {
        hipEventRecord(a); fill<<<wgs, 512>>>(buf, per_wg / 16, nx, 1); hipEventRecord(b);
        hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); if (t < best) best = t;
        hipEventRecord(a); rd<<<wgs, 512>>>(buf, out, per_wg / 16, nx); hipEventRecord(b);
        hipEventSynchronize(b); hipEventElapsedTime(&t, a, b); if (t < bestr) bestr = t;
      }
      const double bytes = (double)per_wg * 32 * rounds * nx;
      printf("rounds %d xcds %d: %.1f MiB  write %.2f us = %.2f TB/s   read %.2f us = %.2f TB/s\n", rounds, nx,
             bytes / 1048576, best * 1e3, bytes / best / 1e9, bestr * 1e3, bytes / bestr / 1e9);
    }
  }
  return 0;
}
