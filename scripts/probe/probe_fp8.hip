// Hardware probe (gfx950): fp8 conversion format, operand / result layout and issue rate of
// v_mfma_scale_f32_32x32x64_f8f6f4.  Build: hipcc -O3 --offload-arch=gfx950 probe_fp8.hip -o _probe_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void cvt_kernel(const float* x, unsigned char* o, int n) {
  const int i = threadIdx.x;
  if (i < n) o[i] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false) & 0xff);
}

// A, B row-major [32][64] fp8 bytes; lane l supplies row l & 31, k bytes [32 * (l >> 5), +32)
__global__ void mfma_kernel(const unsigned char* A, const unsigned char* B, float* C) {
  const int l = threadIdx.x;
  v8i a, b;
  memcpy(&a, A + (l & 31) * 64 + 32 * (l >> 5), 32);
  memcpy(&b, B + (l & 31) * 64 + 32 * (l >> 5), 32);
  v16f acc = {};
  // operand order as in the bf16 kernels: first operand = N fragment (B rows), second = M fragment
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, acc, 0, 0, 0, 127, 0, 127);
  for (int r = 0; r < 16; ++r) {
    // bf16 32x32 result layout with swapped operands: lane -> m = l & 31, n = (r&3) + 8*(r>>2) + 4*(l>>5)
    const int m = l & 31, n = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[m * 32 + n] = acc[r];
  }
}

__global__ void rate_kernel(float* out, int iters) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
    c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
    c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static float fp8_e4m3fn_to_float(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}

int main() {
  // 1. conversion format
  const float xs[] = {0.f, 0.5f, 1.f, 1.75f, 3.1f, 240.f, 256.f, 448.f, 480.f, 1000.f, -448.f, -1.f, 0.002f, 1e-6f};
  const int n = sizeof(xs) / sizeof(xs[0]);
  float* dx; unsigned char* dout;
  hipMalloc(&dx, sizeof(xs)); hipMalloc(&dout, 64);
  hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dx, dout, n);
  unsigned char ho[64];
  hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt %g -> 0x%02x (as OCP e4m3fn = %g)\n", xs[i], ho[i], fp8_e4m3fn_to_float(ho[i]));
  // 2. layout: small integers encoded as OCP e4m3 (0 -> 0x00, 1 -> 0x38, 2 -> 0x40, 3 -> 0x44)
  const unsigned char enc[4] = {0x00, 0x38, 0x40, 0x44};
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  std::vector<int> Ai(32 * 64), Bi(32 * 64);
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { Ai[i] = rand() % 4; Bi[i] = rand() % 4; A[i] = enc[Ai[i]]; B[i] = enc[Bi[i]]; }
  unsigned char *dA, *dB; float* dC;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  std::vector<float> C(1024);
  hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 32; ++m)
    for (int nn = 0; nn < 32; ++nn) {
      int ref = 0;
      for (int k = 0; k < 64; ++k) ref += Ai[m * 64 + k] * Bi[nn * 64 + k];
      if ((float)ref != C[m * 32 + nn]) { if (bad < 5) printf("mismatch C[%d][%d] = %g ref %d\n", m, nn, C[m * 32 + nn], ref); ++bad; }
    }
  printf("layout check: %d mismatches of 1024\n", bad);
  // 3. issue rate
  float* dr; hipMalloc(&dr, 256 * 1024 * 4 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, dr, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel, dim3(1024), dim3(256), 0, 0, dr, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 1024.0 * 4 /*waves*/ * iters * 4.0 * 2 * 32 * 32 * 64;
  printf("f8f6f4 32x32x64 rate: %.3f ms, %.1f TFLOP/s (bf16 dense peak 2500, fp8 peak 5000)\n", ms, flops / ms / 1e9);
  return 0;
}
