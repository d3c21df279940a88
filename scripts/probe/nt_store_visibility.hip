// Round 6 probe: is a NON-TEMPORAL store of a kernel visible to an asynchronous device-to-host copy that is ordered behind the kernel
// on the SAME stream, while another stream keeps the device busy?  (profiles/r06_dw_side_stream.txt "World 2": with hinted stores in
// the per-slice AdamW kernel, an all-gather staged through the copy engine right behind it captured a stale ZeRO-1 slice in 3 of 10
// full-suite runs; without them 0 of 5.)  Stream A: fill kernel (value = iteration; plain or non-temporal stores, 8 bytes per lane as
// adam_store4 writes bf16) -> hipMemcpyAsync to pinned host memory -> synchronize A -> every element must read `iteration`.  Stream B:
// a long-running kernel hammering another buffer the whole time.  Counts stale elements per mode.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/nt_store_visibility.hip -o scripts/probe/_probe_nt_store_visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(unsigned* buf, long n2, unsigned value) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) {
    u32x2 v; v[0] = value; v[1] = value;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(buf) + i);
    else reinterpret_cast<u32x2*>(buf)[i] = v;
  }
}
__global__ __launch_bounds__(256) void burn_kernel(float* p, long n, int rounds) {
  for (int r = 0; r < rounds; ++r)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.0f;
}

// `c` = the stream the copy runs on: `a` itself, or a third stream ordered behind the kernel by an EVENT only (how a gloo
// collective stages a CUDA tensor: record on the caller's stream, wait on its own)
template <bool NT>
long run(hipStream_t a, hipStream_t b, hipStream_t c, unsigned* dev, unsigned* host, float* burn, long n, int iters, bool busy) {
  long stale = 0;
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int it = 1; it <= iters; ++it) {
    if (busy) hipLaunchKernelGGL(burn_kernel, dim3(2048), dim3(256), 0, b, burn, 64L << 20, 1);
    const int grid = (int)((n / 2 + 255) / 256) < 1024 ? (int)((n / 2 + 255) / 256) : 1024;
    hipLaunchKernelGGL((fill_kernel<NT>), dim3(grid), dim3(256), 0, a, dev, n / 2, (unsigned)it);
    if (c != a) { CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(c, ev, 0)); }
    CK(hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, c));
    CK(hipStreamSynchronize(c));
    for (long i = 0; i < n; ++i) stale += host[i] != (unsigned)it;
  }
  CK(hipDeviceSynchronize());
  return stale;
}

int main() {
  const long n = 8L << 20;          // 32 MiB per copy (a ZeRO-1 slice of the micro model is far smaller; a real one is of this order)
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  unsigned *dev, *host; float* burn;
  CK(hipMalloc(&dev, n * 4)); CK(hipHostMalloc(&host, n * 4)); CK(hipMalloc(&burn, (64L << 20) * 4));
  CK(hipMemset(dev, 0, n * 4)); CK(hipMemset(burn, 0, (64L << 20) * 4));
  hipStream_t c; CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
  printf("elements,copy_stream,mode,busy_second_stream,iterations,stale_elements\n");
  for (long m : {4096L, 262144L, n}) {
    const int iters = m == n ? 200 : 3000;
    for (int third = 0; third < 2; ++third)
      for (int busy = 0; busy < 2; ++busy) {
        hipStream_t cs = third ? c : a;
        printf("%ld,%s,plain,%d,%d,%ld\n", m, third ? "third(event)" : "same", busy, iters, run<false>(a, b, cs, dev, host, burn, m, iters, busy));
        printf("%ld,%s,nontemporal,%d,%d,%ld\n", m, third ? "third(event)" : "same", busy, iters, run<true>(a, b, cs, dev, host, burn, m, iters, busy));
      }
  }
  return 0;
}
