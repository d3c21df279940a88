// Round 6 probe: the multi-tensor AdamW stream (adamw_multi_kernel, 35 ms of the 213 ms cfg-3 step at 5.65 TB/s = 0.90 of a
// copy) in a stand-alone harness, to try launch-geometry / pipelining variants without touching the shipped kernel:
// 14 B read (bf16 g, fp32 master, m, v) + 14 B written (master, m, v, bf16 w) per element, the shipped arithmetic.
//   CHUNK   elements per workgroup slice        GROUPS  4-element groups per lane and trip (all loads before the first use)
//   THREADS workgroup size                      NTG     non-temporal loads of the gradient (read once)
//   NTW     non-temporal stores of the 16-bit weight copy
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/adamw_stream.hip -o scripts/probe/_probe_adamw_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../macaw_llm_amd/csrc/softmax.hip"      // the SHIPPED kernel (mk_adamw_multi) in the same harness
#undef MK_ST
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ inline float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ inline u16 f2bf(float f) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(f));
  return (u16)r;
}

template <int CHUNK, int GROUPS, int THREADS, bool NTG, bool NTW>
__global__ __launch_bounds__(THREADS) void adamw_kernel(u16* __restrict__ w16, float* __restrict__ master, float* __restrict__ m,
                                                        float* __restrict__ v, const u16* __restrict__ g, long n, float lr,
                                                        float b1, float b2, float eps, float wd, float ib1, float ib2) {
  const long base = (long)blockIdx.x * CHUNK;
  const long end = base + CHUNK < n ? base + CHUNK : n;
  for (long i0 = base + (long)threadIdx.x * 4; i0 < end; i0 += (long)THREADS * 4 * GROUPS) {
    uint2 gr[GROUPS];
    float4 w4[GROUPS], m4[GROUPS], v4[GROUPS];
    long idx[GROUPS];
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) {
      const long i = i0 + (long)u * THREADS * 4;
      idx[u] = i < end ? i : i0;
    }
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) {
      if (NTG) { const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(g + idx[u])); gr[u] = make_uint2(t[0], t[1]); }
      else gr[u] = *reinterpret_cast<const uint2*>(g + idx[u]);
    }
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) w4[u] = *reinterpret_cast<const float4*>(master + idx[u]);
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) m4[u] = *reinterpret_cast<const float4*>(m + idx[u]);
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) v4[u] = *reinterpret_cast<const float4*>(v + idx[u]);
#pragma unroll
    for (int u = 0; u < GROUPS; ++u) {
      if (u > 0 && idx[u] == i0) continue;
      const float gk[4] = {bf2f((u16)(gr[u].x & 0xffff)), bf2f((u16)(gr[u].x >> 16)), bf2f((u16)(gr[u].y & 0xffff)), bf2f((u16)(gr[u].y >> 16))};
      float ww[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w}, mm[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w},
            vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mm[k] = b1 * mm[k] + (1.f - b1) * gk[k];
        vv[k] = b2 * vv[k] + (1.f - b2) * gk[k] * gk[k];
        float wk = ww[k];
        wk -= lr * wd * wk;
        wk -= lr * (mm[k] * ib1) / (sqrtf(vv[k] * ib2) + eps);
        ww[k] = wk;
      }
      *reinterpret_cast<float4*>(master + idx[u]) = make_float4(ww[0], ww[1], ww[2], ww[3]);
      *reinterpret_cast<float4*>(m + idx[u]) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *reinterpret_cast<float4*>(v + idx[u]) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      const uint2 o = make_uint2((unsigned)f2bf(ww[0]) | ((unsigned)f2bf(ww[1]) << 16), (unsigned)f2bf(ww[2]) | ((unsigned)f2bf(ww[3]) << 16));
      if (NTW) { u32x2 t; t[0] = o.x; t[1] = o.y; __builtin_nontemporal_store(t, reinterpret_cast<u32x2*>(w16 + idx[u])); }
      else *reinterpret_cast<uint2*>(w16 + idx[u]) = o;
    }
  }
}

template <int CHUNK, int GROUPS, int THREADS, bool NTG, bool NTW>
void run(const char* tag, u16* w, float* a, float* b, float* c, const u16* g, long n) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  const unsigned grid = (unsigned)((n + CHUNK - 1) / CHUNK);
  for (int r = 0; r < 4; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((adamw_kernel<CHUNK, GROUPS, THREADS, NTG, NTW>), dim3(grid), dim3(THREADS), 0, 0, w, a, b, c, g, n, 1e-4f, 0.9f,
                       0.999f, 1e-8f, 0.01f, 1.f / 0.1f, 1.f / 0.001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  printf("%-44s %8.3f ms  %.3f TB/s\n", tag, best, 28.0 * n / (best * 1e-3) / 1e12);
}

int main() {
  const long n = 1L << 30;       // 1 Gi elements: 30 GB of traffic per pass
  u16 *w, *g; float *a, *b, *c;
  CK(hipMalloc(&w, n * 2)); CK(hipMalloc(&g, n * 2)); CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
  CK(hipMemset(w, 0, n * 2)); CK(hipMemset(g, 0x3c, n * 2)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(c, 0x3c, n * 4));
  run<32768, 2, 256, false, false>("chunk 32768 groups 2 threads 256 (shipped)", w, a, b, c, g, n);
  run<32768, 4, 256, false, false>("chunk 32768 groups 4 threads 256", w, a, b, c, g, n);
  run<32768, 1, 256, false, false>("chunk 32768 groups 1 threads 256", w, a, b, c, g, n);
  run<131072, 2, 256, false, false>("chunk 131072 groups 2 threads 256", w, a, b, c, g, n);
  run<8192, 2, 256, false, false>("chunk 8192 groups 2 threads 256", w, a, b, c, g, n);
  run<65536, 2, 512, false, false>("chunk 65536 groups 2 threads 512", w, a, b, c, g, n);
  run<131072, 2, 1024, false, false>("chunk 131072 groups 2 threads 1024", w, a, b, c, g, n);
  run<32768, 2, 256, true, false>("shipped + non-temporal gradient loads", w, a, b, c, g, n);
  run<32768, 2, 256, false, true>("shipped + non-temporal 16-bit stores", w, a, b, c, g, n);
  run<32768, 2, 256, true, true>("shipped + both", w, a, b, c, g, n);
  run<32768, 8, 256, false, false>("chunk 32768 groups 8 threads 256", w, a, b, c, g, n);
  run<8192, 2, 256, true, true>("chunk 8192 + both non-temporal", w, a, b, c, g, n);
  run<16384, 2, 256, true, true>("chunk 16384 + both non-temporal", w, a, b, c, g, n);
  run<4096, 2, 256, true, true>("chunk 4096 + both non-temporal", w, a, b, c, g, n);
  run<8192, 1, 256, true, true>("chunk 8192 groups 1 + both non-temporal", w, a, b, c, g, n);
  run<32768, 2, 256, false, false>("chunk 32768 groups 2 threads 256 (shipped, again)", w, a, b, c, g, n);
  run<2048, 2, 256, true, true>("chunk 2048 + both non-temporal", w, a, b, c, g, n);
  run<1024, 1, 256, true, true>("chunk 1024 groups 1 + both non-temporal", w, a, b, c, g, n);
  run<2048, 1, 256, true, true>("chunk 2048 groups 1 + both non-temporal", w, a, b, c, g, n);
  run<4096, 1, 256, true, true>("chunk 4096 groups 1 + both non-temporal", w, a, b, c, g, n);
  run<4096, 2, 256, false, false>("chunk 4096 groups 2, temporal", w, a, b, c, g, n);
  run<2048, 2, 256, false, false>("chunk 2048 groups 2, temporal", w, a, b, c, g, n);
  run<4096, 4, 256, true, true>("chunk 4096 groups 4 + both non-temporal", w, a, b, c, g, n);
  {   // the shipped kernel on the same buffers: one item, then the same bytes cut into 311 items (as the 7B step has)
    struct Item { void* param; float* master; float* m; float* v; const void* grad; long n; };
    for (int n_items : {1, 311}) {
      std::vector<Item> items(n_items);
      std::vector<long> cs(n_items);
      const long CH = mk_adamw_chunk();
      const long per = (n / n_items) / 32768 * 32768;
      long chunks = 0;
      for (int i = 0; i < n_items; ++i) {
        const long off = (long)i * per, cnt = i == n_items - 1 ? n - off : per;
        items[i] = {w + off, a + off, b + off, c + off, g + off, cnt};
        cs[i] = chunks;
        chunks += (cnt + CH - 1) / CH;
      }
      Item* di; long* dc;
      CK(hipMalloc(&di, sizeof(Item) * n_items)); CK(hipMalloc(&dc, 8 * n_items));
      CK(hipMemcpy(di, items.data(), sizeof(Item) * n_items, hipMemcpyHostToDevice));
      CK(hipMemcpy(dc, cs.data(), 8 * n_items, hipMemcpyHostToDevice));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0));
        const int rc = mk_adamw_multi(di, (const int64_t*)dc, n_items, chunks, 1e-4f, 0.9f, 0.999f, 1e-8f, 0.01f, 10, 1.f, MK_BF16, nullptr);
        if (rc) { printf("mk_adamw_multi rc %d\n", rc); return 1; }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
      }
      printf("SHIPPED mk_adamw_multi (slices of %ld), %3d items  %8.3f ms  %.3f TB/s\n", CH, n_items, best, 28.0 * n / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
