// Stand-alone timing of mk_flash_attn_fwd (no Python / torch): used with the TIMING-ONLY variant libraries of
// scripts/probe/build_f8_variants.sh (LD_LIBRARY_PATH=scripts/probe/_probe_f8_<n> scripts/probe/_probe_attn_fwd).
//   hipcc -O2 --offload-arch=gfx950 scripts/probe/attn_fwd_probe.cpp -o scripts/probe/_probe_attn_fwd -Lmacaw_llm_amd -lmacaw_hip
// usage: _probe_attn_fwd [B H S hd causal]   (default 4 32 2048 128 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../../include/macaw_hip.h"
int main(int argc, char** argv) {
  int B = 4, H = 32, S = 2048, hd = 128, causal = 1;
  if (argc >= 6) { B = atoi(argv[1]); H = atoi(argv[2]); S = atoi(argv[3]); hd = atoi(argv[4]); causal = atoi(argv[5]); }
  const long D = (long)H * hd, n = (long)B * S * D;
  std::vector<uint16_t> h(n);
  uint32_t x = 12345u;
  for (long i = 0; i < n; ++i) {      // bf16 of roughly N(0, 0.5): sum of four uniforms
    float f = 0.f;
    for (int k = 0; k < 4; ++k) { x = x * 1664525u + 1013904223u; f += (float)(x >> 8) / 16777216.f - 0.5f; }
    f *= 0.87f;
    uint32_t u; __builtin_memcpy(&u, &f, 4);
    h[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  }
  void *q, *k, *v, *o; float* lse;
  hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2); hipMalloc(&lse, (long)B * H * S * 4);
  hipMemcpy(q, h.data(), n * 2, hipMemcpyHostToDevice);
  for (long i = 0; i < n; ++i) h[i] = (uint16_t)(h[i] ^ (uint16_t)((i * 2654435761u) >> 31 << 15));
  hipMemcpy(k, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(v, h.data() + 0, n * 2, hipMemcpyHostToDevice);
  auto run = [&] {
    return mk_flash_attn_fwd(q, k, v, o, lse, nullptr, B, H, S, S, hd, D, (long)S * D, D, (long)S * D, D, (long)S * D, D,
                             (long)S * D, 1.0f / sqrtf((float)hd), causal, MK_BF16, nullptr);
  };
  for (int i = 0; i < 3; ++i) if (int rc = run()) { printf("rc %d\n", rc); return 1; }
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 10; ++i) run();
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pairs = causal ? ((double)S * S - 0.5 * S * (S - 1)) : (double)S * S;
    printf("B %d H %d S %d hd %d causal %d: %.1f us  %.1f TFLOP/s\n", B, H, S, hd, causal, ms * 100.0,
           4.0 * pairs * hd * B * H / (ms * 1e-4) / 1e12);
  }
  return 0;
}
