// Standalone GEMM harness (no Python / torch start-up): correctness against an fp32 reference
// kernel on sampled rows + timing of mk_gemm per (shape, layout, kernel configuration).
//
//   hipcc -O2 --offload-arch=gfx950 scripts/gemm_bench.cpp -o scripts/probe/_probe_gemm_bench \
//         -Lmacaw_llm_amd -lmacaw_hip -lhipblaslt -Wl,-rpath,'$ORIGIN/../../macaw_llm_amd'
//   scripts/probe/_probe_gemm_bench [shapes-file] > gpurun_out/gemm_bench.csv
//
// shapes file: one "M N K layout cfgs..." per line (layout 0 = NT fwd, 1 = NN grad-input,
// 3 = TT grad-weight, as mk_prof_report prints them); default = the LLaMA-7B shapes of BASELINE
// cfg 3.  Data: A ~ N(0,1), B ~ 0.02 N(0,1) (weights) like the training step, never zeros
// (cdna_hip_programming.md rule 25).
//
// cfg 100 is NOT a configuration of mk_gemm: it runs the vendor library (hipBLASLt, heuristic algorithm) on the
// same operands in the same process -- a YARDSTICK for "what this box reaches on this data" (rule 10: ceilings
// come from a known-good reference on the same hardware).  The product never links or calls it.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/macaw_hip.h"
#include <hipblaslt/hipblaslt.h>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

typedef __bf16 bf16;

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void fill_normal(bf16* p, long n, unsigned seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const unsigned h1 = hash32((unsigned)i * 2654435761U + seed);
    const unsigned h2 = hash32(h1 ^ 0x9e3779b9U);
    const float u1 = ((h1 >> 8) + 1) * (1.0f / 16777217.0f), u2 = (h2 >> 8) * (1.0f / 16777216.0f);
    p[i] = (bf16)(scale * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2));
  }
}
// reference for sampled output rows: ref[s][n] = sum_k A(m_s, k) * B(n, k) with the operand
// storage given by the layout flags (fp32 accumulation in a fixed order)
__global__ void ref_rows(const bf16* A, const bf16* B, float* ref, const int* rows, int nrows, int N,
                         int K, long lda, long ldb, int a_red, int b_red) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (n >= N || s >= nrows) return;
  const int m = rows[s];
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const float a = (float)(a_red ? A[(long)k * lda + m] : A[(long)m * lda + k]);
    const float b = (float)(b_red ? B[(long)k * ldb + n] : B[(long)n * ldb + k]);
    acc += a * b;
  }
  ref[(long)s * N + n] = acc;
}
__global__ void cmp_rows(const bf16* C, long ldc, const float* ref, const int* rows, int nrows, int N,
                         float* out /* [0] max abs err, [1] max |ref|, [2] #bad */) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (n >= N || s >= nrows) return;
  const float r = ref[(long)s * N + n];
  const float c = (float)C[(long)rows[s] * ldc + n];
  const float err = fabsf(c - r);
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(err));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(fabsf(r)));
  // bf16 output rounding (2^-8 relative) + accumulation-order noise
  if (!(err <= 6e-3f * fabsf(r) + 2e-2f)) {
    const float nb = atomicAdd(out + 2, 1.0f);
    if (nb < 6.f) printf("  mismatch row %d col %d: got %g ref %g\n", rows[s], n, c, r);
  }
}


// ---- vendor yardstick (cfg 100): C[M][N] row-major = op(A) op(B)^T  <=>  column-major C^T (N x M) = B_op A_op
struct LtPlan {
  hipblasLtMatmulDesc_t md{}; hipblasLtMatrixLayout_t la{}, lb{}, lc{}; hipblasLtMatmulHeuristicResult_t res{};
  bool ok = false;
};
#define LT(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "%s:%d hipblaslt status %d\n", __FILE__, __LINE__, (int)s_); return p; } } while (0)
static LtPlan lt_plan(hipblasLtHandle_t h, int M, int N, int K, long lda, long ldb, long ldc, int a_red, int b_red,
                      size_t ws_bytes) {
  LtPlan p;
  LT(hipblasLtMatmulDescCreate(&p.md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  // first operand = our B: stored [N][K] (column-major K x N, op T) or, reduction-major, [K][N] (N x K, op N)
  const int32_t opa = b_red ? HIPBLAS_OP_N : HIPBLAS_OP_T;
  // second operand = our A: stored [M][K] (column-major K x M, op N) or [K][M] (M x K, op T)
  const int32_t opb = a_red ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LT(hipblasLtMatmulDescSetAttribute(p.md, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof opa));
  LT(hipblasLtMatmulDescSetAttribute(p.md, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof opb));
  LT(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, b_red ? N : K, b_red ? K : N, ldb));
  LT(hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, a_red ? M : K, a_red ? K : M, lda));
  LT(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, N, M, ldc));
  hipblasLtMatmulPreference_t pref;
  LT(hipblasLtMatmulPreferenceCreate(&pref));
  uint64_t wsb = ws_bytes;
  LT(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof wsb));
  int n = 0;
  LT(hipblasLtMatmulAlgoGetHeuristic(h, p.md, p.la, p.lb, p.lc, p.lc, pref, 1, &p.res, &n));
  hipblasLtMatmulPreferenceDestroy(pref);
  p.ok = n > 0;
  return p;
}
static int lt_run(hipblasLtHandle_t h, const LtPlan& p, const void* A, const void* B, void* C, void* ws, size_t wsb,
                  hipStream_t st) {
  const float one = 1.f, zero = 0.f;
  return (int)hipblasLtMatmul(h, p.md, &one, B, p.la, A, p.lb, &zero, C, p.lc, C, p.lc, &p.res.algo, ws, wsb, st);
}

struct Shape { int M, N, K, layout; std::vector<int> cfgs; };

int main(int argc, char** argv) {
  std::vector<Shape> shapes;
  if (argc > 1) {
    FILE* f = fopen(argv[1], "r");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
    char line[512];
    while (fgets(line, sizeof line, f)) {
      if (line[0] == '#' || strlen(line) < 5) continue;
      Shape s{};
      int off = 0, n = 0;
      if (sscanf(line, "%d %d %d %d%n", &s.M, &s.N, &s.K, &s.layout, &off) < 4) continue;
      int c;
      while (sscanf(line + off, "%d%n", &c, &n) == 1) { s.cfgs.push_back(c); off += n; }
      if (s.cfgs.empty()) s.cfgs = {5, 11};
      shapes.push_back(s);
    }
    fclose(f);
  } else {
    const int M = 4608;
    for (int layout : {0, 1, 3}) {
      const int NK[4][2] = {{12288, 4096}, {4096, 4096}, {22016, 4096}, {4096, 11008}};
      for (auto& nk : NK) {
        Shape s{};
        if (layout == 0) { s.M = M; s.N = nk[0]; s.K = nk[1]; }        // y = x W^T
        else if (layout == 1) { s.M = M; s.N = nk[1]; s.K = nk[0]; }   // dx = dy W
        else { s.M = nk[0]; s.N = nk[1]; s.K = M; }                    // dW = dy^T x
        s.layout = layout;
        s.cfgs = {5, 11};
        shapes.push_back(s);
      }
    }
  }
  const int iters = getenv("GB_ITERS") ? atoi(getenv("GB_ITERS")) : 10;
  const int rounds = getenv("GB_ROUNDS") ? atoi(getenv("GB_ROUNDS")) : 2;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const long WS = 72L << 20;
  void* ws;
  CK(hipMalloc(&ws, WS));
  CK(hipMemset(ws, 0, 4096));
  float* stats;
  CK(hipMalloc(&stats, 16));
  hipblasLtHandle_t lth = nullptr;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (getenv("GB_WRITE_BW")) {   // raw store bandwidth of the chip for epilogue-sized buffers
    for (long mb : {32L, 64L, 256L, 1024L}) {
      void* p;
      CK(hipMalloc(&p, mb << 20));
      for (int kind = 0; kind < 2; ++kind) {
        for (int i = 0; i < 3; ++i) {
          if (kind == 0) CK(hipMemsetAsync(p, 1, mb << 20, st));
          else hipLaunchKernelGGL(fill_normal, dim3(2048), dim3(256), 0, st, (bf16*)p, (mb << 20) / 2, 1u, 1.0f);
        }
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) {
          if (kind == 0) CK(hipMemsetAsync(p, 1, mb << 20, st));
          else hipLaunchKernelGGL(fill_normal, dim3(2048), dim3(256), 0, st, (bf16*)p, (mb << 20) / 2, 1u, 1.0f);
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("write %s %ld MiB: %.2f us, %.2f TB/s\n", kind ? "fill_kernel(2B stores)" : "memset", mb, ms * 100,
               (double)(mb << 20) / (ms / 10 * 1e-3) / 1e12);
      }
      CK(hipFree(p));
    }
  }
  printf("M,N,K,layout,cfg,ms,tflops,max_err,max_ref,bad\n");
  for (auto& s : shapes) {
    const int a_red = (s.layout >> 1) & 1, b_red = s.layout & 1;
    // storage: A is [M][K] (lda = K) or, reduction-major, [K][M] (lda = M); same for B
    // (reduction-major pitches padded to 64 elements, as the engine's pitched buffers are)
    // K-major operands with K % 64 != 0 are pitched to 64 with ZERO pad columns (MK_GEMM_*_KPAD_ZERO)
    const long kpad = (s.K + 63) / 64 * 64;
    const long lda = a_red ? (s.M + 63) / 64 * 64 : kpad, ldb = b_red ? (s.N + 63) / 64 * 64 : kpad;
    const long ldc = (s.N + 63) / 64 * 64;
    const long na = a_red ? lda * s.K : (long)s.M * lda, nb = b_red ? ldb * s.K : (long)s.N * ldb;
    const long nc = (long)s.M * ldc;
    // GB_COLD=1: every timed launch reads a DIFFERENT copy of its operands (enough copies to exceed
    // the 256 MB Infinity Cache several times), as inside a training step where weights and saved
    // activations come from HBM; the default re-reads the same operands (cache-warm)
    const bool cold = getenv("GB_COLD") != nullptr;
    const int ncopy = cold ? (int)std::max<long>(2, (1536L << 20) / ((na + nb) * 2) + 1) : 1;
    bf16 *A, *B, *C;
    CK(hipMalloc(&A, na * 2 * ncopy));
    CK(hipMalloc(&B, nb * 2 * ncopy));
    CK(hipMalloc(&C, nc * 2));
    // forward: A = activations, B = weights; grad-input: A = dy, B = W; grad-weight: both activations
    hipLaunchKernelGGL(fill_normal, dim3(2048), dim3(256), 0, st, A, na, 0x1234u + s.M, 1.0f);
    hipLaunchKernelGGL(fill_normal, dim3(2048), dim3(256), 0, st, B, nb, 0xbeefu + s.N, s.layout == 3 ? 1.0f : 0.02f);
    for (int c = 1; c < ncopy; ++c) {
      CK(hipMemcpyAsync(A + c * na, A, na * 2, hipMemcpyDeviceToDevice, st));
      CK(hipMemcpyAsync(B + c * nb, B, nb * 2, hipMemcpyDeviceToDevice, st));
    }
    if (kpad != s.K) {
      if (!a_red) CK(hipMemset2DAsync(A + s.K, lda * 2, 0, (kpad - s.K) * 2, s.M, st));
      if (!b_red) CK(hipMemset2DAsync(B + s.K, ldb * 2, 0, (kpad - s.K) * 2, s.N, st));
    }
    // sampled rows: edges of the first / last tiles and a spread in between
    std::vector<int> rows;
    for (int r : {0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 257, 511}) if (r < s.M) rows.push_back(r);
    for (int i = 0; i < 20; ++i) rows.push_back((int)(((long)s.M * (2 * i + 1)) / 41));
    for (int r : {s.M - 1, s.M - 2, s.M - 33, s.M - 129, s.M - 257}) if (r >= 0) rows.push_back(r);
    int* drows;
    float* ref;
    CK(hipMalloc(&drows, rows.size() * 4));
    CK(hipMemcpyAsync(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, st));
    CK(hipMalloc(&ref, rows.size() * (long)s.N * 4));
    hipLaunchKernelGGL(ref_rows, dim3((s.N + 255) / 256, rows.size()), dim3(256), 0, st, A, B, ref, drows,
                       (int)rows.size(), s.N, s.K, lda, ldb, a_red, b_red);
    mk_gemm_desc d{};
    d.A = A; d.B = B; d.C = C;
    d.M = s.M; d.N = s.N; d.K = s.K;
    d.lda = lda; d.ldb = ldb; d.ldc = ldc;
    d.a_red_major = a_red; d.b_red_major = b_red;
    d.nb1 = d.nb2 = 1;
    d.alpha = 1.0f;
    d.dtype = MK_BF16;
    d.ws = ws; d.ws_bytes = WS;
    if (kpad != s.K) d.flags = (a_red ? 0 : MK_GEMM_A_KPAD_ZERO) | (b_red ? 0 : MK_GEMM_B_KPAD_ZERO);
    LtPlan ltp;
    for (int c : s.cfgs)
      if (c == 100 && !ltp.ok) {
        if (!lth && hipblasLtCreate(&lth) != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "hipblasLtCreate failed\n"); return 4; }
        ltp = lt_plan(lth, s.M, s.N, s.K, lda, ldb, ldc, a_red, b_red, WS);
        if (!ltp.ok) fprintf(stderr, "hipblaslt: no algorithm for %d %d %d layout %d\n", s.M, s.N, s.K, s.layout);
      }
    std::vector<double> best(s.cfgs.size(), 1e30);
    std::vector<float> err(s.cfgs.size() * 3, 0.f);
    for (int rd = 0; rd < rounds; ++rd) {
      for (size_t ci = 0; ci < s.cfgs.size(); ++ci) {   // interleaved A/B rounds (rule 24)
        const bool vendor = s.cfgs[ci] == 100;
        if (vendor && !ltp.ok) { best[ci] = 0; continue; }
        if (!vendor) mk_gemm_set_cfg(s.cfgs[ci]);
        if (rd == 0) {
          CK(hipMemsetAsync(C, 0xff, nc * 2, st));   // poison: untouched outputs show as NaN
          CK(hipMemsetAsync(stats, 0, 16, st));
          int rc = vendor ? lt_run(lth, ltp, A, B, C, ws, WS, st) : mk_gemm(&d, st);
          if (rc) { fprintf(stderr, "%s rc %d\n", vendor ? "hipblasLtMatmul" : "mk_gemm", rc); return 3; }
          hipLaunchKernelGGL(cmp_rows, dim3((s.N + 255) / 256, rows.size()), dim3(256), 0, st, C, ldc, ref,
                             drows, (int)rows.size(), s.N, stats);
          CK(hipMemcpyAsync(&err[ci * 3], stats, 12, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
        }
        auto launch = [&](int i) {
          mk_gemm_desc dc = d;
          dc.A = A + (long)(i % ncopy) * na;
          dc.B = B + (long)(i % ncopy) * nb;
          if (vendor) lt_run(lth, ltp, dc.A, dc.B, C, ws, WS, st);
          else mk_gemm(&dc, st);
        };
        for (int i = 0; i < 2; ++i) launch(i + 7);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best[ci] = std::min(best[ci], (double)ms / iters);
      }
    }
    for (size_t ci = 0; ci < s.cfgs.size(); ++ci) {
      const double tf = best[ci] > 0 ? 2.0 * s.M * s.N * s.K / (best[ci] * 1e-3) / 1e12 : 0.0;
      printf("%d,%d,%d,%d,%d,%.4f,%.1f,%.4g,%.4g,%.0f\n", s.M, s.N, s.K, s.layout, s.cfgs[ci], best[ci], tf,
             err[ci * 3], err[ci * 3 + 1], err[ci * 3 + 2]);
      fflush(stdout);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(drows)); CK(hipFree(ref));
  }
  mk_gemm_set_cfg(-1);
  return 0;
}
