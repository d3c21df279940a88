// GEMM kernels for gfx950 (MI355X).
//
//   bf16 path : 128x128x64 block tile, 4 waves (2x2), each wave 64x64 through
//               v_mfma_f32_32x32x16_bf16 (2x2 fragments, 64 accumulator VGPRs).
//               Operand tiles are staged global -> registers -> LDS (double
//               buffered, one barrier per K-tile; the next tile's global loads
//               are issued before the MFMA block so HBM latency hides under
//               compute).  Two LDS images, chosen per operand:
//                 K-major  (reduction index contiguous, e.g. x[M,K], W[N,K]):
//                   [128 rows][64 k] with the 16-B chunk index XOR-swizzled by
//                   (row>>1)&7 -> conflict-free ds_read_b128 fragment reads.
//                 red-major (operand stored [K][rows], e.g. W in dx = dy.W, and
//                   both operands of dW = dy^T.x): [64 k][128 rows] with the
//                   chunk index XOR 4*(k&3); fragments come out of LDS through
//                   ds_read_b64_tr_b16 (hardware transpose read), so no
//                   separate transpose pass over HBM is ever needed.
//               MFMA operand order is (N-fragment, M-fragment) so each lane ends
//               up with 4 consecutive n for one m: 8-byte row-major C stores.
//   f32 path  : exact-f32 v_mfma_f32_16x16x4_f32, 64x64x16 tile; parity mode
//               (fp32 end-to-end vs the fp32 oracle) and on-device cross-check.
//
// Replaces: nn.Linear / torch.matmul call sites listed in include/macaw_hip.h.
#include "common.h"
#include "../../include/macaw_hip.h"
#include "gemm_common.h"
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

namespace {
using namespace mkg;

// ------------------------------------------------------------- 16-bit tiles --
constexpr int BM = 128, BN = 128, BK = 64;
#ifndef MK_GEMM_DEFAULT_CFG
#define MK_GEMM_DEFAULT_CFG 5
#endif
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
}  // namespace

// The 16-bit kernels are written once over an element type e16 and instantiated for bf16 and for f16
// (the reference's fp16 checkpoints / `--fp16 True`, train.sh:36): same tiles and schedules, the MFMA
// opcode and the conversions are the only differences (common.h E16<>).
#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#include "gemm_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define gemm_bf16_kernel gemm_f16_kernel
#define gemm_bf16_v2_kernel gemm_f16_v2_kernel
#define gemm_fp8_v2_kernel gemm_fp8_v2_kernel_unused
#include "gemm_impl.inc"
#undef gemm_bf16_kernel
#undef gemm_bf16_v2_kernel
#undef gemm_fp8_v2_kernel
#undef MK_E16_T
#undef MK_E16_NS

namespace {
using namespace mkg;

// ------------------------------------------------------------------- f32 --
// 64x64x16 tile, 4 waves (2x2) of 32x32, v_mfma_f32_16x16x4_f32 (exact f32).
constexpr int FBM = 64, FBN = 64, FBK = 16, FPAD = 4;

template <bool RED_MAJOR>
MK_DEV void f32_tile_load(const float* base, long ld, int row0, int k0, int R, int K,
                          float (*lds)[FBM + FPAD]) {
  const int t = threadIdx.x;
  if constexpr (!RED_MAJOR) {
    const int row = t >> 2, kq = (t & 3) * 4;
    const int gr = row0 + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gk = k0 + kq + j;
      lds[kq + j][row] = (gr < R && gk < K) ? base[(long)gr * ld + gk] : 0.f;
    }
  } else {
    const int kr = t >> 4, mq = (t & 15) * 4;
    const int gk = k0 + kr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = row0 + mq + j;
      lds[kr][mq + j] = (gr < R && gk < K) ? base[(long)gk * ld + gr] : 0.f;
    }
  }
}

template <bool A_RED, bool B_RED>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[FBK][FBM + FPAD];
  __shared__ float Bs[FBK][FBN + FPAD];
  int tm, tn;
  tile_coords(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
  const int z = blockIdx.z, z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const float* A = reinterpret_cast<const float*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const float* B = reinterpret_cast<const float*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  float* C = reinterpret_cast<float*>(g.C) + z1 * g.sC1 + z2 * g.sC2;
  const float* Rp = g.R ? reinterpret_cast<const float*>(g.R) + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int m0 = tm * FBM, n0 = tn * FBN;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int wm0 = (w >> 1) * 32, wn0 = (w & 1) * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < g.K; k0 += FBK) {
    f32_tile_load<A_RED>(A, g.lda, m0, k0, g.M, g.K, As);
    f32_tile_load<B_RED>(B, g.ldb, n0, k0, g.N, g.K, Bs);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kk = ks * 4 + (l >> 4);
      float fm[2], fn[2];
      fm[0] = As[kk][wm0 + (l & 15)];
      fm[1] = As[kk][wm0 + 16 + (l & 15)];
      fn[0] = Bs[kk][wn0 + (l & 15)];
      fn[1] = Bs[kk][wn0 + 16 + (l & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fn[j], fm[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i = n][j = m]: lane holds m = l&15, n = 4*(l>>4) + reg
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm0 + i * 16 + (l & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn0 + j * 16 + 4 * (l >> 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= g.N) continue;
        float v = g.alpha * acc[i][j][e];
        if (g.bias_mode == 1) v += reinterpret_cast<const float*>(g.bias)[n + e];
        else if (g.bias_mode == 2) v += reinterpret_cast<const float*>(g.bias)[m];
        v = apply_act(v, g.act);
        if (Rp) v += Rp[(long)m * g.ldr + n + e];
        float* cp = C + (long)m * g.ldc + n + e;
        if (g.accumulate) v += *cp;
        *cp = v;
      }
    }
  }
}

// -------------------------------------------------------------- transpose --
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* in, T* out, int rows, int cols,
                                                        long ld_in, long ld_out, long s_in,
                                                        long s_out) {
  __shared__ T tile[64][65];
  in += (long)blockIdx.z * s_in;
  out += (long)blockIdx.z * s_out;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < cols) tile[i][tx] = in[(long)r * ld_in + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < rows && c < cols) out[(long)c * ld_out + r] = tile[tx][i];
  }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace


// ---------------------------------------------------------------- profiler --
// Optional per-launch HIP-event timing on the launch stream, used by bench.py for the `roofline`
// figures (kernel time measured live, same stream as the kernel).  kind 0 = mk_gemm, 1 = fused
// attention forward, 2 = fused attention backward (csrc/attention.hip calls mkp::begin / end), 3 = fp8 GEMM.
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int kind, M, N, K, nb, layout, cfg; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
}  // namespace

namespace mkp {
bool on() { return g_prof_on; }
// returns an index into the record list (or -1 when profiling is off); the start event is recorded
int begin(hipStream_t st, int kind, double flops, int M, int N, int K, int nb, int layout, int cfg) {
  if (!g_prof_on) return -1;
  ProfRec rec{};
  if (!g_prof_pool.empty()) { rec.a = g_prof_pool.back().first; rec.b = g_prof_pool.back().second; g_prof_pool.pop_back(); }
  else { (void)hipEventCreate(&rec.a); (void)hipEventCreate(&rec.b); }
  rec.flops = flops; rec.kind = kind; rec.M = M; rec.N = N; rec.K = K; rec.nb = nb; rec.layout = layout; rec.cfg = cfg;
  (void)hipEventRecord(rec.a, st);
  g_prof.push_back(rec);
  return (int)g_prof.size() - 1;
}
void set_cfg(int idx, int cfg) { if (idx >= 0) g_prof[idx].cfg = cfg; }
void end(int idx, hipStream_t st) { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, st); }
}  // namespace mkp

extern "C" int mk_prof_begin(void) {
  for (auto& r : g_prof) g_prof_pool.emplace_back(r.a, r.b);
  g_prof.clear();
  g_prof_on = true;
  return MK_OK;
}
// Synchronises, sums (elapsed ms, flops, launches) over every launch of `kind` since mk_prof_begin.
extern "C" int mk_prof_sum(int kind, double* total_ms, double* total_flops, int64_t* launches) {
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
  for (auto& r : g_prof) {
    if (r.kind != kind) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return MK_ERR_LAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return MK_ERR_LAUNCH;
    ms += t;
    fl += r.flops;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return MK_OK;
}
// mk_gemm launches (kind 0) since mk_prof_begin; stops recording.
extern "C" int mk_prof_end(double* total_ms, double* total_flops, int64_t* launches) {
  g_prof_on = false;
  return mk_prof_sum(0, total_ms, total_flops, launches);
}

// Per-shape breakdown of the launches since mk_prof_begin, written as CSV
// (kind,M,N,K,batch,layout,cfg,launches,total_ms,tflops).  Call before mk_prof_end.
extern "C" int mk_prof_report(const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return MK_ERR_BAD_ARG;
  struct Agg { int kind, M, N, K, nb, layout, cfg; long n; double ms, fl; };
  std::vector<Agg> aggs;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) { fclose(f); return MK_ERR_LAUNCH; }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.a, r.b);
    bool found = false;
    for (auto& a : aggs)
      if (a.kind == r.kind && a.M == r.M && a.N == r.N && a.K == r.K && a.nb == r.nb && a.layout == r.layout && a.cfg == r.cfg) {
        a.n++; a.ms += t; a.fl += r.flops; found = true; break;
      }
    if (!found) aggs.push_back({r.kind, r.M, r.N, r.K, r.nb, r.layout, r.cfg, 1, (double)t, r.flops});
  }
  fprintf(f, "kind,M,N,K,batch,layout,cfg,launches,total_ms,tflops\n");
  for (auto& a : aggs)
    fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%ld,%.4f,%.1f\n", a.kind == 0 ? "gemm" : a.kind == 1 ? "attn_fwd" : a.kind == 2 ? "attn_bwd" : "gemm_fp8",
            a.M, a.N, a.K, a.nb, a.layout, a.cfg, a.n, a.ms, a.ms > 0 ? a.fl / (a.ms * 1e-3) / 1e12 : 0.0);
  fclose(f);
  return MK_OK;
}

namespace { int g_force_cfg = -1; }
namespace {
// kernel for 17 ... 32 token rows: 19 = two 16-token tiles per 16 weight rows (gemm_skinny16_kernel MT = 2: N / 16
// workgroups), 22 = 32 x 32 pipelined (gemm_skinny32p_kernel: N / 32 workgroups, ONE token byte from L2 per weight
// byte instead of two).  Measured cold (scripts/gemm_shapes_decode32.txt, profiles/archive/r03_decode32_cold.csv): 22
// wins where N / 32 still fills the chip (N = 22016: 52 vs 69 us, 32007: 73 vs 91, 12288: 35 vs 37), 19 where it
// does not (N = 4096 = 128 workgroups: 14.5 vs 18.2 us at K = 4096, 33.6 vs 46.2 at K = 11008).
// MK_GEMM_SKINNY32 forces one (read per call: scripts/bench_generate.py switches it in-process).
int skinny32_cfg(int N) {
  const char* e = getenv("MK_GEMM_SKINNY32");
  const int forced = e ? atoi(e) : 0;
  if (forced == 19 || forced == 22) return forced;
  return N >= 8192 ? 22 : 19;
}
}

// tuning / A-B hook (scripts/gemm_bench.cpp, tests): force a kernel configuration for the
// following mk_gemm calls of this process (-1 = automatic choice); same meaning as MK_GEMM_CFG
extern "C" int mk_gemm_set_cfg(int cfg) { g_force_cfg = cfg; return MK_OK; }
// 1 if the kernel configuration is compiled into this library (experiment kernels: build.py MK_EXPERIMENTS)
extern "C" int mk_gemm_has_cfg(int cfg) {
#ifdef MK_WITH_V8
  if (cfg == 14) return 1;
#endif
  return cfg == 0 || cfg == 5 || cfg == 7 || cfg == 11 || cfg == 15;
}
namespace { int g_plan_cus = 0; }
extern "C" int mk_gemm_set_cus(int n) { const int prev = g_plan_cus; g_plan_cus = n > 0 ? n : 0; return prev; }

namespace mkg {
int launch_v7(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool fp8, bool f16);  // gemm_v7.hip
#ifdef MK_WITH_V8   // experiment builds only (build.py MK_EXPERIMENTS=1)
int launch_v8(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool f16);            // gemm_v8.hip
#endif
int launch_v9(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool f16);            // gemm_v9.hip
int v9_mfma16_layouts();                                                                                        // gemm_v9.hip
}

// y[M <= 16, N] = prologue(x) W^T (+ residual): the linear layers of one decode position per sample
// with the normalisation / activation that precedes them folded into the weight-streaming kernel
// (one launch instead of two per linear; the hipGraph of a decode step has 5 kernels per layer).
extern "C" int mk_decode_linear(const void* x, int64_t ldx, const void* W, int64_t ldw, void* y,
                                int64_t ldy, const void* residual, int64_t ldr, int32_t M, int32_t N,
                                int32_t K, int32_t prologue, const void* norm_w, float eps,
                                int32_t dtype, void* stream) {
  if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0) return MK_ERR_BAD_ARG;
  if (prologue < 0 || prologue > 2 || (prologue == 1 && !norm_w)) return MK_ERR_BAD_ARG;
  if ((dtype != MK_BF16 && dtype != MK_F16) || M > (prologue ? 16 : 32) || (K % 64) || (ldx % 8) || (ldw % 8) ||
      !aligned16(x) || !aligned16(W) || (prologue == 1 && !aligned16(norm_w)))
    return MK_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.A = x; g.B = W; g.C = y; g.R = residual;
  g.M = M; g.N = N; g.K = K;
  g.lda = ldx; g.ldb = ldw; g.ldc = ldy; g.ldr = ldr;
  g.alpha = 1.f;
  g.pro_w = norm_w; g.pro_eps = eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool wide = N <= 16 * 256 && K <= 4096;     // as mk_gemm: 16 waves where N / 16 workgroups are few
  const size_t lds = prologue ? (size_t)M * (K + 8) * 2 : 0;   // prepared token rows
  if (lds > 40 * 1024) return MK_ERR_UNSUPPORTED;   // (two workgroups per CU must still fit)
  const int prof = mkp::begin(st, 0, 2.0 * M * N * K, M, N, K, 1, 0, 17 + 10 * prologue);
  if (M > 16) {                     // (no prologue: checked above) same kernels as mk_gemm's skinny path
    if (dtype == MK_F16) e_f16::launch_skinny(g, skinny32_cfg(N), N, st);
    else e_bf16::launch_skinny(g, skinny32_cfg(N), N, st);
  } else if (dtype == MK_F16) e_f16::launch_decode_linear(g, N, prologue, wide, lds, st);
  else e_bf16::launch_decode_linear(g, N, prologue, wide, lds, st);
  mkp::end(prof, st);
  return mk_check_launch();
}
namespace {
// Default kernel choice: a small cost model fitted to scripts/gemm_bench.cpp measurements on
// MI355X (microseconds; round 4 re-fit on COLD operands -- as inside a training step -- from
// profiles/r04_gemm_enc_cold.csv, r04_gemm_kslope.csv).  v7: one 256x256 tile per CU and round,
// (K-tiles x a7 + prologue/epilogue) per round, the last partial round as 128x128 sub-tiles when
// that is at most two sub-rounds.  v2: 128x128 tiles, two workgroups per CU (four with BK = 32 for
// reduction-major x reduction-major), fractional rounds through its K-split tail.  The round-2 fit
// priced a v2 K-tile at 0.92 us (cache-warm operands); cold it is 1.67, and with it every encoder
// shape (K = 512 ... 1024: 8224 x 1024 x 1024 at 485 TFLOP/s on v2, 618 on v7; 8224 x 4096 x 1024
// 641 vs 924; 48000 x 2048 x 512 683 vs 835) was on the wrong kernel.
int pick_cfg(const mk_gemm_desc* d, int nbatch, bool v7_ok, int n_cus) {
  static const bool no_v7 = getenv("MK_GEMM_NO_V7") != nullptr;
  if (!v7_ok || no_v7 || nbatch != 1) return MK_GEMM_DEFAULT_CFG;
  const int layout = (d->a_red_major ? 2 : 0) + (d->b_red_major ? 1 : 0);
  const double nk = (d->K + 63) / 64;
  const long T7 = (long)mk_cdiv(d->M, 256) * mk_cdiv(d->N, 256);
  const long full = T7 / n_cus, R = T7 % n_cus;
  const double a7 = layout == 3 ? 1.53 : layout == 1 ? 1.51 : layout == 2 ? 1.52 : 1.46;
  const double tile7 = nk * a7 + 11.5, sub7 = nk * 0.48 + 7.0;
  double t7 = full * tile7;
  // (a partially filled round of whole tiles runs faster per K-tile: less L2 / power contention)
  if (R > 0) t7 += (4 * R <= 2 * n_cus) ? (double)mk_cdiv((int)(4 * R), n_cus) * sub7
                                         : nk * a7 * (0.55 + 0.45 * R / n_cus) + 11.5;
  const long T2 = (long)mk_cdiv(d->M, 128) * mk_cdiv(d->N, 128);
  const long slots2 = (long)n_cus * (layout == 3 ? 4 : 2);
  const double tile2 = layout == 3 ? nk * 2.18 + 4.0 : nk * (layout == 0 ? 1.67 : 1.45) + 2.0;
  const double a2 = layout == 3 ? 2.18 : layout == 0 ? 1.67 : 1.45;
  const double nk2 = layout == 3 ? 2 * nk : nk;
  double t2;
  if (T2 < slots2) {
    // less than one round of tiles: the K-split tail takes the WHOLE problem (project_audio: K = 122,880;
    // the alignment P V: K = 32,064; 192 x 768 x 36,864): every tile is cut into `sp` K-pieces
    double sp = (double)(slots2 / T2);
    if (sp > nk2 / 2) sp = nk2 / 2;
    if (sp > 64) sp = 64;
    if (sp < 1) sp = 1;
    // a2 is the K-tile time of a workgroup that SHARES its CU with another one; with sp == 1 and T2 barely above the CU
    // count most workgroups have a CU to themselves and run 0.93 ... 1.04 us per K-tile instead of 1.67 (round 6, cold
    // harness, profiles/r06_gemm_cfg2_enc.csv: 4112 x 1024 x 4096 61 us on v2 against 79 on v7 where this model said 109;
    // 4112 x 1024 x 1024 18.7 against 27.4) -- CLIP's out-proj / fc2 at 16 images per GPU (BASELINE cfg 2)
    const double alone = sp > 1 ? 1.0 : 0.57 + 0.43 * std::min(1.0, std::max(0.0, (double)T2 / n_cus - 1.0));
    // fix-up: ONE workgroup per tile reads `sp` fp32 slabs of 64 KiB at a single CU's rate (34 GB/s,
    // profiles/r06_local_overlap_confined.txt): 1.6 us per slab, not 0.65 -- 192 x 768 x 4096 and 192 x 512 x 4096 (grad-input
    // of the modality projections) took 62 / 59 us here against 26 on the 256 x 256 kernel (profiles/r06_gemm_policy_sweep.csv)
    t2 = nk / sp * a2 * alone + 2.0 + (sp > 1 ? 4.0 + 1.6 * sp : 0.0);
  } else {
    t2 = (double)T2 / slots2 * tile2;
    if (const long R2 = T2 % slots2) {   // K-split tail: pieces + the last arriver reading `sp` slabs
      double sp = (double)slots2 / R2;
      if (sp > nk2 / 2) sp = nk2 / 2;
      if (sp > 64) sp = 64;
      t2 += 4.0 + 0.65 * sp;
    }
  }
  if (!(t7 * 0.95 < t2)) return MK_GEMM_DEFAULT_CFG;   // (ties go to v7: the model is pessimistic for it)
  // v9 (gemm_v9.hip: one wave per SIMD, hand-placed K loop, register epilogue): whole tiles, at least one full round,
  // a plain epilogue.  Measured against v7 on cold operands (profiles/r05_gemm_v9_step_cold.csv): reduction-major x
  // reduction-major (grad-weight) +2 ... +8 % on every step shape; inside the cfg-3 step (r05_gemm_v9_instep.txt)
  // grad-weight +3 ... +6 %, grad-input (K-major x reduction-major) +0.5 ... +2.3 %, forward (both K-major) -1.5 %
  // at K <= 12288 (v9's fixed cost per tile is 14 us against 12.5, its K-tile 1.485 us against 1.54).
  // MK_GEMM_V9: 0 = never, 1 = this policy (default), 2 = wherever it is legal.
  static const int v9_mode = [] { const char* e = getenv("MK_GEMM_V9"); return e ? atoi(e) : 1; }();
  // (layouts on the 16 x 16 x 32 loop have an all-in-registers epilogue that also takes the residual)
  const bool m16 = ((mkg::v9_mfma16_layouts() >> layout) & 1) != 0;
  const bool v9_legal = d->dtype != MK_F32 && d->dtype != MK_FP8 && d->M % 256 == 0 && d->N % 256 == 0 && d->K % 64 == 0 &&
                        full >= 1 && !d->scale_a && !d->scale_b &&
                        d->bias_mode == 0 && d->act == 0 && !d->accumulate && (m16 || !d->R);
  // forward (K-major x K-major) on the 16 x 16 x 32 loop, inside the cfg-3 step against v7 (profiles/r06_gemm_v9_mfma16.txt):
  // q|k|v +2.5 %, gate|up +0.7 %, down +3.3 %, but o_proj (288 tiles, K = 4096) -5 %: v9 runs a tail as a SECOND launch
  // behind its whole rounds, which one short round of short tiles does not amortise
  const bool nt16 = m16 && layout == 0 && !(R > 0 && full < 2 && nk < 128);
  if (v9_legal && (v9_mode == 2 || (v9_mode == 1 && (layout == 3 || layout == 1 || nk >= 256 || nt16)))) return 15;
  return 11;
}
}  // namespace

extern "C" int mk_abi_version(void) { return MK_ABI_VERSION; }

extern "C" int mk_gemm(const mk_gemm_desc* d_in, void* stream) {
  const mk_gemm_desc* d = d_in;
  if (!d || !d->A || !d->B || !d->C) return MK_ERR_BAD_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return MK_ERR_BAD_ARG;
  if (d->nb1 < 1 || d->nb2 < 1) return MK_ERR_BAD_ARG;
  if (d->bias_mode && !d->bias) return MK_ERR_BAD_ARG;
  // fp8 (e4m3) operands, bf16 result: both operands K-major, K a multiple of 128, 16-byte aligned
  // rows.  The descriptor is restated in 2-byte units and takes the bf16 data path with the
  // f8f6f4 MFMA (no other kernel handles it: anything that does not fit is an error).
  mk_gemm_desc dd;
  bool fp8 = false;
  if (d->dtype == MK_FP8) {
    if (d->a_red_major || d->b_red_major || (d->K % 128) || (d->lda % 16) || (d->ldb % 16) ||
        (d->sA1 % 16) || (d->sA2 % 16) || (d->sB1 % 16) || (d->sB2 % 16) || !aligned16(d->A) ||
        !aligned16(d->B))
      return MK_ERR_UNSUPPORTED;
    dd = *d;
    dd.K /= 2; dd.lda /= 2; dd.ldb /= 2;
    dd.sA1 /= 2; dd.sA2 /= 2; dd.sB1 /= 2; dd.sB2 /= 2;
    dd.dtype = MK_BF16;
    d = &dd;
    fp8 = true;
  }
  if (d->dtype != MK_F32 && d->dtype != MK_BF16 && d->dtype != MK_F16) return MK_ERR_UNSUPPORTED;
  const bool f16 = d->dtype == MK_F16;      // same kernels, instantiated for _Float16 (e_f16::)
  const bool e16 = d->dtype == MK_BF16 || f16;
  GemmArgs g;
  g.scale_a = d->scale_a; g.scale_b = d->scale_b;
  g.scale_vec = (d->flags & MK_GEMM_SCALE_VEC) ? 1 : 0;
  if (g.scale_vec && (!d->scale_a || !d->scale_b)) return MK_ERR_BAD_ARG;
  if (g.scale_vec && !fp8) return MK_ERR_UNSUPPORTED;   // per-row / per-column scales exist in the e4m3 instantiations only
  g.A = d->A; g.B = d->B; g.C = d->C; g.R = d->R; g.bias = d->bias;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr = d->ldr;
  g.nb2 = d->nb2;
  g.sA1 = d->sA1; g.sA2 = d->sA2; g.sB1 = d->sB1; g.sB2 = d->sB2;
  g.sC1 = d->sC1; g.sC2 = d->sC2; g.sR1 = d->sR1; g.sR2 = d->sR2;
  g.alpha = d->alpha; g.bias_mode = d->bias_mode; g.act = d->act; g.accumulate = d->accumulate;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nbatch = d->nb1 * d->nb2;
  // (kind 3 = fp8 GEMM: priced against the 5 PFLOP/s e4m3 peak by bench.py, kind 0 against 2.5)
  const int prof = mkp::begin(st, fp8 ? 3 : 0, 2.0 * d->M * d->N * d->K * nbatch * (fp8 ? 2 : 1), d->M, d->N,
                              d->K * (fp8 ? 2 : 1), nbatch, d->a_red_major * 2 + d->b_red_major, -1);
  // (measured on generate(): B = 1: 7.5 -> 6.1 ms/token, B = 8: 7.2 -> 6.5; at B = 32 the 32 distinct
  // token rows re-read per workgroup cost more than the tile kernel's wasted rows: 8.6 vs 8.0)
  static const int skinny_max = [] { const char* e = getenv("MK_GEMM_SKINNY_MAX_M"); return e ? atoi(e) : 32; }();
  if (e16 && !fp8 && d->M <= 32 && d->M <= skinny_max && !d->a_red_major && !d->b_red_major && nbatch == 1 &&
      (d->K % 64) == 0 && (d->lda % 8) == 0 && (d->ldb % 8) == 0 && aligned16(d->A) && aligned16(d->B) &&
      !getenv("MK_GEMM_NO_SKINNY")) {
    // cfg 12 = 32 weight rows per workgroup (8 waves), 17 / 18 = 16 rows per workgroup with 8 / 16
    // waves splitting K (measured cold, scripts/gemm_shapes_decode.txt: 4.0 ... 5.7 TB/s against
    // 2.1 ... 3.8; 16 waves where N / 16 workgroups alone would leave a CU with one short wave set)
    int sk = d->M <= 16 ? ((d->N <= 16 * 256 && d->K <= 4096) ? 18 : 17) : skinny32_cfg(d->N);
    if ((g_force_cfg == 12 || g_force_cfg == 19 || g_force_cfg == 22) ||
        ((g_force_cfg == 13 || g_force_cfg == 17 || g_force_cfg == 18) && d->M <= 16)) sk = g_force_cfg;
    mkp::set_cfg(prof, sk);
    if (f16) e_f16::launch_skinny(g, sk, d->N, st);
    else e_bf16::launch_skinny(g, sk, d->N, st);
    mkp::end(prof, st);
    return mk_check_launch();
  }
  if (e16) {
    // kernel configuration: 11 = v7 256x256 quadrant-phase LDS-DMA ring (gemm_v7.hip; big aligned
    // problems), 14 = v8 256x256 with one wave per SIMD (gemm_v8.hip), 5 = v2 issue-lean LDS-DMA 128x128 (aligned operands, K % 64 == 0), 7 = v2 with
    // BK = 32 (reduction-major x reduction-major), 0 = register-staged 128x128 (anything).
    // MK_GEMM_CFG forces one where it is legal.
    static const int env_cfg0 = [] {
      const char* e = getenv("MK_GEMM_CFG");
      return e ? atoi(e) : -1;
    }();
    const int env_cfg = g_force_cfg >= 0 ? g_force_cfg : env_cfg0;
    const auto fits = [&](bool red, long ld, int rows) {
      const long span = red ? (long)d->K * ld * 2 : ((long)rows * ld + d->K) * 2;
      return span < 0x7fffffffL;
    };
    // a K that is not a multiple of the K-tile: rows >= K of a reduction-major operand lie outside
    // its buffer descriptor and load as zeros; a K-major operand must then be zero-padded (flags)
    const long kpad = (d->K + 63) / 64 * 64;
    const bool ktail_a = d->a_red_major || ((d->flags & MK_GEMM_A_KPAD_ZERO) && d->lda >= kpad);
    const bool ktail_b = d->b_red_major || ((d->flags & MK_GEMM_B_KPAD_ZERO) && d->ldb >= kpad);
    const bool v2_ok = aligned16(d->A) && aligned16(d->B) && (d->lda % 8 == 0) && (d->ldb % 8 == 0) &&
                       (d->sA1 % 8 == 0) && (d->sA2 % 8 == 0) && (d->sB1 % 8 == 0) &&
                       (d->sB2 % 8 == 0) &&
                       (d->K % BK == 0 || (ktail_a && ktail_b && d->K > BK)) &&
                       fits(d->a_red_major, d->lda, BM) && fits(d->b_red_major, d->ldb, BN);
    const bool v7_ok = v2_ok && d->K >= 128 && d->M > 128 && d->N > 128 &&
                       fits(d->a_red_major, d->lda, 256) && fits(d->b_red_major, d->ldb, 256);
    static const int dev_cus = [] {
      int dev = 0, cus = 256;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return cus;
    }();
    // (mk_gemm_set_cus: CUs held by other resident kernels -- RCCL channels -- are not planned for)
    const int n_cus = (g_plan_cus > 0 && g_plan_cus < dev_cus) ? g_plan_cus : dev_cus;
    int cfg;
    if (env_cfg >= 0) cfg = env_cfg;
    else cfg = pick_cfg(d, nbatch, v7_ok, n_cus);
    if (fp8 && cfg != 11) cfg = 5;        // fp8 exists on the two LDS-DMA tile kernels only
#ifndef MK_WITH_V8
    if (cfg == 14) cfg = 11;              // gemm_v8 is not part of the shipped library (build.py MK_EXPERIMENTS)
#endif
    if ((cfg == 11 || cfg == 14 || cfg == 15) && !v7_ok) cfg = 5;
    if ((cfg == 14 || cfg == 15) && fp8) cfg = 11;       // (v8 / v9 have no e4m3 instantiation)
    // v9 (hand-placed K loop, gemm_v9.hip) takes whole 256 x 256 x 64 tiles only
    if (cfg == 15 && (d->M % 256 != 0 || d->N % 256 != 0 || d->K % 64 != 0 || d->K < 128)) cfg = 11;
    // the layouts of v9 on the 16 x 16 x 32 loop have a register epilogue for alpha (+ residual) only
    if (cfg == 15 && ((mkg::v9_mfma16_layouts() >> ((d->a_red_major ? 2 : 0) + (d->b_red_major ? 1 : 0))) & 1) &&
        (d->bias_mode != 0 || d->act != 0 || d->accumulate || d->scale_a || d->scale_b))
      cfg = 11;
    if (cfg != 0 && cfg != 5 && cfg != 7 && cfg != 11 && cfg != 14 && cfg != 15) cfg = 5;
    if (cfg >= 5 && !v2_ok) cfg = 0;
    if (fp8 && cfg != 5 && cfg != 11) return MK_ERR_UNSUPPORTED;
    // Measured (profiles/): with BOTH operands reduction-major (dW = dy^T x) the global rows are
    // whole 256-B lines whatever BK is, and BK = 32 (32 KiB LDS -> 4 workgroups per CU) is 17 %
    // faster than BK = 64 on the 128x128 tile; K-major operands would degrade to 64-B segments.
    if (cfg == 5 && d->a_red_major && d->b_red_major && !getenv("MK_GEMM_NO_BK32")) cfg = 7;
    const int bkv = cfg == 7 ? 32 : BK;
    mkp::set_cfg(prof, cfg);
    const bool v8 = cfg == 14;            // 256 x 256 tile, one wave per SIMD (gemm_v8.hip)
    const bool v9 = cfg == 15;            // the same with the K loop placed by hand (gemm_v9.hip)
    const bool t256 = cfg == 11 || v8 || v9;
    const int bm = t256 ? 256 : 128;
    const int bn = t256 ? 256 : BN;
    g.tiles_m = mk_cdiv(d->M, bm);
    g.tiles_n = mk_cdiv(d->N, bn);
    g.a_vec = aligned16(d->A) && (d->lda % 8 == 0) && (d->sA1 % 8 == 0) && (d->sA2 % 8 == 0);
    g.b_vec = aligned16(d->B) && (d->ldb % 8 == 0) && (d->sB1 % 8 == 0) && (d->sB2 % 8 == 0);
    const auto c_aligned = [&](uintptr_t mask, long q) {
      return ((reinterpret_cast<uintptr_t>(d->C) & mask) == 0) && (d->ldc % q == 0) && (d->sC1 % q == 0) &&
             (d->sC2 % q == 0) &&
             (!d->R || (((reinterpret_cast<uintptr_t>(d->R) & mask) == 0) && (d->ldr % q == 0) &&
                        (d->sR1 % q == 0) && (d->sR2 % q == 0)));
    };
    g.c_vec = c_aligned(15, 8) ? 2 : c_aligned(7, 4) ? 1 : 0;
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    g.dp_tiles = g.tiles_m * g.tiles_n;
    g.lin_batch = 0;
    g.split = 1;
    g.kt_per_piece = 0;
    g.ws = nullptr;
    g.counters = nullptr;
    g.ablate = 0;
    {   // A/B switch of gemm_v9's epilogue (MK_V9_LDS_EPI=1: the LDS-transposed form also for plain products)
      static const int v9_lds_epi = getenv("MK_V9_LDS_EPI") ? 1 : 0;
      if (cfg == 15 && v9_lds_epi) g.ablate = 9;
    }
    g.tail8 = 0;
    // resident workgroups per CU of the chosen kernel
    const int slots = n_cus * (t256 ? 1 : (cfg == 7 ? 4 : 2));
    static const bool no_streamk = getenv("MK_GEMM_NO_STREAMK") != nullptr;
    if (t256 && !v8 && !no_streamk) {
      // v7: the last partial round of 256x256 tiles is computed as four 128x128 sub-tiles each
      // (one workgroup per sub-tile, full K, no partial sums) when that takes fewer rounds
      const int T = g.tiles_m * g.tiles_n, R = T % n_cus;
      static const bool no_tail8 = getenv("MK_GEMM_NO_TAIL8") != nullptr;
      if (R > 0 && 8 * R <= n_cus && !d->a_red_major && !no_tail8) {
        // few tail tiles (288 = 256 + 32, 774 = 768 + 6, 1548 = 1536 + 12): 64 x 128 eighths so that
        // the tail runs as ONE short round on up to all CUs (a 32-tile tail as quarters keeps half
        // of the chip idle for a longer round)
        g.dp_tiles = T - R;
        g.tail8 = 1;
        grid.x = g.dp_tiles + 8 * R;
      } else if (R > 0 && 4 * R <= 2 * n_cus) {
        g.dp_tiles = T - R;
        grid.x = g.dp_tiles + 4 * R;
      }
    } else if ((cfg == 5 || cfg == 7) && d->ws && !no_streamk) {
      const int T = g.tiles_m * g.tiles_n * nbatch, nkt = (d->K + bkv - 1) / bkv;
      const int R = T % slots;
      int sp = R > 0 ? slots / R : 1;
      if (sp > nkt / 2) sp = nkt / 2;  // at least two K-tiles per piece
      if (sp > 64) sp = 64;
      const long need = 4096 + (long)R * sp * (64 * 256) * 4;
      if (R > 0 && sp >= 2 && need <= d->ws_bytes) {
        g.dp_tiles = T - R;
        g.split = sp;
        g.kt_per_piece = (nkt + sp - 1) / sp;
        // pieces that would start past the end get nk <= 0 and contribute zeros
        g.counters = reinterpret_cast<int*>(d->ws);
        g.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(d->ws) + 4096);
        if (R * (int)sizeof(int) > 4096) { g.dp_tiles = T; g.split = 1; }
        else {
          grid.x = g.dp_tiles + R * sp;
          if (nbatch > 1) { g.lin_batch = 1; grid.z = 1; }
        }
      }
    }
    g.walkers = g.dp_tiles;
    if (t256 && !v8 && !v9) {
      // more whole tiles than CUs: one WALKING workgroup per planned CU (tile, tile + walkers, ...), which requests
      // the first K-tiles of its next tile before the epilogue of the current one (gemm_v7_impl.inc).  The XCD
      // map needs the stride to be a multiple of 8; batched problems keep one workgroup per tile, and so do
      // tile counts that are not whole rounds (the hardware's dynamic dispatch balances those better).
      static const bool no_walk = getenv("MK_GEMM_NO_WALK") != nullptr;
      if (!no_walk && nbatch == 1 && n_cus % 8 == 0 && g.dp_tiles > n_cus && g.dp_tiles % n_cus == 0) {
        grid.x = n_cus + (grid.x - g.dp_tiles);
        g.walkers = n_cus;
      }
    }
#ifdef MK_WITH_V8
    if (v8) {
      const int rc = mkg::launch_v8(g, d->a_red_major != 0, d->b_red_major != 0, grid, st, f16);
      mkp::end(prof, st);
      return rc;
    }
#endif
    // (the 16 x 16 x 32 layouts' register epilogue moves 8 bytes per lane: C / R must be 8-byte aligned)
    const bool v9_c_ok = !((mkg::v9_mfma16_layouts() >> ((d->a_red_major ? 2 : 0) + (d->b_red_major ? 1 : 0))) & 1) || g.c_vec >= 1;
    if (v9 && g.dp_tiles >= n_cus && v9_c_ok) {
      // whole tiles [0, dp_tiles) on v9, one workgroup each; the spatial tail of the last partial round (planned
      // above exactly as for v7: eighths / quarters of the tiles >= dp_tiles) on v7's sub-tile kernels behind it
      // on the same stream (walkers = 0: every workgroup of that launch is a tail workgroup)
      const int tail_wgs = (int)grid.x - g.walkers;
      dim3 gmain(g.dp_tiles, 1, nbatch);
      // whole rounds of tiles and a register epilogue: one WALKING workgroup per planned CU (tile, tile + n_cus, ...),
      // which requests its next tile's first K-tiles before the epilogue of the current one (gemm_v9_impl.inc)
      static const bool no_walk9 = getenv("MK_GEMM_NO_WALK") != nullptr;
      const bool plain9 = d->bias_mode == 0 && d->act == 0 && !d->R && !d->accumulate && g.c_vec == 2 && !d->scale_a &&
                          !d->scale_b && g.ablate != 9;
      const int lay9 = (d->a_red_major ? 2 : 0) + (d->b_red_major ? 1 : 0);
      const bool regepi9 = plain9 || (((mkg::v9_mfma16_layouts() >> lay9) & 1) && g.c_vec >= 1 && d->bias_mode == 0 && d->act == 0 &&
                                      !d->accumulate && !d->scale_a && !d->scale_b && g.ablate != 9);
      if (!no_walk9 && regepi9 && nbatch == 1 && n_cus % 8 == 0 && g.dp_tiles > n_cus && g.dp_tiles % n_cus == 0)
        gmain.x = n_cus;
      int rc = mkg::launch_v9(g, d->a_red_major != 0, d->b_red_major != 0, gmain, st, f16);
      if (rc == MK_OK && tail_wgs > 0) {
        GemmArgs gt = g;
        gt.walkers = 0;
        rc = mkg::launch_v7(gt, d->a_red_major != 0, d->b_red_major != 0, dim3(tail_wgs, 1, nbatch), st, false, f16);
      }
      mkp::end(prof, st);
      return rc;
    }
    if (t256) {
      // cfg 15 with fewer whole tiles than planned CUs (forced via MK_GEMM_CFG / mk_gemm_set_cfg) runs on v7: the
      // profile must say so
      if (v9) mkp::set_cfg(prof, 11);
      const int rc = mkg::launch_v7(g, d->a_red_major != 0, d->b_red_major != 0, grid, st, fp8, f16);
      mkp::end(prof, st);
      return rc;
    }
    if (f16) (void)e_f16::launch_tile(g, cfg, d->a_red_major != 0, d->b_red_major != 0, false, grid, st);
    else (void)e_bf16::launch_tile(g, cfg, d->a_red_major != 0, d->b_red_major != 0, fp8, grid, st);
  } else {
    g.tiles_m = mk_cdiv(d->M, FBM);
    g.tiles_n = mk_cdiv(d->N, FBN);
    g.a_vec = g.b_vec = g.c_vec = 0;
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch), block(256);
    if (!d->a_red_major && !d->b_red_major)
      MK_LAUNCH((gemm_f32_kernel<false, false>), grid, block, 0, st, g);
    else if (!d->a_red_major && d->b_red_major)
      MK_LAUNCH((gemm_f32_kernel<false, true>), grid, block, 0, st, g);
    else if (d->a_red_major && !d->b_red_major)
      MK_LAUNCH((gemm_f32_kernel<true, false>), grid, block, 0, st, g);
    else
      MK_LAUNCH((gemm_f32_kernel<true, true>), grid, block, 0, st, g);
  }
  mkp::end(prof, st);
  return mk_check_launch();
}

extern "C" int mk_transpose(const void* in, void* out, int32_t rows, int32_t cols, int64_t ld_in,
                            int64_t ld_out, int32_t batch, int64_t s_in, int64_t s_out,
                            int32_t elem_size, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || batch <= 0) return MK_ERR_BAD_ARG;
  dim3 grid(mk_cdiv(cols, 64), mk_cdiv(rows, 64), batch), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (elem_size == 2)
    MK_LAUNCH((transpose_kernel<unsigned short>), grid, block, 0, st,
                       (const unsigned short*)in, (unsigned short*)out, rows, cols, (long)ld_in,
                       (long)ld_out, (long)s_in, (long)s_out);
  else if (elem_size == 4)
    MK_LAUNCH((transpose_kernel<float>), grid, block, 0, st, (const float*)in,
                       (float*)out, rows, cols, (long)ld_in, (long)ld_out, (long)s_in,
                       (long)s_out);
  else
    return MK_ERR_UNSUPPORTED;
  return mk_check_launch();
}
