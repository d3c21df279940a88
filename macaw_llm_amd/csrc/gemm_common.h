// Shared pieces of the GEMM kernels (gemm.hip, gemm_v7.hip): launch arguments, tile order,
// activation and the LDS-transposed accumulator epilogue.
#pragma once
#include "common.h"
#include "../../include/macaw_hip.h"

namespace mkg {

struct GemmArgs {
  const void* A; const void* B; void* C; const void* R; const void* bias;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int nb2;
  long sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  float alpha;
  const float* scale_a; const float* scale_b;  // optional device de-quantisation scales (fp8): scalars
                                               // multiplied into alpha, or (scale_vec) one per row of A /
                                               // per row of B = per output row / output column
  int scale_vec;
  int bias_mode, act, accumulate;
  int tiles_m, tiles_n;
  int a_vec, b_vec;  // 1: 16-byte aligned vector loads allowed
  int c_vec;         // vector C/R access: 2 = 16-byte aligned (pitches / strides % 8 == 0), 1 = 8-byte, 0 = none
  // stream-K tail (v2 kernel only): blocks [0, dp_tiles) own whole tiles; the remaining
  // tiles are cut into `split` K-pieces of `kt_per_piece` K-tiles, one block each.
  int dp_tiles, split, kt_per_piece;
  int walkers;     // v7: workgroups [0, walkers) walk the whole tiles with stride `walkers`, the rest are the tail
  int tail8;       // v7: the spatial tail is cut into 64 x 128 eighths (K-major A, <= 32 tail tiles) instead of quarters
  int lin_batch;   // 1: batch index is folded into the linear tile index (grid.z == 1)
  int ablate;      // debug only (MK_GEMM_ABLATE): 1 = skip global->LDS, 2 = skip barrier wait
  const void* pro_w; float pro_eps;   // skinny kernel prologue (mk_decode_linear): RMSNorm weight / eps
  float* ws;       // fp32 slabs [tail tile][piece][64 regs][256 threads]
  int* counters;   // arrival counter per tail tile (zeroed by the launcher)
};

MK_DEV float apply_act(float v, int act) {
  if (act == 1) return 0.5f * v * (1.0f + mk_erf(v * 0.70710678118654752440f));
  if (act == 2) return v / (1.0f + __expf(-1.702f * v));
  return v;
}

// XCD-aware + grouped tile order (cdna_hip_programming.md T1, bijective form).
MK_DEV int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
MK_DEV void tile_from_index(int wg, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M);
MK_DEV void tile_coords(int bid, int tiles_m, int tiles_n, int& tm, int& tn) {
  tile_from_index(xcd_remap(bid, tiles_m * tiles_n), tiles_m, tiles_n, tm, tn, 8);
}
MK_DEV void tile_from_index(int wg, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = 8) {
  const int per_group = GROUP_M * tiles_n;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsize = min(tiles_m - first_m, GROUP_M);
  const int in_g = wg - group * per_group;
  tm = first_m + in_g % gsize;
  tn = in_g / gsize;
}

// Epilogue for one wave's 64x64 accumulator block (2x2 fragments of 32x32).
// D[i = n][j = m]: lane holds m = l&31, n = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// The accumulator layout gives a lane ONE output row and 4-column groups 8 apart: written
// straight to C that is 16 B per row per instruction (32 different lines each), and a residual /
// bias read in that layout sat in a conditional block per group -- sixteen serialised HBM round
// trips per wave tile (an epilogue with bias + residual cost 15-50 % of a K <= 1024 GEMM).  So the
// tile is transposed through LDS (free after the K loop): each wave stages 32 rows x (FN * 32)
// fp32 in its private 8 KiB (float4 index XOR row: conflict-free both ways), reads them back
// row-major -- 16 lanes per 64-column row -- and every load / store is a full 128-byte line per
// row; the residual rows of a pass group are all requested before the first is used.
// SV: per-row x per-column de-quantisation scales (g.scale_vec) are compiled in -- the fp8 instantiations
// only: carried as a run-time branch the four column scales and the row-scale pointer stay live beside
// the accumulators and cost every 16-bit v2 kernel its fourth wave per SIMD (125 -> 136-142 VGPRs).
template <bool SV, typename ET, int FM, int FN>
MK_DEV void wave_epilogue(const f32x16 (&acc)[FM][FN], const GemmArgs& g, ET* C, const ET* Rp,
                          int m0, int n0, int wm0, int wn0, char* smem) {
  typedef typename E16<ET>::x4 e16x4;
  constexpr int W4 = FN * 8;        // float4 per staged row
  constexpr int RPI = 64 / W4;      // rows per pass
  constexpr int NPASS = 32 / RPI;   // passes per 32-row fragment
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  float alpha = g.alpha;
  const bool svec = SV && g.scale_vec != 0;
  if (!svec) {
    if (g.scale_a) alpha *= g.scale_a[0];
    if (g.scale_b) alpha *= g.scale_b[0];
  }
  __syncthreads();                  // every wave is done with the operand tiles in LDS
  float* buf = reinterpret_cast<float*>(smem) + w * 2048;
  const int srow = l & 31, sh = l >> 5;          // staging: this lane's accumulator row / half
  const int c4 = l % W4, rsub = l / W4;          // read-back: float4 column and row inside a pass
  const int ncol = n0 + wn0 + c4 * 4;            // first of this lane's 4 output columns
  const bool cols_full = ncol + 3 < g.N;
  // fast path: whole wave on aligned, in-range 4-column groups (always true off the N edge)
  const bool fast = g.c_vec && __all(cols_full ? 1 : 0);
  float sb[4] = {1.f, 1.f, 1.f, 1.f};      // per-column de-quantisation scales (fp8, scale_vec)
  if (svec) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sb[e] = alpha * g.scale_b[min(ncol + e, g.N - 1)];
  }
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias_mode == 1) {
    const ET* bp = reinterpret_cast<const ET*>(g.bias);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = (float)bp[min(ncol + e, g.N - 1)];
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    // ---- stage fragment row block i
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cw = (j * 8 + 2 * q + sh) ^ (srow & (W4 - 1));
        *reinterpret_cast<float4*>(buf + (srow * W4 + cw) * 4) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
    const int mbase = m0 + wm0 + i * 32 + rsub;
    if (fast) {
      // residual / accumulate rows of ALL passes requested up front (clamped row, discarded later)
      e16x4 rv[NPASS], cv[NPASS];
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const long mc = min(mbase + p * RPI, g.M - 1);
        if (Rp) rv[p] = *reinterpret_cast<const e16x4*>(Rp + mc * g.ldr + ncol);
        if (g.accumulate) cv[p] = *reinterpret_cast<const e16x4*>(C + mc * g.ldc + ncol);
      }
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPI + rsub, m = mbase + p * RPI;
        const float4 t = *reinterpret_cast<const float4*>(buf + (row * W4 + (c4 ^ (row & (W4 - 1)))) * 4);
        float v[4] = {alpha * t.x, alpha * t.y, alpha * t.z, alpha * t.w};
        if (svec) {
          const float sa = g.scale_a[min(m, g.M - 1)];
          v[0] = t.x * (sa * sb[0]); v[1] = t.y * (sa * sb[1]); v[2] = t.z * (sa * sb[2]); v[3] = t.w * (sa * sb[3]);
        }
        if (g.bias_mode == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bv[e];
        } else if (g.bias_mode == 2) {
          const float bm = (float)reinterpret_cast<const ET*>(g.bias)[min(m, g.M - 1)];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += bm;
        }
        if (g.act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], g.act);
        }
        if (Rp) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[p][e];
        }
        if (g.accumulate) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)cv[p][e];
        }
        if (m < g.M) {
          e16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (ET)v[e];
          *reinterpret_cast<e16x4*>(C + (long)m * g.ldc + ncol) = o;
        }
      }
    } else {
      // N edge / unaligned C: same order of operations, element by element
#pragma unroll 1
      for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPI + rsub, m = mbase + p * RPI;
        const float4 t = *reinterpret_cast<const float4*>(buf + (row * W4 + (c4 ^ (row & (W4 - 1)))) * 4);
        const float tv[4] = {t.x, t.y, t.z, t.w};
        if (m >= g.M) continue;
        float bm = 0.f;
        if (g.bias_mode == 2) bm = (float)reinterpret_cast<const ET*>(g.bias)[m];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = ncol + e;
          if (n >= g.N) continue;
          float x = svec ? tv[e] * (g.scale_a[m] * sb[e]) : alpha * tv[e];
          if (g.bias_mode == 1) x += bv[e];
          else if (g.bias_mode == 2) x += bm;
          if (g.act) x = apply_act(x, g.act);
          if (Rp) x += (float)Rp[(long)m * g.ldr + n];
          ET* cp = C + (long)m * g.ldc + n;
          if (g.accumulate) x += (float)*cp;
          *cp = (ET)x;
        }
      }
    }
  }
}

// The same epilogue with 16-byte accesses, for the 256 x 256 tile kernels (gemm_v7 / gemm_v8: registers to spare;
// the 128 x 128 kernels keep the 8-byte form above -- the wide one costs them their fourth wave per SIMD).
// D[i = n][j = m]: lane holds m = l&31, n = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
// The accumulator layout gives a lane ONE output row and 4-column groups 8 apart: written
// straight to C that is 16 B per row per instruction (32 different lines each), and a residual /
// bias read in that layout sat in a conditional block per group -- sixteen serialised HBM round
// trips per wave tile (an epilogue with bias + residual cost 15-50 % of a K <= 1024 GEMM).  So the
// tile is transposed through LDS (free after the K loop): each wave stages 32 rows x (FN * 32)
// fp32 in its private buffer (float4 index XOR row: conflict-free both ways) and reads them back
// row-major, EIGHT columns = 16 bytes of output per lane (round 4; rounds 1-3 moved 8 bytes per lane:
// `global_store_dwordx2` runs at about half the per-instruction rate of `dwordx4` and the epilogue is
// store-ISSUE-bound -- every wave queues its 512-byte stores behind the other waves'; measured on one
// round of 256 x 256 tiles, 4096 x 4096 x K: 16 us of fixed time per tile on the 8-wave kernel, 34 us
// on the 4-wave one, against 9 us in the vendor library).  Every load / store is a full 128-byte line
// per row (FN = 1: 64-byte halves); the residual rows of a pass group are all requested before the
// first is used.
// SV: per-row x per-column de-quantisation scales (g.scale_vec) are compiled in -- the fp8 instantiations
// only: carried as a run-time branch the four column scales and the row-scale pointer stay live beside
// the accumulators and cost every 16-bit v2 kernel its fourth wave per SIMD (125 -> 136-142 VGPRs).
template <bool SV, typename ET, int FM, int FN>
MK_DEV void wave_epilogue16(const f32x16 (&acc)[FM][FN], const GemmArgs& g, ET* C, const ET* Rp,
                          int m0, int n0, int wm0, int wn0, char* smem, const bool sync = true,
                          const int wave_off = -1) {
  typedef typename E16<ET>::x8 e16x8;
  constexpr int W4 = FN * 8;        // float4 per staged row
  constexpr int LPR = W4 / 2;       // lanes per row on the way out (8 columns each)
  constexpr int RPI = 64 / LPR;     // rows per pass
  constexpr int NPASS = 32 / RPI;   // passes per 32-row fragment
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  float alpha = g.alpha;
  const bool svec = SV && g.scale_vec != 0;
  if (!svec) {
    if (g.scale_a) alpha *= g.scale_a[0];
    if (g.scale_b) alpha *= g.scale_b[0];
  }
  if (sync) __syncthreads();        // every wave is done with the operand tiles in LDS
  // 32 rows x FN * 32 fp32 per wave; wave_off: the caller's byte offset for this wave (a walking workgroup
  // keeps the staging buffers out of the ring slots its next tile is already being loaded into)
  float* buf = wave_off >= 0 ? reinterpret_cast<float*>(smem + wave_off)
                             : reinterpret_cast<float*>(smem) + w * (FN > 2 ? 1024 * FN : 2048);
  const int srow = l & 31, sh = l >> 5;          // staging: this lane's accumulator row / half
  const int c8 = l % LPR, rsub = l / LPR;        // read-back: 8-column group and row inside a pass
  const int ncol = n0 + wn0 + c8 * 8;            // first of this lane's 8 output columns
  const bool cols_full = ncol + 7 < g.N;
  // fast path: whole wave on 16-byte aligned, in-range 8-column groups (always true off the N edge)
  const bool fast = g.c_vec == 2 && __all(cols_full ? 1 : 0);
  // (Round 4, measured and removed: 16-bit staging -- alpha / bias / activation on the accumulator layout, the value
  // rounded BEFORE the trip through LDS, 8 bytes per 4-column group staged: half the LDS bytes, bit-identical.
  // Cold in the harness +0.3 ... +7 % per launch; in the cfg-3 step -1.1 % of GEMM time, with the residual rows
  // read on the accumulator layout as well -1.2 % (gpurun_out/r04/l3, l4).  On the way: laundering the thread
  // index at the top of this function stops the compiler from forming the epilogue's lane invariants at the kernel
  // entry and takes the 256 x 256 kernel from 251 to 213-227 VGPRs -- and costs 0.7 % of GEMM time in the step
  // (170.3 vs 169.1 ms, three alternations on one box, gpurun_out/r04/l5): not kept either.)
  float sb[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};      // per-column de-quantisation scales (fp8, scale_vec)
  if (svec) {
#pragma unroll
    for (int e = 0; e < 8; ++e) sb[e] = alpha * g.scale_b[min(ncol + e, g.N - 1)];
  }
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (g.bias_mode == 1) {
    const ET* bp = reinterpret_cast<const ET*>(g.bias);
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (float)bp[min(ncol + e, g.N - 1)];
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    // ---- stage fragment row block i
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cw = (j * 8 + 2 * q + sh) ^ (srow & (W4 - 1));
        *reinterpret_cast<float4*>(buf + (srow * W4 + cw) * 4) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
      }
    const int mbase = m0 + wm0 + i * 32 + rsub;
    if (fast) {
      // residual / accumulate rows of ALL passes requested up front (clamped row, discarded later)
      e16x8 rv[NPASS], cv[NPASS];
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const long mc = min(mbase + p * RPI, g.M - 1);
        if (Rp) rv[p] = *reinterpret_cast<const e16x8*>(Rp + mc * g.ldr + ncol);
        if (g.accumulate) cv[p] = *reinterpret_cast<const e16x8*>(C + mc * g.ldc + ncol);
      }
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPI + rsub, m = mbase + p * RPI;
        const int x = row & (W4 - 1);
        const float4 t0 = *reinterpret_cast<const float4*>(buf + (row * W4 + ((2 * c8) ^ x)) * 4);
        const float4 t1 = *reinterpret_cast<const float4*>(buf + (row * W4 + ((2 * c8 + 1) ^ x)) * 4);
        const float t[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        float v[8];
        if (svec) {
          const float sa = g.scale_a[min(m, g.M - 1)];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = t[e] * (sa * sb[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = alpha * t[e];
        }
        if (g.bias_mode == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bv[e];
        } else if (g.bias_mode == 2) {
          const float bm = (float)reinterpret_cast<const ET*>(g.bias)[min(m, g.M - 1)];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bm;
        }
        if (g.act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], g.act);
        }
        if (Rp) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rv[p][e];
        }
        if (g.accumulate) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)cv[p][e];
        }
        if (m < g.M) {
          e16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (ET)v[e];
#ifdef MK_EPI_NOSTORE   // timing-only experiment builds: the value is formed, the store is not issued
          asm volatile("" :: "v"(o));
#else
          *reinterpret_cast<e16x8*>(C + (long)m * g.ldc + ncol) = o;
#endif
        }
      }
    } else {
      // N edge / unaligned C: same order of operations, element by element
#pragma unroll 1
      for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPI + rsub, m = mbase + p * RPI;
        const int x = row & (W4 - 1);
        const float4 t0 = *reinterpret_cast<const float4*>(buf + (row * W4 + ((2 * c8) ^ x)) * 4);
        const float4 t1 = *reinterpret_cast<const float4*>(buf + (row * W4 + ((2 * c8 + 1) ^ x)) * 4);
        const float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        if (m >= g.M) continue;
        float bm = 0.f;
        if (g.bias_mode == 2) bm = (float)reinterpret_cast<const ET*>(g.bias)[m];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int n = ncol + e;
          if (n >= g.N) continue;
          float xv = svec ? tv[e] * (g.scale_a[m] * sb[e]) : alpha * tv[e];
          if (g.bias_mode == 1) xv += bv[e];
          else if (g.bias_mode == 2) xv += bm;
          if (g.act) xv = apply_act(xv, g.act);
          if (Rp) xv += (float)Rp[(long)m * g.ldr + n];
          ET* cp = C + (long)m * g.ldc + n;
          if (g.accumulate) xv += (float)*cp;
          *cp = (ET)xv;
        }
      }
    }
  }
}

// The compiler sometimes loses the wave-uniformity of a tile base pointer (then every
// buffer_load ... lds becomes a 12-instruction waterfall loop over the descriptor): pin it.
template <typename T>
MK_DEV const T* uniform_ptr(const T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}

}  // namespace mkg
