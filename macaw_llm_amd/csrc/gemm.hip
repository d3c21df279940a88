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
#include <cstdlib>
#include <utility>
#include <vector>

namespace {
using namespace mkg;

// ------------------------------------------------------------------ bf16 --
constexpr int BM = 128, BN = 128, BK = 64;
#ifndef MK_GEMM_DEFAULT_CFG
#define MK_GEMM_DEFAULT_CFG 5
#endif
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

// 8 bf16 from global with zero fill; `valid` = number of leading valid elements.
// Slow path (edge tiles / unaligned operands): kept out of line so the hot loop
// stays small.
__device__ __attribute__((noinline)) uint4 load_chunk_slow(const bf16* p, int valid, bool vec) {
  if (valid >= 8 && vec) return *reinterpret_cast<const uint4*>(p);
  const unsigned short* s = reinterpret_cast<const unsigned short*>(p);
  unsigned e0 = valid > 0 ? s[0] : 0, e1 = valid > 1 ? s[1] : 0, e2 = valid > 2 ? s[2] : 0,
           e3 = valid > 3 ? s[3] : 0, e4 = valid > 4 ? s[4] : 0, e5 = valid > 5 ? s[5] : 0,
           e6 = valid > 6 ? s[6] : 0, e7 = valid > 7 ? s[7] : 0;
  return make_uint4(e0 | (e1 << 16), e2 | (e3 << 16), e4 | (e5 << 16), e6 | (e7 << 16));
}

// Stage one 128(rows) x 64(k) operand tile: global -> regs.
template <bool RED_MAJOR>
MK_DEV void tile_load(const bf16* base, long ld, int row0, int k0, int R, int K, bool vec,
                      uint4 (&r)[4]) {
  const int tid = threadIdx.x;
  const bool interior = vec && (row0 + 128 <= R) && (k0 + BK <= K);  // block-uniform
  if (interior) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if constexpr (!RED_MAJOR) {
        const int row = c >> 3, kc = c & 7;
        r[i] = *reinterpret_cast<const uint4*>(base + (long)(row0 + row) * ld + k0 + kc * 8);
      } else {
        const int kr = c >> 4, mc = c & 15;
        r[i] = *reinterpret_cast<const uint4*>(base + (long)(k0 + kr) * ld + row0 + mc * 8);
      }
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      uint4 v;
      if constexpr (!RED_MAJOR) {
        const int row = c >> 3, kc = c & 7;
        const int gr = row0 + row, gk = k0 + kc * 8;
        const int valid = (gr < R) ? min(max(K - gk, 0), 8) : 0;
        v = load_chunk_slow(base + (long)gr * ld + gk, valid, vec);
      } else {
        const int kr = c >> 4, mc = c & 15;
        const int gk = k0 + kr, gr = row0 + mc * 8;
        const int valid = (gk < K) ? min(max(R - gr, 0), 8) : 0;
        v = load_chunk_slow(base + (long)gk * ld + gr, valid, vec);
      }
      if (i == 0) r[0] = v; else if (i == 1) r[1] = v; else if (i == 2) r[2] = v; else r[3] = v;
    }
  }
}
// regs -> LDS image (swizzled).
template <bool RED_MAJOR>
MK_DEV void tile_store(char* lds, const uint4 (&r)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    int off;
    if constexpr (!RED_MAJOR) {
      const int row = c >> 3, kc = c & 7;
      off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    } else {
      const int kr = c >> 4, mc = c & 15;
      off = kr * 256 + ((mc ^ (4 * (kr & 3))) << 4);
    }
    *reinterpret_cast<uint4*>(lds + off) = r[i];
  }
}
// Interior tile: global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction,
// no VGPR round trip and no ds_write pass).  The LDS destination of a wave instruction is
// linear (base + lane*16), so the swizzle of the LDS image is applied to the per-lane SOURCE
// address instead (cdna_hip_programming.md rule 21); each 128-B / 256-B global row segment is
// still fetched whole.
template <bool RED_MAJOR>
MK_DEV void tile_glds(const bf16* base, long ld, int row0, int k0, char* lds_tile) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = w + 4 * i;  // 1-KiB piece of the 16-KiB tile
    const bf16* gp;
    if constexpr (!RED_MAJOR) {
      const int row = p * 8 + (l >> 3);
      const int kc = (l & 7) ^ ((row >> 1) & 7);
      gp = base + (long)(row0 + row) * ld + k0 + kc * 8;
    } else {
      const int kr = p * 4 + (l >> 4);
      const int mc = (l & 15) ^ (4 * (kr & 3));
      gp = base + (long)(k0 + kr) * ld + row0 + mc * 8;
    }
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gp,
        (__attribute__((address_space(3))) void*)(lds_tile + p * 1024), 16, 0, 0);
  }
}
// Fragment for v_mfma_f32_32x32x16_bf16: lane l holds row (l&31), k = 8*(l>>5)+j.
template <bool RED_MAJOR>
MK_DEV bf16x8 frag_load(const char* lds, int row_base, int ks) {
  const int l = threadIdx.x & 63;
  if constexpr (!RED_MAJOR) {
    const int row = row_base + (l & 31);
    const int kc = ks * 2 + (l >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    // two transpose reads of a [4 k][16 rows] block each
    const int li = l & 15;
    const int col = row_base + 16 * ((l >> 4) & 1) + 4 * (li & 3);
    const int kr0 = ks * 16 + 8 * (l >> 5) + (li >> 2);
    bf16x8 out;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int kr = kr0 + 4 * r;
      const int off = kr * 256 + (((col >> 3) ^ (4 * (kr & 3))) << 4) + ((col & 7) << 1);
      bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
          (__attribute__((address_space(3))) bf16x4*)(lds + off));
      out[4 * r + 0] = t[0]; out[4 * r + 1] = t[1]; out[4 * r + 2] = t[2]; out[4 * r + 3] = t[3];
    }
    return out;
  }
}


template <bool A_RED, bool B_RED, bool GLDS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  tile_coords(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
  const int z = blockIdx.z, z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const bf16* A = reinterpret_cast<const bf16*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16* B = reinterpret_cast<const bf16*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  bf16* C = reinterpret_cast<bf16*>(g.C) + z1 * g.sC1 + z2 * g.sC2;
  const bf16* Rp = g.R ? reinterpret_cast<const bf16*>(g.R) + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, w = tid >> 6;
  const int wm0 = (w >> 1) * 64, wn0 = (w & 1) * 64;

  // LDS map: [A0 | B0 | A1 | B1], 16 KiB each.

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (g.K + BK - 1) / BK;
  auto compute = [&](const char* la, const char* lb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 fm[2], fn[2];
      fm[0] = frag_load<A_RED>(la, wm0, ks);
      fm[1] = frag_load<A_RED>(la, wm0 + 32, ks);
      fn[0] = frag_load<B_RED>(lb, wn0, ks);
      fn[1] = frag_load<B_RED>(lb, wn0 + 32, ks);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fn[j], fm[i], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (GLDS) {
    // Pipeline: [wait tile kt landed; barrier] -> issue tile kt+1 (LDS-DMA, other buffer) ->
    // 16 MFMAs on tile kt.  One barrier per K-tile, loads in flight during the MFMA block.
    const bool a_rows = g.a_vec && (m0 + BM <= g.M);
    const bool b_rows = g.b_vec && (n0 + BN <= g.N);
    auto stage = [&](int kt, char* buf) {
      const int k0 = kt * BK;
      const bool kfull = (k0 + BK <= g.K);
      if (a_rows && kfull) tile_glds<A_RED>(A, g.lda, m0, k0, buf);
      else {
        uint4 r[4];
        tile_load<A_RED>(A, g.lda, m0, k0, g.M, g.K, g.a_vec, r);
        tile_store<A_RED>(buf, r);
      }
      if (b_rows && kfull) tile_glds<B_RED>(B, g.ldb, n0, k0, buf + TILE_BYTES);
      else {
        uint4 r[4];
        tile_load<B_RED>(B, g.ldb, n0, k0, g.N, g.K, g.b_vec, r);
        tile_store<B_RED>(buf + TILE_BYTES, r);
      }
    };
    stage(0, smem);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, smem + (cur ^ 1) * (2 * TILE_BYTES));
      const char* la = smem + cur * (2 * TILE_BYTES);
      compute(la, la + TILE_BYTES);
    }
  } else {
    uint4 ra[4], rb[4];
    tile_load<A_RED>(A, g.lda, m0, 0, g.M, g.K, g.a_vec, ra);
    tile_load<B_RED>(B, g.ldb, n0, 0, g.N, g.K, g.b_vec, rb);
    tile_store<A_RED>(smem, ra);
    tile_store<B_RED>(smem + TILE_BYTES, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const bool more = (kt + 1 < nk);
      if (more) {
        tile_load<A_RED>(A, g.lda, m0, (kt + 1) * BK, g.M, g.K, g.a_vec, ra);
        tile_load<B_RED>(B, g.ldb, n0, (kt + 1) * BK, g.N, g.K, g.b_vec, rb);
      }
      const char* la = smem + cur * (2 * TILE_BYTES);
      compute(la, la + TILE_BYTES);
      if (more) {
        char* na = smem + (cur ^ 1) * (2 * TILE_BYTES);
        tile_store<A_RED>(na, ra);
        tile_store<B_RED>(na + TILE_BYTES, rb);
      }
      __syncthreads();
    }
  }

  wave_epilogue(acc, g, C, Rp, m0, n0, wm0, wn0, smem);
}

// ------------------------------------------------ v2: issue-lean LDS-DMA kernel --
// PMC on the kernels above showed ~5.6 VALU instructions per MFMA (swizzle / 64-bit address
// arithmetic recomputed every K-tile) and only ~40 % matrix-pipe occupancy: instruction issue,
// not LDS or L2 bandwidth, was the limiter.  v2 keeps the 128x128x64 tile / 4 waves / 2
// workgroups per CU, but moves every per-lane address computation out of the K loop:
//   * global -> LDS by buffer_load_dwordx4 ... lds with a per-lane voffset computed ONCE and a
//     scalar soffset advanced by SALU per K-tile (no VALU, no VGPR staging, no ds_write);
//   * fragment LDS offsets precomputed per lane (K-major: 8 per operand; reduction-major: 2 per
//     operand + immediates), the two LDS stages addressed through immediate offsets by
//     unrolling the K loop by two;
//   * M / N edge tiles need no predicates: K-major rows are clamped to the last valid row
//     (garbage only reaches discarded outputs), reduction-major over-reads stay inside the
//     buffer or hit the SRD bound (returns 0).
// Requires 16-byte aligned operands and K % 64 == 0; anything else runs the generic kernel.
// K-major LDS image of a [128][BKv] tile: 16-B chunk swizzle (conflict-free ds_read_b128)
template <int BKv>
MK_DEV int kswz(int row, int kc) {
  if constexpr (BKv == 64) return kc ^ ((row >> 1) & 7);
  else return kc ^ ((row >> 2) & 3);
}
template <bool RED_MAJOR, int BKv, int NWv>
MK_DEV void v2_voffsets(int row0, int R, long ld, int w, int l, int (&voff)[4]) {
  constexpr int CR = BKv / 8;          // 16-B chunks per K-major row
#pragma unroll
  for (int i = 0; i < BKv / 4 / NWv; ++i) {
    const int p = w + NWv * i;
    if constexpr (!RED_MAJOR) {
      const int row = p * (64 / CR) + l / CR;
      const int kc = kswz<BKv>(row, l % CR);
      const int gr = min(row0 + row, R - 1) - row0;  // may be negative only if row0 >= R (never)
      voff[i] = (int)((long)gr * ld * 2 + kc * 16);
    } else {
      const int kr = p * 4 + (l >> 4);
      const int mc = (l & 15) ^ (4 * (kr & 3));
      voff[i] = (int)((long)kr * ld * 2 + mc * 16);
    }
  }
}
template <bool RED_MAJOR, int BKv>
MK_DEV void v2_frag_offsets(int wrow0, int l, int (&off)[2][4]) {
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    if constexpr (!RED_MAJOR) {
      const int row = wrow0 + f * 32 + (l & 31);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int kc = (ks * 2 + (l >> 5)) % (BKv / 8);
        off[f][ks] = row * (BKv * 2) + (kswz<BKv>(row, kc) << 4);
      }
    } else {
      const int li = l & 15;
      const int col = wrow0 + f * 32 + 16 * ((l >> 4) & 1) + 4 * (li & 3);
      const int kr = 8 * (l >> 5) + (li >> 2);
      off[f][0] = kr * 256 + (((col >> 3) ^ (4 * (kr & 3))) << 4) + ((col & 7) << 1);
      off[f][1] = off[f][2] = off[f][3] = 0;
    }
  }
}
// NWv = 4: waves 2 x 2 of 64 x 64 (2 x 2 fragments).  NWv = 8: waves 2 (M) x 4 (N) of 64 x 32
// (2 x 1 fragments, 32 accumulator registers) -- the same tile and LDS image with twice the
// waves per SIMD to hide LDS / barrier latency (the kernel is latency- not bandwidth-bound).
template <bool A_RED, bool B_RED, int BKv, int NWv, bool FP8 = false>
MK_DEV void v2_body(const GemmArgs& g) {
  constexpr int TILE_B = 128 * BKv * 2;   // bytes per operand tile
  constexpr int NP = BKv / 4 / NWv;       // LDS-DMA pieces per wave per operand tile
  constexpr int NKS = BKv / 16;           // MFMA k-steps per tile
  constexpr int FN = NWv == 8 ? 1 : 2;    // N fragments per wave
  constexpr int NT = NWv * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  int piece = -1, tail_idx = 0, zlin = 0;
  // K % BKv != 0 only reaches this kernel with BOTH operands reduction-major: rows >= K are
  // then outside the buffer descriptors and load as zeros
  int kt_begin = 0, kt_end = (g.K + BKv - 1) / BKv;
  {
    const int bid = blockIdx.x;
    int t;
    if (bid < g.dp_tiles) {
      t = xcd_remap(bid, g.dp_tiles);
    } else {
      const int r = bid - g.dp_tiles;
      tail_idx = r / g.split;
      piece = r - tail_idx * g.split;
      t = g.dp_tiles + tail_idx;
      kt_begin = piece * g.kt_per_piece;
      kt_end = min(kt_end, kt_begin + g.kt_per_piece);
    }
    if (g.lin_batch) {
      const int per = g.tiles_m * g.tiles_n;
      zlin = t / per;
      t -= zlin * per;
    }
    tile_from_index(t, g.tiles_m, g.tiles_n, tm, tn, (g.ablate >> 8) ? (g.ablate >> 8) : 8);
  }
  const int z = g.lin_batch ? zlin : (int)blockIdx.z, z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const bf16* A = reinterpret_cast<const bf16*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16* B = reinterpret_cast<const bf16*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  bf16* C = reinterpret_cast<bf16*>(g.C) + z1 * g.sC1 + z2 * g.sC2;
  const bf16* Rp = g.R ? reinterpret_cast<const bf16*>(g.R) + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int m0 = tm * BM, n0 = tn * BN;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = NWv == 8 ? (w >> 2) * 64 : (w >> 1) * 64;
  const int wn0 = NWv == 8 ? (w & 3) * 32 : (w & 1) * 64;

  // (the hardware range check is per DWORD: with an odd number of valid rows the last element of
  // the last k-row shares its dword with the first out-of-range one and would load as zero, so
  // the bound is rounded up to the dword; that element lies inside the pitch since ld % 8 == 0)
  // buffer descriptors (tile-relative bases keep voffset small; num_records bounds the
  // reduction-major over-read of the last K row)
  const bf16* abase = A_RED ? A + m0 : A + (long)m0 * g.lda;
  const bf16* bbase = B_RED ? B + n0 : B + (long)n0 * g.ldb;
  const long a_bytes = A_RED ? ((long)(g.K - 1) * g.lda + ((g.M - m0 + 1) & ~1)) * 2
                             : ((long)(min(g.M - m0, BM) - 1) * g.lda + ((g.K + 1) & ~1)) * 2;
  const long b_bytes = B_RED ? ((long)(g.K - 1) * g.ldb + ((g.N - n0 + 1) & ~1)) * 2
                             : ((long)(min(g.N - n0, BN) - 1) * g.ldb + ((g.K + 1) & ~1)) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)abase, 0, (int)min(a_bytes, 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bbase, 0, (int)min(b_bytes, 0x7fffffffL), 0x00020000);
  int voffA[4], voffB[4];
  v2_voffsets<A_RED, BKv, NWv>(m0, g.M, g.lda, w, l, voffA);
  v2_voffsets<B_RED, BKv, NWv>(n0, g.N, g.ldb, w, l, voffB);
  const int stepA = A_RED ? (int)(BKv * g.lda * 2) : BKv * 2;  // bytes per K-tile
  const int stepB = B_RED ? (int)(BKv * g.ldb * 2) : BKv * 2;
  int offA[2][4], offB[2][4];
  v2_frag_offsets<A_RED, BKv>(wm0, l, offA);
  v2_frag_offsets<B_RED, BKv>(wn0, l, offB);

  f32x16 acc[2][FN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int sA = kt_begin * stepA, sB = kt_begin * stepB;  // scalar byte offsets of the next K-tile
  auto issue = [&](int stage) {
    char* la = smem + stage * (2 * TILE_B) + w * 1024;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsA, (__attribute__((address_space(3))) void*)(la + i * (NWv * 1024)), 16, voffA[i], sA, 0, 0);
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (__attribute__((address_space(3))) void*)(la + TILE_B + i * (NWv * 1024)), 16,
          voffB[i], sB, 0, 0);
    sA += stepA;
    sB += stepB;
  };
#define MK_V2_FRAG(RED, OFF, F, KS, BASE)                                                        \
  [&]() -> bf16x8 {                                                                              \
    if constexpr (!(RED)) {                                                                      \
      return *reinterpret_cast<const bf16x8*>(smem + (BASE) + OFF[F][KS]);                       \
    } else {                                                                                     \
      bf16x8 o_;                                                                                 \
      bf16x4 t0_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                     \
          (__attribute__((address_space(3))) bf16x4*)(smem + (BASE) + (KS) * 16 * 256 + OFF[F][0])); \
      bf16x4 t1_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                     \
          (__attribute__((address_space(3))) bf16x4*)(smem + (BASE) + ((KS) * 16 + 4) * 256 + OFF[F][0])); \
      o_[0] = t0_[0]; o_[1] = t0_[1]; o_[2] = t0_[2]; o_[3] = t0_[3];                            \
      o_[4] = t1_[0]; o_[5] = t1_[1]; o_[6] = t1_[2]; o_[7] = t1_[3];                            \
      return o_;                                                                                 \
    }                                                                                            \
  }()
#define MK_V2_LOAD4(DST, KS, STAGE)                                                              \
  do {                                                                                           \
    DST[0] = MK_V2_FRAG(A_RED, offA, 0, KS, (STAGE) * 2 * TILE_B);                           \
    DST[1] = MK_V2_FRAG(A_RED, offA, 1, KS, (STAGE) * 2 * TILE_B);                           \
    DST[2] = MK_V2_FRAG(B_RED, offB, 0, KS, (STAGE) * 2 * TILE_B + TILE_B);              \
    if constexpr (FN == 2) DST[3] = MK_V2_FRAG(B_RED, offB, 1, KS, (STAGE) * 2 * TILE_B + TILE_B); \
  } while (0)
#define MK_V2_MFMA4(F)                                                                           \
  do {                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                               \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[2], F[0], acc[0][0], 0, 0, 0);         \
    if constexpr (FN == 2)                                                                       \
      acc[0][FN - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[3], F[0], acc[0][FN - 1], 0, 0, 0); \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[2], F[1], acc[1][0], 0, 0, 0);         \
    if constexpr (FN == 2)                                                                       \
      acc[1][FN - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F[3], F[1], acc[1][FN - 1], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                               \
  } while (0)
// fp8 (e4m3) operands: the tile bytes, LDS image and the two 16-byte reads per fragment are
// those of two bf16 k-steps; they feed ONE v_mfma_scale_f32_32x32x64_f8f6f4 (64 fp8 k-slots,
// twice the bf16 MFMA rate; any k-slot permutation is fine as long as A and B agree, and both
// are K-major here).  Scales are 2^0: the per-tensor scales are applied in the epilogue.
#define MK_V2_CAT(LO, HI)                                                                        \
  [&]() -> i32x8 {                                                                               \
    const i32x4 lo_ = __builtin_bit_cast(i32x4, LO), hi_ = __builtin_bit_cast(i32x4, HI);        \
    i32x8 r_;                                                                                    \
    r_[0] = lo_[0]; r_[1] = lo_[1]; r_[2] = lo_[2]; r_[3] = lo_[3];                              \
    r_[4] = hi_[0]; r_[5] = hi_[1]; r_[6] = hi_[2]; r_[7] = hi_[3];                              \
    return r_;                                                                                   \
  }()
#define MK_V2_MFMA4_FP8(F, G)                                                                    \
  do {                                                                                           \
    const i32x8 a0_ = MK_V2_CAT(F[0], G[0]), a1_ = MK_V2_CAT(F[1], G[1]);                        \
    const i32x8 b0_ = MK_V2_CAT(F[2], G[2]), b1_ = MK_V2_CAT(F[3], G[3]);                        \
    __builtin_amdgcn_s_setprio(1);                                                               \
    acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b0_, a0_, acc[0][0], 0, 0, 0, 127, 0, 127); \
    acc[0][FN - 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b1_, a0_, acc[0][FN - 1], 0, 0, 0, 127, 0, 127); \
    acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b0_, a1_, acc[1][0], 0, 0, 0, 127, 0, 127); \
    acc[1][FN - 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b1_, a1_, acc[1][FN - 1], 0, 0, 0, 127, 0, 127); \
    __builtin_amdgcn_s_setprio(0);                                                               \
  } while (0)
// fragments of k-step ks+1 are requested before the MFMAs of k-step ks (two register sets)
#define MK_V2_COMPUTE(STAGE)                                                                     \
  do {                                                                                           \
    bf16x8 fa_[4], fb_[4];                                                                       \
    MK_V2_LOAD4(fa_, 0, STAGE);                                                                  \
    MK_V2_LOAD4(fb_, 1, STAGE);                                                                  \
    if constexpr (FP8) {                                                                         \
      bf16x8 fc_[4], fd_[4];                                                                     \
      MK_V2_LOAD4(fc_, 2, STAGE);                                                                \
      MK_V2_LOAD4(fd_, 3, STAGE);                                                                \
      MK_V2_MFMA4_FP8(fa_, fb_);                                                                 \
      MK_V2_MFMA4_FP8(fc_, fd_);                                                                 \
      break;                                                                                     \
    }                                                                                            \
    MK_V2_MFMA4(fa_);                                                                            \
    if (NKS == 4) {                                                                              \
      MK_V2_LOAD4(fa_, 2, STAGE);                                                                \
      MK_V2_MFMA4(fb_);                                                                          \
      MK_V2_LOAD4(fb_, 3, STAGE);                                                                \
      MK_V2_MFMA4(fa_);                                                                          \
    }                                                                                            \
    MK_V2_MFMA4(fb_);                                                                            \
  } while (0)
#define MK_V2_SYNC()                                          \
  do {                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_s_barrier();                             \
    asm volatile("" ::: "memory");                            \
  } while (0)

  const int nk = kt_end - kt_begin;
  // Pairs of K-tiles (stage 0 then stage 1) in a single-exit loop, odd tail afterwards: a
  // mid-loop `break` made the compiler shuttle all 64 accumulator registers between two
  // register sets every iteration (32 v_mov_b64 + MFMA-drain s_nops per pair).
  if (nk > 0) issue(0);
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    MK_V2_SYNC();
    issue(1);
    MK_V2_COMPUTE(0);
    MK_V2_SYNC();
    if (kt + 2 < nk) issue(0);
    MK_V2_COMPUTE(1);
  }
  if (kt < nk) {
    MK_V2_SYNC();
    MK_V2_COMPUTE(0);
  }
#undef MK_V2_SYNC
#undef MK_V2_COMPUTE
#undef MK_V2_MFMA4
#undef MK_V2_MFMA4_FP8
#undef MK_V2_CAT
#undef MK_V2_LOAD4
#undef MK_V2_FRAG
  if (piece >= 0) {
    // ---- stream-K tail: publish this piece's fp32 accumulators, last arriver reduces ----
    // (placement-independent agent-scope release/acquire, cdna_hip_programming.md G16)
    // Slabs are stored WRITE-THROUGH (sc1) so no L2 write-back fence is needed (the release
    // fence flushes the whole XCD L2 and made the tail slower than the quantisation it fixes);
    // every wave drains its stores, one lane bumps the tile's arrival counter, the last
    // arriver does ONE agent-scope acquire and reads the slabs.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    float* slab0 = g.ws + (long)tail_idx * g.split * (64 * 256);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(slab0 + (long)piece * (64 * 256)), 0, 64 * 256 * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int e = (i * FN + j) * 4 + q4;
          u32x4 v;
          v[0] = __float_as_uint(acc[i][j][4 * q4]); v[1] = __float_as_uint(acc[i][j][4 * q4 + 1]);
          v[2] = __float_as_uint(acc[i][j][4 * q4 + 2]); v[3] = __float_as_uint(acc[i][j][4 * q4 + 3]);
          __builtin_amdgcn_raw_buffer_store_b128(v, rsS, (e * NT + (int)threadIdx.x) * 16, 0, 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) {
      const int old = __hip_atomic_fetch_add(g.counters + tail_idx, 1, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
      *flag = (old == g.split - 1);
    }
    __syncthreads();
    if (!*flag) return;
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      g.counters[tail_idx] = 0;   // self-cleaning: the next launch on this stream finds zeros
    }
    __syncthreads();
    // deterministic: sum the slabs in piece order regardless of who arrived last
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int pc = 0; pc < g.split; ++pc) {
      const float* sl = slab0 + (long)pc * (64 * 256);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int e = (i * FN + j) * 4 + q4;
            const float4 v = *reinterpret_cast<const float4*>(sl + ((long)e * NT + threadIdx.x) * 4);
            acc[i][j][4 * q4] += v.x; acc[i][j][4 * q4 + 1] += v.y;
            acc[i][j][4 * q4 + 2] += v.z; acc[i][j][4 * q4 + 3] += v.w;
          }
    }
  }
  wave_epilogue(acc, g, C, Rp, m0, n0, wm0, wn0, smem);
}
template <bool A_RED, bool B_RED, int BKv>
__global__ __launch_bounds__(256, 2) void gemm_bf16_v2_kernel(GemmArgs g) {
  v2_body<A_RED, B_RED, BKv, 4>(g);
}
// fp8 e4m3 x fp8 e4m3 -> bf16, both operands K-major; GemmArgs dimensions are in 2-byte units
// (K / 2, lda / 2, ldb / 2): the data path is byte-identical to the bf16 kernel.
__global__ __launch_bounds__(256, 2) void gemm_fp8_v2_kernel(GemmArgs g) {
  v2_body<false, false, 64, 4, true>(g);
}

// ------------------------------------------------ skinny M (decode step) --
// y[M <= 32, N] = x W^T: one token per sample against every weight row, i.e. pure weight streaming
// (13.5 GB per generated token at 7B).  The 128-row tile kernels push W through LDS for 128 output
// rows of which <= 32 exist (measured 2.25 TB/s); here the MFMA roles are swapped -- 32 WEIGHT rows
// are the M dimension of v_mfma_f32_32x32x16_bf16, the (clamped) token rows the N dimension -- so
// a W fragment goes from HBM straight into the registers of the one wave that uses it (8 lanes of
// 16 B = a full 128-byte line per row and 64-element K block), x comes from L2, and the 8 waves of
// a workgroup split K and reduce through LDS.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs g) {
  __shared__ float red[NW][16][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, h = l >> 5, r32 = l & 31;
  const int n0 = blockIdx.x * 32;
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const bf16* wp = B + (long)min(n0 + r32, g.N - 1) * g.ldb + 8 * h;
  const bf16* xp = A + (long)min(r32, g.M - 1) * g.lda + 8 * h;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nkb = g.K / 64;
  int kb = w;
  // two K blocks per trip: 8 weight + 8 token loads of 16 B in flight per lane
  for (; kb + NW < nkb; kb += 2 * NW) {
    bf16x8 wf[8], xf[8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        wf[4 * u + j] = *reinterpret_cast<const bf16x8*>(wp + (kb + u * NW) * 64 + 16 * j);
        xf[4 * u + j] = *reinterpret_cast<const bf16x8*>(xp + (kb + u * NW) * 64 + 16 * j);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[j], acc, 0, 0, 0);
  }
  if (kb < nkb) {
    bf16x8 wf[4], xf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wf[j] = *reinterpret_cast<const bf16x8*>(wp + kb * 64 + 16 * j);
      xf[j] = *reinterpret_cast<const bf16x8*>(xp + kb * 64 + 16 * j);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], xf[j], acc, 0, 0, 0);
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) red[w][e][l] = acc[e];
  __syncthreads();
  // wave w finishes accumulator slots e = 2w' and 2w'+1 (two consecutive output columns of row m)
  const int m = r32;
  constexpr int SPW = 16 / NW;     // slots per wave (2 for NW = 8)
  float alpha = g.alpha;
  bf16* C = reinterpret_cast<bf16*>(g.C);
  const bf16* Rp = reinterpret_cast<const bf16*>(g.R);
#pragma unroll
  for (int t = 0; t < SPW; ++t) {
    const int e = w * SPW + t;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww][e][l];   // fixed order: deterministic
    const int n = n0 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (m >= g.M || n >= g.N) continue;
    v *= alpha;
    if (g.bias_mode == 1) v += (float)reinterpret_cast<const bf16*>(g.bias)[n];
    else if (g.bias_mode == 2) v += (float)reinterpret_cast<const bf16*>(g.bias)[m];
    if (g.act) v = apply_act(v, g.act);
    if (Rp) v += (float)Rp[(long)m * g.ldr + n];
    bf16* cp = C + (long)m * g.ldc + n;
    if (g.accumulate) v += (float)*cp;
    *cp = (bf16)v;
  }
}

// Same idea with 16 weight rows per workgroup (v_mfma_f32_16x16x32_bf16: 16 weight rows x 16 token
// columns x 32 k): N / 16 workgroups instead of N / 32 -- the decode GEMMs stream 33 ... 262 MB
// and last 6 ... 50 us, so what matters is that EVERY CU pulls from the first microsecond -- and a
// two-stage register pipeline: the loads of trip i + 1 are in flight under the MFMAs of trip i
// (U K-blocks of 64 per trip and wave: 2 U weight + 2 U token loads of 16 B per lane).
// PRO: what the token operand is (mk_decode_linear; 0 for mk_gemm):
//   1  RMSNorm of the A rows, fused: y = w * rnd(x * rstd) with the rounding points of
//      rmsnorm_fwd_kernel (modeling.py:100-105)
//   2  SwiGLU of A = [gate | up] ([M, 2K]): x = rnd(rnd(silu(gate)) * up) (swiglu2d_fwd_kernel,
//      modeling.py:140)
// With a prologue the workgroup first writes the M prepared token rows to LDS ([M][K + 8] bf16; M x K
// elements of work per workgroup, from L2) while its first weight fragments are already in flight,
// and the MFMA token operand is read from there (doing it per fragment in registers repeats the
// conversion for all 16 token lanes: measured 1.6x SLOWER than the separate kernels).
// MT = 2: two tiles of 16 token rows (M <= 32) share every weight fragment (PRO = 0 only).
template <int NW, int U, int PRO, int NBUF = 2, int MT = 1>
__global__ __launch_bounds__(NW * 64) void gemm_skinny16_kernel(GemmArgs g) {
  static_assert(MT == 1 || PRO == 0, "two token tiles: plain token operand only");
  extern __shared__ __attribute__((aligned(16))) char sk_smem[];
  __shared__ float red[NW][4 * MT][64];
  __shared__ float ssq[NW][16];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, r16 = l & 15, kq = l >> 4;
  const int n0 = blockIdx.x * 16;
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const bf16* wp = B + (long)min(n0 + r16, g.N - 1) * g.ldb + 8 * kq;
  const int trow = min(r16, g.M - 1);
  const bf16* xp = A + (long)trow * g.lda + 8 * kq;
  const bf16* xp2 = A + (long)min(16 + r16, g.M - 1) * g.lda + 8 * kq;   // MT == 2: token rows 16 ... 31
  const int ldt = g.K + 8;                                  // LDS token row pitch (elements)
  const bf16* tp = reinterpret_cast<const bf16*>(sk_smem) + trow * ldt + 8 * kq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  const int nkb = g.K / 64;
  constexpr int XN = PRO == 0 ? 2 * U * MT : 1;
  // trip t of wave w covers K blocks w + NW * (t * U + u), u < U (neighbouring waves read
  // neighbouring 128-byte lines of a row)
  auto load = [&](bf16x8 (&wf)[2 * U], bf16x8 (&xf)[XN], int kb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = min(kb + u * NW, nkb - 1);       // clamped: the MFMA of a clamped block is skipped
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        wf[2 * u + hh] = *reinterpret_cast<const bf16x8*>(wp + kk * 64 + 32 * hh);
        if constexpr (PRO == 0) xf[2 * u + hh] = *reinterpret_cast<const bf16x8*>(xp + kk * 64 + 32 * hh);
        if constexpr (MT == 2) xf[2 * U + 2 * u + hh] = *reinterpret_cast<const bf16x8*>(xp2 + kk * 64 + 32 * hh);
      }
    }
  };
  auto mma = [&](const bf16x8 (&wf)[2 * U], const bf16x8 (&xf)[XN], int kb) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (kb + u * NW < nkb) {
        bf16x8 t0, t1;
        if constexpr (PRO == 0) {
          t0 = xf[2 * u]; t1 = xf[2 * u + 1];
        } else {
          t0 = *reinterpret_cast<const bf16x8*>(tp + (kb + u * NW) * 64);
          t1 = *reinterpret_cast<const bf16x8*>(tp + (kb + u * NW) * 64 + 32);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u], t0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u + 1], t1, acc, 0, 0, 0);
        if constexpr (MT == 2) {
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u], xf[2 * U + 2 * u], acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2 * u + 1], xf[2 * U + 2 * u + 1], acc2, 0, 0, 0);
        }
      }
  };
  // ring of NBUF register buffers: NBUF - 1 trips of this wave are in flight under the MFMAs of one
  bf16x8 wbuf[NBUF][2 * U], xbuf[NBUF][XN];
  constexpr int STEP = NW * U;
  int kb = w;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (kb + i * STEP < nkb) load(wbuf[i], xbuf[i], kb + i * STEP);
  if constexpr (PRO != 0) {
    bf16* ts = reinterpret_cast<bf16*>(sk_smem);
    const int nch = g.K / 8;                                // 16-byte chunks per row
    for (int m = 0; m < g.M; ++m) {
      const bf16* xr = A + (long)m * g.lda;
      float rstd = 1.f;
      if constexpr (PRO == 1) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < nch; c += NW * 64) {
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xr + c * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += (float)xv[e] * (float)xv[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        __syncthreads();                                    // ssq of the previous row consumed
        if (l == 0) ssq[w][0] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += ssq[ww][0];   // fixed order: deterministic
        rstd = rsqrtf(tot / (float)g.K + g.pro_eps);
      }
      for (int c = threadIdx.x; c < nch; c += NW * 64) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(xr + c * 8);
        bf16x8 bv;
        if constexpr (PRO == 1) bv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(g.pro_w) + c * 8);
        else bv = *reinterpret_cast<const bf16x8*>(xr + g.K + c * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = (float)av[e], b = (float)bv[e];
          if constexpr (PRO == 1) o[e] = (bf16)(b * rnd<bf16>(a * rstd));
          else o[e] = (bf16)(rnd<bf16>(a / (1.f + __expf(-a))) * b);
        }
        *reinterpret_cast<bf16x8*>(ts + m * ldt + c * 8) = o;
      }
    }
    __syncthreads();
  }
  while (kb < nkb) {
#pragma unroll
    for (int i = 0; i < NBUF; ++i) {
      if (kb + (NBUF - 1) * STEP < nkb) load(wbuf[(i + NBUF - 1) % NBUF], xbuf[(i + NBUF - 1) % NBUF], kb + (NBUF - 1) * STEP);
      mma(wbuf[i], xbuf[i], kb);
      kb += STEP;
      if (kb >= nkb) break;
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[w][e][l] = acc[e];
    if constexpr (MT == 2) red[w][4 + e][l] = acc2[e];
  }
  __syncthreads();
  // D[i = weight row][j = token]: lane holds j = l & 15, i = 4 * (l >> 4) + e
  float alpha = g.alpha;
  bf16* C = reinterpret_cast<bf16*>(g.C);
  const bf16* Rp = reinterpret_cast<const bf16*>(g.R);
  for (int t = threadIdx.x; t < 256 * MT; t += NW * 64) {
    const int e = (t >> 6) & 3, ll = t & 63, mt = t >> 8;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) v += red[ww][4 * mt + e][ll];   // fixed order: deterministic
    const int m = 16 * mt + (ll & 15), n = n0 + 4 * (ll >> 4) + e;
    if (m >= g.M || n >= g.N) continue;
    v *= alpha;
    if (g.bias_mode == 1) v += (float)reinterpret_cast<const bf16*>(g.bias)[n];
    else if (g.bias_mode == 2) v += (float)reinterpret_cast<const bf16*>(g.bias)[m];
    if (g.act) v = apply_act(v, g.act);
    if (Rp) v += (float)Rp[(long)m * g.ldr + n];
    bf16* cp = C + (long)m * g.ldc + n;
    if (g.accumulate) v += (float)*cp;
    *cp = (bf16)v;
  }
}

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
// attention forward, 2 = fused attention backward (csrc/attention.hip calls mkp::begin / end).
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
    fprintf(f, "%s,%d,%d,%d,%d,%d,%d,%ld,%.4f,%.1f\n", a.kind == 0 ? "gemm" : a.kind == 1 ? "attn_fwd" : "attn_bwd",
            a.M, a.N, a.K, a.nb, a.layout, a.cfg, a.n, a.ms, a.ms > 0 ? a.fl / (a.ms * 1e-3) / 1e12 : 0.0);
  fclose(f);
  return MK_OK;
}

namespace { int g_force_cfg = -1; }
// tuning / A-B hook (scripts/gemm_bench.cpp, tests): force a kernel configuration for the
// following mk_gemm calls of this process (-1 = automatic choice); same meaning as MK_GEMM_CFG
extern "C" int mk_gemm_set_cfg(int cfg) { g_force_cfg = cfg; return MK_OK; }

namespace mkg {
int launch_v7(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool fp8);  // gemm_v7.hip
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
  if (dtype != MK_BF16 || M > (prologue ? 16 : 32) || (K % 64) || (ldx % 8) || (ldw % 8) || !aligned16(x) || !aligned16(W) ||
      (prologue == 1 && !aligned16(norm_w)))
    return MK_ERR_UNSUPPORTED;
  GemmArgs g{};
  g.A = x; g.B = W; g.C = y; g.R = residual;
  g.M = M; g.N = N; g.K = K;
  g.lda = ldx; g.ldb = ldw; g.ldc = ldy; g.ldr = ldr;
  g.alpha = 1.f;
  g.pro_w = norm_w; g.pro_eps = eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int prof = mkp::begin(st, 0, 2.0 * M * N * K, M, N, K, 1, 0, 17 + 10 * prologue);
  const dim3 g16(mk_cdiv(N, 16));
  const bool wide = N <= 16 * 256 && K <= 4096;     // as mk_gemm: 16 waves where N / 16 workgroups are few
  const size_t lds = prologue ? (size_t)M * (K + 8) * 2 : 0;   // prepared token rows
  if (lds > 40 * 1024) return MK_ERR_UNSUPPORTED;   // (two workgroups per CU must still fit)
  if (M > 16) {
    MK_LAUNCH((gemm_skinny16_kernel<8, 2, 0, 2, 2>), g16, dim3(512), 0, st, g);
  } else if (prologue == 0) {
    if (wide) MK_LAUNCH((gemm_skinny16_kernel<16, 2, 0>), g16, dim3(1024), 0, st, g);
    else MK_LAUNCH((gemm_skinny16_kernel<8, 2, 0, 3>), g16, dim3(512), 0, st, g);
  } else if (prologue == 1) {
    if (wide) MK_LAUNCH((gemm_skinny16_kernel<16, 2, 1>), g16, dim3(1024), lds, st, g);
    else MK_LAUNCH((gemm_skinny16_kernel<8, 2, 1, 3>), g16, dim3(512), lds, st, g);
  } else {
    if (wide) MK_LAUNCH((gemm_skinny16_kernel<16, 2, 2>), g16, dim3(1024), lds, st, g);
    else MK_LAUNCH((gemm_skinny16_kernel<8, 2, 2, 3>), g16, dim3(512), lds, st, g);
  }
  mkp::end(prof, st);
  return mk_check_launch();
}
namespace {
// Default kernel choice: a small cost model fitted to scripts/gemm_bench.cpp measurements on
// MI355X (profiles/r02_gemm_shapes.csv; microseconds).  v7: one 256x256 tile per CU and round,
// (K-tiles x a7 + prologue/epilogue) per round, the last partial round as 128x128 sub-tiles when
// that is at most two sub-rounds.  v2: 128x128 tiles, two workgroups per CU (four with BK = 32 for
// reduction-major x reduction-major), fractional rounds through its K-split tail.
int pick_cfg(const mk_gemm_desc* d, int nbatch, bool v7_ok, int n_cus) {
  static const bool no_v7 = getenv("MK_GEMM_NO_V7") != nullptr;
  if (!v7_ok || no_v7 || nbatch != 1) return MK_GEMM_DEFAULT_CFG;
  const int layout = (d->a_red_major ? 2 : 0) + (d->b_red_major ? 1 : 0);
  const double nk = (d->K + 63) / 64;
  const long T7 = (long)mk_cdiv(d->M, 256) * mk_cdiv(d->N, 256);
  const long full = T7 / n_cus, R = T7 % n_cus;
  const double a7 = layout == 3 ? 1.63 : layout == 1 ? 1.53 : layout == 2 ? 1.57 : 1.49;
  const double tile7 = nk * a7 + 14.0, sub7 = nk * 0.48 + 7.0;
  double t7 = full * tile7;
  // (a partially filled round of whole tiles runs faster per K-tile: less L2 / power contention)
  if (R > 0) t7 += (4 * R <= 2 * n_cus) ? (double)mk_cdiv((int)(4 * R), n_cus) * sub7
                                         : nk * a7 * (0.55 + 0.45 * R / n_cus) + 14.0;
  const long T2 = (long)mk_cdiv(d->M, 128) * mk_cdiv(d->N, 128);
  const long slots2 = (long)n_cus * (layout == 3 ? 4 : 2);
  const double tile2 = layout == 3 ? nk * 1.77 + 8.0 : nk * (layout == 0 ? 0.92 : 1.03) + 5.0;
  double rounds2 = (double)T2 / slots2;
  if (rounds2 < 0.25) rounds2 = 0.25;
  double t2 = rounds2 * tile2;
  if (const long R2 = T2 % slots2) {   // K-split tail: pieces + the last arriver reading `sp` slabs
    double sp = (double)slots2 / R2;
    const double nk2 = layout == 3 ? 2 * nk : nk;
    if (sp > nk2 / 2) sp = nk2 / 2;
    if (sp > 64) sp = 64;
    t2 += 4.0 + 0.65 * sp;
  }
  return t7 * 0.95 < t2 ? 11 : MK_GEMM_DEFAULT_CFG;   // (ties go to v7: the model is pessimistic for it)
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
  if (d->dtype != MK_F32 && d->dtype != MK_BF16) return MK_ERR_UNSUPPORTED;
  GemmArgs g;
  g.scale_a = d->scale_a; g.scale_b = d->scale_b;
  g.scale_vec = (d->flags & MK_GEMM_SCALE_VEC) ? 1 : 0;
  if (g.scale_vec && (!d->scale_a || !d->scale_b)) return MK_ERR_BAD_ARG;
  g.A = d->A; g.B = d->B; g.C = d->C; g.R = d->R; g.bias = d->bias;
  g.M = d->M; g.N = d->N; g.K = d->K;
  g.lda = d->lda; g.ldb = d->ldb; g.ldc = d->ldc; g.ldr = d->ldr;
  g.nb2 = d->nb2;
  g.sA1 = d->sA1; g.sA2 = d->sA2; g.sB1 = d->sB1; g.sB2 = d->sB2;
  g.sC1 = d->sC1; g.sC2 = d->sC2; g.sR1 = d->sR1; g.sR2 = d->sR2;
  g.alpha = d->alpha; g.bias_mode = d->bias_mode; g.act = d->act; g.accumulate = d->accumulate;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nbatch = d->nb1 * d->nb2;
  const int prof = mkp::begin(st, 0, 2.0 * d->M * d->N * d->K * nbatch * (fp8 ? 2 : 1), d->M, d->N,
                              d->K * (fp8 ? 2 : 1), nbatch, d->a_red_major * 2 + d->b_red_major, -1);
  // (measured on generate(): B = 1: 7.5 -> 6.1 ms/token, B = 8: 7.2 -> 6.5; at B = 32 the 32 distinct
  // token rows re-read per workgroup cost more than the tile kernel's wasted rows: 8.6 vs 8.0)
  static const int skinny_max = [] { const char* e = getenv("MK_GEMM_SKINNY_MAX_M"); return e ? atoi(e) : 32; }();
  if (d->dtype == MK_BF16 && !fp8 && d->M <= 32 && d->M <= skinny_max && !d->a_red_major && !d->b_red_major && nbatch == 1 &&
      (d->K % 64) == 0 && (d->lda % 8) == 0 && (d->ldb % 8) == 0 && aligned16(d->A) && aligned16(d->B) &&
      !getenv("MK_GEMM_NO_SKINNY")) {
    // cfg 12 = 32 weight rows per workgroup (8 waves), 17 / 18 = 16 rows per workgroup with 8 / 16
    // waves splitting K (measured cold, scripts/gemm_shapes_decode.txt: 4.0 ... 5.7 TB/s against
    // 2.1 ... 3.8; 16 waves where N / 16 workgroups alone would leave a CU with one short wave set)
    int sk = d->M <= 16 ? ((d->N <= 16 * 256 && d->K <= 4096) ? 18 : 17) : 19;
    if (g_force_cfg >= 12 && g_force_cfg <= 19 && (g_force_cfg == 12 || g_force_cfg == 19 || d->M <= 16)) sk = g_force_cfg;
    mkp::set_cfg(prof, sk);
    const dim3 g16(mk_cdiv(d->N, 16));
    if (sk == 19) MK_LAUNCH((gemm_skinny16_kernel<8, 2, 0, 2, 2>), g16, dim3(512), 0, st, g);   // 17 ... 32 token rows
    else if (sk == 17) MK_LAUNCH((gemm_skinny16_kernel<8, 2, 0, 3>), g16, dim3(512), 0, st, g);
    else if (sk == 18) MK_LAUNCH((gemm_skinny16_kernel<16, 2, 0>), g16, dim3(1024), 0, st, g);
    else if (sk == 13) MK_LAUNCH((gemm_skinny16_kernel<8, 2, 0, 2>), g16, dim3(512), 0, st, g);
    else MK_LAUNCH((gemm_skinny_kernel<8>), dim3(mk_cdiv(d->N, 32)), dim3(512), 0, st, g);
    mkp::end(prof, st);
    return mk_check_launch();
  }
  if (d->dtype == MK_BF16) {
    // kernel configuration: 11 = v7 256x256 quadrant-phase LDS-DMA ring (gemm_v7.hip; big aligned
    // problems), 5 = v2 issue-lean LDS-DMA 128x128 (aligned operands, K % 64 == 0), 7 = v2 with
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
    static const int n_cus = [] {
      int dev = 0, cus = 256;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      return cus;
    }();
    int cfg;
    if (env_cfg >= 0) cfg = env_cfg;
    else cfg = pick_cfg(d, nbatch, v7_ok, n_cus);
    if (fp8 && cfg != 11) cfg = 5;        // fp8 exists on the two LDS-DMA tile kernels only
    if (cfg == 11 && !v7_ok) cfg = 5;
    if (cfg != 0 && cfg != 5 && cfg != 7 && cfg != 11) cfg = 5;
    if (cfg >= 5 && !v2_ok) cfg = 0;
    if (fp8 && cfg != 5 && cfg != 11) return MK_ERR_UNSUPPORTED;
    // Measured (profiles/): with BOTH operands reduction-major (dW = dy^T x) the global rows are
    // whole 256-B lines whatever BK is, and BK = 32 (32 KiB LDS -> 4 workgroups per CU) is 17 %
    // faster than BK = 64 on the 128x128 tile; K-major operands would degrade to 64-B segments.
    if (cfg == 5 && d->a_red_major && d->b_red_major && !getenv("MK_GEMM_NO_BK32")) cfg = 7;
    const int bkv = cfg == 7 ? 32 : BK;
    mkp::set_cfg(prof, cfg);
    const bool t256 = cfg == 11;
    const int bm = t256 ? 256 : 128;
    const int bn = t256 ? 256 : BN;
    g.tiles_m = mk_cdiv(d->M, bm);
    g.tiles_n = mk_cdiv(d->N, bn);
    g.a_vec = aligned16(d->A) && (d->lda % 8 == 0) && (d->sA1 % 8 == 0) && (d->sA2 % 8 == 0);
    g.b_vec = aligned16(d->B) && (d->ldb % 8 == 0) && (d->sB1 % 8 == 0) && (d->sB2 % 8 == 0);
    g.c_vec = ((reinterpret_cast<uintptr_t>(d->C) & 7) == 0) && (d->ldc % 4 == 0) &&
              (d->sC1 % 4 == 0) && (d->sC2 % 4 == 0) &&
              (!d->R || (((reinterpret_cast<uintptr_t>(d->R) & 7) == 0) && (d->ldr % 4 == 0) &&
                         (d->sR1 % 4 == 0) && (d->sR2 % 4 == 0)));
    dim3 grid(g.tiles_m * g.tiles_n, 1, nbatch);
    g.dp_tiles = g.tiles_m * g.tiles_n;
    g.lin_batch = 0;
    g.split = 1;
    g.kt_per_piece = 0;
    g.ws = nullptr;
    g.counters = nullptr;
    g.ablate = 0;
    // resident workgroups per CU of the chosen kernel
    const int slots = n_cus * (t256 ? 1 : (cfg == 7 ? 4 : 2));
    static const bool no_streamk = getenv("MK_GEMM_NO_STREAMK") != nullptr;
    if (t256 && !no_streamk) {
      // v7: the last partial round of 256x256 tiles is computed as four 128x128 sub-tiles each
      // (one workgroup per sub-tile, full K, no partial sums) when that takes fewer rounds
      const int T = g.tiles_m * g.tiles_n, R = T % n_cus;
      if (R > 0 && 4 * R <= 2 * n_cus) {
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
#define MK_REG(AR, BR)                                                                        \
  do {                                                                                        \
    static bool attr_done = false;                                                            \
    if (!attr_done) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<AR, BR, false>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);  \
      attr_done = true;                                                                       \
    }                                                                                         \
    MK_LAUNCH((gemm_bf16_kernel<AR, BR, false>), grid, dim3(256), 4 * TILE_BYTES, st, g);     \
  } while (0)
#define MK_V2(AR, BR)                                                                         \
  do {                                                                                        \
    static bool attr_done = false;                                                            \
    if (!attr_done) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_v2_kernel<AR, BR, 64>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);  \
      attr_done = true;                                                                       \
    }                                                                                         \
    MK_LAUNCH((gemm_bf16_v2_kernel<AR, BR, 64>), grid, dim3(256), 4 * TILE_BYTES, st, g);     \
  } while (0)
#define MK_V2F8()                                                                             \
  do {                                                                                        \
    static bool attr_done = false;                                                            \
    if (!attr_done) {                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fp8_v2_kernel),           \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);  \
      attr_done = true;                                                                       \
    }                                                                                         \
    MK_LAUNCH(gemm_fp8_v2_kernel, grid, dim3(256), 4 * TILE_BYTES, st, g);                    \
  } while (0)
#define MK_V2S(AR, BR)                                                                        \
  MK_LAUNCH((gemm_bf16_v2_kernel<AR, BR, 32>), grid, dim3(256), 2 * TILE_BYTES, st, g)
#define MK_LAYOUT(AR, BR)                                    \
  do {                                                       \
    if (cfg == 7) MK_V2S(AR, BR);                            \
    else if (cfg == 5) MK_V2(AR, BR);                        \
    else MK_REG(AR, BR);                                     \
  } while (0)
    if (t256) {
      const int rc = mkg::launch_v7(g, d->a_red_major != 0, d->b_red_major != 0, grid, st, fp8);
      mkp::end(prof, st);
      return rc;
    }
    if (fp8) MK_V2F8();
    else if (!d->a_red_major && !d->b_red_major) MK_LAYOUT(false, false);
    else if (!d->a_red_major && d->b_red_major) MK_LAYOUT(false, true);
    else if (d->a_red_major && !d->b_red_major) MK_LAYOUT(true, false);
    else MK_LAYOUT(true, true);
#undef MK_LAYOUT
#undef MK_V2
#undef MK_V2S
#undef MK_V2F8
#undef MK_REG
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
