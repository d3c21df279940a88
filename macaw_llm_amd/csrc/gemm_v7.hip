// v7: the 256 x 256 x 64 bf16 MFMA GEMM of the big projections (gfx950 / MI355X).
//
// Why another kernel: the 128 x 128 v2 kernel pulls 64 KiB per K-tile and workgroup pair through
// the L2 -> LDS-DMA path for 1024 MFMA cycles per SIMD and is bound by that fill rate and by its
// two barriers per K-tile (profiles/r01_gemm_pmc.md: MFMA pipe 41-49 % busy).  A 256 x 256 tile
// halves the bytes per FLOP; what it needs in exchange is a schedule in which no wave ever waits
// for a load it has just issued:
//
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 fragments of v_mfma_f32_32x32x16_bf16
//     (128 accumulator registers), one workgroup per CU, two waves per SIMD.
//   * A K-tile is computed BY OUTPUT QUADRANT (rows lo/hi x columns lo/hi of the wave tile, full
//     K = 64 each), not by k-step, in TWO PHASES of 16 MFMAs:  A: (A_lo,B_lo) (A_lo,B_hi)   B:
//     (A_hi,B_hi) (A_hi,B_lo).  LDS reads per phase: 16 / 8 x 16 B per lane.  The B operand of a
//     K-tile is therefore dead after phase A and A after the reads of phase B, which is what lets
//     the NEXT-BUT-ONE tile's data be requested while this tile is still being computed:
//   * LDS = ten 16-KiB half-tile slots (all 160 KiB): A has 3 tile slots, B has 2.  A half-tile
//     is 128 rows x 64 k = sixteen 1-KiB `buffer_load_dwordx4 ... lds` pieces, two per wave.
//     Phase A requests both halves of A(T+2) (into the slot A(T-1) left in phase B(T-1)), phase B
//     both halves of B(T+2) (into the slot B(T) left in phase A(T)).  Reads retire (lgkmcnt 0)
//     before the barrier that closes the segment issuing them, so every slot is rewritten at least
//     one full phase after its last read; every request has two phases (one K-tile) to land, and
//     the only wait in the loop is ONE counted `s_waitcnt vmcnt(8)` per K-tile (phase B:
//     everything older than A(T+2), B(T+2) has landed = tile T+1 complete); vmcnt never drains to
//     0 inside the loop and the barriers are raw `s_barrier`s, so the LDS-DMA queue stays full
//     across them (cdna_hip_programming.md "8-phase template", T3+T4, "Pipelining across
//     barriers").  Four phases of 8 MFMAs (the template's granularity) measured 3 % slower: each
//     barrier interval costs ~30 cycles on top of its 256 / 512 MFMA cycles.
//   * The two M-halves of the workgroup (waves 0-3 / 4-7 = one wave of each per SIMD) run one
//     barrier out of phase: while one half issues its ds_reads and LDS-DMA, the other half owns
//     the matrix pipe (s_setprio around the MFMA cluster).
//   * LDS images are those of the v2 kernel per half-tile: K-major [128][64] with the 16-B chunk
//     XOR (row>>1)&7 (conflict-free ds_read_b128), reduction-major [64 k][128 rows] read through
//     ds_read_b64_tr_b16; the swizzle is applied to the per-lane SOURCE address of the DMA.
//   * Epilogue (alpha, bias, activation, residual, accumulate) = mkg::wave_epilogue, transposed
//     through LDS so that every C / R access is a full 128-byte line.
//   * Tile-count quantisation on the 256-CU chip: the last partial round of tiles is NOT cut
//     along K (measured: each K-piece writes a 256-KiB fp32 slab at the CU's ~10 B/clk store rate
//     and the last arriver re-reads 8 of them at ~100 GB/s -- 4608x4096x4096 ran at 800 TFLOP/s)
//     but SPATIALLY into four 128 x 128 sub-tiles over the full K, one workgroup each, computed
//     by a small 5-stage LDS-DMA loop below (v7_subtile): no partial sums, no workspace, no
//     inter-workgroup traffic, bit-identical to an unsplit tile.
//
// Requires 16-byte aligned operands / pitches; a K tail (K % 64 != 0) only with operands that
// read as zeros beyond K (reduction-major: outside the descriptor; K-major: zero pad columns).
// Replaces the nn.Linear matmuls of modeling.py:134-140,159-162,597 and their gradients.
#include "gemm_common.h"

namespace {
using namespace mkg;

constexpr int BM7 = 256, BN7 = 256, BK7 = 64;
constexpr int HALF_BYTES = 128 * BK7 * 2;        // 16 KiB: 128 rows x 64 k
constexpr int A_SLOT = 2 * HALF_BYTES;           // 32 KiB
constexpr int B_BASE7 = 3 * A_SLOT;              // A: 3 slots, then B: 2 slots
constexpr int LDS7 = 5 * A_SLOT;                 // 163,840 B

// per-lane byte offset (relative to the tile's first row / column) of this lane's 16 bytes in
// piece p (0..15) of half-tile `half`; the XOR puts the swizzled LDS image behind a LINEAR
// LDS-DMA destination (rule 21: swizzle the source, read with the same involution)
template <bool RED_MAJOR>
MK_DEV int v7_voffset(int row0, int R, long ld, int half, int p, int l) {
  if constexpr (!RED_MAJOR) {
    const int r = p * 8 + (l >> 3);                       // row inside the half image
    const int kc = (l & 7) ^ ((r >> 1) & 7);
    const int gr = min(row0 + half * 128 + r, R - 1) - row0;  // clamp: garbage rows are never stored
    return (int)((long)gr * ld * 2 + kc * 16);
  } else {
    const int kr = p * 4 + (l >> 4);
    const int mc = (l & 15) ^ (4 * (kr & 3));
    return (int)((long)kr * ld * 2 + (half * 128 + mc * 8) * 2);
  }
}

// LDS read addressing of one operand.  K-major: one base per k-step (the swizzle depends on it),
// fragments of 32 rows are +4096 apart.  Reduction-major: one base per 32-row fragment (its
// swizzle depends on the column), k-steps are +4096 apart, the second transpose read +1024.
template <bool RED_MAJOR>
MK_DEV void v7_read_offsets(int base_row, int l, int (&off)[4]) {
  if constexpr (!RED_MAJOR) {
    const int row = base_row + (l & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kc = ks * 2 + (l >> 5);
      off[ks] = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    }
  } else {
    const int li = l & 15;
    const int kr = 8 * (l >> 5) + (li >> 2);
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int col = base_row + f * 32 + 16 * ((l >> 4) & 1) + 4 * (li & 3);
      off[f] = kr * 256 + (((col >> 3) ^ (4 * (kr & 3))) << 4) + ((col & 7) << 1);
    }
  }
}

// fragment F (32 rows) of k-step KS; ADDR = LDS byte addresses (see v7_read_offsets)
#define V7_FRAG(RED, ADDR, F, KS)                                                                 \
  [&]() -> bf16x8 {                                                                               \
    if constexpr (!(RED)) {                                                                       \
      return *reinterpret_cast<const bf16x8*>(smem + ADDR[KS] + (F) * 4096);                      \
    } else {                                                                                      \
      bf16x8 o_;                                                                                  \
      bf16x4 t0_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                      \
          (__attribute__((address_space(3))) bf16x4*)(smem + ADDR[F] + (KS) * 4096));            \
      bf16x4 t1_ = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                      \
          (__attribute__((address_space(3))) bf16x4*)(smem + ADDR[F] + (KS) * 4096 + 1024));     \
      o_[0] = t0_[0]; o_[1] = t0_[1]; o_[2] = t0_[2]; o_[3] = t0_[3];                             \
      o_[4] = t1_[0]; o_[5] = t1_[1]; o_[6] = t1_[2]; o_[7] = t1_[3];                             \
      return o_;                                                                                  \
    }                                                                                             \
  }()

// ---- spatial tail: one 128 x 128 sub-tile (quadrant `sub & 3` of tail tile `sub >> 2`) over the
// full K.  8 waves = 2 (M) x 4 (N) of 64 x 32; a stage = one A and one B half-tile image (32 KiB),
// five stages, LDS-DMA four K-tiles ahead, one barrier and one counted vmcnt per K-tile.
template <bool A_RED, bool B_RED, bool FP8 = false>
MK_DEV void v7_subtile(const GemmArgs& g, int sub) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 2 * HALF_BYTES;
  int tm, tn;
  tile_from_index(g.dp_tiles + (sub >> 2), g.tiles_m, g.tiles_n, tm, tn, 8);
  const int m0 = tm * BM7 + ((sub >> 1) & 1) * 128, n0 = tn * BN7 + (sub & 1) * 128;
  if (m0 >= g.M || n0 >= g.N) return;     // quadrant outside the matrix (workgroup-uniform)
  const int z = blockIdx.z, z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const bf16* A = reinterpret_cast<const bf16*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16* B = reinterpret_cast<const bf16*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  bf16* C = reinterpret_cast<bf16*>(g.C) + z1 * g.sC1 + z2 * g.sC2;
  const bf16* Rp = g.R ? reinterpret_cast<const bf16*>(g.R) + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm0 = (w >> 2) * 64, wn0 = (w & 3) * 32;
  const bf16* abase = uniform_ptr(A_RED ? A + m0 : A + (long)m0 * g.lda);
  const bf16* bbase = uniform_ptr(B_RED ? B + n0 : B + (long)n0 * g.ldb);
  const long a_bytes = A_RED ? ((long)(g.K - 1) * g.lda + ((g.M - m0 + 1) & ~1)) * 2
                             : ((long)(min(g.M - m0, 128) - 1) * g.lda + ((g.K + 1) & ~1)) * 2;
  const long b_bytes = B_RED ? ((long)(g.K - 1) * g.ldb + ((g.N - n0 + 1) & ~1)) * 2
                             : ((long)(min(g.N - n0, 128) - 1) * g.ldb + ((g.K + 1) & ~1)) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)abase, 0, (int)min(a_bytes, 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bbase, 0, (int)min(b_bytes, 0x7fffffffL), 0x00020000);
  int voA[2], voB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    voA[i] = v7_voffset<A_RED>(m0, g.M, g.lda, 0, w + 8 * i, l);
    voB[i] = v7_voffset<B_RED>(n0, g.N, g.ldb, 0, w + 8 * i, l);
  }
  const int stepA = A_RED ? (int)(BK7 * g.lda * 2) : BK7 * 2;
  const int stepB = B_RED ? (int)(BK7 * g.ldb * 2) : BK7 * 2;
  int offA[4], offB[4];
  v7_read_offsets<A_RED>(wm0, l, offA);
  v7_read_offsets<B_RED>(wn0, l, offB);
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
  const int nk = (g.K + BK7 - 1) / BK7;
  int kA = 0, kB = 0, st_in = 0;   // next K-tile to request and its stage
  auto issue = [&]() {
    char* dst = smem + st_in * STAGE + w * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsA, (__attribute__((address_space(3))) void*)(dst + i * 8192), 16, voA[i], kA, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (__attribute__((address_space(3))) void*)(dst + HALF_BYTES + i * 8192), 16, voB[i], kB, 0, 0);
    kA += stepA; kB += stepB;
    st_in = st_in == 4 ? 0 : st_in + 1;
  };
#pragma unroll 1
  for (int s = 0; s < 4 && s < nk; ++s) issue();
  // fragments of K-tile t + 1 are read (second register set) before the MFMAs of K-tile t
  bf16x8 fa0[2][4], fb0[4], fa1[2][4], fb1[4];
#define V7S_WAIT(AHEAD)                                                        \
  do {                                                                         \
    if ((AHEAD) >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         \
    else if ((AHEAD) == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
    __builtin_amdgcn_sched_barrier(0);                                         \
    __builtin_amdgcn_s_barrier();                                              \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define V7S_READ(FA, FB, ST)                                                   \
  do {                                                                         \
    int aad[4], bad[4];                                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                            \
      aad[i] = (ST) * STAGE + offA[i];                                         \
      bad[i] = (ST) * STAGE + HALF_BYTES + offB[i];                            \
    }                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                         \
      FB[ks] = V7_FRAG(B_RED, bad, 0, ks);                                     \
      FA[0][ks] = V7_FRAG(A_RED, aad, 0, ks);                                  \
      FA[1][ks] = V7_FRAG(A_RED, aad, 1, ks);                                  \
    }                                                                          \
  } while (0)
#define V7S_CAT(LO, HI)                                                        \
  [&]() -> i32x8 {                                                             \
    const i32x4 lo_ = __builtin_bit_cast(i32x4, LO), hi_ = __builtin_bit_cast(i32x4, HI); \
    i32x8 r_;                                                                  \
    r_[0] = lo_[0]; r_[1] = lo_[1]; r_[2] = lo_[2]; r_[3] = lo_[3];            \
    r_[4] = hi_[0]; r_[5] = hi_[1]; r_[6] = hi_[2]; r_[7] = hi_[3];            \
    return r_;                                                                 \
  }()
#define V7S_MMA(FA, FB)                                                        \
  do {                                                                         \
    if constexpr (FP8) {                                                       \
      _Pragma("unroll") for (int kp = 0; kp < 2; ++kp) {                       \
        const i32x8 b8_ = V7S_CAT(FB[2 * kp], FB[2 * kp + 1]);                 \
        acc[0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(           \
            b8_, V7S_CAT(FA[0][2 * kp], FA[0][2 * kp + 1]), acc[0][0], 0, 0, 0, 127, 0, 127); \
        acc[1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(           \
            b8_, V7S_CAT(FA[1][2 * kp], FA[1][2 * kp + 1]), acc[1][0], 0, 0, 0, 127, 0, 127); \
      }                                                                        \
    } else {                                                                   \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                       \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks], FA[0][ks], acc[0][0], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks], FA[1][ks], acc[1][0], 0, 0, 0); \
      }                                                                        \
    }                                                                          \
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]));                       \
  } while (0)
// one K-tile t: stage t + 1 must have landed (requests t + 2, t + 3 may stay in flight); after the
// barrier every wave has finished its reads of stage t - 1, whose slot request t + 4 reuses
#define V7S_STEP(FX, BX, FY, BY)                                               \
  do {                                                                         \
    if (t + 1 < nk) {                                                          \
      V7S_WAIT(nk - 2 - t);                                                    \
      V7S_READ(FY, BY, st1);                                                   \
    }                                                                          \
    V7S_MMA(FX, BX);                                                           \
    if (t + 4 < nk) issue();                                                   \
    st1 = st1 == 4 ? 0 : st1 + 1;                                              \
    ++t;                                                                       \
  } while (0)
  V7S_WAIT(nk - 1);
  V7S_READ(fa0, fb0, 0);
  int st1 = 1, t = 0;
  while (t < nk) {
    V7S_STEP(fa0, fb0, fa1, fb1);
    if (t >= nk) break;
    V7S_STEP(fa1, fb1, fa0, fb0);
  }
#undef V7S_STEP
#undef V7S_MMA
#undef V7S_CAT
#undef V7S_READ
#undef V7S_WAIT
  wave_epilogue(acc, g, C, Rp, m0, n0, wm0, wn0, smem);
}

template <bool A_RED, bool B_RED, bool FP8 = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_v7_kernel(GemmArgs g) {
  static_assert(!FP8 || (!A_RED && !B_RED), "fp8 operands are K-major");
  constexpr int FM = 4, FN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int tm, tn;
  const int kt_begin = 0, kt_end = (g.K + BK7 - 1) / BK7;   // K tail: see mk_gemm (zeros)
  {
    const int bid = blockIdx.x;
    if (bid >= g.dp_tiles) {          // spatial tail: quadrant (bid' & 3) of tail tile (bid' >> 2)
      v7_subtile<A_RED, B_RED, FP8>(g, xcd_remap(bid - g.dp_tiles, (int)gridDim.x - g.dp_tiles));
      return;
    }
    tile_from_index(xcd_remap(bid, g.dp_tiles), g.tiles_m, g.tiles_n, tm, tn, 8);
  }
  const int z = blockIdx.z, z1 = z / g.nb2, z2 = z - z1 * g.nb2;
  const bf16* A = reinterpret_cast<const bf16*>(g.A) + z1 * g.sA1 + z2 * g.sA2;
  const bf16* B = reinterpret_cast<const bf16*>(g.B) + z1 * g.sB1 + z2 * g.sB2;
  bf16* C = reinterpret_cast<bf16*>(g.C) + z1 * g.sC1 + z2 * g.sC2;
  const bf16* Rp = g.R ? reinterpret_cast<const bf16*>(g.R) + z1 * g.sR1 + z2 * g.sR2 : nullptr;
  const int m0 = tm * BM7, n0 = tn * BN7;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 2, wc = w & 3;      // wave row (= stagger group) / wave column
  const int wm0 = wr * 128, wn0 = wc * 64;

  // buffer descriptors over the tile's rows; num_records bounds the over-read of edge tiles
  const bf16* abase = uniform_ptr(A_RED ? A + m0 : A + (long)m0 * g.lda);
  const bf16* bbase = uniform_ptr(B_RED ? B + n0 : B + (long)n0 * g.ldb);
  const long a_bytes = A_RED ? ((long)(g.K - 1) * g.lda + ((g.M - m0 + 1) & ~1)) * 2
                             : ((long)(min(g.M - m0, BM7) - 1) * g.lda + ((g.K + 1) & ~1)) * 2;
  const long b_bytes = B_RED ? ((long)(g.K - 1) * g.ldb + ((g.N - n0 + 1) & ~1)) * 2
                             : ((long)(min(g.N - n0, BN7) - 1) * g.ldb + ((g.K + 1) & ~1)) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)abase, 0, (int)min(a_bytes, 0x7fffffffL), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)bbase, 0, (int)min(b_bytes, 0x7fffffffL), 0x00020000);
  // this wave's two pieces (w and w + 8) of each half-tile
  int voA[2][2], voB[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      voA[h][i] = v7_voffset<A_RED>(m0, g.M, g.lda, h, w + 8 * i, l);
      voB[h][i] = v7_voffset<B_RED>(n0, g.N, g.ldb, h, w + 8 * i, l);
    }
  const int stepA = A_RED ? (int)(BK7 * g.lda * 2) : BK7 * 2;   // bytes per K-tile
  const int stepB = B_RED ? (int)(BK7 * g.ldb * 2) : BK7 * 2;
  int offA[4], offB[4];
  v7_read_offsets<A_RED>(0, l, offA);                 // rows of this wave's A half image
  v7_read_offsets<B_RED>((wc & 1) * 64, l, offB);     // columns inside this wave's B half image

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = kt_end - kt_begin;
  // LDS-DMA of this wave's share of one half-tile: slot byte offset `dst`, K-tile byte offset `ko`
  auto dma_a = [&](int dst, int half, int ko) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsA, (__attribute__((address_space(3))) void*)(smem + dst + half * HALF_BYTES + (w + 8 * i) * 1024),
          16, voA[half][i], ko, 0, 0);
  };
  auto dma_b = [&](int dst, int half, int ko) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rsB, (__attribute__((address_space(3))) void*)(smem + dst + half * HALF_BYTES + (w + 8 * i) * 1024),
          16, voB[half][i], ko, 0, 0);
  };

#define V7_BAR()                               \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)
// 8 MFMAs: rows I0, I0+1 (fragments held in fa_) x column fragment J (held in FB)
// fp8 (e4m3) operands, FP8 = true: the tile bytes, the LDS image and the 16-byte fragment reads are
// those of the bf16 kernel (GemmArgs in 2-byte units, K-major operands only); two consecutive reads
// are concatenated into the 32-byte operand of ONE v_mfma_scale_f32_32x32x64_f8f6f4 (64 fp8 k-slots,
// scales 2^0: the de-quantisation scales are applied in the epilogue; any k-slot permutation is
// legal because A and B use the same one).  Same instruction-issue time per K-tile, twice the K.
#define V7_CAT(LO, HI)                                                                            \
  [&]() -> i32x8 {                                                                                \
    const i32x4 lo_ = __builtin_bit_cast(i32x4, LO), hi_ = __builtin_bit_cast(i32x4, HI);         \
    i32x8 r_;                                                                                     \
    r_[0] = lo_[0]; r_[1] = lo_[1]; r_[2] = lo_[2]; r_[3] = lo_[3];                               \
    r_[4] = hi_[0]; r_[5] = hi_[1]; r_[6] = hi_[2]; r_[7] = hi_[3];                               \
    return r_;                                                                                    \
  }()
#define V7_MMA(I0, J, FB)                                                                         \
  do {                                                                                            \
    if constexpr (FP8) {                                                                          \
      _Pragma("unroll") for (int kp_ = 0; kp_ < 2; ++kp_) {                                       \
        const i32x8 b8_ = V7_CAT(FB[2 * kp_], FB[2 * kp_ + 1]);                                   \
        acc[I0][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                             \
            b8_, V7_CAT(fa_[0][2 * kp_], fa_[0][2 * kp_ + 1]), acc[I0][J], 0, 0, 0, 127, 0, 127); \
        acc[I0 + 1][J] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                         \
            b8_, V7_CAT(fa_[1][2 * kp_], fa_[1][2 * kp_ + 1]), acc[I0 + 1][J], 0, 0, 0, 127, 0, 127); \
      }                                                                                           \
    } else {                                                                                      \
      _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                       \
        acc[I0][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks_], fa_[0][ks_], acc[I0][J], 0, 0, 0); \
        acc[I0 + 1][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FB[ks_], fa_[1][ks_], acc[I0 + 1][J], 0, 0, 0); \
      }                                                                                           \
    }                                                                                             \
    /* register-only MFMAs are otherwise sunk / hoisted across the barriers (s5.7 item 3) */      \
    asm volatile("" : "+v"(acc[I0][J]), "+v"(acc[I0 + 1][J]));                                    \
  } while (0)
// One K-tile.  FULL: tile T+2 exists (steady state); NEXT: tile T+1 exists (wave-uniform).
#define V7_TILE(FULL, NEXT)                                                                      \
  do {                                                                                            \
    bf16x8 fa_[2][4], fbl_[4], fbh_[4];                                                           \
    int aad_[4], bad_[4];                                                                         \
    const int a2_ = aslot >= 1 ? aslot - 1 : 2;                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                            \
      aad_[i_] = aslot * A_SLOT + wr * HALF_BYTES + offA[i_];                                     \
      bad_[i_] = B_BASE7 + bslot * A_SLOT + (wc >> 1) * HALF_BYTES + offB[i_];                    \
    }                                                                                             \
    /* ---- phase A ---- */                                                                       \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) fbl_[ks_] = V7_FRAG(B_RED, bad_, 0, ks_); \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                         \
      fa_[0][ks_] = V7_FRAG(A_RED, aad_, 0, ks_);                                                 \
      fa_[1][ks_] = V7_FRAG(A_RED, aad_, 1, ks_);                                                 \
    }                                                                                             \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) fbh_[ks_] = V7_FRAG(B_RED, bad_, 1, ks_); \
    if (FULL) { dma_a(a2_ * A_SLOT, 0, kA2); dma_a(a2_ * A_SLOT, 1, kA2); }                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    V7_BAR();                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                \
    V7_MMA(0, 0, fbl_);                                                                           \
    V7_MMA(0, 1, fbh_);                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                \
    V7_BAR();                                                                                     \
    /* ---- phase B ---- */                                                                       \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                         \
      fa_[0][ks_] = V7_FRAG(A_RED, aad_, 2, ks_);                                                 \
      fa_[1][ks_] = V7_FRAG(A_RED, aad_, 3, ks_);                                                 \
    }                                                                                             \
    if (FULL) {                                                                                   \
      dma_b(B_BASE7 + bslot * A_SLOT, 0, kB2); dma_b(B_BASE7 + bslot * A_SLOT, 1, kB2);           \
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                            \
    } else if (NEXT) {                                                                            \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    }                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    V7_BAR();                                                                                     \
    asm volatile("" : "+v"(fbl_[0]), "+v"(fbl_[1]), "+v"(fbl_[2]), "+v"(fbl_[3]));                \
    __builtin_amdgcn_s_setprio(1);                                                                \
    V7_MMA(2, 1, fbh_);                                                                           \
    V7_MMA(2, 0, fbl_);                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                \
    V7_BAR();                                                                                     \
    aslot = aslot == 2 ? 0 : aslot + 1;                                                           \
    bslot ^= 1;                                                                                   \
    kA2 += stepA; kB2 += stepB;                                                                   \
  } while (0)

  // ---- prologue: tile 0 complete, tile 1 requested
  int kA2 = (kt_begin + 2) * stepA, kB2 = (kt_begin + 2) * stepB;
  int aslot = 0, bslot = 0;
  if (nk > 0) {
    const int kA0 = kt_begin * stepA, kB0 = kt_begin * stepB;
    dma_a(0, 0, kA0);
    dma_a(0, 1, kA0);
    dma_b(B_BASE7, 0, kB0);
    dma_b(B_BASE7, 1, kB0);
    if (nk > 1) {
      dma_a(A_SLOT, 0, kA0 + stepA);
      dma_a(A_SLOT, 1, kA0 + stepA);
      dma_b(B_BASE7 + A_SLOT, 0, kB0 + stepB);
      dma_b(B_BASE7 + A_SLOT, 1, kB0 + stepB);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  V7_BAR();                    // tile 0 landed for every wave's pieces
  if (wr == 1) V7_BAR();       // stagger: the second M-half runs one barrier behind
  if (nk > 0) {
#pragma unroll 1
    for (int T = 0; T < nk; ++T) {
      const bool full = T + 2 < nk, next = T + 1 < nk;
      V7_TILE(full, next);
    }
  }
  if (wr == 0) V7_BAR();       // re-align the two halves
#undef V7_TILE
#undef V7_MMA
#undef V7_CAT
#undef V7_BAR

  wave_epilogue(acc, g, C, Rp, m0, n0, wm0, wn0, smem);
}

template <bool A_RED, bool B_RED, bool FP8 = false>
int launch(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_v7_kernel<A_RED, B_RED, FP8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS7) != hipSuccess)
      return MK_ERR_LAUNCH;
    attr_done = true;
  }
  MK_LAUNCH((gemm_bf16_v7_kernel<A_RED, B_RED, FP8>), grid, dim3(512), LDS7, st, g);
  return mk_check_launch();
}

}  // namespace

namespace mkg {
int launch_v7(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool fp8) {
  if (fp8) return (a_red || b_red) ? MK_ERR_UNSUPPORTED : launch<false, false, true>(g, grid, st);
  if (!a_red && !b_red) return launch<false, false>(g, grid, st);
  if (!a_red && b_red) return launch<false, true>(g, grid, st);
  if (a_red && !b_red) return launch<true, false>(g, grid, st);
  return launch<true, true>(g, grid, st);
}
}  // namespace mkg
