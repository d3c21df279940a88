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

// The kernel is written once over a 16-bit element type e16 and instantiated for bf16 and for f16
// (the reference's fp16 checkpoints / `--fp16 True`): same LDS images, same schedule, the MFMA opcode
// (v_mfma_f32_32x32x16_bf16 / _f16) and the epilogue conversion are the only differences.
#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#include "gemm_v7_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define gemm_bf16_v7_kernel gemm_f16_v7_kernel
#include "gemm_v7_impl.inc"
#undef gemm_bf16_v7_kernel
#undef MK_E16_T
#undef MK_E16_NS

namespace mkg {
int launch_v7(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool fp8, bool f16) {
  if (f16) {
    if (fp8) return MK_ERR_UNSUPPORTED;
    if (!a_red && !b_red) return e_f16::launch<false, false>(g, grid, st);
    if (!a_red && b_red) return e_f16::launch<false, true>(g, grid, st);
    if (a_red && !b_red) return e_f16::launch<true, false>(g, grid, st);
    return e_f16::launch<true, true>(g, grid, st);
  }
  if (fp8) return (a_red || b_red) ? MK_ERR_UNSUPPORTED : e_bf16::launch<false, false, true>(g, grid, st);
  if (!a_red && !b_red) return e_bf16::launch<false, false>(g, grid, st);
  if (!a_red && b_red) return e_bf16::launch<false, true>(g, grid, st);
  if (a_red && !b_red) return e_bf16::launch<true, false>(g, grid, st);
  return e_bf16::launch<true, true>(g, grid, st);
}
}  // namespace mkg
