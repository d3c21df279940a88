// v8: the 256 x 256 x 64 MFMA GEMM with ONE wave per SIMD (gfx950 / MI355X).
//
// v7 (gemm_v7.hip) runs eight waves of 128 x 64 per 256 x 256 tile, two per SIMD, which take turns
// on the matrix pipe: every K-tile moves 192 KiB from LDS into registers and crosses four barriers,
// and its load segment (8 LDS-DMA pieces + 12 ds_reads per wave) does not fit beside the partner's
// 512 MFMA cycles (1.49 us per K-tile against 1.08 of MFMA work).  v8 is the other corner of the
// design space, the one the vendor's own tuned gfx950 kernels sit in (MT256x256x64, 256 threads,
// 8 x 8 fragments per wave, direct-to-LDS):
//
//   * 4 waves = 2 (M) x 2 (N), wave tile 128 x 128 = 4 x 4 fragments of v_mfma_f32_32x32x16
//     (256 accumulator registers of the 512 a lone wave on a SIMD owns), 160 KiB LDS, one workgroup
//     per CU.  LDS reads per K-tile: 128 KiB instead of 192 (each operand byte feeds 128 columns).
//   * ONE barrier per K-tile.  Software pipeline inside the wave: the fragments of k-step s + 1 are
//     read (second register set) between the 16 MFMAs of k-step s, the LDS-DMA pieces of the tiles
//     two ahead are sprinkled between the MFMAs (the order is pinned with sched_group_barrier), so
//     the matrix pipe is fed by a single in-order stream with <= 2 other instructions per 32-cycle
//     MFMA slot.
//   * LDS ring as v7: A three tile slots, B two (5 x 32 KiB).  The barrier sits between k-steps 2
//     and 3 of tile T: before it every wave waits (counted vmcnt) for its pieces of tile T + 1 and
//     for its last reads of tile T; after it k-step 0 of tile T + 1 is read, B(T + 2) is requested
//     into the slot B(T) has just left and (during the next tile) A(T + 3) into A(T)'s.
//   * LDS images, source-side swizzle, buffer descriptors and the LDS-transposed epilogue are v7's
//     (gemm_lds_image.inc, mkg::wave_epilogue).
//
// Replaces the nn.Linear matmuls of modeling.py:134-140,159-162,597 and their gradients (same
// contract as gemm_v7.hip).
#include "gemm_common.h"

#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#include "gemm_v8_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define gemm_bf16_v8_kernel gemm_f16_v8_kernel
#include "gemm_v8_impl.inc"
#undef gemm_bf16_v8_kernel
#undef MK_E16_T
#undef MK_E16_NS

namespace mkg {
int launch_v8(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool f16) {
  if (f16) {
    if (!a_red && !b_red) return e_f16::launch_v8<false, false>(g, grid, st);
    if (!a_red && b_red) return e_f16::launch_v8<false, true>(g, grid, st);
    if (a_red && !b_red) return e_f16::launch_v8<true, false>(g, grid, st);
    return e_f16::launch_v8<true, true>(g, grid, st);
  }
  if (!a_red && !b_red) return e_bf16::launch_v8<false, false>(g, grid, st);
  if (!a_red && b_red) return e_bf16::launch_v8<false, true>(g, grid, st);
  if (a_red && !b_red) return e_bf16::launch_v8<true, false>(g, grid, st);
  return e_bf16::launch_v8<true, true>(g, grid, st);
}
}  // namespace mkg
