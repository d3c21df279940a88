// v9: the 256 x 256 x 64 MFMA GEMM with ONE wave per SIMD and a HAND-PLACED K loop (gfx950 / MI355X).
//
// gemm_v8.hip is the same decomposition (4 waves of 128 x 128 = 4 x 4 fragments of v_mfma_f32_32x32x16, 256
// accumulator registers in the AGPR half of the file, A three / B two 32-KiB LDS slots filled by
// buffer_load ... lds, ONE barrier per K-tile) scheduled by hipcc: its loop needs 38-39 cycles per MFMA where the
// vendor's generated assembly needs 35 and a hand-placed stream 32.4 (MI355X_MICROARCH.md "one wave per SIMD";
// DESIGN 4.1).  The disassembly says why: four fragment reads behind the first MFMA of a k-step, three LDS-DMA
// pieces with their M0 writes and s_nops between two MFMAs, counted lgkmcnt waits in the middle of the k-step.
// v9 keeps v8's C++ for everything that runs once per tile (tile order, descriptors, the first LDS-DMA requests,
// the LDS-transposed epilogue) and replaces the K loop by one inline-asm statement whose text is generated
// (scripts/gen_v9_loop.py -> gemm_v9_loop.inc): per K-tile and wave 64 MFMAs, 32 ds_read_b128 (K-major operands;
// 2 ds_read_b64_tr_b16 per fragment of a reduction-major one), 16 LDS-DMA pieces, ~30 SALU, 8 VALU, one wait +
// barrier -- every one of them assigned to an MFMA gap, at most four per gap, the M0 write of a piece one gap
// ahead of its load, no wait inside a k-step.
//
// Round 6: the loops run on v_mfma_f32_16x16x32 (V9_MFMA16 bit mask in the generated file: all four layouts): per K-tile
// and wave 128 MFMAs of 16 cycles in four quarters (quarter q = B fragments 4 (q & 1) .. + 3 x all eight A fragments of
// k32-step q >> 1), the same 32 ds_read_b128 / 64 transposing reads, the same LDS-DMA stream and slot / wait protocol.
// It is the MFMA shape of every vendor kernel for these shapes (profiles/r06_vendor_isa.txt) and it draws less power per
// FLOP (profiles/r06_mfma_power.txt): 1.50 -> 1.38 us per K-tile, 11.4 -> 9.4 us fixed, bit-identical results
// (profiles/r06_gemm_v9_mfma16.txt).  The accumulator layout (a lane owns 4 consecutive columns of one row) gives an
// all-in-registers epilogue of 8-byte accesses for alpha (+ residual).
//
// Whole tiles only (M % 256 == N % 256 == 0, K % 64 == 0): the launcher in gemm.hip sends everything else to v7.
//
// Replaces the nn.Linear matmuls of /root/reference/modeling.py:134-140,159-162,597 and their gradients (same
// contract as gemm_v7.hip).
#include "gemm_common.h"
#include "gemm_v9_loop.inc"

#define MK_E16_T bf16
#define MK_E16_NS e_bf16
#define MK_V9_SFX "bf16"
#include "gemm_v9_impl.inc"
#undef MK_E16_T
#undef MK_E16_NS
#undef MK_V9_SFX
#define MK_E16_T _Float16
#define MK_E16_NS e_f16
#define MK_V9_SFX "f16"
#define gemm_bf16_v9_kernel gemm_f16_v9_kernel
#include "gemm_v9_impl.inc"
#undef gemm_bf16_v9_kernel
#undef MK_E16_T
#undef MK_E16_NS
#undef MK_V9_SFX

namespace mkg {
// layouts whose K loop runs on 16 x 16 x 32 MFMAs with the all-in-registers epilogue (any epilogue, walking allowed):
// bit (2 a_red + b_red) (scripts/gen_v9_loop.py)
int v9_mfma16_layouts() {
#ifdef V9_MFMA16
  return V9_MFMA16;
#else
  return 0;
#endif
}
int launch_v9(const GemmArgs& g, bool a_red, bool b_red, dim3 grid, hipStream_t st, bool f16) {
  if (f16) {
    if (!a_red && !b_red) return e_f16::launch_v9<false, false>(g, grid, st);
    if (!a_red && b_red) return e_f16::launch_v9<false, true>(g, grid, st);
    if (a_red && !b_red) return e_f16::launch_v9<true, false>(g, grid, st);
    return e_f16::launch_v9<true, true>(g, grid, st);
  }
  if (!a_red && !b_red) return e_bf16::launch_v9<false, false>(g, grid, st);
  if (!a_red && b_red) return e_bf16::launch_v9<false, true>(g, grid, st);
  if (a_red && !b_red) return e_bf16::launch_v9<true, false>(g, grid, st);
  return e_bf16::launch_v9<true, true>(g, grid, st);
}
}  // namespace mkg
