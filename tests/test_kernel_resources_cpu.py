"""Occupancy budgets of the shipped gfx950 kernels, read from the built library's own code-object
metadata (scripts/kernel_resources.py; no GPU, no recompilation).

Why this exists: in round 3 a run-time branch in the shared GEMM epilogue (per-row / per-column fp8
scales) raised every 16-bit 128x128 kernel from 125 to 136-142 VGPRs = from four to three waves per
SIMD, and the BK = 32 instantiations -- LDS leaves room for five workgroups per CU, registers decide --
lost 12-24 % (profiles/r02_cfg5_last_step.txt vs r03: 11.4 -> 14.1 ms).  Parity tests cannot see that."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts"))
import kernel_resources as kr  # noqa: E402

LLVM_TOOLS = all(os.path.exists(os.path.join(kr.LLVM, t)) for t in ("llvm-objcopy", "llvm-readelf"))
pytestmark = pytest.mark.skipif(not (LLVM_TOOLS and os.path.exists(kr.LIB)),
                                reason="needs the built library and ROCm's llvm-objcopy / llvm-readelf")


@pytest.fixture(scope="module")
def ks():
    from macaw_llm_amd import build
    build.build()                       # no-op when the stamp matches the sources
    return kr.kernels()


def _pick(ks, *subs):
    out = {k: v for k, v in ks.items() if all(s in k for s in subs)}
    assert out, f"no kernel matches {subs}"
    return out


def test_every_kernel_of_the_library_is_listed_and_both_element_types_exist(ks):
    assert len(ks) > 200
    for ns in ("e_bf16::", "e_f16::"):
        for stem in ("v7_kernel", "v2_kernel", "flash_fwd", "flash_bwd_dkv", "decode_step_attn", "gemm_skinny16"):
            _pick(ks, ns, stem)


def test_the_128x128_gemm_kernels_keep_four_waves_per_simd(ks):
    """512 VGPRs per SIMD lane / 4 waves = 128.  Matters for BK = 32 (32 KiB of LDS per workgroup: five fit, so
    registers decide); BK = 64 needs 64 KiB of LDS (two workgroups per CU whatever the registers) -- its NT
    instantiation is allowed the 140 it has always had."""
    for k, v in _pick(ks, "gemm_", "_v2_kernel<").items():
        if "fp8" in k:
            continue
        limit = 144 if k.endswith("<false, false, 64>(mkg::GemmArgs)") else 128
        assert v["vgpr_count"] <= limit, (k, v)


def test_the_256x256_gemm_kernels_fit_two_workgroups_of_eight_waves_per_cu(ks):
    for k, v in _pick(ks, "_v7_kernel<").items():
        assert v["vgpr_count"] <= 256 and v.get("vgpr_spill_count", 0) == 0, (k, v)
        assert v.get("private_segment_fixed_size", 0) == 0, (k, v)


def test_the_one_wave_per_simd_gemm_keeps_its_accumulators_in_registers(ks):
    """v8: 256 accumulator registers (AGPRs) + fragments / addresses in the other half of the 512-entry file, no
    spill and NO scratch: with LLVM's default `#pragma unroll` budget the 4 x 4-fragment epilogue stays rolled,
    indexes the accumulators dynamically and the whole tile goes through scratch (build.FILE_FLAGS)."""
    found = {k: v for k, v in ks.items() if "_v8_kernel<" in k}
    if not found:
        pytest.skip("gemm_v8 is an experiment kernel (MK_EXPERIMENTS=1); its successor is pinned below")
    assert len(found) == 8                                  # 4 layouts x {bf16, f16}
    for k, v in found.items():
        assert 256 < v["vgpr_count"] <= 512, (k, v)         # (the note counts VGPRs + AGPRs of the unified file)
        assert v.get("vgpr_spill_count", 0) == 0 and v.get("private_segment_fixed_size", 0) == 0, (k, v)
        assert v["max_flat_workgroup_size"] == 256, (k, v)


def test_the_hand_placed_gemm_owns_the_accumulator_file_and_nothing_spills(ks):
    """v9: the asm block owns a[0:255] (256 AGPRs) and v[0:87]; the compiler's code around it must stay inside the
    arch VGPRs it has left (<= 256 in all, so that it never parks a value in an AGPR the asm block is using), without
    spills or scratch, one wave per SIMD."""
    found = _pick(ks, "_v9_kernel<")
    assert len(found) == 8                                  # 4 layouts x {bf16, f16}
    for k, v in found.items():
        assert v.get("agpr_count", 256) == 256, (k, v)
        assert 256 + 88 <= v["vgpr_count"] <= 512, (k, v)   # (the note counts VGPRs + AGPRs of the unified file)
        assert v.get("vgpr_spill_count", 0) == 0 and v.get("private_segment_fixed_size", 0) == 0, (k, v)
        assert v["max_flat_workgroup_size"] == 256, (k, v)


def test_no_hot_kernel_spills(ks):
    """two known exceptions: the non-causal head-dim-128 dq kernel (4 VGPRs, 20 B of scratch; not on any
    BASELINE configuration's path -- LLaMA is causal, the alignment attention has no backward through it) and the
    fp16 causal short-sequence backward with RoPE inside (both forms; <= 4 lane-index values stored once before phase 1 and
    read back once in phase 2, outside every loop; the bf16 instantiation -- BASELINE's dtype -- has none)"""
    spilled = {k: v for k, v in ks.items() if v.get("vgpr_spill_count", 0)}      # (SGPR -> VGPR-lane spills cost nothing)
    allowed = [k for k in spilled if "flash_bwd_dq" in k and "<128, false, false>" in k]
    allowed += [k for k in spilled if ("flash_bwd_short_f16_kernel<128, true, 1>" in k or "flash_bwd_short_f16_kernel<128, true, 2>" in k)
                and spilled[k]["vgpr_spill_count"] <= 4]
    unexpected = {k: v for k, v in spilled.items() if k not in allowed}
    assert not unexpected, unexpected


def test_attention_and_streaming_kernels_keep_their_occupancy(ks):
    for k, v in _pick(ks, "flash_fwd").items():
        if "flash_fwd4x64" in k:                           # (MK_EXPERIMENTS builds) one wave per SIMD by design: 512 registers, no scratch
            assert 256 < v["vgpr_count"] <= 512 and v.get("private_segment_fixed_size", 0) == 0, (k, v)
            assert v["max_flat_workgroup_size"] == 256, (k, v)
            continue
        assert v["vgpr_count"] <= 256, (k, v)              # two waves per SIMD (two 4-wave workgroups or one of 8 waves)
    for k, v in _pick(ks, "adamw_multi_kernel").items():
        assert v["vgpr_count"] <= 72, (k, v)               # HBM-bound: >= 7 waves per SIMD in flight
    for stem in ("gemm_skinny16_kernel", "gemm_skinny32p_kernel"):
        for k, v in _pick(ks, stem).items():
            assert v["vgpr_count"] <= 128, (k, v)          # weight streaming: two 8-wave workgroups per CU


def test_no_compiler_code_touches_the_hand_placed_gemms_accumulator_registers():
    """v9's 256 accumulators live in a0..a255 ACROSS asm statements (K loop -> compiler code that sets
    up / requests the next tile -> the V9_ACC_READ asms) and are declared only as clobbers, so for the compiler the
    AGPRs are free in between: a VGPR -> AGPR copy or spill placed there would silently corrupt the tile, and the
    metadata counts above cannot see it.  Disassemble every *_v9_kernel: the ONLY instructions with an AGPR operand are
    the MFMAs, `v_accvgpr_write_b32 aN, 0` (the loops' zeroing: 2 loop forms x 256) and `v_accvgpr_read_b32` with
    every aN read equally often (2 epilogue forms x 256); no load, no copy from a VGPR."""
    import collections
    import re
    dis = kr.disassemble("v9_kernel")
    assert len(dis) == 8
    areg = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    for sym, lines in dis.items():
        ops, reads = collections.Counter(), collections.Counter()
        for ins in lines:
            if not areg.search(ins):
                continue
            op = ins.split()[0]
            ops[op] += 1
            if op == "v_accvgpr_write_b32":
                assert re.search(r"\ba\d+, 0$", ins), (sym, ins)          # zeroing only, never a value from a VGPR
            elif op == "v_accvgpr_read_b32":
                reads[int(areg.search(ins).group(1))] += 1
            else:
                assert op.startswith(("v_mfma_f32_32x32x16_", "v_mfma_f32_16x16x32_")), (sym, ins)   # no ds_read / buffer_load / v_mov into an AGPR
        assert ops["v_accvgpr_write_b32"] == 512 and ops["v_accvgpr_read_b32"] == 512, (sym, dict(ops))
        assert sorted(reads) == list(range(256)) and set(reads.values()) == {2}, sym
