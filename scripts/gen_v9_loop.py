#!/usr/bin/env python3
"""Generates macaw_llm_amd/csrc/gemm_v9_loop.inc: the hand-placed K loop of the 4-wave 256 x 256 x 64 GEMM
(gemm_v9.hip) as ONE inline-asm statement per operand layout.

Why a generator: the loop is 64 MFMAs per K-tile with every other instruction (fragment reads, LDS-DMA pieces and
their M0 writes, slot arithmetic, address updates) assigned to a specific MFMA gap -- hipcc cannot be steered to
that (gemm_v8: 38-39 cycles per MFMA against 32.4 for a hand-placed stream, MI355X_MICROARCH.md "one wave per
SIMD").  The placement rule lives HERE (function `kstep`), the output is committed next to the
kernel and `tests/test_v9_gen_cpu.py` checks that the committed file is what this script writes and that the
addressing identities the asm relies on hold against the C++ formulas of gemm_lds_image.inc.

Register map of the asm block (fixed physical registers, all in the clobber list):
  a[0:255]   accumulators, fragment (i, j) at a[(4 i + j) * 16 ...]           (i: A fragment row, j: B fragment)
  v[0:15]    A fragments set 0      v[16:31] B fragments set 0
  v[32:47]   A fragments set 1      v[48:63] B fragments set 1
  v[64:71]   LDS-DMA voffsets of A  [h][i]       v[72:79] of B
  v[80:83]   LDS read addresses of A (K-major: per k-step; reduction-major: per fragment), v[84:87] of B
  s64 aoff (A slot of tile T: 0 / 32 Ki / 64 Ki)   s65 boff (0 / 32 Ki)   s66 kA2   s67 kB2 (K-tile byte offsets of T + 2)
  s68 loop counter   s69 M0 base of the current piece group   s70 a1off   s71 a2off   s72 bnext   s73 dA   s74 dB   s75 kB1

Round 6: the loops that ship run on v_mfma_f32_16x16x32 (MFMA16 mask; functions *16 below: `loop_text16`, `body16`, `quarter`);
the round-5 loops on v_mfma_f32_32x32x16 (`loop_text`, `body`, `kstep`) are still generated with --no-mfma16 and still
simulated by the tests.
"""
import sys

# timing-only ablations for experiment builds (scripts/probe/build_v9_variants.sh); the shipped text has none set
NO_DMA = NO_READ = NO_BAR = False
MFMA16 = 15         # bit mask of the layouts (2 a_red + b_red) whose loop runs on v_mfma_f32_16x16x32 (round 6: K-tile 1.50 ->
                    # 1.38 us, profiles/r06_gemm_v9_mfma16.txt); --mfma16=<mask>, --no-mfma16 = round 5's file

HALF = 16384
SLOT = 32768
B_BASE = 3 * SLOT

FA = [[0, 32][s] for s in range(2)]      # first VGPR of A fragments, set s
FB = [[16, 48][s] for s in range(2)]
VOA, VOB, ADA, ADB = 64, 72, 80, 84


class Emit:
    def __init__(self):
        self.lines = []

    def __call__(self, s):
        self.lines.append(s)


def frag_reads(red, set_base, addr_base, ks):
    """instructions that read the 4 fragments of k-step `ks` of one operand into the register set at set_base"""
    out = []
    for f in range(4):
        r = set_base + 4 * f
        if not red:
            off = f * 4096
            out.append(f"ds_read_b128 v[{r}:{r + 3}], v{addr_base + ks}" + (f" offset:{off}" if off else ""))
        else:
            off = ks * 4096
            out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{addr_base + f}" + (f" offset:{off}" if off else ""))
            out.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{addr_base + f} offset:{off + 1024}")
    return out


def reads_of_kstep(a_red, b_red, s, ks):
    """fragment reads of k-step ks into set s, in the order the MFMAs consume them (j outer, i inner:
    B0, A0..A3, B1, B2, B3)"""
    ra = frag_reads(a_red, FA[s], ADA, ks)
    rb = frag_reads(b_red, FB[s], ADB, ks)
    na, nb = (2 if a_red else 1), (2 if b_red else 1)
    return rb[:nb] + ra + rb[nb:]


def mfmas(s, first_zero=False):
    out = []
    for j in range(4):
        for i in range(4):
            acc = (4 * i + j) * 16
            c = "0" if first_zero else f"a[{acc}:{acc + 15}]"
            out.append(f"v_mfma_f32_32x32x16_@SFX@ a[{acc}:{acc + 15}], v[{FB[s] + 4 * j}:{FB[s] + 4 * j + 3}], "
                       f"v[{FA[s] + 4 * i}:{FA[s] + 4 * i + 3}], {c}")
    return out


def kstep(e, mset, reads, dma, salu, valu):
    """one k-step: 16 MFMAs on register set `mset`; the fillers by gap (gap g follows MFMA g):
       reads (<= 24)  : gaps 0 .. 7 (one per gap) -- or two / three per gap when an operand is reduction-major
       dma (0 or 4)   : the M0 write in gaps 8 / 10 / 12 / 14, the buffer_load ... lds in gaps 9 / 11 / 13 / 15
       salu, valu     : one entry per gap, SALU in gaps 0 .. 7 (instructions that communicate through SCC are ONE
                        entry and stay adjacent), VALU from gap 8 on
    At most 4 entries follow any MFMA (the budget of a lone wave is ~5 issues per 32-cycle gap)."""
    gaps = [[] for _ in range(16)]
    if NO_READ:
        reads = []
    if NO_DMA:
        dma = []
    per = (len(reads) + 7) // 8 if reads else 0
    for k, r in enumerate(reads):
        gaps[k // per].append(r)
    for k, (m0w, ld) in enumerate(dma):
        gaps[8 + 2 * k].append(m0w)
        gaps[9 + 2 * k].append(ld)
    assert len(salu) <= 8
    for g, ins in enumerate(salu):          # gap g <- entry g: everything a piece group needs is formed by gap 7
        gaps[g].append(ins)
    g = 8 if reads else 0
    for ins in valu:
        gaps[g].append(ins)
        g += 1
    ms = mfmas(mset)
    e("s_waitcnt lgkmcnt(0)")
    for k in range(16):
        e(ms[k])
        for ins in gaps[k]:
            for part in ins.split(" ; "):
                e(part)
        assert len(gaps[k]) <= 4, (k, gaps[k])


def dma_group(op, dst_expr_sgpr, vo_base, soff):
    """4 pieces: (M0 write, load).  s69 holds the LDS address of piece 0; pieces are 4 KiB apart."""
    rs = "%[rsA]" if op == "A" else "%[rsB]"
    out = []
    for i in range(4):
        m0w = f"s_add_u32 m0, s69, {i * 4096}" if i else "s_mov_b32 m0, s69"
        out.append((m0w, f"buffer_load_dwordx4 v{vo_base + i}, {rs}, {soff} offen lds"))
    return out


def body(e, a_red, b_red, mode):
    """one K-tile T.  mode FULL: tiles T + 1 and T + 2 exist; NEXT: T + 1 only; LAST: neither."""
    nxt = mode != "LAST"
    full = mode == "FULL"
    # ---- k-step 0: MFMA set 0, reads k-step 1 -> set 1, LDS-DMA B(T + 1) half 1
    salu = []
    if nxt:
        salu += ["s_add_u32 s70, s64, 0x8000 ; s_cmp_eq_u32 s70, 0x18000 ; s_cselect_b32 s70, 0, s70",
                 "s_xor_b32 s72, s65, 0x8000",
                 "s_sub_u32 s75, s67, %[stB]",
                 "s_add_u32 s69, %[wv], s72",
                 f"s_add_u32 s69, s69, {B_BASE + HALF}",
                 "s_sub_u32 s73, s70, s64",
                 "s_sub_u32 s74, s72, s65"]
    if full:
        salu += ["s_sub_u32 s71, s64, 0x8000 ; s_cmp_eq_u32 s64, 0 ; s_cselect_b32 s71, 0x10000, s71"]
    kstep(e, 0, reads_of_kstep(a_red, b_red, 1, 1), dma_group("B", None, VOB + 4, "s75") if nxt else [], salu, [])
    # ---- k-step 1: MFMA set 1, reads k-step 2 -> set 0, LDS-DMA A(T + 2) half 0
    if full:
        e("s_add_u32 s69, %[wv], s71")
    kstep(e, 1, reads_of_kstep(a_red, b_red, 0, 2), dma_group("A", None, VOA, "s66") if full else [], [], [])
    # ---- k-step 2: MFMA set 0, reads k-step 3 -> set 1, LDS-DMA A(T + 2) half 1; then the read addresses move
    # to tile T + 1 (every address register had its last use for tile T in this k-step's reads)
    if full:
        e(f"s_add_u32 s69, s69, {HALF}")
    valu = []
    if nxt:
        valu = [f"v_add_u32 v{ADA + k}, s73, v{ADA + k}" for k in range(4)] + \
               [f"v_add_u32 v{ADB + k}, s74, v{ADB + k}" for k in range(4)]
    kstep(e, 0, reads_of_kstep(a_red, b_red, 1, 3), dma_group("A", None, VOA + 4, "s66") if full else [], [], valu)
    # ---- the one barrier of the tile: this wave's pieces of tile T + 1 have landed (everything older than the
    # eight A(T + 2) pieces), its reads of tile T are complete
    if nxt:
        e("s_waitcnt vmcnt(8) lgkmcnt(0)" if full else "s_waitcnt vmcnt(0) lgkmcnt(0)")
        if not NO_BAR:
            e("s_barrier")
        e("s_add_u32 s69, %[wv], s65")
        e(f"s_add_u32 s69, s69, {B_BASE}")
    # ---- k-step 3: MFMA set 1, reads k-step 0 of tile T + 1 -> set 0, LDS-DMA B(T + 2) half 0 into B(T)'s slot
    salu = []
    if nxt:
        salu = ["s_mov_b32 s64, s70", "s_mov_b32 s65, s72", "s_add_u32 s66, s66, %[stA]", "s_add_u32 s67, s67, %[stB]"]
        # (s65 is overwritten only after the M0 base of this k-step's pieces was formed above; s67 only after the
        #  last piece has been issued: see the order check in `order_ok`)
    kstep3(e, a_red, b_red, nxt, full, salu)


def kstep3(e, a_red, b_red, nxt, full, salu):
    reads = reads_of_kstep(a_red, b_red, 0, 0) if (nxt and not NO_READ) else []
    dma = dma_group("B", None, VOB, "s67") if (full and not NO_DMA) else []
    gaps = [[] for _ in range(16)]
    per = (len(reads) + 7) // 8 if reads else 0
    for k, r in enumerate(reads):
        gaps[k // per].append(r)
    for k, (m0w, ld) in enumerate(dma):
        gaps[8 + 2 * k].append(m0w)
        gaps[9 + 2 * k].append(ld)
    # bookkeeping: s64 / s65 early, the K-tile offsets behind the last piece (gap 15 reads s67)
    if salu:
        gaps[0].append(salu[0])
        gaps[1].append(salu[1])
        gaps[15].append(salu[2])
        gaps[15].append(salu[3])
    ms = mfmas(1)
    e("s_waitcnt lgkmcnt(0)")
    for k in range(16):
        e(ms[k])
        for ins in gaps[k]:
            e(ins)
        assert len(gaps[k]) <= 4


def prologue(e, a_red, b_red, walk=False):
    e(f"v_mov_b32 v{VOA}, %[voA]")
    e(f"v_mov_b32 v{VOB}, %[voB]")
    for op, vo, red in (("A", VOA, a_red), ("B", VOB, b_red)):
        st = f"%[i{op}]"
        for i in range(1, 4):
            e(f"v_add_u32 v{vo + i}, {st}, v{vo + i - 1}")
        if not red:                       # h = 1: 128 rows further = four piece strides
            e(f"v_add_u32 v{vo + 4}, {st}, v{vo + 3}")
            for i in range(1, 4):
                e(f"v_add_u32 v{vo + 4 + i}, {st}, v{vo + 3 + i}")
        else:                             # h = 1: 128 columns further = 256 bytes
            for i in range(4):
                e(f"v_add_u32 v{vo + 4 + i}, 0x100, v{vo + i}")
    for op, ad, red in (("A", ADA, a_red), ("B", ADB, b_red)):
        e(f"v_mov_b32 v{ad}, %[ad{op}]")
        sh = 6 if red else 5
        for k in range(1, 4):
            e(f"v_xor_b32 v{ad + k}, {hex(k << sh)}, v{ad}")
    e("s_mov_b32 s64, 0")
    e("s_mov_b32 s65, 0")
    e("s_lshl_b32 s66, %[stA], 1")
    e("s_lshl_b32 s67, %[stB], 1")
    e("s_sub_u32 s68, %[nk], 2")
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    # a walking workgroup's later tiles: the epilogue stores of the previous tile are still in flight behind the
    # requests of this one, and loads and stores share the counter without a guaranteed order between them --
    # only vmcnt(0) says "tile 0 has landed" there (it waits for A(1) / B(1) half 0 and the stores as well)
    e("s_waitcnt vmcnt(0)" if walk else "s_waitcnt vmcnt(12)")
    e("s_barrier")
    for r in reads_of_kstep(a_red, b_red, 0, 0):
        e(r)


# ------------------------------------------------------------------------------------------------------------------
# EXPERIMENT (--mfma16, round 6): the K-major x K-major loop on v_mfma_f32_16x16x32_* -- the shape every vendor kernel
# uses (profiles/r06_vendor_isa.txt) and the one that draws 4 % less power per FLOP (profiles/r06_mfma_power.txt).
# Same tile, same LDS image, same LDS-DMA stream, same slot / wait protocol; per K-tile 128 MFMAs of 16 cycles in four
# QUARTERS of 32 (quarter q = B fragments 4 (q & 1) .. + 3 of k32-step q >> 1, all 8 A fragments), 32 ds_read_b128.
# Register map:  a[(8 i + j) * 4 ..] fragment (i, j) of 16 x 16;  set s: A fragments v[64 s + 4 i ..], B fragments
# v[64 s + 32 + 4 j ..];  v[128:135] / v[136:143] LDS-DMA voffsets of A / B;  v[144:145] / v[148:149] LDS read addresses
# of A / B per k32-step (the swizzle depends on it: + (1 << 6)); fragments of 16 rows are + 2048 apart.
SA16, SB16 = [0, 64], [32, 96]
VOA16, VOB16, ADA16, ADB16 = 128, 136, 144, 152       # 8 read-address registers per operand (a K-major one uses 2)
NV16 = 160                                            # VGPRs of the asm block: v[0:159]


def frag_reads16(red, set_base, ad, h):
    """the reads of the 8 fragments (16 rows x 32 k) of k32-step h of one operand.  K-major: one ds_read_b128 each, address
    register ad + h, fragments + 2048 apart.  Reduction-major: two transposing reads each (k rows + 0 .. 3, + 4 .. 7 of the
    lane's group of 8), address register ad + f (f = 2 a + b: the swizzle of the image depends on both bits), k32-steps
    + 8192 apart"""
    out = []
    for f in range(8):
        r = set_base + 4 * f
        if not red:
            out.append(f"ds_read_b128 v[{r}:{r + 3}], v{ad + h}" + (f" offset:{f * 2048}" if f else ""))
        else:
            off = h * 8192
            out.append(f"ds_read_b64_tr_b16 v[{r}:{r + 1}], v{ad + f}" + (f" offset:{off}" if off else ""))
            out.append(f"ds_read_b64_tr_b16 v[{r + 2}:{r + 3}], v{ad + f} offset:{off + 1024}")
    return out


def mfmas16(s, q):
    out = []
    for j in range(4 * (q & 1), 4 * (q & 1) + 4):
        for i in range(8):
            acc = (8 * i + j) * 4
            out.append(f"v_mfma_f32_16x16x32_@SFX@ a[{acc}:{acc + 3}], v[{SB16[s] + 4 * j}:{SB16[s] + 4 * j + 3}], "
                       f"v[{SA16[s] + 4 * i}:{SA16[s] + 4 * i + 3}], a[{acc}:{acc + 3}]")
    return out


def quarter(e, q, reads, dma, salu, valu, wait, tail_salu=()):
    """32 MFMAs = 16 slots of two; the fillers of a slot are split between its two MFMAs.  Reads are spread evenly over the
    16 slots (<= 2 per slot), LDS-DMA pieces sit in slots 8 .. 15 (M0 write, then the load in the next slot), SALU entries
    in slots 0 .. 7 (an SCC chain is ONE entry and stays adjacent), VALU entries one per slot from slot 0 on when the quarter
    has no reads, else from slot 8; `tail_salu` goes behind everything else of slot 15."""
    slots = [[] for _ in range(16)]
    if NO_READ:
        reads = []
    if NO_DMA:
        dma = []
    per = (len(reads) + 15) // 16 if reads else 0
    for k, r in enumerate(reads):
        slots[k // per].append(r)
    for k, (m0w, ld) in enumerate(dma):
        slots[8 + 2 * k].append(m0w)
        slots[9 + 2 * k].append(ld)
    assert len(salu) <= 8
    for g, ins in enumerate(salu):
        slots[g].append(ins)
    g = 0 if not reads else 8
    for ins in valu:
        slots[g % 16].append(ins)
        g += 1
    for ins in tail_salu:
        slots[15].append(ins)
    ms = mfmas16(q >> 1, q)
    if wait:
        e("s_waitcnt lgkmcnt(0)")
    for g in range(16):
        assert len(slots[g]) <= (6 if g == 15 else 4), (q, g, slots[g])
        half = (len(slots[g]) + 1) // 2
        e(ms[2 * g])
        for ins in slots[g][:half]:
            for part in ins.split(" ; "):
                e(part)
        e(ms[2 * g + 1])
        for ins in slots[g][half:]:
            for part in ins.split(" ; "):
                e(part)


def dma_group16(op, vo_base, soff):
    rs = "%[rsA]" if op == "A" else "%[rsB]"
    out = []
    for i in range(4):
        m0w = f"s_add_u32 m0, s69, {i * 4096}" if i else "s_mov_b32 m0, s69"
        out.append((m0w, f"buffer_load_dwordx4 v{vo_base + i}, {rs}, {soff} offen lds"))
    return out


def body16(e, a_red, b_red, mode):
    """one K-tile T in four quarters.  Quarter q computes B fragments 4 (q & 1) .. + 3 x all A fragments of k32-step q >> 1
    out of register set q >> 1.  Reads: quarter 0 fetches A, quarter 1 B of k32-step 1 into set 1; quarter 3 (behind the
    tile's barrier) all of k32-step 0 of tile T + 1 into set 0.  LDS-DMA groups and the slot arithmetic as in `body`."""
    nxt = mode != "LAST"
    full = mode == "FULL"
    salu = []
    if nxt:
        salu += ["s_add_u32 s70, s64, 0x8000 ; s_cmp_eq_u32 s70, 0x18000 ; s_cselect_b32 s70, 0, s70",
                 "s_xor_b32 s72, s65, 0x8000",
                 "s_sub_u32 s75, s67, %[stB]",
                 "s_add_u32 s69, %[wv], s72",
                 f"s_add_u32 s69, s69, {B_BASE + HALF}",
                 "s_sub_u32 s73, s70, s64",
                 "s_sub_u32 s74, s72, s65"]
    if full:
        salu += ["s_sub_u32 s71, s64, 0x8000 ; s_cmp_eq_u32 s64, 0 ; s_cselect_b32 s71, 0x10000, s71"]
    quarter(e, 0, frag_reads16(a_red, SA16[1], ADA16, 1), dma_group16("B", VOB16 + 4, "s75") if nxt else [], salu, [],
            wait=True)
    if full:
        e("s_add_u32 s69, %[wv], s71")
    quarter(e, 1, frag_reads16(b_red, SB16[1], ADB16, 1), dma_group16("A", VOA16, "s66") if full else [], [], [],
            wait=False)
    if full:
        e(f"s_add_u32 s69, s69, {HALF}")
    valu = []
    if nxt:     # the read addresses move to tile T + 1's slots (their last reads of tile T were issued in quarters 0 / 1)
        valu = [f"v_add_u32 v{ADA16 + k}, s73, v{ADA16 + k}" for k in range(8 if a_red else 2)] + \
               [f"v_add_u32 v{ADB16 + k}, s74, v{ADB16 + k}" for k in range(8 if b_red else 2)]
    quarter(e, 2, [], dma_group16("A", VOA16 + 4, "s66") if full else [], [], valu, wait=True)
    if nxt:
        e("s_waitcnt vmcnt(8) lgkmcnt(0)" if full else "s_waitcnt vmcnt(0) lgkmcnt(0)")
        if not NO_BAR:
            e("s_barrier")
        e("s_add_u32 s69, %[wv], s65")
        e(f"s_add_u32 s69, s69, {B_BASE}")
    salu = ["s_mov_b32 s64, s70", "s_mov_b32 s65, s72"] if nxt else []
    tail = ["s_add_u32 s66, s66, %[stA]", "s_add_u32 s67, s67, %[stB]"] if nxt else []
    reads = (frag_reads16(a_red, SA16[0], ADA16, 0) + frag_reads16(b_red, SB16[0], ADB16, 0)) if nxt else []
    quarter(e, 3, reads, dma_group16("B", VOB16, "s67") if full else [], salu, [], wait=False, tail_salu=tail)


def prologue16(e, a_red, b_red, walk):
    e(f"v_mov_b32 v{VOA16}, %[voA]")
    e(f"v_mov_b32 v{VOB16}, %[voB]")
    for op, vo, red in (("A", VOA16, a_red), ("B", VOB16, b_red)):
        st = f"%[i{op}]"
        for i in range(1, 4):
            e(f"v_add_u32 v{vo + i}, {st}, v{vo + i - 1}")
        if not red:                       # h = 1: 128 rows further = four piece strides
            e(f"v_add_u32 v{vo + 4}, {st}, v{vo + 3}")
            for i in range(1, 4):
                e(f"v_add_u32 v{vo + 4 + i}, {st}, v{vo + 3 + i}")
        else:                             # h = 1: 128 columns further = 256 bytes
            for i in range(4):
                e(f"v_add_u32 v{vo + 4 + i}, 0x100, v{vo + i}")
    for op, ad, red in (("A", ADA16, a_red), ("B", ADB16, b_red)):
        e(f"v_mov_b32 v{ad}, %[ad{op}]")
        if not red:
            e(f"v_xor_b32 v{ad + 1}, 0x40, v{ad}")                      # k32-step 1: 16-byte chunk + 4
        else:
            for f in range(1, 8):                                      # fragment f = 2 a + b: chunk ^ (4 a + 2 b)
                e(f"v_xor_b32 v{ad + f}, {hex(((f >> 1) << 6) | ((f & 1) << 5))}, v{ad}")
    e("s_mov_b32 s64, 0")
    e("s_mov_b32 s65, 0")
    e("s_lshl_b32 s66, %[stA], 1")
    e("s_lshl_b32 s67, %[stB], 1")
    e("s_sub_u32 s68, %[nk], 2")
    for r in range(256):
        e(f"v_accvgpr_write_b32 a{r}, 0")
    e("s_waitcnt vmcnt(0)" if walk else "s_waitcnt vmcnt(12)")
    e("s_barrier")
    for r in frag_reads16(a_red, SA16[0], ADA16, 0) + frag_reads16(b_red, SB16[0], ADB16, 0):
        e(r)


def loop_text16(a_red, b_red, walk=False):
    e = Emit()
    prologue16(e, a_red, b_red, walk)
    e("s_cmp_eq_u32 s68, 0")
    e("s_cbranch_scc1 .Lv9n%=")
    e(".p2align 6")
    e(".Lv9l%=:")
    body16(e, a_red, b_red, "FULL")
    e("s_sub_u32 s68, s68, 1")
    e("s_cmp_lg_u32 s68, 0")
    e("s_cbranch_scc1 .Lv9l%=")
    e(".Lv9n%=:")
    body16(e, a_red, b_red, "NEXT")
    body16(e, a_red, b_red, "LAST")
    e("s_nop 15")
    e("s_nop 15")
    return e.lines


def acc_read_macros16():
    out = []
    for i in range(8):
        for j in range(8):
            base = (8 * i + j) * 4
            txt = "".join(f"v_accvgpr_read_b32 %{e}, a{base + e}\\n\\t" for e in range(4))
            outs = ", ".join(f'"=v"((X)[{e}])' for e in range(4))
            out.append(f"#define V9_ACC16_READ_{i}_{j}(X) asm volatile(\"{txt}\" : {outs})")
    return "\n".join(out)


def loop_text(a_red, b_red, walk=False):
    e = Emit()
    prologue(e, a_red, b_red, walk)
    e("s_cmp_eq_u32 s68, 0")
    e("s_cbranch_scc1 .Lv9n%=")
    e(".p2align 6")
    e(".Lv9l%=:")
    body(e, a_red, b_red, "FULL")
    e("s_sub_u32 s68, s68, 1")
    e("s_cmp_lg_u32 s68, 0")
    e("s_cbranch_scc1 .Lv9l%=")
    e(".Lv9n%=:")
    body(e, a_red, b_red, "NEXT")
    body(e, a_red, b_red, "LAST")
    e("s_nop 15")
    e("s_nop 15")
    return e.lines


def order_ok(lines):
    """static checks of the emitted stream: (1) every LDS-DMA is preceded by an M0 write with exactly one
    instruction (the wait state) or more in between and no other M0 write after it; (2) SCC producer / consumer
    pairs are adjacent; (3) no s_add / s_sub between s_cmp and s_cselect / s_cbranch"""
    last_m0 = None
    for n, l in enumerate(lines):
        if " m0," in l:
            last_m0 = n
        if "buffer_load" in l:
            assert last_m0 is not None and n - last_m0 >= 2, (n, l)
            last_m0 = None if False else last_m0
        if l.startswith("s_cselect") or l.startswith("s_cbranch_scc"):
            assert lines[n - 1].startswith("s_cmp"), (n, lines[n - 1], l)
    return True


def c_string(lines):
    out = []
    for l in lines:
        if "@SFX@" in l:
            a, b = l.split("@SFX@")
            out.append(f'  "{a}" MK_V9_SFX "{b}\\n\\t"')
        else:
            out.append(f'  "{l}\\n\\t"')
    return " \\\n".join(out)


def acc_read_macros():
    """the epilogue's side: fragment (i, j) out of the accumulator file, 16 registers per statement"""
    out = []
    for i in range(4):
        for j in range(4):
            base = (4 * i + j) * 16
            txt = "".join(f"v_accvgpr_read_b32 %{e}, a{base + e}\\n\\t" for e in range(16))
            outs = ", ".join(f'"=v"((X)[{e}])' for e in range(16))
            out.append(f"#define V9_ACC_READ_{i}_{j}(X) asm volatile(\"{txt}\" : {outs})")
    return "\n".join(out)


def main(path):
    parts = ["// GENERATED by scripts/gen_v9_loop.py -- do not edit (tests/test_v9_gen_cpu.py compares).",
             "// The K loop of gemm_v9 as inline asm, one string per operand layout; MK_V9_SFX = \"bf16\" / \"f16\".",
             "// Register map and placement rule: see the generator."]
    for a_red in (0, 1):
        for b_red in (0, 1):
            for walk in (False, True):
                if (MFMA16 >> (2 * a_red + b_red)) & 1:     # this layout runs on the 16 x 16 x 32 loop below
                    parts.append(f"#define V9_LOOP_TEXT_{a_red}{b_red}{'_W' if walk else ''} \"\"")
                    continue
                lines = loop_text(bool(a_red), bool(b_red), walk)
                order_ok(lines)
                parts.append(f"#define V9_LOOP_TEXT_{a_red}{b_red}{'_W' if walk else ''} \\\n" + c_string(lines))
    if MFMA16:
        parts.append(f"#define V9_MFMA16 {MFMA16}      // bit (2 a_red + b_red): this layout's K loop is the 16 x 16 x 32 one")
        for a_red in (0, 1):
            for b_red in (0, 1):
                if not (MFMA16 >> (2 * a_red + b_red)) & 1:
                    continue
                for walk in (False, True):
                    lines = loop_text16(bool(a_red), bool(b_red), walk)
                    order_ok(lines)
                    parts.append(f"#define V9_LOOP16_TEXT_{a_red}{b_red}{'_W' if walk else ''} \\\n" + c_string(lines))
        clob16 = ", ".join([f'"v{r}"' for r in range(NV16)] + [f'"a{r}"' for r in range(256)] +
                           [f'"s{r}"' for r in range(64, 76)] + ['"scc"', '"memory"'])
        parts.append(f"#define V9_LOOP16_CLOBBERS {clob16}")
        parts.append(acc_read_macros16())
    clob = ", ".join([f'"v{r}"' for r in range(88)] + [f'"a{r}"' for r in range(256)] +
                     [f'"s{r}"' for r in range(64, 76)] + ['"scc"', '"memory"'])
    parts.append(f"#define V9_LOOP_CLOBBERS {clob}")
    parts.append(acc_read_macros())
    txt = "\n".join(parts) + "\n"
    if path == "-":
        sys.stdout.write(txt)
    else:
        with open(path, "w") as f:
            f.write(txt)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for a in sys.argv[1:]:
        if a == "--nodma":
            NO_DMA = True
        elif a == "--noread":
            NO_READ = True
        elif a == "--nobar":
            NO_BAR = True
        elif a.startswith("--mfma16"):
            MFMA16 = int(a.split("=")[1]) if "=" in a else 15
        elif a == "--no-mfma16":
            MFMA16 = 0
        elif a.startswith("--"):
            sys.exit(f"unknown option {a}")
    main(args[0] if args else "macaw_llm_amd/csrc/gemm_v9_loop.inc")
