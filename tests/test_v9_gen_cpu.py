"""CPU checks of the generated K loop of gemm_v9 (scripts/gen_v9_loop.py -> macaw_llm_amd/csrc/gemm_v9_loop.inc):

  * the committed .inc is what the generator writes;
  * the addressing identities the asm relies on (voffsets of a wave's pieces = piece 0 + strides, read addresses
    of k-steps / fragments = address 0 XOR a constant) hold against the C++ formulas of gemm_lds_image.inc;
  * a single-wave SIMULATION of the instruction stream for several K-tile counts: every MFMA consumes fragments
    of the tile and k-step it is supposed to, read from an LDS slot whose LDS-DMA had been waited for (counted
    vmcnt) and published by a barrier, every LDS-DMA is issued into a slot only after a barrier that follows the
    last read of its previous content, every piece of every K-tile is requested exactly once, and the scalar
    protocol (M0 one instruction ahead of its load, SCC pairs adjacent) is respected.

Round 6: the shipped loops run on v_mfma_f32_16x16x32 (gen.loop_text16: 128 MFMAs per K-tile in four quarters, new
fragment addressing, one more swizzle bit in the reduction-major image); the identities and the simulation are repeated for
them at the end of this file.  The round-5 loops (gen.loop_text, `--no-mfma16`) stay in the generator and stay tested.

No GPU involved: the numerics are covered by tests/test_kernels_gpu.py::test_gemm_v9_*."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_v9_loop", os.path.join(ROOT, "scripts", "gen_v9_loop.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def test_committed_inc_is_generated():
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gen.main("-")
    with open(os.path.join(ROOT, "macaw_llm_amd", "csrc", "gemm_v9_loop.inc")) as f:
        assert f.read() == buf.getvalue(), "run scripts/gen_v9_loop.py"


# ---- the C++ formulas (gemm_lds_image.inc), restated ------------------------------------------------------------
def v7_voffset(red, row0, R, ld, half, p, l):
    if not red:
        r = p * 8 + (l >> 3)
        kc = (l & 7) ^ ((r >> 1) & 7)
        gr = min(row0 + half * 128 + r, R - 1) - row0
        return gr * ld * 2 + kc * 16
    kr = p * 4 + (l >> 4)
    mc = (l & 15) ^ (4 * (kr & 3))
    return kr * ld * 2 + (half * 128 + mc * 8) * 2


def v7_read_offsets(red, l):
    off = []
    if not red:
        row = l & 31
        for ks in range(4):
            kc = ks * 2 + (l >> 5)
            off.append(row * 128 + ((kc ^ ((row >> 1) & 7)) << 4))
    else:
        li = l & 15
        kr = 8 * (l >> 5) + (li >> 2)
        for f in range(4):
            col = f * 32 + 16 * ((l >> 4) & 1) + 4 * (li & 3)
            off.append(kr * 256 + (((col >> 3) ^ (4 * (kr & 3))) << 4) + ((col & 7) << 1))
    return off


@pytest.mark.parametrize("red", [False, True])
def test_addressing_identities(red):
    ld = 4608
    for w in range(4):
        for l in range(64):
            base = v7_voffset(red, 512, 4096, ld, 0, w, l)
            for h in range(2):
                for i in range(4):
                    want = v7_voffset(red, 512, 4096, ld, h, w + 4 * i, l)
                    got = base + ((128 * h + 32 * i) * ld * 2 if not red else 16 * i * ld * 2 + 256 * h)
                    assert want == got
    for l in range(64):
        off = v7_read_offsets(red, l)
        for k in range(4):
            assert off[k] == off[0] ^ (k << (6 if red else 5))


# ---- single-wave simulation ----------------------------------------------------------------------------------
HALF, SLOT, B_BASE = 16384, 32768, 98304


class Sim:
    def __init__(self, a_red, b_red, nk, w=0, m16=False):
        self.a_red, self.b_red, self.nk, self.w = a_red, b_red, nk, w
        # register map of the stream being simulated: round 5's 32 x 32 x 16 loop or the 16 x 16 x 32 one
        self.m16 = m16
        self.ada, self.adb = (gen.ADA16, gen.ADB16) if m16 else (gen.ADA, gen.ADB)
        self.voa, self.vob = (gen.VOA16, gen.VOB16) if m16 else (gen.VOA, gen.VOB)
        self.stA = 4096 if a_red else 128        # bytes per K-tile (any distinct positive numbers do)
        self.stB = 8192 if b_red else 128
        self.s = {}
        self.v = {}
        self.m0 = None
        self.m0_age = 99
        self.scc = None
        self.scc_fresh = False
        self.t = 0
        self.barriers = []                       # times
        # LDS halves: key = half-slot index (address // 16384); value = dict(op, tile, half, landed_t, pub)
        self.lds = {}
        self.vmq = []                            # outstanding LDS-DMA pieces, oldest first
        self.lgkm = []                           # outstanding ds_reads (register ranges)
        self.regs = {}                           # first VGPR of a 2- or 4-register fragment part -> tag
        self.last_read = {}                      # half-slot -> time of the last ds_read
        self.requested = {}
        self.mfma_log = []
        # what the C++ prologue has issued: tile 0 (A, B interleaved per piece), A(1), B(1) half 0
        for h in range(2):
            for i in range(4):
                self._dma("A", 0 * SLOT + h * HALF + (w + 4 * i) * 1024, 0, h, i)
                self._dma("B", B_BASE + h * HALF + (w + 4 * i) * 1024, 0, h, i)
        for h in range(2):
            for i in range(4):
                self._dma("A", SLOT + h * HALF + (w + 4 * i) * 1024, 1, h, i)
        for i in range(4):
            self._dma("B", B_BASE + SLOT + (w + 4 * i) * 1024, 1, 0, i)

    def _dma(self, op, lds_addr, tile, half, piece):
        hs = lds_addr // HALF
        assert 0 <= tile < self.nk, ("request of a K-tile that does not exist", op, tile)
        key = (op, tile, half, piece)
        assert key not in self.requested, ("piece requested twice", key)
        self.requested[key] = self.t
        assert (lds_addr % HALF) // 1024 == self.w + 4 * piece and (lds_addr % 1024) == 0
        assert (op == "A") == (lds_addr < B_BASE) and lds_addr < B_BASE + 2 * SLOT
        assert ((lds_addr % SLOT) // HALF) == half
        # WAR: the previous content of this half-slot was last read before a barrier that precedes this request
        lr = self.last_read.get(hs)
        if lr is not None:
            assert any(lr < b < self.t or b == self.t for b in self.barriers if b > lr), ("WAR", op, tile, half, piece)
        self.vmq.append(dict(op=op, hs=hs, tile=tile, half=half, piece=piece))

    def val(self, tok):
        tok = tok.strip()
        if tok.startswith("s") and tok[1:].isdigit():
            return self.s[int(tok[1:])]
        if tok == "m0":
            return self.m0
        return int(tok, 0)

    def run(self, lines):
        labels = {l[:-1]: n for n, l in enumerate(lines) if l.endswith(":")}
        pc = 0
        steps = 0
        while pc < len(lines):
            steps += 1
            assert steps < 200000
            l = lines[pc]
            pc += 1
            self.t += 1
            self.m0_age += 1
            if l.endswith(":") or l.startswith(".p2align") or l.startswith("s_nop") or l.startswith("v_accvgpr_write"):
                continue
            op, _, rest = l.partition(" ")
            a = [x.strip() for x in rest.split(",")] if rest else []
            fresh = self.scc_fresh
            self.scc_fresh = False
            if op in ("s_mov_b32", "s_add_u32", "s_sub_u32", "s_xor_b32", "s_lshl_b32"):
                if op == "s_mov_b32":
                    r = self.val(a[1])
                else:
                    x, y = self.val(a[1]), self.val(a[2])
                    r = {"s_add_u32": x + y, "s_sub_u32": x - y, "s_xor_b32": x ^ y, "s_lshl_b32": x << y}[op]
                r &= 0xffffffff
                if a[0] == "m0":
                    self.m0, self.m0_age = r, 0
                else:
                    self.s[int(a[0][1:])] = r
            elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32"):
                x, y = self.val(a[0]), self.val(a[1])
                self.scc = (x == y) if op == "s_cmp_eq_u32" else (x != y)
                self.scc_fresh = True
            elif op == "s_cselect_b32":
                assert fresh, "SCC consumer not adjacent to its s_cmp"
                self.s[int(a[0][1:])] = self.val(a[1]) if self.scc else self.val(a[2])
            elif op == "s_cbranch_scc1":
                assert fresh
                if self.scc:
                    pc = labels[a[0]]
            elif op in ("v_mov_b32", "v_add_u32", "v_xor_b32"):
                if op == "v_mov_b32":
                    r = self.vval(a[1])
                else:
                    x, y = self.vval(a[1]), self.vval(a[2])
                    r = (x + y) if op == "v_add_u32" else (x ^ y)
                self.v[int(a[0][1:])] = r & 0xffffffff
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    n = int(m.group(1))
                    while len(self.vmq) > n:
                        p = self.vmq.pop(0)
                        e = self.lds.setdefault(p["hs"], dict(op=None, tile=None, pieces=set(), landed=None))
                        if (e["op"], e["tile"]) != (p["op"], p["tile"]):
                            e.update(op=p["op"], tile=p["tile"], pieces=set(), landed=None)
                        e["pieces"].add(p["piece"])
                        if len(e["pieces"]) == 4:
                            e["landed"] = self.t
                if "lgkmcnt(0)" in rest:
                    for r0, n_, tag in self.lgkm:
                        self.regs[r0] = (n_, tag)
                    self.lgkm = []
            elif op == "s_barrier":
                self.barriers.append(self.t)
            elif op in ("ds_read_b128", "ds_read_b64_tr_b16"):
                m = re.match(r"v\[(\d+):(\d+)\]", a[0])
                r0, r1 = int(m.group(1)), int(m.group(2))
                addr_reg = int(a[1].split()[0][1:])
                off = int(a[1].split("offset:")[1]) if "offset:" in a[1] else 0
                addr = self.v[addr_reg] + off
                hs = addr // HALF
                e = self.lds.get(hs)
                assert e is not None and e["landed"] is not None, ("read of a half-slot that never landed", l)
                # RAW across waves: a barrier lies between this wave's wait for the pieces and the read
                assert any(e["landed"] <= b <= self.t for b in self.barriers), ("RAW: no barrier after the wait", l)
                self.last_read[hs] = self.t
                is_a = addr < B_BASE
                red = self.a_red if is_a else self.b_red
                which = addr_reg - (self.ada if is_a else self.adb)
                if self.m16:
                    # K-major: address register = k32-step, fragments + 2048; reduction-major: address register =
                    # fragment, k32-steps + 8192, the second transposing read + 1024
                    assert 0 <= which < (8 if red else 2), l
                    if not red:
                        ks, frag, part = which, off // 2048, 0
                        assert off % 2048 == 0 and frag < 8
                    else:
                        frag, ks, part = which, off // 8192, off % 8192
                        assert part in (0, 1024) and ks < 2
                    tag = ("A" if is_a else "B", e["tile"], ks, frag, part)
                elif not red:
                    tag = ("A" if is_a else "B", e["tile"], which, off // 4096, 0)
                else:
                    tag = ("A" if is_a else "B", e["tile"], off // 4096, which, off % 4096)
                for rr in range(r0, r1 + 1):
                    self.regs.pop(rr, None)
                self.lgkm.append((r0, r1 - r0 + 1, tag))
            elif op == "buffer_load_dwordx4":
                assert "lds" in l and self.m0_age >= 2, ("M0 written right in front of its LDS-DMA", l)
                vo = int(a[0][1:])
                is_a = "%[rsA]" in a[1] or a[1] == "RSA"
                base = self.voa if is_a else self.vob
                assert base <= vo < base + 8 and (vo < self.vob) == is_a
                h, i = (vo - base) // 4, (vo - base) % 4
                soff = self.val(a[2].split()[0])
                st = self.stA if is_a else self.stB
                assert soff % st == 0
                self._dma("A" if is_a else "B", self.m0, soff // st, h, i)
            elif op.startswith("v_mfma"):
                m = re.findall(r"\[(\d+):(\d+)\]", l)
                acc, fb, fa = int(m[0][0]), int(m[1][0]), int(m[2][0])
                parts_a = self.frag(fa, self.a_red)
                parts_b = self.frag(fb, self.b_red)
                self.mfma_log.append((acc // (4 if self.m16 else 16), parts_a, parts_b))
            else:
                raise AssertionError(f"unmodelled instruction: {l}")

    def vval(self, tok):
        tok = tok.strip()
        if tok.startswith("v") and tok[1:].isdigit():
            return self.v[int(tok[1:])]
        return self.val(tok)

    def frag(self, r0, red):
        """the (operand, tile, k-step, fragment) a 4-register MFMA operand holds; every part must have arrived"""
        if not red:
            assert r0 in self.regs and self.regs[r0][0] == 4, ("operand not in registers", r0)
            return self.regs[r0][1][:4]
        lo, hi = self.regs.get(r0), self.regs.get(r0 + 2)
        assert lo and hi and lo[0] == 2 and hi[0] == 2, ("operand not in registers", r0)
        assert lo[1][:4] == hi[1][:4] and lo[1][4] + 1024 == hi[1][4]
        return lo[1][:4]


def _lines(a_red, b_red, walk=False):
    out = []
    for l in gen.loop_text(a_red, b_red, walk):
        for k, v in (("%[stA]", None), ("%[stB]", None)):
            pass
        out.append(l.replace("@SFX@", "bf16").replace("%=", "X"))
    return out


@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("nk", [2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("w", [0, 1, 2, 3])
@pytest.mark.parametrize("walk", [False, True])
def test_simulated_stream_is_consistent(a_red, b_red, nk, w, walk):
    sim = Sim(a_red, b_red, nk, w)
    wr, wc = w >> 1, w & 1                       # this wave reads A half wr and B half wc
    sub = {"%[stA]": str(sim.stA), "%[stB]": str(sim.stB), "%[nk]": str(nk), "%[wv]": str(w * 1024),
           "%[iA]": "1000000", "%[iB]": "2000000", "%[voA]": "0", "%[voB]": "0", "%[adA]": str(wr * HALF),
           "%[adB]": str(B_BASE + wc * HALF), "%[rsB]": "RSB", "%[rsA]": "RSA"}
    lines = []
    for l in _lines(a_red, b_red, walk):
        for k, v in sub.items():
            l = l.replace(k, v)
        lines.append(l)
    sim.run(lines)
    # every piece of every K-tile of both operands was requested exactly once
    want = {(op, t, h, i) for op in "AB" for t in range(nk) for h in range(2) for i in range(4)}
    assert set(sim.requested) == want
    assert not sim.vmq or all(False for _ in ()), "pieces still in flight at the end"
    # 64 MFMAs per K-tile: tile T, k-step ks, accumulator (i, j) <- A fragment i x B fragment j, each exactly once
    assert len(sim.mfma_log) == 64 * nk
    seen = set()
    for n, (acc, pa, pb) in enumerate(sim.mfma_log):
        T, ks = n // 64, (n % 64) // 16
        i, j = acc // 4, acc % 4
        assert pa == ("A", T, ks, i) and pb == ("B", T, ks, j), (n, acc, pa, pb)
        seen.add((T, ks, i, j))
    assert len(seen) == 64 * nk


# ---- the 16 x 16 x 32 loop (round 6) -----------------------------------------------------------------------------
def v9_voffset16_red(row0, R, ld, half, p, l):
    kr = p * 4 + (l >> 4)
    mc = (l & 15) ^ (4 * (kr & 3)) ^ (2 * ((kr >> 3) & 1))
    return kr * ld * 2 + (half * 128 + mc * 8) * 2


def v9_read_offset16(red, l):
    if not red:
        row = l & 15
        return row * 128 + (((l >> 4) ^ ((row >> 1) & 7)) << 4)
    g, li = l >> 4, l & 15
    t, c = li >> 2, li & 3
    return (8 * g + t) * 256 + ((((4 * t) ^ (2 * (g & 1)) ^ (c >> 1))) << 4) + ((c & 1) << 3)


def test_addressing_identities_of_the_16x16x32_loop():
    """gemm_v9_impl.inc v9_voffset<true, true> / v9_read_offset16 against what the generated asm assumes: a wave's LDS-DMA
    voffsets = piece 0's + strides (the extra swizzle bit of the reduction-major image is the same for every piece of a
    wave); a K-major fragment f of k32-step h is read at (address ^ (h << 6)) + 2048 f and holds rows 16 f + (l & 15),
    16-byte chunk 4 h + (l >> 4); the two transposing reads of a reduction-major fragment f = 2 a + b at
    (address ^ ((a << 6) | (b << 5))) + 8192 h (+ 1024) hold k-rows 32 h + 8 (l >> 4) + ((l & 15) >> 2) (+ 4) and columns
    16 f + 4 (l & 3) .. + 3 of the image the LDS-DMA wrote; and the 32 lanes of a half-wave hit 64 different banks."""
    ld = 4608
    for w in range(4):
        for l in range(64):
            base = v9_voffset16_red(512, 4096, ld, 0, w, l)
            for h in range(2):
                for i in range(4):
                    assert v9_voffset16_red(512, 4096, ld, h, w + 4 * i, l) == base + 16 * i * ld * 2 + 256 * h
    for h in range(2):
        for f in range(8):
            for l in range(64):
                addr = (v9_read_offset16(False, l) ^ (h << 6)) + 2048 * f
                row, chunk = addr // 128, (addr % 128) // 16
                assert row == 16 * f + (l & 15) and chunk ^ ((row >> 1) & 7) == 4 * h + (l >> 4)
            a, b = f >> 1, f & 1
            for part in (0, 1024):
                addrs = []
                for l in range(64):
                    g, li = l >> 4, l & 15
                    addr = (v9_read_offset16(True, l) ^ ((a << 6) | (b << 5))) + 8192 * h + part
                    kr, chunk, byte = addr // 256, (addr % 256) // 16, addr % 16
                    # the image: LDS (k-row, chunk) holds global chunk  chunk ^ 4 (kr & 3) ^ 2 ((kr >> 3) & 1)
                    col = (chunk ^ (4 * (kr & 3)) ^ (2 * ((kr >> 3) & 1))) * 8 + byte // 2
                    assert kr == 32 * h + 8 * g + (li >> 2) + part // 256 and col == 16 * f + 4 * (li & 3)
                    addrs.append(addr)
                for half in range(2):
                    banks = [((x // 4) + d) % 64 for x in addrs[32 * half:32 * half + 32] for d in range(2)]
                    assert len(set(banks)) == 64


def _lines16(a_red, b_red, walk=False):
    return [l.replace("@SFX@", "bf16").replace("%=", "X") for l in gen.loop_text16(a_red, b_red, walk)]


@pytest.mark.parametrize("a_red,b_red", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("nk", [2, 3, 4, 5, 8, 9])
@pytest.mark.parametrize("w", [0, 1, 2, 3])
@pytest.mark.parametrize("walk", [False, True])
def test_simulated_stream_of_the_16x16x32_loop_is_consistent(a_red, b_red, nk, w, walk):
    """the same single-wave simulation for the shipped loop: 128 MFMAs per K-tile -- tile T, k32-step h, accumulator (i, j)
    <- A fragment i x B fragment j of exactly that tile and step, each once -- every fragment read from a half-slot whose
    LDS-DMA was waited for and published by a barrier, every LDS-DMA into a slot only behind a barrier that follows the
    last read of its previous content, every piece of every K-tile requested exactly once."""
    sim = Sim(a_red, b_red, nk, w, m16=True)
    wr, wc = w >> 1, w & 1
    sub = {"%[stA]": str(sim.stA), "%[stB]": str(sim.stB), "%[nk]": str(nk), "%[wv]": str(w * 1024),
           "%[iA]": "1000000", "%[iB]": "2000000", "%[voA]": "0", "%[voB]": "0", "%[adA]": str(wr * HALF),
           "%[adB]": str(B_BASE + wc * HALF), "%[rsB]": "RSB", "%[rsA]": "RSA"}
    lines = []
    for l in _lines16(a_red, b_red, walk):
        for k, v in sub.items():
            l = l.replace(k, v)
        lines.append(l)
    sim.run(lines)
    want = {(op, t, h, i) for op in "AB" for t in range(nk) for h in range(2) for i in range(4)}
    assert set(sim.requested) == want
    assert len(sim.mfma_log) == 128 * nk
    seen = set()
    for n, (acc, pa, pb) in enumerate(sim.mfma_log):
        T, h = n // 128, (n % 128) // 64
        i, j = acc // 8, acc % 8
        assert pa == ("A", T, h, i) and pb == ("B", T, h, j), (n, acc, pa, pb)
        seen.add((T, h, i, j))
    assert len(seen) == 128 * nk
