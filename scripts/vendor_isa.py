"""STUDY TOOL (nothing here is linked or shipped): per-K-tile instruction mix of the bf16 gfx950 kernels that hipBLASLt's
heuristic selects for the LLaMA-7B step shapes, next to gemm_v9's generated loop (VERDICT r5 "Next" item 3a).

    scripts/gpu_call_r06.sh a vendor          # GPU: which kernel per shape (rocprofv3 kernel trace of the harness, cfg 100)
    python scripts/vendor_isa.py gpurun_out/r06/a/vendor_kernels_by_grid.txt > profiles/r06_vendor_isa.txt

The vendor code objects are compressed offload bundles (CCOB): clang-offload-bundler unbundles them; kernels carry size-0
symbols, so a kernel's text is [its address, the next FUNC symbol).  The main loop of a Tensile kernel is the text between
label_LoopBeginL_0 and label_LoopBeginL_1 (PGR2 kernels unroll two K-tiles: _0 / _1) or LoopBeginL .. LoopEndL."""
import bisect
import collections
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIBDIR = "/opt/rocm/lib/hipblaslt/library"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def unbundle(contraction, td):
    src = os.path.join(LIBDIR, f"TensileLibrary_BB_BB_HA_Bias_SAV_UA_Type_BB_HPA_Contraction_l_{contraction}_Cijk_Dijk_gfx950.co")
    dst = os.path.join(td, contraction + ".co")
    if not os.path.exists(dst):
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={src}", f"--output={dst}"], check=True)
    return dst


def kernel_text(co, name):
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], check=True, capture_output=True, text=True).stdout
    rows = sorted((int(f[1], 16), f[7]) for f in (l.split() for l in syms.splitlines()) if len(f) >= 8 and f[3] == "FUNC")
    addrs = [a for a, _ in rows]
    a = next(a for a, n in rows if n == name)
    b = addrs[bisect.bisect_right(addrs, a)]
    out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f"--start-address={a}",
                          f"--stop-address={b}", co], check=True, capture_output=True, text=True).stdout
    return out.splitlines(), b - a


def loop_body(lines):
    lab = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^[0-9a-f]+ <(label_\w+)>:", l))}
    if "label_LoopBeginL_0" in lab and "label_LoopBeginL_1" in lab:
        return lines[lab["label_LoopBeginL_0"]:lab["label_LoopBeginL_1"]], "LoopBeginL_0 .. LoopBeginL_1 (one of two unrolled K-tiles)"
    return lines[lab["label_LoopBeginL"]:lab["label_LoopEndL"]], "LoopBeginL .. LoopEndL"


def classify(op, ins):
    if op.startswith("v_mfma"):
        return op
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return op
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return op
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return op + (" ... lds" if ins.rstrip().endswith("lds") else "")
    if op == "s_barrier" or op == "s_waitcnt":
        return op
    if op.startswith("v_"):
        return "VALU:" + op
    if op.startswith("s_"):
        return "SALU / branch"
    return op


def mix(body):
    c = collections.Counter()
    for l in body:
        l = l.split("//")[0].strip()
        if not l or l.endswith(":") or re.match(r"^[0-9a-f]+ <", l):
            continue
        c[classify(l.split()[0], l)] += 1
    return c


def v9_mix(layout):
    """the same classes for gemm_v9's generated K loop: the steady-state K-tile of V9_LOOP_TEXT_<layout> = the text between
    its labels .Lv9l (loop head) and .Lv9n (the last tiles)"""
    txt = open(os.path.join(ROOT, "macaw_llm_amd", "csrc", "gemm_v9_loop.inc")).read()
    blk = txt[txt.index(f"#define V9_LOOP_TEXT_{layout} \\"):]
    blk = blk[:blk.index("\n#define", 10)].replace('" MK_V9_SFX "', "bf16")
    ins = [s.strip() for s in re.findall(r'"((?:[^"\\]|\\[^n])*?)\\n\\t"', blk)]
    lo, hi = ins.index(".Lv9l%=:"), ins.index(".Lv9n%=:")
    c = collections.Counter()
    for s in ins[lo + 1:hi]:
        if s and not s.endswith(":"):
            c[classify(s.split()[0], s)] += 1
    return c, hi - lo - 1, len(ins)


def fmt(c):
    keys = sorted(c, key=lambda k: (not k.startswith("v_mfma"), not k.startswith("ds_"), not k.startswith("buffer"), k))
    return "\n".join(f"    {c[k]:5d}  {k}" for k in keys)


def main():
    picked = collections.OrderedDict()
    for l in open(sys.argv[1]):
        f = l.split()
        picked.setdefault(f[5], []).append((int(f[0]), float(f[1]), f[2], f[3], f[4]))
    print(__doc__.split("\n\n")[0])
    print()
    with tempfile.TemporaryDirectory() as td:
        for name, uses in picked.items():
            contraction = re.match(r"Cijk_(\w+?_\w+?)_BBS", name).group(1)
            layout = {"Alik_Bljk": "NT (forward: both operands K-major)", "Ailk_Bljk": "NN (grad-input: A K-major, B reduction-major)",
                      "Ailk_Bjlk": "TT (grad-weight: both reduction-major)"}[contraction]
            co = unbundle(contraction, td)
            lines, size = kernel_text(co, name)
            body, where = loop_body(lines)
            c = mix(body)
            p = dict(re.findall(r"_([A-Z]+[a-z]?)(\d[\dx_]*)", name))
            mt = re.search(r"_MT(\d+x\d+x\d+)", name).group(1)
            wt = re.search(r"_MIWT(\d+_\d+)", name).group(1)
            wg = re.search(r"_WG(\d+_\d+)_", name).group(1)
            print(f"== vendor {layout}: macro tile {mt}, MFMA 16x16x32 tiles per wave {wt}, workgroup {wg} "
                  f"({'direct-to-LDS' if 'DTLA1' in name else 'register-staged global loads'}; "
                  f"{'custom main-loop schedule (CMS); ' if '_CMS_' in name else ''}code {size} B)")
            for n, us, grid, wgs, lds in uses:
                print(f"   launched {n} x: grid {int(grid) // int(wgs)} workgroups of {wgs} threads, LDS {lds} B, {us} us per launch")
            print(f"   main loop = {where}: {sum(c.values())} instructions per K-tile (64) and wave")
            print(fmt(c))
            print(f"   full name: {name}")
            print()
    for lay, what in (("00", "NT"), ("01", "NN"), ("11", "TT")):
        c, n, tot = v9_mix(lay)
        print(f"== gemm_v9 {what} (macaw_llm_amd/csrc/gemm_v9_loop.inc V9_LOOP_TEXT_{lay}: macro tile 256x256x64, 4 waves, wave tile "
              f"128x128, MFMA 32x32x16; steady-state K-tile .Lv9l .. .Lv9n = {n} instructions per K-tile and wave, whole asm "
              f"block {tot})")
        print(fmt(c))
        print()


if __name__ == "__main__":
    main()
