"""Per-kernel resource usage (VGPRs, spills, scratch, static LDS) of the gfx950 code objects inside the
built library, read from the library itself: the .hip_fatbin section is a sequence of clang offload
bundles, each holding one code object per target; its AMDGPU metadata note lists every kernel.

    python scripts/kernel_resources.py [pattern]     # table, optionally filtered by a substring

tests/test_kernel_resources_cpu.py pins the occupancy-relevant budgets with this (a run-time branch
added to the GEMM epilogue in round 3 cost every 128x128 kernel its fourth wave per SIMD -- 125 -> 136
VGPRs, -12 ... -24 % on the shapes that use it -- and nothing but a profile showed it)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "macaw_llm_amd", "libmacaw_hip.so")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(lib=LIB, arch="gfx950"):
    """yield the bytes of every `arch` code object bundled into `lib`"""
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, lib],
                       check=True)
        data = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, data)] + [len(data)]
    for a, b in zip(starts, starts[1:]):
        blob = data[a:b]
        (n,) = struct.unpack_from("<Q", blob, len(magic))
        p = len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            p += 24
            target = blob[p:p + idlen].decode()
            p += idlen
            if arch in target and size:
                yield blob[off:off + size]


def kernels(lib=LIB):
    """{demangled kernel name: {field: int}} over all code objects of the library"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib)):
            path = os.path.join(td, f"co{i}.o")
            open(path, "wb").write(co)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", path], check=True,
                                   capture_output=True, text=True).stdout
            name, cur = None, {}
            for line in notes.split("\n"):
                m = re.match(r"\s+\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s+\.(\w+):\s+(\d+)\s*$", line)
                if m and m.group(1) in FIELDS:
                    cur[m.group(1)] = int(m.group(2))
                if re.match(r"\s+\.wavefront_size:", line) and name:
                    out[name] = cur
                    name, cur = None, {}
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d.replace("(anonymous namespace)::", ""): out[n] for n, d in zip(names, dem)}


def disassemble(substr, lib=LIB):
    """{mangled symbol: [instruction lines]} for every function of the library whose mangled name contains `substr`
    (llvm-objdump -d of the code objects that define such a symbol)"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib)):
            if substr.encode() not in co:
                continue
            path = os.path.join(td, f"co{i}.o")
            open(path, "wb").write(co)
            txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", path], check=True,
                                 capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1) if substr in m.group(1) else None
                    if cur:
                        out[cur] = []
                elif cur and line.strip():
                    out[cur].append(line.split("//")[0].strip())
    return out


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    ks = kernels()
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'scratch':>7} {'lds':>7}  kernel")
    for k in sorted(ks):
        if pat in k:
            v = ks[k]
            print(f"{v.get('vgpr_count', 0):5d} {v.get('agpr_count', 0):5d} {v.get('sgpr_count', 0):5d} "
                  f"{v.get('vgpr_spill_count', 0):6d} {v.get('private_segment_fixed_size', 0):7d} "
                  f"{v.get('group_segment_fixed_size', 0):7d}  {k[:150]}")
