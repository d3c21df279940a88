"""Build the C-ABI shared library `libmacaw_hip.so` for gfx950 with hipcc.

A plain `.so` (no torch ABI coupling: torch here is built against ROCm 7.0, the
system is ROCm 7.2) loaded through ctypes.  Built in-tree so it travels to the
GPU box with the repo snapshot.  `python -m macaw_llm_amd.build` rebuilds.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "csrc" / "_obj"
LIB = PKG / "libmacaw_hip.so"
ARCH = "gfx950"
# MK_EXPERIMENTS=1 also builds the kernels that lost their A/B and stay in the tree as measured experiments:
# gemm_v8.hip (the 4-wave 256 x 256 GEMM scheduled by hipcc: superseded by gemm_v9's hand-placed loop; `mk_gemm_set_cfg(14)`
# falls back to v7 without it, tests/test_kernels_gpu.py::test_gemm_v8_* skip) and flash_fwd4x64_kernel (attention forward on one
# wave per SIMD: loses to the 8-wave kernel, profiles/r05_attn8.txt F; MK_ATTN_FWD4X64=1 selects it in such a build)
EXPERIMENTS = bool(os.environ.get("MK_EXPERIMENTS"))
SOURCES = ["gemm.hip", "gemm_v7.hip", *(["gemm_v8.hip"] if EXPERIMENTS else []), "gemm_v9.hip", "norm.hip",
           "elementwise.hip", "softmax.hip", "attention.hip", "decode.hip", "preprocess.hip"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *os.environ.get("MK_EXTRA_FLAGS", "").split(),
         *(["-DMK_WITH_V8", "-DMK_WITH_FWD4X64"] if EXPERIMENTS else []), "-Wno-unused-result"]


# per-source flags.  gemm_v8.hip: the LDS-transposed epilogue of a 4 x 4-fragment wave tile exceeds LLVM's
# default `#pragma unroll` size budget; left rolled, the fragment-row loop indexes the 256 accumulator
# registers dynamically and the whole accumulator goes through scratch memory.
FILE_FLAGS = {"gemm_v8.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
              "gemm_v9.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: cannot build libmacaw_hip.so")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = [CSRC / s for s in SOURCES]
    deps = srcs + [CSRC / "common.h", CSRC / "gemm_common.h", PKG.parent / "include" / "macaw_hip.h",
                   *sorted(CSRC.glob("*.inc"))]
    stamp = OBJ / "stamp.txt"
    dig = _digest(deps)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ.mkdir(parents=True, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(src.name, []), "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    stamp.write_text(dig)
    if verbose:
        print(f"[macaw_llm_amd.build] built {LIB} ({LIB.stat().st_size/1e6:.1f} MB)", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
