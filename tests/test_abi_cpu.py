"""CPU checks of the C-ABI boundary (no kernel is launched): the library builds for gfx950,
loads, exports every symbol include/macaw_hip.h declares, the ctypes mirror of mk_gemm_desc has
the C layout, argument validation returns error codes instead of aborting, and the product ops
refuse CPU tensors (there is no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "macaw_hip.h")


@pytest.fixture(scope="module")
def lib():
    from macaw_llm_amd import build, lib as L
    build.build()
    return L.load()


def declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"^int (mk_[a-z0-9_]+)\(", src, flags=re.M)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from macaw_llm_amd import lib as L
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} not exported by libmacaw_hip.so"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature"
    assert set(L.SIGNATURES) <= set(syms), set(L.SIGNATURES) - set(syms)
    assert lib.mk_abi_version() == L.ABI_VERSION


def test_gemm_desc_layout_matches_c(tmp_path):
    """compile a tiny C program against the public header and compare sizeof/offsetof"""
    from macaw_llm_amd.lib import GemmDesc
    fields = [f[0] for f in GemmDesc._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){',
            'printf("%zu\\n", sizeof(mk_gemm_desc));']
    prog += [f'printf("%zu\\n", offsetof(mk_gemm_desc, {f}));' for f in fields]
    prog += ['return 0;}']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == C.sizeof(GemmDesc)
    for f, off in zip(fields, out[1:]):
        assert getattr(GemmDesc, f).offset == int(off), f


def test_argument_validation_returns_error_codes(lib):
    from macaw_llm_amd.lib import GemmDesc
    assert lib.mk_gemm(None, None) == -1                     # MK_ERR_BAD_ARG
    d = GemmDesc()
    assert lib.mk_gemm(C.byref(d), None) == -1               # null operands
    assert lib.mk_rmsnorm_fwd(None, None, None, None, None, None, 4, 8, 1e-6, 1, None) == -1
    assert lib.mk_softmax_fwd(None, None, None, None, 1, 1, 1, 1, 1, 0, 0.0, 0, 1, None) == -1
    assert lib.mk_cast(None, 0, None, 1, 10, None) == -1
    assert lib.mk_adamw(None, None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 1, None) == -1


def test_product_ops_refuse_cpu_tensors():
    from macaw_llm_amd import ops
    from macaw_llm_amd.lib import MacawHipError
    x = torch.randn(4, 8)
    with pytest.raises(MacawHipError):
        ops.linear_fwd(x, torch.randn(3, 8))
    with pytest.raises(MacawHipError):
        ops.rmsnorm_fwd(x, torch.ones(8), 1e-6)


def test_product_path_never_imports_the_oracle():
    """the oracle is test infrastructure: importing the product package must not pull it in"""
    code = ("import sys; import macaw_llm_amd.modeling, macaw_llm_amd.engine, macaw_llm_amd.ops, "
            "macaw_llm_amd.bucketed, macaw_llm_amd.train, macaw_llm_amd.optim, macaw_llm_amd.factory, macaw_llm_amd.preprocess; "
            "bad=[m for m in sys.modules if m=='oracle' or m.startswith('oracle.')]; print(bad); "
            "sys.exit(1 if bad else 0)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    for fn in os.listdir(os.path.join(ROOT, "macaw_llm_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "macaw_llm_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from macaw_llm_amd import lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", tmp_path / "nope.so")
    with pytest.raises(L.MacawHipError):
        L.load()
