"""The C-ABI library builds, loads without a GPU and exports every symbol include/perphil_hip.h
declares.  No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "perphil_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pph_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_binding_list():
    from perphil_amd import _ffi

    assert _declared() == sorted(_ffi.EXPORTS)


def test_library_exports_every_declared_symbol():
    from perphil_amd import _ffi

    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} missing from libperphil_hip.so"


def test_struct_layouts_match_header():
    from perphil_amd import _ffi

    # pph_solver_cfg: 4 int32, 2 double, 4 int32, 4 double, 2 int32, 1 double ; pph_solve_info: 4 int32, 2 double
    assert ctypes.sizeof(_ffi.SolverCfg) == 4 * 4 + 2 * 8 + 4 * 4 + 4 * 8 + 2 * 4 + 8 + 2 * 4
    assert _ffi.SolverCfg.inner_reduction.offset == 88
    assert _ffi.SolverCfg.inner_norm.offset == 96
    assert ctypes.sizeof(_ffi.SolveInfo) == 4 * 4 + 2 * 8
    assert _ffi.SolverCfg.rtol.offset == 16 and _ffi.SolverCfg.inner_rtol.offset == 48
    assert _ffi.SolverCfg.picard_max_it.offset == 80


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device the context constructor raises; nothing computes on the CPU instead."""
    from perphil_amd import _ffi

    have_gpu = os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)
    if have_gpu:
        pytest.skip("a GPU is present: covered by the gpu-marked tests")
    with pytest.raises(RuntimeError):
        _ffi.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "perphil_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                # condition numbers are host-side analysis in the reference too (SciPy svds / eigsh,
                # src/perphil/solvers/conditioning.py:155-218): conditioning.py may use them, nothing else may
                if f != "conditioning.py":
                    assert "scipy.sparse.linalg" not in src, f"{f}: CPU solver in the product path"
