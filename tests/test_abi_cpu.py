"""The C-ABI shared library loads without a GPU and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from mpas_seaice_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "seaice_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sih_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = declared_symbols()
    assert len(syms) >= 18
    handle = lib.load()
    for s in syms:
        assert hasattr(handle, s), s
    assert sorted(lib.EXPORTS) == [s for s in syms if s in lib.EXPORTS]
    assert set(syms) <= set(lib.EXPORTS) | {"sih_ctx"}


def test_struct_sizes_match_header_layout():
    # the ctypes mirrors must have the natural C layout of the header's structs
    assert ctypes.sizeof(lib.MeshDesc) == 8 * 4 + 4 + 4 + 8 + 18 * 8 + 8
    assert ctypes.sizeof(lib.EvpConfig) == 8 + 15 * 4 + 4 + 8 + 6 * 4 + 4 * 4 + 2 * 8 + 8  # last: one int32 and padding
    assert ctypes.sizeof(lib.IrDesc) == 10 * 8 + 2 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.SeaiceHipError):
        lib.load()


def test_fortran_shims_typecheck():
    """The MPAS-side shim modules (fortran/shim/*.F) pass flang's semantic analysis against the interface
    module of the library and declarations of the framework procedures they call."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fc = "/opt/rocm/lib/llvm/bin/amdflang"
    if not os.path.exists(fc) or shutil.which("make") is None:
        pytest.skip("amdflang or make not available")
    r = subprocess.run(["make", "-C", os.path.join(root, "fortran"), "check-shims"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "error" not in r.stderr.lower(), r.stderr[-2000:]
