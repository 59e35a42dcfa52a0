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


def test_advection_shim_keeps_the_reference_abort_convention():
    """A forced run-time abort of the remap (the library call replaced by one that returns SIH_ERR_IR) takes the
    advection shim through the reference's sequence (mpas_seaice_advection_incremental_remap.F:3182-3183,
    mpas_seaice_error.F:44-115): the message as an error, seaice_critical_error_write_block for the failing block,
    seaice_check_critical_error -- the vote every rank takes part in -- and only inside it the critical stop."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fc = "/opt/rocm/lib/llvm/bin/amdflang"
    if not os.path.exists(fc) or shutil.which("make") is None or shutil.which("gcc") is None:
        pytest.skip("amdflang, gcc or make not available")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run(["make", "-C", os.path.join(root, "fortran"), "check-abort-path"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln.strip() for ln in r.stdout.splitlines()]
    want = ["mpas_log_write type 3: IR: Negative mass in IR",
            "seaice_critical_error_write_block abortFlag=T",
            "seaice_check_critical_error (vote of all ranks) abortFlag=T",
            "mpas_log_write type 4: Runtime error"]
    pos = [lines.index(w) for w in want]
    assert pos == sorted(pos), lines
    assert "seaice_run_advection returned" not in r.stdout  # the failing rank stops inside the vote routine
