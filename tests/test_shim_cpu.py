"""The case file of the shim harness, without a GPU: fortran/harness/shim_harness must read every pool of a domain
written by fortran_case.write_pools, find what seaice_mesh_pool_create asks its pools for, and stop with the framework's
critical-error convention at the first call that needs a device."""
import os
import subprocess

import numpy as np
import pytest

from mpas_seaice_amd import cases, fortran_case
from util import get_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "harness", "shim_harness")


def test_case_file_round_trip_and_device_less_stop(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: tests/test_shim_gpu.py runs the harness to the end")
    if not os.path.exists(EXE):
        pytest.fail("fortran/harness/shim_harness is not built (run __graft_entry__.build())")
    dom = cases.make_domain(get_mesh("geo4"), None)
    case = str(tmp_path / "case.bin")
    fortran_case.write_pools(case, dom)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([EXE, case, str(tmp_path / "res.bin")], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 3, out.stdout + out.stderr  # mpas_log_write(..., MPAS_LOG_CRIT)
    assert "no HIP device is visible" in out.stdout


def test_field_records_cover_both_time_levels():
    from mpas_seaice_amd import pools
    dom = cases.make_domain(get_mesh("geo4"), None)
    b = dom.blocklist[0]
    pools.allocate_upwind_pools(b)
    recs = {(p, k): lv for p, k, lv in fortran_case._field_records(b)}
    assert len(recs[("tracers", "iceAreaCategory")]) == 2 and len(recs[("velocity_solver", "uVelocity")]) == 1
    assert not any(p.endswith("_level2") for p, _ in recs)
    assert recs[("velocity_solver", "dynamicsTimeStep")][0].shape == (1,)
    assert recs[("mesh", "verticesOnCell")][0].dtype == np.int32
