"""The Fortran host (ISO_C_BINDING interface module + driver, built with amdflang) drives the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from mpas_seaice_amd import cases, fortran_case
from util import get_mesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "fortran", "seaice_hip_driver")


def test_fortran_driver_matches_oracle(oracle, tmp_path):
    if not os.path.exists(EXE):
        pytest.fail("fortran/seaice_hip_driver is not built (run __graft_entry__.build())")
    mesh = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 60}
    ref = cases.make_domain(mesh, "QU480", configs=cfg)
    case = str(tmp_path / "case.bin")
    res = str(tmp_path / "result.bin")
    fortran_case.write_case(case, ref.blocklist[0], ref.configs)
    from oracle import ir_flow
    oracle.seaice_init_velocity_solver(ref)
    ir_flow.seaice_init_advection(ref)
    oracle.seaice_run_velocity_solver(ref)
    ir_flow.seaice_run_advection(ref)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([EXE, case, res], capture_output=True, text=True, timeout=200, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "seaice_hip_driver ok" in out.stdout
    got = fortran_case.read_result(res, ref.blocklist[0])
    b = ref.blocklist[0]
    nC, nV = mesh.nCells, mesh.nVertices
    for name, pool, n in (("uVelocity", "velocity_solver", nV), ("vVelocity", "velocity_solver", nV),
                          ("stress11", "velocity_variational", nC), ("stress22", "velocity_variational", nC),
                          ("stress12", "velocity_variational", nC), ("divergence", "velocity_solver", nC),
                          ("shear", "velocity_solver", nC), ("iceAreaCategory", "tracers", nC),
                          ("iceVolumeCategory", "tracers", nC), ("snowVolumeCategory", "tracers", nC),
                          ("surfaceTemperature", "tracers", nC), ("iceEnthalpy", "tracers", nC),
                          ("iceSalinity", "tracers", nC), ("snowEnthalpy", "tracers", nC)):
        r = b.structs[pool][name][..., :n]
        err = np.abs(got[name] - r).max() / np.abs(r).max()
        assert err <= 1e-10, (name, err)
    # the driver's last leg: one upwind step from the advected state, surface temperature in kelvin
    from oracle import upwind_flow
    b.structs["tracers"]["surfaceTemperature"][..., :nC] += 273.15
    upwind_flow.seaice_init_advection_upwind(ref)
    upwind_flow.seaice_run_advection_upwind(ref)
    for name, key in (("iceAreaCategory", "upwindArea"), ("surfaceTemperature", "upwindTemperature"),
                      ("iceVolumeCategory", "upwindVolume"), ("snowVolumeCategory", "upwindSnow")):
        r = b.structs["tracers"][name][..., :nC]
        err = np.abs(got[key] - r).max() / np.abs(r).max()
        assert err <= 1e-10, (key, err)
    assert np.abs(got["upwindVolume"]).max() > 0.0


def test_fortran_host_step_with_transfers_at_qu60(tmp_path):
    """A dynamics step at QU60 (120 subcycles) as a Fortran host sees it, every input uploaded and every pool field of
    the reference downloaded: the batched calls from page-locked memory, which the shim modules use, against one
    blocking call per field from ordinary memory."""
    if not os.path.exists(EXE):
        pytest.fail("fortran/seaice_hip_driver is not built (run __graft_entry__.build())")
    mesh = get_mesh("QU60")
    dom = cases.make_domain(mesh, "QU60")
    case = str(tmp_path / "case.bin")
    fortran_case.write_case(case, dom.blocklist[0], dom.configs)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([EXE, case, str(tmp_path / "result.bin"), "time"], capture_output=True, text=True,
                         timeout=280, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    ms = {}
    for ln in out.stdout.splitlines():
        if ln.startswith("host step ms, "):
            k, v = ln[len("host step ms, "):].rsplit(":", 1)
            ms[k.strip()] = float(v)
    print(out.stdout)
    per_field = ms["one blocking call per field, ordinary memory"]
    batched = ms["batched calls, page-locked memory"]
    alone = ms["solver alone"]
    lazy = ms["batched calls, page-locked memory, strains and stresses left on the device"]
    assert alone < lazy < batched < per_field
    assert batched <= 14.0, ms  # 7 ms of solver + 69 MB up + 137 MB down over the host link
    # the downloads behind the computation (sih_evp_run's early outputs, sih_fields_get_start): the same bits in every
    # array, the solver's call returns sooner, and the whole step (solver + advection, everything delivered) is shorter
    assert "fields that differ from the plain ones: 0" in out.stdout
    ret = ms["velocity solver, downloads behind the computation, until it returns"]
    plain = ms["velocity solver, batched, page-locked, all outputs before it returns"]
    step_over = ms["velocity solver + advection, downloads behind the computation, all outputs delivered"]
    step_plain = ms["velocity solver + advection, plain"]
    keep = ms["velocity solver, downloads behind the computation, solver state kept on the device, until it returns"]
    assert keep < ret < plain and step_over < step_plain, ms
    assert ret <= 10.0 and keep <= 9.2, ms
