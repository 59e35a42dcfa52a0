"""The reference's advection test case on the device: a cosine bell (or slotted cylinder) of ice carried once
around the sphere by solid-body rotation, final state against initial state
(/root/reference/testing_and_setup/testcases/advection/create_ics.py:36-112, advection_error_convergence.py).
Checked: the error falls with resolution for the incremental remap, the remap beats first-order upwind on the
same mesh, area is conserved and stays within its initial bounds, and the device agrees with the CPU oracle
after the whole revolution."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, velocity_solver as vs
from util import get_mesh

pytestmark = pytest.mark.gpu
DAYS = 120.0


def _case(meshname, ic, advection_type, dt):
    mesh = get_mesh(meshname)
    dom = cases.make_domain(mesh, meshname, nCategories=1, configs={
        "config_rotate_cartesian_grid": False, "config_advection_type": advection_type, "config_dt": dt})
    b = dom.blocklist[0]
    m, tr, v = b.structs["mesh"], b.structs["tracers"], b.structs["velocity_solver"]
    nC, nV = b.dimensions["nCells"], b.dimensions["nVertices"]
    R = float(m["sphere_radius"])
    x, y, z = m["xCell"][:nC] / R, m["yCell"][:nC] / R, m["zCell"][:nC] / R
    r = np.sqrt(z * z + x * x)
    area = np.zeros(nC)
    if ic == "cosine_bell":
        inside = (r < 1.0 / 3.0) & (y > 0.0)
        area[inside] = 0.5 * (1.0 + np.cos(np.pi * r[inside] / (1.0 / 3.0)))
    else:  # slotted cylinder
        area[(r < 0.5) & (y > 0.0)] = 1.0
        area[(np.abs(x) < 1.0 / 12.0) & (z > -2.0 / 6.0)] = 0.0
    for a in tr.values():
        a[:] = 0.0
    tr["iceAreaCategory"][0, 0, :nC] = area
    tr["iceVolumeCategory"][0, 0, :nC] = area * 1.0
    tr["surfaceTemperature"][0, 0, :nC] = 1.0
    U = 2.0 * np.pi * R / (DAYS * 86400.0)
    v["uVelocity"][:nV] = U * np.cos(m["latVertex"][:nV])
    v["vVelocity"][:] = 0.0
    v["dynamicsTimeStep"][0] = dt
    return dom, area


def _cleanup(dom):
    """What the column physics does between two advection calls in the model (the ITD clean-up removes
    concentrations below 1e-11).  Without it the bell's tail decays to ~1e-22, where the limited gradient's
    rounding makes a quadrature value -1e-25 and the remap stops with the reference's own abort, "IR negative
    reconstructed ice area" (incremental_remap.F:6903-6925; the oracle stops at the same step)."""
    tr = dom.blocklist[0].structs["tracers"]
    small = tr["iceAreaCategory"] < 1.0e-11
    for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"):
        tr[k][small] = 0.0


def _revolve(dom, dt, flow=None):
    nsteps = int(round(DAYS * 86400.0 / dt))
    if flow is not None:  # CPU oracle
        for _ in range(nsteps):
            flow.seaice_run_advection(dom)
            _cleanup(dom)
        return
    vs.seaice_mesh_pool_create(dom)
    try:
        adv.seaice_init_advection(dom)
        for _ in range(nsteps):
            adv.seaice_run_advection(dom)
            _cleanup(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def _l2(dom, initial):
    b = dom.blocklist[0]
    nC = b.dimensions["nCells"]
    w = b.structs["mesh"]["areaCell"][:nC]
    a = b.structs["tracers"]["iceAreaCategory"][0, 0, :nC]
    return np.sqrt((w * (a - initial) ** 2).sum() / (w * initial ** 2).sum()), a, w


def test_cosine_bell_once_around_the_sphere():
    err = {}
    for meshname, dt in (("QU480", 28800.0), ("QU240", 14400.0)):
        for kind in ("incremental_remap", "upwind"):
            dom, a0 = _case(meshname, "cosine_bell", kind, dt)
            _revolve(dom, dt)
            e, a, w = _l2(dom, a0)
            err[(meshname, kind)] = e
            assert abs((w * a).sum() - (w * a0).sum()) <= 1e-8 * (w * a0).sum(), (meshname, kind)  # the clean-up's share
            assert a.min() >= -1e-12 and a.max() <= a0.max() * (1.0 + 1e-10), (meshname, kind)
    print("relative L2 error after one revolution:", {k: round(float(v), 4) for k, v in err.items()})
    assert err[("QU240", "incremental_remap")] < 0.6 * err[("QU480", "incremental_remap")], err
    for meshname in ("QU480", "QU240"):
        assert err[(meshname, "incremental_remap")] < 0.6 * err[(meshname, "upwind")], err
    assert err[("QU240", "incremental_remap")] < 0.25, err


def test_slotted_cylinder_matches_the_oracle_after_a_revolution():
    from oracle import ir_flow
    dt = 28800.0
    dev, a0 = _case("QU480", "slotted_cylinder", "incremental_remap", dt)
    ref, _ = _case("QU480", "slotted_cylinder", "incremental_remap", dt)
    ir_flow.seaice_init_advection(ref)
    _revolve(ref, dt, flow=ir_flow)
    _revolve(dev, dt)
    nC = dev.blocklist[0].dimensions["nCells"]
    for k in ("iceAreaCategory", "iceVolumeCategory"):
        x, y = dev.blocklist[0].structs["tracers"][k][..., :nC], ref.blocklist[0].structs["tracers"][k][..., :nC]
        assert np.abs(x - y).max() <= 1e-10 * np.abs(y).max(), k
    e, a, w = _l2(dev, a0)
    assert a.min() >= -1e-12 and a.max() <= 1.0 + 1e-10  # the limiter keeps the discontinuous shape in bounds
    assert abs((w * a).sum() - (w * a0).sum()) <= 1e-8 * (w * a0).sum()
