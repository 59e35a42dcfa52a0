"""Upwind advection (config_advection_type = 'upwind') on the device against the CPU oracle, through the host
mirror and the C ABI.  Tolerance 1e-10 relative to each field's largest value; the oracle's own parity status
is "unpinned" (oracle/upwind_oracle.c)."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, parallel, velocity_solver as vs
from mpas_seaice_amd.advection_upwind import CHAIN
from mpas_seaice_amd.lib import SeaiceHipError
from util import compare_domains, get_mesh

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _fields():
    f = [("velocity_solver", "normalVectorEdge"), ("boundary", "interiorEdge")]
    for k in CHAIN:
        f += [("tracers", k), ("tracers_level2", k), ("tracer_tendencies", k + "Tend"),
              ("tracer_edge_fluxes", k + "EdgeFlux"), ("tracer_conservation", k + "Cons"),
              ("tracer_conservation_level2", k + "Cons")]
    return f + [("tracers", "iceEnthalpy"), ("tracers_level2", "iceEnthalpy"), ("tracers", "snowEnthalpy")]


def _make(meshname, name, warm, configs=None, nblocks=None):
    cfg = {"config_advection_type": "upwind", "config_conservation_check": True}
    cfg.update(configs or {})
    mesh = get_mesh(meshname)
    if nblocks:
        dom, halos, _ = parallel.make_partitioned_domain(mesh, name, nblocks, configs=cfg, open_water_band=(-0.2, 0.3))
    else:
        dom, halos = cases.make_domain(mesh, name, configs=cfg, open_water_band=(-0.2, 0.3)), None
    for b in dom.blocklist:
        nV = b.dimensions["nVertices"]
        m, v = b.structs["mesh"], b.structs["velocity_solver"]
        v["dynamicsTimeStep"][0] = dom.configs["config_dt"]
        if bool(m["on_a_sphere"]):
            la, lo = m["latVertex"][:nV], m["lonVertex"][:nV]
            v["uVelocity"][:nV] = 0.4 * np.cos(la) + 0.1 * np.sin(2.0 * lo)
            v["vVelocity"][:nV] = 0.2 * np.sin(lo) * np.cos(la)
        else:
            X, Y = m["xVertex"][:nV] / m["xVertex"][:nV].max(), m["yVertex"][:nV] / m["yVertex"][:nV].max()
            v["uVelocity"][:nV] = 0.3 * np.sin(2.0 * np.pi * Y) + 0.1
            v["vVelocity"][:nV] = -0.2 * np.cos(2.0 * np.pi * X)
        if warm:
            b.structs["tracers"]["surfaceTemperature"] += 273.15
    return dom, halos


def _run_device(dom, steps=1):
    vs.seaice_mesh_pool_create(dom)
    try:
        adv.seaice_init_advection(dom)
        for _ in range(steps):
            adv.seaice_run_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


@pytest.mark.parametrize("meshname,name,warm", [("QU240", "QU240", True), ("QU240", "QU240", False),
                                                ("var12", None, True), ("hex_24_20", None, True)])
def test_upwind_matches_oracle(meshname, name, warm):
    from oracle import upwind_flow
    ref, _ = _make(meshname, name, warm)
    dev, _ = _make(meshname, name, warm)
    upwind_flow.seaice_init_advection_upwind(ref)
    for _ in range(2):
        upwind_flow.seaice_run_advection_upwind(ref)
    _run_device(dev, steps=2)
    compare_domains(dev, ref, _fields(), TOL)
    a = dev.blocklist[0].structs["tracers"]["iceAreaCategory"]
    assert a.max() > 0.1 and np.isfinite(a).all()
    if warm:
        assert dev.blocklist[0].structs["tracers"]["iceVolumeCategory"].max() > 0.0


def test_upwind_after_the_velocity_solver(oracle):
    """the reference's order: velocity solver, then advection with its velocities"""
    from oracle import upwind_flow
    cfg = {"config_elastic_subcycle_number": 30}
    ref, _ = _make("QU240", "QU240", True, configs=cfg)
    dev, _ = _make("QU240", "QU240", True, configs=cfg)
    oracle.seaice_init_velocity_solver(ref)
    upwind_flow.seaice_init_advection_upwind(ref)
    oracle.seaice_run_velocity_solver(ref)
    upwind_flow.seaice_run_advection_upwind(ref)
    vs.seaice_mesh_pool_create(dev)
    try:
        vs.seaice_init_velocity_solver(dev)
        adv.seaice_init_advection(dev)
        vs.seaice_run_velocity_solver(dev)
        adv.seaice_run_advection(dev)
    finally:
        vs.seaice_mesh_pool_destroy(dev)
    compare_domains(dev, ref, [("velocity_solver", "uVelocity"), ("velocity_solver", "vVelocity")] + _fields(), TOL)


def test_upwind_on_three_blocks_is_bit_identical_on_owned_cells():
    one, _ = _make("QU240", "QU240", True)
    _run_device(one, steps=2)
    dom, halos = _make("QU240", "QU240", True, nblocks=3)
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, 3)
        adv.seaice_init_advection(dom)
        for _ in range(2):
            adv.seaice_run_advection(dom)
            dom.comm.halo_exchange(dom, list(CHAIN))  # the caller's halo update between steps
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC = one.blocklist[0].dimensions["nCells"]
    for k in CHAIN:
        g = parallel.gather_global(dom.blocklist, halos, "tracers", k, nC, "cell")
        assert np.array_equal(g, one.blocklist[0].structs["tracers"][k][..., :nC]), k
    tot = np.zeros(one.blocklist[0].dimensions["nCategories"])
    for b in dom.blocklist:
        tot += b.structs["tracer_conservation_level2"]["iceAreaCategoryCons"][0]
    ref = one.blocklist[0].structs["tracer_conservation_level2"]["iceAreaCategoryCons"][0]
    assert np.allclose(tot, ref, rtol=1e-12)


def test_upwind_error_paths():
    dom, _ = _make("QU480", "QU480", True)
    vs.seaice_mesh_pool_create(dom)
    try:
        ctx = dom.blocklist[0].ctx
        with pytest.raises(SeaiceHipError, match="sih_upwind_init has not run"):
            ctx.upwind_prepare()
        adv.seaice_init_advection(dom)
        with pytest.raises(SeaiceHipError, match="sih_upwind_prepare must run first"):
            ctx.upwind_advect(100.0)
        with pytest.raises(SeaiceHipError, match="not a tracer of the upwind chain"):
            ctx.upwind_get_sums("iceEnthalpy", 5)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
