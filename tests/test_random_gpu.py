"""Seeded random states (patchy ice with holes and missing categories, noisy forcing, non-zero initial velocity)
on meshes with 5-, 6- and 7-sided cells: the device against the CPU oracle for a velocity-solver step followed
by an incremental-remap step.  If the state makes the remap stop with one of the reference's aborts, the device
has to stop with the same message."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, velocity_solver as vs
from mpas_seaice_amd.lib import SeaiceHipError
from util import compare_domains, get_mesh

pytestmark = pytest.mark.gpu
TOL = 1e-10
STATE = [("velocity_solver", k) for k in ("uVelocity", "vVelocity", "stressDivergenceU", "oceanStressCellU", "divergence",
                                          "shear", "solveStress", "solveVelocity")] + \
        [("velocity_variational", k) for k in ("stress11", "stress22", "stress12")] + \
        [("tracers", k) for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature",
                                  "iceEnthalpy", "iceSalinity", "snowEnthalpy")]


def _randomise(dom, seed):
    rng = np.random.Generator(np.random.PCG64(20241020 + seed))
    b = dom.blocklist[0]
    nC, nV, nCat = b.dimensions["nCells"], b.dimensions["nVertices"], b.dimensions["nCategories"]
    tr = b.structs["tracers"]
    hole = rng.random(nC) < 0.25                      # ice-free cells scattered through the pack
    a = tr["iceAreaCategory"][0, :, :nC] * rng.uniform(0.2, 1.0, (nCat, nC))
    a[rng.random((nCat, nC)) < 0.15] = 0.0            # single categories missing
    a[:, hole] = 0.0
    h = rng.uniform(0.2, 3.0, (nCat, nC))
    tr["iceAreaCategory"][0, :, :nC] = a
    tr["iceVolumeCategory"][0, :, :nC] = a * h
    tr["snowVolumeCategory"][0, :, :nC] = a * rng.uniform(0.0, 0.4, (nCat, nC))
    tr["surfaceTemperature"][0, :, :nC] = rng.uniform(-25.0, 0.0, (nCat, nC))
    for k, lo, hi in (("iceEnthalpy", -3.0e8, -1.0e8), ("iceSalinity", 0.0, 8.0), ("snowEnthalpy", -1.3e8, -1.0e8)):
        tr[k][..., :nC] = rng.uniform(lo, hi, tr[k][..., :nC].shape)
    at, oc, v = b.structs["atmos_coupling"], b.structs["ocean_coupling"], b.structs["velocity_solver"]
    at["uAirVelocity"][:nC] += rng.normal(0.0, 3.0, nC)
    at["vAirVelocity"][:nC] += rng.normal(0.0, 3.0, nC)
    oc["uOceanVelocity"][:nC] += rng.normal(0.0, 0.05, nC)
    oc["vOceanVelocity"][:nC] += rng.normal(0.0, 0.05, nC)
    v["uVelocity"][:nV] = rng.normal(0.0, 0.05, nV)
    v["vVelocity"][:nV] = rng.normal(0.0, 0.05, nV)


@pytest.mark.parametrize("meshname,seed", [("var12", 1), ("var12", 2), ("QU240", 3), ("hex_24_20", 4)])
def test_random_state_step(oracle, meshname, seed):
    from oracle import ir_flow
    cfg = {"config_elastic_subcycle_number": 40, "config_dt": 1800.0, "config_conservation_check": True}
    doms = []
    for _ in range(2):
        d = cases.make_domain(get_mesh(meshname), None, configs=cfg)
        _randomise(d, seed)
        doms.append(d)
    ref, dev = doms
    oracle.seaice_init_velocity_solver(ref)
    ir_flow.seaice_init_advection(ref)
    oracle.seaice_run_velocity_solver(ref)
    abort = None
    try:
        ir_flow.seaice_run_advection(ref)
    except ir_flow.IRError as e:
        abort = str(e)
    vs.seaice_mesh_pool_create(dev)
    try:
        vs.seaice_init_velocity_solver(dev)
        adv.seaice_init_advection(dev)
        vs.seaice_run_velocity_solver(dev)
        if abort is None:
            adv.seaice_run_advection(dev)
        else:
            with pytest.raises(SeaiceHipError) as info:
                adv.seaice_run_advection(dev)
            assert abort.replace("MPAS-seaice: ", "") in str(info.value)
    finally:
        vs.seaice_mesh_pool_destroy(dev)
    fields = STATE if abort is None else [f for f in STATE if f[0] != "tracers"]
    worst = compare_domains(dev, ref, fields, TOL)
    print(meshname, seed, "abort:", abort, "worst rel Linf:", worst)
