"""The velocity solver on quadrilateral meshes (vertexDegree 4: four cells, four partial sums of the stress divergence
and four corner strains meet at a vertex), as the reference's own quadrilateral test cases use it
(/root/reference/testing_and_setup/testcases/square/1D_velocity_quad, square_quadhex, ridging_1D): device against the
CPU oracle through the host mirror and the C ABI, bar 1e-10 relative L-infinity.  "Parity unpinned" like the rest of the
oracle; the oracle's vertex loops take vertexDegree as the reference's do."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, column, pools, velocity_solver as vs
from mpas_seaice_amd.lib import SeaiceHipError
from mpas_seaice_amd.velocity_solver import RUN_OUTPUTS, STATIC_OUTPUTS
from util import compare_domains, get_mesh

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _pair(oracle, meshname, cfg, run=True, **kw):
    mesh = get_mesh(meshname)
    ref = cases.make_domain(mesh, None, configs=cfg, **kw)
    oracle.seaice_init_velocity_solver(ref)
    if run:
        oracle.seaice_run_velocity_solver(ref)
    dom = cases.make_domain(mesh, None, configs=cfg, **kw)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        if run:
            vs.seaice_run_velocity_solver(dom)
    except Exception:
        vs.seaice_mesh_pool_destroy(dom)
        raise
    return dom, ref


@pytest.mark.parametrize("meshname,cfg", [
    ("quad_24_18", {"config_dt": 600.0, "config_elastic_subcycle_number": 60}),
    ("quad_24_18", {"config_dt": 600.0, "config_elastic_subcycle_number": 60, "config_constitutive_relation_type": "evp_revised",
                    "config_variational_denominator_type": "alternate"}),
    ("quad_24_18", {"config_dt": 600.0, "config_elastic_subcycle_number": 30, "config_average_variational_strain": True}),
    ("quad_24_18", {"config_dt": 600.0, "config_elastic_subcycle_number": 30, "config_variational_basis": "pwl"}),
    ("llquad_48_20", {"config_elastic_subcycle_number": 60, "config_rotate_cartesian_grid": False}),
])
def test_velocity_solver_on_quadrilaterals_matches_oracle(oracle, meshname, cfg):
    kw = {"open_water_band": (-0.2, 0.3)} if meshname.startswith("ll") else {}
    dom, ref = _pair(oracle, meshname, cfg, **kw)
    try:
        rep = []
        compare_domains(dom, ref, STATIC_OUTPUTS, 1e-13, rep)
        compare_domains(dom, ref, RUN_OUTPUTS, TOL, rep)
        b = dom.blocklist[0]
        u = b.structs["velocity_solver"]["uVelocity"]
        assert np.isfinite(u).all() and np.abs(u).max() > 1e-4
        print(meshname, "worst", sorted(rep, key=lambda r: -r[2])[:3], "iterationForm", b.ctx.info("iterationForm"))
        assert int(b.ctx.info("iterationForm")) == 0   # the two-kernel form: tiles of the one-launch forms assume three cells a vertex
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_dynamics_step_on_squares_velocity_advection_ridging(oracle):
    """What the reference's ridging_1D case runs on its mesh of squares: wind along x drives the ice against the wall
    at the end of the domain; velocity solver, incremental remap and ridging for a few steps, device against oracle."""
    from oracle import ir_flow, ridge_flow
    mesh = get_mesh("quad_32_12")
    cfg = {"config_dt": 1800.0, "config_elastic_subcycle_number": 60, "config_use_column_package": True,
           "config_conservation_check": True}

    def make():
        dom = cases.make_domain(mesh, None, configs=cfg)
        b = dom.blocklist[0]
        pools.allocate_ridging_pools(b)
        nC = b.dimensions["nCells"]
        af = b.structs["atmos_coupling"]
        af["uAirVelocity"][:nC] = 10.0       # ridging_1D/create_grid.py: uAirVelocity 10, vAirVelocity 0
        af["vAirVelocity"][:nC] = 0.0
        b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - b.structs["tracers"]["iceAreaCategory"][0, :, :nC].sum(axis=0)
        return dom
    ref, dom = make(), make()
    oracle.seaice_init_velocity_solver(ref)
    ir_flow.seaice_init_advection(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        for step in range(3):
            oracle.seaice_run_velocity_solver(ref)
            ir_flow.seaice_run_advection(ref)
            ridge_flow.column_ridging(ref)
            vs.seaice_run_velocity_solver(dom, tracers_on_device=(step > 0))
            adv.seaice_run_advection(dom, upload=(step == 0), download=False)
            column.column_ridging(dom, upload=("cells" if step == 0 else False), download=True)
        compare_domains(dom, ref, [("velocity_solver", "uVelocity"), ("velocity_solver", "vVelocity")], TOL)
        compare_domains(dom, ref, [("tracers", n) for n in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory",
                                                          "surfaceTemperature", "iceEnthalpy")], TOL)
        b = dom.blocklist[0]
        nC = b.dimensions["nCells"]
        a = b.structs["tracers"]["iceAreaCategory"][0, :, :nC].sum(axis=0)
        assert np.abs(a + b.structs["icestate"]["openWaterArea"][:nC] - 1.0).max() < 1e-11
        assert b.structs["ridging"]["areaLossRidge"][:nC].max() > 0.0
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_weak_scheme_refuses_quadrilateral_meshes():
    dom = cases.make_domain(get_mesh("quad_12_9"), None, configs={"config_strain_scheme": "weak",
                                                                  "config_stress_divergence_scheme": "weak"})
    vs.seaice_mesh_pool_create(dom)
    try:
        with pytest.raises(SeaiceHipError, match="vertexDegree"):
            vs.seaice_init_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_one_dimensional_velocity_case_on_squares(oracle):
    """The reference's 1D_velocity_quad case (testcases/square/1D_velocity_quad/calc_testcase_ic.py): one category
    whose concentration rises linearly in x, twice as much volume, a uniform wind along x, no ocean current, no
    Coriolis force, and velocity boundaries that copy the second row onto the outer one at the bottom and at the top.
    The answer has no structure in y: every column of vertices carries the same velocity, and v stays zero."""
    nx, ny = 40, 20
    mesh = get_mesh(f"quad_{nx}_{ny}")
    cfg = {"config_dt": 600.0, "config_elastic_subcycle_number": 120, "config_use_special_boundaries_velocity": True}

    def make():
        dom = cases.make_domain(mesh, None, nCategories=1, configs=cfg)
        b = dom.blocklist[0]
        m = b.structs["mesh"]
        nC, nV = b.dimensions["nCells"], b.dimensions["nVertices"]
        x = m["xCell"][:nC]
        Lx = m["xVertex"][:nV].max() - m["xVertex"][:nV].min()
        tr = b.structs["tracers"]
        tr["iceAreaCategory"][0, 0, :nC] = x / Lx
        tr["iceVolumeCategory"][0, 0, :nC] = 2.0 * tr["iceAreaCategory"][0, 0, :nC]
        tr["snowVolumeCategory"][0, 0, :nC] = 0.0
        at, oc = b.structs["atmos_coupling"], b.structs["ocean_coupling"]
        at["uAirVelocity"][:nC], at["vAirVelocity"][:nC] = 1.0, 0.0
        oc["uOceanVelocity"][:nC] = oc["vOceanVelocity"][:nC] = 0.0
        oc["seaSurfaceTiltU"][:nC] = oc["seaSurfaceTiltV"][:nC] = 0.0
        m["fVertex"][:] = 0.0
        b.structs["velocity_solver"]["uVelocity"][:] = 0.0
        b.structs["velocity_solver"]["vVelocity"][:] = 0.0
        for k in ("stress11", "stress22", "stress12"):
            b.structs["velocity_variational"][k][:] = 0.0
        sb = b.structs["special_boundaries"]
        typ, src = sb["vertexBoundaryType"], sb["vertexBoundarySource"]
        for i in range(nx + 1):       # vertex (i, j) has the number j (nx + 1) + i
            typ[i], src[i] = 1, (ny - 1) * (nx + 1) + i + 1           # bottom row from the row below the top one
            typ[ny * (nx + 1) + i], src[ny * (nx + 1) + i] = 1, (nx + 1) + i + 1   # top row from the second row
        return dom
    ref, dom = make(), make()
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        vs.seaice_run_velocity_solver(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    compare_domains(dom, ref, [("velocity_solver", "uVelocity")], TOL)
    vd, vr = (d_.blocklist[0].structs["velocity_solver"]["vVelocity"] for d_ in (dom, ref))
    scale = np.abs(ref.blocklist[0].structs["velocity_solver"]["uVelocity"]).max()
    assert np.abs(vd - vr).max() <= TOL * scale      # v is rounding noise around zero: measured against the size of u
    u = dom.blocklist[0].structs["velocity_solver"]["uVelocity"][:(nx + 1) * (ny + 1)].reshape(ny + 1, nx + 1)
    v = dom.blocklist[0].structs["velocity_solver"]["vVelocity"][:(nx + 1) * (ny + 1)].reshape(ny + 1, nx + 1)
    assert np.abs(u).max() > 1e-4
    inner = u[1:-1]
    assert np.abs(inner - inner[0][None, :]).max() <= 1e-10 * np.abs(u).max()     # no structure in y
    assert np.abs(v[1:-1]).max() <= 1e-10 * np.abs(u).max()
    assert np.array_equal(u[0], u[ny - 1]) and np.array_equal(u[ny], u[1])         # the boundary rows are copies
