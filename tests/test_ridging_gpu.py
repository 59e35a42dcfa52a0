"""The column package's ridging on the device (sih_column_ridge: ridge_ice of /root/reference/src/column/ice_mechred.F90
for every owned cell) against the CPU oracle, through the host mirror and the C ABI; tolerance 1e-10 relative to each
field's largest value (exp and sqrt differ in the last place between device and host).  The oracle is "parity
unpinned" (oracle/ridge_oracle.c); tests/test_ridging_cpu.py holds what pins it."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, column, pools, time_integration, velocity_solver as vs
from mpas_seaice_amd.lib import SeaiceHipError
from oracle import ir_flow, ridge_flow
from util import get_mesh, rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-10
TRACERS = ["iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature", "iceEnthalpy",
           "iceSalinity", "snowEnthalpy"]


def _cold_snow(b):
    """the idealised snow enthalpy stands for snow at and above the melting point, which the ITD clean-up would send into
    the ocean: 2.5e7 J/m3 colder, i.e. between -20 and -50 degrees"""
    nC = b.dimensions["nCells"]
    b.structs["tracers"]["snowEnthalpy"][..., :nC] -= 0.25e8


def _make(mesh, name, extra=None):
    cfg = {"config_elastic_subcycle_number": 60, "config_use_column_package": True}
    cfg.update(extra or {})
    dom = cases.make_domain(mesh, name, configs=cfg, open_water_band=(-0.2, 0.3))
    for b in dom.blocklist:
        pools.allocate_ridging_pools(b)
        nC = b.dimensions["nCells"]
        # open water as the column physics' aggregation left it before the dynamics (the advection changes the areas,
        # and the ridging closes the sum again)
        b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - b.structs["tracers"]["iceAreaCategory"][0, :, :nC].sum(axis=0)
        b.structs["ocean_fluxes"]["oceanFreshWaterFlux"][:nC] = 1.0e-6
        _cold_snow(b)
    return dom


def _compare(dom, ref, nC):
    b, r = dom.blocklist[0], ref.blocklist[0]
    worst = 0.0
    pairs = [("tracers", t) for t in TRACERS] + [("icestate", "openWaterArea"), ("ocean_fluxes", "oceanFreshWaterFlux"),
                                                 ("ocean_fluxes", "oceanHeatFlux"), ("ocean_fluxes", "oceanSaltFlux"),
                                                 ("tracers_aggregate", "iceAreaCell")]
    assert np.array_equal(b.structs["biogeochemistry"]["newlyFormedIce"][:, :nC], r.structs["biogeochemistry"]["newlyFormedIce"][:, :nC])
    pairs += [("ridging", k) for k in pools.RIDGING_CELL_FIELDS + pools.RIDGING_CATEGORY_FIELDS]
    for pool, name in pairs:
        e = rel_linf(b.structs[pool][name][..., :nC], r.structs[pool][name][..., :nC])
        assert e <= TOL, (pool, name, e)
        worst = max(worst, e)
    return worst


def _compare_aggregate(dom, ref, nC):
    """what seaice_column_update_state leaves: aggregate state, cell values of the tracers, transport tendencies"""
    b, r = dom.blocklist[0], ref.blocklist[0]
    pairs = list(column.AGGREGATE_FIELDS) + [("tracers_aggregate", n + "Cell") for n in column.AGGREGATED_TRACERS]
    for pool, name in pairs:
        e = rel_linf(b.structs[pool][name][..., :nC], r.structs[pool][name][..., :nC])
        assert e <= TOL, (pool, name, e)


@pytest.mark.parametrize("part,redist", [("exponential", "exponential"), ("Thorndike75", "Hibler80")])
def test_ridging_after_velocity_solver_and_advection_matches_oracle(oracle, part, redist):
    """QU240: the idealised forcing gives converging and shearing regions (ridgeConvergence, ridgeShear of the velocity
    solver's post-cycle), an open-water band and ice edges.  Velocity solver, incremental remap and ridging on each
    side; on the device nothing visits the host between the three."""
    mesh = get_mesh("QU240")
    extra = {"config_ridging_participation_function": part, "config_ridging_redistribution_function": redist}
    ref = _make(mesh, "QU240", extra)
    oracle.seaice_init_velocity_solver(ref)
    ir_flow.seaice_init_advection(ref)
    oracle.seaice_run_velocity_solver(ref)
    ir_flow.seaice_run_advection(ref)
    rg = ref.blocklist[0].structs["ridging"]
    assert rg["ridgeConvergence"].max() > 0.0 and rg["ridgeShear"].max() > 0.0
    ridge_flow.column_ridging(ref)

    dom = _make(mesh, "QU240", extra)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        vs.seaice_run_velocity_solver(dom)
        adv.seaice_run_advection(dom, download=False)
        column.column_ridging(dom, upload="cells", download=True)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC = mesh.nCells
    worst = _compare(dom, ref, nC)
    tr = dom.blocklist[0].structs["tracers"]
    a = tr["iceAreaCategory"][0, :, :nC]
    aice0 = dom.blocklist[0].structs["icestate"]["openWaterArea"][:nC]
    assert np.abs(a.sum(axis=0) + aice0 - 1.0).max() < 1e-11 and a.min() >= 0.0 and a.sum(axis=0).max() <= 1.0 + 1e-11
    assert dom.blocklist[0].structs["ridging"]["areaLossRidge"][:nC].max() > 0.0
    print("ridging", part, redist, "worst rel Linf", worst, "iterations", dom.solver_state["ridge_iterations"])


def test_ridging_conserves_volume_and_energy_on_the_device():
    mesh = get_mesh("QU240")
    dom = _make(mesh, "QU240")
    b = dom.blocklist[0]
    nC, nI = mesh.nCells, b.dimensions["nIceLayers"]
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        vs.seaice_run_velocity_solver(dom)
        adv.seaice_run_advection(dom)  # pools hold the advected state: the reference point of the sums below
        tr = b.structs["tracers"]
        vi = tr["iceVolumeCategory"][0, :, :nC].copy()
        ei0 = (tr["iceEnthalpy"][:, :, :nC] * vi[None] / nI).sum(axis=(0, 1))
        vsn0 = tr["snowVolumeCategory"][0, :, :nC].sum(axis=0)
        f0 = b.structs["ocean_fluxes"]["oceanFreshWaterFlux"][:nC].copy()
        column.column_ridging(dom, upload="cells", download=True, cleanup=False)   # ridge_ice alone: the clean-up may zap areas below 1e-11
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    v = tr["iceVolumeCategory"][0, :, :nC]
    assert np.abs(v.sum(axis=0) - vi.sum(axis=0)).max() <= 1e-13 * vi.sum(axis=0).max()
    ei1 = (tr["iceEnthalpy"][:, :, :nC] * v[None] / nI).sum(axis=(0, 1))
    assert np.abs(ei1 - ei0).max() <= 1e-12 * np.abs(ei0).max()
    dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
    lost = vsn0 - tr["snowVolumeCategory"][0, :, :nC].sum(axis=0)
    fresh = b.structs["ocean_fluxes"]["oceanFreshWaterFlux"][:nC] - f0
    assert np.abs(lost - fresh * dt / 330.0).max() <= 1e-12 * vsn0.max()


def test_dynamics_step_with_the_ridging_on_the_device_keeps_the_tracers_there(oracle):
    """Two dynamics subcycles of seaice_timestep (velocity solver -> advection -> ridging, twice): on the device the
    tracers go up once and come back once; the oracle does the same steps on the pools."""
    mesh = get_mesh("QU240")
    extra = {"config_dynamics_subcycle_number": 2, "config_elastic_subcycle_number": 40}
    ref = _make(mesh, "QU240", extra)
    oracle.seaice_init_velocity_solver(ref)
    ir_flow.seaice_init_advection(ref)
    for _ in range(2):
        ridge_flow.column_reinitialize_diagnostics_dynamics(ref)
        oracle.seaice_run_velocity_solver(ref)
        ir_flow.seaice_run_advection(ref)
        ridge_flow.column_dynamics_time_integration(ref)
    dom = _make(mesh, "QU240", extra)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        time_integration.seaice_timestep_dynamics(dom, column_dynamics="device")
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC, nV = mesh.nCells, mesh.nVertices
    _compare(dom, ref, nC)
    _compare_aggregate(dom, ref, nC)
    tend = dom.blocklist[0].structs["diagnostics"]["iceVolumeTendencyTransport"][:nC]
    assert np.abs(tend).max() > 0.0   # the second subcycle's change of the cell's ice volume per second
    for k in ("uVelocity", "vVelocity"):
        e = rel_linf(dom.blocklist[0].structs["velocity_solver"][k][:nV], ref.blocklist[0].structs["velocity_solver"][k][:nV])
        assert e <= TOL, (k, e)


def test_tracers_on_other_tracers_are_refused():
    mesh = get_mesh("QU480")
    dom = _make(mesh, "QU480")
    b = dom.blocklist[0]
    nC, nCat = b.dimensions["nCells"], b.dimensions["nCategories"]
    b.structs["tracers"]["levelIceArea"] = np.full((1, nCat, nC + 1), 0.5, order="F")
    b.structs["tracers"]["levelIceVolume"] = np.full((1, nCat, nC + 1), 0.5, order="F")
    b.structs["tracers"]["pondArea"] = np.full((1, nCat, nC + 1), 0.1, order="F")   # carried on levelIceArea
    dom.configs["config_use_level_meltponds"] = True
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        with pytest.raises(SeaiceHipError, match="carried on another tracer"):
            column.column_ridging(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def _synthetic(mesh, nCategories=5, conv=4.0e-6, shear=2.0e-6, dt=3600.0):
    """strong closing set by hand (no velocity solver): several percent of area per step, so that cells run out of
    open water and thin ice and the reference's iteration (ridge_ice :253-398) has to go round again"""
    dom = cases.make_domain(mesh, None, nCategories=nCategories, open_water_band=(-0.2, 0.3),
                            configs={"config_use_column_package": True})
    b = dom.blocklist[0]
    pools.allocate_ridging_pools(b)
    nC = mesh.nCells
    b.structs["velocity_solver"]["dynamicsTimeStep"][0] = dt
    a = b.structs["tracers"]["iceAreaCategory"][0, :, :nC]
    b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - a.sum(axis=0)
    lat = np.asarray(mesh["latCell"])
    lon = np.asarray(mesh["lonCell"])
    b.structs["ridging"]["ridgeConvergence"][:nC] = conv * np.sin(3.0 * lat) * np.cos(2.0 * lon)   # both signs
    b.structs["ridging"]["ridgeShear"][:nC] = shear * np.cos(lat) ** 2
    _cold_snow(b)
    return dom


@pytest.mark.parametrize("nCat", [5, 1, 7])
def test_strong_closing_set_by_hand_matches_oracle(nCat):
    mesh = get_mesh("QU480")
    ref, dom = _synthetic(mesh, nCat), _synthetic(mesh, nCat)
    ridge_flow.column_ridging(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        column.column_ridging(dom, upload=True, download=True)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC = mesh.nCells
    worst = _compare(dom, ref, nC)
    a = dom.blocklist[0].structs["tracers"]["iceAreaCategory"][0, :, :nC]
    assert np.abs(a.sum(axis=0) + dom.blocklist[0].structs["icestate"]["openWaterArea"][:nC] - 1.0).max() < 1e-11
    assert dom.solver_state["ridge_iterations"] == int(ref.solver_state["ridge_iterations"].max())
    print("nCategories", nCat, "worst", worst, "iterations", dom.solver_state["ridge_iterations"])


def test_itd_cleanup_cases_match_oracle():
    """What the clean-up of colpkg_step_ridge does beyond sorting the ice into its categories: a category with an area
    of 1e-12 goes into the ocean fluxes and is marked as new ice, snow that is too warm is sent to the ocean, and ice
    whose thickness left its category's range (here: volume doubled by hand) moves up.  Device against oracle."""
    mesh = get_mesh("QU480")

    def make():
        dom = _synthetic(mesh, 5, conv=1.0e-7, shear=1.0e-7)
        b = dom.blocklist[0]
        nC = mesh.nCells
        tr = b.structs["tracers"]
        ridge_flow.column_ridging(dom)          # a sorted state to start from (oracle on both sides: same input)
        a = tr["iceAreaCategory"][0, :, :nC]
        cells = np.flatnonzero(a[2] > 0.01)
        small, warm, thick = cells[:25], cells[25:50], cells[50:75]
        tr["iceAreaCategory"][0, 2, small] = 1.0e-12
        tr["iceVolumeCategory"][0, 2, small] = 2.0e-12
        tr["snowVolumeCategory"][0, 2, small] = 2.0e-13
        tr["snowEnthalpy"][0, 1, warm] = -330.0 * 3.34e5 * 0.5
        tr["iceVolumeCategory"][0, 1, thick] *= 2.0
        b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - tr["iceAreaCategory"][0, :, :nC].sum(axis=0)
        return dom, (small, warm, thick)
    ref, (small, warm, thick) = make()
    dom, _ = make()
    ridge_flow.column_ridging(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        column.column_ridging(dom, upload=True, download=True)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC = mesh.nCells
    _compare(dom, ref, nC)
    b = dom.blocklist[0]
    tr = b.structs["tracers"]
    assert np.all(tr["iceAreaCategory"][0, 2, small] == 0.0) and np.all(b.structs["biogeochemistry"]["newlyFormedIce"][2, small] == 1)
    # (where the ridging or the sorting brought in enough cold snow the mixture may stay: nearly all are zapped)
    assert (tr["snowVolumeCategory"][0, 1, warm] == 0.0).sum() >= 20
    assert b.structs["ocean_fluxes"]["oceanSaltFlux"][small].min() > 0.0
    lim = b.structs["initial"]["categoryThicknessLimits"]
    a, v = tr["iceAreaCategory"][0, :, :nC], tr["iceVolumeCategory"][0, :, :nC]
    for n in range(4):
        ok = a[n] > 1e-11
        assert (v[n][ok] / a[n][ok] <= lim[n + 1] * (1 + 1e-14)).all()


def test_ridge_ice_alone_without_the_cleanup_matches_oracle():
    mesh = get_mesh("QU480")
    ref, dom = _synthetic(mesh, 5), _synthetic(mesh, 5)
    ridge_flow.column_ridging(ref, cleanup=False)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        column.column_ridging(dom, upload=True, download=True, cleanup=False)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    _compare(dom, ref, mesh.nCells)


def test_update_state_alone_matches_oracle_bit_for_bit():
    """seaice_column_update_state on its own (no ridging) from the pools: sums in category order, one division per
    tracer -- nothing that may differ between host and device, so the comparison is exact.  The open-water band has
    cells without ice: their surface temperature is the freezing temperature, their other cell values zero."""
    mesh = get_mesh("QU240")
    extra = {"config_use_column_ridging": False}
    ref = _make(mesh, "QU240", extra)
    dom = _make(mesh, "QU240", extra)
    nC = mesh.nCells
    for d in (ref, dom):
        b = d.blocklist[0]
        b.structs["velocity_solver"]["dynamicsTimeStep"][0] = 900.0   # (set by the velocity solver's initialisation)
        b.structs["diagnostics"]["iceAreaTendencyTransport"][:nC] = 0.5
        b.structs["diagnostics"]["iceVolumeTendencyTransport"][:nC] = 1.0
        b.structs["ocean_coupling"]["seaFreezingTemperature"][:nC] = -1.8 + 0.1 * np.cos(np.arange(nC))
    ridge_flow.column_dynamics_time_integration(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        adv.seaice_init_advection(dom)   # the tracer list
        column.seaice_column_dynamics_time_integration(dom, upload=True, download=True)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    b, r = dom.blocklist[0], ref.blocklist[0]
    for pool, name in list(column.AGGREGATE_FIELDS) + [("tracers_aggregate", n + "Cell") for n in column.AGGREGATED_TRACERS]:
        assert np.array_equal(b.structs[pool][name][..., :nC], r.structs[pool][name][..., :nC]), (pool, name)
    ta = b.structs["tracers_aggregate"]
    empty = ta["iceAreaCell"][:nC] == 0.0
    assert empty.sum() > 50
    assert np.array_equal(ta["surfaceTemperatureCell"][:nC][empty], b.structs["ocean_coupling"]["seaFreezingTemperature"][:nC][empty])
    assert not ta["iceEnthalpyCell"][:, :nC][:, empty].any() and ta["iceEnthalpyCell"][:, :nC][:, ~empty].min() < 0.0
    dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
    assert np.array_equal(b.structs["diagnostics"]["iceAreaTendencyTransport"][:nC], (ta["iceAreaCell"][:nC] - 0.5) / dt)
