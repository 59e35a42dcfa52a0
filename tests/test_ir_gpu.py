"""GPU parity of the HIP incremental-remap advection against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, parallel, velocity_solver as vs
from util import get_mesh, rel_linf

pytestmark = pytest.mark.gpu

# tracer parity bar of the project: 1e-10 relative L-infinity, fp64
TOL = 1e-10
TRACERS = ["iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature", "iceEnthalpy",
           "iceSalinity", "snowEnthalpy"]


def _ir():
    from oracle import ir_flow
    return ir_flow


def make_pair(oracle, meshname, name=None, configs=None, **kw):
    mesh = get_mesh(meshname)
    ref = cases.make_domain(mesh, name, configs=configs, **kw)
    oracle.seaice_init_velocity_solver(ref)
    _ir().seaice_init_advection(ref)
    dom = cases.make_domain(mesh, name, configs=configs, **kw)
    vs.seaice_mesh_pool_create(dom)
    vs.seaice_init_velocity_solver(dom)
    adv.seaice_init_advection(dom)
    return dom, ref


def copy_velocity(ref, dom):
    for br, bd in zip(ref.blocklist, dom.blocklist):
        for k in ("uVelocity", "vVelocity"):
            bd.structs["velocity_solver"][k][:] = br.structs["velocity_solver"][k]


def compare_tracers(dom, ref, tol=TOL):
    rep = {}
    for bd, br in zip(dom.blocklist, ref.blocklist):
        nC = bd.dimensions["nCells"]
        for n in TRACERS:
            x, y = bd.structs["tracers"][n][..., :nC], br.structs["tracers"][n][..., :nC]
            e = rel_linf(x, y)
            rep[n] = max(rep.get(n, 0.0), e)
            assert np.isfinite(x).all(), n
            assert e <= tol, f"{n}: rel Linf {e:.3e} > {tol:.1e}"
    return rep


def dget(b, name, lead, n, dtype=np.float64):
    a = np.zeros((lead, n + 1), dtype=dtype, order="F")
    b.ctx.get(name, a)
    return a[:, :n]


@pytest.mark.parametrize("meshname", ["QU480", "hex_12_10"])
def test_init_geometry_matches_oracle(oracle, meshname):
    dom, ref = make_pair(oracle, meshname, "QU480" if meshname == "QU480" else None)
    try:
        b, r = dom.blocklist[0], ref.blocklist[0]
        d = b.dimensions
        nC, nV, nE, M = d["nCells"], d["nVertices"], d["nEdges"], d["maxEdges"]
        ir = r.structs["incremental_remap"]
        rep = {}
        for nm in ("xVertexOnCell", "yVertexOnCell"):
            rep[nm] = rel_linf(dget(b, nm, M, nC), ir[nm][:, :nC])
        for nm in ("xVertexOnEdge", "yVertexOnEdge"):
            rep[nm] = rel_linf(dget(b, nm, 6, nE), ir[nm][:6, :nE])
        rep["geomAvgCell"] = max(rel_linf(dget(b, "geomAvgCell", 14, nC)[k], ir["geomAvgCell"][:nC, k])
                                 for k in range(14))
        rep["minLength"] = rel_linf(dget(b, "minLengthEdgesOnVertex", 1, nV)[0], ir["minLengthEdgesOnVertex"][:nV])
        remap = dget(b, "remapEdge", 1, nE, np.int32)[0]
        assert np.array_equal(remap, ir["remapEdge"][:nE])
        print(rep)
        assert max(rep.values()) <= 1e-14, rep
    finally:
        vs.seaice_mesh_pool_destroy(dom)


@pytest.mark.parametrize("meshname,name,cfg,kw", [
    ("QU480", "QU480", {}, {}),
    ("QU480", "QU480", {}, {"open_water_band": (-0.3, 0.4)}),
    ("QU240", "QU240", {"config_elastic_subcycle_number": 30}, {}),
    ("hex_24_20", None, {"config_dt": 600.0, "config_elastic_subcycle_number": 60}, {}),
    # variable resolution, cells with 5, 6 and 7 edges (maxEdges = 7)
    ("var12", None, {"config_dt": 1800.0, "config_elastic_subcycle_number": 60}, {}),
])
def test_advection_step_matches_oracle(oracle, meshname, name, cfg, kw):
    """One incremental-remap step from identical velocities: every advected tracer, plus the
    intermediate departure triangles, edge fluxes and reconstructions."""
    ir = _ir()
    cfg = dict(cfg, config_conservation_check=True)
    dom, ref = make_pair(oracle, meshname, name, configs=cfg, **kw)
    try:
        oracle.seaice_run_velocity_solver(ref)
        copy_velocity(ref, dom)
        diag = {}
        ir.seaice_run_advection(ref, diagnostics=diag)
        adv.seaice_run_advection(dom)
        rep = compare_tracers(dom, ref)
        b, r = dom.blocklist[0], ref.blocklist[0]
        d = b.dimensions
        nC, nE, nCat = d["nCells"], d["nEdges"], d["nCategories"]
        rir = r.structs["incremental_remap"]
        ta = dget(b, "triangleArea", 6, nE)
        rep["triangleArea"] = rel_linf(ta, rir["triangleArea"][:, :nE])
        assert np.array_equal(ta != 0.0, rir["triangleArea"][:, :nE] != 0.0)
        xt = dget(b, "xTriangle", 18, nE).reshape((3, 6, nE), order="F")
        live = (rir["triangleArea"][:, :nE] != 0.0)[None, :, :]
        rep["xTriangle"] = rel_linf(np.where(live, xt, 0.0), np.where(live, diag["xTriangleVertices"][:3, :, :nE], 0.0))
        W = diag["work"]
        b.ctx.ir_unpack_reconstruction()
        for n, L in (("iceAreaCategory", 1), ("iceVolumeCategory", 1), ("iceEnthalpy", 7)):
            lead = L * nCat
            rep[n + "Gradx"] = rel_linf(dget(b, n + "Gradx", lead, nC),
                                        W[n].xGrad[..., :nC].reshape((lead, nC), order="F"))
            rep[n + "Center"] = rel_linf(dget(b, n + "Center", lead, nC),
                                         W[n].center[..., :nC].reshape((lead, nC), order="F"))
            rep[n + "EdgeFlux"] = rel_linf(dget(b, n + "EdgeFlux", lead, nE),
                                           W[n].edgeFlux[..., :nE].reshape((lead, nE), order="F"))
        rep["BaryX"] = rel_linf(dget(b, "iceVolumeCategoryBarycenterx", nCat, nC),
                                W["iceVolumeCategory"].xBary[..., :nC].reshape((nCat, nC), order="F"))
        print({k: float(f"{v:.2e}") for k, v in rep.items()})
        assert max(rep.values()) <= TOL, rep
        # device-side conservation sums agree with the oracle's
        s0, s1 = dom.solver_state["ir_sums"]["iceVolumeCategory"]
        assert np.abs(s1 - s0).max() <= 1e-11 * np.abs(s0).max()
        o0 = r._ir_sums["iceVolumeCategory"][0].reshape(-1, order="F")
        assert rel_linf(s0, o0) < 1e-12
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_dynamics_and_advection_two_steps(oracle):
    """seaice_run_velocity_solver + seaice_run_advection, twice, each side with its own velocities."""
    ir = _ir()
    dom, ref = make_pair(oracle, "QU480", "QU480", configs={"config_elastic_subcycle_number": 60})
    try:
        for _ in range(2):
            oracle.seaice_run_velocity_solver(ref)
            ir.seaice_run_advection(ref)
            vs.seaice_run_velocity_solver(dom)
            adv.seaice_run_advection(dom)
        rep = compare_tracers(dom, ref)
        print(rep)
        u, ur = (x.blocklist[0].structs["velocity_solver"]["uVelocity"] for x in (dom, ref))
        assert rel_linf(u, ur) <= TOL
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_no_renumbering_gives_same_bits(oracle):
    mesh = get_mesh("QU480")
    outs = []
    for renumber in (True, False):
        dom = cases.make_domain(mesh, "QU480")
        vs.seaice_mesh_pool_create(dom, renumber=renumber)
        try:
            vs.seaice_init_velocity_solver(dom)
            adv.seaice_init_advection(dom)
            vs.seaice_run_velocity_solver(dom)
            adv.seaice_run_advection(dom)
            outs.append({n: dom.blocklist[0].structs["tracers"][n].copy() for n in TRACERS})
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    for n in TRACERS:
        assert np.array_equal(outs[0][n], outs[1][n]), n


@pytest.mark.parametrize("nblocks", [2, 4])
def test_multiblock_advection_matches_single_block_oracle(oracle, nblocks):
    """Blocks with two halo layers, halo updates of velocity and tracers through the device pack /
    unpack path: owned cells of every block equal the single-block oracle."""
    ir = _ir()
    mesh = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 30, "config_monotonicity_check": nblocks == 4}
    ref = cases.make_domain(mesh, "QU480", configs=cfg)
    oracle.seaice_init_velocity_solver(ref)
    ir.seaice_init_advection(ref)
    oracle.seaice_run_velocity_solver(ref)
    ir.seaice_run_advection(ref)
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU480", nblocks, configs=cfg)
    vs.seaice_mesh_pool_create(dom)
    try:
        dom.comm = parallel.LocalComm(dom, halos, nblocks)
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        vs.seaice_run_velocity_solver(dom)
        adv.seaice_run_advection(dom)
        nC = mesh.nCells
        for n in TRACERS:
            g = parallel.gather_global(dom.blocklist, halos, "tracers", n, nC, "cell")
            e = rel_linf(g, ref.blocklist[0].structs["tracers"][n][..., :nC])
            assert e <= TOL, (n, e)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_unsupported_and_error_paths():
    mesh = get_mesh("QU480")
    dom = cases.make_domain(mesh, "QU480")
    vs.seaice_mesh_pool_create(dom)
    try:
        from mpas_seaice_amd.lib import SeaiceHipError
        b = dom.blocklist[0]
        with pytest.raises(SeaiceHipError):
            b.ctx.ir_run(100.0)                      # before init
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        with pytest.raises(SeaiceHipError):        # child listed before its parent
            b.ctx.ir_set_tracers(["iceAreaCategory", "iceEnthalpy", "iceVolumeCategory"],
                                 ["None", "iceVolumeCategory", "iceAreaCategory"], [1, 7, 1])
        # a negative concentration somewhere aborts like the reference ("IR negative reconstructed ice area")
        adv.seaice_init_advection(dom)
        nV = b.dimensions["nVertices"]
        b.structs["velocity_solver"]["uVelocity"][:nV] = 0.3
        b.structs["tracers"]["iceAreaCategory"][0, 0, 100:140] = -0.5
        with pytest.raises(SeaiceHipError):
            adv.seaice_run_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


@pytest.mark.parametrize("meshname,name,cfg", [
    ("QU480", "QU480", {}),
    ("hex_24_20", None, {"config_dt": 600.0, "config_elastic_subcycle_number": 60}),
    ("var12", None, {"config_dt": 1800.0, "config_elastic_subcycle_number": 60}),
])
def test_monotonicity_check_matches_oracle(oracle, meshname, name, cfg):
    """config_monotonicity_check: the bounds (local min / max of the old values, extended by a layer in the
    reference's in-place order) are the oracle's bit for bit, and the step passes the check."""
    ir = _ir()
    cfg = dict(cfg, config_monotonicity_check=True)
    dom, ref = make_pair(oracle, meshname, name, configs=cfg, open_water_band=(-0.3, 0.4))
    try:
        oracle.seaice_run_velocity_solver(ref)
        copy_velocity(ref, dom)
        ir.seaice_run_advection(ref)
        adv.seaice_run_advection(dom)
        compare_tracers(dom, ref)
        nCs = dom.blocklist[0].dimensions["nCellsSolve"]
        pd, pr = (x.blocklist[0].structs["tracer_monotonicity"] for x in (dom, ref))
        seen = 0
        for n in TRACERS[1:]:
            for suffix in ("LocalMin", "LocalMax"):
                a, b = pd[n + suffix][..., :nCs], pr[n + suffix][..., :nCs]
                assert np.array_equal(a, b), (n + suffix, np.abs(a - b).max())
                seen += int(np.abs(b).max() > 0)
        assert seen >= 10
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_monotonicity_violation_is_reported(oracle):
    from mpas_seaice_amd.lib import SeaiceHipError
    cfg = {"config_monotonicity_check": True}
    dom, ref = make_pair(oracle, "QU480", "QU480", configs=cfg)
    try:
        oracle.seaice_run_velocity_solver(ref)
        copy_velocity(ref, dom)
        b = dom.blocklist[0]
        ctx, tp = b.ctx, b.structs["tracers"]
        ctx.put_many([("uVelocity", b.structs["velocity_solver"]["uVelocity"]),
                      ("vVelocity", b.structs["velocity_solver"]["vVelocity"])] + [(n, tp[n]) for n in TRACERS])
        dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
        ctx.ir_run(dt, False, True)
        with pytest.raises(SeaiceHipError, match="call sih_ir_check_monotonicity"):
            ctx.ir_run(dt, False, True)
        # push one new value far outside the old neighbourhood range
        T = np.zeros_like(tp["surfaceTemperature"])
        ctx.get("surfaceTemperature", T)
        cell = int(np.argmax(tp["iceAreaCategory"][0, 2, :-1] > 0.05))
        T[0, 2, cell] += 50.0
        ctx.put("surfaceTemperature", T)
        with pytest.raises(SeaiceHipError, match="monotonicity violation: tracer = surfaceTemperature"):
            ctx.ir_check_monotonicity()
        with pytest.raises(SeaiceHipError, match="no step with the monotonicity check is pending"):
            ctx.ir_check_monotonicity()
        ctx.ir_run(dt, False, False)  # the context is usable again
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_cfl_warning_count(oracle):
    """find_departure_points (:5325-5345) only warns when a trajectory is longer than the shortest edge at
    the vertex; the library reports how many vertices that happened at."""
    ir = _ir()
    dom, ref = make_pair(oracle, "hex_12_10", None, configs={"config_dt": 600.0})
    try:
        for d in (dom, ref):
            vsol = d.blocklist[0].structs["velocity_solver"]
            vsol["uVelocity"][:] = 0.0
            vsol["vVelocity"][:] = 0.0
            vsol["uVelocity"][5:9] = 1.0e3  # 600 s * 1 km/s = 600 km against ~60 km edges
        diag = {}
        ir.seaice_run_advection(ref, diagnostics=diag)
        try:
            adv.seaice_run_advection(dom)
        except Exception:
            pass  # such a flow may also trip the remap's own aborts; the count is set before them
        assert int(dom.blocklist[0].ctx.info("irCflWarnings")) == diag["cflWarnings"] == 4
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_more_layers_than_one_chunk(oracle):
    """10 ice layers and 3 snow layers: the kernels handle layers in chunks of 8, so the 7-layer default never
    reaches the second chunk; one category fewer as well."""
    ir = _ir()
    cfg = {"config_nIceLayers": 10, "config_nSnowLayers": 3, "config_elastic_subcycle_number": 30,
           "config_conservation_check": True}
    dom, ref = make_pair(oracle, "QU480", "QU480", configs=cfg, nCategories=4, open_water_band=(-0.2, 0.3))
    try:
        assert dom.blocklist[0].structs["tracers"]["iceEnthalpy"].shape[:2] == (10, 4)
        for _ in range(2):
            oracle.seaice_run_velocity_solver(ref)
            ir.seaice_run_advection(ref)
            vs.seaice_run_velocity_solver(dom)
            adv.seaice_run_advection(dom)
        rep = compare_tracers(dom, ref)
        assert max(rep.values()) <= TOL, rep
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def _add_optional_tracers(dom, seed=7, vertical=True):
    """Optional tracers of the reference's list, three parents deep, with and without layers."""
    rng = np.random.Generator(np.random.PCG64(seed))
    for b in dom.blocklist:
        nC, nCat = b.dimensions["nCells"], b.dimensions["nCategories"]
        tr = b.structs["tracers"]

        def field(L, lo, hi):
            a = np.zeros((L, nCat, nC + 1), order="F")
            a[..., :nC] = rng.uniform(lo, hi, (L, nCat, nC))
            return a
        tr["iceAge"] = field(1, 0.0, 3.0e7)
        tr["levelIceArea"] = field(1, 0.2, 1.0)
        tr["levelIceVolume"] = field(1, 0.2, 1.0)
        tr["pondArea"] = field(1, 0.0, 0.5)            # child of levelIceArea with config_use_level_meltponds
        tr["pondDepth"] = field(1, 0.0, 0.3)           # three parents: pondArea, levelIceArea, iceAreaCategory
        tr["brineFraction"] = field(1, 0.5, 1.0)
        if vertical:
            tr["verticalNitrateIce"] = field(8, 0.0, 5.0)  # layers under a parent without layers, three parents
            tr["verticalNitrateSnow"] = field(2, 0.0, 5.0)
            tr["snowScatteringAerosol"] = field(2, 0.0, 1.0e-6)


def test_optional_tracers_three_parents_deep(oracle):
    ir = _ir()
    mesh = get_mesh("var12")
    cfg = {"config_elastic_subcycle_number": 30, "config_dt": 1800.0, "config_conservation_check": True,
           "config_use_level_meltponds": True}
    ref = cases.make_domain(mesh, None, configs=cfg, open_water_band=(-0.2, 0.3))
    dom = cases.make_domain(mesh, None, configs=cfg, open_water_band=(-0.2, 0.3))
    _add_optional_tracers(ref)
    _add_optional_tracers(dom)
    oracle.seaice_init_velocity_solver(ref)
    ir.seaice_init_advection(ref)
    names = [t.tracerName for t in ref.solver_state["tracersHead"]]
    assert names[:4] == ["iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature"]
    assert {t.tracerName: t.nParents for t in ref.solver_state["tracersHead"]}["pondDepth"] == 3
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        for _ in range(2):
            oracle.seaice_run_velocity_solver(ref)
            ir.seaice_run_advection(ref)
            vs.seaice_run_velocity_solver(dom)
            adv.seaice_run_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    nC = mesh.nCells
    for n in names:
        x = dom.blocklist[0].structs["tracers"][n][..., :nC]
        y = ref.blocklist[0].structs["tracers"][n][..., :nC]
        assert np.isfinite(x).all(), n
        assert rel_linf(x, y) <= TOL, (n, rel_linf(x, y))
    assert len(names) == 16


@pytest.mark.parametrize("with_column", [False, True])
def test_timestep_dynamics_loop(oracle, with_column):
    """the caller of both paths: two dynamics subcycles per time step, with and without a host routine in the
    place of the column package's ridging between them (time_integration.F:138-162)"""
    from mpas_seaice_amd import time_integration as ti
    ir = _ir()
    cfg = {"config_dynamics_subcycle_number": 2, "config_elastic_subcycle_number": 30}

    def column(domain):  # stands for seaice_column_dynamics_time_integration: any host-side change of the pools
        for b in domain.blocklist:
            tr = b.structs["tracers"]
            small = tr["iceAreaCategory"] < 0.05
            for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"):
                tr[k][small] = 0.0
            tr["iceEnthalpy"] *= 0.999
    dom, ref = make_pair(oracle, "QU480", "QU480", configs=cfg, open_water_band=(-0.2, 0.3))
    try:
        for _ in range(2):
            oracle.seaice_run_velocity_solver(ref)
            ir.seaice_run_advection(ref)
            if with_column:
                column(ref)
        ti.seaice_timestep_dynamics(dom, column_dynamics=column if with_column else None)
        rep = compare_tracers(dom, ref)
        for k in ("uVelocity", "vVelocity"):
            x, y = dom.blocklist[0].structs["velocity_solver"][k], ref.blocklist[0].structs["velocity_solver"][k]
            assert rel_linf(x, y) <= TOL, k
        assert abs(dom.blocklist[0].structs["velocity_solver"]["dynamicsTimeStep"][0] - 0.5 * dom.configs["config_dt"]) < 1e-9
    finally:
        vs.seaice_mesh_pool_destroy(dom)


@pytest.mark.parametrize("optional", [False, True])
def test_flux_kernel_variants_give_the_same_bits(oracle, monkeypatch, optional):
    """SIH_IR_FLUX_MODE: 2 (default) one thread per (edge, category) walking the tracer tree with the ancestors'
    values at the quadrature points kept, 1 one thread per (edge, tracer plane group), 0 one thread per (edge,
    category) with every tracer from scratch.  Same sums in the same order.  With the optional tracers the tree
    is three parents deep and has siblings with children on two levels."""
    results = []
    cfg = {"config_dt": 1800.0, "config_elastic_subcycle_number": 20}
    if optional:
        cfg["config_use_level_meltponds"] = True
    names = None
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("SIH_IR_FLUX_MODE", mode)
        dom = cases.make_domain(get_mesh("var12"), None, configs=cfg, open_water_band=(-0.2, 0.3))
        if optional:
            _add_optional_tracers(dom, vertical=False)
        vs.seaice_mesh_pool_create(dom)
        try:
            vs.seaice_init_velocity_solver(dom)
            adv.seaice_init_advection(dom)
            if mode == "2":
                assert dom.blocklist[0].ctx.info("irFluxWalk") == 1.0
            vs.seaice_run_velocity_solver(dom)
            adv.seaice_run_advection(dom)
            names = [t.tracerName for t in dom.solver_state["tracersHead"]]
        finally:
            vs.seaice_mesh_pool_destroy(dom)
        results.append(dom.blocklist[0].structs["tracers"])
    assert len(names) == (13 if optional else 7)
    for n in names:
        assert np.array_equal(results[0][n], results[1][n]), n
        assert np.array_equal(results[0][n], results[2][n]), n


@pytest.mark.parametrize("meshname,name,cfg", [
    ("hex_24_20", None, {"config_dt": 600.0, "config_elastic_subcycle_number": 60}),
    ("QU480", "QU480", {}),
])
def test_recover_tracer_means_check(oracle, meshname, name, cfg):
    """config_recover_tracer_means_check: the integrals of the reconstructions and the number of entries the
    reference would warn about are those of the oracle; the check changes nothing in the advection."""
    cfg = dict(cfg, config_recover_tracer_means_check=True)
    dom, ref = make_pair(oracle, meshname, name, configs=cfg)
    try:
        oracle.seaice_run_velocity_solver(ref)
        copy_velocity(ref, dom)
        diag = {}
        _ir().seaice_run_advection(ref, diagnostics=diag)
        with pytest.warns(UserWarning) if diag["recoverMeansWarnings"] else _nullcontext():
            adv.seaice_run_advection(dom)
        assert dom.solver_state["ir_recover_means_warnings"] == diag["recoverMeansWarnings"]
        compare_tracers(dom, ref)
        b = dom.blocklist[0]
        nC, nCat = b.dimensions["nCells"], b.dimensions["nCategories"]
        for t in dom.solver_state["tracersHead"]:
            want = diag[t.tracerName + "RecoveredProduct"][..., :nC].reshape(t.nLayers * nCat, nC, order="F")
            got = dget(b, t.tracerName + "RecoveredProduct", t.nLayers * nCat, nC)
            # device planes are (category, layer); the oracle's leading index is (layer, category)
            got = got.reshape(nCat, t.nLayers, nC).transpose(1, 0, 2).reshape(t.nLayers * nCat, nC, order="F")
            assert rel_linf(got, want) <= TOL, (t.tracerName, rel_linf(got, want))
    finally:
        vs.seaice_mesh_pool_destroy(dom)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def test_recover_tracer_means_counts_warnings(oracle):
    """Enthalpies of 0 between neighbours of +-3e8 J/m3: the products of the means are 0, the integrals of the
    (steep) reconstructions are roundoff of 1e-8, which the reference reports.  Device and oracle do the same
    operations in the same order, so they count the same entries."""
    mesh = get_mesh("hex_12_10")
    cfg = {"config_dt": 600.0, "config_elastic_subcycle_number": 60, "config_recover_tracer_means_check": True}
    ref = cases.make_domain(mesh, None, configs=cfg)
    dom = cases.make_domain(mesh, None, configs=cfg)
    rng = np.random.default_rng(11)
    e = ref.blocklist[0].structs["tracers"]["iceEnthalpy"]
    e[...] = rng.uniform(-3e8, 3e8, e.shape)
    e[..., rng.uniform(size=e.shape[-1]) < 0.3] = 0.0
    dom.blocklist[0].structs["tracers"]["iceEnthalpy"][...] = e
    oracle.seaice_init_velocity_solver(ref)
    _ir().seaice_init_advection(ref)
    vs.seaice_mesh_pool_create(dom)
    try:
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        oracle.seaice_run_velocity_solver(ref)
        copy_velocity(ref, dom)
        for _ in range(2):  # the count starts from zero in every step
            diag = {}
            _ir().seaice_run_advection(ref, diagnostics=diag)
            assert diag["recoverMeansWarnings"] > 0
            with pytest.warns(UserWarning, match="does not recover the mean"):
                adv.seaice_run_advection(dom)
            assert dom.solver_state["ir_recover_means_warnings"] == diag["recoverMeansWarnings"]
    finally:
        vs.seaice_mesh_pool_destroy(dom)
