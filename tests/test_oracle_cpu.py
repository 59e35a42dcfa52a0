"""The CPU oracle against the analytic known answers of the reference's idealised test cases
and against invariants of the variational scheme (no golden vectors exist in the reference)."""
import numpy as np
import pytest

import analytic
from mpas_seaice_amd import cases, mesh as meshmod
from mpas_seaice_amd.cases import rotated_lonlat
from util import get_mesh


def test_analytic_restatements_match_reference_samples():
    s = analytic.SPHERE_SAMPLE
    got = analytic.velocities_strains_analytical(s["lat"], s["lon"])
    for g, k in zip(got, ("u", "v", "e11", "e22", "e12")):
        assert abs(g - s[k]) < 5e-11 * max(1.0, abs(s[k])), k
    p = analytic.PLANE_SAMPLE
    got = analytic.planar_fields(p["x"], p["y"])
    for g, k in zip(got, ("u", "v", "e11", "e22", "e12", "divu", "divv")):
        assert abs(g - p[k]) < 5e-11 * max(1.0, abs(p[k])), k


def _init(oracle, mesh, configs=None):
    dom = cases.make_domain(mesh, None, configs=configs)
    oracle.seaice_init_velocity_solver(dom)
    return dom


def _strain(oracle, dom, u, v):
    b = dom.blocklist[0]
    d, m, vv = b.dimensions, b.structs["mesh"], b.structs["velocity_variational"]
    nC, nV, M = d["nCells"], d["nVertices"], d["maxEdges"]
    uu = np.zeros(nV + 1)
    vw = np.zeros(nV + 1)
    uu[:nV], vw[:nV] = u, v
    ss = np.ones(nC + 1, dtype=np.int32)
    e11, e22, e12 = (np.zeros((M, nC + 1), order="F") for _ in range(3))
    oracle.call("ora_strain_tensor_variational", nC, M, m["nEdgesOnCell"], m["verticesOnCell"], e11, e22, e12, uu, vw,
                vv["basisGradientU"], vv["basisGradientV"], vv["tanLatVertexRotatedOverRadius"], ss)
    return e11, e22, e12


def sphere_strain_error(oracle, n):
    mesh = meshmod.make_sphere_mesh(n, radius=1.0)
    dom = _init(oracle, mesh)
    b = dom.blocklist[0]
    m = b.structs["mesh"]
    nC, nV = mesh.nCells, mesh.nVertices
    lon, lat = rotated_lonlat(m["xVertex"][:nV], m["yVertex"][:nV], m["zVertex"][:nV], 1.0)
    u, v, a11, a22, a12 = analytic.velocities_strains_analytical(lat, lon)
    e11, e22, e12 = _strain(oracle, dom, u, v)
    voc = m["verticesOnCell"].astype(np.int64) - 1
    err2, nrm2 = 0.0, 0.0
    for i in range(mesh.maxEdges):
        act = (m["nEdgesOnCell"][:nC] > i)
        vi = voc[i, :nC][act]
        keep = np.abs(lat[vi]) < np.deg2rad(70.0)  # the reference's norm also stays away from the grid poles
        for e, a in ((e11, a11), (e22, a22), (e12, a12)):
            err2 += ((e[i, :nC][act][keep] - a[vi][keep]) ** 2).sum()
            nrm2 += (a[vi][keep] ** 2).sum()
    return np.sqrt(err2 / nrm2)


def test_sphere_strain_converges_to_spherical_harmonic_answer(oracle):
    e8, e16, e32 = (sphere_strain_error(oracle, n) for n in (8, 16, 32))
    assert e32 < e16 < e8
    assert e32 < 0.08
    # at least first-order convergence of the L2 norm under mesh refinement
    assert np.log2(e16 / e32) > 0.9, (e8, e16, e32)


def test_analytic_stress_divergence_is_the_derivative_of_the_analytic_strain():
    """The second-derivative formulas restated from the reference's stress-divergence case, checked against
    central differences of the (sample-pinned) analytic strains with the same metric terms."""
    lat, lon, h = 0.7, 1.1, 1e-5
    divu, divv = analytic.stress_divergence_analytical(lat, lon)

    def e(la, lo):
        return analytic.velocities_strains_analytical(la, lo)[2:]
    d_lon = [(a - b) / (2 * h) for a, b in zip(e(lat, lon + h), e(lat, lon - h))]
    d_lat = [(a - b) / (2 * h) for a, b in zip(e(lat + h, lon), e(lat - h, lon))]
    e11, e22, e12 = e(lat, lon)
    c, t = np.cos(lat), np.tan(lat)
    fu = d_lon[0] / c + d_lat[2] - 2.0 * t * e12
    fv = d_lon[2] / c + d_lat[1] + t * e11 - t * e22
    assert abs(fu - divu) < 1e-6 * abs(divu) and abs(fv - divv) < 1e-6 * abs(divv)


def sphere_stress_divergence_error(oracle, n):
    """stress := analytic strain at the cell corners; the variational divergence against the analytic one
    (the reference's spherical_operators/stress_divergence case)"""
    mesh = meshmod.make_sphere_mesh(n, radius=1.0)
    dom = _init(oracle, mesh)
    b = dom.blocklist[0]
    d, m, vv = b.dimensions, b.structs["mesh"], b.structs["velocity_variational"]
    nC, nV, M, D = mesh.nCells, mesh.nVertices, d["maxEdges"], d["vertexDegree"]
    lon, lat = rotated_lonlat(m["xVertex"][:nV], m["yVertex"][:nV], m["zVertex"][:nV], 1.0)
    _, _, a11, a22, a12 = analytic.velocities_strains_analytical(lat, lon)
    du, dv = analytic.stress_divergence_analytical(lat, lon)
    voc = m["verticesOnCell"].astype(np.int64) - 1
    s11, s22, s12 = (np.zeros((M, nC + 1), order="F") for _ in range(3))
    for i in range(M):
        act = m["nEdgesOnCell"][:nC] > i
        vi = voc[i, :nC][act]
        s11[i, :nC][act], s22[i, :nC][act], s12[i, :nC][act] = a11[vi], a22[vi], a12[vi]
    divU, divV = np.zeros(nV + 1), np.zeros(nV + 1)
    sv = np.ones(nV + 1, dtype=np.int32)
    oracle.call("ora_stress_divergence_variational", nV, D, M, m["nEdgesOnCell"], m["cellsOnVertex"],
                vv["cellVerticesAtVertex"], divU, divV, s11, s22, s12, vv["basisIntegralsU"], vv["basisIntegralsV"],
                vv["basisIntegralsMetric"], vv["variationalDenominator"], vv["tanLatVertexRotatedOverRadius"], sv)
    keep = np.abs(lat) < np.deg2rad(70.0)
    err2 = ((divU[:nV][keep] - du[keep]) ** 2).sum() + ((divV[:nV][keep] - dv[keep]) ** 2).sum()
    return np.sqrt(err2 / ((du[keep] ** 2).sum() + (dv[keep] ** 2).sum()))


def test_sphere_stress_divergence_converges_to_analytic_answer(oracle):
    e8, e16, e32 = (sphere_stress_divergence_error(oracle, n) for n in (8, 16, 32))
    assert e32 < e16 < e8, (e8, e16, e32)
    assert np.log2(e16 / e32) > 0.9, (e8, e16, e32)


def test_planar_operators_linear_exact_and_sinusoid_converges(oracle):
    errs = []
    for nx in (16, 32):
        mesh = meshmod.make_planar_hex_mesh(nx, nx, 1.0 / nx)
        dom = _init(oracle, mesh, {"config_include_metric_terms": False})
        b = dom.blocklist[0]
        m = b.structs["mesh"]
        nC, nV = mesh.nCells, mesh.nVertices
        x, y = m["xVertex"][:nV], m["yVertex"][:nV]
        # linear field: the Wachspress gradient is exact
        u, v, a11, a22, a12, _, _ = analytic.planar_fields(x, y, "linear")
        e11, e22, e12 = _strain(oracle, dom, u, v)
        assert np.abs(e11[:, :nC] - 1.0).max() < 1e-11
        assert np.abs(e22[:, :nC] - 1.0).max() < 1e-11
        assert np.abs(e12[:, :nC]).max() < 1e-11
        u, v, a11, a22, a12, du, dv = analytic.planar_fields(x, y, "sinusoid", A=1.0)
        e11, e22, e12 = _strain(oracle, dom, u, v)
        voc = m["verticesOnCell"].astype(np.int64) - 1
        err = max(np.abs(e11[i, :nC] - a11[voc[i, :nC]]).max() for i in range(6)) / np.abs(a11).max()
        errs.append(err)
    assert errs[1] < 0.6 * errs[0]


def test_basis_invariants(oracle):
    mesh = get_mesh("QU480")
    dom = _init(oracle, mesh)
    b = dom.blocklist[0]
    vv = b.structs["velocity_variational"]
    nC = mesh.nCells
    gU, gV = vv["basisGradientU"][:, :, :nC], vv["basisGradientV"][:, :, :nC]
    scale = np.abs(gU).max()
    # partition of unity: the basis gradients sum to zero at every corner
    assert np.abs(gU.sum(axis=0)).max() < 1e-12 * scale
    assert np.abs(gV.sum(axis=0)).max() < 1e-12 * scale
    # only the basis vertex and its two neighbours carry a gradient (wachspress.F:1178-1191)
    nz = (gU != 0).sum(axis=0)
    assert nz.max() == 3
    SU, SM = vv["basisIntegralsU"][:, :, :nC], vv["basisIntegralsMetric"][:, :, :nC]
    # sum_j dN_j/dx = 0 under the integral, and sum_sj int N_s N_j = planar cell area ~ areaCell
    assert np.abs(SU.sum(axis=1)).max() < 1e-9 * np.abs(SU).max()
    rel = np.abs(SM.sum(axis=(0, 1)) / mesh["areaCell"] - 1.0).max()
    assert rel < 5e-3
    # metric integrals are symmetric up to rounding
    assert np.abs(SM - SM.transpose(1, 0, 2)).max() < 1e-12 * np.abs(SM).max()


def test_full_step_is_finite_and_deterministic(oracle):
    mesh = get_mesh("geo8")
    out = []
    for threads in (1, 4):
        oracle.set_num_threads(threads)
        dom = cases.make_domain(mesh, None, configs={"config_dt": 7200.0})
        oracle.seaice_init_velocity_solver(dom)
        oracle.seaice_run_velocity_solver(dom)
        out.append(dom.blocklist[0].structs["velocity_solver"]["uVelocity"].copy())
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 1e-3
    assert np.array_equal(out[0], out[1])


def test_average_strains_on_vertex_properties(oracle):
    """seaice_average_strains_on_vertex (variational.F:684-763): a corner strain that is the same around a
    vertex stays put; otherwise every corner at the vertex gets the area-weighted mean."""
    mesh = get_mesh("hex_8_6")
    dom = cases.make_domain(mesh, None)
    oracle.seaice_init_velocity_solver(dom)
    b = dom.blocklist[0]
    d, m, vv = b.dimensions, b.structs["mesh"], b.structs["velocity_variational"]
    nC, nV, M = d["nCells"], d["nVertices"], d["maxEdges"]
    rng = np.random.default_rng(3)
    s = [np.asfortranarray(rng.standard_normal((M, nC + 1))) for _ in range(3)]
    s0 = [a.copy() for a in s]
    oracle.call("ora_average_strains_on_vertex", nC, d["nVerticesSolve"], 3, M, m["cellsOnVertex"],
                vv["cellVerticesAtVertex"], m["areaCell"], *s)
    cov, cva, area = m["cellsOnVertex"], vv["cellVerticesAtVertex"], m["areaCell"]
    for v in range(nV):
        cs = [(cov[k, v] - 1, cva[k, v] - 1) for k in range(3) if cov[k, v] <= nC]
        for a0, a1 in zip(s0, s):
            mean = sum(a0[j, c] * area[c] for c, j in cs) / sum(area[c] for c, _ in cs)
            for c, j in cs:
                assert abs(a1[j, c] - mean) <= 1e-15 * max(1.0, abs(mean))
    again = [a.copy(order="F") for a in s]
    oracle.call("ora_average_strains_on_vertex", nC, d["nVerticesSolve"], 3, M, m["cellsOnVertex"],
                vv["cellVerticesAtVertex"], m["areaCell"], *again)
    for a1, a2 in zip(s, again):  # idempotent up to rounding of the mean of equal numbers
        assert np.abs(a1 - a2).max() <= 4e-16 * np.abs(a1).max()


def test_special_boundaries_velocity_is_the_sequential_in_place_loop(oracle):
    """special_boundaries.F:301-324 restated in plain Python on random types and sources (chains included)"""
    rng = np.random.default_rng(9)
    n = 200
    typ = rng.choice([0, 0, 1, 2, 3], size=n).astype(np.int32)
    src = rng.integers(1, n + 1, size=n).astype(np.int32)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    eu, ev = u.copy(), v.copy()
    for i in range(n):
        if typ[i] == 1:
            eu[i], ev[i] = eu[src[i] - 1], ev[src[i] - 1]
        elif typ[i] == 2:
            eu[i], ev[i] = -eu[src[i] - 1], -ev[src[i] - 1]
        elif typ[i] == 3:
            eu[i], ev[i] = 0.0, 0.0
    oracle.call("ora_set_special_boundaries_velocity", n, typ, src, u, v)
    assert np.array_equal(u, eu) and np.array_equal(v, ev)


@pytest.mark.parametrize("mesh_name", ["QU480", "hex_8_6", "var12"])
def test_pwl_basis_invariants(oracle, mesh_name):
    """Piecewise-linear basis (velocity_solver_pwl.F:44-373): the basis functions sum to one, so their
    gradients sum to zero at every corner and the strain of a linear velocity field is exact; the
    metric integrals sum to the cell area; the U / V integrals of each basis function sum (over stress
    corners) to the integral of its gradient, sum_s grad(s) * subArea(s)."""
    mesh = get_mesh(mesh_name)
    dom = cases.make_domain(mesh, mesh_name if mesh_name.startswith("QU") else None,
                            configs={"config_variational_basis": "pwl"})
    oracle.seaice_init_velocity_solver(dom)
    b = dom.blocklist[0]
    vv, m = b.structs["velocity_variational"], b.structs["mesh"]
    nC, M = b.dimensions["nCells"], b.dimensions["maxEdges"]
    gu, gv = vv["basisGradientU"][:, :, :nC], vv["basisGradientV"][:, :, :nC]
    scale = np.abs(gu).max()
    assert np.abs(gu.sum(axis=0)).max() <= 1e-13 * scale and np.abs(gv.sum(axis=0)).max() <= 1e-13 * scale
    sm = vv["basisIntegralsMetric"][:, :, :nC].sum(axis=(0, 1))
    assert np.abs(sm / m["areaCell"][:nC] - 1.0).max() <= 1e-13
    # linear field u = a x + b y in the local frame of each cell: du/dx = a, du/dy = b at every corner
    xl, yl = b._xLocal, b._yLocal
    ne = m["nEdgesOnCell"][:nC]
    for c in range(0, nC, max(1, nC // 50)):
        n = ne[c]
        u = 0.3 * xl[:n, c] - 1.7 * yl[:n, c] + 4.0
        for g in range(n):
            assert abs((u * gu[:n, g, c]).sum() - 0.3) <= 1e-10
            assert abs((u * gv[:n, g, c]).sum() + 1.7) <= 1e-10


@pytest.mark.parametrize("mesh_name", ["hex_8_6", "QU480"])
def test_weak_operators(oracle, mesh_name):
    """Weak scheme (velocity_solver_weak.F): polygon normals are unit vectors that close (sum n dv = 0 on
    the plane); the strain of a linear velocity field on the plane is exact in every interior cell; the
    divergence of a uniform stress field vanishes at interior vertices."""
    mesh = get_mesh(mesh_name)
    cfg = {"config_strain_scheme": "weak", "config_stress_divergence_scheme": "weak"}
    dom = cases.make_domain(mesh, mesh_name if mesh_name.startswith("QU") else None, configs=cfg)
    oracle.seaice_init_velocity_solver(dom)
    b = dom.blocklist[0]
    d, m, vw = b.dimensions, b.structs["mesh"], b.structs["velocity_weak"]
    nC, nV, M = d["nCells"], d["nVertices"], d["maxEdges"]
    ne = m["nEdgesOnCell"][:nC]
    npoly = vw["normalVectorPolygon"][:, :, :nC]
    live = np.arange(M)[:, None] < ne[None, :]
    assert np.abs(np.sqrt((npoly ** 2).sum(axis=0))[live] - 1.0).max() < 1e-14
    ntri = vw["normalVectorTriangle"][:, :, :nV]
    interior = b.structs["boundary"]["interiorVertex"][:nV] == 1
    assert np.abs(np.sqrt((ntri ** 2).sum(axis=0))[:, interior] - 1.0).max() < 1e-14
    if mesh_name.startswith("hex"):
        dv = m["dvEdge"][np.clip(m["edgesOnCell"][:, :nC] - 1, 0, d["nEdges"] - 1)]
        closure = (npoly * np.where(live, dv, 0.0)[None, :, :]).sum(axis=1)
        assert np.abs(closure).max() < 1e-9 * dv.max()
        # linear velocity u = a x + b y + c, v = e x + f y + g
        a_, b_, e_, f_ = 2.0e-6, -1.0e-6, 0.5e-6, 3.0e-6
        u = np.zeros(nV + 1)
        v = np.zeros(nV + 1)
        u[:nV] = a_ * m["xVertex"][:nV] + b_ * m["yVertex"][:nV] + 0.3
        v[:nV] = e_ * m["xVertex"][:nV] + f_ * m["yVertex"][:nV] - 0.1
        s = [np.zeros(nC + 1) for _ in range(3)]
        solve = np.ones(nC + 1, dtype=np.int32)
        oracle.call("ora_strain_tensor_weak", nC, M, m["nEdgesOnCell"], m["verticesOnCell"], m["edgesOnCell"],
                    m["verticesOnEdge"], m["dvEdge"], m["areaCell"], 0.0, s[0], s[1], s[2], u, v,
                    vw["normalVectorPolygon"], vw["latCellRotated"], solve)
        assert np.abs(s[0][:nC] - a_).max() < 1e-12 * abs(a_) * 1e3
        assert np.abs(s[1][:nC] - f_).max() < 1e-12 * abs(f_) * 1e3
        assert np.abs(s[2][:nC] - 0.5 * (b_ + e_)).max() < 1e-15
        # uniform stress: no force on interior vertices
        st = [np.full(nC + 1, x) for x in (3.0, -2.0, 0.7)]
        for x in st:
            x[nC] = 0.0
        dU, dV = np.zeros(nV + 1), np.zeros(nV + 1)
        sv = np.zeros(nV + 1, dtype=np.int32)
        sv[:nV] = interior
        oracle.call("ora_stress_divergence_weak", nV, 3, m["cellsOnVertex"], m["edgesOnVertex"], m["cellsOnEdge"],
                    m["areaTriangle"], m["dcEdge"], 0.0, dU, dV, st[0], st[1], st[2], vw["normalVectorTriangle"],
                    vw["latVertexRotated"], sv)
        scale = 3.0 * m["dcEdge"].max() / m["areaTriangle"][:nV].min()
        assert np.abs(dU[:nV][interior]).max() < 1e-12 * scale and np.abs(dV[:nV][interior]).max() < 1e-12 * scale


def test_special_boundaries_tracers_sequential_in_place():
    """seaice_set_special_boundaries_tracers: zeroed cells, copied cells, and a copy whose source was itself
    replaced earlier in the loop (special_boundaries.F:460-478)."""
    from mpas_seaice_amd import special_boundaries as sbm
    dom = cases.make_domain(get_mesh("QU480"), "QU480", configs={"config_use_special_boundaries_tracers": True})
    b = dom.blocklist[0]
    sb, tr = b.structs["special_boundaries"], b.structs["tracers"]
    ids = b.structs["mesh"]["indexToCellID"]
    sb["tracerBoundaryType"][[4, 10, 20]] = [sbm.TRACER_BOUNDARY_ZERO, sbm.TRACER_BOUNDARY_SET, sbm.TRACER_BOUNDARY_SET]
    sb["tracerBoundarySource"][10] = ids[30]   # cell 11 takes cell 31
    sb["tracerBoundarySource"][20] = ids[10]   # cell 21 takes cell 11, which by then holds cell 31's values
    before = {k: v.copy(order="F") for k, v in tr.items()}
    sbm.seaice_init_special_boundaries(dom)
    assert sb["tracerBoundarySourceLocal"][10] == 31 and sb["tracerBoundarySourceLocal"][20] == 11
    sbm.seaice_set_special_boundaries_tracers(dom)
    for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"):
        assert np.abs(tr[k][:, :, 4]).max() == 0.0
        assert np.array_equal(tr[k][:, :, 10], before[k][:, :, 30])
        assert np.array_equal(tr[k][:, :, 20], before[k][:, :, 30])
        untouched = np.ones(tr[k].shape[-1], dtype=bool)
        untouched[[4, 10, 20]] = False
        assert np.array_equal(tr[k][..., untouched], before[k][..., untouched])
    assert np.array_equal(tr["iceEnthalpy"], before["iceEnthalpy"])


def _column_strength(oracle, aicen, vicen, aice0, opts):
    """one cell through ora_ice_strength_column"""
    n = len(aicen)
    a = np.zeros((1, n, 2), order="F")
    v = np.zeros((1, n, 2), order="F")
    a[0, :, 0], v[0, :, 0] = aicen, vicen
    P = np.zeros(2)
    oracle.call("ora_ice_strength_column", 1, n, np.ones(2, dtype=np.int32), np.array([sum(aicen), 0.0]),
                np.array([sum(vicen), 0.0]), np.array([aice0, 0.0]), a, v, opts[0], opts[1], opts[2], float(opts[3]),
                float(opts[4]), P)
    return P[0]


def test_column_package_ice_strength_known_values(oracle):
    """colpkg_ice_strength restated: closed forms for one category of thickness h and no open water.
    All ice then participates (apartic(1) = 1 for both participation functions) and the strength is
    Cf Cp (-h^2 + <h_r^2> / k) / (1 - 1/k) with k the mean ridge thickness ratio."""
    Cf, Cp = 17.0, 0.5 * 9.80616 * (1026.0 - 917.0) * 917.0 / 1026.0
    h = 2.0
    # exponential redistribution: hrmin = min(2h, h + 1), hrexp = mu sqrt(h), k = (hrmin + hrexp) / h
    hrmin, hrexp = min(2 * h, h + 1.0), 3.0 * np.sqrt(h)
    k = (hrmin + hrexp) / h
    want = Cf * Cp * (-h * h + (hrmin ** 2 + 2 * hrmin * hrexp + 2 * hrexp ** 2) / k) / (1.0 - 1.0 / k)
    for partic in (0, 1):
        got = _column_strength(oracle, [1.0], [h], 0.0, (1, partic, 1, 3.0, Cf))
        assert abs(got - want) <= 1e-12 * want, (partic, got, want)
    # Hibler 1980 redistribution: ridges uniform between hrmin and hrmax = 2 sqrt(25 h)
    hrmax = 2.0 * np.sqrt(25.0 * h)
    k = 0.5 * (hrmin + hrmax) / h
    want = Cf * Cp * (-h * h + (hrmax ** 3 - hrmin ** 3) / (3.0 * (hrmax - hrmin)) / k) / (1.0 - 1.0 / k)
    got = _column_strength(oracle, [1.0], [h], 0.0, (1, 1, 0, 3.0, Cf))
    assert abs(got - want) <= 1e-12 * want
    # kstrength = 0 is Hibler's formula
    got = _column_strength(oracle, [0.3, 0.5], [0.3, 1.5], 0.2, (0, 1, 1, 3.0, Cf))
    assert abs(got - 2.75e4 * 1.8 * np.exp(-20.0 * (1.0 - 0.8))) <= 1e-9
    # open water takes part in the closing first: with 20 % open water the pack is much weaker than without
    weak = _column_strength(oracle, [0.3, 0.5], [0.3, 1.5], 0.2, (1, 1, 1, 3.0, Cf))
    strong = _column_strength(oracle, [0.375, 0.625], [0.375, 1.875], 0.0, (1, 1, 1, 3.0, Cf))
    assert 0.0 < weak < 0.1 * strong


def test_trapezoidal_integration_rule(oracle):
    """config_wachspress_integration_type = 'trapezoidal': the rule integrates constants and linear functions on
    the unit triangle exactly, and the basis integrals it produces approach the order-8 Dunavant ones."""
    for order in (1, 4, 9):
        u, v, w, norm = oracle.integration_rule({"config_wachspress_integration_type": "trapezoidal",
                                                 "config_wachspress_integration_order": order})
        assert u.size == (order + 1) * (order + 2) // 2
        assert abs(w.sum() / norm - 0.5) < 1e-14                 # area of the unit triangle
        assert abs((w * u).sum() / norm - 1.0 / 6.0) < 1e-14     # integral of x
        assert abs((w * v).sum() / norm - 1.0 / 6.0) < 1e-14
    with pytest.raises(ValueError):
        oracle.integration_rule({"config_wachspress_integration_type": "fekete", "config_wachspress_integration_order": 7})
    mesh = get_mesh("QU480")
    ref = _init(oracle, mesh)
    SM8 = ref.blocklist[0].structs["velocity_variational"]["basisIntegralsMetric"]
    err = []
    for order in (4, 16):
        dom = _init(oracle, mesh, {"config_wachspress_integration_type": "trapezoidal",
                                   "config_wachspress_integration_order": order})
        SM = dom.blocklist[0].structs["velocity_variational"]["basisIntegralsMetric"]
        err.append(np.abs(SM - SM8).max() / np.abs(SM8).max())
    assert err[1] < 0.25 * err[0] and err[1] < 5e-3, err
