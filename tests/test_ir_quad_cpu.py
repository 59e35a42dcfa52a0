"""The incremental remap on quadrilateral meshes (vertexDegree 4) in the CPU oracle: the extra side edges E5 / E6 and
cells C5 / C6 of /root/reference/src/shared/mpas_seaice_advection_incremental_remap.F:1133-1172, 1479-1578 and the
split of a side triangle where the line between the departure points crosses E5 / E6 (:5807-5958).  "Parity
unpinned" like the rest of the oracle; pinned here by the geometry of the stencil, exact translation of a linear field,
conservation, bounds, and uniform tracers staying uniform."""
import numpy as np
import pytest

from mpas_seaice_amd import cases, mesh as meshmod
from util import get_mesh


def _ir():
    from oracle import ir_flow
    return ir_flow


def quad_domain(meshname, dt, configs=None, **kw):
    """a domain for the advection alone (the velocity solver is for vertexDegree 3): velocities are set by the test"""
    # the band lies around the equator, where the rotated frame of config_rotate_cartesian_grid has its poles (the
    # reference's local east / north frames are singular there): the band is advected in the unrotated frame
    cfg = {"config_conservation_check": True, "config_rotate_cartesian_grid": False}
    cfg.update(configs or {})
    dom = cases.make_domain(get_mesh(meshname), None, configs=cfg, **kw)
    for b in dom.blocklist:
        b.structs["velocity_solver"]["dynamicsTimeStep"][0] = dt
    return dom


def set_velocity(b, fn):
    m = b.structs["mesh"]
    nV = b.dimensions["nVertices"]
    u, v = fn(m, nV)
    b.structs["velocity_solver"]["uVelocity"][:nV] = u
    b.structs["velocity_solver"]["vVelocity"][:nV] = v


def swirl(m, nV, speed=0.5):
    """planar: solid rotation about the domain centre plus a shear, so that the two departure points of an edge move
    differently and their line crosses the side edges and the extensions of the main edge"""
    x, y = m["xVertex"][:nV], m["yVertex"][:nV]
    cx, cy = 0.5 * x.max(), 0.5 * y.max()
    r = max(cx, cy)
    return (-speed * (y - cy) / r + 0.3 * speed * np.sin(6.0 * np.pi * y / y.max()),
            speed * (x - cx) / r + 0.3 * speed * np.sin(4.0 * np.pi * x / x.max()))


def zonal_wavy(m, nV, speed=20.0):
    lat, lon = m["latVertex"][:nV], m["lonVertex"][:nV]
    return speed * np.cos(lat) * (1.0 + 0.3 * np.sin(3.0 * lon)), 0.4 * speed * np.sin(2.0 * lon) * np.cos(lat)


@pytest.mark.parametrize("meshname", ["quad_12_9", "llquad_24_10"])
def test_quad_mesh_generators_are_consistent(meshname):
    m = get_mesh(meshname)
    rep = meshmod.check_mesh(m)
    assert rep["voc_cov_consistent"]
    N, nE, T = m.nCells, m.nEdges, m.nVertices
    assert m["vertexDegree"] == 4 and m["maxEdges"] == 4
    voe, coe, eoc, voc = (np.asarray(m[k]).astype(np.int64) - 1 for k in ("verticesOnEdge", "cellsOnEdge", "edgesOnCell", "verticesOnCell"))
    for c in range(N):   # the k-th edge of a cell joins its k-th and (k+1)-th vertex and has the cell on one side
        for k in range(4):
            e = eoc[k, c]
            assert {voe[0, e], voe[1, e]} == {voc[k, c], voc[(k + 1) % 4, c]}
            assert c in (coe[0, e], coe[1, e])
    eov = np.asarray(m["edgesOnVertex"]).astype(np.int64) - 1
    for v in range(T):
        es = [e for e in eov[:, v] if e < nE]
        assert len(es) == sum(1 for e in range(nE) if v in (voe[0, e], voe[1, e])) if T < 200 else len(es) >= 2
    if m["on_a_sphere"]:
        band = 2.0 * np.sin(np.deg2rad(60.0)) * 2.0 * np.pi * m["sphere_radius"] ** 2
        assert abs(m["areaCell"].sum() / band - 1.0) < 1e-12 and abs(m["areaTriangle"].sum() / band - 1.0) < 1e-12


def test_remap_stencil_of_an_interior_edge_follows_the_diagram():
    ir = _ir()
    dom = quad_domain("quad_12_9", 600.0)
    ir.seaice_init_advection(dom)
    b = dom.blocklist[0]
    m, r = b.structs["mesh"], b.structs["incremental_remap"]
    nE, nC = b.dimensions["nEdges"], b.dimensions["nCells"]
    dc = 1.28e6 / 12
    checked = 0
    for e in range(nE):
        if r["remapEdge"][e] != 1:
            continue
        cells, edges = r["cellsOnEdgeRemap"][:, e], r["edgesOnEdgeRemap"][:, e]
        if (cells > nC).any() or (edges > nE).any() or (edges < 1).any():
            continue
        checked += 1
        assert len(set(cells)) == 6 and len(set(edges) | {e + 1}) == 7
        v1, v2 = m["verticesOnEdge"][:, e] - 1
        t = np.array([m["xVertex"][v2] - m["xVertex"][v1], m["yVertex"][v2] - m["yVertex"][v1]]) / dc   # along the edge
        n = np.array([-t[1], t[0]])                                                                          # to the left
        mid = np.array([m["xEdge"][e], m["yEdge"][e]])
        want = {0: 0.5 * n, 1: -0.5 * n, 2: 0.5 * n - t, 3: 0.5 * n + t, 4: -0.5 * n - t, 5: -0.5 * n + t}  # C1..C6
        for k, off in want.items():
            c = cells[k] - 1
            assert np.allclose([m["xCell"][c], m["yCell"][c]], mid + off * dc, atol=1e-6 * dc), (e, k)
        # the far vertices V3..V8 in the edge's frame (x/yVertexOnEdge rows 3..8)
        far = {2: -0.5 * t + n, 3: 0.5 * t + n, 4: -0.5 * t - n, 5: 0.5 * t - n, 6: -1.5 * t, 7: 1.5 * t}
        for k, off in far.items():
            assert np.allclose([r["xVertexOnEdge"][k, e], r["yVertexOnEdge"][k, e]], off * dc, atol=1e-6 * dc), (e, k)
    assert checked > 50


def test_planar_translation_of_linear_field_is_exact_on_squares():
    ir = _ir()
    dom = quad_domain("quad_16_12", 600.0)
    ir.seaice_init_advection(dom)
    b = dom.blocklist[0]
    m = b.structs["mesh"]
    nC, nV, nCat = b.dimensions["nCells"], b.dimensions["nVertices"], b.dimensions["nCategories"]
    u0, v0 = 6.0, -3.5     # a good tenth of a cell per step, diagonally: side triangles in the corner cells
    b.structs["velocity_solver"]["uVelocity"][:nV] = u0
    b.structs["velocity_solver"]["vVelocity"][:nV] = v0
    Lx = m["xCell"][:nC].max()
    f = lambda x, y: 0.1 + 0.05 * x / Lx + 0.03 * y / Lx   # noqa: E731
    tp = b.structs["tracers"]
    for k in tp:
        tp[k][:] = 0.0
    for k in range(nCat):
        tp["iceAreaCategory"][0, k, :nC] = f(m["xCell"][:nC], m["yCell"][:nC]) * (1.0 + 0.1 * k)
        tp["iceVolumeCategory"][0, k, :nC] = 2.0 * tp["iceAreaCategory"][0, k, :nC]
    ir.seaice_run_advection(dom)
    r = b.structs["incremental_remap"]
    ntri = (r["triangleArea"][:, :b.dimensions["nEdges"]] != 0.0).sum(axis=0)
    assert ntri.max() == 3                      # two central triangles and one in a corner cell
    coc = m["cellsOnCell"][:, :nC]
    interior = (coc <= nC).all(axis=0)
    inner = interior.copy()
    for i in range(4):
        inner &= interior[np.minimum(coc[i] - 1, nC - 1)]
    exact = f(m["xCell"][:nC] - u0 * 600.0, m["yCell"][:nC] - v0 * 600.0)
    for k in range(nCat):
        err = np.abs(tp["iceAreaCategory"][0, k, :nC] - exact * (1.0 + 0.1 * k))[inner].max()
        assert err < 1e-10, err
        h = tp["iceVolumeCategory"][0, k, :nC][inner] / tp["iceAreaCategory"][0, k, :nC][inner]
        assert np.abs(h - 2.0).max() < 1e-11


@pytest.mark.parametrize("meshname,flow,dt", [("quad_24_18", swirl, 7200.0), ("llquad_48_20", zonal_wavy, 3600.0)])
def test_conservation_bounds_and_uniform_tracers_on_quadrilaterals(meshname, flow, dt):
    ir = _ir()
    dom = quad_domain(meshname, dt, open_water_band=(-0.2, 0.3) if meshname.startswith("ll") else None)
    ir.seaice_init_advection(dom)
    b = dom.blocklist[0]
    set_velocity(b, flow)
    nC, nE = b.dimensions["nCells"], b.dimensions["nEdges"]
    tp = b.structs["tracers"]
    a = tp["iceAreaCategory"]
    tp["surfaceTemperature"][..., :nC] = -3.0            # uniform tracers on area, volume and snow
    tp["iceEnthalpy"][..., :nC] = -2.5e8
    tp["snowEnthalpy"][..., :nC] = -1.2e8
    area = b.structs["mesh"]["areaCell"][:nC]
    before = {k: v.copy() for k, v in tp.items()}
    for _ in range(3):
        ir.seaice_run_advection(dom)   # raises on a conservation error > 1e-11 of any mass * tracer product
    r = b.structs["incremental_remap"]
    ntri = (r["triangleArea"][:, :nE] != 0.0).sum(axis=0)
    cells = r["cellsOnEdgeRemap"][:, :nE]
    in56 = ((r["iCellTriangle"][:, :nE] == cells[4][None]) | (r["iCellTriangle"][:, :nE] == cells[5][None])) & (r["triangleArea"][:, :nE] != 0.0)
    print(meshname, "triangles per edge", np.bincount(ntri), "triangles in C5/C6", int(in56.sum()))
    assert in56.any()                                     # the quadrilateral-only source cells are in use
    a1 = tp["iceAreaCategory"][..., :nC]
    assert np.abs(a1 - before["iceAreaCategory"][..., :nC]).max() > 1e-4
    assert a1.min() >= 0.0
    for nm in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"):
        s0, s1 = (before[nm][0, :, :nC] * area).sum(), (tp[nm][0, :, :nC] * area).sum()
        assert abs(s1 - s0) <= 1e-11 * abs(s0)
    ok = a1 > 1e-11
    assert np.abs(tp["surfaceTemperature"][..., :nC][ok] + 3.0).max() < 1e-11
    okL = np.broadcast_to(tp["iceVolumeCategory"][..., :nC] > 1e-11, tp["iceEnthalpy"][..., :nC].shape)
    assert np.abs(tp["iceEnthalpy"][..., :nC][okL] / -2.5e8 - 1.0).max() < 1e-11
    # thickness stays within the range of the neighbourhood (two rings, as the reference's check allows)
    a0 = before["iceAreaCategory"][..., :nC]
    h0 = np.where(a0 > 0, before["iceVolumeCategory"][..., :nC] / np.where(a0 > 0, a0, 1.0), 0.0)
    h1 = np.where(ok, tp["iceVolumeCategory"][..., :nC] / np.where(ok, a1, 1.0), 0.0)
    assert h1[ok].max() <= h0[a0 > 0].max() * (1 + 1e-9) and h1[ok].min() >= h0[a0 > 0].min() * (1 - 1e-9)


def test_split_side_triangles_appear_under_strong_rotation():
    """The line between the two departure points crosses E5 / E6 only when they lie on opposite sides of the main
    edge's line beyond one of its ends: a strong local rotation.  Such edges get the two-part side triangle."""
    ir = _ir()
    dom = quad_domain("quad_24_18", 7200.0)
    ir.seaice_init_advection(dom)
    b = dom.blocklist[0]
    rng = np.random.default_rng(5)
    nV, nE = b.dimensions["nVertices"], b.dimensions["nEdges"]
    b.structs["velocity_solver"]["uVelocity"][:nV] = rng.uniform(-2.5, 2.5, nV)   # up to a third of a cell per step
    b.structs["velocity_solver"]["vVelocity"][:nV] = rng.uniform(-2.5, 2.5, nV)
    ir.seaice_run_advection(dom)
    r = b.structs["incremental_remap"]
    ntri = (r["triangleArea"][:, :nE] != 0.0).sum(axis=0)
    print("triangles per edge", np.bincount(ntri))
    assert ntri.max() >= 4 and ntri.max() <= 6
