"""The incremental remap on quadrilateral meshes (vertexDegree 4) on the device against the CPU oracle, through the
host mirror and the C ABI: set-up geometry with the six side edges and cells of an edge, departure triangles with the
split at E5 / E6 (/root/reference/src/shared/mpas_seaice_advection_incremental_remap.F:5807-5958), and the advected
tracers.  Bar: 1e-10 relative L-infinity (stated below); the arithmetic is the oracle's statement for statement, so the
differences are expected to be zero.  The oracle is "parity unpinned" (tests/test_ir_quad_cpu.py holds what pins it)."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, velocity_solver as vs
from mpas_seaice_amd.lib import SeaiceHipError
from test_ir_gpu import TRACERS, dget
from test_ir_quad_cpu import quad_domain, set_velocity, swirl, zonal_wavy
from util import rel_linf

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _ir():
    from oracle import ir_flow
    return ir_flow


def random_flow(m, nV, speed=2.5):
    rng = np.random.default_rng(5)
    return rng.uniform(-speed, speed, nV), rng.uniform(-speed, speed, nV)


def _pair(meshname, dt, **kw):
    ref = quad_domain(meshname, dt, **kw)
    _ir().seaice_init_advection(ref)
    dom = quad_domain(meshname, dt, **kw)
    vs.seaice_mesh_pool_create(dom)
    adv.seaice_init_advection(dom)
    return dom, ref


@pytest.mark.parametrize("meshname", ["quad_12_9", "llquad_24_10"])
def test_quad_geometry_matches_oracle(meshname):
    dom, ref = _pair(meshname, 600.0)
    try:
        b, r = dom.blocklist[0], ref.blocklist[0]
        d = b.dimensions
        nC, nV, nE, M = d["nCells"], d["nVertices"], d["nEdges"], d["maxEdges"]
        ir = r.structs["incremental_remap"]
        rep = {}
        for nm in ("xVertexOnCell", "yVertexOnCell"):
            rep[nm] = rel_linf(dget(b, nm, M, nC), ir[nm][:, :nC])
        remap = dget(b, "remapEdge", 1, nE, np.int32)[0]
        assert np.array_equal(remap, ir["remapEdge"][:nE])
        on = remap == 1
        for nm in ("xVertexOnEdge", "yVertexOnEdge"):   # eight vertices per edge: V1, V2 and the far ends of E1..E6
            rep[nm] = rel_linf(dget(b, nm, 8, nE)[:, on], ir[nm][:8, :nE][:, on])
        rep["geomAvgCell"] = max(rel_linf(dget(b, "geomAvgCell", 14, nC)[k], ir["geomAvgCell"][:nC, k]) for k in range(14))
        rep["minLength"] = rel_linf(dget(b, "minLengthEdgesOnVertex", 1, nV)[0], ir["minLengthEdgesOnVertex"][:nV])
        print(meshname, rep)
        assert max(rep.values()) <= 1e-14, rep
    finally:
        vs.seaice_mesh_pool_destroy(dom)


@pytest.mark.parametrize("meshname,flow,dt,kw", [
    ("quad_24_18", swirl, 7200.0, {}),
    ("quad_24_18", random_flow, 7200.0, {}),                                   # split side triangles, up to 6 an edge
    ("llquad_48_20", zonal_wavy, 3600.0, {"open_water_band": (-0.2, 0.3)}),
    ("llquad_96_40", zonal_wavy, 1800.0, {}),
])
def test_quad_advection_matches_oracle(meshname, flow, dt, kw):
    dom, ref = _pair(meshname, dt, **kw)
    try:
        for d_ in (dom, ref):
            set_velocity(d_.blocklist[0], flow)
        b, r = dom.blocklist[0], ref.blocklist[0]
        nC, nE = b.dimensions["nCells"], b.dimensions["nEdges"]
        steps = 1 if flow is random_flow else 2
        diag = {}
        for _ in range(steps):
            _ir().seaice_run_advection(ref, diagnostics=diag)
            adv.seaice_run_advection(dom)
        rep = {}
        for n in TRACERS:
            x, y = b.structs["tracers"][n][..., :nC], r.structs["tracers"][n][..., :nC]
            assert np.isfinite(x).all(), n
            rep[n] = rel_linf(x, y)
        rir = r.structs["incremental_remap"]
        ta = dget(b, "triangleArea", 6, nE)
        assert np.array_equal(ta != 0.0, rir["triangleArea"][:, :nE] != 0.0)
        rep["triangleArea"] = rel_linf(ta, rir["triangleArea"][:, :nE])
        live = (rir["triangleArea"][:, :nE] != 0.0)[None, :, :]
        for nm, key in (("xTriangle", "xTriangleVertices"), ("yTriangle", "yTriangleVertices")):
            t = dget(b, nm, 18, nE).reshape((3, 6, nE), order="F")
            rep[nm] = rel_linf(np.where(live, t, 0.0), np.where(live, diag[key][:3, :, :nE], 0.0))
        ntri = (ta != 0.0).sum(axis=0)
        print(meshname, flow.__name__, "triangles per edge", np.bincount(ntri), {k: float(f"{v:.1e}") for k, v in rep.items()})
        assert max(rep.values()) <= TOL, rep
        if flow is random_flow:
            assert ntri.max() >= 4
        s0, s1 = dom.solver_state["ir_sums"]["iceVolumeCategory"]
        assert np.abs(s1 - s0).max() <= 1e-11 * np.abs(s0).max()
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_upwind_advection_refuses_quadrilateral_meshes():
    dom = quad_domain("quad_12_9", 600.0)
    vs.seaice_mesh_pool_create(dom)
    try:
        dom.configs["config_advection_type"] = "upwind"
        with pytest.raises(SeaiceHipError, match="vertexDegree"):
            adv.seaice_init_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
