"""The persistent subcycle kernel (one launch for all iterations of a small block; tiles of 64 cells resident on
the chip, neighbour-to-neighbour hand-offs) against the two-kernel form: the same bits."""
import numpy as np
import pytest

from mpas_seaice_amd import cases, parallel, velocity_solver as vs
from util import get_mesh

pytestmark = pytest.mark.gpu

FIELDS = [("uVelocity", "vertex", 1), ("vVelocity", "vertex", 1), ("stress11", "cell", 0), ("stress22", "cell", 0),
          ("stress12", "cell", 0)]


def _state(b):
    d = b.dimensions
    out = {}
    for name, loc, _ in FIELDS:
        n = d["nVertices"] if loc == "vertex" else d["nCells"]
        a = np.zeros(n + 1) if loc == "vertex" else np.zeros((d["maxEdges"], n + 1), order="F")
        b.ctx.get(name, a)
        out[name] = a
    return out


def _prepared(mesh, name, configs=None, **kw):
    dom = cases.make_domain(mesh, name, configs=configs, **kw)
    vs.seaice_mesh_pool_create(dom)
    vs.seaice_init_velocity_solver(dom)
    b = dom.blocklist[0]
    vs._upload(dom, b)
    b.ctx.pre_subcycle(-1)
    return dom, b


@pytest.mark.parametrize("meshname,cfg,kw", [
    ("QU240", {}, {}),
    ("QU240", {"config_constitutive_relation_type": "evp_revised"}, {"open_water_band": (-0.2, 0.3)}),
    ("var16", {"config_dt": 1800.0}, {}),
    ("hex_24_20", {"config_dt": 600.0}, {}),
])
def test_persistent_kernel_gives_the_bits_of_the_two_kernel_form(meshname, cfg, kw):
    mesh = get_mesh(meshname)
    name = meshname if meshname.startswith("QU") else None
    print("case", meshname, cfg, kw, flush=True)
    res = []
    for persistent in (False, True):
        dom, b = _prepared(mesh, name, configs=cfg, **kw)
        try:
            if persistent:
                assert b.ctx.info("persistentKernel") == 1.0
            # several launches in a row, with iterations of the two-kernel form in between
            for n in (7, 1, 13, 30):
                if persistent:
                    b.ctx.subcycle_persistent(n)
                else:
                    b.ctx.subcycle(n, last=False)
                b.ctx.subcycle(2, last=False)
            b.ctx.sync()
            res.append(_state(b))
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    assert np.abs(res[0]["uVelocity"]).max() > 1e-4
    for k in res[0]:
        assert np.isfinite(res[1][k]).all(), k
        assert np.array_equal(res[0][k], res[1][k]), (meshname, k, np.abs(res[0][k] - res[1][k]).max())


def test_persistent_kernel_on_a_block_with_deep_halos():
    """One block of a 4-way partition with 4 halo layers: the halo vertices that see all their cells are updated
    too (computeHaloVertices), by the tiles that own them."""
    mesh = get_mesh("QU240")
    cfg = {"config_num_halos": 4, "config_halo_exchange_interval": 4}
    res = []
    for persistent in (False, True):
        dom, halo, _ = parallel.make_partitioned_domain(mesh, "QU240", 4, configs=cfg, only_block=1)
        vs.seaice_mesh_pool_create(dom)
        try:
            vs.seaice_init_velocity_solver(dom)
            b = dom.blocklist[0]
            vs._upload(dom, b)
            b.ctx.pre_subcycle(-1)
            if persistent:
                assert b.ctx.info("persistentKernel") == 1.0
                b.ctx.subcycle_persistent(4)
            else:
                b.ctx.subcycle(4, last=False)
            b.ctx.sync()
            res.append(_state(b))
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_a_block_too_large_to_be_resident_keeps_the_two_kernel_form():
    from mpas_seaice_amd import lib as sihlib
    mesh = get_mesh("QU60")
    dom, b = _prepared(mesh, "QU60")
    try:
        assert b.ctx.info("persistentKernel") == 0.0
        with pytest.raises(sihlib.SeaiceHipError, match="does not qualify"):
            b.ctx.subcycle_persistent(3)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_small_tiles_hold_a_block_that_tiles_of_64_cells_cannot():
    """QU120 (40 962 cells): 641 tiles of 64 cells are more than stay resident; 976 tiles of 42 cells, with the rows of
    the metric integrals read from memory in every iteration instead of kept in LDS, fit four to a CU.  Same bits as
    the two-kernel form, odd and even launches, interleaved with two-kernel iterations."""
    mesh = get_mesh("QU120")
    res = []
    for persistent in (False, True):
        dom, b = _prepared(mesh, "QU120", open_water_band=(-0.2, 0.3))
        try:
            if persistent:
                assert b.ctx.info("persistentKernel") == 1.0, b.ctx.info("persistentReason")
                assert b.ctx.info("persistentTileCells") == 42.0 and b.ctx.info("persistentTiles") == 976.0
            for n in (7, 1, 12):
                if persistent:
                    b.ctx.subcycle_persistent(n)
                else:
                    b.ctx.subcycle(n, last=False)
                b.ctx.subcycle(2, last=False)
            b.ctx.sync()
            res.append(_state(b))
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    assert np.abs(res[0]["uVelocity"]).max() > 1e-4
    for k in res[0]:
        assert np.isfinite(res[1][k]).all(), k
        assert np.array_equal(res[0][k], res[1][k]), (k, np.abs(res[0][k] - res[1][k]).max())


def test_small_tiles_for_seven_sided_meshes():
    """A variable-resolution mesh with 5-, 6- and 7-sided cells (maxEdges 7), 31 362 cells: tiles of 64 cells need
    100 KB of LDS, one to a CU, and 491 of them are more than a device holds; tiles of 32 cells x 7 corners (four waves)
    with the metric rows streamed fit four to a CU.  Same bits as the two-kernel form."""
    mesh = get_mesh("var56")
    assert int(mesh.maxEdges) == 7
    res = []
    for persistent in (False, True):
        dom, b = _prepared(mesh, None, configs={"config_dt": 900.0}, open_water_band=(-0.2, 0.3))
        try:
            if persistent:
                assert b.ctx.info("persistentKernel") == 1.0, b.ctx.info("persistentReason")
                assert b.ctx.info("persistentTileCells") == 32.0 and b.ctx.info("persistentTiles") == 981.0
            for n in (7, 1, 12):
                if persistent:
                    b.ctx.subcycle_persistent(n)
                else:
                    b.ctx.subcycle(n, last=False)
                b.ctx.subcycle(2, last=False)
            b.ctx.sync()
            res.append(_state(b))
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    assert np.abs(res[0]["uVelocity"]).max() > 1e-4
    for k in res[0]:
        assert np.isfinite(res[1][k]).all(), k
        assert np.array_equal(res[0][k], res[1][k]), (k, np.abs(res[0][k] - res[1][k]).max())
