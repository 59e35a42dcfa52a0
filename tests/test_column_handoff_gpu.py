"""The hand-off to the column package's ridging (sih_column_get / sih_column_put): after the velocity solver and
the advection have run on the device, the per-cell records the reference assembles on the host
(set_cice_physics_tracer_array_category, mpas_seaice_column.F:6534-6575; the arguments of colpkg_step_ridge,
:3480-3530) come from the device for the owned cells only, and what the ridging leaves goes back the same way."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, lib as sihlib, parallel, velocity_solver as vs
from util import get_mesh

pytestmark = pytest.mark.gpu

# the reference's order in trcrn: surfaceTemperature, iceEnthalpy, snowEnthalpy, iceSalinity
TRCRN = ["surfaceTemperature", "iceEnthalpy", "snowEnthalpy", "iceSalinity"]
CATEGORY = ["iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"]


def _field(b, name):
    n, _ = b.ctx.field_count(name)
    nC = b.dimensions["nCells"]
    a = np.zeros((n // nC, nC), order="F")
    b.ctx.get(name, a)
    return a


def _run_dynamics(dom):
    vs.seaice_mesh_pool_create(dom)
    vs.seaice_init_velocity_solver(dom)
    adv.seaice_init_advection(dom)
    vs.seaice_run_velocity_solver(dom)
    adv.seaice_run_advection(dom, upload=True, download=False)


def _column_case(dom, b, ridging_outputs=True):
    nC, nOwn = b.dimensions["nCells"], b.dimensions["nCellsSolve"]
    nCat = b.dimensions["nCategories"]
    # ---- trcrn(nTracers, nCategories) of every owned cell
    fields = {n: _field(b, n) for n in TRCRN + CATEGORY + ["ridgeConvergence", "ridgeShear"]}
    layers = [fields[n].shape[0] // nCat for n in TRCRN]
    S = sum(layers)
    rec = np.zeros((S, nCat, nOwn), order="F")
    b.ctx.column_get(TRCRN, nCat, 0, nOwn, rec)
    s0 = 0
    for n, L in zip(TRCRN, layers):
        want = fields[n].reshape((L, nCat, nC), order="F")[:, :, :nOwn]  # the host array (layer, category, cell)
        assert np.array_equal(rec[s0:s0 + L], want), n
        s0 += L
    assert np.abs(rec).max() > 0
    # ---- aicen, vicen, vsnon and the two cell fields
    cat = np.zeros((len(CATEGORY), nCat, nOwn), order="F")
    b.ctx.column_get(CATEGORY, nCat, 0, nOwn, cat)
    for k, n in enumerate(CATEGORY):
        assert np.array_equal(cat[k], fields[n][:, :nOwn]), n
    rid = np.zeros((2, 1, nOwn), order="F")
    b.ctx.column_get(["ridgeConvergence", "ridgeShear"], 1, 0, nOwn, rid)
    assert np.array_equal(rid[0, 0], fields["ridgeConvergence"][0, :nOwn])
    assert np.array_equal(rid[1, 0], fields["ridgeShear"][0, :nOwn])
    assert np.abs(rid).max() > 0 or not ridging_outputs
    # ---- a window of cells, then the way back: only those cells change, halo cells included in "the others"
    first, n = nOwn // 3, nOwn // 2
    win = np.zeros((S, nCat, n), order="F")
    b.ctx.column_get(TRCRN, nCat, first, n, win)
    assert np.array_equal(win, rec[:, :, first:first + n])
    b.ctx.column_put(TRCRN, nCat, first, n, np.asfortranarray(win * 0.5 + 1.0))
    s0 = 0
    for name, L in zip(TRCRN, layers):
        now = _field(b, name).reshape((L, nCat, nC), order="F")
        want = fields[name].reshape((L, nCat, nC), order="F").copy()
        want[:, :, first:first + n] = want[:, :, first:first + n] * 0.5 + 1.0
        assert np.array_equal(now, want), name
        s0 += L
    # ---- argument checks
    with pytest.raises(sihlib.SeaiceHipError, match="count"):
        b.ctx.column_get(TRCRN, nCat, 0, nOwn, np.zeros((S, nCat, nOwn + 1), order="F"))
    with pytest.raises(sihlib.SeaiceHipError, match="multiple"):
        b.ctx.column_get(["ridgeShear"], nCat, 0, nOwn, np.zeros((1, nCat, nOwn), order="F"))
    with pytest.raises(sihlib.SeaiceHipError, match="unknown"):
        b.ctx.column_get(["noSuchField"], 1, 0, nOwn, np.zeros((1, 1, nOwn), order="F"))


def test_ridging_records_of_a_single_block():
    cfg = {"config_use_column_package": True, "config_use_column_vertical_thermodynamics": True,
           "config_elastic_subcycle_number": 20}
    dom = cases.make_domain(get_mesh("QU240"), "QU240", configs=cfg)
    b = dom.blocklist[0]
    area = b.structs["tracers"]["iceAreaCategory"][0].sum(axis=0)
    b.structs["icestate"]["openWaterArea"][:] = 1.0 - area
    b.structs["icestate"]["iceAreaCellInitial"][:] = area
    try:
        _run_dynamics(dom)
        _column_case(dom, b)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_ridging_records_of_a_block_with_halo_cells():
    """one block of a 4-way partition: the owned cells are the first nCellsSolve of the block's host numbering, the
    library's own numbering is another one (renumbered for locality)"""
    cfg = {"config_elastic_subcycle_number": 20}
    dom, halo, _ = parallel.make_partitioned_domain(get_mesh("QU240"), "QU240", 4, configs=cfg, only_block=2)
    b = dom.blocklist[0]
    assert b.dimensions["nCellsSolve"] < b.dimensions["nCells"]
    try:
        _run_dynamics(dom)
        _column_case(dom, b, False)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
