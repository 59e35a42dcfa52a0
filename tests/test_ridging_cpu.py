"""The ridging oracle (oracle/ridge_oracle.c, a restatement of ridge_ice and its routines in
/root/reference/src/column/ice_mechred.F90; "parity unpinned": the reference holds no vectors for it) against what
pins it: total area 1 after ridging, conservation of ice volume, of ice energy and salt (volume tracers), of snow
volume up to the half that ridge_shift throws into the ocean, and closed forms for a single category."""
import numpy as np
import pytest

from mpas_seaice_amd import cases, pools
from oracle import ridge_flow
from util import get_mesh


def _domain(part, redist, conv=2.0e-7, shear=1.0e-7):
    mesh = get_mesh("geo8")
    cfg = {"config_ridging_participation_function": part, "config_ridging_redistribution_function": redist,
           "config_dt": 3600.0}
    dom = cases.make_domain(mesh, None, configs=cfg, open_water_band=(-0.2, 0.3))
    b = dom.blocklist[0]
    pools.allocate_ridging_pools(b)
    nC = mesh.nCells
    b.structs["velocity_solver"]["dynamicsTimeStep"][0] = 3600.0
    a = b.structs["tracers"]["iceAreaCategory"][0, :, :nC]
    b.structs["icestate"]["openWaterArea"][:nC] = 1.0 - a.sum(axis=0)
    lat = np.asarray(mesh["latCell"])
    b.structs["ridging"]["ridgeConvergence"][:nC] = conv * np.maximum(np.sin(3.0 * lat), 0.0)
    b.structs["ridging"]["ridgeShear"][:nC] = shear * np.cos(lat) ** 2
    return mesh, dom, b


@pytest.mark.parametrize("part,redist", [("exponential", "exponential"), ("Thorndike75", "Hibler80"),
                                         ("Thorndike75", "exponential")])
def test_ridging_conserves_and_closes_the_area(part, redist):
    mesh, dom, b = _domain(part, redist)
    nC = mesh.nCells
    tr = b.structs["tracers"]
    vi = tr["iceVolumeCategory"][0, :, :nC].copy()
    vsn = tr["snowVolumeCategory"][0, :, :nC].copy()
    nI = b.dimensions["nIceLayers"]
    ei0 = (tr["iceEnthalpy"][:, :, :nC] * vi[None] / nI).sum(axis=(0, 1))
    si0 = (tr["iceSalinity"][:, :, :nC] * vi[None] / nI).sum(axis=(0, 1))
    a_before = tr["iceAreaCategory"][0, :, :nC].copy()
    ridge_flow.column_ridging(dom)
    a, v, s = (tr[k][0, :, :nC] for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory"))
    aice0 = b.structs["icestate"]["openWaterArea"][:nC]
    assert np.abs(a.sum(axis=0) + aice0 - 1.0).max() < 1e-11          # asum = 1 (ridge_ice :387-398)
    assert a.min() >= 0.0 and aice0.min() >= 0.0
    assert np.abs(v.sum(axis=0) - vi.sum(axis=0)).max() <= 1e-13 * vi.sum(axis=0).max()
    ei1 = (tr["iceEnthalpy"][:, :, :nC] * v[None] / nI).sum(axis=(0, 1))
    si1 = (tr["iceSalinity"][:, :, :nC] * v[None] / nI).sum(axis=(0, 1))
    assert np.abs(ei1 - ei0).max() <= 1e-12 * np.abs(ei0).max()
    assert np.abs(si1 - si0).max() <= 1e-12 * np.abs(si0).max()
    # half of the ridged snow goes to the ocean (fsnowrdg = 0.5): as fresh water over the thermodynamic step
    fresh = b.structs["ocean_fluxes"]["oceanFreshWaterFlux"][:nC]
    lost = vsn.sum(axis=0) - s.sum(axis=0)
    assert np.abs(lost - fresh * 3600.0 / 330.0).max() <= 1e-12 * vsn.sum(axis=0).max()
    rg = b.structs["ridging"]
    assert rg["areaLossRidge"][:nC].max() > 0.0 and (rg["areaGainRidge"][:nC] <= rg["areaLossRidge"][:nC] + 1e-20).all()
    assert np.abs(a - a_before).max() > 1e-6
    assert dom.solver_state["ridge_iterations"].max() <= 3


def test_no_forcing_and_closed_area_changes_nothing():
    """asum = 1 and no convergence or shear: closing and opening rates are zero, every category keeps its state."""
    mesh, dom, b = _domain("exponential", "exponential", conv=0.0, shear=0.0)
    nC = mesh.nCells
    before = {k: v.copy() for k, v in b.structs["tracers"].items()}
    ridge_flow.column_ridging(dom)
    ice = before["iceAreaCategory"][0, :, :nC] > 1e-11  # an empty category gets the freezing temperature and zeros
    for k, v in b.structs["tracers"].items():
        diff = np.abs(v[..., :nC] - before[k][..., :nC]) * ice[None]
        assert diff.max() <= 1e-12 * max(np.abs(before[k]).max(), 1.0), k
    empty = ~ice
    assert np.all(b.structs["tracers"]["surfaceTemperature"][0, :, :nC][empty] == -1.8)  # colpkg_compute_tracers :226
    # rounding of the area sum gives closing rates of 1e-16/dt (ridge_ice :253-259): nothing above that
    assert np.abs(b.structs["ridging"]["areaLossRidge"]).max() < 1e-18


def test_single_category_closed_form():
    """One category of thickness h = 1 m covering the whole cell, Hibler80 redistribution, pure convergence: the ridge
    is uniform between hrmin = min(2h, h + 1) = 2 and hrmax = 2 sqrt(25 h) = 10, so krdg = 6 and a unit of ridging
    ice becomes 1/6 of ridge: with all of the participation on the one category, the net closing rate c removes
    c dt of area, i.e. the category loses ardg1 = c dt / (1 - 1/6) and gains ardg1 / 6."""
    mesh = get_mesh("geo4")
    dom = cases.make_domain(mesh, None, nCategories=1, configs={
        "config_ridging_participation_function": "Thorndike75", "config_ridging_redistribution_function": "Hibler80"})
    b = dom.blocklist[0]
    pools.allocate_ridging_pools(b)
    nC = mesh.nCells
    tr = b.structs["tracers"]
    tr["iceAreaCategory"][0, 0, :nC] = 1.0
    tr["iceVolumeCategory"][0, 0, :nC] = 1.0
    tr["snowVolumeCategory"][0, 0, :nC] = 0.1
    b.structs["icestate"]["openWaterArea"][:nC] = 0.0
    b.structs["velocity_solver"]["dynamicsTimeStep"][0] = 1000.0
    c = 1.0e-5
    b.structs["ridging"]["ridgeConvergence"][:nC] = c
    b.structs["initial"]["categoryThicknessLimits"][:] = [0.0, 100.0]
    ridge_flow.column_ridging(dom)
    rg = b.structs["ridging"]
    assert np.allclose(rg["ratioRidgeThicknessToIce"][0, :nC], 6.0, rtol=1e-14)
    ardg1 = c * 1000.0 / (1.0 - 1.0 / 6.0)
    assert np.allclose(rg["areaLossRidge"][:nC] * 1000.0, ardg1, rtol=1e-12)
    assert np.allclose(rg["areaGainRidge"][:nC] * 1000.0, ardg1 / 6.0, rtol=1e-12)
    assert np.allclose(tr["iceAreaCategory"][0, 0, :nC], 1.0 - c * 1000.0, rtol=1e-12)
    assert np.allclose(tr["iceVolumeCategory"][0, 0, :nC], 1.0, rtol=1e-13)
    # closing with no open water left: the reference's second iteration finds asum = 1 - c dt < 1 and opens the rest
    assert np.allclose(b.structs["icestate"]["openWaterArea"][:nC], c * 1000.0, rtol=1e-9)
