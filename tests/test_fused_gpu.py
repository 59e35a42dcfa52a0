"""The fused subcycle kernel (one launch per iteration: the vertices of iteration t - 1, then the cells of iteration
t, over tiles of 64 cells) against the two-kernel form: the same bits."""
import numpy as np
import pytest

from mpas_seaice_amd import cases, lib as sihlib, parallel, velocity_solver as vs
from test_persistent_gpu import _prepared, _state
from util import get_mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("meshname,cfg,kw", [
    ("QU240", {}, {}),
    ("QU240", {"config_constitutive_relation_type": "evp_revised"}, {"open_water_band": (-0.2, 0.3)}),
    ("QU240", {"config_constitutive_relation_type": "linear"}, {}),
    ("var16", {"config_dt": 1800.0}, {}),
    ("hex_24_20", {"config_dt": 600.0}, {}),
    ("QU120", {}, {"open_water_band": (-0.2, 0.3)}),
])
def test_fused_kernel_gives_the_bits_of_the_two_kernel_form(meshname, cfg, kw):
    mesh = get_mesh(meshname)
    name = meshname if meshname.startswith("QU") else None
    print("case", meshname, cfg, kw, flush=True)
    res = []
    for fused in (False, True):
        dom, b = _prepared(mesh, name, configs=cfg, **kw)
        try:
            if fused:
                assert b.ctx.info("fusedKernel") == 1.0
            # loops of odd and even length (the velocities alternate between two arrays), a single iteration, and
            # iterations of the two-kernel form in between
            for n in (7, 1, 2, 12, 30):
                if fused:
                    b.ctx.subcycle_fused(n)
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


def test_fused_kernel_on_a_block_with_deep_halos():
    """One block of a 4-way partition with 4 halo layers: the halo vertices that see all their cells are updated
    too (computeHaloVertices), by the tiles that own them."""
    mesh = get_mesh("QU240")
    cfg = {"config_num_halos": 4, "config_halo_exchange_interval": 4}
    res = []
    for fused in (False, True):
        dom, halo, _ = parallel.make_partitioned_domain(mesh, "QU240", 4, configs=cfg, only_block=1)
        vs.seaice_mesh_pool_create(dom)
        try:
            vs.seaice_init_velocity_solver(dom)
            b = dom.blocklist[0]
            vs._upload(dom, b)
            b.ctx.pre_subcycle(-1)
            for n in (4, 3):
                if fused:
                    b.ctx.subcycle_fused(n)
                else:
                    b.ctx.subcycle(n, last=False)
            b.ctx.sync()
            res.append(_state(b))
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_whole_steps_choose_the_fused_kernel_for_a_block_too_large_to_be_resident(monkeypatch):
    """QU120 on one GPU (40 962 cells: 641 tiles, more than stay resident): sih_evp_run takes the fused kernel for
    all iterations but the last by itself, and the step gives the bits of the two-kernel form."""
    monkeypatch.setenv("SIH_PERSIST_SMALL_TILES", "0")  # (with tiles of 42 cells QU120 just fits the persistent kernel)
    mesh = get_mesh("QU120")
    res = []
    for mode in (0, -1):
        dom = cases.make_domain(mesh, "QU120", configs={"config_elastic_subcycle_number": 24}, open_water_band=(-0.2, 0.3))
        vs.seaice_mesh_pool_create(dom)
        try:
            vs.seaice_init_velocity_solver(dom)
            b = dom.blocklist[0]
            b.ctx.set_persistent(mode)
            assert b.ctx.info("persistentKernel") == 0.0 and b.ctx.info("fusedKernel") == 1.0
            vs.seaice_run_velocity_solver(dom)
            assert (b.ctx.info("fusedLaunches") > 0) == (mode == -1)
            sv, vv = b.structs["velocity_solver"], b.structs["velocity_variational"]
            res.append({"u": sv["uVelocity"].copy(), "v": sv["vVelocity"].copy(), "s11": vv["stress11"].copy(),
                        "s12": vv["stress12"].copy(), "div": sv["divergence"].copy(), "shear": sv["shear"].copy()})
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    assert np.abs(res[0]["u"]).max() > 1e-4
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k


def test_a_block_with_an_option_outside_the_fused_kernel_is_refused():
    mesh = get_mesh("QU240")
    dom, b = _prepared(mesh, "QU240", configs={"config_strain_scheme": "weak", "config_stress_divergence_scheme": "weak"})
    try:
        assert b.ctx.info("fusedKernel") == 0.0
        with pytest.raises(sihlib.SeaiceHipError, match="does not qualify"):
            b.ctx.subcycle_fused(3)
    finally:
        vs.seaice_mesh_pool_destroy(dom)


def test_cells_without_the_marks_of_the_set_up_in_every_kernel_form(oracle, monkeypatch):
    """The subcycle kernels leave out a corner's own gradient and the first corner's rows of basisIntegralsU / V only in
    cells the set-up has marked (where the partition of unity holds to rounding); no mesh of these tests has an
    unmarked cell, so SIH_TEST_UNMARK_EVERY withholds the marks from every third cell: marked and unmarked cells then
    sit side by side in every wave.  The four kernel forms must still give the same bits, and the oracle's velocities
    to 1e-10."""
    monkeypatch.setenv("SIH_TEST_UNMARK_EVERY", "3")
    mesh = get_mesh("QU240")
    res = {}
    for form in ("split", "cell", "fused", "persistent"):
        if form == "cell":
            monkeypatch.setenv("SIH_CELL_KERNEL", "fused")   # one thread per cell
        else:
            monkeypatch.delenv("SIH_CELL_KERNEL", raising=False)
        dom, b = _prepared(mesh, "QU240")
        try:
            assert b.ctx.info("derivedGradient") == 1.0 and b.ctx.info("derivedIntegralRow") == 1.0
            n_marks, _ = b.ctx.field_count("_nEdgesDerive")
            marks = np.zeros(n_marks, dtype=np.int32)
            b.ctx.get("_nEdgesDerive", marks)
            marks = marks[:mesh.nCells] & 768
            assert (marks == 0).sum() == (mesh.nCells + 2) // 3 and (marks == 768).sum() == mesh.nCells - (mesh.nCells + 2) // 3
            for n in (7, 1, 12):
                if form == "fused":
                    b.ctx.subcycle_fused(n)
                elif form == "persistent":
                    b.ctx.subcycle_persistent(n)
                else:
                    b.ctx.subcycle(n, last=False)
                b.ctx.subcycle(2, last=False)
            b.ctx.sync()
            res[form] = _state(b)
        finally:
            vs.seaice_mesh_pool_destroy(dom)
    assert np.abs(res["split"]["uVelocity"]).max() > 1e-4
    for form in ("cell", "fused", "persistent"):
        for k in res["split"]:
            assert np.array_equal(res["split"][k], res[form][k]), (form, k)
    monkeypatch.delenv("SIH_CELL_KERNEL", raising=False)
    small = get_mesh("QU480")
    cfg = {"config_elastic_subcycle_number": 30}
    ref = cases.make_domain(small, "QU480", configs=cfg)
    oracle.seaice_init_velocity_solver(ref)
    oracle.seaice_run_velocity_solver(ref)
    dev = cases.make_domain(small, "QU480", configs=cfg)
    vs.seaice_mesh_pool_create(dev)
    try:
        vs.seaice_init_velocity_solver(dev)
        vs.seaice_run_velocity_solver(dev)
    finally:
        vs.seaice_mesh_pool_destroy(dev)
    for k in ("uVelocity", "vVelocity"):
        a = dev.blocklist[0].structs["velocity_solver"][k]
        r = ref.blocklist[0].structs["velocity_solver"][k]
        assert np.abs(a - r).max() <= 1e-10 * np.abs(r).max(), k
