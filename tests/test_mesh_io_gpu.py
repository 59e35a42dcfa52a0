"""The device path driven from files: an MPAS grid file and a METIS partition file, as a host model reads them
(x1.N.grid.nc, graph.info.part.N), and a restart written after the step."""
import numpy as np
import pytest

from mpas_seaice_amd import advection as adv, cases, mesh_io, parallel, velocity_solver as vs
from util import get_mesh

pytestmark = pytest.mark.gpu

TRACERS = ["iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature", "iceEnthalpy",
           "iceSalinity", "snowEnthalpy"]
STATE = [("velocity_solver", "uVelocity"), ("velocity_solver", "vVelocity"), ("velocity_variational", "stress11"),
         ("velocity_variational", "stress22"), ("velocity_variational", "stress12"), ("velocity_solver", "shear")]


def _run(mesh, nblocks, cell_owner):
    cfg = {"config_elastic_subcycle_number": 60}
    dom, halos, _ = parallel.make_partitioned_domain(mesh, "QU240", nblocks, configs=cfg, cell_owner=cell_owner)
    vs.seaice_mesh_pool_create(dom, device=0)
    try:
        dom.comm = parallel.LocalComm(dom, halos, nblocks)
        vs.seaice_init_velocity_solver(dom)
        adv.seaice_init_advection(dom)
        vs.seaice_run_velocity_solver(dom)
        adv.seaice_run_advection(dom)
    finally:
        vs.seaice_mesh_pool_destroy(dom)
    out = {}
    for pool, name in STATE + [("tracers", t) for t in TRACERS]:
        loc = parallel.field_loc(name)
        n = mesh.nVertices if loc == "vertex" else mesh.nCells
        out[name] = parallel.gather_global(dom.blocklist, halos, pool, name, n, loc)
    return dom, out


def test_velocity_solver_and_advection_on_a_mesh_and_partition_read_from_files(tmp_path):
    mesh = get_mesh("QU240")
    grid = str(tmp_path / "x1.10242.grid.nc")
    mesh_io.write_mpas_mesh(mesh, grid)
    mesh_io.write_graph_info(mesh, str(tmp_path / "graph.info"))
    # a partition as METIS would write it: one block number per line, here two bands of latitude bent by longitude
    owner = ((np.asarray(mesh["zCell"]) + 0.3 * np.asarray(mesh["xCell"])) > 0.0).astype(np.int64)
    part = str(tmp_path / "graph.info.part.2")
    np.savetxt(part, owner, fmt="%d")

    loaded = mesh_io.read_mpas_mesh(grid)
    got_owner = mesh_io.read_graph_partition(part, loaded.nCells)
    assert np.array_equal(got_owner, owner)
    dom_f, from_files = _run(loaded, 2, got_owner)
    _, in_memory = _run(mesh, 2, owner)
    for k, a in in_memory.items():
        assert np.isfinite(a).all(), k
        assert np.array_equal(a, from_files[k]), k
    assert np.abs(in_memory["uVelocity"]).max() > 1e-3

    # the state the reference keeps in its restart stream survives the trip through a restart file
    one = cases.make_domain(loaded, "QU240", configs={"config_elastic_subcycle_number": 60})
    b = one.blocklist[0]
    for pool, name in STATE[:5]:
        n = from_files[name].shape[-1]
        b.structs[pool][name][..., :n] = from_files[name]
    rst = str(tmp_path / "restart.nc")
    mesh_io.write_restart(one, rst)
    other = cases.make_domain(loaded, "QU240")
    mesh_io.read_restart(other, rst)
    for pool, name in STATE[:5]:
        n = from_files[name].shape[-1]
        assert np.array_equal(other.blocklist[0].structs[pool][name][..., :n], from_files[name]), name
