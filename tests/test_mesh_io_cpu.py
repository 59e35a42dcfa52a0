"""MPAS grid / partition / restart files: round trips through the NetCDF reader and writer."""
import numpy as np

from mpas_seaice_amd import cases, mesh_io, parallel
from mpas_seaice_amd.partition import partition_mesh
from util import get_mesh


def test_mesh_file_round_trip(tmp_path):
    for name in ("QU480", "hex_8_6", "var12"):
        mesh = get_mesh(name)
        p = str(tmp_path / (name + ".grid.nc"))
        mesh_io.write_mpas_mesh(mesh, p)
        back = mesh_io.read_mpas_mesh(p)
        for k in ("nCells", "nVertices", "nEdges", "maxEdges", "vertexDegree", "on_a_sphere", "sphere_radius"):
            assert back[k] == mesh[k], (name, k)
        for k in ("nEdgesOnCell", "verticesOnCell", "edgesOnCell", "cellsOnCell", "cellsOnVertex", "cellsOnEdge",
                  "verticesOnEdge", "xCell", "zVertex", "areaCell", "areaTriangle", "dcEdge", "dvEdge", "fVertex",
                  "kiteAreasOnVertex", "coeffs_reconstruct"):
            assert np.array_equal(np.asarray(back[k]), np.asarray(mesh[k])), (name, k)
        assert back["verticesOnCell"].flags["F_CONTIGUOUS"] and back["verticesOnCell"].dtype == np.int32


def test_unit_sphere_file_is_rescaled(tmp_path):
    mesh = get_mesh("QU480")
    unit = dict(mesh)
    R = mesh["sphere_radius"]
    for k in ("xCell", "yCell", "zCell", "xVertex", "yVertex", "zVertex", "xEdge", "yEdge", "zEdge", "dcEdge", "dvEdge"):
        unit[k] = mesh[k] / R
    for k in ("areaCell", "areaTriangle", "kiteAreasOnVertex"):
        unit[k] = mesh[k] / R ** 2
    unit["sphere_radius"] = 1.0
    p = str(tmp_path / "unit.grid.nc")
    mesh_io.write_mpas_mesh(unit, p)
    back = mesh_io.read_mpas_mesh(p, earth_radius=R)
    assert back["sphere_radius"] == R
    assert np.allclose(back["areaCell"], mesh["areaCell"], rtol=1e-14)
    assert np.allclose(back["dvEdge"], mesh["dvEdge"], rtol=1e-14)


def test_graph_files_drive_the_partition(tmp_path):
    mesh = get_mesh("QU480")
    g = str(tmp_path / "graph.info")
    mesh_io.write_graph_info(mesh, g)
    head = open(g).readline().split()
    assert int(head[0]) == mesh.nCells and int(head[1]) == mesh.nEdges
    owner = (np.arange(mesh.nCells) * 3) // mesh.nCells
    part = str(tmp_path / "graph.info.part.3")
    np.savetxt(part, owner, fmt="%d")
    got = mesh_io.read_graph_partition(part, mesh.nCells)
    assert np.array_equal(got, owner)
    meshes, halos = partition_mesh(mesh, 3, num_halos=2, cell_owner=got)
    assert sum(h.nOwned["cell"] for h in halos) == mesh.nCells
    for r, h in enumerate(halos):
        assert np.all(owner[h.globalID["cell"][:h.nOwned["cell"]]] == r)


def test_restart_round_trip(tmp_path, oracle):
    mesh = get_mesh("QU480")
    dom = cases.make_domain(mesh, "QU480", configs={"config_elastic_subcycle_number": 10})
    oracle.seaice_init_velocity_solver(dom)
    oracle.seaice_run_velocity_solver(dom)
    p = str(tmp_path / "restart.nc")
    mesh_io.write_restart(dom, p)
    other = cases.make_domain(mesh, "QU480", configs={"config_elastic_subcycle_number": 10})
    for pool, name, _e, _o in mesh_io.RESTART_FIELDS:
        other.blocklist[0].structs[pool][name][...] = 0
    mesh_io.read_restart(other, p)
    b0, b1 = dom.blocklist[0], other.blocklist[0]
    for pool, name, ent, _o in mesh_io.RESTART_FIELDS + mesh_io._tracer_fields(b0):
        n = b0.dimensions[ent]
        assert np.array_equal(b0.structs[pool][name][..., :n], b1.structs[pool][name][..., :n]), name
    assert np.abs(b1.structs["velocity_variational"]["stress11"]).max() > 0


def test_seaice_init_mesh_rescales_and_sets_coriolis():
    """seaice_init_mesh on the pools: a unit-sphere mesh expanded to config_earth_radius equals the mesh built at
    that radius; fVertex = 2 omega sin(lat); a restart refuses a mesh of another radius."""
    from mpas_seaice_amd import cases, mesh as meshmod, mesh_init
    unit = meshmod.make_sphere_mesh(8, radius=1.0)
    full = meshmod.make_sphere_mesh(8, radius=6371229.0)
    dom = cases.make_domain(unit, None)
    ref = cases.make_domain(full, None)
    dom.blocklist[0].structs["mesh"]["fVertex"][:] = 0.0
    mesh_init.seaice_init_mesh(dom)
    m, r = dom.blocklist[0].structs["mesh"], ref.blocklist[0].structs["mesh"]
    assert m["sphere_radius"] == 6371229.0
    for k in ("xCell", "zVertex", "yEdge", "areaCell", "areaTriangle", "dvEdge", "dcEdge", "kiteAreasOnVertex"):
        assert np.allclose(m[k], r[k], rtol=1e-12, atol=1e-6 * np.abs(r[k]).max() * 1e-6), k
    nV = unit.nVertices
    assert np.allclose(m["fVertex"][:nV], 2.0 * 7.29212e-5 * np.sin(m["latVertex"][:nV]), rtol=0, atol=1e-18)
    dom2 = cases.make_domain(unit, None, configs={"config_do_restart": True})
    import pytest
    with pytest.raises(RuntimeError, match="inconsistent radii"):
        mesh_init.seaice_init_mesh(dom2)
