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


def test_files_are_checked_by_a_reader_and_writer_that_share_no_code_with_mesh_io(tmp_path):
    """mesh_io goes through scipy's NetCDF module both ways, so a round trip proves little about the bytes.  Here
    tests/cdf_raw.py -- written from the file format's layout alone -- reads what mesh_io wrote (entity dimension
    first, 0 for a missing neighbour, the leading pool index last in restart fields) and writes files mesh_io has
    never seen (a grid with other dimension order in the header, 0 and out-of-range neighbour marks, classic format;
    a restart with two records of which the first counts), which mesh_io must read into the pools' layout: Fortran
    order, n + 1 sentinels."""
    import cdf_raw
    mesh = get_mesh("hex_8_6")  # a planar mesh with a boundary: missing neighbours exist
    nC, nV, nE, M = mesh.nCells, mesh.nVertices, mesh.nEdges, int(mesh.maxEdges)
    p = str(tmp_path / "grid.nc")
    mesh_io.write_mpas_mesh(mesh, p)
    dims, gatts, var = cdf_raw.read(p)
    assert dims["nCells"] == nC and dims["nVertices"] == nV and dims["maxEdges"] == M and dims["TWO"] == 2
    assert gatts["on_a_sphere"].strip() == "NO"
    names, a = var["verticesOnCell"]
    assert names == ("nCells", "maxEdges") and a.shape == (nC, M) and a.dtype == np.int32
    pool = np.asarray(mesh["verticesOnCell"])  # (maxEdges, nCells), n + 1 = none
    assert np.array_equal(np.where(pool.T > nV, 0, pool.T), a)
    names, a = var["cellsOnVertex"]
    assert names == ("nVertices", "vertexDegree") and (a == 0).any() and a.max() <= nC
    assert np.array_equal(np.where(np.asarray(mesh["cellsOnVertex"]).T > nC, 0, np.asarray(mesh["cellsOnVertex"]).T), a)
    assert var["xCell"][0] == ("nCells",) and np.array_equal(var["xCell"][1], np.asarray(mesh["xCell"]))
    assert np.array_equal(var["dvEdge"][1], np.asarray(mesh["dvEdge"]))

    # ---- a grid file mesh_io did not write
    def conn(k, target):
        a = np.asarray(mesh[k]).T.astype(np.int32).copy()
        a[a > target] = 0
        return a
    voc = conn("verticesOnCell", nV)
    voc[3, int(mesh["nEdgesOnCell"][3]):] = nV + 7  # junk past nEdgesOnCell, as real grid files have
    variables = [("xCell", ("nCells",), np.asarray(mesh["xCell"], dtype=np.float64))]
    for k in ("yCell", "zCell", "areaCell"):
        variables.append((k, ("nCells",), np.asarray(mesh[k], dtype=np.float64)))
    for k in ("xVertex", "yVertex", "zVertex", "areaTriangle"):
        variables.append((k, ("nVertices",), np.asarray(mesh[k], dtype=np.float64)))
    for k in ("xEdge", "yEdge", "zEdge", "dcEdge", "dvEdge"):
        variables.append((k, ("nEdges",), np.asarray(mesh[k], dtype=np.float64)))
    variables += [("nEdgesOnCell", ("nCells",), np.asarray(mesh["nEdgesOnCell"], dtype=np.int32)),
                  ("verticesOnCell", ("nCells", "maxEdges"), voc),
                  ("edgesOnCell", ("nCells", "maxEdges"), conn("edgesOnCell", nE)),
                  ("cellsOnCell", ("nCells", "maxEdges"), conn("cellsOnCell", nC)),
                  ("cellsOnVertex", ("nVertices", "vertexDegree"), conn("cellsOnVertex", nC)),
                  ("cellsOnEdge", ("nEdges", "TWO"), conn("cellsOnEdge", nC)),
                  ("verticesOnEdge", ("nEdges", "TWO"), conn("verticesOnEdge", nV))]
    q = str(tmp_path / "foreign.grid.nc")
    cdf_raw.write(q, [("TWO", 2), ("vertexDegree", 3), ("maxEdges", M), ("nEdges", nE), ("nVertices", nV),
                      ("nCells", nC)], {"on_a_sphere": "NO", "sphere_radius": 1.0}, variables, version=1)
    back = mesh_io.read_mpas_mesh(q)
    for k in ("nEdgesOnCell", "verticesOnCell", "edgesOnCell", "cellsOnCell", "cellsOnVertex", "cellsOnEdge",
              "verticesOnEdge", "xCell", "zVertex", "areaCell", "areaTriangle", "dcEdge", "dvEdge", "edgesOnVertex"):
        assert np.array_equal(np.asarray(back[k]), np.asarray(mesh[k])), k
    assert back["verticesOnCell"].flags["F_CONTIGUOUS"] and back["verticesOnCell"].shape == (M, nC)
    assert not back["on_a_sphere"]

    # ---- restart files both ways
    dom = cases.make_domain(mesh, None, configs={"config_dt": 600.0})
    b = dom.blocklist[0]
    rng = np.random.Generator(np.random.PCG64(20241020))
    b.structs["velocity_solver"]["uVelocity"][:nV] = rng.standard_normal(nV)
    b.structs["velocity_variational"]["stress12"][:, :nC] = rng.standard_normal((M, nC))
    b.structs["tracers"]["iceEnthalpy"][..., :nC] = rng.standard_normal(b.structs["tracers"]["iceEnthalpy"][..., :nC].shape)
    b.structs["velocity_solver"]["solveVelocityPrevious"][:nV] = rng.integers(0, 2, nV)
    r = str(tmp_path / "restart.nc")
    mesh_io.write_restart(dom, r)
    dims, _, var = cdf_raw.read(r)
    assert dims["Time"] is None
    names, a = var["stress12"]
    assert names == ("Time", "nCells", "maxEdges") and a.shape == (1, nC, M)
    assert np.array_equal(a[0], b.structs["velocity_variational"]["stress12"][:, :nC].T)
    names, a = var["iceEnthalpy"]
    assert names == ("Time", "nCells", "nCategories", "nIceLayers")
    assert np.array_equal(a[0], np.transpose(b.structs["tracers"]["iceEnthalpy"][..., :nC]))
    assert var["solveVelocityPrevious"][1].dtype == np.int32
    # the same fields written by the raw writer (two records; the restart's is the first), read by mesh_io
    variables = []
    for k, (names, a) in var.items():
        two = np.concatenate([a, a * 0 - 3], axis=0)
        variables.append((k, names, two.astype(np.int32) if a.dtype.kind == "i" else two.astype(np.float64)))
    dl = [("Time", None)] + [(k, v) for k, v in dims.items() if k != "Time"]
    r2 = str(tmp_path / "foreign.restart.nc")
    cdf_raw.write(r2, dl, {}, variables, version=2)
    other = cases.make_domain(mesh, None, configs={"config_dt": 600.0})
    mesh_io.read_restart(other, r2)
    for pool, name, ent, _ in mesh_io.RESTART_FIELDS:
        n = nV if ent == "nVertices" else nC
        assert np.array_equal(other.blocklist[0].structs[pool][name][..., :n], b.structs[pool][name][..., :n]), name
    for t in ("iceAreaCategory", "iceEnthalpy", "snowEnthalpy"):
        assert np.array_equal(other.blocklist[0].structs["tracers"][t][..., :nC], b.structs["tracers"][t][..., :nC]), t
