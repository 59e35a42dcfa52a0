"""MPAS mesh, partition and restart files for the host side of the device library.

The reference gets its mesh pool from the MPAS framework's stream reader (``<grid>.nc`` named by the
``input`` / ``restart`` streams of /root/reference/src/Registry.xml:1863-2060; mesh fields
Registry.xml:2251-2367) and its block decomposition from ``graph.info.part.<N>`` written by METIS
(``config_block_decomp_file_prefix``).  This module reads and writes those formats so that a run can
start from the same files:

* ``read_mpas_mesh`` / ``write_mpas_mesh``: NetCDF classic or 64-bit-offset files (what
  ``scipy.io.netcdf_file`` understands; HDF5-based NetCDF-4 and CDF-5 files must be converted with
  ``nccopy -k classic`` / ``-k 64-bit-offset`` first).  File variables are C-ordered with the entity
  dimension first, e.g. ``verticesOnCell(nCells, maxEdges)``, and mark a missing neighbour with 0; the
  mesh pool in memory is Fortran-ordered ``(maxEdges, nCells)`` with ``n+1`` for a missing neighbour,
  which is what the framework hands to the reference code (mpas_seaice_mesh.F:471).
* ``read_graph_partition``: one block number per line, cells in global order.
* ``write_graph_info``: the METIS input the partition is made from (cell adjacency).
* ``write_restart`` / ``read_restart``: the restart stream's dynamics fields (Registry.xml:1910-2000:
  uVelocity, vVelocity, stress11/22/12, solveVelocityPrevious, oceanStressCellU/V) and the advected
  tracers, with the framework's dimension order (Time, entity, ...).
"""
from __future__ import annotations

import numpy as np
from scipy.io import netcdf_file

from .mesh import Mesh, add_reconstruction_coefficients, edges_on_vertex

OMEGA = 7.29212e-5  # omega, mpas_seaice_constants.F:54 (used for fCell / fVertex / fEdge when the file has none)

_CONN = {  # name: (entity dimension, second dimension, sentinel dimension)
    "verticesOnCell": ("nCells", "maxEdges", "nVertices"),
    "edgesOnCell": ("nCells", "maxEdges", "nEdges"),
    "cellsOnCell": ("nCells", "maxEdges", "nCells"),
    "cellsOnVertex": ("nVertices", "vertexDegree", "nCells"),
    "edgesOnVertex": ("nVertices", "vertexDegree", "nEdges"),
    "cellsOnEdge": ("nEdges", "TWO", "nCells"),
    "verticesOnEdge": ("nEdges", "TWO", "nVertices"),
}
_REAL1 = {
    "nCells": ("xCell", "yCell", "zCell", "latCell", "lonCell", "areaCell", "fCell"),
    "nVertices": ("xVertex", "yVertex", "zVertex", "latVertex", "lonVertex", "areaTriangle", "fVertex"),
    "nEdges": ("xEdge", "yEdge", "zEdge", "latEdge", "lonEdge", "dcEdge", "dvEdge", "fEdge"),
}


def _attr(f, name, default=None):
    v = getattr(f, name, default)
    if isinstance(v, bytes):
        v = v.decode()
    return v


def read_mpas_mesh(path, earth_radius=None, reconstruct=True) -> Mesh:
    """Mesh pool of one unpartitioned domain from an MPAS grid / restart file.

    ``earth_radius``: rescale a unit-sphere mesh as seaice_init's rescale_mesh does
    (config_earth_radius); None keeps the file's lengths.
    """
    f = netcdf_file(path, "r", mmap=False)
    try:
        dims = f.dimensions
        m = Mesh()
        nC, nV, nE = int(dims["nCells"]), int(dims["nVertices"]), int(dims["nEdges"])
        M, D = int(dims["maxEdges"]), int(dims["vertexDegree"])
        m["nCells"], m["nVertices"], m["nEdges"] = nC, nV, nE
        m["nCellsSolve"], m["nVerticesSolve"], m["nEdgesSolve"] = nC, nV, nE
        m["maxEdges"], m["vertexDegree"] = M, D
        sphere = str(_attr(f, "on_a_sphere", "YES")).strip().upper() in ("YES", "T", "TRUE", "1")
        m["on_a_sphere"] = sphere
        m["sphere_radius"] = float(_attr(f, "sphere_radius", 1.0))
        size = {"nCells": nC, "nVertices": nV, "nEdges": nE}
        m["nEdgesOnCell"] = np.array(f.variables["nEdgesOnCell"][:], dtype=np.int32)
        for name, (ent, _second, target) in _CONN.items():
            if name not in f.variables:
                continue
            a = np.array(f.variables[name][:], dtype=np.int32)  # (entity, k), 0 = none
            a = np.where((a < 1) | (a > size[target]), size[target] + 1, a)
            m[name] = np.asfortranarray(a.T)
        # entries beyond nEdgesOnCell are undefined in grid files: give them the sentinel
        slot = np.arange(M)[:, None] >= m["nEdgesOnCell"][None, :]
        for name, target in (("verticesOnCell", nV), ("edgesOnCell", nE), ("cellsOnCell", nC)):
            if name in m:
                m[name][slot] = target + 1
        for ent, names in _REAL1.items():
            for name in names:
                if name in f.variables:
                    m[name] = np.array(f.variables[name][:], dtype=np.float64)
        if "kiteAreasOnVertex" in f.variables:
            m["kiteAreasOnVertex"] = np.asfortranarray(np.array(f.variables["kiteAreasOnVertex"][:], dtype=np.float64).T)
        for name in ("indexToCellID", "indexToVertexID", "indexToEdgeID"):
            if name in f.variables:
                m[name] = np.array(f.variables[name][:], dtype=np.int32)
    finally:
        f.close()
    if "edgesOnVertex" not in m:
        m["edgesOnVertex"] = edges_on_vertex(m)
    if earth_radius is not None and sphere and m["sphere_radius"] != earth_radius:
        s = float(earth_radius) / m["sphere_radius"]  # rescale_mesh: lengths by s, areas by s^2
        for k in ("xCell", "yCell", "zCell", "xVertex", "yVertex", "zVertex", "xEdge", "yEdge", "zEdge", "dcEdge",
                  "dvEdge"):
            if k in m:
                m[k] = m[k] * s
        for k in ("areaCell", "areaTriangle", "kiteAreasOnVertex"):
            if k in m:
                m[k] = m[k] * (s * s)
        m["sphere_radius"] = float(earth_radius)
    for ent, n in (("Cell", nC), ("Vertex", nV), ("Edge", nE)):
        for pre in ("lat", "lon"):
            m.setdefault(pre + ent, np.zeros(n))
        if "f" + ent not in m:
            m["f" + ent] = 2.0 * OMEGA * np.sin(m["lat" + ent]) if sphere else np.zeros(n)
    if "kiteAreasOnVertex" not in m:
        cov = m["cellsOnVertex"]
        cnt = np.maximum((cov <= nC).sum(axis=0), 1)
        m["kiteAreasOnVertex"] = np.asfortranarray(np.where(cov <= nC, m["areaTriangle"][None, :] / cnt[None, :], 0.0))
    if reconstruct:
        add_reconstruction_coefficients(m)
    return m


def write_mpas_mesh(mesh, path, version=2):
    """The mesh pool as an MPAS grid file (NetCDF classic: version 1, 64-bit offset: version 2)."""
    f = netcdf_file(path, "w", version=version)
    try:
        nC, nV, nE = int(mesh["nCells"]), int(mesh["nVertices"]), int(mesh["nEdges"])
        for d, n in (("nCells", nC), ("nVertices", nV), ("nEdges", nE), ("maxEdges", int(mesh["maxEdges"])),
                     ("vertexDegree", int(mesh["vertexDegree"])), ("TWO", 2)):
            f.createDimension(d, n)
        f.on_a_sphere = "YES" if mesh["on_a_sphere"] else "NO"
        f.sphere_radius = float(mesh["sphere_radius"])
        size = {"nCells": nC, "nVertices": nV, "nEdges": nE}
        v = f.createVariable("nEdgesOnCell", "i", ("nCells",))
        v[:] = np.asarray(mesh["nEdgesOnCell"], dtype=np.int32)
        for name, (ent, second, target) in _CONN.items():
            if name not in mesh:
                continue
            a = np.asarray(mesh[name], dtype=np.int32).T.copy()
            a[a > size[target]] = 0
            v = f.createVariable(name, "i", (ent, second))
            v[:] = a
        for ent, names in _REAL1.items():
            for name in names:
                if name in mesh:
                    v = f.createVariable(name, "d", (ent,))
                    v[:] = np.asarray(mesh[name], dtype=np.float64)
        if "kiteAreasOnVertex" in mesh:
            v = f.createVariable("kiteAreasOnVertex", "d", ("nVertices", "vertexDegree"))
            v[:] = np.asarray(mesh["kiteAreasOnVertex"], dtype=np.float64).T
        for name, ent in (("indexToCellID", "nCells"), ("indexToVertexID", "nVertices"), ("indexToEdgeID", "nEdges")):
            v = f.createVariable(name, "i", (ent,))
            v[:] = np.asarray(mesh[name], dtype=np.int32) if name in mesh else np.arange(1, size[ent] + 1,
                                                                                         dtype=np.int32)
    finally:
        f.close()


def read_graph_partition(path, nCells):
    """``graph.info.part.<N>``: the block (0-based) that owns each cell, cells in global order."""
    owner = np.loadtxt(path, dtype=np.int64, ndmin=1)
    if owner.size != nCells:
        raise ValueError(f"{path}: {owner.size} entries for a mesh of {nCells} cells")
    if owner.min() < 0:
        raise ValueError(f"{path}: negative block number")
    return owner


def write_graph_info(mesh, path):
    """``graph.info``: METIS graph of the cell adjacency (header: vertices, edges; then one line of
    1-based neighbours per cell)."""
    nC = int(mesh["nCells"])
    coc = np.asarray(mesh["cellsOnCell"])
    ne = np.asarray(mesh["nEdgesOnCell"])
    lines, total = [], 0
    for c in range(nC):
        nb = [int(x) for x in coc[:ne[c], c] if 1 <= x <= nC]
        total += len(nb)
        lines.append(" ".join(str(x) for x in nb))
    with open(path, "w") as fh:
        fh.write(f"{nC} {total // 2}\n")
        fh.write("\n".join(lines) + "\n")


# ---------------------------------------------------------------------------------------- restart
RESTART_FIELDS = [  # (pool, field, entity dimension, other dimensions as stored in the file)
    ("velocity_solver", "uVelocity", "nVertices", ()),
    ("velocity_solver", "vVelocity", "nVertices", ()),
    ("velocity_solver", "solveVelocityPrevious", "nVertices", ()),
    ("velocity_solver", "oceanStressCellU", "nCells", ()),
    ("velocity_solver", "oceanStressCellV", "nCells", ()),
    ("velocity_variational", "stress11", "nCells", ("maxEdges",)),
    ("velocity_variational", "stress22", "nCells", ("maxEdges",)),
    ("velocity_variational", "stress12", "nCells", ("maxEdges",)),
]


def _tracer_fields(block):
    out = []
    for name, a in block.structs["tracers"].items():
        lead = "ONE" if a.shape[0] == 1 else ("nIceLayers" if a.shape[0] == block.dimensions["nIceLayers"]
                                              else "nSnowLayers")
        out.append(("tracers", name, "nCells", ("nCategories", lead)))
    return out


def write_restart(domain, path, version=2):
    """Restart fields of a single-block domain, dimension order as in MPAS files: (Time, entity, ...)
    with the pool's leading (fastest) index last."""
    if len(domain.blocklist) != 1:
        raise ValueError("write_restart: gather the blocks into one domain first")
    b = domain.blocklist[0]
    d = b.dimensions
    f = netcdf_file(path, "w", version=version)
    try:
        f.createDimension("Time", None)
        for name in ("nCells", "nVertices", "maxEdges", "nCategories", "nIceLayers", "nSnowLayers"):
            f.createDimension(name, int(d[name]))
        f.createDimension("ONE", 1)
        for pool, name, ent, others in RESTART_FIELDS + _tracer_fields(b):
            a = b.structs[pool][name][..., :d[ent]]
            kind = "i" if a.dtype.kind == "i" else "d"
            v = f.createVariable(name, kind, ("Time", ent) + tuple(others))
            v[0] = np.ascontiguousarray(np.transpose(a))  # (entity, ..., leading index last)
    finally:
        f.close()


def read_restart(domain, path):
    b = domain.blocklist[0]
    d = b.dimensions
    f = netcdf_file(path, "r", mmap=False)
    try:
        for pool, name, ent, _others in RESTART_FIELDS + _tracer_fields(b):
            if name not in f.variables:
                raise KeyError(f"{path}: restart field {name} is missing")
            a = np.array(f.variables[name][0])
            b.structs[pool][name][..., :d[ent]] = np.transpose(a)
    finally:
        f.close()
