"""Host mirror of ``seaice_init_mesh`` (/root/reference/src/shared/mpas_seaice_mesh.F:44-70), what the
reference does to the mesh pool before the velocity solver and the advection are initialised: ``rescale_mesh``
(:82-310), ``coriolis_parameter`` (:316-358) and ``init_boundary`` (:372-420).  The first two work on the host
pools, once; the interior flags of ``init_boundary`` are computed by the device library during its own
initialisation (``interiorVertex`` comes back with ``seaice_init_velocity_solver``, ``interiorEdge`` with the
upwind advection's initialisation), so nothing is done for them here.

Call it before ``seaice_mesh_pool_create``: the library takes its copy of the mesh there.
"""
from __future__ import annotations

import numpy as np

OMEGA = 7.29212e-5  # seaice_constants: angular rotation rate of the Earth [1/s]


def rescale_mesh(domain):
    """:82-310.  Positions are projected onto the new radius point by point, areas and lengths scale with the
    ratio of the radii; areaTriangle is rebuilt from the rescaled kite areas."""
    cfg = domain.configs
    new = float(cfg["config_earth_radius"])
    for b in domain.blocklist:
        m = b.structs["mesh"]
        if not bool(m["on_a_sphere"]):
            continue
        old = float(m["sphere_radius"])
        if cfg.get("config_do_restart", False):
            if old != new:
                raise RuntimeError(f"rescale_mesh: inconsistent radii: sphere_radius: {old}, earthRadius: {new}")
            continue
        d = b.dimensions
        for loc, n in (("Cell", d["nCells"]), ("Vertex", d["nVertices"]), ("Edge", d["nEdges"])):
            x, y, z = (m[c + loc] for c in "xyz")
            norm = np.sqrt(x[:n] ** 2 + y[:n] ** 2 + z[:n] ** 2)
            for a in (x, y, z):
                a[:n] = (a[:n] / norm) * new
        nC, nV, nE = d["nCells"], d["nVertices"], d["nEdges"]
        m["areaCell"][:nC] = (m["areaCell"][:nC] / (old ** 2)) * new ** 2
        kite = m["kiteAreasOnVertex"]
        kite[:, :nV] = (kite[:, :nV] / old ** 2) * new ** 2
        tri = np.zeros(nV)
        for k in range(kite.shape[0]):  # summed in vertexDegree order like the reference's loop
            tri = tri + kite[k, :nV]
        m["areaTriangle"][:nV] = tri
        m["dvEdge"][:nE] = (m["dvEdge"][:nE] / old) * new
        m["dcEdge"][:nE] = (m["dcEdge"][:nE] / old) * new
        m["sphere_radius"] = new


def coriolis_parameter(domain):
    """:316-358"""
    if not domain.configs.get("config_calculate_coriolis", True):
        return
    for b in domain.blocklist:
        m = b.structs["mesh"]
        nV = b.dimensions["nVertices"]
        m["fVertex"][:nV] = 2.0 * OMEGA * np.sin(m["latVertex"][:nV])


def seaice_init_mesh(domain):
    """:44-70"""
    rescale_mesh(domain)
    coriolis_parameter(domain)
