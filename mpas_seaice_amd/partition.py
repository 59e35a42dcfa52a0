"""Decomposition of a global mesh into MPAS-style blocks with halo layers and exchange lists.

Stands in for what the MPAS framework does from ``graph.info.part.N`` files (absent from the
reference repository): every block lists its owned cells first, then ``num_halos`` layers of halo
cells (config_num_halos = 2, /root/reference/src/Registry.xml:337); vertices and edges of all block
cells follow the same owned-first order, an entity being owned by the block that owns its first
cell (cellsOnVertex(1,v) / cellsOnEdge(1,e)).  Connectivity entries that leave the block become
the ``n+1`` sentinel.  Cells are cut into blocks along a Morton curve.
"""
from __future__ import annotations

import numpy as np

from .mesh import Mesh, _spatial_sort_key


def _unit_positions(mesh):
    p = np.stack([mesh["xCell"], mesh["yCell"], mesh["zCell"]], axis=1).astype(np.float64)
    lo, hi = p.min(axis=0), p.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    return (p - lo) / span * 2.0 - 1.0


def cell_owner_by_curve(mesh, nblocks):
    key = _spatial_sort_key(_unit_positions(mesh))
    order = np.argsort(key, kind="stable")
    owner = np.empty(mesh.nCells, dtype=np.int32)
    bounds = np.linspace(0, mesh.nCells, nblocks + 1).astype(np.int64)
    for r in range(nblocks):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


class BlockHalo:
    """Exchange lists of one block: per location ('cell' | 'vertex' | 'edge') and neighbour block,
    1-based local indices to send (owned here) and to receive (halo here), both in ascending global
    id order so that the two sides agree without further negotiation."""

    def __init__(self):
        self.globalID = {}
        self.send = {"cell": {}, "vertex": {}, "edge": {}}
        self.recv = {"cell": {}, "vertex": {}, "edge": {}}


def partition_mesh(mesh: Mesh, nblocks: int, num_halos: int = 2, cell_owner=None):
    """Returns (list of local Mesh, list of BlockHalo)."""
    nC, nV, nE, M = mesh.nCells, mesh.nVertices, mesh.nEdges, mesh.maxEdges
    if cell_owner is None:
        cell_owner = cell_owner_by_curve(mesh, nblocks)
    coc = mesh["cellsOnCell"].astype(np.int64) - 1      # (M, nC)
    voc = mesh["verticesOnCell"].astype(np.int64) - 1
    eoc = mesh["edgesOnCell"].astype(np.int64) - 1
    cov = mesh["cellsOnVertex"].astype(np.int64) - 1    # (3, nV)
    coe = mesh["cellsOnEdge"].astype(np.int64) - 1      # (2, nE)
    voe = mesh["verticesOnEdge"].astype(np.int64) - 1
    nedge = mesh["nEdgesOnCell"].astype(np.int64)
    valid = np.arange(M)[:, None] < nedge[None, :]

    def first_valid_owner(conn, n_ent):
        own = np.full(n_ent, -1, dtype=np.int64)
        for k in range(conn.shape[0] - 1, -1, -1):
            ok = (conn[k] >= 0) & (conn[k] < nC)
            own[ok] = cell_owner[conn[k][ok]]
        return own
    vert_owner = first_valid_owner(cov, nV)
    edge_owner = first_valid_owner(coe, nE)

    meshes, halos = [], []
    block_sets = []
    for r in range(nblocks):
        owned = np.nonzero(cell_owner == r)[0]
        in_block = np.zeros(nC, dtype=bool)
        in_block[owned] = True
        layers = [owned]
        frontier = owned
        for _ in range(num_halos):
            nb = coc[:, frontier][valid[:, frontier]]
            nb = nb[(nb >= 0) & (nb < nC)]
            nb = np.unique(nb)
            new = nb[~in_block[nb]]
            in_block[new] = True
            layers.append(new)
            frontier = new
        cells = np.concatenate(layers)
        # vertices / edges of all block cells, owned first, each group in ascending global id
        bv = np.unique(voc[:, cells][valid[:, cells]])
        be = np.unique(eoc[:, cells][valid[:, cells]])
        bv = bv[(bv >= 0) & (bv < nV)]
        be = be[(be >= 0) & (be < nE)]
        verts = np.concatenate([bv[vert_owner[bv] == r], bv[vert_owner[bv] != r]])
        edges = np.concatenate([be[edge_owner[be] == r], be[edge_owner[be] != r]])
        block_sets.append((cells, verts, edges, len(owned), int((vert_owner[bv] == r).sum()),
                           int((edge_owner[be] == r).sum())))

    for r in range(nblocks):
        cells, verts, edges, nCs, nVs, nEs = block_sets[r]
        lc, lv, le = len(cells), len(verts), len(edges)
        cmap = np.full(nC + 1, lc, dtype=np.int64)
        cmap[cells] = np.arange(lc)
        vmap = np.full(nV + 1, lv, dtype=np.int64)
        vmap[verts] = np.arange(lv)
        emap = np.full(nE + 1, le, dtype=np.int64)
        emap[edges] = np.arange(le)

        def remap(conn, ents, mp, fill):
            sub = conn[:, ents].copy()
            bad = (sub < 0) | (sub >= len(mp) - 1)
            sub[bad] = len(mp) - 1
            out = mp[sub]
            return np.asfortranarray((out + 1).astype(np.int32))
        m = Mesh()
        m["nCells"], m["nVertices"], m["nEdges"] = lc, lv, le
        m["nCellsSolve"], m["nVerticesSolve"], m["nEdgesSolve"] = nCs, nVs, nEs
        m["maxEdges"], m["vertexDegree"] = M, 3
        m["on_a_sphere"], m["sphere_radius"] = mesh["on_a_sphere"], mesh["sphere_radius"]
        m["nEdgesOnCell"] = mesh["nEdgesOnCell"][cells].astype(np.int32)
        vloc = remap(voc, cells, vmap, lv)
        eloc = remap(eoc, cells, emap, le)
        cloc = remap(coc, cells, cmap, lc)
        inval = ~valid[:, cells]
        vloc[inval] = lv + 1
        eloc[inval] = le + 1
        cloc[inval] = lc + 1
        m["verticesOnCell"], m["edgesOnCell"], m["cellsOnCell"] = vloc, eloc, cloc
        m["cellsOnVertex"] = remap(cov, verts, cmap, lc)
        m["cellsOnEdge"] = remap(coe, edges, cmap, lc)
        m["verticesOnEdge"] = remap(voe, edges, vmap, lv)
        if "edgesOnVertex" in mesh:
            m["edgesOnVertex"] = remap(mesh["edgesOnVertex"].astype(np.int64) - 1, verts, emap, le)
        for k in ("xCell", "yCell", "zCell", "latCell", "lonCell", "areaCell", "fCell"):
            m[k] = np.ascontiguousarray(mesh[k][cells])
        for k in ("xVertex", "yVertex", "zVertex", "latVertex", "lonVertex", "areaTriangle", "fVertex"):
            m[k] = np.ascontiguousarray(mesh[k][verts])
        m["kiteAreasOnVertex"] = np.asfortranarray(mesh["kiteAreasOnVertex"][:, verts])
        if "coeffs_reconstruct" in mesh:
            m["coeffs_reconstruct"] = np.asfortranarray(mesh["coeffs_reconstruct"][:, :, cells])
        for k in ("xEdge", "yEdge", "zEdge", "latEdge", "lonEdge", "dcEdge", "dvEdge", "fEdge"):
            m[k] = np.ascontiguousarray(mesh[k][edges])
        if not mesh["on_a_sphere"]:  # extent of the whole planar domain, for test cases defined on it
            m["planarExtent"] = (float(mesh["xCell"].min()), float(mesh["xCell"].max()),
                                 float(mesh["yCell"].min()), float(mesh["yCell"].max()))
        m["indexToCellID"] = (cells + 1).astype(np.int32)
        m["indexToVertexID"] = (verts + 1).astype(np.int32)
        meshes.append(m)

        h = BlockHalo()
        h.globalID = {"cell": cells, "vertex": verts, "edge": edges}
        h.nOwned = {"cell": nCs, "vertex": nVs, "edge": nEs}
        halos.append(h)

    # exchange lists
    owners = {"cell": cell_owner, "vertex": vert_owner, "edge": edge_owner}
    for loc in ("cell", "vertex", "edge"):
        own = owners[loc]
        local_of = []
        for r in range(nblocks):
            gids = halos[r].globalID[loc]
            mp = {}
            local_of.append(dict(zip(gids.tolist(), range(1, len(gids) + 1))))
        for r in range(nblocks):
            gids = halos[r].globalID[loc]
            nown = halos[r].nOwned[loc]
            hal = gids[nown:]
            src = own[hal]
            for s in np.unique(src):
                s = int(s)
                g = np.sort(hal[src == s])
                halos[r].recv[loc][s] = np.array([local_of[r][x] for x in g.tolist()], dtype=np.int32)
                halos[s].send[loc][r] = np.array([local_of[s][x] for x in g.tolist()], dtype=np.int32)
    return meshes, halos
