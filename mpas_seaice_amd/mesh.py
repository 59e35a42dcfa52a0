"""Synthetic MPAS meshes (the reference ships none; its meshes are downloaded).

Builds full-sphere quasi-uniform Voronoi meshes (geodesic class-I subdivision of the
icosahedron, frequency n -> 10 n^2 + 2 cells; n = 16, 32, 64, 128 give the 2562 / 10242 /
40962 / 163842-cell sizes the reference's convergence tests use for 480 / 240 / 120 / 60 km,
/root/reference/testing_and_setup/testcases/advection/advection_error_convergence.py:113)
and planar hexagonal meshes, with the connectivity and geometry fields of the reference's
``mesh`` pool (/root/reference/src/Registry.xml:2251-2367).

Conventions (all index arrays are 1-based like MPAS, missing neighbour = n+1 sentinel):
  * verticesOnCell / edgesOnCell / cellsOnCell are counter-clockwise seen from outside;
    edgesOnCell(i,c) joins verticesOnCell(i,c) and verticesOnCell(i+1,c)
    (the convention /root/reference/src/shared/mpas_seaice_velocity_solver_pwl.F:161-167 relies on)
    and cellsOnCell(i,c) is the cell across edgesOnCell(i,c).
  * cellsOnVertex is counter-clockwise; kiteAreasOnVertex(k,v) belongs to cellsOnVertex(k,v).
  * the edge normal points from cellsOnEdge(1,e) to cellsOnEdge(2,e); verticesOnEdge(2,e) is
    to the left of that normal.
Arrays are numpy, Fortran-ordered for 2-D fields so that ``a[i-1, c-1]`` is MPAS ``a(i,c)``
and the memory layout equals the Fortran one.
"""
from __future__ import annotations

import numpy as np

EARTH_RADIUS = 6371229.0  # config_earth_radius, /root/reference/src/Registry.xml:447
OMEGA = 7.29212e-5        # /root/reference/src/shared/mpas_seaice_constants.F:54


# ----------------------------------------------------------------------------- geodesic points
def _icosahedron():
    phi = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([
        [-1, phi, 0], [1, phi, 0], [-1, -phi, 0], [1, -phi, 0],
        [0, -1, phi], [0, 1, phi], [0, -1, -phi], [0, 1, -phi],
        [phi, 0, -1], [phi, 0, 1], [-phi, 0, -1], [-phi, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1)[:, None]
    f = np.array([
        [0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
        [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
        [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
        [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    # a small fixed rotation so that no generator sits exactly on a pole of either the
    # geographic or the rotated (x' = -z, y' = y, z' = x) grid
    a, b = 0.31, 0.17
    rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    v = v @ (rz @ rx).T
    return v, f


def geodesic_triangulation(n: int):
    """Points (N,3) on the unit sphere and CCW triangles (T,3) for frequency ``n``."""
    v, f = _icosahedron()
    # lattice (p,q): weights (n-p-q, p, q) on the face corners (A,B,C)
    pq = np.array([(p, q) for p in range(n + 1) for q in range(n + 1 - p)], dtype=np.int64)
    lat_index = -np.ones((n + 1, n + 1), dtype=np.int64)
    lat_index[pq[:, 0], pq[:, 1]] = np.arange(len(pq))
    w = np.stack([n - pq[:, 0] - pq[:, 1], pq[:, 0], pq[:, 1]], axis=1)  # (L,3)
    L = len(pq)
    nf = len(f)
    ids = np.broadcast_to(f[:, None, :], (nf, L, 3)).copy()
    ws = np.broadcast_to(w[None, :, :], (nf, L, 3)).copy()
    ids[ws == 0] = 12  # sentinel corner for zero weights
    order = np.argsort(ids, axis=2, kind="stable")
    ids_s = np.take_along_axis(ids, order, axis=2)
    ws_s = np.take_along_axis(ws, order, axis=2)
    # interior points are unique to their face: the three corner ids identify the face
    key = ((ids_s[..., 0] * 13 + ids_s[..., 1]) * 13 + ids_s[..., 2]) * (n + 1) ** 2 \
        + ws_s[..., 0] * (n + 1) + ws_s[..., 1]
    key = key.reshape(-1)
    uniq, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    # positions evaluated with corners in sorted-id order (identical for duplicates)
    vext = np.vstack([v, np.zeros((1, 3))])
    # weights sin(w_i * theta) (theta = angle between icosahedron corners) reproduce great-circle
    # interpolation along the face edges and give far more uniform cells than flat weights
    theta = np.arccos(1.0 / np.sqrt(5.0))
    sw = np.sin(ws_s.astype(np.float64) * (theta / float(n)))
    pos = (vext[ids_s[..., 0]] * sw[..., 0:1] + vext[ids_s[..., 1]] * sw[..., 1:2]
           + vext[ids_s[..., 2]] * sw[..., 2:3])
    pos = pos.reshape(-1, 3)[first]
    pos /= np.linalg.norm(pos, axis=1)[:, None]
    gid = inverse.reshape(nf, L)
    # triangles
    up = np.array([(p, q) for p in range(n) for q in range(n - p)], dtype=np.int64)
    dn = np.array([(p, q) for p in range(n - 1) for q in range(n - 1 - p)], dtype=np.int64).reshape(-1, 2)
    tri_l = [np.stack([lat_index[up[:, 0], up[:, 1]], lat_index[up[:, 0] + 1, up[:, 1]],
                       lat_index[up[:, 0], up[:, 1] + 1]], axis=1)]
    if len(dn):
        tri_l.append(np.stack([lat_index[dn[:, 0] + 1, dn[:, 1]], lat_index[dn[:, 0] + 1, dn[:, 1] + 1],
                               lat_index[dn[:, 0], dn[:, 1] + 1]], axis=1))
    tri_l = np.vstack(tri_l)  # (n^2, 3) lattice indices
    tris = gid[:, tri_l].reshape(-1, 3)
    # make sure orientation is CCW seen from outside
    a, b, c = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
    s = np.einsum("ij,ij->i", np.cross(b - a, c - a), a + b + c)
    flip = s < 0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return pos, tris


# ----------------------------------------------------------------------------- geometry helpers
def _sph_tri_area(a, b, c):
    """Unsigned area of spherical triangles on the unit sphere (rows of a, b, c are unit vectors)."""
    num = np.abs(np.einsum("ij,ij->i", a, np.cross(b, c)))
    den = 1.0 + np.einsum("ij,ij->i", a, b) + np.einsum("ij,ij->i", b, c) + np.einsum("ij,ij->i", c, a)
    return 2.0 * np.arctan2(num, den)


def _arc(a, b):
    return 2.0 * np.arcsin(np.clip(0.5 * np.linalg.norm(a - b, axis=1), 0.0, 1.0))


def _unit(x):
    return x / np.linalg.norm(x, axis=1)[:, None]


class Mesh(dict):
    """A dict of MPAS mesh-pool arrays with attribute access to dimensions."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e


def _spatial_sort_key(xyz):
    """Morton key of points on the unit sphere via a cube-face-free 3-D quantisation."""
    q = np.clip(((xyz * 0.5 + 0.5) * 1023.0).astype(np.int64), 0, 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def _build_from_triangulation(pos, tris, radius, on_a_sphere, planar_vertices=None,
                              boundary_ok=False, max_edges=None):
    """Voronoi (dual) mesh of a closed CCW triangulation. ``pos`` are unit-sphere points."""
    N = len(pos)
    T = len(tris)
    # directed edge a->b -> triangle on its left
    a = tris.reshape(-1)
    b = tris[:, [1, 2, 0]].reshape(-1)
    c = tris[:, [2, 0, 1]].reshape(-1)
    t_of = np.repeat(np.arange(T), 3)
    dkey = a * N + b
    order = np.argsort(dkey)
    dkey_s = dkey[order]

    def left_tri(u, v):
        k = u * N + v
        i = np.searchsorted(dkey_s, k)
        ok = (i < len(dkey_s))
        i = np.minimum(i, len(dkey_s) - 1)
        ok &= dkey_s[i] == k
        return np.where(ok, t_of[order][i], -1), np.where(ok, c[order][i], -1)

    # undirected edges
    lo = np.minimum(a, b)
    hi = np.maximum(a, b)
    ekey = lo * N + hi
    euniq, einv = np.unique(ekey, return_inverse=True)
    nE = len(euniq)
    e_c1 = euniq // N
    e_c2 = euniq % N

    def edge_id(u, v):
        k = np.minimum(u, v) * N + np.maximum(u, v)
        return np.searchsorted(euniq, k)

    # cell degree and CCW rings
    deg = np.bincount(a, minlength=N)
    M = int(deg.max()) if max_edges is None else int(max_edges)
    # start: the incident directed edge a->b with the smallest triangle id
    start = np.full(N, -1, dtype=np.int64)
    # iterate reversed so the smallest index wins
    idx = np.arange(len(a))[::-1]
    start[a[idx]] = idx
    cur_b = b[start]
    nEdgesOnCell = deg.astype(np.int32)
    verticesOnCell = np.full((M, N), T + 1, dtype=np.int32, order="F")
    edgesOnCell = np.full((M, N), nE + 1, dtype=np.int32, order="F")
    cellsOnCell = np.full((M, N), N + 1, dtype=np.int32, order="F")
    me = np.arange(N)
    for i in range(int(deg.max())):
        act = deg > i
        t, cc = left_tri(me[act], cur_b[act])
        assert (t >= 0).all(), "open triangulation"
        verticesOnCell[i, act] = t + 1
        # edge i joins vertex i and vertex i+1: it is the edge (cell, third corner)
        edgesOnCell[i, act] = edge_id(me[act], cc) + 1
        cellsOnCell[i, act] = cc + 1
        nb = cur_b.copy()
        nb[act] = cc
        cur_b = nb

    # vertices (triangles)
    cellsOnVertex = np.asfortranarray((tris.T + 1).astype(np.int32))
    pa, pb, pc = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
    if planar_vertices is None:
        vpos = _unit(np.cross(pb - pa, pc - pa))
    else:
        vpos = planar_vertices
    # edges
    t_left, _ = left_tri(e_c1, e_c2)    # triangle left of c1->c2  -> verticesOnEdge(2)
    t_right, _ = left_tri(e_c2, e_c1)   # triangle left of c2->c1  -> verticesOnEdge(1)
    cellsOnEdge = np.asfortranarray(np.stack([e_c1 + 1, e_c2 + 1]).astype(np.int32))
    verticesOnEdge = np.asfortranarray(np.stack([t_right + 1, t_left + 1]).astype(np.int32))
    epos = _unit(pos[e_c1] + pos[e_c2])
    # edgesOnVertex(k,v): edge joining cellsOnVertex(k,v) and cellsOnVertex(k+1,v)
    edgesOnVertex = np.asfortranarray(np.stack([
        edge_id(tris[:, 0], tris[:, 1]), edge_id(tris[:, 1], tris[:, 2]),
        edge_id(tris[:, 2], tris[:, 0])]).astype(np.int32) + 1)

    m = Mesh()
    m["nCells"], m["nVertices"], m["nEdges"] = N, T, nE
    m["nCellsSolve"], m["nVerticesSolve"], m["nEdgesSolve"] = N, T, nE
    m["maxEdges"], m["vertexDegree"] = M, 3
    m["on_a_sphere"], m["sphere_radius"] = bool(on_a_sphere), float(radius)
    m["nEdgesOnCell"] = nEdgesOnCell
    m["verticesOnCell"], m["edgesOnCell"], m["cellsOnCell"] = verticesOnCell, edgesOnCell, cellsOnCell
    m["cellsOnVertex"], m["edgesOnVertex"] = cellsOnVertex, edgesOnVertex
    m["cellsOnEdge"], m["verticesOnEdge"] = cellsOnEdge, verticesOnEdge
    m["_unit_cell"], m["_unit_vertex"], m["_unit_edge"] = pos, vpos, epos
    m["_tris"] = tris
    return m


def _finish_sphere(m, radius):
    pos, vpos, epos, tris = m["_unit_cell"], m["_unit_vertex"], m["_unit_edge"], m["_tris"]
    N, T, nE, M = m.nCells, m.nVertices, m.nEdges, m.maxEdges
    voc = m["verticesOnCell"].astype(np.int64) - 1
    # areas on the unit sphere
    areaCell = np.zeros(N)
    for i in range(M):
        act = m["nEdgesOnCell"] > i
        nxt = np.where(m["nEdgesOnCell"] > i + 1, i + 1, 0)
        v1 = voc[i, act]
        v2 = voc[nxt[act], np.nonzero(act)[0]]
        areaCell[act] += _sph_tri_area(pos[act], vpos[v1], vpos[v2])
    kite = np.zeros((3, T), order="F")
    eov = m["edgesOnVertex"].astype(np.int64) - 1
    for k in range(3):
        ck = pos[tris[:, k]]
        e_next = epos[eov[k]]            # edge (c_k, c_k+1)
        e_prev = epos[eov[(k + 2) % 3]]  # edge (c_k-1, c_k)
        kite[k] = _sph_tri_area(ck, e_next, vpos) + _sph_tri_area(ck, vpos, e_prev)
    c1 = m["cellsOnEdge"][0].astype(np.int64) - 1
    c2 = m["cellsOnEdge"][1].astype(np.int64) - 1
    v1 = m["verticesOnEdge"][0].astype(np.int64) - 1
    v2 = m["verticesOnEdge"][1].astype(np.int64) - 1
    dc = _arc(pos[c1], pos[c2])
    dv = _arc(vpos[v1], vpos[v2])
    # rescale the unit sphere exactly the way the reference does at start-up
    # (/root/reference/src/shared/mpas_seaice_mesh.F:206-259)
    old_r = 1.0

    def scale(p):
        nrm = np.sqrt(p[:, 0] ** 2 + p[:, 1] ** 2 + p[:, 2] ** 2)
        return (p / nrm[:, None]) * radius
    pc, pv, pe = scale(pos), scale(vpos), scale(epos)
    m["xCell"], m["yCell"], m["zCell"] = (np.ascontiguousarray(pc[:, i]) for i in range(3))
    m["xVertex"], m["yVertex"], m["zVertex"] = (np.ascontiguousarray(pv[:, i]) for i in range(3))
    m["xEdge"], m["yEdge"], m["zEdge"] = (np.ascontiguousarray(pe[:, i]) for i in range(3))
    m["areaCell"] = (areaCell / old_r ** 2) * radius ** 2
    kite = (kite / old_r ** 2) * radius ** 2
    m["kiteAreasOnVertex"] = np.asfortranarray(kite)
    m["areaTriangle"] = kite[0] + kite[1] + kite[2]
    m["dcEdge"] = (dc / old_r) * radius
    m["dvEdge"] = (dv / old_r) * radius
    for nm, p in (("Cell", pc), ("Vertex", pv), ("Edge", pe)):
        m["lat" + nm] = np.arcsin(np.clip(p[:, 2] / radius, -1.0, 1.0))
        lon = np.arctan2(p[:, 1], p[:, 0])
        m["lon" + nm] = np.where(lon < 0.0, lon + 2.0 * np.pi, lon)
    m["fVertex"] = 2.0 * OMEGA * np.sin(m["latVertex"])  # mpas_seaice_mesh.F:349
    m["fCell"] = 2.0 * OMEGA * np.sin(m["latCell"])
    m["fEdge"] = 2.0 * OMEGA * np.sin(m["latEdge"])
    for k in ("_unit_cell", "_unit_vertex", "_unit_edge", "_tris"):
        del m[k]
    return m


def _reorder(pos, tris, reorder):
    if not reorder:
        return pos, tris
    perm = np.argsort(_spatial_sort_key(pos), kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    pos = pos[perm]
    tris = inv[tris]
    # triangles sorted by their smallest corner: vertices numbered near their cells
    tperm = np.argsort(tris.min(axis=1) * 4 + np.argmin(tris, axis=1), kind="stable")
    return pos, tris[tperm]


def lloyd_relax(pos, tris, iters):
    """Fixed-topology Lloyd iterations: move generators to the area centroid of their cell."""
    for _ in range(iters):
        pa, pb, pc = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
        vpos = _unit(np.cross(pb - pa, pc - pa))
        acc = np.zeros_like(pos)
        # each triangle corner k contributes the kite (c_k, m_k, v, m_k-1) centroid weighted by area
        mids = [_unit(pa + pb), _unit(pb + pc), _unit(pc + pa)]
        corners = [pa, pb, pc]
        for k in range(3):
            ck = corners[k]
            e_next, e_prev = mids[k], mids[(k + 2) % 3]
            for (p, q, r) in ((ck, e_next, vpos), (ck, vpos, e_prev)):
                ar = _sph_tri_area(p, q, r)
                cen = _unit(p + q + r)
                np.add.at(acc, tris[:, k], cen * ar[:, None])
        pos = _unit(acc)
    return pos


def make_sphere_mesh(n: int, radius: float = EARTH_RADIUS, reorder: bool = False,
                     lloyd_iters: int = 0) -> Mesh:
    """Quasi-uniform full-sphere mesh with 10 n^2 + 2 cells, rescaled to ``radius``.

    ``reorder=False`` keeps the generator's face-by-face numbering (poor locality, which is what
    an arbitrary MPAS file looks like to the solver); ``reorder=True`` numbers cells along a
    Morton curve.
    """
    pos, tris = geodesic_triangulation(n)
    if lloyd_iters:
        pos = lloyd_relax(pos, tris, lloyd_iters)
    pos, tris = _reorder(pos, tris, reorder)
    m = _build_from_triangulation(pos, tris, radius, True)
    m = _finish_sphere(m, radius)
    add_reconstruction_coefficients(m)
    return m


def _delaunay_on_sphere(pos):
    """Delaunay triangulation of points on the unit sphere = their convex hull; triangles CCW seen from outside."""
    from scipy.spatial import ConvexHull
    tris = ConvexHull(pos).simplices.astype(np.int64)
    pa, pb, pc = pos[tris[:, 0]], pos[tris[:, 1]], pos[tris[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(pb - pa, pc - pa), pa + pb + pc) < 0.0
    tris[flip] = tris[flip][:, [0, 2, 1]]
    return tris


def make_variable_sphere_mesh(n: int, ratio: float = 3.0, jitter: float = 0.22, radius: float = EARTH_RADIUS,
                              lloyd_iters: int = 2, seed: int = 20241020) -> Mesh:
    """Variable-resolution full-sphere Voronoi mesh with cells of 5, 6 and 7 (occasionally 8) edges: a stand-in
    for the refined-region meshes (oRRS 18-6 km) the reference is run on, which are not redistributable.

    The generators of the geodesic mesh are pulled towards the north pole by the map
    z' = (z + a) / (1 + a z), a = (ratio - 1) / (ratio + 1) (spacing ratio ``ratio`` between the two poles),
    displaced by a seeded pseudo-random fraction ``jitter`` of the local spacing, triangulated again
    (Delaunay) and smoothed by a few Lloyd iterations with re-triangulation.
    """
    pos, _ = geodesic_triangulation(n)
    a = (ratio - 1.0) / (ratio + 1.0)
    z = pos[:, 2]
    z2 = (z + a) / (1.0 + a * z)
    s = np.sqrt(np.maximum(1.0 - z2 * z2, 0.0)) / np.sqrt(np.maximum(1.0 - z * z, 1e-300))
    pos = _unit(np.stack([pos[:, 0] * s, pos[:, 1] * s, z2], axis=1) + 1e-300)
    spacing = np.sqrt(4.0 * np.pi / len(pos)) * (1.0 - a * a) / (1.0 + a * z) ** 1.0 / np.sqrt(1.0 - a * a)
    rng = np.random.Generator(np.random.PCG64(seed))
    pos = _unit(pos + jitter * spacing[:, None] * rng.standard_normal(pos.shape) / np.sqrt(3.0))
    tris = _delaunay_on_sphere(pos)
    for _ in range(lloyd_iters):
        pos = lloyd_relax(pos, tris, 1)
        tris = _delaunay_on_sphere(pos)
    pos, tris = _reorder(pos, tris, False)
    m = _build_from_triangulation(pos, tris, radius, True)
    m = _finish_sphere(m, radius)
    add_reconstruction_coefficients(m)
    return m


QU_FREQ = {"QU480": 16, "QU240": 32, "QU120": 64, "QU60": 128, "QU30": 256, "QU15": 512}


def make_named_mesh(name: str, **kw) -> Mesh:
    """QU480 ... QU15: the quasi-uniform geodesic meshes; var<n>: the variable-resolution mesh with 5-, 6- and
    7-sided cells (maxEdges 7) on 10 n^2 + 2 generators (var128: 163 842 cells, var256: 655 362)."""
    if name.startswith("var"):
        return make_variable_sphere_mesh(int(name[3:]), **kw)
    return make_sphere_mesh(QU_FREQ[name], **kw)


# ----------------------------------------------------------------------------- planar hex mesh
def make_planar_hex_mesh(nx: int, ny: int, dc: float) -> Mesh:
    """Non-periodic planar mesh of regular hexagons (nx by ny cells, centre spacing ``dc``).

    Boundary cells have fewer than 6 real neighbours; their missing neighbours / vertices that
    would lie outside are still present as mesh vertices (every cell is a full hexagon) but the
    vertices on the rim touch fewer than 3 cells and carry the nCells+1 sentinel in
    cellsOnVertex, so ``interiorVertex`` is 0 there
    (/root/reference/src/shared/mpas_seaice_mesh.F:461-486).
    """
    h = dc / np.sqrt(3.0)  # hexagon circumradius (pointy-top)
    cx = np.zeros((ny, nx))
    cy = np.zeros((ny, nx))
    for j in range(ny):
        cx[j] = (np.arange(nx) + 0.5 * (j % 2)) * dc
        cy[j] = j * 1.5 * h
    cx = cx.reshape(-1)
    cy = cy.reshape(-1)
    N = nx * ny
    ang = np.deg2rad(30.0 + 60.0 * np.arange(6))  # CCW corners of a pointy-top hexagon
    vx = cx[:, None] + h * np.cos(ang)[None, :]
    vy = cy[:, None] + h * np.sin(ang)[None, :]
    key = np.round(vx / (dc * 0.25)).astype(np.int64) * 100003 + np.round(vy / (h * 0.25)).astype(np.int64)
    uniq, first, inv = np.unique(key.reshape(-1), return_index=True, return_inverse=True)
    T = len(uniq)
    xV = vx.reshape(-1)[first]
    yV = vy.reshape(-1)[first]
    voc = inv.reshape(N, 6)
    M = 6
    verticesOnCell = np.asfortranarray((voc.T + 1).astype(np.int32))
    # cells on vertex (CCW), sentinel N+1 where missing
    cov = [[] for _ in range(T)]
    for c in range(N):
        for i in range(6):
            cov[voc[c, i]].append(c)
    cellsOnVertex = np.full((3, T), N + 1, dtype=np.int32, order="F")
    for v in range(T):
        cs = cov[v]
        a = np.arctan2(cy[cs] - yV[v], cx[cs] - xV[v])
        cs = [cs[i] for i in np.argsort(a)]
        for k, c in enumerate(cs):
            cellsOnVertex[k, v] = c + 1
    # edges: between consecutive vertices of each cell
    e1 = voc
    e2 = np.roll(voc, -1, axis=1)
    ek = np.minimum(e1, e2).astype(np.int64) * T + np.maximum(e1, e2)
    euniq, efirst, einv = np.unique(ek.reshape(-1), return_index=True, return_inverse=True)
    nE = len(euniq)
    edgesOnCell = np.asfortranarray((einv.reshape(N, 6).T + 1).astype(np.int32))
    cellsOnEdge = np.full((2, nE), N + 1, dtype=np.int32, order="F")
    cnt = np.zeros(nE, dtype=np.int64)
    for c in range(N):
        for i in range(6):
            e = einv.reshape(N, 6)[c, i]
            cellsOnEdge[cnt[e], e] = c + 1
            cnt[e] += 1
    cellsOnCell = np.full((6, N), N + 1, dtype=np.int32, order="F")
    for c in range(N):
        for i in range(6):
            e = einv.reshape(N, 6)[c, i]
            pair = cellsOnEdge[:, e] - 1
            other = pair[0] if pair[1] == c else pair[1]
            cellsOnCell[i, c] = other + 1 if other != c else N + 1
    verticesOnEdge = np.zeros((2, nE), dtype=np.int32, order="F")
    ev1 = (euniq // T).astype(np.int64)
    ev2 = (euniq % T).astype(np.int64)
    xE = 0.5 * (xV[ev1] + xV[ev2])
    yE = 0.5 * (yV[ev1] + yV[ev2])
    for e in range(nE):
        c1 = cellsOnEdge[0, e] - 1
        # normal from c1 outwards through the edge; vertex 2 on its left
        nx_, ny_ = xE[e] - cx[c1], yE[e] - cy[c1]
        tx, ty = -ny_, nx_
        if (xV[ev2[e]] - xV[ev1[e]]) * tx + (yV[ev2[e]] - yV[ev1[e]]) * ty > 0:
            verticesOnEdge[:, e] = (ev1[e] + 1, ev2[e] + 1)
        else:
            verticesOnEdge[:, e] = (ev2[e] + 1, ev1[e] + 1)
    m = Mesh()
    m["nCells"], m["nVertices"], m["nEdges"] = N, T, nE
    m["nCellsSolve"], m["nVerticesSolve"], m["nEdgesSolve"] = N, T, nE
    m["maxEdges"], m["vertexDegree"] = M, 3
    m["on_a_sphere"], m["sphere_radius"] = False, 1.0
    m["nEdgesOnCell"] = np.full(N, 6, dtype=np.int32)
    m["verticesOnCell"], m["edgesOnCell"], m["cellsOnCell"] = verticesOnCell, edgesOnCell, cellsOnCell
    m["cellsOnVertex"], m["cellsOnEdge"], m["verticesOnEdge"] = cellsOnVertex, cellsOnEdge, verticesOnEdge
    m["xCell"], m["yCell"], m["zCell"] = cx, cy, np.zeros(N)
    m["xVertex"], m["yVertex"], m["zVertex"] = xV, yV, np.zeros(T)
    m["xEdge"], m["yEdge"], m["zEdge"] = xE, yE, np.zeros(nE)
    hex_area = 1.5 * np.sqrt(3.0) * h * h
    m["areaCell"] = np.full(N, hex_area)
    kite = np.where(cellsOnVertex <= N, hex_area / 6.0, 0.0)
    m["kiteAreasOnVertex"] = np.asfortranarray(kite)
    m["areaTriangle"] = np.full(T, hex_area / 2.0)
    m["dcEdge"] = np.full(nE, dc)
    m["dvEdge"] = np.full(nE, h)
    for nm, n_ in (("Cell", N), ("Vertex", T), ("Edge", nE)):
        m["lat" + nm] = np.zeros(n_)
        m["lon" + nm] = np.zeros(n_)
    m["fVertex"] = np.zeros(T)
    m["fCell"] = np.zeros(N)
    m["fEdge"] = np.zeros(nE)
    add_reconstruction_coefficients(m)
    return m


# ----------------------------------------------------------------------------- quadrilateral meshes
def _structured_quad_connectivity(nx: int, ny: int, periodic_x: bool):
    """Connectivity of nx by ny quadrilateral cells (vertexDegree 4, maxEdges 4), 1-based with the n+1 sentinel.

    Cell (i, j) has the vertices (i, j), (i+1, j), (i+1, j+1), (i, j+1) counter-clockwise and its k-th edge joins
    its k-th and (k+1)-th vertex (bottom, right, top, left), the convention of the other generators of this module.
    ``periodic_x`` closes the mesh in i (a band around a sphere); in j it always ends at a boundary."""
    nvx = nx if periodic_x else nx + 1
    N, T = nx * ny, nvx * (ny + 1)
    nH = nx * (ny + 1)
    nE = nH + nvx * ny

    def vid(i, j):
        return j * nvx + (i % nvx if periodic_x else i)

    def cid(i, j):
        if j < 0 or j >= ny:
            return N
        if periodic_x:
            return j * nx + i % nx
        return j * nx + i if 0 <= i < nx else N

    def hid(i, j):
        return j * nx + (i % nx if periodic_x else i)

    def wid(i, j):
        return nH + j * nvx + (i % nvx if periodic_x else i)

    voc = np.zeros((4, N), dtype=np.int32, order="F")
    eoc = np.zeros((4, N), dtype=np.int32, order="F")
    coc = np.zeros((4, N), dtype=np.int32, order="F")
    for j in range(ny):
        for i in range(nx):
            c = j * nx + i
            voc[:, c] = (vid(i, j) + 1, vid(i + 1, j) + 1, vid(i + 1, j + 1) + 1, vid(i, j + 1) + 1)
            eoc[:, c] = (hid(i, j) + 1, wid(i + 1, j) + 1, hid(i, j + 1) + 1, wid(i, j) + 1)
            coc[:, c] = (cid(i, j - 1) + 1, cid(i + 1, j) + 1, cid(i, j + 1) + 1, cid(i - 1, j) + 1)
    coe = np.full((2, nE), N + 1, dtype=np.int32, order="F")
    voe = np.zeros((2, nE), dtype=np.int32, order="F")
    # cell 1 is the lower-numbered existing cell and lies to the left of vertex 1 -> vertex 2
    for j in range(ny + 1):
        for i in range(nx):
            e = hid(i, j)
            south, north = cid(i, j - 1), cid(i, j)
            if south < N:
                coe[:, e] = (south + 1, north + 1)
                voe[:, e] = (vid(i + 1, j) + 1, vid(i, j) + 1)      # heading west: the southern cell on the left
            else:
                coe[:, e] = (north + 1, N + 1)
                voe[:, e] = (vid(i, j) + 1, vid(i + 1, j) + 1)
    for j in range(ny):
        for i in range(nvx):
            e = wid(i, j)
            west, east = cid(i - 1, j), cid(i, j)
            if west < N and east < N and east < west:                  # the periodic seam: cell 1 = the lower number
                coe[:, e] = (east + 1, west + 1)
                voe[:, e] = (vid(i, j + 1) + 1, vid(i, j) + 1)      # heading south: the eastern cell on the left
            elif west < N:
                coe[:, e] = (west + 1, east + 1)
                voe[:, e] = (vid(i, j) + 1, vid(i, j + 1) + 1)      # heading north: the western cell on the left
            else:
                coe[:, e] = (east + 1, N + 1)
                voe[:, e] = (vid(i, j + 1) + 1, vid(i, j) + 1)
    cov = np.full((4, T), N + 1, dtype=np.int32, order="F")
    eov = np.full((4, T), nE + 1, dtype=np.int32, order="F")
    for j in range(ny + 1):
        for i in range(nvx):
            v = vid(i, j)
            cells = [cid(i, j), cid(i - 1, j), cid(i - 1, j - 1), cid(i, j - 1)]                      # NE, NW, SW, SE
            edges = [hid(i, j) if (periodic_x or i < nx) else nE, wid(i, j) if j < ny else nE,
                     hid(i - 1, j) if (periodic_x or i > 0) else nE, wid(i, j - 1) if j > 0 else nE]   # E, N, W, S
            cells = [c for c in cells if c < N]
            edges = [e for e in edges if e < nE]
            cov[:len(cells), v] = np.asarray(cells) + 1
            eov[:len(edges), v] = np.asarray(edges) + 1
    return dict(N=N, T=T, nE=nE, nvx=nvx, voc=voc, eoc=eoc, coc=coc, coe=coe, voe=voe, cov=cov, eov=eov)


def _quad_mesh_from(con, on_a_sphere, radius):
    m = Mesh()
    N, T, nE = con["N"], con["T"], con["nE"]
    m["nCells"], m["nVertices"], m["nEdges"] = N, T, nE
    m["nCellsSolve"], m["nVerticesSolve"], m["nEdgesSolve"] = N, T, nE
    m["maxEdges"], m["vertexDegree"] = 4, 4
    m["on_a_sphere"], m["sphere_radius"] = bool(on_a_sphere), float(radius)
    m["nEdgesOnCell"] = np.full(N, 4, dtype=np.int32)
    m["verticesOnCell"], m["edgesOnCell"], m["cellsOnCell"] = con["voc"], con["eoc"], con["coc"]
    m["cellsOnVertex"], m["edgesOnVertex"] = con["cov"], con["eov"]
    m["cellsOnEdge"], m["verticesOnEdge"] = con["coe"], con["voe"]
    return m


def make_planar_quad_mesh(nx: int, ny: int, dc: float) -> Mesh:
    """Non-periodic planar mesh of nx by ny squares of side ``dc`` (vertexDegree 4): the shape of the reference's
    ``square`` test-case meshes (/root/reference/testing_and_setup/testcases/square/square_quadhex), for the
    quadrilateral branches of the incremental remap."""
    con = _structured_quad_connectivity(nx, ny, False)
    m = _quad_mesh_from(con, False, 1.0)
    N, T, nE = con["N"], con["T"], con["nE"]
    ci, cj = np.meshgrid(np.arange(nx), np.arange(ny))
    m["xCell"], m["yCell"], m["zCell"] = ((ci + 0.5) * dc).reshape(-1), ((cj + 0.5) * dc).reshape(-1), np.zeros(N)
    vi, vj = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1))
    m["xVertex"], m["yVertex"], m["zVertex"] = (vi * dc).reshape(-1).astype(float), (vj * dc).reshape(-1).astype(float), np.zeros(T)
    voe = con["voe"].astype(np.int64) - 1
    m["xEdge"] = 0.5 * (m["xVertex"][voe[0]] + m["xVertex"][voe[1]])
    m["yEdge"] = 0.5 * (m["yVertex"][voe[0]] + m["yVertex"][voe[1]])
    m["zEdge"] = np.zeros(nE)
    m["areaCell"] = np.full(N, dc * dc)
    m["kiteAreasOnVertex"] = np.asfortranarray(np.where(con["cov"] <= N, 0.25 * dc * dc, 0.0))
    m["areaTriangle"] = np.full(T, dc * dc)
    m["dcEdge"] = np.full(nE, dc)
    m["dvEdge"] = np.full(nE, dc)
    for nm, n_ in (("Cell", N), ("Vertex", T), ("Edge", nE)):
        m["lat" + nm] = np.zeros(n_)
        m["lon" + nm] = np.zeros(n_)
        m["f" + nm] = np.zeros(n_)
    add_reconstruction_coefficients(m)
    return m


def make_latlon_quad_mesh(nlon: int, nlat: int, lat_max_deg: float = 60.0, radius: float = EARTH_RADIUS) -> Mesh:
    """A band of nlon by nlat latitude-longitude quadrilaterals between -lat_max and +lat_max on the sphere, closed in
    longitude, with boundaries along the two outer parallels (vertexDegree 4).  Stands in for the reference's
    spherical quadrilateral meshes (the gx3 / gx1 cases named at
    /root/reference/src/shared/mpas_seaice_advection_incremental_remap.F:105-119, not in the repository): on the
    sphere the edges E5 / E6 beside an edge along a parallel bend away from its line, which is what the split of
    side triangles (:5807-5958) is for."""
    con = _structured_quad_connectivity(nlon, nlat, True)
    m = _quad_mesh_from(con, True, radius)
    N, T, nE = con["N"], con["T"], con["nE"]
    dlon = 2.0 * np.pi / nlon
    L = np.deg2rad(lat_max_deg)
    dlat = 2.0 * L / nlat

    def sph(lon, lat):
        return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1)

    ci, cj = np.meshgrid(np.arange(nlon), np.arange(nlat))
    pc = sph(((ci + 0.5) * dlon).reshape(-1), (-L + (cj + 0.5) * dlat).reshape(-1))
    vi, vj = np.meshgrid(np.arange(nlon), np.arange(nlat + 1))
    pv = sph((vi * dlon).reshape(-1), (-L + vj * dlat).reshape(-1))
    voe = con["voe"].astype(np.int64) - 1
    coe = con["coe"].astype(np.int64) - 1
    pe = _unit(pv[voe[0]] + pv[voe[1]])
    m["xCell"], m["yCell"], m["zCell"] = (np.ascontiguousarray(radius * pc[:, k]) for k in range(3))
    m["xVertex"], m["yVertex"], m["zVertex"] = (np.ascontiguousarray(radius * pv[:, k]) for k in range(3))
    m["xEdge"], m["yEdge"], m["zEdge"] = (np.ascontiguousarray(radius * pe[:, k]) for k in range(3))
    lat_lo = (-L + cj * dlat).reshape(-1)
    m["areaCell"] = radius ** 2 * dlon * (np.sin(lat_lo + dlat) - np.sin(lat_lo))
    latv = (-L + vj * dlat).reshape(-1)
    lo, hi = np.maximum(latv - 0.5 * dlat, -L), np.minimum(latv + 0.5 * dlat, L)
    m["areaTriangle"] = radius ** 2 * dlon * (np.sin(hi) - np.sin(lo))
    cov = con["cov"].astype(np.int64) - 1
    ncv = np.maximum((cov < N).sum(axis=0), 1)
    m["kiteAreasOnVertex"] = np.asfortranarray(np.where(cov < N, (m["areaTriangle"] / ncv)[None, :], 0.0))
    m["dvEdge"] = radius * _arc(pv[voe[0]], pv[voe[1]])
    both = coe[1] < N
    c2 = np.where(both, coe[1], coe[0])
    m["dcEdge"] = np.where(both, radius * _arc(pc[coe[0]], pc[c2]), radius * dlat)
    for nm, p in (("Cell", pc), ("Vertex", pv), ("Edge", pe)):
        m["lat" + nm] = np.arcsin(np.clip(p[:, 2], -1.0, 1.0))
        m["lon" + nm] = np.mod(np.arctan2(p[:, 1], p[:, 0]), 2.0 * np.pi)
        m["f" + nm] = 2.0 * OMEGA * np.sin(m["lat" + nm])
    add_reconstruction_coefficients(m)
    return m


# ----------------------------------------------------------------------------- reconstruction
def edges_on_vertex(m: Mesh):
    """edgesOnVertex (vertexDegree, nVertices) from verticesOnEdge, nEdges+1 where a vertex has fewer edges."""
    nV, nE, D = m["nVertices"], m["nEdges"], int(m["vertexDegree"])
    eov = np.full((D, nV), nE + 1, dtype=np.int32, order="F")
    cnt = np.zeros(nV, dtype=np.int64)
    voe = np.asarray(m["verticesOnEdge"]) - 1
    for e in range(nE):
        for k in range(2):
            v = voe[k, e]
            if 0 <= v < nV and cnt[v] < D:
                eov[cnt[v], v] = e + 1
                cnt[v] += 1
    return eov


def add_reconstruction_coefficients(m: Mesh):
    """``coeffs_reconstruct(3, maxEdges, nCells)``: weights that turn the normal components of a
    gradient on the edges of a cell into the gradient vector at the cell centre (global x, y, z).

    In the reference this array is the output of the MPAS framework operator ``mpas_init_reconstruct``
    (radial basis functions), called from
    /root/reference/src/shared/mpas_seaice_advection_incremental_remap.F:744-746; the framework is not
    part of the reference repository.  Here a least-squares reconstruction plays that role: with n_e
    the unit vector from cellsOnEdge(1,e) to cellsOnEdge(2,e) projected on the tangent plane at the
    cell centre and r the local vertical, G = A^-1 sum_e n_e g_e with A = sum_e n_e n_e^T + r r^T,
    which is exact for linear fields.  The same array feeds the device library and the CPU oracle.
    """
    nC, M = m["nCells"], m["maxEdges"]
    xyz = np.stack([m["xCell"], m["yCell"], m["zCell"]], axis=1)
    coe = np.asarray(m["cellsOnEdge"]).astype(np.int64) - 1
    eoc = np.asarray(m["edgesOnCell"]).astype(np.int64) - 1
    ne = np.asarray(m["nEdgesOnCell"]).astype(np.int64)
    if m["on_a_sphere"]:
        r = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
    else:
        r = np.tile(np.array([0.0, 0.0, 1.0]), (nC, 1))
    nvec = np.zeros((nC, M, 3))
    for i in range(M):
        e = np.clip(eoc[i], 0, m["nEdges"] - 1)
        c1, c2 = coe[0, e], coe[1, e]
        ok = (i < ne) & (c1 >= 0) & (c1 < nC) & (c2 >= 0) & (c2 < nC)
        c1c, c2c = np.clip(c1, 0, nC - 1), np.clip(c2, 0, nC - 1)
        d = xyz[c2c] - xyz[c1c]
        d = d - (d * r).sum(axis=1, keepdims=True) * r
        nrm = np.linalg.norm(d, axis=1, keepdims=True)
        d = np.where(nrm > 0, d / np.where(nrm > 0, nrm, 1.0), 0.0)
        nvec[:, i, :] = np.where(ok[:, None], d, 0.0)
    A = np.einsum("cik,cil->ckl", nvec, nvec) + np.einsum("ck,cl->ckl", r, r)
    Ainv = np.linalg.inv(A)
    coeffs = np.einsum("ckl,cil->cik", Ainv, nvec)        # (nC, M, 3)
    m["coeffs_reconstruct"] = np.asfortranarray(coeffs.transpose(2, 1, 0))
    if "edgesOnVertex" not in m:
        m["edgesOnVertex"] = edges_on_vertex(m)


# ----------------------------------------------------------------------------- checks
def check_mesh(m: Mesh) -> dict:
    """Consistency diagnostics used by the tests."""
    out = {}
    N, T, nE = m.nCells, m.nVertices, m.nEdges
    if m.on_a_sphere:
        r = m.sphere_radius
        out["area_cells_rel"] = abs(m["areaCell"].sum() / (4 * np.pi * r * r) - 1.0)
        out["area_tri_rel"] = abs(m["areaTriangle"].sum() / (4 * np.pi * r * r) - 1.0)
        out["euler"] = N - nE + T
    voc = m["verticesOnCell"].astype(np.int64) - 1
    cov = m["cellsOnVertex"].astype(np.int64) - 1
    # every (cell, vertex-on-cell) pair must be mirrored in cellsOnVertex
    ok = True
    for i in range(m.maxEdges):
        act = m["nEdgesOnCell"] > i
        v = voc[i, act]
        c = np.nonzero(act)[0]
        ok &= bool(((cov[:, v] == c[None, :]).sum(axis=0) == 1).all())
    out["voc_cov_consistent"] = ok
    return out
