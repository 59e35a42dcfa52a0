"""TEST INFRASTRUCTURE ONLY: the reference's incremental-remap call flow executed with the CPU oracle.

Walks the same sequence as /root/reference/src/shared/mpas_seaice_advection_incremental_remap.F
(seaice_init_advection_incremental_remap :165, seaice_run_advection_incremental_remap :2338,
incremental_remap_block :2740) over the pool containers of ``mpas_seaice_amd.pools`` and calls
``liboracle.so`` (ir_oracle.c) for every loop.  Parity status: see the header of ir_oracle.c
("parity unpinned" at bit level).
"""
from __future__ import annotations

import ctypes

import numpy as np

from mpas_seaice_amd import tracers as trc
from . import evp_flow

NTRI, NQP, NVER, NEER = 6, 6, 8, 6
call = evp_flow.call


def calli(name, *args):
    """Like evp_flow.call for routines that return an int."""
    fn = getattr(evp_flow.lib(), name)
    fn.restype = ctypes.c_int
    cargs = []
    for a in args:
        if a is None:
            cargs.append(ctypes.c_void_p(0))
        elif isinstance(a, np.ndarray):
            cargs.append(ctypes.c_void_p(a.ctypes.data))
        elif isinstance(a, (bool, np.bool_, int, np.integer)):
            cargs.append(ctypes.c_int(int(a)))
        elif isinstance(a, (float, np.floating)):
            cargs.append(ctypes.c_double(float(a)))
        else:
            raise TypeError(f"{name}: unsupported argument {type(a)}")
    return int(fn(*cargs))


class IRError(RuntimeError):
    """Stands for an MPAS_LOG_CRIT abort of the reference."""


def seaice_init_advection(domain):
    """seaice_init_advection_incremental_remap :165-816"""
    cfg = domain.configs
    # check_halo_layer_number :829-874 (two layers on hexagonal meshes, three on quadrilateral ones); as in the device
    # mirror it only matters where there are halos at all
    need = 3 if int(domain.blocklist[0].dimensions["vertexDegree"]) == 4 else 2
    if int(cfg.get("config_num_halos", need)) < need and len(domain.blocklist) > 1:
        raise IRError("MPAS-seaice: IR too few halo layers for incremental remapping")
    trc.seaice_add_tracers_to_linked_list(domain)
    for b in domain.blocklist:
        d, m = b.dimensions, b.structs["mesh"]
        if d["vertexDegree"] not in (3, 4):
            raise IRError("IR: vertexDegree must be 3 (hexagonal meshes) or 4 (quadrilateral meshes)")
        nC, nV, nE, M = d["nCells"], d["nVertices"], d["nEdges"], d["maxEdges"]
        sphere = bool(m["on_a_sphere"])
        rot = bool(cfg["config_rotate_cartesian_grid"]) and sphere
        ir = b.structs["incremental_remap"]
        xyz = {}
        for loc, n in (("Cell", nC), ("Vertex", nV), ("Edge", nE)):
            x, y, z = m["x" + loc], m["y" + loc], m["z" + loc]
            if rot:
                xr, yr, zr = np.zeros_like(x), np.zeros_like(y), np.zeros_like(z)
                call("ira_rotate_global_vectors", x, y, z, xr, yr, zr, n)
                x, y, z = xr, yr, zr
            xyz[loc] = (x, y, z)
        for loc, n in (("Cell", nC), ("Vertex", nV), ("Edge", nE)):
            ir[f"trans{loc}ToGlobal"] = np.zeros((3, 3, n + 1), order="F")
            ir[f"transGlobalTo{loc}"] = np.zeros((3, 3, n + 1), order="F")
            if sphere:
                call("ira_define_transformations", *xyz[loc], n, ir[f"trans{loc}ToGlobal"], ir[f"transGlobalTo{loc}"])
        ir["xVertexOnCell"] = np.zeros((M, nC + 1), order="F")
        ir["yVertexOnCell"] = np.zeros((M, nC + 1), order="F")
        bad = calli("ira_vertex_on_cell", nC, M, m["nEdgesOnCell"], m["verticesOnCell"], *xyz["Vertex"], *xyz["Cell"],
                    sphere, ir["transGlobalToCell"], ir["xVertexOnCell"], ir["yVertexOnCell"])
        if bad:
            raise IRError("IR requires convex cells with vertices arranged CCW around the cell")
        ir["remapEdge"] = np.zeros(nE + 1, dtype=np.int32)
        ir["cellsOnEdgeRemap"] = np.zeros((NEER, nE + 1), dtype=np.int32, order="F")
        ir["edgesOnEdgeRemap"] = np.zeros((NEER, nE + 1), dtype=np.int32, order="F")
        ir["xVertexOnEdge"] = np.zeros((NVER, nE + 1), order="F")
        ir["yVertexOnEdge"] = np.zeros((NVER, nE + 1), order="F")
        ir["minLengthEdgesOnVertex"] = np.zeros(nV + 1)
        bad = calli("ira_get_geometry", nE, nV, nC, d["nCellsSolve"], M, m["nEdgesOnCell"], *xyz["Cell"], *xyz["Vertex"],
                    *xyz["Edge"], m["cellsOnEdge"], m["verticesOnEdge"], m["edgesOnCell"], m["edgesOnVertex"],
                    int(d["vertexDegree"]), sphere, ir["transGlobalToEdge"], ir["remapEdge"], ir["cellsOnEdgeRemap"], ir["edgesOnEdgeRemap"],
                    ir["xVertexOnEdge"], ir["yVertexOnEdge"], ir["minLengthEdgesOnVertex"])
        if bad:
            raise IRError("cell 1 is not in the left half-plane")
        ir["geomAvgCell"] = np.zeros((nC + 1, 14), order="F")
        call("ira_geometric_cell_averages", ir["xVertexOnCell"], ir["yVertexOnCell"], nC, M, m["edgesOnCell"],
             m["nEdgesOnCell"], m["dcEdge"], m["dvEdge"], ir["geomAvgCell"])
        if (m["areaCell"][:d["nCellsSolve"]] <= 0.0).any():
            raise IRError("IR: Bad cell area")
    # coeffs_reconstruct comes with the mesh; the reference exchanges its halo here (:770)


class _Work:
    """Per-tracer scratch of tracer_type (:41-109) for one block."""

    def __init__(self, t, nCat, nC, nE):
        L = t.nLayers

        def rc():
            return np.zeros((L, nCat, nC + 1), order="F")
        self.mask = np.zeros((L, nCat, nC + 1), dtype=np.int32, order="F")
        self.center, self.xGrad, self.yGrad = rc(), rc(), rc()
        self.xBary, self.yBary = rc(), rc()
        self.prod = rc()
        self.edgeFlux = np.zeros((L, nCat, nE + 1), order="F")
        self.triVal = np.zeros((L, nCat, NQP, NTRI, nE + 1), order="F")


def seaice_run_advection(domain, updateHaloInit=True, updateHaloFinal=False, diagnostics=None):
    """seaice_run_advection_incremental_remap :2338-2667"""
    tracersHead = domain.solver_state["tracersHead"]
    halo = getattr(domain, "halo", None)
    if updateHaloInit and halo is not None and domain.configs["config_use_halo_exch"]:
        halo(domain, "velocity_solver", ["uVelocity", "vVelocity"])
        halo(domain, "tracers", [t.tracerName for t in tracersHead])
    for b in domain.blocklist:
        _convert(b, tracersHead, "ira_volume_to_thickness")
    for b in domain.blocklist:
        dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
        incremental_remap_block(domain, b, dt, tracersHead, diagnostics)
    if domain.configs["config_conservation_check"]:
        for t in tracersHead:
            init = sum(b._ir_sums[t.tracerName][0] for b in domain.blocklist)
            fin = sum(b._ir_sums[t.tracerName][1] for b in domain.blocklist)
            big = np.abs(init) > 1.0e-11
            ratio = np.where(big, (fin - init) / np.where(big, init, 1.0), 0.0)
            if (np.abs(ratio) > 1.0e-11).any():
                raise IRError("IR: Tracer conservation error: tracer = " + t.tracerName)
    if domain.configs["config_monotonicity_check"]:
        check_tracer_monotonicity(domain, tracersHead)
    for b in domain.blocklist:
        _convert(b, tracersHead, "ira_thickness_to_volume")
    if updateHaloFinal and halo is not None:
        halo(domain, "tracers", [t.tracerName for t in tracersHead])


def check_tracer_monotonicity(domain, tracersHead):
    """check_tracer_monotonicity :8416-8749 (thicknesses, not volumes, are compared)"""
    halo = getattr(domain, "halo", None)
    for t in tracersHead:
        if t.nParents == 0:
            continue  # monotonicity holds for tracers but not for the mass-like field
        if halo is not None:
            halo(domain, "tracer_monotonicity", [t.tracerName + "LocalMin", t.tracerName + "LocalMax"])
        for b in domain.blocklist:
            d, m = b.dimensions, b.structs["mesh"]
            mono = b.structs["tracer_monotonicity"]
            bad = calli("ira_check_monotonicity", d["nCells"], d["nCellsSolve"], d["maxEdges"], t.nLayers,
                        d["nCategories"], m["nEdgesOnCell"], m["cellsOnCell"], b.structs["tracers"][t.tracerName],
                        b._ir_masks[t.tracerName], mono[t.tracerName + "LocalMin"], mono[t.tracerName + "LocalMax"])
            if bad:
                raise IRError("IR advection, monotonicity violation: tracer = " + t.tracerName)


def _convert(b, tracersHead, routine):
    """:2502-2544 and :2608-2650"""
    tp = b.structs["tracers"]
    nCat = b.dimensions["nCategories"]
    work = None
    for t in tracersHead:
        if t.tracerName == "iceAreaCategory":
            work = tp[t.tracerName].copy(order="F")
        elif t.tracerName in ("iceVolumeCategory", "snowVolumeCategory"):
            call(routine, b.dimensions["nCells"] + 1, nCat, work, tp[t.tracerName])


def incremental_remap_block(domain, b, dt, tracersHead, diagnostics=None):
    """incremental_remap_block :2740-3389"""
    cfg = domain.configs
    d, m, ir, tp = b.dimensions, b.structs["mesh"], b.structs["incremental_remap"], b.structs["tracers"]
    vs = b.structs["velocity_solver"]
    nC, nCs, nV, nE, M, nCat = d["nCells"], d["nCellsSolve"], d["nVertices"], d["nEdges"], d["maxEdges"], d["nCategories"]
    sphere = bool(m["on_a_sphere"])
    rot = bool(cfg["config_rotate_cartesian_grid"])
    W = {t.tracerName: _Work(t, nCat, nC, nE) for t in tracersHead}
    maskCell = np.zeros(nC + 1, dtype=np.int32)

    # local mins and maxes of the old values for the monotonicity check :2999-3021
    if cfg["config_monotonicity_check"]:
        mono = b.structs.setdefault("tracer_monotonicity", {})
        for t in tracersHead:
            w = W[t.tracerName]
            call_masks(nC, t, nCat, tp, t.parent, 0.0, w.mask, maskCell)
            lo = np.zeros((t.nLayers, nCat, nC + 1), order="F")
            hi = np.zeros((t.nLayers, nCat, nC + 1), order="F")
            call("ira_local_min_max", nC, nCs, M, t.nLayers, nCat, m["nEdgesOnCell"], m["cellsOnCell"],
                 tp[t.tracerName], w.mask, lo, hi)
            mono[t.tracerName + "LocalMin"], mono[t.tracerName + "LocalMax"] = lo, hi

    # make_masks :3023-3029
    for t in tracersHead:
        w = W[t.tracerName]
        if t.parent is None:
            call_masks(nC, t, nCat, tp, None, 1.0e-11, w.mask, maskCell)
        else:
            call_masks(nC, t, nCat, tp, t.parent, 1.0e-11, w.mask, maskCell)

    # construct_linear_tracer_fields :3580-4189
    geom = ir["geomAvgCell"]
    xB0 = np.zeros((1, nCat, nC + 1), order="F")
    yB0 = np.zeros((1, nCat, nC + 1), order="F")
    call("ira_centroid_barycenter", nC, nCat, geom, xB0, yB0)
    for t in tracersHead:
        w = W[t.tracerName]
        L = t.nLayers
        arr = tp[t.tracerName]
        call("ira_compute_gradient", nC, M, L, nCat, m["nEdgesOnCell"], m["edgesOnCell"], m["cellsOnCell"],
             m["cellsOnEdge"], m["dcEdge"], m["coeffs_reconstruct"], sphere, rot, ir["transGlobalToCell"], maskCell,
             arr, w.mask, w.xGrad, w.yGrad)
        if t.parent is None:
            xb, yb, Lb = xB0, yB0, 1
        else:
            pw = W[t.parent.tracerName]
            xb, yb, Lb = pw.xBary, pw.yBary, t.parent.nLayers
        call("ira_limit_gradient", nC, M, L, nCat, Lb, m["nEdgesOnCell"], m["cellsOnCell"], ir["xVertexOnCell"],
             ir["yVertexOnCell"], maskCell, arr, w.mask, xb, yb, w.xGrad, w.yGrad)
        call("ira_center_value", nC, L, nCat, Lb, arr, w.xGrad, w.yGrad, xb, yb, w.center)
        if t.hasChild:
            if t.nParents == 3:
                raise IRError("IR, construct_linear_tracer_fields: Tracers with 3 parents should not have children")
            chain = []
            p = t
            while p is not None:
                chain.insert(0, p)
                p = p.parent
            _barycenter(nC, L, nCat, geom, maskCell, chain, tp, W, w)

    # find_departure_points :5255-5349
    ir["departurePoint"] = np.zeros((2, nV + 1), order="F")
    warn = calli("ira_departure_points", nV, dt, vs["uVelocity"], vs["vVelocity"], ir["minLengthEdgesOnVertex"],
                 ir["departurePoint"])
    # find_departure_triangles :5365-6254, get_triangle_quadrature_points :6546-6652
    ir["maskEdge"] = np.zeros(nE + 1, dtype=np.int32)
    ir["xTriangle"] = np.zeros((NQP, NTRI, nE + 1), order="F")
    ir["yTriangle"] = np.zeros((NQP, NTRI, nE + 1), order="F")
    ir["iCellTriangle"] = np.zeros((NTRI, nE + 1), dtype=np.int32, order="F")
    ir["triangleArea"] = np.zeros((NTRI, nE + 1), order="F")
    errs = calli("ira_departure_triangles", nE, nC, M, m["nEdgesOnCell"], ir["xVertexOnEdge"], ir["yVertexOnEdge"],
                 ir["xVertexOnCell"], ir["yVertexOnCell"], m["verticesOnCell"], m["verticesOnEdge"], ir["remapEdge"],
                 ir["cellsOnEdgeRemap"], ir["edgesOnEdgeRemap"], int(d["vertexDegree"]), ir["departurePoint"], sphere,
                 ir["maskEdge"],
                 ir["xTriangle"], ir["yTriangle"], ir["iCellTriangle"], ir["triangleArea"])
    if errs:
        raise IRError("IR: basis vectors for IR coordinate transformations must not be parallel")
    if diagnostics is not None:
        diagnostics["xTriangleVertices"] = ir["xTriangle"].copy()
        diagnostics["yTriangleVertices"] = ir["yTriangle"].copy()
        diagnostics["cflWarnings"] = warn
    call("ira_quadrature_points", nE, ir["maskEdge"], ir["xTriangle"], ir["yTriangle"])

    # integrate_fluxes_over_triangles :6667-6967
    for t in tracersHead:
        w = W[t.tracerName]
        pv = W[t.parent.tracerName].triVal if t.parent is not None else None
        Lp = t.parent.nLayers if t.parent is not None else 1
        neg = calli("ira_integrate_fluxes", nE, t.nLayers, nCat, Lp, ir["maskEdge"], ir["xTriangle"], ir["yTriangle"],
                    ir["triangleArea"], ir["iCellTriangle"], w.center, w.xGrad, w.yGrad, pv, w.triVal, w.edgeFlux,
                    t.tracerName == "iceAreaCategory")
        if neg:
            raise IRError("MPAS-seaice: IR negative reconstructed ice area")

    # compute_mass_tracer_products :6982-7109
    for t in tracersHead:
        w = W[t.tracerName]
        pp = W[t.parent.tracerName].prod if t.parent is not None else None
        Lp = t.parent.nLayers if t.parent is not None else 1
        calli("ira_mass_tracer_product", nC, t.nLayers, nCat, Lp, pp, tp[t.tracerName], w.prod)
    # recover_tracer_means :3233-3252 (a debugging option: the reference only logs warnings)
    if cfg.get("config_recover_tracer_means_check", False):
        total = 0
        for t in tracersHead:
            chain = []
            p = t
            while p is not None:
                chain.insert(0, p)
                p = p.parent
            rec = np.zeros((t.nLayers, nCat, nC + 1), order="F")
            total += _recover_means(nC, M, t.nLayers, nCat, m, ir, chain, W, W[t.tracerName].prod, rec)
            if diagnostics is not None:
                diagnostics[t.tracerName + "RecoveredProduct"] = rec
        b._ir_recover_warnings = total
        if diagnostics is not None:
            diagnostics["recoverMeansWarnings"] = total
    sums = {}
    if cfg["config_conservation_check"]:
        for t in tracersHead:
            s0 = np.zeros((t.nLayers, nCat), order="F")
            call("ira_sum_tracer", nCs, t.nLayers, nCat, m["areaCell"], W[t.tracerName].prod, s0)
            sums[t.tracerName] = [s0, None]

    # update_mass_and_tracers :7125-7523
    for t in tracersHead:
        w = W[t.tracerName]
        pp = W[t.parent.tracerName].prod if t.parent is not None else None
        Lp = t.parent.nLayers if t.parent is not None else 1
        assert t.nLayers * nCat <= 512
        bad = calli("ira_update_tracer", nCs, M, t.nLayers, nCat, Lp, m["nEdgesOnCell"], m["edgesOnCell"],
                    m["cellsOnEdge"], m["areaCell"], w.edgeFlux, pp, w.prod, tp[t.tracerName])
        if bad:
            raise IRError("IR: Negative mass in IR")
    if cfg["config_conservation_check"]:
        for t in tracersHead:
            w = W[t.tracerName]
            pp = W[t.parent.tracerName].prod if t.parent is not None else None
            Lp = t.parent.nLayers if t.parent is not None else 1
            calli("ira_mass_tracer_product", nC, t.nLayers, nCat, Lp, pp, tp[t.tracerName], w.prod)
            s1 = np.zeros((t.nLayers, nCat), order="F")
            call("ira_sum_tracer", nCs, t.nLayers, nCat, m["areaCell"], w.prod, s1)
            sums[t.tracerName][1] = s1
    b._ir_sums = sums

    # zap_small_mass :8764-8886
    maskCat = np.zeros((nCat, nC + 1), dtype=np.int32, order="F")
    call("ira_zap_small_mass", nCs, nCat, tp[tracersHead[0].tracerName], maskCat, nC + 1)
    for t in tracersHead[1:]:
        call("ira_zap_tracer", nCs, t.nLayers, nCat, maskCat, tp[t.tracerName])
    if diagnostics is not None:
        diagnostics["work"] = W
        diagnostics["maskCell"] = maskCell
    b._ir_masks = {t.tracerName: W[t.tracerName].mask for t in tracersHead}  # from the old values, threshold 1e-11


def call_masks(nC, t, nCat, tp, parent, threshold, mask, maskCell):
    Lp = parent.nLayers if parent is not None else 1
    parr = tp[parent.tracerName] if parent is not None else None
    calli("ira_make_masks", nC, t.nLayers, nCat, Lp, tp[t.tracerName], parr, float(threshold), mask, maskCell)


def _recover_means(nC, M, L, nCat, m, ir, chain, W, prod, rec):
    n = len(chain)
    Lk = np.array([p.nLayers for p in chain], dtype=np.int32)
    PtrArr = ctypes.c_void_p * n

    def ptrs(get):
        return PtrArr(*[get(W[p.tracerName]).ctypes.data for p in chain])

    def vp(a):
        return ctypes.c_void_p(a.ctypes.data)
    fn = evp_flow.lib().ira_recover_tracer_means
    fn.restype = ctypes.c_int
    return int(fn(ctypes.c_int(nC), ctypes.c_int(M), ctypes.c_int(L), ctypes.c_int(nCat), vp(m["nEdgesOnCell"]),
                  vp(m["edgesOnCell"]), vp(ir["xVertexOnCell"]), vp(ir["yVertexOnCell"]), vp(m["dcEdge"]),
                  vp(m["dvEdge"]), ctypes.c_int(n), vp(Lk), ptrs(lambda w: w.center), ptrs(lambda w: w.xGrad),
                  ptrs(lambda w: w.yGrad), vp(prod), vp(rec)))


def _barycenter(nC, L, nCat, geom, maskCell, chain, tp, W, w):
    n = len(chain)
    Lk = np.array([p.nLayers for p in chain], dtype=np.int32)
    PtrArr = ctypes.c_void_p * n

    def ptrs(get):
        return PtrArr(*[get(p).ctypes.data for p in chain])
    fn = evp_flow.lib().ira_barycenter
    fn.restype = None
    fn(ctypes.c_int(nC), ctypes.c_int(L), ctypes.c_int(nCat), ctypes.c_void_p(geom.ctypes.data),
       ctypes.c_void_p(maskCell.ctypes.data), ctypes.c_int(n), ctypes.c_void_p(Lk.ctypes.data),
       ptrs(lambda p: tp[p.tracerName]), ptrs(lambda p: W[p.tracerName].center),
       ptrs(lambda p: W[p.tracerName].xGrad), ptrs(lambda p: W[p.tracerName].yGrad),
       ctypes.c_void_p(w.xBary.ctypes.data), ctypes.c_void_p(w.yBary.ctypes.data))
