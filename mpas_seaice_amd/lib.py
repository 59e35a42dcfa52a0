"""ctypes binding of libseaice_hip.so (the C ABI declared in include/seaice_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised (the reference treats the same conditions as MPAS_LOG_CRIT aborts).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SIH_LIB_PATH") or os.path.join(_HERE, "libseaice_hip.so")  # SIH_LIB_PATH: another build
_lib = None


class SeaiceHipError(RuntimeError):
    pass


class MeshDesc(ctypes.Structure):
    _fields_ = [
        ("nCells", ctypes.c_int32), ("nCellsSolve", ctypes.c_int32),
        ("nVertices", ctypes.c_int32), ("nVerticesSolve", ctypes.c_int32),
        ("nEdges", ctypes.c_int32), ("nEdgesSolve", ctypes.c_int32),
        ("maxEdges", ctypes.c_int32), ("vertexDegree", ctypes.c_int32),
        ("onASphere", ctypes.c_int32), ("sphereRadius", ctypes.c_double),
        ("nEdgesOnCell", ctypes.c_void_p), ("verticesOnCell", ctypes.c_void_p),
        ("cellsOnCell", ctypes.c_void_p), ("edgesOnCell", ctypes.c_void_p),
        ("cellsOnVertex", ctypes.c_void_p), ("cellsOnEdge", ctypes.c_void_p),
        ("verticesOnEdge", ctypes.c_void_p),
        ("xCell", ctypes.c_void_p), ("yCell", ctypes.c_void_p), ("zCell", ctypes.c_void_p),
        ("xVertex", ctypes.c_void_p), ("yVertex", ctypes.c_void_p), ("zVertex", ctypes.c_void_p),
        ("areaCell", ctypes.c_void_p), ("areaTriangle", ctypes.c_void_p), ("fVertex", ctypes.c_void_p),
        ("dvEdge", ctypes.c_void_p), ("dcEdge", ctypes.c_void_p),
        ("renumber", ctypes.c_int32),
    ]


class EvpConfig(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_double),
        ("dynamicsSubcycleNumber", ctypes.c_int32), ("elasticSubcycleNumber", ctypes.c_int32),
        ("constitutiveRelation", ctypes.c_int32), ("variationalBasis", ctypes.c_int32),
        ("variationalDenominator", ctypes.c_int32), ("wachspressIntegrationOrder", ctypes.c_int32),
        ("rotateCartesianGrid", ctypes.c_int32), ("includeMetricTerms", ctypes.c_int32),
        ("useAirStress", ctypes.c_int32), ("useOceanStress", ctypes.c_int32),
        ("oceanStressType", ctypes.c_int32), ("surfaceTilt", ctypes.c_int32),
        ("calcVelocityMasks", ctypes.c_int32), ("useColumnPackage", ctypes.c_int32),
        ("nCategories", ctypes.c_int32), ("minEdgeLengthGlobal", ctypes.c_double),
        ("computeHaloVertices", ctypes.c_int32), ("averageVariationalStrain", ctypes.c_int32),
        ("useSpecialBoundariesVelocity", ctypes.c_int32), ("useSpecialBoundariesVelocityMasks", ctypes.c_int32),
        ("strainScheme", ctypes.c_int32), ("stressDivergenceScheme", ctypes.c_int32),
        ("useColumnVerticalThermodynamics", ctypes.c_int32), ("iceStrengthFormulation", ctypes.c_int32),
        ("ridgingParticipation", ctypes.c_int32), ("ridgingRedistribution", ctypes.c_int32),
        ("ridgingEfoldingScale", ctypes.c_double), ("ridgingWorkRatio", ctypes.c_double),
        ("wachspressIntegrationType", ctypes.c_int32),
    ]


class IrDesc(ctypes.Structure):
    _fields_ = [
        ("edgesOnCell", ctypes.c_void_p), ("cellsOnEdge", ctypes.c_void_p), ("verticesOnEdge", ctypes.c_void_p),
        ("edgesOnVertex", ctypes.c_void_p),
        ("xEdge", ctypes.c_void_p), ("yEdge", ctypes.c_void_p), ("zEdge", ctypes.c_void_p),
        ("dcEdge", ctypes.c_void_p), ("dvEdge", ctypes.c_void_p), ("coeffsReconstruct", ctypes.c_void_p),
        ("nCategories", ctypes.c_int32), ("rotateCartesianGrid", ctypes.c_int32),
    ]


class RidgeConfig(ctypes.Structure):
    _fields_ = [
        ("dt", ctypes.c_double), ("dynamicsSubcycleNumber", ctypes.c_int32), ("participation", ctypes.c_int32),
        ("redistribution", ctypes.c_int32), ("efoldingScale", ctypes.c_double), ("nCategories", ctypes.c_int32),
        ("categoryThicknessLimits", ctypes.c_void_p), ("itdCleanup", ctypes.c_int32),
    ]


EXPORTS = [
    "sih_block_create", "sih_block_destroy", "sih_last_error", "sih_set_stream", "sih_sync",
    "sih_evp_init", "sih_evp_static_changed", "sih_field_put", "sih_field_get", "sih_field_count",
    "sih_evp_pre_subcycle", "sih_evp_subcycle", "sih_evp_post_subcycle", "sih_evp_run",
    "sih_evp_subcycle_timed", "sih_evp_subcycle_graph", "sih_halo_set_lists", "sih_halo_pack", "sih_halo_unpack", "sih_get_info",
    "sih_ir_init", "sih_ir_set_tracers", "sih_ir_run", "sih_ir_get_sums", "sih_ir_run_timed",
    "sih_fields_put", "sih_fields_get", "sih_evp_overlap_setup", "sih_evp_subcycle_parts",
    "sih_evp_set_special_boundaries", "sih_ir_check_monotonicity", "sih_evp_set_edge_mesh",
    "sih_ir_unpack_reconstruction", "sih_host_alloc", "sih_host_free", "sih_upwind_init", "sih_upwind_prepare", "sih_upwind_advect", "sih_upwind_run", "sih_upwind_get_sums",
    "sih_evp_defer_special_boundaries", "sih_evp_apply_special_boundaries",
    "sih_evp_strain_tensor", "sih_evp_stress_divergence",
    "sih_comm_handle_size", "sih_comm_create", "sih_comm_get_handle", "sih_comm_connect", "sih_comm_destroy",
    "sih_halo_exchange", "sih_halo_exchange_start", "sih_halo_exchange_finish", "sih_comm_set_timeout",
    "sih_comm_status", "sih_evp_subcycle_exchange", "sih_field_get_noalloc", "sih_device_count", "sih_evp_subcycle_persistent", "sih_evp_subcycle_fused", "sih_column_get", "sih_column_put", "sih_evp_set_persistent",
    "sih_fields_get_start", "sih_fields_get_finish", "sih_evp_run_early_outputs",
    "sih_column_ridge", "sih_column_ridge_init", "sih_column_update_state", "sih_column_reinitialize_diagnostics",
]


def load():
    """Load the shared library (no GPU needed for loading / symbol lookup)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeaiceHipError(
            f"{LIB_PATH} is missing: build it with `python -m mpas_seaice_amd.build` "
            "(there is no CPU fallback for the solver)")
    # PyTorch wheels bundle their own HIP runtime under the same SONAME (libamdhip64.so.7) but load it by
    # another file name.  Two HIP runtimes in one process cannot both own the GPU, so when torch is
    # part of the process (device buffers, torch.distributed) it must be loaded first: this library
    # then binds to the runtime that is already there.  A host without torch uses /opt/rocm's.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.sih_last_error.restype = ctypes.c_char_p
    lib.sih_device_count.argtypes = []
    for fn in (lib.sih_column_get, lib.sih_column_put):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                       ctypes.c_void_p, ctypes.c_size_t]
    lib.sih_field_count.restype = ctypes.c_size_t
    lib.sih_field_count.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    lib.sih_block_create.argtypes = [ctypes.POINTER(MeshDesc), ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.sih_block_destroy.argtypes = [ctypes.c_void_p]
    lib.sih_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sih_sync.argtypes = [ctypes.c_void_p]
    lib.sih_evp_init.argtypes = [ctypes.c_void_p, ctypes.POINTER(EvpConfig)]
    lib.sih_evp_static_changed.argtypes = [ctypes.c_void_p]
    lib.sih_field_put.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.sih_field_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.sih_field_get_noalloc.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.sih_fields_get_finish.argtypes = [ctypes.c_void_p]
    lib.sih_column_ridge.argtypes = [ctypes.c_void_p, ctypes.POINTER(RidgeConfig)]
    lib.sih_column_ridge_init.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_column_update_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    lib.sih_column_reinitialize_diagnostics.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for fn in (lib.sih_fields_put, lib.sih_fields_get, lib.sih_fields_get_start, lib.sih_evp_run_early_outputs):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                       ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)]
    lib.sih_evp_pre_subcycle.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_subcycle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.sih_evp_post_subcycle.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_subcycle_graph.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_subcycle_persistent.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_subcycle_fused.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_set_persistent.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_set_special_boundaries.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
    lib.sih_evp_overlap_setup.argtypes = [ctypes.c_void_p]
    lib.sih_evp_defer_special_boundaries.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_evp_apply_special_boundaries.argtypes = [ctypes.c_void_p]
    lib.sih_evp_subcycle_parts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int32),
                                           ctypes.c_int]
    lib.sih_evp_run.argtypes = [ctypes.c_void_p]
    lib.sih_evp_strain_tensor.argtypes = [ctypes.c_void_p]
    lib.sih_evp_stress_divergence.argtypes = [ctypes.c_void_p]
    lib.sih_evp_subcycle_timed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                           ctypes.POINTER(ctypes.c_double)]
    lib.sih_halo_set_lists.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                       ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.sih_halo_pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
    lib.sih_halo_unpack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
    lib.sih_get_info.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
    lib.sih_comm_handle_size.restype = ctypes.c_size_t
    lib.sih_comm_handle_size.argtypes = []
    lib.sih_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sih_comm_get_handle.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sih_comm_connect.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sih_comm_destroy.argtypes = [ctypes.c_void_p]
    for fn in (lib.sih_halo_exchange, lib.sih_halo_exchange_start, lib.sih_halo_exchange_finish):
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    lib.sih_comm_set_timeout.argtypes = [ctypes.c_void_p, ctypes.c_double]
    lib.sih_comm_status.argtypes = [ctypes.c_void_p]
    lib.sih_evp_subcycle_exchange.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
    lib.sih_ir_init.argtypes = [ctypes.c_void_p, ctypes.POINTER(IrDesc)]
    lib.sih_evp_set_edge_mesh.argtypes = [ctypes.c_void_p, ctypes.POINTER(IrDesc)]
    lib.sih_ir_set_tracers.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                       ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p]
    lib.sih_ir_run.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    lib.sih_ir_check_monotonicity.argtypes = [ctypes.c_void_p]
    lib.sih_ir_unpack_reconstruction.argtypes = [ctypes.c_void_p]
    lib.sih_ir_get_sums.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.sih_ir_run_timed.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                                     ctypes.c_void_p]
    lib.sih_host_alloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    lib.sih_host_free.argtypes = [ctypes.c_void_p]
    lib.sih_upwind_init.argtypes = [ctypes.c_void_p, ctypes.POINTER(IrDesc)]
    lib.sih_upwind_prepare.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.sih_upwind_advect.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    lib.sih_upwind_run.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    lib.sih_upwind_get_sums.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_void_p]
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sih_last_error()
        raise SeaiceHipError(f"{what}: error {rc}: {msg.decode() if msg else ''}")


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


def pinned_empty(shape, dtype=np.float64, order="F"):
    """A numpy array in page-locked host memory (sih_host_alloc); freed with the array."""
    import weakref
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    p = ctypes.c_void_p()
    check(load().sih_host_alloc(ctypes.byref(p), max(n, 1) * dt.itemsize), "sih_host_alloc")
    buf = (ctypes.c_char * (max(n, 1) * dt.itemsize)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape, order=order)
    weakref.finalize(buf, load().sih_host_free, p)  # the array keeps `buf` alive through its base
    return a


class BlockContext:
    """One MPAS block resident on one GPU (wraps a ``sih_ctx*``)."""

    def __init__(self, mesh_pool, dims, device=0, renumber=True):
        lib = load()
        self._lib = lib
        self._keep = []
        d = MeshDesc()
        for k in ("nCells", "nCellsSolve", "nVertices", "nVerticesSolve", "nEdges", "nEdgesSolve", "maxEdges",
                  "vertexDegree"):
            setattr(d, k, int(dims[k]))
        d.onASphere = int(bool(mesh_pool["on_a_sphere"]))
        d.sphereRadius = float(mesh_pool["sphere_radius"])
        for k in ("nEdgesOnCell", "verticesOnCell", "cellsOnCell", "edgesOnCell", "cellsOnVertex", "cellsOnEdge",
                  "verticesOnEdge"):
            a = mesh_pool.get(k)
            if a is not None:
                assert a.dtype == np.int32 and a.flags["F_CONTIGUOUS"], k
                self._keep.append(a)
            setattr(d, k, _ptr(a))
        for k in ("xCell", "yCell", "zCell", "xVertex", "yVertex", "zVertex", "areaCell", "areaTriangle", "fVertex",
                  "dvEdge", "dcEdge"):
            a = mesh_pool.get(k)
            if a is not None:
                assert a.dtype == np.float64, k
            setattr(d, k, _ptr(a))
        d.renumber = int(bool(renumber))
        self.dims = dict(dims)
        h = ctypes.c_void_p()
        check(lib.sih_block_create(ctypes.byref(d), int(device), ctypes.byref(h)), "sih_block_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.sih_block_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- thin wrappers -------------------------------------------------------------------------
    def evp_init(self, cfg: EvpConfig):
        check(self._lib.sih_evp_init(self.handle, ctypes.byref(cfg)), "sih_evp_init")

    def static_changed(self):
        check(self._lib.sih_evp_static_changed(self.handle), "sih_evp_static_changed")

    def field_count(self, name):
        es = ctypes.c_int(0)
        n = self._lib.sih_field_count(self.handle, name.encode(), ctypes.byref(es))
        return int(n), int(es.value)

    def put(self, name, arr):
        n, es = self.field_count(name)
        if n == 0:
            raise SeaiceHipError(f"unknown field {name}")
        assert arr.dtype == (np.int32 if es == 4 else np.float64), (name, arr.dtype)
        assert arr.flags["F_CONTIGUOUS"] or arr.flags["C_CONTIGUOUS"], name
        assert arr.size >= n, (name, arr.size, n)
        check(self._lib.sih_field_put(self.handle, name.encode(), _ptr(arr), n), f"sih_field_put({name})")

    def get(self, name, arr):
        n, es = self.field_count(name)
        if n == 0:
            raise SeaiceHipError(f"unknown field {name}")
        assert arr.dtype == (np.int32 if es == 4 else np.float64), (name, arr.dtype)
        assert arr.flags["F_CONTIGUOUS"] or arr.flags["C_CONTIGUOUS"], name
        assert arr.size >= n, (name, arr.size, n)
        check(self._lib.sih_field_get(self.handle, name.encode(), _ptr(arr), n), f"sih_field_get({name})")

    def _many(self, fn, what, items):
        """items: iterable of (name, array); one library call for all of them."""
        items = list(items)
        if not items:
            return
        k = len(items)
        names = (ctypes.c_char_p * k)()
        hosts = (ctypes.c_void_p * k)()
        counts = (ctypes.c_size_t * k)()
        for i, (name, arr) in enumerate(items):
            n, es = self.field_count(name)
            if n == 0:
                raise SeaiceHipError(f"unknown field {name}")
            assert arr.dtype == (np.int32 if es == 4 else np.float64), (name, arr.dtype)
            assert arr.flags["F_CONTIGUOUS"] or arr.flags["C_CONTIGUOUS"], name
            assert arr.size >= n, (name, arr.size, n)
            names[i], hosts[i], counts[i] = name.encode(), arr.ctypes.data, n
        check(fn(self.handle, k, names, hosts, counts), what)

    def put_many(self, items):
        self._many(self._lib.sih_fields_put, "sih_fields_put", items)

    def get_many(self, items):
        self._many(self._lib.sih_fields_get, "sih_fields_get", items)

    def get_many_start(self, items):
        """the download drains behind whatever is launched next; the arrays are the library's until get_many_finish()"""
        self._pending_get = [a for _, a in items]  # keep the arrays alive
        self._many(self._lib.sih_fields_get_start, "sih_fields_get_start", items)

    def evp_run_early_outputs(self, items):
        """sih_evp_run with the download of `items` (fields final after the pre-cycle) behind the iterations"""
        self._pending_get = [a for _, a in items]
        self._many(self._lib.sih_evp_run_early_outputs, "sih_evp_run_early_outputs", items)

    def get_many_finish(self):
        check(self._lib.sih_fields_get_finish(self.handle), "sih_fields_get_finish")
        self._pending_get = None

    def pre_subcycle(self, stage=-1):
        check(self._lib.sih_evp_pre_subcycle(self.handle, stage), "sih_evp_pre_subcycle")

    def subcycle(self, n, last=False):
        check(self._lib.sih_evp_subcycle(self.handle, int(n), int(bool(last))), "sih_evp_subcycle")

    def subcycle_graph(self, n):
        check(self._lib.sih_evp_subcycle_graph(self.handle, int(n)), "sih_evp_subcycle_graph")

    def set_persistent(self, mode):
        """0 two-kernel form, 1 persistent kernel, 2 persistent kernel with the halo exchange inside, -1 environment"""
        check(self._lib.sih_evp_set_persistent(self.handle, int(mode)), "sih_evp_set_persistent")

    def subcycle_persistent(self, n):
        check(self._lib.sih_evp_subcycle_persistent(self.handle, int(n)), "sih_evp_subcycle_persistent")

    def subcycle_fused(self, n):
        check(self._lib.sih_evp_subcycle_fused(self.handle, int(n)), "sih_evp_subcycle_fused")

    def set_special_boundaries(self, vertexBoundaryType=None, vertexBoundarySourceLocal=None,
                               solveVelocitySpecialBoundaries=None, solveStressSpecialBoundaries=None):
        arrs = (vertexBoundaryType, vertexBoundarySourceLocal, solveVelocitySpecialBoundaries,
                solveStressSpecialBoundaries)
        for a in arrs:
            assert a is None or (a.dtype == np.int32 and a.ndim == 1), "int32 vectors expected"
        check(self._lib.sih_evp_set_special_boundaries(self.handle, *[_ptr(a) for a in arrs]),
              "sih_evp_set_special_boundaries")

    def defer_special_boundaries(self, on=True):
        check(self._lib.sih_evp_defer_special_boundaries(self.handle, int(bool(on))),
              "sih_evp_defer_special_boundaries")

    def apply_special_boundaries(self):
        check(self._lib.sih_evp_apply_special_boundaries(self.handle), "sih_evp_apply_special_boundaries")

    def overlap_setup(self):
        check(self._lib.sih_evp_overlap_setup(self.handle), "sih_evp_overlap_setup")

    def subcycle_parts(self, parts, use_graph=False):
        arr = (ctypes.c_int32 * len(parts))(*parts)
        check(self._lib.sih_evp_subcycle_parts(self.handle, len(parts), arr, int(bool(use_graph))),
              "sih_evp_subcycle_parts")

    def strain_tensor(self):
        check(self._lib.sih_evp_strain_tensor(self.handle), "sih_evp_strain_tensor")

    def stress_divergence(self):
        check(self._lib.sih_evp_stress_divergence(self.handle), "sih_evp_stress_divergence")

    def post_subcycle(self, stage=-1):
        check(self._lib.sih_evp_post_subcycle(self.handle, stage), "sih_evp_post_subcycle")

    def run(self):
        check(self._lib.sih_evp_run(self.handle), "sih_evp_run")

    def subcycle_timed(self, n, use_graph=False, sample_every=0):
        t = ctypes.c_double(0)
        tc = ctypes.c_double(0)
        tv = ctypes.c_double(0)
        check(self._lib.sih_evp_subcycle_timed(self.handle, int(n), int(bool(use_graph)), int(sample_every),
                                               ctypes.byref(t), ctypes.byref(tc), ctypes.byref(tv)),
              "sih_evp_subcycle_timed")
        return t.value, tc.value, tv.value

    def sync(self):
        check(self._lib.sih_sync(self.handle), "sih_sync")

    def set_stream(self, stream_ptr):
        check(self._lib.sih_set_stream(self.handle, ctypes.c_void_p(stream_ptr) if stream_ptr else None),
              "sih_set_stream")

    def halo_set_lists(self, location, send_idx, recv_idx, send_counts=None, recv_counts=None):
        s = np.ascontiguousarray(send_idx, dtype=np.int32)
        r = np.ascontiguousarray(recv_idx, dtype=np.int32)
        npeers = 0
        sc = rc = None
        if send_counts is not None:
            sc = np.ascontiguousarray(send_counts, dtype=np.int32)
            rc = np.ascontiguousarray(recv_counts, dtype=np.int32)
            npeers = sc.size
        check(self._lib.sih_halo_set_lists(self.handle, int(location), s.size, _ptr(s) if s.size else None,
                                           r.size, _ptr(r) if r.size else None, npeers,
                                           _ptr(sc) if npeers else None, _ptr(rc) if npeers else None),
              "sih_halo_set_lists")

    def _names(self, names):
        arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        return arr

    def halo_pack(self, location, names, device_ptr):
        check(self._lib.sih_halo_pack(self.handle, int(location), len(names), self._names(names),
                                      ctypes.c_void_p(device_ptr)), "sih_halo_pack")

    def halo_unpack(self, location, names, device_ptr):
        check(self._lib.sih_halo_unpack(self.handle, int(location), len(names), self._names(names),
                                        ctypes.c_void_p(device_ptr)), "sih_halo_unpack")

    # -- halo exchange inside the library --------------------------------------------------------
    def comm_create(self, rank, nranks, max_cell_planes, max_vertex_planes):
        check(self._lib.sih_comm_create(self.handle, int(rank), int(nranks), int(max_cell_planes),
                                        int(max_vertex_planes)), "sih_comm_create")

    def comm_handle(self):
        buf = ctypes.create_string_buffer(self._lib.sih_comm_handle_size())
        check(self._lib.sih_comm_get_handle(self.handle, buf), "sih_comm_get_handle")
        return buf.raw

    def comm_connect(self, handles):
        """handles: the comm_handle() bytes of every rank, in rank order"""
        blob = b"".join(handles)
        assert len(blob) == len(handles) * self._lib.sih_comm_handle_size()
        check(self._lib.sih_comm_connect(self.handle, blob), "sih_comm_connect")

    def comm_destroy(self):
        check(self._lib.sih_comm_destroy(self.handle), "sih_comm_destroy")

    def comm_set_timeout(self, ms):
        check(self._lib.sih_comm_set_timeout(self.handle, float(ms)), "sih_comm_set_timeout")

    def comm_status(self):
        check(self._lib.sih_comm_status(self.handle), "sih_comm_status")

    def halo_exchange(self, names):
        check(self._lib.sih_halo_exchange(self.handle, len(names), self._names(names)), "sih_halo_exchange")

    def halo_exchange_start(self, names):
        check(self._lib.sih_halo_exchange_start(self.handle, len(names), self._names(names)),
              "sih_halo_exchange_start")

    def halo_exchange_finish(self, names):
        check(self._lib.sih_halo_exchange_finish(self.handle, len(names), self._names(names)),
              "sih_halo_exchange_finish")

    def subcycle_exchange(self, n, last=False, interval=1, overlap=False, use_graph=True):
        check(self._lib.sih_evp_subcycle_exchange(self.handle, int(n), int(bool(last)), int(interval),
                                                  int(bool(overlap)), int(bool(use_graph))),
              "sih_evp_subcycle_exchange")

    # -- incremental remap ---------------------------------------------------------------------
    @staticmethod
    def _edge_desc(mesh_pool):
        d = IrDesc()
        for k in ("edgesOnCell", "cellsOnEdge", "verticesOnEdge", "edgesOnVertex"):
            a = mesh_pool[k]
            assert a.dtype == np.int32 and a.flags["F_CONTIGUOUS"], k
            setattr(d, k, _ptr(a))
        for k in ("xEdge", "yEdge", "zEdge", "dcEdge", "dvEdge"):
            a = mesh_pool[k]
            assert a.dtype == np.float64, k
            setattr(d, k, _ptr(a))
        return d

    def set_edge_mesh(self, mesh_pool):
        """edge positions and connectivity for the weak scheme (before evp_init)"""
        d = self._edge_desc(mesh_pool)
        check(self._lib.sih_evp_set_edge_mesh(self.handle, ctypes.byref(d)), "sih_evp_set_edge_mesh")

    def ir_init(self, mesh_pool, nCategories, rotate):
        d = self._edge_desc(mesh_pool)
        a = mesh_pool["coeffs_reconstruct"]
        assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"]
        d.coeffsReconstruct = _ptr(a)
        d.nCategories = int(nCategories)
        d.rotateCartesianGrid = int(bool(rotate))
        check(self._lib.sih_ir_init(self.handle, ctypes.byref(d)), "sih_ir_init")

    def ir_set_tracers(self, names, parents, nLayers):
        nl = np.ascontiguousarray(nLayers, dtype=np.int32)
        check(self._lib.sih_ir_set_tracers(self.handle, len(names), self._names(names), self._names(parents),
                                           _ptr(nl)), "sih_ir_set_tracers")

    def _column(self, fn, names, n_categories, first_cell, n_cells, arr):
        assert arr.dtype == np.float64 and (arr.flags["F_CONTIGUOUS"] or arr.flags["C_CONTIGUOUS"])
        arr_t = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
        check(fn(self.handle, len(names), arr_t, int(n_categories), int(first_cell), int(n_cells), _ptr(arr),
                 ctypes.c_size_t(arr.size)), fn.__name__)

    def column_get(self, names, n_categories, first_cell, n_cells, arr):
        """records of the listed fields for host cells [first_cell, first_cell + n_cells): arr(S, nCategories, nCells)
        in Fortran order"""
        self._column(self._lib.sih_column_get, names, n_categories, first_cell, n_cells, arr)

    def column_put(self, names, n_categories, first_cell, n_cells, arr):
        self._column(self._lib.sih_column_put, names, n_categories, first_cell, n_cells, arr)

    def ir_run(self, dt, check_conservation=False, check_monotonicity=False, check_recover_means=False):
        checks = ((1 if check_conservation else 0) | (2 if check_monotonicity else 0)
                  | (4 if check_recover_means else 0))
        check(self._lib.sih_ir_run(self.handle, float(dt), checks), "sih_ir_run")

    def ridge_prepare_fields(self, nCategories):
        check(self._lib.sih_column_ridge_init(self.handle, int(nCategories)), "sih_column_ridge_init")

    def column_ridge(self, dt, ndtd, participation, redistribution, efolding_scale, thickness_limits, cleanup=True):
        """sih_column_ridge: colpkg_step_ridge (ridge_ice, then the ITD clean-up unless ``cleanup`` is false) for every
        owned cell on the device fields"""
        lim = np.ascontiguousarray(thickness_limits, dtype=np.float64)
        cfg = RidgeConfig(float(dt), int(ndtd), int(participation), int(redistribution), float(efolding_scale),
                          int(lim.size - 1), lim.ctypes.data, 1 if cleanup else 0)
        check(self._lib.sih_column_ridge(self.handle, ctypes.byref(cfg)), "sih_column_ridge")

    def column_update_state(self, nCategories, dt, tendencies=True):
        """sih_column_update_state: the aggregate state of every owned cell (colpkg_aggregate) and, with
        ``tendencies``, the transport tendencies"""
        check(self._lib.sih_column_update_state(self.handle, int(nCategories), float(dt), 1 if tendencies else 0),
              "sih_column_update_state")

    def column_reinitialize_diagnostics(self, nCategories):
        check(self._lib.sih_column_reinitialize_diagnostics(self.handle, int(nCategories)),
              "sih_column_reinitialize_diagnostics")

    def ir_unpack_reconstruction(self):
        check(self._lib.sih_ir_unpack_reconstruction(self.handle), "sih_ir_unpack_reconstruction")

    def ir_check_monotonicity(self):
        check(self._lib.sih_ir_check_monotonicity(self.handle), "sih_ir_check_monotonicity")

    def ir_get_sums(self, name, n):
        a = np.zeros(n)
        b = np.zeros(n)
        check(self._lib.sih_ir_get_sums(self.handle, name.encode(), _ptr(a), _ptr(b)), "sih_ir_get_sums")
        return a, b

    def ir_run_timed(self, dt):
        t = ctypes.c_double(0)
        ph = np.zeros(5)
        check(self._lib.sih_ir_run_timed(self.handle, float(dt), ctypes.byref(t), _ptr(ph)), "sih_ir_run_timed")
        return t.value, ph

    # -- upwind advection -----------------------------------------------------------------------
    def upwind_init(self, mesh_pool, nCategories, rotate):
        d = self._edge_desc(mesh_pool)
        d.nCategories = int(nCategories)
        d.rotateCartesianGrid = int(bool(rotate))
        check(self._lib.sih_upwind_init(self.handle, ctypes.byref(d)), "sih_upwind_init")

    def upwind_prepare(self, check_conservation=False):
        check(self._lib.sih_upwind_prepare(self.handle, int(bool(check_conservation))), "sih_upwind_prepare")

    def upwind_advect(self, dt, check_conservation=False):
        check(self._lib.sih_upwind_advect(self.handle, float(dt), int(bool(check_conservation))), "sih_upwind_advect")

    def upwind_run(self, dt, check_conservation=False):
        check(self._lib.sih_upwind_run(self.handle, float(dt), int(bool(check_conservation))), "sih_upwind_run")

    def upwind_get_sums(self, name, n):
        a = np.zeros(n)
        b = np.zeros(n)
        check(self._lib.sih_upwind_get_sums(self.handle, name.encode(), _ptr(a), _ptr(b)), "sih_upwind_get_sums")
        return a, b

    def info(self, key):
        v = ctypes.c_double(0)
        check(self._lib.sih_get_info(self.handle, key.encode(), ctypes.byref(v)), "sih_get_info")
        return v.value
