"""Host mirror of the reference's advection driver for the device library.

``seaice_init_advection`` / ``seaice_run_advection`` follow
/root/reference/src/shared/mpas_seaice_advection.F:40-140 (``config_advection_type = 'upwind'`` goes to
``advection_upwind``) and, for ``config_advection_type = 'incremental_remap'``, /root/reference/src/shared/mpas_seaice_advection_incremental_remap.F
(seaice_init_advection_incremental_remap :165, seaice_run_advection_incremental_remap :2338).  The
per-block arithmetic runs in libseaice_hip.so (sih_ir_*); this module keeps the reference's call
sequence: halo updates of velocity and tracers, the block loop, the conservation check with its global
sums, the optional final halo update.  There is no CPU path.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import tracers as trc
from .lib import SeaiceHipError


def seaice_init_advection(domain):
    """mpas_seaice_advection.F:40-76"""
    cfg = domain.configs
    if not cfg.get("config_use_advection", True):
        return
    if cfg["config_advection_type"] == "upwind":  # mpas_seaice_advection.F:64-66
        from . import advection_upwind
        return advection_upwind.seaice_init_advection_upwind(domain)
    if cfg["config_advection_type"] != "incremental_remap":
        raise SeaiceHipError("config_advection_type: unknown option " + str(cfg["config_advection_type"]))
    # check_halo_layer_number, incremental_remap.F:829-874
    need = 3 if int(domain.blocklist[0].dimensions["vertexDegree"]) == 4 else 2   # SCVT meshes 2, quadrilateral ones 3
    if getattr(domain, "comm", None) is not None and int(cfg.get("config_num_halos", need)) < need:
        raise SeaiceHipError("MPAS-seaice: IR too few halo layers for incremental remapping (config_num_halos)")
    tracersHead = trc.seaice_add_tracers_to_linked_list(domain)
    if cfg["config_monotonicity_check"]:
        # the reference's tracer_monotonicity pool (incremental_remap_tracers.F:1045-1110)
        for b in domain.blocklist:
            pool = b.structs.setdefault("tracer_monotonicity", {})
            for t in tracersHead:
                shape = (t.nLayers, b.dimensions["nCategories"], b.dimensions["nCells"] + 1)
                pool[t.tracerName + "LocalMin"] = np.zeros(shape, order="F")
                pool[t.tracerName + "LocalMax"] = np.zeros(shape, order="F")
    for b in domain.blocklist:
        if getattr(b, "ctx", None) is None:
            raise SeaiceHipError("seaice_init_advection: seaice_mesh_pool_create must run first")
        m = b.structs["mesh"]
        if "coeffs_reconstruct" not in m:
            raise SeaiceHipError("mesh pool lacks coeffs_reconstruct (the output of mpas_init_reconstruct)")
        b.ctx.ir_init(m, b.dimensions["nCategories"],
                      bool(cfg["config_rotate_cartesian_grid"]) and bool(m["on_a_sphere"]))
        b.ctx.ir_set_tracers([t.tracerName for t in tracersHead], [t.parentName for t in tracersHead],
                             [t.nLayers for t in tracersHead])


def _n(b, loc):
    return b.dimensions[{"cell": "nCells", "vertex": "nVertices"}[loc]]


def seaice_run_advection(domain, updateHaloInit=True, updateHaloFinal=False, upload=True, download=True):
    """mpas_seaice_advection.F:90-136 -> seaice_run_advection_incremental_remap (:2338-2667).

    ``upload`` / ``download`` move uVelocity, vVelocity and the tracers between the host pools and the
    device around the call; a host that keeps its state on the device between the velocity solver and
    the advection passes ``upload=False``.
    """
    cfg = domain.configs
    if not cfg.get("config_use_advection", True):
        return
    if cfg["config_advection_type"] == "upwind":  # mpas_seaice_advection.F:115-119
        from . import advection_upwind
        return advection_upwind.seaice_run_advection_upwind(domain, upload=upload, download=download)
    tracersHead = domain.solver_state["tracersHead"]
    names = [t.tracerName for t in tracersHead]
    comm = getattr(domain, "comm", None)
    if upload:
        for b in domain.blocklist:
            vs, tp = b.structs["velocity_solver"], b.structs["tracers"]
            b.ctx.put_many([("uVelocity", vs["uVelocity"]), ("vVelocity", vs["vVelocity"])]
                           + [(n, tp[n]) for n in names])
    # initial halo updates of velocity, mass and tracers, :2436-2487
    if updateHaloInit and comm is not None and cfg["config_use_halo_exch"]:
        comm.halo_exchange(domain, ["uVelocity", "vVelocity"])
        comm.halo_exchange(domain, names)
    check = bool(cfg["config_conservation_check"])
    mono = bool(cfg["config_monotonicity_check"])
    recover = bool(cfg["config_recover_tracer_means_check"])
    if recover:
        domain.solver_state["ir_recover_means_warnings"] = 0
    for b in domain.blocklist:
        dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
        b.ctx.ir_run(dt, check, mono, recover)
        if recover:  # recover_tracer_means :7541-7982 only logs warnings; the count stands for them here
            n = int(b.ctx.info("irRecoverMeansWarnings"))
            domain.solver_state["ir_recover_means_warnings"] = \
                domain.solver_state.get("ir_recover_means_warnings", 0) + n
            if n:
                warnings.warn(f"IR: Reconstructed tracer does not recover the mean in {n} entries")
    if check:  # check_tracer_conservation, :8126-8253 (global sums over blocks and ranks)
        nCat = domain.blocklist[0].dimensions["nCategories"]
        for t in tracersHead:
            init = np.zeros(t.nLayers * nCat)
            fin = np.zeros(t.nLayers * nCat)
            for b in domain.blocklist:
                a0, a1 = b.ctx.ir_get_sums(t.tracerName, t.nLayers * nCat)
                init += a0
                fin += a1
            if comm is not None and hasattr(comm, "allreduce_sum"):
                init, fin = comm.allreduce_sum(init), comm.allreduce_sum(fin)
            big = np.abs(init) > 1.0e-11
            ratio = np.where(big, (fin - init) / np.where(big, init, 1.0), 0.0)
            if (np.abs(ratio) > 1.0e-11).any():
                raise SeaiceHipError("IR: Tracer conservation error: tracer = " + t.tracerName)
            domain.solver_state.setdefault("ir_sums", {})[t.tracerName] = (init, fin)
    if mono:  # check_tracer_monotonicity :8416-8749, outside the block loop because of its halo update
        if comm is not None and cfg["config_use_halo_exch"]:
            for t in tracersHead:
                if t.parent is not None:
                    comm.halo_exchange(domain, [t.tracerName + "LocalMin", t.tracerName + "LocalMax"])
        try:
            for b in domain.blocklist:
                b.ctx.ir_check_monotonicity()
        finally:
            if download:  # the extended bounds, as the reference leaves them in its pool
                for b in domain.blocklist:
                    pool = b.structs["tracer_monotonicity"]
                    b.ctx.get_many((n, pool[n]) for t in tracersHead if t.parent is not None
                                   for n in (t.tracerName + "LocalMin", t.tracerName + "LocalMax"))
    if updateHaloFinal and comm is not None:
        comm.halo_exchange(domain, names)
    if download:
        for b in domain.blocklist:
            tp = b.structs["tracers"]
            b.ctx.get_many((n, tp[n]) for n in names)
