"""Host mirror of the reference's upwind advection driver for the device library.

``seaice_init_advection_upwind`` / ``seaice_run_advection_upwind`` follow
/root/reference/src/shared/mpas_seaice_advection_upwind.F (:79-131, :385-518).  The per-block arithmetic runs
in libseaice_hip.so (sih_upwind_*); this module keeps what belongs to the pools: the tracer connectivity
table, the halo exchange between preparation and advection, the bookkeeping of the two time levels and the
tracer_conservation aggregates.  There is no CPU path.
"""
from __future__ import annotations

import numpy as np

from . import pools
from .lib import SeaiceHipError

# define_tracer_connectivities, :145-189: (child, parent) in the reference's order
TRACER_CONNECTIVITIES = (("iceAreaCategory", "none"), ("surfaceTemperature", "iceAreaCategory"),
                         ("iceVolumeCategory", "surfaceTemperature"), ("snowVolumeCategory", "iceVolumeCategory"))
CHAIN = tuple(c for c, _ in TRACER_CONNECTIVITIES)


def seaice_init_advection_upwind(domain):
    """:79-131"""
    cfg = domain.configs
    domain.solver_state["tracerConnectivities"] = TRACER_CONNECTIVITIES
    for b in domain.blocklist:
        if getattr(b, "ctx", None) is None:
            raise SeaiceHipError("seaice_init_advection_upwind: seaice_mesh_pool_create must run first")
        pools.allocate_upwind_pools(b)
        m = b.structs["mesh"]
        b.ctx.upwind_init(m, b.dimensions["nCategories"],
                          bool(cfg["config_rotate_cartesian_grid"]) and bool(m["on_a_sphere"]))
        b.ctx.get_many([("normalVectorEdge", b.structs["velocity_solver"]["normalVectorEdge"]),
                        ("interiorEdge", b.structs["boundary"]["interiorEdge"])])


def seaice_run_advection_upwind(domain, upload=True, download=True):
    """:385-518.  ``upload`` / ``download`` as in ``advection.seaice_run_advection``."""
    cfg = domain.configs
    comm = getattr(domain, "comm", None)
    check = bool(cfg["config_conservation_check"])
    if upload:
        for b in domain.blocklist:
            vs, tp = b.structs["velocity_solver"], b.structs["tracers"]
            b.ctx.put_many([("uVelocity", vs["uVelocity"]), ("vVelocity", vs["vVelocity"])]
                           + [(n, tp[n]) for n in CHAIN])
    for b in domain.blocklist:
        b.ctx.upwind_prepare(check)                      # prepare_advection :441-453
    if comm is not None and cfg["config_use_halo_exch"]:
        comm.halo_exchange(domain, list(CHAIN))          # halo_exchange_advection :459
    for b in domain.blocklist:
        dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
        b.ctx.upwind_advect(dt, check)                   # :479-513
    for b in domain.blocklist:
        tr1, tr2 = b.structs["tracers"], b.structs["tracers_level2"]
        # MPAS_pool_shift_time_levels acts on the whole tracers pool (:2035): the fields outside the chain
        # come back as their second level, which initialize_timelevel_variables zeroed (:1711-1713)
        nC = b.dimensions["nCells"]
        for k in pools.UPWIND_POOL_TRACERS:
            if k in CHAIN:
                tr2[k][..., nC:] = 0.0  # the spare slot the device never touches; zeroed with the rest at :1707-1710
            else:
                tr2[k][:] = 0.0
        pools.mpas_pool_shift_time_levels(b, "tracers")
        tr1, tr2 = b.structs["tracers"], b.structs["tracers_level2"]
        if download:
            items = []
            for k in CHAIN:
                items += [(k, tr1[k]), (k + "Level2", tr2[k]), (k + "Tend", b.structs["tracer_tendencies"][k + "Tend"]),
                          (k + "EdgeFlux", b.structs["tracer_edge_fluxes"][k + "EdgeFlux"])]
            b.ctx.get_many(items)
        if check:
            nCat = b.dimensions["nCategories"]
            for k in CHAIN:
                a0, a1 = b.ctx.upwind_get_sums(k, nCat)
                b.structs["tracer_conservation"][k + "Cons"][0, :] = a0
                b.structs["tracer_conservation_level2"][k + "Cons"][0, :] = a1
