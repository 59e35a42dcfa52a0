"""Host mirror of the reference's special-boundaries module
(/root/reference/src/shared/mpas_seaice_special_boundaries.F).

``seaice_init_special_boundaries`` turns the global source ids of the ``special_boundaries`` pool into
local ones (init_special_boundaries_velocity, :84-155) and, for blocks that live on the device, hands
the lists and replacement masks to the library, whose subcycle loop then applies them before the first
and after every iteration (seaice_set_special_boundaries_velocity :253-331, ..._velocity_masks :345-401,
called from mpas_seaice_velocity_solver.F:2441-2456).
"""
from __future__ import annotations

import numpy as np

VELOCITY_BOUNDARY_NONE, VELOCITY_BOUNDARY_PERIODIC, VELOCITY_BOUNDARY_REVERSE, VELOCITY_BOUNDARY_ZERO = 0, 1, 2, 3
TRACER_BOUNDARY_NONE, TRACER_BOUNDARY_ZERO, TRACER_BOUNDARY_SET = 0, 1, 2  # :42-45


def seaice_init_special_boundaries(domain):
    cfg = domain.configs
    use_vel = bool(cfg["config_use_special_boundaries_velocity"])
    use_masks = bool(cfg["config_use_special_boundaries_velocity_masks"])
    several = use_vel and getattr(domain, "comm", None) is not None and any(
        getattr(b, "ctx", None) is not None for b in domain.blocklist)
    if several and (int(cfg.get("config_halo_exchange_interval", 1)) != 1
                    or cfg.get("config_overlap_halo_exchange", False)):
        # the reference applies them after the halo update that ends every iteration (velocity_solver.F:2449-2456)
        from .lib import SeaiceHipError
        raise SeaiceHipError("special velocity boundaries on several blocks need one plain halo exchange per "
                             "subcycle (config_halo_exchange_interval = 1, no overlap)")
    if bool(cfg.get("config_use_special_boundaries_tracers", False)):  # init_special_boundaries_tracers, :169-239
        for b in domain.blocklist:
            sb = b.structs["special_boundaries"]
            nC = b.dimensions["nCells"]
            ids = b.structs["mesh"]["indexToCellID"][:nC]
            g2l = np.zeros(int(ids.max()) + 1, dtype=np.int64)
            g2l[ids] = np.arange(1, nC + 1)
            marked = sb["tracerBoundaryType"][:nC] != TRACER_BOUNDARY_NONE
            src = sb["tracerBoundarySource"][:nC]
            sb["tracerBoundarySourceLocal"][:nC][marked] = g2l[np.clip(src[marked], 0, g2l.size - 1)]
    for b in domain.blocklist:
        sb = b.structs["special_boundaries"]
        nV = b.dimensions["nVertices"]
        if use_vel:
            ids = b.structs["mesh"]["indexToVertexID"][:nV]
            g2l = np.zeros(int(ids.max()) + 1, dtype=np.int64)
            g2l[ids] = np.arange(1, nV + 1)
            typ = sb["vertexBoundaryType"][:nV]
            marked = typ != VELOCITY_BOUNDARY_NONE
            src = sb["vertexBoundarySource"][:nV]
            # a source that is not in this block maps to 0: only an outermost halo vertex can have one when the
            # sources are neighbours, and the library leaves such a vertex as the exchange delivered it
            inside = (src[marked] >= 0) & (src[marked] < g2l.size)
            sb["vertexBoundarySourceLocal"][:nV][marked] = np.where(inside, g2l[np.clip(src[marked], 0, g2l.size - 1)], 0)
        ctx = getattr(b, "ctx", None)
        if ctx is not None and (use_vel or use_masks):
            nC = b.dimensions["nCells"]
            ctx.set_special_boundaries(
                np.ascontiguousarray(sb["vertexBoundaryType"][:nV]) if use_vel else None,
                np.ascontiguousarray(sb["vertexBoundarySourceLocal"][:nV]) if use_vel else None,
                np.ascontiguousarray(sb["solveVelocitySpecialBoundaries"][:nV]) if use_masks else None,
                np.ascontiguousarray(sb["solveStressSpecialBoundaries"][:nC]) if use_masks else None)
            if several:
                ctx.defer_special_boundaries(True)


def seaice_set_special_boundaries_tracers(domain):
    """:415-485, called at the end of every time step (mpas_seaice_time_integration.F:173) on the host pools:
    marked cells lose their ice or take the three state tracers of their source cell, in cell order and in
    place, exactly like the reference's loop (a source that was itself replaced earlier hands on its new
    value)."""
    if not bool(domain.configs.get("config_use_special_boundaries_tracers", False)):
        return
    for b in domain.blocklist:
        sb, tr = b.structs["special_boundaries"], b.structs["tracers"]
        nC = b.dimensions["nCells"]
        typ = sb["tracerBoundaryType"][:nC]
        src = sb["tracerBoundarySourceLocal"]
        fields = [tr[k] for k in ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory")]
        for i in np.nonzero(typ != TRACER_BOUNDARY_NONE)[0]:
            if typ[i] == TRACER_BOUNDARY_ZERO:
                for a in fields:
                    a[:, :, i] = 0.0
            elif typ[i] == TRACER_BOUNDARY_SET:
                j = int(src[i]) - 1
                for a in fields:
                    a[:, :, i] = a[:, :, j]
