"""Host mirror of the dynamics part of the reference's time step, the caller of the two device paths
(/root/reference/src/shared/mpas_seaice_time_integration.F:131-173): the loop over dynamics subcycles, each a
velocity solve followed by the advection and the column package's ridging, and the tracers' special
boundaries at the end of the step.

The column package is not part of this library.  ``column_dynamics`` stands for
``seaice_column_dynamics_time_integration`` (:158): a callable that works on the host pools.  When it is given,
the tracers come back to the host after every advection and go up again before the next velocity solve, as
in the reference, where every routine works on the pools.  Without it, state stays on the device from the first
velocity solve to the last advection of the step.
"""
from __future__ import annotations

from . import advection as adv
from . import special_boundaries as sb
from . import velocity_solver as vs


def seaice_timestep_dynamics(domain, column_dynamics=None):
    """:131-173 without the column physics before and after the dynamics.  ``column_dynamics="device"``: the ridging
    runs on the device (column.seaice_column_dynamics_time_integration), and from the first velocity solve to the last
    ridging of the step the tracers never visit the host: every velocity solve after the first takes ice area and
    volumes from the device, every advection and ridging works on what the stage before left there, and the pools
    get tracers and ridging diagnostics once, at the end."""
    cfg = domain.configs
    if column_dynamics == "device":
        return _timestep_dynamics_on_device(domain)
    if cfg.get("config_use_dynamics", True):
        n = int(cfg["config_dynamics_subcycle_number"])
        for i in range(n):                                    # :138-162
            vs.seaice_run_velocity_solver(domain)             # :146
            if cfg.get("config_use_advection", True):         # :151-152
                # The pools get the advected tracers back every time (the next velocity solve reads ice area
                # and volumes from them).  Going up, the first advection of the step takes everything from
                # the pools; later ones find velocities and tracers on the device unless the column physics
                # has touched the pools in between.
                adv.seaice_run_advection(domain, upload=(i == 0 or column_dynamics is not None))
            if column_dynamics is not None:
                column_dynamics(domain)                       # :158
    sb.seaice_set_special_boundaries_tracers(domain)          # :173


def _timestep_dynamics_on_device(domain):
    from . import column
    cfg = domain.configs
    if cfg.get("config_use_dynamics", True):
        n = int(cfg["config_dynamics_subcycle_number"])
        for i in range(n):
            last = i == n - 1
            # :142: ridging rates to zero, transport tendencies from the aggregate state; the first time on the pools
            # (they go up with the cell fields of the first ridging), later on what the device holds
            column.seaice_column_reinitialize_diagnostics_dynamics(domain, on_device=(i > 0))
            vs.seaice_run_velocity_solver(domain, tracers_on_device=(i > 0))
            if cfg.get("config_use_advection", True):
                adv.seaice_run_advection(domain, upload=(i == 0), download=False)
            # the first ridging of the step brings the host's cell fields (ocean state, fluxes, open water); later ones
            # go on from what the one before left on the device; the tracers are on the device either way
            column.seaice_column_dynamics_time_integration(domain, upload=("cells" if i == 0 else False), download=last)
    sb.seaice_set_special_boundaries_tracers(domain)
