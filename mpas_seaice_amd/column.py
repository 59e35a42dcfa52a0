"""Host mirror of the column package's part of the dynamics step, for what this library runs on the device:
``seaice_column_dynamics_time_integration`` (/root/reference/src/shared/mpas_seaice_column.F:1139-1194) ->
``column_ridging`` (:3283-3556) -> ``colpkg_step_ridge`` -> ``ridge_ice``
(/root/reference/src/column/ice_mechred.F90:80-597).

The ridging works on the device fields of the advected tracers, so that between the advection and the next velocity
solve nothing has to visit the host.  Scope and limits: see ``sih_column_ridge`` in include/seaice_hip.h (standard
physics tracers, no strata), and so does the aggregation that follows it, ``seaice_column_update_state`` (:4279-4377 ->
``seaice_column_aggregate`` :4391-4494 -> ``colpkg_aggregate``, /root/reference/src/column/ice_colpkg.F90:3366-3480)."""
from __future__ import annotations

from . import pools
from .lib import SeaiceHipError

PARTICIPATION = {"Thorndike75": 0, "exponential": 1}
REDISTRIBUTION = {"Hibler80": 0, "exponential": 1}
INPUT_FIELDS = (("ridging", "ridgeConvergence"), ("ridging", "ridgeShear"), ("icestate", "openWaterArea"),
                ("ocean_coupling", "seaFreezingTemperature"), ("ocean_fluxes", "oceanFreshWaterFlux"),
                ("ocean_fluxes", "oceanHeatFlux"), ("ocean_fluxes", "oceanSaltFlux"), ("ridging", "fractionNewRidgeArea"),
                ("ridging", "fractionNewRidgeVolume"), ("biogeochemistry", "newlyFormedIce"),
                ("diagnostics", "iceAreaTendencyTransport"), ("diagnostics", "iceVolumeTendencyTransport"))
OUTPUT_FIELDS = (("icestate", "openWaterArea"), ("ocean_fluxes", "oceanFreshWaterFlux"), ("ocean_fluxes", "oceanHeatFlux"),
                 ("ocean_fluxes", "oceanSaltFlux"), ("biogeochemistry", "newlyFormedIce"), ("tracers_aggregate", "iceAreaCell")) \
    + tuple(("ridging", k) for k in pools.RIDGING_CELL_FIELDS + pools.RIDGING_CATEGORY_FIELDS)


def column_ridging(domain, upload=True, download=True, cleanup=True):
    """:3283-3556.  ``upload``: True, the tracers and every cell field the ridging reads go from the pools to the device
    first; "cells", only the cell fields that no device routine computes (ocean state, accumulated fluxes, open water,
    the fractions of the last ridging): for a host that has just run velocity solver and advection on the device;
    False, nothing: the device holds what the ridging before this one left (open water, fluxes accumulated so far).
    ``download``: tracers, open water, fluxes and diagnostics come back.  ``cleanup``: colpkg_step_ridge whole, i.e. the
    ITD clean-up (cleanup_itd: rebin, zap_small_areas, zap_snow_temperature) after ridge_ice; False: ridge_ice alone."""
    cfg = domain.configs
    for key, table in (("config_ridging_participation_function", PARTICIPATION),
                       ("config_ridging_redistribution_function", REDISTRIBUTION)):
        if cfg[key] not in table:
            raise SeaiceHipError(f"{key} unknown: {cfg[key]}")
    names = [t.tracerName for t in domain.solver_state["tracersHead"]]
    for b in domain.blocklist:
        pools.allocate_ridging_pools(b)
        tp = b.structs["tracers"]
        dt = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0])
        limits = b.structs["initial"]["categoryThicknessLimits"]
        # the fields only the host knows (ocean state, accumulated fluxes, open water); the rates of the velocity
        # solver and the tracers are on the device already unless the caller says otherwise
        b.ctx.ridge_prepare_fields(b.dimensions["nCategories"])
        items = []
        if upload is True:
            items = [(n, b.structs[p][n]) for p, n in INPUT_FIELDS] + [(n, tp[n]) for n in names]
        elif upload == "cells":
            items = [(n, b.structs[p][n]) for p, n in INPUT_FIELDS if n not in ("ridgeConvergence", "ridgeShear")]
        if items:
            b.ctx.put_many(items)
        b.ctx.column_ridge(dt, int(cfg["config_dynamics_subcycle_number"]),
                           PARTICIPATION[cfg["config_ridging_participation_function"]],
                           REDISTRIBUTION[cfg["config_ridging_redistribution_function"]],
                           float(cfg["config_ridiging_efolding_scale"]), limits, cleanup=cleanup)
        limits[-1] = 1.0e8  # ridge_prep writes it into the host's array (ice_mechred.F90:675)
        domain.solver_state["ridge_iterations"] = int(b.ctx.info("ridgeIterations"))
        if download:
            b.ctx.get_many([(n, tp[n]) for n in names] + [(n, b.structs[p][n]) for p, n in OUTPUT_FIELDS])


AGGREGATE_FIELDS = (("tracers_aggregate", "iceAreaCell"), ("tracers_aggregate", "iceVolumeCell"),
                    ("tracers_aggregate", "snowVolumeCell"), ("icestate", "openWaterArea"),
                    ("diagnostics", "iceAreaTendencyTransport"), ("diagnostics", "iceVolumeTendencyTransport"))
# category tracers whose cell values the `tracers_aggregate` pool holds as <name>Cell (get_cice_physics_tracer_array_cell
# :7007-7183, the standard physics set)
AGGREGATED_TRACERS = ("surfaceTemperature", "iceEnthalpy", "iceSalinity", "snowEnthalpy")


def seaice_column_update_state(domain, stateUpdateType="transport", dt=None, upload=False, download=True):
    """:4279-4377 on the device, for the category tracers the device holds: the aggregate state of every owned cell
    and the transport tendencies.  ``upload``: tracers, freezing temperature and the tendencies' start values go up
    first (a host that has not just run the ridging on the device)."""
    if stateUpdateType != "transport":
        raise SeaiceHipError("seaice_column_update_state: Unknown update type: " + stateUpdateType)
    names = [t.tracerName for t in domain.solver_state["tracersHead"]]
    for b in domain.blocklist:
        pools.allocate_ridging_pools(b)
        tp = b.structs["tracers"]
        step = float(b.structs["velocity_solver"]["dynamicsTimeStep"][0]) if dt is None else float(dt)
        nCat = b.dimensions["nCategories"]
        if upload:
            b.ctx.ridge_prepare_fields(nCat)
            b.ctx.column_update_state(nCat, step, tendencies=False)  # creates the fields; overwritten below
            b.ctx.put_many([(n, tp[n]) for n in names]
                           + [("seaFreezingTemperature", b.structs["ocean_coupling"]["seaFreezingTemperature"])]
                           + [(n, b.structs["diagnostics"][n]) for n in ("iceAreaTendencyTransport",
                                                                         "iceVolumeTendencyTransport")])
        b.ctx.column_update_state(nCat, step, tendencies=True)
        if download:
            ta = b.structs["tracers_aggregate"]
            b.ctx.get_many([(n, b.structs[p][n]) for p, n in AGGREGATE_FIELDS]
                           + [(n + "Cell", ta[n + "Cell"]) for n in AGGREGATED_TRACERS if n in names])


def seaice_column_reinitialize_diagnostics_dynamics(domain, on_device=False):
    """:13569-13723 for the fields of this path, at the start of a dynamics subcycle: the ridging rates back to zero,
    the transport tendencies set to the aggregate state before the dynamics.  On the pools, or (``on_device``) on the
    device fields, for a step whose state stays there between two dynamics subcycles."""
    if not domain.configs["config_use_column_package"]:
        return
    for b in domain.blocklist:
        pools.allocate_ridging_pools(b)
        if on_device:
            b.ctx.column_reinitialize_diagnostics(b.dimensions["nCategories"])
            continue
        rg, ta, dg = b.structs["ridging"], b.structs["tracers_aggregate"], b.structs["diagnostics"]
        for k in pools.RIDGING_CELL_FIELDS:
            rg[k][:] = 0.0
        dg["iceAreaTendencyTransport"][:] = ta["iceAreaCell"]
        dg["iceVolumeTendencyTransport"][:] = ta["iceVolumeCell"]
        dg["iceAgeTendencyTransport"][:] = 0.0


def seaice_column_dynamics_time_integration(domain, upload=True, download=True):
    """:1139-1194: the ridging, then the aggregate state and the transport tendencies (column_combine_snow_ice_tracers
    belongs to config_use_vertical_tracers, outside this library)"""
    cfg = domain.configs
    if not cfg["config_use_column_package"]:
        return
    ridging = cfg.get("config_use_column_ridging", True)
    if ridging:
        column_ridging(domain, upload=upload, download=download)
    seaice_column_update_state(domain, "transport", upload=(upload is True and not ridging), download=download)
