"""Host-side mirror of the reference's velocity-solver interface, driving the HIP library.

Same entry points, argument meaning and call order as the reference modules they replace:

  seaice_mesh_pool_create(domain)          /root/reference/src/shared/mpas_seaice_mesh_pool.F:76
  seaice_init_velocity_solver(domain)      /root/reference/src/shared/mpas_seaice_velocity_solver.F:85
  seaice_run_velocity_solver(domain,clock) /root/reference/src/shared/mpas_seaice_velocity_solver.F:562
  seaice_mesh_pool_destroy(domain)         /root/reference/src/shared/mpas_seaice_mesh_pool.F:188

The pools stay the source of truth at dynamics-step granularity: inputs are uploaded at entry and
every field the reference solver leaves in the pools is downloaded at exit, so restart / analysis /
coupling code sees the same state.  This is the logic the Fortran ISO_C_BINDING shim carries in a
real MPAS build; it is written in Python here because the MPAS framework the Fortran shim needs
(pools, dmpar) is not available outside a host model.
"""
from __future__ import annotations

import os

import numpy as np

from . import lib as _lib

RELATION = {"evp": 1, "evp_revised": 2, "linear": 3}

# (pool, field) moved host -> device at the start of every seaice_run_velocity_solver
RUN_INPUTS = [
    ("tracers", "iceAreaCategory"), ("tracers", "iceVolumeCategory"), ("tracers", "snowVolumeCategory"),
    ("ocean_coupling", "uOceanVelocity"), ("ocean_coupling", "vOceanVelocity"),
    ("ocean_coupling", "landIceMask"), ("ocean_coupling", "landIceMaskVertex"),
    ("atmos_coupling", "uAirVelocity"), ("atmos_coupling", "vAirVelocity"), ("atmos_coupling", "airDensity"),
    ("velocity_solver", "uVelocity"), ("velocity_solver", "vVelocity"),
    ("velocity_solver", "solveVelocityPrevious"),
    ("velocity_solver", "stressDivergenceU"), ("velocity_solver", "stressDivergenceV"),
    ("velocity_solver", "oceanStressU"), ("velocity_solver", "oceanStressV"),
    ("velocity_variational", "stress11"), ("velocity_variational", "stress22"),
    ("velocity_variational", "stress12"),
]
# (pool, field) moved device -> host at the end
RUN_OUTPUTS = [
    ("velocity_solver", k) for k in (
        "uVelocity", "vVelocity", "uVelocityInitial", "vVelocityInitial", "solveStress", "solveVelocity",
        "solveVelocityPrevious", "icePressure", "stressDivergenceU", "stressDivergenceV", "airStressCellU",
        "airStressCellV", "airStressVertexU", "airStressVertexV", "uOceanVelocityVertex", "vOceanVelocityVertex",
        "oceanStressU", "oceanStressV", "oceanStressCoeff", "oceanStressCellU", "oceanStressCellV",
        "surfaceTiltForceU", "surfaceTiltForceV", "totalMassVertexfVertex", "divergence", "shear")
] + [
    ("velocity_variational", k) for k in (
        "strain11", "strain22", "strain12", "stress11", "stress22", "stress12", "replacementPressure",
        "principalStress1", "principalStress2")
] + [
    ("icestate", k) for k in ("iceAreaVertex", "totalMassVertex", "iceAreaCellInitial", "totalMassCell")
] + [
    ("tracers_aggregate", k) for k in ("iceAreaCell", "iceVolumeCell", "snowVolumeCell")
]
STATIC_OUTPUTS = [
    ("boundary", "interiorVertex"),
] + [("velocity_variational", k) for k in (
    "tanLatVertexRotatedOverRadius", "cellVerticesAtVertex", "basisGradientU", "basisGradientV",
    "basisIntegralsU", "basisIntegralsV", "basisIntegralsMetric", "variationalDenominator")]


# velocity_weak pool: pool key (Registry name_in_code) -> device field (Registry name), Registry.xml:3730-3744
WEAK_FIELDS = {"strain11": "strain11weak", "strain22": "strain22weak", "strain12": "strain12weak",
               "stress11": "stress11weak", "stress22": "stress22weak", "stress12": "stress12weak",
               "principalStress1": "principalStress1Weak", "principalStress2": "principalStress2Weak",
               "replacementPressure": "replacementPressureWeak"}
WEAK_STATIC = ("normalVectorPolygon", "normalVectorTriangle", "latCellRotated", "latVertexRotated")


def _weak(domain):
    """both schemes weak: the per-cell fields of the velocity_weak pool carry the state"""
    return domain.configs["config_stress_divergence_scheme"] == "weak"


def _weak_strain(domain):
    """weak strain (alone or with the weak divergence): the weak initialisation is needed"""
    return domain.configs["config_strain_scheme"] == "weak"


def _items(domain, b, pairs):
    """(device field, host array) for (pool, key) pairs; with the weak scheme the per-corner fields of the
    velocity_variational pool give way to the per-cell ones of the velocity_weak pool"""
    out = []
    for pool, name in pairs:
        if _weak(domain) and pool == "velocity_variational":
            if name in WEAK_FIELDS:
                out.append((WEAK_FIELDS[name], b.structs["velocity_weak"][name]))
            continue
        out.append((name, b.structs[pool][name]))
    return out


def _evp_config(domain, block):
    cfg = domain.configs
    c = _lib.EvpConfig()
    c.dt = float(cfg["config_dt"])
    c.dynamicsSubcycleNumber = int(cfg["config_dynamics_subcycle_number"])
    c.elasticSubcycleNumber = int(cfg["config_elastic_subcycle_number"])
    rel = cfg["config_constitutive_relation_type"]
    if rel not in RELATION:
        raise _lib.SeaiceHipError("config_constitutive_relation_type unknown: " + str(rel))
    c.constitutiveRelation = RELATION[rel]
    schemes = {"variational": 0, "weak": 1}
    for key in ("config_strain_scheme", "config_stress_divergence_scheme"):
        if cfg[key] not in schemes:
            raise _lib.SeaiceHipError(f"{key} unknown: {cfg[key]}")
    c.strainScheme = schemes[cfg["config_strain_scheme"]]
    c.stressDivergenceScheme = schemes[cfg["config_stress_divergence_scheme"]]
    c.averageVariationalStrain = int(bool(cfg["config_average_variational_strain"]))
    c.useSpecialBoundariesVelocity = int(bool(cfg["config_use_special_boundaries_velocity"]))
    c.useSpecialBoundariesVelocityMasks = int(bool(cfg["config_use_special_boundaries_velocity_masks"]))
    basis = cfg["config_variational_basis"]
    if basis == "wachspress":
        c.variationalBasis = 1
    elif basis == "pwl":
        c.variationalBasis = 2
    elif basis == "none":
        c.variationalBasis = 0
    else:
        raise _lib.SeaiceHipError("Unknown variational basis type: " + str(basis))
    den = cfg["config_variational_denominator_type"]
    if den not in ("original", "alternate"):
        raise _lib.SeaiceHipError("Unknown variational denominator type: " + str(den))
    c.variationalDenominator = 1 if den == "alternate" else 0
    c.wachspressIntegrationOrder = int(cfg["config_wachspress_integration_order"])
    itype = cfg["config_wachspress_integration_type"]
    types = {"dunavant": 1, "trapezoidal": 2, "fekete": 3}   # get_integration_factors, wachspress.F:1250-1283
    if itype not in types:
        raise _lib.SeaiceHipError("get_integration_factors: Unknown wachspress integration type: " + str(itype))
    c.wachspressIntegrationType = types[itype]
    c.rotateCartesianGrid = int(bool(cfg["config_rotate_cartesian_grid"]))
    c.includeMetricTerms = int(bool(cfg["config_include_metric_terms"]))
    c.useAirStress = int(bool(cfg["config_use_air_stress"]))
    c.useOceanStress = int(bool(cfg["config_use_ocean_stress"]))
    ost = cfg["config_ocean_stress_type"]
    if ost not in ("quadratic", "linear"):
        raise _lib.SeaiceHipError("config_ocean_stress_type unknown: " + str(ost))
    c.oceanStressType = 1 if ost == "quadratic" else 2
    c.surfaceTilt = 0 if not cfg["config_use_surface_tilt"] else (1 if cfg["config_geostrophic_surface_tilt"] else 2)
    c.calcVelocityMasks = int(bool(cfg["config_calc_velocity_masks"]))
    c.useColumnPackage = int(bool(cfg["config_use_column_package"]))
    c.useColumnVerticalThermodynamics = int(bool(cfg["config_use_column_vertical_thermodynamics"]))
    # colpkg_ice_strength options (mpas_seaice_column.F:11432-11460)
    for attr, key, table in (
            ("iceStrengthFormulation", "config_ice_strength_formulation", {"Hibler79": 0, "Rothrock75": 1}),
            ("ridgingParticipation", "config_ridging_participation_function", {"Thorndike75": 0, "exponential": 1}),
            ("ridgingRedistribution", "config_ridging_redistribution_function", {"Hibler80": 0, "exponential": 1})):
        if cfg[key] not in table:
            raise _lib.SeaiceHipError(key + " unknown: " + str(cfg[key]))
        setattr(c, attr, table[cfg[key]])
    c.ridgingEfoldingScale = float(cfg["config_ridiging_efolding_scale"])
    c.ridgingWorkRatio = float(cfg["config_ratio_ridging_work_to_PE"])
    c.nCategories = int(block.dimensions["nCategories"])
    c.minEdgeLengthGlobal = float(domain.solver_state.get("dvEdgeMinGlobal", 0.0))
    k = int(cfg.get("config_halo_exchange_interval", 1))
    if k < 1 or (k > 1 and k > int(cfg["config_num_halos"])):
        raise _lib.SeaiceHipError("config_halo_exchange_interval must be between 1 and config_num_halos")
    c.computeHaloVertices = int(k > 1)
    return c


def seaice_mesh_pool_create(domain, device=0, renumber=True):
    """One device context per block.  Unlike the reference (mesh_pool.F:103-105) several blocks per
    process are accepted; ``device`` may be an int or a list with one ordinal per block."""
    for i, b in enumerate(domain.blocklist):
        dev = device[i] if isinstance(device, (list, tuple)) else device
        b.ctx = _lib.BlockContext(b.structs["mesh"], b.dimensions, device=dev, renumber=renumber)


def seaice_mesh_pool_destroy(domain):
    for b in domain.blocklist:
        if getattr(b, "ctx", None) is not None:
            b.ctx.close()
            b.ctx = None


def seaice_mesh_pool_update(domain):
    """mesh_pool.F:261-281: in the reference's directive-based device builds this refreshes the device copies of
    the masks, the ice strength, the velocities and the stresses from the pools before the subcycle loop
    (velocity_solver.F:668).  Here those fields are computed or uploaded on the device already; the call is
    kept for hosts that change them in the pools by hand between the pre-cycle and the loop."""
    for b in domain.blocklist:
        b.ctx.put_many(_items(domain, b, [
            ("velocity_solver", "solveStress"), ("velocity_solver", "solveVelocity"), ("velocity_solver", "icePressure"),
            ("velocity_solver", "uVelocity"), ("velocity_solver", "vVelocity"), ("velocity_variational", "stress11"),
            ("velocity_variational", "stress12"), ("velocity_variational", "stress22")]))


def seaice_init_velocity_solver(domain):
    cfg = domain.configs
    # global minimum edge length (mpas_dmpar_min_real in seaice_init_evp, constitutive_relation.F:132-149)
    dvmin = min(float(b.structs["mesh"]["dvEdge"][:b.dimensions["nEdgesSolve"]].min()) for b in domain.blocklist)
    comm = getattr(domain, "comm", None)
    if comm is not None:
        dvmin = comm.allreduce_min(dvmin)
    domain.solver_state["dvEdgeMinGlobal"] = dvmin
    for b in domain.blocklist:
        vs = b.structs["velocity_solver"]
        vs["dynamicsTimeStep"][0] = cfg["config_dt"] / float(cfg["config_dynamics_subcycle_number"])
        vs["elasticTimeStep"][0] = vs["dynamicsTimeStep"][0] / float(cfg["config_elastic_subcycle_number"])
        if not cfg["config_use_velocity_solver"]:
            continue
        if _weak_strain(domain):
            b.ctx.set_edge_mesh(b.structs["mesh"])
        b.ctx.evp_init(_evp_config(domain, b))
        if cfg["config_variational_basis"] == "none":
            vv = b.structs["velocity_variational"]
            for k in ("basisGradientU", "basisGradientV", "basisIntegralsU", "basisIntegralsV", "basisIntegralsMetric"):
                b.ctx.put(k, vv[k])
            b.ctx.static_changed()
        for pool, name in STATIC_OUTPUTS:
            b.ctx.get(name, b.structs[pool][name])
        if _weak_strain(domain):
            for name in WEAK_STATIC:
                b.ctx.get(name, b.structs["velocity_weak"][name])
    if cfg["config_use_velocity_solver"]:
        # seaice_init calls this before the solver's init (mpas_seaice_initialize.F:91); the device lists need
        # the initialised context, so it runs here
        from .special_boundaries import seaice_init_special_boundaries
        seaice_init_special_boundaries(domain)
    if comm is not None:
        comm.halo_exchange(domain, ["interiorVertex"])
        for b in domain.blocklist:
            b.ctx.get("interiorVertex", b.structs["boundary"]["interiorVertex"])
        if cfg["config_use_velocity_solver"] and cfg.get("config_overlap_halo_exchange", False):
            for b in domain.blocklist:
                b.ctx.overlap_setup()


def _upload(domain, b, skip=()):
    cfg = domain.configs
    items = [(p, n) for p, n in RUN_INPUTS if n not in skip]
    if cfg["config_use_surface_tilt"] and not cfg["config_geostrophic_surface_tilt"]:
        items += [("ocean_coupling", "seaSurfaceTiltU"), ("ocean_coupling", "seaSurfaceTiltV")]
    if not cfg["config_calc_velocity_masks"]:
        items += [("velocity_solver", "solveStress"), ("velocity_solver", "solveVelocity")]
    if cfg["config_use_column_package"] and cfg["config_use_column_vertical_thermodynamics"]:
        # what the column physics has left in the pools for the solver: open water for colpkg_ice_strength
        # (velocity_solver.F:1451), the ice area at the start of the step (:808-823) and the air stress (:1562-1565)
        items += [("icestate", "openWaterArea"), ("icestate", "iceAreaCellInitial"),
                  ("velocity_solver", "airStressCellU"), ("velocity_solver", "airStressCellV")]
    b.ctx.put_many(_items(domain, b, items))


def _download(domain, b):
    items = list(RUN_OUTPUTS)
    if _weak_strain(domain) and not _weak(domain):  # mixed scheme: the weak strains are diagnostics of it
        b.ctx.get_many([(WEAK_FIELDS[k], b.structs["velocity_weak"][k]) for k in ("strain11", "strain22", "strain12")]
                       + [(k, b.structs["velocity_weak_variational"][k])
                          for k in ("strain11Vertex", "strain22Vertex", "strain12Vertex")])
    if domain.configs["config_use_column_package"]:
        items += [("ridging", "ridgeConvergence"), ("ridging", "ridgeShear")]
    b.ctx.get_many(_items(domain, b, items))


def velocity_solver_pre_subcycle(domain):
    comm = getattr(domain, "comm", None)
    ex = comm is not None and domain.configs["config_use_halo_exch"]
    after = {0: ["iceAreaCellInitial", "totalMassCell"], 1: ["solveVelocity"], 2: ["uVelocity", "vVelocity"],
             3: ["icePressure", "airStressCellU", "airStressCellV"], 4: []}
    if not ex:
        for b in domain.blocklist:
            b.ctx.pre_subcycle(-1)
        return
    if int(domain.configs.get("config_halo_exchange_interval", 1)) > 1:
        # with several subcycles between exchanges the stress mask of the outer halo cells, which the
        # reference leaves to local (possibly incomplete) neighbour information, must be the owner's
        after[1] = ["solveVelocity", "solveStress"]
    for stage in range(5):
        for b in domain.blocklist:
            b.ctx.pre_subcycle(stage)
        if after[stage]:
            comm.halo_exchange(domain, after[stage])


def subcycle_velocity_solver(domain, clock=None):
    n = int(domain.configs["config_elastic_subcycle_number"])
    comm = getattr(domain, "comm", None)
    ex = comm is not None and domain.configs["config_use_halo_exch"]
    if not ex:
        # as sih_evp_run does: all iterations but the last replayed from a captured graph (SIH_EVP_GRAPH=0: not)
        graph = n > 2 and os.environ.get("SIH_EVP_GRAPH", "1") != "0"
        for b in domain.blocklist:
            if graph:
                b.ctx.subcycle_graph(n - 1)
                b.ctx.subcycle(1, last=True)
            else:
                b.ctx.subcycle(n, last=True)
        return
    k = int(domain.configs.get("config_halo_exchange_interval", 1))
    names = ["uVelocity", "vVelocity"] if k == 1 else ["uVelocity", "vVelocity", "stress11", "stress22", "stress12"]
    if getattr(comm, "in_library", False) and not _weak_strain(domain) \
            and not domain.configs["config_average_variational_strain"]:
        # the whole loop, exchanges included, runs inside the library (sih_evp_subcycle_exchange)
        if hasattr(comm, "agree_on_exchange_inside") and not hasattr(comm, "exchange_inside"):
            comm.agree_on_exchange_inside(domain)
        for b in domain.blocklist:
            b.ctx.subcycle_exchange(n, last=True, interval=k,
                                    overlap=bool(domain.configs.get("config_overlap_halo_exchange", False)),
                                    use_graph=bool(domain.configs.get("config_subcycle_graph", False)))
        return
    if domain.configs.get("config_overlap_halo_exchange", False) and hasattr(comm, "start_exchange"):
        subcycle_overlapped(domain, comm, n - 1, k, names,
                            use_graph=bool(domain.configs.get("config_subcycle_graph", False)))
        for b in domain.blocklist:
            b.ctx.subcycle(1, last=True)
        comm.halo_exchange(domain, names)
        return
    sb = bool(domain.configs["config_use_special_boundaries_velocity"])
    graph = k > 1 and bool(domain.configs.get("config_subcycle_graph", False))
    done = 0
    while done < n:
        m = min(k, n - done)
        for b in domain.blocklist:
            if graph and m == k and done + m < n:
                b.ctx.subcycle_graph(m)  # the k iterations between two exchanges as one graph launch
            else:
                b.ctx.subcycle(m, last=(done + m == n))
        done += m
        comm.halo_exchange(domain, names)
        if sb:  # seaice_set_special_boundaries_velocity after the halo update, velocity_solver.F:2449-2456
            for b in domain.blocklist:
                b.ctx.apply_special_boundaries()


def subcycle_overlapped(domain, comm, n, k, names, use_graph=False):
    """n iterations (none the last of the dynamics step), exchanging every k of them, with every
    exchange hidden behind work that needs no halo value.

    Around an exchange the iteration is cut into pieces (sih_evp_subcycle_parts): the vertices the
    neighbours need are updated first and the messages start; while they travel the block finishes its
    other vertices and runs the next iteration's strain/stress on the cells that touch no halo vertex;
    after the wait and the unpack it completes that iteration on the remaining cells.  The reference
    exchanges after every iteration with nothing overlapped (velocity_solver.F:2543-2584); the values
    are the same."""
    CA, CI, CR, VA, VS, VR = range(6)
    blocks = domain.blocklist
    pending, done = None, 0
    while done < n:
        m = min(k, n - done)
        if pending is None:
            parts = [CA, VA] * (m - 1) + [CA, VS]
        else:
            pending.finish()  # the first iteration of this chunk already has its interior cells
            parts = [CR, VS] if m == 1 else [CR, VA] + [CA, VA] * (m - 2) + [CA, VS]
        for b in blocks:
            b.ctx.subcycle_parts(parts, use_graph)
        pending = comm.start_exchange(domain, names)
        done += m
        more = done < n
        for b in blocks:
            b.ctx.subcycle_parts([VR, CI] if more else [VR], use_graph)
    if pending is not None:
        pending.finish()


def velocity_solver_post_subcycle(domain):
    comm = getattr(domain, "comm", None)
    ex = comm is not None and domain.configs["config_use_halo_exch"]
    for b in domain.blocklist:
        b.ctx.post_subcycle(0)
    if ex and domain.configs["config_use_ocean_stress"]:
        comm.halo_exchange(domain, ["oceanStressU", "oceanStressV"])
    for b in domain.blocklist:
        b.ctx.post_subcycle(1)


def seaice_run_velocity_solver(domain, clock=None, tracers_on_device=False):
    """``tracers_on_device``: ice area, ice volume and snow volume per category are on the device as the advection
    and the ridging left them (a step whose column physics ran on the device): they are not taken from the pools."""
    if not domain.configs["config_use_velocity_solver"]:
        return
    for b in domain.blocklist:
        _upload(domain, b, skip=("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory") if tracers_on_device else ())
    velocity_solver_pre_subcycle(domain)
    subcycle_velocity_solver(domain, clock)
    velocity_solver_post_subcycle(domain)
    for b in domain.blocklist:
        _download(domain, b)
