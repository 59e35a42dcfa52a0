"""Host-side stand-in for the MPAS block / pool containers the reference solver is written against.

The reference reaches every field through ``MPAS_pool_get_subpool(block % structs, "<pool>")`` and
``MPAS_pool_get_array(pool, "<name>")`` (e.g. /root/reference/src/shared/mpas_seaice_velocity_solver.F:
1218-1236).  Here a pool is a dict name -> numpy array, a block owns a dict of pools plus its
dimensions, and a domain owns the block list and the configs -- enough structure for the host
mirror of ``seaice_init_velocity_solver`` / ``seaice_run_velocity_solver`` to read like the
reference.  Field names, pools and dimension order follow /root/reference/src/Registry.xml
(:2251-2367 mesh, :2439-2524 tracers, :3683-3688 boundary, :3691-3727 velocity_solver,
:3747-3765 velocity_variational).

Arrays are allocated like MPAS does: one extra slot past nCells / nVertices / nEdges so that the
``n+1`` "no neighbour" sentinel can be dereferenced, Fortran-ordered, float64 / int32.
"""
from __future__ import annotations

import numpy as np

# defaults of the namelist options the velocity solver and advection read
# (/root/reference/src/Registry.xml:317-340, 566-670)
DEFAULT_CONFIGS = {
    "config_dt": 3600.0,
    "config_dynamics_subcycle_number": 1,
    "config_elastic_subcycle_number": 120,
    "config_use_velocity_solver": True,
    "config_strain_scheme": "variational",
    "config_stress_divergence_scheme": "variational",
    "config_constitutive_relation_type": "evp",
    "config_variational_basis": "wachspress",
    "config_variational_denominator_type": "original",
    "config_wachspress_integration_type": "dunavant",
    "config_wachspress_integration_order": 8,
    "config_average_variational_strain": False,
    "config_rotate_cartesian_grid": True,
    "config_include_metric_terms": True,
    "config_use_air_stress": True,
    "config_use_ocean_stress": True,
    "config_use_surface_tilt": True,
    "config_geostrophic_surface_tilt": True,
    "config_ocean_stress_type": "quadratic",
    "config_calc_velocity_masks": True,
    "config_use_column_package": False,
    "config_use_column_vertical_thermodynamics": False,
    # ice strength of the column package (Registry.xml:1685-1708), read when both options above are set
    "config_ice_strength_formulation": "Rothrock75",
    "config_ridging_participation_function": "exponential",
    "config_ridging_redistribution_function": "exponential",
    "config_ridiging_efolding_scale": 3.0,
    "config_ratio_ridging_work_to_PE": 17.0,
    # ridging of the column package (Registry.xml:1685-1708; column_ridging runs when both are set)
    "config_use_column_ridging": True,
    "config_use_vertical_tracers": False,   # column_combine_snow_ice_tracers (biogeochemistry): outside this library
    "config_use_ice_age": False,            # the ice-age tendency of seaice_column_update_state: not carried
    "config_use_special_boundaries_velocity": False,
    "config_use_special_boundaries_velocity_masks": False,
    "config_use_special_boundaries_tracers": False,
    "config_use_dynamics": True,
    "config_use_halo_exch": True,
    "config_aggregate_halo_exch": False,
    "config_reuse_halo_exch": False,
    "config_num_halos": 2,
    # not a reference option: number of subcycles between halo exchanges of the velocity solver.  k > 1 needs
    # config_num_halos >= k halo layers; the solver then also updates halo vertices/cells redundantly and
    # exchanges uVelocity, vVelocity and stress11/22/12 every k subcycles (results identical to k = 1)
    "config_halo_exchange_interval": 1,
    # not reference options: hide each of those exchanges behind the subcycle work that needs no halo value
    # (same results), and replay the kernel sequences between exchanges from HIP graphs
    "config_overlap_halo_exchange": False,
    "config_subcycle_graph": False,
    "config_use_advection": True,
    "config_advection_type": "incremental_remap",
    "config_conservation_check": False,
    "config_monotonicity_check": False,
    "config_recover_tracer_means_check": False,
    "config_recover_tracer_means_check": False,
    "config_nIceLayers": 7,
    "config_nSnowLayers": 1,
    "config_earth_radius": 6371229.0,
    "config_do_restart": False,
    "config_calculate_coriolis": True,
}


class Block:
    def __init__(self, mesh, configs, nCategories=5, blockID=0):
        self.blockID = blockID
        self.configs = configs
        self.dimensions = {k: int(mesh[k]) for k in (
            "nCells", "nCellsSolve", "nVertices", "nVerticesSolve", "nEdges", "nEdgesSolve",
            "maxEdges", "vertexDegree")}
        self.dimensions["nCategories"] = nCategories
        self.dimensions["ONE"] = 1
        self.dimensions["nIceLayers"] = int(configs.get("config_nIceLayers", 7))
        self.dimensions["nSnowLayers"] = int(configs.get("config_nSnowLayers", 1))
        self.structs = {}
        self.halo = None  # filled by the partitioner for multi-block runs
        self._alloc(mesh)

    # -- allocation -------------------------------------------------------------------------
    def _alloc(self, mesh):
        d = self.dimensions
        nC, nV, nE, M, D, nCat = d["nCells"], d["nVertices"], d["nEdges"], d["maxEdges"], d["vertexDegree"], d["nCategories"]

        def rc(*lead):
            return np.zeros(tuple(lead) + (nC + 1,), dtype=np.float64, order="F")

        def rv(*lead):
            return np.zeros(tuple(lead) + (nV + 1,), dtype=np.float64, order="F")

        def ic():
            return np.zeros(nC + 1, dtype=np.int32)

        def iv(*lead):
            return np.zeros(tuple(lead) + (nV + 1,), dtype=np.int32, order="F")

        meshp = {"on_a_sphere": mesh["on_a_sphere"], "sphere_radius": mesh["sphere_radius"]}
        if "planarExtent" in mesh:  # block of a partitioned planar mesh (test cases are defined on the whole domain)
            meshp["planarExtent"] = mesh["planarExtent"]

        def pad1(a, fill):
            out = np.full(a.shape[:-1] + (a.shape[-1] + 1,), fill, dtype=a.dtype, order="F")
            out[..., :-1] = a
            return out
        for k in ("xCell", "yCell", "zCell", "latCell", "lonCell", "areaCell", "fCell"):
            meshp[k] = pad1(np.asarray(mesh[k], dtype=np.float64), 0.0)
        for k in ("xVertex", "yVertex", "zVertex", "latVertex", "lonVertex", "areaTriangle", "fVertex",
                  "kiteAreasOnVertex"):
            meshp[k] = pad1(np.asarray(mesh[k], dtype=np.float64), 0.0)
        for k in ("xEdge", "yEdge", "zEdge", "dcEdge", "dvEdge", "latEdge", "lonEdge", "fEdge"):
            meshp[k] = pad1(np.asarray(mesh[k], dtype=np.float64), 0.0)
        meshp["nEdgesOnCell"] = pad1(mesh["nEdgesOnCell"].astype(np.int32), 0)
        meshp["verticesOnCell"] = pad1(mesh["verticesOnCell"].astype(np.int32), nV + 1)
        meshp["edgesOnCell"] = pad1(mesh["edgesOnCell"].astype(np.int32), nE + 1)
        meshp["cellsOnCell"] = pad1(mesh["cellsOnCell"].astype(np.int32), nC + 1)
        meshp["cellsOnVertex"] = pad1(mesh["cellsOnVertex"].astype(np.int32), nC + 1)
        meshp["cellsOnEdge"] = pad1(mesh["cellsOnEdge"].astype(np.int32), nC + 1)
        meshp["verticesOnEdge"] = pad1(mesh["verticesOnEdge"].astype(np.int32), nV + 1)
        if "edgesOnVertex" in mesh:
            meshp["edgesOnVertex"] = pad1(mesh["edgesOnVertex"].astype(np.int32), nE + 1)
        # output of the framework operator mpas_init_reconstruct (called at
        # mpas_seaice_advection_incremental_remap.F:744-746); supplied with the mesh here
        if "coeffs_reconstruct" in mesh:
            meshp["coeffs_reconstruct"] = pad1(np.asarray(mesh["coeffs_reconstruct"], dtype=np.float64), 0.0)
        # global ids (mesh pool, Registry.xml indexToCellID / indexToVertexID); identity for an unpartitioned mesh
        meshp["indexToCellID"] = pad1(np.asarray(mesh["indexToCellID"], dtype=np.int32) if "indexToCellID" in mesh
                                      else np.arange(1, nC + 1, dtype=np.int32), 0)
        meshp["indexToVertexID"] = pad1(np.asarray(mesh["indexToVertexID"], dtype=np.int32)
                                        if "indexToVertexID" in mesh else np.arange(1, nV + 1, dtype=np.int32), 0)
        self.structs["mesh"] = meshp
        # special_boundaries pool (mpas_seaice_special_boundaries.F:121-126, 382-383)
        self.structs["special_boundaries"] = {
            "vertexBoundaryType": iv(), "vertexBoundarySource": iv(), "vertexBoundarySourceLocal": iv(),
            "solveVelocitySpecialBoundaries": iv(), "solveStressSpecialBoundaries": ic(),
            "tracerBoundaryType": ic(), "tracerBoundarySource": ic(), "tracerBoundarySourceLocal": ic()}

        self.structs["boundary"] = {"interiorVertex": iv(), "interiorCell": ic(),
                                    "interiorEdge": np.zeros(nE + 1, dtype=np.int32)}
        vs = {"dynamicsTimeStep": np.zeros(1), "elasticTimeStep": np.zeros(1)}
        for k in ("uVelocity", "vVelocity", "uVelocityInitial", "vVelocityInitial", "stressDivergenceU",
                  "stressDivergenceV", "airStressVertexU", "airStressVertexV", "uOceanVelocityVertex",
                  "vOceanVelocityVertex", "oceanStressU", "oceanStressV", "oceanStressCoeff",
                  "seaSurfaceTiltVertexU", "seaSurfaceTiltVertexV", "surfaceTiltForceU", "surfaceTiltForceV",
                  "totalMassVertexfVertex"):
            vs[k] = rv()
        for k in ("icePressure", "airStressCellU", "airStressCellV", "oceanStressCellU", "oceanStressCellV",
                  "divergence", "shear", "uVelocityCell", "vVelocityCell"):
            vs[k] = rc()
        vs["solveStress"] = ic()
        vs["dynamicallyLockedCellsMask"] = ic()
        vs["solveVelocity"] = iv()
        vs["solveVelocityPrevious"] = iv()
        self.structs["velocity_solver"] = vs

        vv = {}
        for k in ("strain11", "strain22", "strain12", "stress11", "stress22", "stress12",
                  "principalStress1", "principalStress2", "replacementPressure"):
            vv[k] = rc(M)
        for k in ("basisIntegralsU", "basisIntegralsV", "basisIntegralsMetric", "basisGradientU", "basisGradientV"):
            vv[k] = rc(M, M)
        vv["tanLatVertexRotatedOverRadius"] = rv()
        vv["variationalDenominator"] = rv()
        vv["cellVerticesAtVertex"] = iv(D)
        self.structs["velocity_variational"] = vv

        # velocity_weak pool (Registry.xml:3730-3744; keys are the name_in_code values)
        vw = {"normalVectorPolygon": rc(2, M), "normalVectorTriangle": rv(2, D), "latCellRotated": rc(),
              "latVertexRotated": rv()}
        for k in ("strain11", "strain22", "strain12", "stress11", "stress22", "stress12", "principalStress1",
                  "principalStress2", "replacementPressure"):
            vw[k] = rc()
        self.structs["velocity_weak"] = vw
        # velocity_weak_variational pool (Registry.xml:3768-3772)
        self.structs["velocity_weak_variational"] = {k: rv() for k in ("strain11Vertex", "strain22Vertex",
                                                                        "strain12Vertex")}

        self.structs["icestate"] = {k: (rv() if "Vertex" in k else rc()) for k in (
            "iceAreaVertex", "totalMassVertex", "iceAreaCellInitial", "totalMassCell", "openWaterArea")}
        self.structs["tracers_aggregate"] = {k: rc() for k in ("iceAreaCell", "iceVolumeCell", "snowVolumeCell")}
        # the tracers that are always advected (mpas_seaice_advection_incremental_remap_tracers.F:218-224;
        # Registry.xml:2439-2524); optional ones are added to this pool by the caller
        nIce, nSnow = d["nIceLayers"], d["nSnowLayers"]
        self.structs["tracers"] = {k: rc(1, nCat) for k in ("iceAreaCategory", "iceVolumeCategory",
                                                            "snowVolumeCategory", "surfaceTemperature")}
        self.structs["tracers"]["iceEnthalpy"] = rc(nIce, nCat)
        self.structs["tracers"]["iceSalinity"] = rc(nIce, nCat)
        self.structs["tracers"]["snowEnthalpy"] = rc(nSnow, nCat)
        self.structs["incremental_remap"] = {}
        self.structs["ocean_coupling"] = {"uOceanVelocity": rc(), "vOceanVelocity": rc(),
                                          "seaSurfaceTiltU": rc(), "seaSurfaceTiltV": rc(),
                                          "landIceMask": ic(), "landIceMaskVertex": iv()}
        self.structs["atmos_coupling"] = {"uAirVelocity": rc(), "vAirVelocity": rc(), "airDensity": rc()}
        self.structs["ridging"] = {"ridgeConvergence": rc(), "ridgeShear": rc()}


RIDGING_CELL_FIELDS = ("areaLossRidge", "areaGainRidge", "iceVolumeRidged", "openingRateRidge")
RIDGING_CATEGORY_FIELDS = ("ridgeParticipationFunction", "ratioRidgeThicknessToIce", "fractionNewRidgeArea",
                           "fractionNewRidgeVolume", "areaLossRidgeCategory", "areaGainRidgeCategory",
                           "iceVolumeRidgedCategory", "raftingIceArea", "raftingIceVolume")


def category_thickness_limits(nCategories):
    """hin_max(0:ncat) of the original category scheme (kcatbound = 0 with the linear-remapping ITD,
    colpkg_init_itd, /root/reference/src/column/ice_colpkg.F90:175-202): for five categories 0, 0.645, 1.391, 2.470,
    4.567, 9.334 m."""
    r = float(nCategories)
    cc1, cc3 = 3.0 / r, 3.0
    cc2 = 15.0 * cc1
    h = np.zeros(nCategories + 1)
    for n in range(1, nCategories + 1):
        x1 = (n - 1) / r
        h[n] = h[n - 1] + cc1 + cc2 * (1.0 + np.tanh(cc3 * (x1 - 1.0)))
    return h


TRANSPORT_TENDENCY_FIELDS = ("iceAreaTendencyTransport", "iceVolumeTendencyTransport", "iceAgeTendencyTransport")


def allocate_ridging_pools(block):
    """What column_ridging reaches beyond the tracers (/root/reference/src/shared/mpas_seaice_column.F:3418-3456): the
    diagnostics of the `ridging` pool (Registry dims nCells or nCategories x nCells), the ocean fluxes the ridging adds
    to, the freezing temperature of the ocean, the category limits of the `initial` pool and `newlyFormedIce`."""
    if "ocean_fluxes" in block.structs:
        return
    d = block.dimensions
    nC, nCat = d["nCells"], d["nCategories"]
    rg = block.structs["ridging"]
    for k in RIDGING_CELL_FIELDS:
        rg[k] = np.zeros(nC + 1)
    for k in RIDGING_CATEGORY_FIELDS:
        rg[k] = np.zeros((nCat, nC + 1), order="F")
    block.structs["ocean_fluxes"] = {"oceanFreshWaterFlux": np.zeros(nC + 1), "oceanHeatFlux": np.zeros(nC + 1),
                                     "oceanSaltFlux": np.zeros(nC + 1)}
    # the ITD clean-up at the end of colpkg_step_ridge marks categories it emptied (Registry.xml:3962)
    block.structs["biogeochemistry"] = {"newlyFormedIce": np.zeros((nCat, nC + 1), dtype=np.int32, order="F")}
    block.structs["ocean_coupling"]["seaFreezingTemperature"] = np.full(nC + 1, -1.8)
    block.structs["initial"] = {"categoryThicknessLimits": category_thickness_limits(nCat)}
    # what seaice_column_update_state fills after the ridging (:4279-4494): the cell values of the category tracers
    # (`tracers_aggregate`, Registry dims nCells or layers x nCells) and the transport tendencies of `diagnostics`
    ta = block.structs["tracers_aggregate"]
    ta["surfaceTemperatureCell"] = np.zeros(nC + 1)
    ta["iceEnthalpyCell"] = np.zeros((d["nIceLayers"], nC + 1), order="F")
    ta["iceSalinityCell"] = np.zeros((d["nIceLayers"], nC + 1), order="F")
    ta["snowEnthalpyCell"] = np.zeros((d["nSnowLayers"], nC + 1), order="F")
    dg = block.structs.setdefault("diagnostics", {})
    for k in TRANSPORT_TENDENCY_FIELDS:
        dg[k] = np.zeros(nC + 1)


class Domain:
    """``domain % blocklist`` + ``domain % configs`` (one process may hold several blocks)."""

    def __init__(self, meshes, configs=None, nCategories=5):
        cfg = dict(DEFAULT_CONFIGS)
        if configs:
            cfg.update(configs)
        self.configs = cfg
        if not isinstance(meshes, (list, tuple)):
            meshes = [meshes]
        self.blocklist = [Block(m, cfg, nCategories=nCategories, blockID=i) for i, m in enumerate(meshes)]
        self.dminfo = None      # communicator description for multi-process runs
        self.solver_state = {}  # module-level "save" variables of the reference modules


def mpas_pool_get_subpool(block, name):
    return block.structs[name]


def mpas_pool_get_array(pool, name):
    return pool[name]


# tracers the upwind advection knows about (mpas_seaice_advection_upwind.F:1691-1697)
UPWIND_POOL_TRACERS = ("iceAreaCategory", "iceVolumeCategory", "snowVolumeCategory", "surfaceTemperature",
                       "iceEnthalpy", "iceSalinity", "snowEnthalpy")


def allocate_upwind_pools(block):
    """The pools only the upwind advection touches, allocated on first use: the second time level of
    ``tracers`` and ``tracer_conservation`` (time_levs="2", Registry.xml:2439, :3351), ``tracer_tendencies``
    (:2706-2714), ``tracer_edge_fluxes`` (:2788-2795) and ``normalVectorEdge`` of the velocity_solver pool."""
    if "tracers_level2" in block.structs:
        return
    d = block.dimensions
    nC, nE, M, nCat = d["nCells"], d["nEdges"], d["maxEdges"], d["nCategories"]
    tr = block.structs["tracers"]
    block.structs["tracers_level2"] = {k: np.zeros_like(v, order="F") for k, v in tr.items()}
    block.structs["tracer_tendencies"] = {k + "Tend": np.zeros_like(tr[k], order="F") for k in UPWIND_POOL_TRACERS}
    block.structs["tracer_edge_fluxes"] = {
        k + "EdgeFlux": np.zeros(tr[k].shape[:2] + (nE + 1,), order="F") for k in UPWIND_POOL_TRACERS}
    for lev in ("tracer_conservation", "tracer_conservation_level2"):
        block.structs[lev] = {k + "Cons": np.zeros(tr[k].shape[:2], order="F") for k in UPWIND_POOL_TRACERS}
    block.structs["velocity_solver"]["normalVectorEdge"] = np.zeros((2, M, nC + 1), order="F")


def mpas_pool_shift_time_levels(block, pool):
    """MPAS_pool_shift_time_levels for a pool with two time levels: the levels trade places."""
    a, b = block.structs[pool], block.structs[pool + "_level2"]
    for k in a:
        if k in b:
            a[k], b[k] = b[k], a[k]


def pin_pools(domain, pools=None):
    """Move the arrays of the pools (all, or the named ones) into page-locked host memory, contents kept:
    the per-step transfers between pools and device then run at the link's DMA rate.  Optional; needs the
    device library (there is no pageable fallback inside this call)."""
    from . import lib
    for b in domain.blocklist:
        for pname, pool in b.structs.items():
            if pools is not None and pname not in pools:
                continue
            for k, a in list(pool.items()):
                if isinstance(a, np.ndarray) and a.ndim >= 1 and a.size > 1 and a.dtype in (np.float64, np.int32):
                    order = "F" if a.flags["F_CONTIGUOUS"] else "C"
                    p = lib.pinned_empty(a.shape, a.dtype, order=order)
                    p[...] = a
                    pool[k] = p
