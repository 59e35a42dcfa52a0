/*
 * seaice_hip.h -- C ABI of the MI355X-native sea-ice dynamics library (libseaice_hip.so).
 *
 * This is the boundary a host model binds to (Fortran ISO_C_BINDING, ctypes, ...).  It replaces,
 * for one MPAS block per context, the reference's velocity-solver entry points:
 *
 *   seaice_mesh_pool_create / _destroy   src/shared/mpas_seaice_mesh_pool.F:76, 188
 *   seaice_init_velocity_solver          src/shared/mpas_seaice_velocity_solver.F:85
 *   seaice_run_velocity_solver           src/shared/mpas_seaice_velocity_solver.F:562
 *     velocity_solver_pre_subcycle       :613     subcycle_velocity_solver  :2404
 *     velocity_solver_post_subcycle      :3360
 *   mpas_dmpar_field_halo_exch (pack / unpack halves only; the transport stays with the host)
 * and the incremental-remap advection entry points:
 *   seaice_init_advection / seaice_run_advection   src/shared/mpas_seaice_advection.F:40, :90
 *     seaice_init_advection_incremental_remap      src/shared/mpas_seaice_advection_incremental_remap.F:165
 *     seaice_run_advection_incremental_remap       :2338      incremental_remap_block  :2740
 *
 * Conventions
 *   - Plain pointers and sizes only.  Host arrays use the reference's layout: Fortran order, float64
 *     (RKIND) / int32, connectivity 1-based with "nX+1" meaning "no neighbour".  Arrays may be
 *     longer than the logical extent (MPAS allocates nCells+1); only the logical part is touched.
 *   - Fields live on the device under their Registry.xml names (src/Registry.xml:3691-3765 and
 *     the pools listed in sih_field_put).  The library keeps its own renumbered, structure-of-arrays
 *     copies; every put/get converts from/to the reference layout and numbering.
 *   - Every function returns SIH_OK (0) or a positive error code; sih_last_error() gives the text.
 *     An error corresponds to the reference's MPAS_LOG_CRIT abort: the caller must not continue.
 *   - One host thread drives a context at a time (the reference solver is not re-entrant either:
 *     module-level save variables, src/shared/mpas_seaice_velocity_solver.F:26-58).
 */
#ifndef SEAICE_HIP_H
#define SEAICE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sih_ctx sih_ctx;

enum {
    SIH_OK = 0,
    SIH_ERR_ARG = 1,        /* bad argument / unknown option (reference: "... unknown", MPAS_LOG_CRIT) */
    SIH_ERR_HIP = 2,        /* HIP runtime failure */
    SIH_ERR_STATE = 3,      /* call out of order (e.g. run before init) */
    SIH_ERR_UNSUPPORTED = 4, /* option of the reference that is outside this library's scope */
    SIH_ERR_IR = 5,         /* a run-time abort of the advection (negative mass / area, degenerate geometry) */
    SIH_ERR_COMM = 6,       /* a neighbour's halo message did not arrive within the time limit */
    SIH_ERR_COLUMN = 7      /* a run-time abort of the column package's ridging (its l_stop with stop_label) */
};

/* Mesh description of one block: the `mesh` pool fields the dynamics read
 * (src/Registry.xml:2251-2367).  All pointers are borrowed for the duration of the call. */
typedef struct sih_mesh_desc {
    int32_t nCells, nCellsSolve;
    int32_t nVertices, nVerticesSolve;
    int32_t nEdges, nEdgesSolve;
    int32_t maxEdges, vertexDegree;
    int32_t onASphere;          /* mesh attribute on_a_sphere */
    double sphereRadius;        /* mesh attribute sphere_radius (after rescale_mesh) */
    const int32_t *nEdgesOnCell;   /* (nCells) */
    const int32_t *verticesOnCell; /* (maxEdges, nCells) */
    const int32_t *cellsOnCell;    /* (maxEdges, nCells) */
    const int32_t *edgesOnCell;    /* (maxEdges, nCells), may be NULL if advection is not used */
    const int32_t *cellsOnVertex;  /* (vertexDegree, nVertices) */
    const int32_t *cellsOnEdge;    /* (2, nEdges), may be NULL */
    const int32_t *verticesOnEdge; /* (2, nEdges), may be NULL */
    const double *xCell, *yCell, *zCell;       /* (nCells) */
    const double *xVertex, *yVertex, *zVertex; /* (nVertices) */
    const double *areaCell;                    /* (nCells) */
    const double *areaTriangle;                /* (nVertices) */
    const double *fVertex;                     /* (nVertices) */
    const double *dvEdge, *dcEdge;             /* (nEdges), may be NULL */
    int32_t renumber; /* 1: library renumbers cells/vertices internally for locality; 0: keep order */
} sih_mesh_desc;

/* Options read by the velocity solver (src/shared/mpas_seaice_velocity_solver.F:103-135, 792-798,
 * 1378-1383; defaults in src/Registry.xml:566-647). */
typedef struct sih_evp_config {
    double dt;                        /* config_dt */
    int32_t dynamicsSubcycleNumber;   /* config_dynamics_subcycle_number */
    int32_t elasticSubcycleNumber;    /* config_elastic_subcycle_number */
    int32_t constitutiveRelation;     /* 1 evp, 2 evp_revised, 3 linear */
    int32_t variationalBasis;         /* 1 wachspress, 2 pwl (both computed on device), 0 none (host supplies it) */
    int32_t variationalDenominator;   /* 0 original, 1 alternate */
    int32_t wachspressIntegrationOrder; /* order of the rule; dunavant 8 is the reference default */
    int32_t rotateCartesianGrid;
    int32_t includeMetricTerms;
    int32_t useAirStress;
    int32_t useOceanStress;
    int32_t oceanStressType;          /* 1 quadratic, 2 linear */
    int32_t surfaceTilt;              /* 0 none, 1 geostrophic, 2 ssh gradient */
    int32_t calcVelocityMasks;
    int32_t useColumnPackage;         /* config_use_column_package: ridgeConvergence / ridgeShear output; with
                                         useColumnVerticalThermodynamics the ice strength is the column
                                         package's (see below) */
    int32_t nCategories;
    double minEdgeLengthGlobal;       /* global min of dvEdge (mpas_dmpar_min_real in seaice_init_evp);
                                         <= 0: use this block's own minimum */
    int32_t computeHaloVertices;      /* 0: vertex loops cover the owned vertices only, as in the reference
                                         (halo values arrive by exchange).  1: they also cover every halo
                                         vertex whose three cells are inside the block; the values equal the
                                         owner's bit for bit and let a block with H halo layers run H
                                         subcycles between exchanges of uVelocity, vVelocity and stress11/22/12 */
    int32_t averageVariationalStrain; /* config_average_variational_strain (velocity_solver.F:2756-2766) */
    int32_t useSpecialBoundariesVelocity;      /* config_use_special_boundaries_velocity */
    int32_t useSpecialBoundariesVelocityMasks; /* config_use_special_boundaries_velocity_masks */
    int32_t strainScheme;             /* config_strain_scheme: 0 variational, 1 weak */
    int32_t stressDivergenceScheme;   /* config_stress_divergence_scheme: 0 variational, 1 weak.  Weak strain
                                         with variational divergence interpolates the cell strains to the
                                         corners (velocity_solver.F:2768-2784); variational strain with weak
                                         divergence is an error in the reference too (:191-194) */
    /* Ice strength (velocity_solver.F:1419-1458).  Without the column package, or without its vertical
     * thermodynamics, Hibler's P* h exp(-C* (1 - a)).  With both, colpkg_ice_strength
     * (src/column/ice_colpkg.F90:3498-3610) from the category areas and volumes and the field openWaterArea,
     * including its ridge_itd (src/column/ice_mechred.F90:741-1012).  In that setting the fields
     * iceAreaCellInitial and airStressCellU / airStressCellV are inputs too: the reference's column physics
     * sets them before the velocity solver runs (velocity_solver.F:808-823, :1562-1565). */
    int32_t useColumnVerticalThermodynamics; /* config_use_column_vertical_thermodynamics */
    int32_t iceStrengthFormulation;   /* config_ice_strength_formulation (kstrength): 0 Hibler79, 1 Rothrock75 */
    int32_t ridgingParticipation;     /* config_ridging_participation_function (krdg_partic): 0 Thorndike75, 1 exponential */
    int32_t ridgingRedistribution;    /* config_ridging_redistribution_function (krdg_redist): 0 Hibler80, 1 exponential */
    double ridgingEfoldingScale;      /* config_ridiging_efolding_scale (mu_rdg), 3.0 */
    double ridgingWorkRatio;          /* config_ratio_ridging_work_to_PE (Cf), 17.0 */
    int32_t wachspressIntegrationType; /* config_wachspress_integration_type: 0 or 1 dunavant (orders 1-10, 12),
                                          2 trapezoidal (orders 1..21), 3 fekete (orders 1-6, 8, 9);
                                          wachspress.F:1223-1941 */
} sih_evp_config;

/* ---- life cycle ----------------------------------------------------------------------------- */
/* replaces seaice_mesh_pool_create (mesh_pool.F:76): uploads connectivity/geometry, builds the
 * internal numbering.  `device` is the HIP device ordinal. */
int sih_block_create(const sih_mesh_desc *mesh, int device, sih_ctx **ctx);
/* number of HIP devices this process sees (0 when there is none): a host with one rank per GPU takes
 * device = node-local rank modulo this count */
int sih_device_count(void);
/* replaces seaice_mesh_pool_destroy (mesh_pool.F:188) */
int sih_block_destroy(sih_ctx *ctx);
const char *sih_last_error(void);
/* run on a caller-owned HIP stream (hipStream_t); NULL restores the context's own stream */
int sih_set_stream(sih_ctx *ctx, void *hipStream);
int sih_sync(sih_ctx *ctx);

/* Page-locked host memory (hipHostMalloc) for arrays that travel every step: sih_field(s)_put / _get from
 * and to it run at the link's DMA rate instead of the pageable-memory rate (about 2x up, 3x down on an
 * MI355X box).  No counterpart in the reference, where the framework allocates the pools; optional. */
int sih_host_alloc(void **ptr, size_t bytes);
int sih_host_free(void *ptr);

/* ---- initialisation ------------------------------------------------------------------------- */
/* Edge description of the mesh for the weak scheme (velocity_solver_weak.F; its normal vectors
 * mesh.F:703-1606 need edge positions and the edge <-> cell / vertex connectivity): the same struct the
 * advection takes; nCategories and rotateCartesianGrid are ignored here.  Call before sih_evp_init when
 * sih_evp_config.strainScheme = 1; arrays are copied. */
struct sih_ir_desc;
int sih_evp_set_edge_mesh(sih_ctx *ctx, const struct sih_ir_desc *edges);
/* replaces the per-block part of seaice_init_velocity_solver (velocity_solver.F:85-252) and
 * interior_vertices (mesh.F:423): time steps, EVP constants, metric terms, cellVerticesAtVertex,
 * local coordinates, Wachspress basis gradients/integrals, variational denominator.
 * One deviation from the reference's bits, by default: basisIntegralsMetric(s, j), the integral of N_s N_j, which the
 * reference rounds separately for (s, j) and (j, s) (wachspress.F:442-457), is made symmetric bit for bit by mirroring
 * the entries with s <= j after the variational denominator has been formed (the other half changes in the last place
 * at most), so that the subcycle kernels fetch one side of the diagonal.  SIH_METRIC_INTEGRALS=as_computed in the
 * environment keeps all entries as computed; a host-supplied basis (variationalBasis 0) is used as it is.
 * A second one inside the subcycle kernel only: in cells where the three non-zero basis gradients of every corner sum
 * to zero to 1e-13 of the largest (the basis functions sum to one), the corner's own gradient is formed as minus the
 * sum of the other two instead of being fetched; basisGradientU / V are untouched.  SIH_BASIS_GRADIENTS=as_computed
 * switches that off.
 * A third, also inside the subcycle kernels only (all their forms alike): the inner sums of the stress divergence
 * (variational.F:1151-1170) add the terms with basisIntegralsU / V and the metric terms separately, and in cells where
 * basisIntegralsU / V summed over the velocity corners vanish to 1e-13 of the largest entry for every stress corner
 * (again the basis functions summing to one) the first corner's sum is its metric terms minus the other corners' sums,
 * so that its rows of basisIntegralsU / V are not fetched; the arrays are untouched.  SIH_BASIS_INTEGRALS=as_computed
 * keeps the reference's statement.  All three keep the velocities within 1e-10 of the reference arithmetic. */
int sih_evp_init(sih_ctx *ctx, const sih_evp_config *cfg);
/* re-pack the static basis after the host replaced basisGradient* / basisIntegrals* with its own
 * (e.g. the piecewise-linear basis computed by the host model) */
int sih_evp_static_changed(sih_ctx *ctx);

/* Special boundaries of idealised domains (special_boundaries.F:253-401): before the first and after
 * every subcycle iteration uVelocity/vVelocity of the marked vertices are replaced by those of a source
 * vertex (type 1 periodic: copy, 2 reverse: negated) or by zero (type 3), and solveVelocity / solveStress
 * by the given masks.  Arrays in the reference layout (nVertices / nCells entries, sources 1-based local
 * indices = vertexBoundarySourceLocal); the reference's loop is sequential and in place, and a source that
 * is itself a boundary vertex is resolved the same way here.  A NULL pair leaves that part off; which
 * parts are used is set by sih_evp_config.useSpecialBoundaries*.  Call after sih_evp_init. */
int sih_evp_set_special_boundaries(sih_ctx *ctx, const int32_t *vertexBoundaryType,
                                   const int32_t *vertexBoundarySourceLocal,
                                   const int32_t *solveVelocitySpecialBoundaries,
                                   const int32_t *solveStressSpecialBoundaries);
/* Several blocks: seaice_set_special_boundaries_velocity follows the halo update that ends an iteration
 * (velocity_solver.F:2449-2456).  sih_evp_defer_special_boundaries(ctx, 1) takes the application out of
 * sih_evp_subcycle; the host then exchanges uVelocity, vVelocity after every iteration and calls
 * sih_evp_apply_special_boundaries.  A marked vertex whose vertexBoundarySourceLocal is 0 (its source is not in
 * this block) is left as the exchange delivered it. */
int sih_evp_defer_special_boundaries(sih_ctx *ctx, int on);
int sih_evp_apply_special_boundaries(sih_ctx *ctx);

/* ---- fields --------------------------------------------------------------------------------- */
/* Copy a named field host -> device (put) or device -> host (get) in the reference layout.
 * `count` is the logical element count and is checked.  Names (pool : fields):
 *   velocity_solver : uVelocity vVelocity uVelocityInitial vVelocityInitial solveStress solveVelocity
 *       solveVelocityPrevious icePressure stressDivergenceU stressDivergenceV airStressCellU
 *       airStressCellV airStressVertexU airStressVertexV uOceanVelocityVertex vOceanVelocityVertex
 *       oceanStressU oceanStressV oceanStressCoeff oceanStressCellU oceanStressCellV
 *       seaSurfaceTiltVertexU seaSurfaceTiltVertexV surfaceTiltForceU surfaceTiltForceV
 *       totalMassVertexfVertex divergence shear
 *   velocity_variational : strain11 strain22 strain12 stress11 stress22 stress12 replacementPressure
 *       principalStress1 principalStress2 (maxEdges,nCells); basisGradientU basisGradientV
 *       basisIntegralsU basisIntegralsV basisIntegralsMetric (maxEdges,maxEdges,nCells);
 *       tanLatVertexRotatedOverRadius variationalDenominator (nVertices);
 *       cellVerticesAtVertex (vertexDegree,nVertices)
 *   icestate : iceAreaVertex totalMassVertex iceAreaCellInitial totalMassCell
 *   tracers_aggregate : iceAreaCell iceVolumeCell snowVolumeCell
 *   tracers : iceAreaCategory iceVolumeCategory snowVolumeCategory (ONE,nCategories,nCells)
 *   ocean_coupling : uOceanVelocity vOceanVelocity seaSurfaceTiltU seaSurfaceTiltV landIceMask
 *       landIceMaskVertex
 *   atmos_coupling : uAirVelocity vAirVelocity airDensity
 *   boundary : interiorVertex          ridging : ridgeConvergence ridgeShear */
int sih_field_put(sih_ctx *ctx, const char *name, const void *host, size_t count);
int sih_field_get(sih_ctx *ctx, const char *name, void *host, size_t count);
/* The same for nFields fields in one call: all copies and layout kernels are queued back to back and
 * the stream is waited for once, instead of once per field.  This is what a host that refreshes a
 * whole pool per time step (the argument lists of the mpas_pool_get_array blocks in
 * mpas_seaice_velocity_solver.F:613-671 and :2870-2960) should call.  Nothing is moved when any
 * name or count is wrong. */
int sih_fields_put(sih_ctx *ctx, int nFields, const char *const *names, const void *const *hosts,
                   const size_t *counts);
int sih_fields_get(sih_ctx *ctx, int nFields, const char *const *names, void *const *hosts, const size_t *counts);
/* A download that drains behind whatever the host starts next.  No counterpart in the reference (its pools live in
 * host memory); it serves the order of seaice_timestep (mpas_seaice_time_integration.F:147-153): the fields of the
 * velocity solver that nothing reads before the end of the step (strains, replacement pressure, principal stresses,
 * masks, forcing terms: what the analysis members and output streams take from the pools) travel while
 * seaice_run_advection computes.  sih_fields_get_start queues the read of the fields behind the work queued on the
 * context's stream so far and returns; work queued afterwards waits for that read (a few layout kernels), not for the
 * copies to the host.  The host arrays must stay allocated and untouched until sih_fields_get_finish has returned; one
 * such download at a time (a second start first waits for the one before).  Page-locked arrays (sih_host_alloc) make
 * the copies asynchronous; from ordinary memory the call may take as long as sih_fields_get. */
int sih_fields_get_start(sih_ctx *ctx, int nFields, const char *const *names, void *const *hosts, const size_t *counts);
int sih_fields_get_finish(sih_ctx *ctx);
/* sih_evp_run (below) with such a download started between the pre-cycle and the iterations, for the fields that
 * velocity_solver_pre_subcycle (mpas_seaice_velocity_solver.F:613-671) computes and nothing later in the step changes:
 * solveStress solveVelocity icePressure airStressCellU/V airStressVertexU/V uOceanVelocityVertex vOceanVelocityVertex
 * surfaceTiltForceU/V totalMassVertexfVertex iceAreaVertex totalMassVertex iceAreaCellInitial totalMassCell
 * iceAreaCell iceVolumeCell snowVolumeCell uVelocityInitial vVelocityInitial.  They reach the host while the subcycle
 * loop runs. */
int sih_evp_run_early_outputs(sih_ctx *ctx, int nFields, const char *const *names, void *const *hosts,
                              const size_t *counts);
/* Hand-off to the column package between advection and ridging (seaice_column_dynamics_time_integration,
 * mpas_seaice_time_integration.F:158 -> column_ridging, mpas_seaice_column.F:3391-3560), which works cell by cell
 * on the owned cells and takes, per cell, ridgeConvergence / ridgeShear (written by the velocity solver's
 * final_divergence_shear, mpas_seaice_velocity_solver_variational.F:1300-1319), the category arrays
 * iceAreaCategory(1,:,iCell) ... and trcrn(nTracers, nCategories) assembled from the tracer fields by
 * set_cice_physics_tracer_array_category (mpas_seaice_column.F:6534-6575).  Instead of downloading every tracer
 * field whole (halo cells included) and assembling on the host, sih_column_get gathers on the device, for host
 * cells [firstCell, firstCell + nCells) (0-based; the owned cells are 0 .. nCellsSolve-1), one record per cell:
 *     host[((cell - firstCell) * nCategories + category) * S + slot]
 * where the listed fields' layers follow each other in list order (S = sum of their layer counts; a field of
 * leading size lead has lead / nCategories layers).  In Fortran that is host(S, nCategories, nCells): host(:,:,i)
 * is colpkg_step_ridge's trcrn for cell i when the names are given in the reference's tracer order, and with
 * nCategories = 1 and the names ridgeConvergence, ridgeShear it is host(2, 1, nCells).  sih_column_put is the way
 * back (get_cice_physics_tracer_array_category, :6723-6760): the device fields of those cells take the values of
 * the records, so the next velocity solve and advection see what the ridging left, and nothing else is touched.
 * count = nCells * nCategories * S, checked. */
int sih_column_get(sih_ctx *ctx, int nFields, const char *const *names, int nCategories, int firstCell, int nCells,
                   double *host, size_t count);
int sih_column_put(sih_ctx *ctx, int nFields, const char *const *names, int nCategories, int firstCell, int nCells,
                   const double *host, size_t count);
/* The ridging itself on the device: ridge_ice (src/column/ice_mechred.F90:80-597, with ridge_prep :640, ridge_itd
 * :741, ridge_shift :1030 and colpkg_compute_tracers, src/column/ice_colpkg_tracers.F90:163) for every owned cell, as
 * column_ridging runs it through colpkg_step_ridge (src/shared/mpas_seaice_column.F:3468-3530,
 * src/column/ice_colpkg.F90:3296-3325).  Works on the device fields of the advected tracers (sih_ir_set_tracers must
 * have been called: iceAreaCategory, iceVolumeCategory, snowVolumeCategory and every tracer carried on one of them --
 * surfaceTemperature, iceEnthalpy, iceSalinity, snowEnthalpy of the standard physics; a tracer carried on another
 * tracer returns SIH_ERR_UNSUPPORTED) and on these cell fields, which the call creates if they do not exist yet:
 *   in:     ridgeConvergence ridgeShear (left by the velocity solver with config_use_column_package)
 *           seaFreezingTemperature (ocean_coupling pool)
 *   in/out: openWaterArea (icestate)  oceanFreshWaterFlux oceanHeatFlux (ocean_fluxes)
 *           fractionNewRidgeArea fractionNewRidgeVolume (nCategories,nCells; ridging pool)
 *   out:    areaLossRidge areaGainRidge iceVolumeRidged openingRateRidge (nCells), ridgeParticipationFunction
 *           ratioRidgeThicknessToIce areaLossRidgeCategory areaGainRidgeCategory iceVolumeRidgedCategory raftingIceArea
 *           raftingIceVolume (nCategories,nCells)
 * so that velocity solver -> advection -> ridging -> next velocity solver runs without a copy to the host.  Where the
 * reference sets l_stop the call returns SIH_ERR_COLUMN with the reference's stop_label as the message
 * ("Ridging error: aice0 < 0", "Ridging error: ardg > aicen", "ridge_ice: Exceeded max number of ridging iterations",
 * "ridge_ice: total area > 1", "ice: ITD cleanup error in colpkg_step_ridge").  With itdCleanup set the ITD clean-up
 * that colpkg_step_ridge runs after ridge_ice (cleanup_itd, src/column/ice_itd.F90:754-1020) follows in the same
 * launch; it adds
 *   in/out: oceanSaltFlux (ocean_fluxes), newlyFormedIce (int32, nCategories,nCells; biogeochemistry pool: set to 1
 *           for a category the clean-up emptied, otherwise left as it is)
 *   out:    iceAreaCell (tracers_aggregate)
 * The aggregation of the other cell means (seaice_column_update_state) stays with the host's column package. */
typedef struct {
    double dt;                  /* dynamicsTimeStep */
    int32_t dynamicsSubcycleNumber; /* config_dynamics_subcycle_number (ndtd: the fluxes are averaged over the thermodynamic step) */
    int32_t participation;      /* config_ridging_participation_function: 0 Thorndike75, 1 exponential */
    int32_t redistribution;     /* config_ridging_redistribution_function: 0 Hibler80, 1 exponential */
    double efoldingScale;       /* config_ridiging_efolding_scale (mu_rdg) */
    int32_t nCategories;
    const double *categoryThicknessLimits; /* hin_max(0:nCategories) of the `initial` pool, host memory */
    int32_t itdCleanup;         /* 1: colpkg_step_ridge whole, i.e. the ITD clean-up (cleanup_itd, ice_itd.F90:754-1020:
                                   rebin, zap_small_areas, zap_snow_temperature) after ridge_ice; 0: ridge_ice alone */
} sih_ridge_config;
int sih_column_ridge(sih_ctx *ctx, const sih_ridge_config *cfg);
/* creates the cell fields listed above (zero-filled) without running anything, so that a host can put its ocean state
 * and accumulated fluxes before the first sih_column_ridge */
int sih_column_ridge_init(sih_ctx *ctx, int nCategories);
/* seaice_column_update_state(domain, "transport", dt, 0) (src/shared/mpas_seaice_column.F:4279-4377), which follows the
 * ridging in seaice_column_dynamics_time_integration (:1139-1194): for every owned cell colpkg_aggregate
 * (src/column/ice_colpkg.F90:3366-3480, through seaice_column_aggregate :4391-4494) -- iceAreaCell, iceVolumeCell,
 * snowVolumeCell, openWaterArea and the cell value of every category tracer, written to the cell field
 * "<tracer>Cell" with the tracer's layers (surfaceTemperatureCell, iceEnthalpyCell, iceSalinityCell, snowEnthalpyCell
 * of the `tracers_aggregate` pool; created on first use) -- and, with tendencies != 0, the transport tendencies
 * iceAreaTendencyTransport = (iceAreaCell - iceAreaTendencyTransport) / dt and the same for the volume (:4352-4355;
 * the ice age, config_use_ice_age, is not carried).  Same tracer set as sih_column_ridge; asynchronous on the
 * context's stream. */
int sih_column_update_state(sih_ctx *ctx, int nCategories, double dt, int tendencies);
/* seaice_column_reinitialize_diagnostics_dynamics (:13569-13723) for the fields of this path, at the start of a
 * dynamics subcycle: areaLossRidge, areaGainRidge, iceVolumeRidged, openingRateRidge = 0; iceAreaTendencyTransport =
 * iceAreaCell, iceVolumeTendencyTransport = iceVolumeCell, iceAgeTendencyTransport = 0 (whole arrays).  The velocity
 * solver's diagnostics the reference zeroes there as well are all set by sih_evp_run. */
int sih_column_reinitialize_diagnostics(sih_ctx *ctx, int nCategories);
/* sih_field_get for callers that cannot allocate: code under HIP graph capture elsewhere in the process, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The layout conversion is
 * staged in a 16 MiB array that is part of the library's code object on every GPU, so the call makes no
 * hipMalloc / hipFree (sih_field_get grows its staging buffer on demand); a field of more than 16 MiB is refused
 * with SIH_ERR_ARG.  All calls on one GPU share that array: calls that overlap on one GPU (two host threads, two
 * contexts) race. */
int sih_field_get_noalloc(sih_ctx *ctx, const char *name, void *host, size_t count);
/* logical element count of a field, and 8 (float64) or 4 (int32) in *elemSize; 0 if unknown */
size_t sih_field_count(sih_ctx *ctx, const char *name, int *elemSize);

/* ---- time step ------------------------------------------------------------------------------ */
/* velocity_solver_pre_subcycle (velocity_solver.F:613-671) split where the reference exchanges
 * halos, so that a multi-block host can exchange between stages:
 *   stage 0: aggregate_mass_and_area, iceAreaCellInitial          -> halo iceAreaCellInitial,totalMassCell
 *   stage 1: cell->vertex area/mass, stress and velocity masks    -> halo solveVelocity
 *   stage 2: ocean velocity to vertices, new_ice_velocities       -> halo uVelocity,vVelocity
 *   stage 3: ice_strength (Hibler), constant_air_stress           -> halo icePressure, airStressCellU/V
 *   stage 4: air stress to vertices, coriolis, ocean_stress, surface_tilt, init_subcycle_variables
 * stage -1 runs all of them back to back (single block). */
int sih_evp_pre_subcycle(sih_ctx *ctx, int stage);
/* n iterations of single_subcycle_velocity_solver (velocity_solver.F:2478-2592) without the halo
 * exchange: strain, EVP stress, stress divergence, ocean stress coefficient, velocity solve.
 * `last` != 0 marks the final iteration of the dynamics step: strain, replacementPressure,
 * stressDivergenceU/V and oceanStressCoeff are only materialised then (they are overwritten every
 * iteration and read by nothing else in between). */
int sih_evp_subcycle(sih_ctx *ctx, int n, int last);
/* the same n iterations (last = 0) replayed from a HIP graph captured on first use; asynchronous on the
 * context's stream.  For blocks so small that launch overhead dominates (many-GPU strong scaling). */
int sih_evp_subcycle_graph(sih_ctx *ctx, int n);
/* the same n iterations in ONE launch of a persistent kernel: every workgroup keeps a tile of 64 cells and all the
 * vertices of those cells on the chip for all iterations (basis gradients, stresses and vertex constants in
 * registers, integral rows and velocities in LDS); only the per-corner sums of the stress divergence pass through
 * memory, and a vertex waits only for the sums it reads.  For blocks small enough to be resident at once (two
 * workgroups a CU: 32 768 cells on 256 CUs; with maxEdges 6 also tiles of 42 cells, four workgroups a CU, the rows
 * of the metric integrals read from memory in every iteration: 43 008 cells; sih_get_info "persistentKernel" = 1,
 * "persistentTileCells"), where the two-kernel form is bound by its kernel boundaries and by re-reading tables that
 * never change.  Same bits as sih_evp_subcycle.
 * sih_evp_subcycle_graph, sih_evp_run and sih_evp_subcycle_exchange use it by themselves for loops of two or more
 * iterations where a block qualifies and its context is the only one of the process on the device (the launches
 * of other blocks would take the places of its workgroups); the environment variable SIH_PERSISTENT=0 switches that off;
 * sih_evp_subcycle_exchange then exchanges between the chunks without overlap.  Returns SIH_ERR_UNSUPPORTED where
 * the block does not qualify; a launch that cannot complete (its waits are bounded: SIH_PERSIST_TIMEOUT_MS,
 * default 2000) is reported by sih_sync / sih_evp_run as SIH_ERR_STATE.  The GPU must not be shared with other
 * work that could keep some of the workgroups from being resident. */
int sih_evp_subcycle_persistent(sih_ctx *ctx, int n);
/* the same n iterations (last = 0) as n + 1 launches of one kernel over the same tiles: a launch is "the vertices
 * of iteration t - 1, then the cells of iteration t" (velocity_solver.F:2478-2592 cut between the stress divergence's
 * inner and outer sums instead of after the velocity solve), the first launch the cells only, the last the vertices
 * only.  For blocks too large to stay resident in the persistent kernel, or not alone on their device: half the
 * launches of the two-kernel form and a shorter chain of dependent memory accesses in each.  Same
 * bits as sih_evp_subcycle; nothing waits inside a launch, so it has no residency condition and no time limit.
 * sih_evp_subcycle_graph, sih_evp_run and sih_evp_subcycle_exchange (between its exchanges, without overlap) choose
 * it by themselves where the persistent kernel does not apply and the block has at most SIH_FUSED_MAX_CELLS cells
 * (sih_get_info "fusedKernel" = 1: default scheme, compact gradients, at most 64 * maxEdges vertices in a tile of 64
 * cells); mode 0 of sih_evp_set_persistent and SIH_FUSED=0 keep the two-kernel form.  Returns SIH_ERR_UNSUPPORTED
 * where the block does not qualify. */
int sih_evp_subcycle_fused(sih_ctx *ctx, int n);
/* which form sih_evp_subcycle_graph, sih_evp_run and sih_evp_subcycle_exchange choose for this block where it
 * qualifies: 0 the two-kernel form; 1 (the default) the persistent kernel with the halo exchanges between its
 * launches; 2 the persistent kernel with the exchanges of sih_evp_subcycle_exchange inside it: one launch for all
 * iterations but the last, in which the tiles at the block's edge store their velocities and stresses straight into
 * the neighbours' memory and poll for theirs (for at most SIH_HALO_TIMEOUT_MS, default 10 s, then the launch ends
 * with an error), and the tiles inside never wait for a neighbour.  Mode 2 speaks
 * another protocol than the exchange between launches, so all blocks that exchange with each other must run it or
 * none: the host asks every block (sih_get_info "persistentExchangeReady" = 1 when the block qualifies and its
 * transport is connected), takes the minimum over the ranks and sets mode 2 on all of them or on none.  -1 goes
 * back to the environment (SIH_PERSISTENT=0 is mode 0, SIH_PERSIST_EXCHANGE=1 is mode 2). */
int sih_evp_set_persistent(sih_ctx *ctx, int mode);
/* Pieces of one subcycle iteration (none the last of the dynamics step), for a host that overlaps the
 * halo exchange the reference does after every iteration (velocity_solver.F:2543-2584) with work that
 * needs no halo value.  sih_evp_overlap_setup derives four entity subsets from the lists given to
 * sih_halo_set_lists (call it after them and after sih_evp_init):
 *   interior cells : no vertex in the vertex receive list and not in the cell receive list
 *   rim cells      : the others
 *   send vertices  : updated vertices that are in the vertex send list
 *   rest vertices  : the other vertices the vertex kernel updates
 * sih_evp_subcycle_parts queues the given sequence of pieces on the context's stream (useGraph != 0:
 * as one HIP graph, captured per distinct sequence on first use).  A typical sequence around an
 * exchange is  CELLS_ALL, VERTICES_SEND | pack, start transfer | VERTICES_REST, CELLS_INTERIOR |
 * wait, unpack | CELLS_RIM, VERTICES_ALL ...  Every entity is computed once per iteration from the
 * same inputs as by sih_evp_subcycle, so the results are identical. */
enum {
    SIH_PART_CELLS_ALL = 0,
    SIH_PART_CELLS_INTERIOR = 1,
    SIH_PART_CELLS_RIM = 2,
    SIH_PART_VERTICES_ALL = 3,
    SIH_PART_VERTICES_SEND = 4,
    SIH_PART_VERTICES_REST = 5
};
int sih_evp_overlap_setup(sih_ctx *ctx);
int sih_evp_subcycle_parts(sih_ctx *ctx, int nParts, const int32_t *parts, int useGraph);
/* The reference's two public operators on their own, as its operator test cases
 * (testing_and_setup/testcases/spherical_operators/{strain,stress_divergence}, square/operators_strain) drive them:
 *   sih_evp_strain_tensor      seaice_strain_tensor_variational (velocity_solver_variational.F:575-670): strain11,
 *                              strain22, strain12 at the corners of every cell with solveStress = 1, from
 *                              uVelocity / vVelocity
 *   sih_evp_stress_divergence  seaice_stress_divergence_variational (:1064-1184): stressDivergenceU / V at the owned
 *                              vertices with solveVelocity = 1, from stress11, stress22, stress12 as they are
 * Both are asynchronous on the context's stream; the masks are whatever the fields hold. */
int sih_evp_strain_tensor(sih_ctx *ctx);
int sih_evp_stress_divergence(sih_ctx *ctx);
/* velocity_solver_post_subcycle (velocity_solver.F:3360-3380):
 *   stage 0: final_divergence_shear, principal stresses, ocean stress on vertices -> halo oceanStressU/V
 *   stage 1: ocean stress vertex->cell, rescale;  stage -1: both */
int sih_evp_post_subcycle(sih_ctx *ctx, int stage);
/* whole seaice_run_velocity_solver for a block that needs no halo exchange; all iterations but the last are
 * replayed from a HIP graph captured at the first call (environment SIH_EVP_GRAPH=0: every launch separate) */
int sih_evp_run(sih_ctx *ctx);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Like sih_evp_subcycle(n, 0) but brackets the launches with HIP events on the context's stream:
 * totalMs = whole loop; every `sampleEvery`-th iteration additionally times its two kernels
 * (cellMs / vertexMs = mean duration of the sampled launches).  useGraph != 0 replays a captured
 * HIP graph of the n iterations instead of launching them one by one (no per-kernel samples then). */
int sih_evp_subcycle_timed(sih_ctx *ctx, int n, int useGraph, int sampleEvery, double *totalMs,
                           double *cellMs, double *vertexMs);

/* ---- halo exchange (pack / unpack only) -------------------------------------------------------
 * location: 0 cells, 1 vertices (pack / unpack also take 2: cell and vertex fields in one message, per
 * neighbour [cell planes][vertex planes]; both lists must name the same neighbours in the same order).
 * sendIdx / recvIdx are 1-based local indices, all neighbours
 * concatenated; sendCounts / recvCounts (nPeers entries each, may be NULL with nPeers = 0 for a single
 * segment) give the number of entries per neighbour in that order. */
int sih_halo_set_lists(sih_ctx *ctx, int location, int nSend, const int32_t *sendIdx, int nRecv,
                       const int32_t *recvIdx, int nPeers, const int32_t *sendCounts,
                       const int32_t *recvCounts);
/* gather the listed entries of `nFields` fields (all of one location and type) into a DEVICE buffer laid
 * out neighbour-major, so that each neighbour's message is contiguous:
 *   buf[off_p * nFields + f * cnt_p + i] = field_f(sendIdx[off_p + i]),  off_p = sum of counts before p;
 * unpack scatters the same layout (with the recv counts) into field_f(recvIdx[...]).
 * Both are asynchronous on the context's stream. */
int sih_halo_pack(sih_ctx *ctx, int location, int nFields, const char *const *names, void *deviceBuf);
int sih_halo_unpack(sih_ctx *ctx, int location, int nFields, const char *const *names, const void *deviceBuf);

/* ---- halo exchange inside the library -----------------------------------------------------------
 * The transport of mpas_dmpar_exch_group_full_halo_exch for blocks that live in different processes (one block
 * per GPU) or on different devices of one process: the per-subcycle update of uVelocity / vVelocity
 * (velocity_solver.F:2543-2584), the once-per-step updates of the pre- and post-cycle (:842-937, :1298-1317,
 * :1480-1616, :3764-3783) and the tracer halo of the advection (incremental_remap.F:2456-2484).  A block's
 * neighbours store their messages straight into its receive buffer (mapped through HIP IPC, over xGMI between
 * GPUs) and raise a sequence number there; the receiver waits for those numbers on the device.  Nothing
 * synchronises with the host, and both launches of an exchange can be captured in a HIP graph.
 *
 * Set-up: sih_halo_set_lists for cells and vertices with nPeers = nRanks (entry p = rank p, zero counts where
 * nothing is exchanged), then sih_comm_create (maxCellPlanes / maxVertexPlanes: the largest number of values
 * per cell / per vertex any one exchange will carry, e.g. 3 * maxEdges cell planes for the three stress
 * fields, nLayers * nCategories summed over the tracers for the tracer halo), then every rank's handle
 * (sih_comm_handle_size bytes, from sih_comm_get_handle) is gathered by whatever the host has -- MPI_Allgather in
 * an MPAS host, a file, a socket -- and the table, in rank order, is given to sih_comm_connect.  All ranks
 * must make the same sequence of exchanges.  Halos must be mutual, as those of MPAS blocks are: a block that sends
 * to a rank it receives nothing from (or the reverse) is refused by sih_comm_create with SIH_ERR_ARG. */
size_t sih_comm_handle_size(void);
int sih_comm_create(sih_ctx *ctx, int rank, int nRanks, int maxCellPlanes, int maxVertexPlanes);
int sih_comm_get_handle(sih_ctx *ctx, void *handle);
int sih_comm_connect(sih_ctx *ctx, const void *allHandles);
/* releases the tables of the transport.  The receive buffer itself (uncached device memory, a few megabytes) is not
 * given back to the runtime but kept for the next sih_comm_create on the same device: memory that ordinary allocations
 * took over from a freed uncached buffer was seen to show kernels the payload of earlier exchanges. */
int sih_comm_destroy(sih_ctx *ctx);
/* One halo update of the named fields (cell and vertex fields may be mixed; all float64 or all int32),
 * asynchronous on the context's stream.  _start sends only, _finish waits for the neighbours and fills the halo:
 * work queued between the two runs while the messages travel and must not touch halo entries of the fields. */
int sih_halo_exchange(sih_ctx *ctx, int nFields, const char *const *names);
int sih_halo_exchange_start(sih_ctx *ctx, int nFields, const char *const *names);
int sih_halo_exchange_finish(sih_ctx *ctx, int nFields, const char *const *names);
/* single_subcycle_velocity_solver with its halo update (velocity_solver.F:2478-2592) n times, everything on the
 * device: the iterations and, after every `interval` of them, the update of uVelocity / vVelocity through
 * sih_halo_exchange.  interval = 1 is the reference's order.  interval = k > 1 needs computeHaloVertices and k halo
 * layers in the block: stress11 / stress22 / stress12 then travel with the velocities and the results are the
 * same.  last != 0: the n-th iteration is the final one of the dynamics step (see sih_evp_subcycle) and is
 * followed by an update of its own.  overlap != 0 (after sih_evp_overlap_setup): the vertices the neighbours need
 * are updated first and sent from a second stream while the context's stream finishes the other vertices and
 * the next iteration's interior cells.  useGraph != 0: every distinct chunk (iterations + exchange) is captured
 * once and replayed as one graph launch.  Asynchronous; check sih_comm_status at the end of the step. */
int sih_evp_subcycle_exchange(sih_ctx *ctx, int n, int last, int interval, int overlap, int useGraph);
/* The device-side wait is bounded (default 10 s, environment SIH_HALO_TIMEOUT_MS): when a neighbour's message does
 * not arrive the halo is left as it was and the failure is recorded.  sih_comm_status waits for the stream and
 * returns SIH_ERR_COMM if that has happened since sih_comm_create (the reference's counterpart is an MPI error
 * in the framework's exchange, fatal there too). */
int sih_comm_set_timeout(sih_ctx *ctx, double milliseconds);
int sih_comm_status(sih_ctx *ctx);

/* ---- incremental-remap advection ---------------------------------------------------------------
 * Replaces, for one block, seaice_init_advection_incremental_remap and the per-block part of
 * seaice_run_advection_incremental_remap (src/shared/mpas_seaice_advection_incremental_remap.F:165,
 * :2338) with incremental_remap_block (:2740).  Hexagonal meshes (vertexDegree 3) and quadrilateral meshes
 * (vertexDegree 4: six side edges and cells per edge, :1133-1172, and the split of side triangles, :5807-5958).  The
 * velocity solver takes a quadrilateral block in its variational scheme on one block; the upwind advection refuses it. */

/* mesh-pool fields the advection reads beyond sih_mesh_desc, and the options it reads
 * (incremental_remap.F:257-340, :2899-2901).  Pointers are borrowed for the duration of the call. */
typedef struct sih_ir_desc {
    const int32_t *edgesOnCell;    /* (maxEdges, nCells) */
    const int32_t *cellsOnEdge;    /* (2, nEdges) */
    const int32_t *verticesOnEdge; /* (2, nEdges) */
    const int32_t *edgesOnVertex;  /* (vertexDegree, nVertices) */
    const double *xEdge, *yEdge, *zEdge; /* (nEdges) */
    const double *dcEdge, *dvEdge;       /* (nEdges) */
    const double *coeffsReconstruct;     /* (3, maxEdges, nCells): mesh-pool field coeffs_reconstruct, the
                                            output of the framework's mpas_init_reconstruct (:744-746) */
    int32_t nCategories;
    int32_t rotateCartesianGrid;         /* config_rotate_cartesian_grid */
} sih_ir_desc;

/* seaice_init_advection_incremental_remap (:165-816): local frames, vertex coordinates relative to
 * cells and edges, remap connectivity, geometric cell averages.  Fails with SIH_ERR_ARG where the
 * reference aborts (cells not convex / not counter-clockwise, cell 1 of an edge not in its left
 * half-plane, non-positive cell area). */
int sih_ir_init(sih_ctx *ctx, const sih_ir_desc *desc);
/* The ordered tracer list (seaice_add_tracers_to_linked_list / set_tracer_order,
 * incremental_remap_tracers.F:158-526, :860-922): names[0] is the mass-like field (parents[0] = "None"),
 * every parent precedes its children, nLayers[i] is the size of the leading dimension of tracer i
 * (1 for the (ONE, nCategories, nCells) fields).  Each tracer becomes a field of that name with
 * nLayers*nCategories values per cell for sih_field_put / sih_field_get. */
int sih_ir_set_tracers(sih_ctx *ctx, int nTracers, const char *const *names, const char *const *parents,
                       const int32_t *nLayers);
/* One call of the block part of seaice_run_advection_incremental_remap (:2489-2650): volume ->
 * thickness, incremental_remap_block with time step dt (uVelocity / vVelocity and the tracer fields are
 * read on the device; halos must be current), thickness -> volume.  Returns SIH_ERR_IR where the
 * reference aborts ("IR negative reconstructed ice area", "IR: Negative mass in IR", parallel basis
 * vectors).  `checks` is a bit set: 1 (config_conservation_check) also evaluates sum_tracers (:7998)
 * before and after the update; 2 (config_monotonicity_check) records the local minima and maxima of the
 * old values (tracer_local_min_max :8268) in the fields <tracer>LocalMin / <tracer>LocalMax and leaves
 * the thickness -> volume conversion to sih_ir_check_monotonicity, which must be called next; 4
 * (config_recover_tracer_means_check) integrates every reconstruction over its cell (recover_tracer_means :7541),
 * leaves the result in the fields <tracer>RecoveredProduct and counts the entries that differ from the product
 * of the means by more than 1e-11 relative and absolute, where the reference logs a warning
 * (sih_get_info "irRecoverMeansWarnings"). */
enum { SIH_IR_CHECK_CONSERVATION = 1, SIH_IR_CHECK_MONOTONICITY = 2, SIH_IR_CHECK_RECOVER_MEANS = 4 };
int sih_ir_run(sih_ctx *ctx, double dt, int checks);
/* check_tracer_monotonicity (:8416-8749) after sih_ir_run(.., checks & 2): extends the local minima and
 * maxima by one layer of neighbours exactly as the reference's in-place loop does, compares every new
 * tracer value that had a meaningful old value against them (tolerance 1e-11 * max(1, |bound|)), then
 * converts thickness back to volume.  A multi-block host exchanges the halos of the LocalMin / LocalMax
 * fields of every tracer with a parent between the two calls (:8495-8496).  Returns SIH_ERR_IR with
 * "IR advection, monotonicity violation" where the reference aborts. */
int sih_ir_check_monotonicity(sih_ctx *ctx);
/* inspection: copies every tracer's reconstruction of the last sih_ir_run (value at the cell's origin and
 * limited gradient, linear_tracer_field :3785-4202), which the library keeps as per-cell records, into the
 * plane fields <tracer>Center, <tracer>Gradx, <tracer>Grady for sih_field_get */
int sih_ir_unpack_reconstruction(sih_ctx *ctx);
/* sum_tracers results of the last sih_ir_run(.., checks & 1): nLayers*nCategories values each */
int sih_ir_get_sums(sih_ctx *ctx, const char *tracer, double *initSums, double *finalSums);
/* sih_ir_run bracketed by HIP events on the context's stream: total and per-phase durations in ms
 * (phases: 0 conversion+masks, 1 reconstruction, 2 departure triangles, 3 fluxes, 4 update+zap) */
int sih_ir_run_timed(sih_ctx *ctx, double dt, double *totalMs, double *phaseMs5);

/* ---- upwind advection ----------------------------------------------------------------------------
 * Replaces, for one block, seaice_init_advection_upwind and the per-block parts of
 * seaice_run_advection_upwind (src/shared/mpas_seaice_advection_upwind.F:79, :385), the alternative the
 * reference selects with config_advection_type = 'upwind' (mpas_seaice_advection.F:64, :115).  The tracers are
 * the reference's chain (define_tracer_connectivities :145-189): iceAreaCategory, surfaceTemperature,
 * iceVolumeCategory, snowVolumeCategory, each a field of nCategories values per cell.  Beside them the
 * library keeps, per tracer, the second time level "<tracer>Level2", the tendency "<tracer>Tend" (cells) and
 * the edge flux "<tracer>EdgeFlux" (nCategories values per edge), as the reference leaves them in its
 * tracers / tracer_tendencies / tracer_edge_fluxes pools, plus normalVectorEdge (2, maxEdges, nCells),
 * edgeVelocity (nEdges) and interiorEdge (nEdges, int). */

/* seaice_init_advection_upwind (:79-131): the edge normals seaice_normal_vectors_polygon(.., rotate, .false.)
 * (mesh.F:760) and interior_edges (mesh.F:567) for every edge of the block.  coeffsReconstruct may be NULL. */
int sih_upwind_init(sih_ctx *ctx, const sih_ir_desc *desc);
/* prepare_advection (:1549-1619): zeroes the second time level, tendencies and edge fluxes, computes the edge
 * velocities from uVelocity / vVelocity, converts the volumes to thickness on every cell of the block;
 * checks & 1 also evaluates initial_conservation (:2454).  A multi-block host exchanges the halos of the four
 * tracers between this call and sih_upwind_advect (halo_exchange_advection :1786). */
int sih_upwind_prepare(sih_ctx *ctx, int checks);
/* the tracer loop (:479-504) and finalize_advection (:1840): per tracer the upwind fluxes, tendencies and new
 * values with time step dt, then the time levels trade places ("<tracer>" holds the new values,
 * "<tracer>Level2" the old ones multiplied by their parents), the products are divided back (owned cells) and
 * thickness becomes volume; checks & 1 also evaluates final_conservation (:2571) */
int sih_upwind_advect(sih_ctx *ctx, double dt, int checks);
/* sih_upwind_prepare followed by sih_upwind_advect, for a block that needs no halo exchange */
int sih_upwind_run(sih_ctx *ctx, double dt, int checks);
/* the aggregates of the last run with checks & 1 (tracer_conservation pool, time levels 1 and 2):
 * nCategories values each */
int sih_upwind_get_sums(sih_ctx *ctx, const char *tracer, double *initSums, double *finalSums);

/* introspection for tests and reports.  Keys: sparseGradient, symmetricMetric, derivedGradient, derivedIntegralRow (1 when
 * the set-up found the layout reduction of that name to apply, see sih_evp_init), splitCellKernel, persistentKernel (and persistentPerCU,
 * persistentTiles, persistentReason, persistentArrived, persistentLaunches, persistentExchangeLaunches,
 * persistentExchangeReady (see sih_evp_set_persistent), fusedKernel and fusedLaunches (sih_evp_subcycle_fused), persistentTileCells, iterationForm (what the last loop of iterations ran as: 0 two
 * kernels an iteration, 1 the persistent kernel, 2 the fused kernel): workgroups of
 * the persistent subcycle kernel a CU holds, its tiles, why the block does not qualify (1 option, 2 too many tiles,
 * 3 too many vertices in a tile, 5 not resident together in the trial launch), workgroups that arrived in a failed
 * trial, launches so far and those with the halo exchange inside), dampingTimescale,
 * elasticTimeStep, dynamicsTimeStep, numericalInertiaCoefficient, padC, padV, minDvEdge, and
 * irRecoverMeansWarnings (see sih_ir_run), irFluxWalk (1 when the flux kernel that walks the tracer tree is in use:
 * every tracer with children has one layer and there are at most 32 layer fields), irCflWarnings: number of vertices whose departure distance in the last sih_ir_run exceeded the
 * shortest adjacent edge (the reference logs "Potential CFL violation in IR advection" and continues,
 * incremental_remap.F:5330-5345). */
int sih_get_info(sih_ctx *ctx, const char *key, double *value);

#ifdef __cplusplus
}
#endif
#endif /* SEAICE_HIP_H */
