!> ISO_C_BINDING view of include/seaice_hip.h: the interface a Fortran host (the MPAS-Seaice shim) uses
!> to drive libseaice_hip.  Names, argument order and struct layouts mirror the C header one to one.
module seaice_hip_interface

  use iso_c_binding

  implicit none

  public

  integer(c_int), parameter :: SIH_OK = 0, SIH_ERR_ARG = 1, SIH_ERR_HIP = 2, SIH_ERR_STATE = 3, &
                               SIH_ERR_UNSUPPORTED = 4, SIH_ERR_IR = 5, SIH_ERR_COMM = 6

  !> struct sih_mesh_desc
  type, bind(C) :: sih_mesh_desc
     integer(c_int32_t) :: nCells, nCellsSolve
     integer(c_int32_t) :: nVertices, nVerticesSolve
     integer(c_int32_t) :: nEdges, nEdgesSolve
     integer(c_int32_t) :: maxEdges, vertexDegree
     integer(c_int32_t) :: onASphere
     real(c_double)     :: sphereRadius
     type(c_ptr) :: nEdgesOnCell, verticesOnCell, cellsOnCell, edgesOnCell
     type(c_ptr) :: cellsOnVertex, cellsOnEdge, verticesOnEdge
     type(c_ptr) :: xCell, yCell, zCell
     type(c_ptr) :: xVertex, yVertex, zVertex
     type(c_ptr) :: areaCell, areaTriangle, fVertex
     type(c_ptr) :: dvEdge, dcEdge
     integer(c_int32_t) :: renumber
  end type sih_mesh_desc

  !> struct sih_evp_config
  type, bind(C) :: sih_evp_config
     real(c_double)     :: dt
     integer(c_int32_t) :: dynamicsSubcycleNumber, elasticSubcycleNumber
     integer(c_int32_t) :: constitutiveRelation, variationalBasis, variationalDenominator
     integer(c_int32_t) :: wachspressIntegrationOrder
     integer(c_int32_t) :: rotateCartesianGrid, includeMetricTerms
     integer(c_int32_t) :: useAirStress, useOceanStress, oceanStressType, surfaceTilt
     integer(c_int32_t) :: calcVelocityMasks, useColumnPackage, nCategories
     real(c_double)     :: minEdgeLengthGlobal
     integer(c_int32_t) :: computeHaloVertices
     integer(c_int32_t) :: averageVariationalStrain = 0
     integer(c_int32_t) :: useSpecialBoundariesVelocity = 0, useSpecialBoundariesVelocityMasks = 0
     integer(c_int32_t) :: strainScheme = 0, stressDivergenceScheme = 0   ! 0 variational, 1 weak
     ! ice strength of the column package (colpkg_ice_strength): used when useColumnPackage and
     ! useColumnVerticalThermodynamics are both set; kstrength, krdg_partic, krdg_redist, mu_rdg, Cf
     integer(c_int32_t) :: useColumnVerticalThermodynamics = 0
     integer(c_int32_t) :: iceStrengthFormulation = 1, ridgingParticipation = 1, ridgingRedistribution = 1
     real(c_double)     :: ridgingEfoldingScale = 3.0_c_double, ridgingWorkRatio = 17.0_c_double
     integer(c_int32_t) :: wachspressIntegrationType = 1   ! 1 dunavant, 2 trapezoidal (orders 1..21), 3 fekete
  end type sih_evp_config

  !> struct sih_ir_desc
  type, bind(C) :: sih_ir_desc
     type(c_ptr) :: edgesOnCell, cellsOnEdge, verticesOnEdge, edgesOnVertex
     type(c_ptr) :: xEdge, yEdge, zEdge
     type(c_ptr) :: dcEdge, dvEdge, coeffsReconstruct
     integer(c_int32_t) :: nCategories, rotateCartesianGrid
  end type sih_ir_desc

  ! sih_ridge_config of seaice_hip.h: the arguments of colpkg_step_ridge that are not fields
  type, bind(C) :: sih_ridge_config
     real(c_double)     :: dt
     integer(c_int32_t) :: dynamicsSubcycleNumber
     integer(c_int32_t) :: participation, redistribution   ! 0 Thorndike75 / Hibler80, 1 exponential
     real(c_double)     :: efoldingScale
     integer(c_int32_t) :: nCategories
     type(c_ptr)        :: categoryThicknessLimits        ! nCategories + 1 doubles
     integer(c_int32_t) :: itdCleanup = 1                 ! 1: the ITD clean-up of colpkg_step_ridge after ridge_ice
  end type sih_ridge_config

  interface

     function sih_column_ridge_init(ctx, nCategories) bind(C, name="sih_column_ridge_init") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nCategories
       integer(c_int) :: ierr
     end function sih_column_ridge_init

     function sih_column_ridge(ctx, cfg) bind(C, name="sih_column_ridge") result(ierr)
       import :: c_int, c_ptr, sih_ridge_config
       type(c_ptr), value :: ctx
       type(sih_ridge_config), intent(in) :: cfg
       integer(c_int) :: ierr
     end function sih_column_ridge

     function sih_column_update_state(ctx, nCategories, dt, tendencies) bind(C, name="sih_column_update_state") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nCategories
       real(c_double), value :: dt
       integer(c_int), value :: tendencies
       integer(c_int) :: ierr
     end function sih_column_update_state

     function sih_column_reinitialize_diagnostics(ctx, nCategories) &
          bind(C, name="sih_column_reinitialize_diagnostics") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nCategories
       integer(c_int) :: ierr
     end function sih_column_reinitialize_diagnostics

     function sih_block_create(mesh, device, ctx) bind(C, name="sih_block_create") result(ierr)
       import :: sih_mesh_desc, c_int, c_ptr
       type(sih_mesh_desc), intent(in) :: mesh
       integer(c_int), value :: device
       type(c_ptr), intent(out) :: ctx
       integer(c_int) :: ierr
     end function sih_block_create

     function sih_block_destroy(ctx) bind(C, name="sih_block_destroy") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_block_destroy

     function sih_last_error() bind(C, name="sih_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function sih_last_error

     function sih_set_stream(ctx, hipStream) bind(C, name="sih_set_stream") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, hipStream
       integer(c_int) :: ierr
     end function sih_set_stream

     function sih_sync(ctx) bind(C, name="sih_sync") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_sync

     function sih_evp_init(ctx, cfg) bind(C, name="sih_evp_init") result(ierr)
       import :: sih_evp_config, c_int, c_ptr
       type(c_ptr), value :: ctx
       type(sih_evp_config), intent(in) :: cfg
       integer(c_int) :: ierr
     end function sih_evp_init

     function sih_evp_static_changed(ctx) bind(C, name="sih_evp_static_changed") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_static_changed

     ! edge description of the mesh for the weak scheme (before sih_evp_init)
     function sih_evp_set_edge_mesh(ctx, edges) bind(C, name="sih_evp_set_edge_mesh") result(ierr)
       import :: sih_ir_desc, c_int, c_ptr
       type(c_ptr), value :: ctx
       type(sih_ir_desc), intent(in) :: edges
       integer(c_int) :: ierr
     end function sih_evp_set_edge_mesh

     ! arrays of the special_boundaries pool (c_loc; c_null_ptr for a part that is switched off)
     function sih_evp_set_special_boundaries(ctx, vertexBoundaryType, vertexBoundarySourceLocal, &
          solveVelocitySpecialBoundaries, solveStressSpecialBoundaries) &
          bind(C, name="sih_evp_set_special_boundaries") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: vertexBoundaryType, vertexBoundarySourceLocal
       type(c_ptr), value :: solveVelocitySpecialBoundaries, solveStressSpecialBoundaries
       integer(c_int) :: ierr
     end function sih_evp_set_special_boundaries

     ! several blocks: velocity boundaries after the host's halo update instead of inside the iteration
     function sih_evp_defer_special_boundaries(ctx, on) bind(C, name="sih_evp_defer_special_boundaries") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
       integer(c_int) :: ierr
     end function sih_evp_defer_special_boundaries

     function sih_evp_apply_special_boundaries(ctx) bind(C, name="sih_evp_apply_special_boundaries") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_apply_special_boundaries

     function sih_evp_overlap_setup(ctx) bind(C, name="sih_evp_overlap_setup") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_overlap_setup

     function sih_evp_subcycle_parts(ctx, nParts, parts, useGraph) bind(C, name="sih_evp_subcycle_parts") result(ierr)
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nParts
       integer(c_int32_t), dimension(*), intent(in) :: parts
       integer(c_int), value :: useGraph
       integer(c_int) :: ierr
     end function sih_evp_subcycle_parts

     function sih_field_put(ctx, name, host, count) bind(C, name="sih_field_put") result(ierr)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: name
       type(c_ptr), value :: host
       integer(c_size_t), value :: count
       integer(c_int) :: ierr
     end function sih_field_put

     function sih_field_get(ctx, name, host, count) bind(C, name="sih_field_get") result(ierr)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: name
       type(c_ptr), value :: host
       integer(c_size_t), value :: count
       integer(c_int) :: ierr
     end function sih_field_get

     ! names: array of C pointers to NUL-terminated strings (c_loc of character(kind=c_char) targets);
     ! hosts: array of c_loc(array); counts: logical element counts
     function sih_fields_put(ctx, nFields, names, hosts, counts) bind(C, name="sih_fields_put") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), dimension(*), intent(in) :: hosts
       integer(c_size_t), dimension(*), intent(in) :: counts
       integer(c_int) :: ierr
     end function sih_fields_put

     function sih_fields_get(ctx, nFields, names, hosts, counts) bind(C, name="sih_fields_get") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), dimension(*), intent(in) :: hosts
       integer(c_size_t), dimension(*), intent(in) :: counts
       integer(c_int) :: ierr
     end function sih_fields_get

     ! the same download, draining behind whatever the host starts next; the arrays are the library's until
     ! sih_fields_get_finish has returned
     function sih_fields_get_start(ctx, nFields, names, hosts, counts) bind(C, name="sih_fields_get_start") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), dimension(*), intent(in) :: hosts
       integer(c_size_t), dimension(*), intent(in) :: counts
       integer(c_int) :: ierr
     end function sih_fields_get_start

     ! sih_evp_run with the download of the listed fields (final after the pre-cycle) behind the iterations
     function sih_evp_run_early_outputs(ctx, nFields, names, hosts, counts) &
          bind(C, name="sih_evp_run_early_outputs") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), dimension(*), intent(in) :: hosts
       integer(c_size_t), dimension(*), intent(in) :: counts
       integer(c_int) :: ierr
     end function sih_evp_run_early_outputs

     function sih_fields_get_finish(ctx) bind(C, name="sih_fields_get_finish") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_fields_get_finish

     function sih_field_count(ctx, name, elemSize) bind(C, name="sih_field_count") result(n)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: name
       integer(c_int), intent(out) :: elemSize
       integer(c_size_t) :: n
     end function sih_field_count

     function sih_evp_pre_subcycle(ctx, stage) bind(C, name="sih_evp_pre_subcycle") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: stage
       integer(c_int) :: ierr
     end function sih_evp_pre_subcycle

     function sih_evp_subcycle(ctx, n, last) bind(C, name="sih_evp_subcycle") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n, last
       integer(c_int) :: ierr
     end function sih_evp_subcycle

     function sih_evp_post_subcycle(ctx, stage) bind(C, name="sih_evp_post_subcycle") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: stage
       integer(c_int) :: ierr
     end function sih_evp_post_subcycle

     function sih_evp_run(ctx) bind(C, name="sih_evp_run") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_run

     function sih_evp_subcycle_timed(ctx, n, useGraph, sampleEvery, totalMs, cellMs, vertexMs) &
          bind(C, name="sih_evp_subcycle_timed") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n, useGraph, sampleEvery
       real(c_double), intent(out) :: totalMs, cellMs, vertexMs
       integer(c_int) :: ierr
     end function sih_evp_subcycle_timed

     function sih_halo_set_lists(ctx, location, nSend, sendIdx, nRecv, recvIdx, nPeers, sendCounts, recvCounts) &
          bind(C, name="sih_halo_set_lists") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: location, nSend, nRecv, nPeers
       type(c_ptr), value :: sendIdx, recvIdx, sendCounts, recvCounts
       integer(c_int) :: ierr
     end function sih_halo_set_lists

     function sih_halo_pack(ctx, location, nFields, names, deviceBuf) bind(C, name="sih_halo_pack") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: location, nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), value :: deviceBuf
       integer(c_int) :: ierr
     end function sih_halo_pack

     function sih_halo_unpack(ctx, location, nFields, names, deviceBuf) bind(C, name="sih_halo_unpack") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: location, nFields
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), value :: deviceBuf
       integer(c_int) :: ierr
     end function sih_halo_unpack

     ! hand-off to the column package's ridging: host(S, nCategories, nCells) records of the listed fields for
     ! the host cells firstCell .. firstCell+nCells-1 (0-based), gathered / scattered on the device
     function sih_column_get(ctx, nFields, names, nCategories, firstCell, nCells, host, count) &
          bind(C, name="sih_column_get") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields, nCategories, firstCell, nCells
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), value :: host
       integer(c_size_t), value :: count
       integer(c_int) :: ierr
     end function sih_column_get

     function sih_column_put(ctx, nFields, names, nCategories, firstCell, nCells, host, count) &
          bind(C, name="sih_column_put") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields, nCategories, firstCell, nCells
       type(c_ptr), dimension(*), intent(in) :: names
       type(c_ptr), value :: host
       integer(c_size_t), value :: count
       integer(c_int) :: ierr
     end function sih_column_put

     function sih_evp_subcycle_graph(ctx, n) bind(C, name="sih_evp_subcycle_graph") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int) :: ierr
     end function sih_evp_subcycle_graph

     ! n iterations in one launch of the persistent kernel (small blocks; SIH_ERR_UNSUPPORTED where a block does
     ! not qualify: query sih_get_info "persistentKernel")
     function sih_evp_subcycle_persistent(ctx, n) bind(C, name="sih_evp_subcycle_persistent") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int) :: ierr
     end function sih_evp_subcycle_persistent

     ! n iterations as n + 1 launches of the fused kernel (vertices of one iteration, cells of the next; blocks too
     ! large for the persistent kernel; SIH_ERR_UNSUPPORTED where a block does not qualify: "fusedKernel")
     function sih_evp_subcycle_fused(ctx, n) bind(C, name="sih_evp_subcycle_fused") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int) :: ierr
     end function sih_evp_subcycle_fused

     ! 0 two-kernel form, 1 persistent kernel, 2 persistent kernel with the halo exchange inside (default where a
     ! block qualifies), -1 back to the environment; the same on every block of a run
     function sih_evp_set_persistent(ctx, mode) bind(C, name="sih_evp_set_persistent") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: mode
       integer(c_int) :: ierr
     end function sih_evp_set_persistent

     function sih_ir_init(ctx, desc) bind(C, name="sih_ir_init") result(ierr)
       import :: sih_ir_desc, c_int, c_ptr
       type(c_ptr), value :: ctx
       type(sih_ir_desc), intent(in) :: desc
       integer(c_int) :: ierr
     end function sih_ir_init

     !> names / parents: arrays of C pointers to NUL-terminated strings; nLayers: int32 array
     function sih_ir_set_tracers(ctx, nTracers, names, parents, nLayers) bind(C, name="sih_ir_set_tracers") &
          result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nTracers
       type(c_ptr), dimension(*), intent(in) :: names, parents
       type(c_ptr), value :: nLayers
       integer(c_int) :: ierr
     end function sih_ir_set_tracers

     ! checks: 1 = config_conservation_check, 2 = config_monotonicity_check (then sih_ir_check_monotonicity
     ! must follow), 4 = config_recover_tracer_means_check (warnings counted in sih_get_info
     ! "irRecoverMeansWarnings"), or their sum
     function sih_ir_run(ctx, dt, checks) bind(C, name="sih_ir_run") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), value :: checks
       integer(c_int) :: ierr
     end function sih_ir_run

     function sih_ir_check_monotonicity(ctx) bind(C, name="sih_ir_check_monotonicity") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_ir_check_monotonicity

     !> page-locked host memory for arrays exchanged every step (use with c_f_pointer)
     function sih_host_alloc(ptr, bytes) bind(C, name="sih_host_alloc") result(ierr)
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), intent(out) :: ptr
       integer(c_size_t), value :: bytes
       integer(c_int) :: ierr
     end function sih_host_alloc

     function sih_host_free(ptr) bind(C, name="sih_host_free") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ptr
       integer(c_int) :: ierr
     end function sih_host_free

     function sih_ir_unpack_reconstruction(ctx) bind(C, name="sih_ir_unpack_reconstruction") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_ir_unpack_reconstruction

     function sih_ir_get_sums(ctx, tracer, initSums, finalSums) bind(C, name="sih_ir_get_sums") result(ierr)
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: tracer
       type(c_ptr), value :: initSums, finalSums
       integer(c_int) :: ierr
     end function sih_ir_get_sums

     ! ---- upwind advection (config_advection_type = 'upwind') ----
     function sih_upwind_init(ctx, desc) bind(C, name="sih_upwind_init") result(ierr)
       import :: sih_ir_desc, c_int, c_ptr
       type(c_ptr), value :: ctx
       type(sih_ir_desc), intent(in) :: desc
       integer(c_int) :: ierr
     end function sih_upwind_init

     ! checks: 1 = config_conservation_check
     function sih_upwind_prepare(ctx, checks) bind(C, name="sih_upwind_prepare") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: checks
       integer(c_int) :: ierr
     end function sih_upwind_prepare

     function sih_upwind_advect(ctx, dt, checks) bind(C, name="sih_upwind_advect") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), value :: checks
       integer(c_int) :: ierr
     end function sih_upwind_advect

     function sih_upwind_run(ctx, dt, checks) bind(C, name="sih_upwind_run") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), value :: checks
       integer(c_int) :: ierr
     end function sih_upwind_run

     function sih_upwind_get_sums(ctx, tracer, initSums, finalSums) bind(C, name="sih_upwind_get_sums") &
          result(ierr)
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: tracer
       type(c_ptr), value :: initSums, finalSums
       integer(c_int) :: ierr
     end function sih_upwind_get_sums

     function sih_ir_run_timed(ctx, dt, totalMs, phaseMs5) bind(C, name="sih_ir_run_timed") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       real(c_double), intent(out) :: totalMs
       type(c_ptr), value :: phaseMs5
       integer(c_int) :: ierr
     end function sih_ir_run_timed

     function sih_device_count() bind(C, name="sih_device_count") result(n)
       import :: c_int
       integer(c_int) :: n
     end function sih_device_count

     function sih_field_get_noalloc(ctx, name, host, count) bind(C, name="sih_field_get_noalloc") result(ierr)
       import :: c_int, c_ptr, c_char, c_size_t
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: name
       type(c_ptr), value :: host
       integer(c_size_t), value :: count
       integer(c_int) :: ierr
     end function sih_field_get_noalloc

     ! the reference's two public operators on their own (velocity_solver_variational.F:575, :1064)
     function sih_evp_strain_tensor(ctx) bind(C, name="sih_evp_strain_tensor") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_strain_tensor

     function sih_evp_stress_divergence(ctx) bind(C, name="sih_evp_stress_divergence") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_evp_stress_divergence

     ! ---- halo exchange inside the library (the transport of mpas_dmpar_exch_group_full_halo_exch) ----
     function sih_comm_handle_size() bind(C, name="sih_comm_handle_size") result(n)
       import :: c_size_t
       integer(c_size_t) :: n
     end function sih_comm_handle_size

     function sih_comm_create(ctx, rank, nRanks, maxCellPlanes, maxVertexPlanes) bind(C, name="sih_comm_create") &
          result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: rank, nRanks, maxCellPlanes, maxVertexPlanes
       integer(c_int) :: ierr
     end function sih_comm_create

     function sih_comm_get_handle(ctx, handle) bind(C, name="sih_comm_get_handle") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, handle
       integer(c_int) :: ierr
     end function sih_comm_get_handle

     function sih_comm_connect(ctx, allHandles) bind(C, name="sih_comm_connect") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, allHandles
       integer(c_int) :: ierr
     end function sih_comm_connect

     function sih_comm_destroy(ctx) bind(C, name="sih_comm_destroy") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_comm_destroy

     function sih_halo_exchange(ctx, nFields, names) bind(C, name="sih_halo_exchange") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       integer(c_int) :: ierr
     end function sih_halo_exchange

     function sih_halo_exchange_start(ctx, nFields, names) bind(C, name="sih_halo_exchange_start") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       integer(c_int) :: ierr
     end function sih_halo_exchange_start

     function sih_halo_exchange_finish(ctx, nFields, names) bind(C, name="sih_halo_exchange_finish") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nFields
       type(c_ptr), dimension(*), intent(in) :: names
       integer(c_int) :: ierr
     end function sih_halo_exchange_finish

     function sih_comm_set_timeout(ctx, milliseconds) bind(C, name="sih_comm_set_timeout") result(ierr)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: milliseconds
       integer(c_int) :: ierr
     end function sih_comm_set_timeout

     function sih_comm_status(ctx) bind(C, name="sih_comm_status") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: ierr
     end function sih_comm_status

     !> n iterations of single_subcycle_velocity_solver with the halo update of uVelocity / vVelocity after every
     !> `interval` of them (velocity_solver.F:2478-2592), all on the device
     function sih_evp_subcycle_exchange(ctx, n, last, interval, overlap, useGraph) &
          bind(C, name="sih_evp_subcycle_exchange") result(ierr)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: n, last, interval, overlap, useGraph
       integer(c_int) :: ierr
     end function sih_evp_subcycle_exchange

     function sih_get_info(ctx, key, value) bind(C, name="sih_get_info") result(ierr)
       import :: c_int, c_ptr, c_char, c_double
       type(c_ptr), value :: ctx
       character(kind=c_char), dimension(*), intent(in) :: key
       real(c_double), intent(out) :: value
       integer(c_int) :: ierr
     end function sih_get_info

  end interface

contains

  !> sih_last_error() as a Fortran string
  function sih_error_message() result(msg)
    character(len=:), allocatable :: msg
    type(c_ptr) :: p
    character(kind=c_char), dimension(:), pointer :: s
    integer :: n, i
    p = sih_last_error()
    msg = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, [1024])
    n = 0
    do while (n < 1024)
       if (s(n+1) == c_null_char) exit
       n = n + 1
    enddo
    allocate(character(len=n) :: msg)
    do i = 1, n
       msg(i:i) = s(i)
    enddo
  end function sih_error_message

end module seaice_hip_interface
