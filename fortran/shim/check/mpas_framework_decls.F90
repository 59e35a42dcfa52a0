! Declarations of the parts of the MPAS framework, and of the one sea-ice module that stays in the build, which
! fortran/shim/*.F use: types with the components the shims touch and procedures with empty bodies.  Their only
! purpose is `make check-shims` (amdflang -fsyntax-only), which type-checks the shim modules against these
! interfaces, and `make check-abort-path`, which links the advection shim with them to walk its error path on the
! CPU: for that the handful of procedures the path goes through answer with fixed values and the log and abort
! routines print what they are asked to do.  Nothing here is a substitute for the framework.

module mpass_derived_types
  implicit none
  integer, parameter :: RKIND = selected_real_kind(12)
  integer, parameter :: StrKIND = 512
  integer, parameter :: MPAS_LOG_OUT = 1, MPAS_LOG_WARN = 2, MPAS_LOG_ERR = 3, MPAS_LOG_CRIT = 4
  type MPAS_pool_type
     integer :: size = 0
  end type MPAS_pool_type
  type MPAS_pool_field_info_type
     integer :: fieldType, nDims, nTimeLevels, nHaloLayers
     logical :: isActive
  end type MPAS_pool_field_info_type
  type MPAS_Clock_type
     integer :: direction = 0
  end type MPAS_Clock_type
  ! exchange lists of a field (mpas_derived_types of the framework): per halo layer a linked list over the
  ! neighbouring processes; endPointID is the neighbour's rank, srcList / destList pair local entries with
  ! positions in the message buffer
  type mpas_exchange_list
     integer :: endPointID, nList
     integer, dimension(:), pointer :: srcList => null(), destList => null()
     type(mpas_exchange_list), pointer :: next => null()
  end type mpas_exchange_list
  type mpas_exchange_list_pointer
     type(mpas_exchange_list), pointer :: exchList => null()
  end type mpas_exchange_list_pointer
  type mpas_multihalo_exchange_list
     type(mpas_exchange_list_pointer), dimension(:), pointer :: halos => null()
  end type mpas_multihalo_exchange_list
  type field1DReal
     real(kind=RKIND), dimension(:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field1DReal
  type field3DReal
     real(kind=RKIND), dimension(:,:,:), pointer :: array => null()
  end type field3DReal
  type MPAS_streamManager_type
     integer :: numStreams = 0
  end type MPAS_streamManager_type
  type block_type
     type(MPAS_pool_type), pointer :: structs => null(), dimensions => null(), configs => null()
     type(block_type), pointer :: prev => null(), next => null()
  end type block_type
  type dm_info
     integer :: nprocs = 1, my_proc_id = 0, comm = 0
  end type dm_info
  type domain_type
     type(block_type), pointer :: blocklist => null()
     type(MPAS_pool_type), pointer :: configs => null()
     type(dm_info), pointer :: dminfo => null()
     type(MPAS_streamManager_type), pointer :: streamManager => null()
  end type domain_type
end module mpass_derived_types

module mpass_pool_routines
  use mpass_derived_types
  implicit none
  interface MPAS_pool_get_array
     module procedure get_r0, get_r1, get_r2, get_r3, get_i1, get_i2
  end interface MPAS_pool_get_array
  interface MPAS_pool_get_config
     module procedure cfg_r, cfg_i, cfg_l, cfg_c
  end interface MPAS_pool_get_config
  interface MPAS_pool_get_dimension
     module procedure dim_0
  end interface MPAS_pool_get_dimension
  interface MPAS_pool_get_field
     module procedure fld_r3, fld_r1
  end interface MPAS_pool_get_field
  ! fixed answers for check-abort-path
  logical, target, save :: yes = .true., no = .false.
  real(kind=RKIND), target, save :: someReal = 900.0_RKIND
  real(kind=RKIND), dimension(4), target, save :: someVector = 0.0_RKIND
  character(len=StrKIND), target, save :: someString
  type(MPAS_pool_type), target, save :: somePool
contains
  subroutine get_r0(inPool, key, scalar, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), pointer :: scalar
    integer, intent(in), optional :: timeLevel
    scalar => someReal   ! e.g. dynamicsTimeStep
  end subroutine get_r0
  subroutine get_r1(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:), pointer :: array
    integer, intent(in), optional :: timeLevel
    array => someVector
  end subroutine get_r1
  subroutine get_r2(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
  end subroutine get_r2
  subroutine get_r3(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:,:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
  end subroutine get_r3
  subroutine get_i1(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, dimension(:), pointer :: array
    integer, intent(in), optional :: timeLevel
  end subroutine get_i1
  subroutine get_i2(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, dimension(:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
  end subroutine get_i2
  subroutine cfg_r(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), pointer :: value
  end subroutine cfg_r
  subroutine cfg_i(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, pointer :: value
  end subroutine cfg_i
  subroutine cfg_l(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    logical, pointer :: value
    if (key == "config_use_advection" .or. key == "config_full_abort_write") then
       value => yes
    else
       value => no
    endif
  end subroutine cfg_l
  subroutine cfg_c(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    character(len=StrKIND), pointer :: value
    someString = "incremental_remap"   ! config_advection_type
    value => someString
  end subroutine cfg_c
  subroutine dim_0(inPool, key, dim)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, pointer :: dim
  end subroutine dim_0
  subroutine fld_r3(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field3DReal), pointer :: field
    integer, intent(in), optional :: timeLevel
  end subroutine fld_r3
  subroutine fld_r1(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field1DReal), pointer :: field
    integer, intent(in), optional :: timeLevel
  end subroutine fld_r1
  subroutine MPAS_pool_get_subpool(inPool, key, subPool)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_type), pointer :: subPool
    subPool => somePool
  end subroutine MPAS_pool_get_subpool
  subroutine MPAS_pool_get_field_info(inPool, key, info)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_field_info_type), intent(out) :: info
    info % fieldType = 0; info % nDims = 1; info % nTimeLevels = 1; info % nHaloLayers = 0; info % isActive = .true.
  end subroutine MPAS_pool_get_field_info
  subroutine MPAS_pool_shift_time_levels(inPool)
    type(MPAS_pool_type), intent(inout) :: inPool
  end subroutine MPAS_pool_shift_time_levels
end module mpass_pool_routines

module mpass_timer
  implicit none
contains
  subroutine mpas_timer_start(timer_name)
    character(len=*), intent(in) :: timer_name
  end subroutine mpas_timer_start
  subroutine mpas_timer_stop(timer_name)
    character(len=*), intent(in) :: timer_name
  end subroutine mpas_timer_stop
end module mpass_timer

module mpass_timekeeping
  use mpass_derived_types
  implicit none
end module mpass_timekeeping

module mpass_log
  use mpass_derived_types
  implicit none
contains
  subroutine mpas_log_write(message, messageType, masterOnly, flushNow, intArgs, realArgs, logicArgs, err)
    character(len=*), intent(in) :: message
    integer, intent(in), optional :: messageType
    logical, intent(in), optional :: masterOnly, flushNow
    integer, dimension(:), intent(in), optional :: intArgs
    real(kind=RKIND), dimension(:), intent(in), optional :: realArgs
    logical, dimension(:), intent(in), optional :: logicArgs
    integer, intent(out), optional :: err
    integer :: kind
    kind = MPAS_LOG_OUT
    if (present(messageType)) kind = messageType
    write(*,'(a,i0,2a)') "mpas_log_write type ", kind, ": ", trim(message)
    if (kind == MPAS_LOG_CRIT) stop 3   ! the framework aborts the run here
  end subroutine mpas_log_write
end module mpass_log

module mpass_dmpar
  use mpass_derived_types
  implicit none
  interface MPAS_dmpar_exch_halo_field
     module procedure exch_r3
  end interface MPAS_dmpar_exch_halo_field
contains
  subroutine MPAS_dmpar_field_halo_exch(domain, fieldName, timeLevel, haloLayers)
    type(domain_type), intent(inout) :: domain
    character(len=*), intent(in) :: fieldName
    integer, intent(in), optional :: timeLevel
    integer, dimension(:), intent(in), optional :: haloLayers
  end subroutine MPAS_dmpar_field_halo_exch
  subroutine mpas_dmpar_min_real(dminfo, r, global_min)
    type(dm_info), intent(in) :: dminfo
    real(kind=RKIND), intent(in) :: r
    real(kind=RKIND), intent(out) :: global_min
  end subroutine mpas_dmpar_min_real
  subroutine mpas_dmpar_min_int(dminfo, i, global_min)
    type(dm_info), intent(in) :: dminfo
    integer, intent(in) :: i
    integer, intent(out) :: global_min
  end subroutine mpas_dmpar_min_int
  subroutine mpas_dmpar_sum_real(dminfo, r, global_sum)
    type(dm_info), intent(in) :: dminfo
    real(kind=RKIND), intent(in) :: r
    real(kind=RKIND), intent(out) :: global_sum
  end subroutine mpas_dmpar_sum_real
  subroutine exch_r3(field, haloLayersIn)
    type(field3DReal), pointer :: field
    integer, dimension(:), intent(in), optional :: haloLayersIn
  end subroutine exch_r3
end module mpass_dmpar

! the constants of the MPI library's own module that seaice_hip_transfer names; MPI_Allgather, MPI_Barrier,
! MPI_Comm_split_type, MPI_Comm_rank and MPI_Comm_free are called through implicit interfaces, as with most
! implementations of that module
module mpi
  implicit none
  integer, parameter :: MPI_SUCCESS = 0, MPI_BYTE = 1, MPI_COMM_TYPE_SHARED = 2, MPI_INFO_NULL = 3
end module mpi

! src/shared/mpas_seaice_error.F:44, :80 of the reference: stays in the build, called by the advection shim
module seaice_error
  use mpass_derived_types
  implicit none
contains
  subroutine seaice_critical_error_write_block(domain, block, abortFlag)
    type(domain_type), intent(in) :: domain
    type(block_type), intent(in) :: block
    logical, intent(in) :: abortFlag
    write(*,'(a,l1)') "seaice_critical_error_write_block abortFlag=", abortFlag
  end subroutine seaice_critical_error_write_block
  subroutine seaice_check_critical_error(domain, abortFlag)
    use mpass_log, only: mpas_log_write
    type(domain_type), intent(in) :: domain
    logical, intent(in) :: abortFlag
    write(*,'(a,l1)') "seaice_check_critical_error (vote of all ranks) abortFlag=", abortFlag
    if (abortFlag) call mpas_log_write("Runtime error", messageType=MPAS_LOG_CRIT)   ! mpas_seaice_error.F:111
  end subroutine seaice_check_critical_error
end module seaice_error

module mpass_rbf_interpolation
  use mpass_derived_types
  implicit none
contains
  subroutine mpas_rbf_interp_initialize(meshPool)
    type(MPAS_pool_type), intent(in) :: meshPool
  end subroutine mpas_rbf_interp_initialize
end module mpass_rbf_interpolation

module mpass_vector_reconstruction
  use mpass_derived_types
  implicit none
contains
  subroutine mpas_init_reconstruct(meshPool, includeHalos)
    type(MPAS_pool_type), intent(in) :: meshPool
    logical, intent(in), optional :: includeHalos
  end subroutine mpas_init_reconstruct
end module mpass_vector_reconstruction

! The tracer list of the incremental remap (src/shared/mpas_seaice_advection_incremental_remap_tracers.F:29-139,
! :158, :936, :1131) stays in the build; the shims use these components and entry points of it.
module seaice_advection_incremental_remap_tracers
  use mpass_derived_types
  implicit none
  type tracer_type
     character(len=StrKIND) :: tracerName, parentName
     integer :: ndims
     real(kind=RKIND), dimension(:,:), pointer :: array2D => null()
     real(kind=RKIND), dimension(:,:,:), pointer :: array3D => null()
     type(tracer_type), pointer :: next => null()
  end type tracer_type
  type(tracer_type), pointer :: tracersHead
contains
  subroutine seaice_add_tracers_to_linked_list(domain)
    type(domain_type), intent(in) :: domain
  end subroutine seaice_add_tracers_to_linked_list
  subroutine seaice_set_tracer_array_pointers(tracersHead, block, timeLevelIn)
    type(tracer_type), pointer, intent(in) :: tracersHead
    type(block_type), intent(inout) :: block
    integer, intent(in), optional :: timeLevelIn
  end subroutine seaice_set_tracer_array_pointers
  subroutine seaice_update_tracer_halo(tracersHead, domain, timeLevelIn)
    type(tracer_type), pointer, intent(in) :: tracersHead
    type(domain_type), intent(inout) :: domain
    integer, intent(in), optional :: timeLevelIn
  end subroutine seaice_update_tracer_halo
end module seaice_advection_incremental_remap_tracers
