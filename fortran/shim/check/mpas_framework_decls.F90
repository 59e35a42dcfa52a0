! Declarations of the parts of the MPAS framework, and of the one sea-ice module that stays in the build, which
! fortran/shim/*.F use: types with the components the shims touch and procedures with empty bodies.  Their only
! purpose is `make check-shims` (amdflang -fsyntax-only), which type-checks the shim modules against these
! interfaces; nothing here is linked, run, or a substitute for the framework.

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
  type field3DReal
     real(kind=RKIND), dimension(:,:,:), pointer :: array => null()
  end type field3DReal
  type block_type
     type(MPAS_pool_type), pointer :: structs => null(), dimensions => null(), configs => null()
     type(block_type), pointer :: prev => null(), next => null()
  end type block_type
  type dm_info
     integer :: nprocs = 1, my_proc_id = 0
  end type dm_info
  type domain_type
     type(block_type), pointer :: blocklist => null()
     type(MPAS_pool_type), pointer :: configs => null()
     type(dm_info), pointer :: dminfo => null()
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
     module procedure fld_r3
  end interface MPAS_pool_get_field
contains
  subroutine get_r0(inPool, key, scalar, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), pointer :: scalar
    integer, intent(in), optional :: timeLevel
  end subroutine get_r0
  subroutine get_r1(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:), pointer :: array
    integer, intent(in), optional :: timeLevel
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
  end subroutine cfg_l
  subroutine cfg_c(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    character(len=StrKIND), pointer :: value
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
  subroutine MPAS_pool_get_subpool(inPool, key, subPool)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_type), pointer :: subPool
  end subroutine MPAS_pool_get_subpool
  subroutine MPAS_pool_get_field_info(inPool, key, info)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_field_info_type), intent(out) :: info
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
