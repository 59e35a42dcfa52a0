! A small, working stand-in for the parts of the MPAS framework that fortran/shim/*.F use, so that the shim modules
! can be linked and RUN against libseaice_hip without MPAS: pools are name -> entry tables that hold sub-pools,
! dimensions, configs and fields with one or two time levels; one block, one rank; the halo routines and the
! reductions are those of a single rank (nothing to exchange, identity); mpas_log_write prints and stops on a
! critical message.  Test scaffolding for the shims only: it has none of the framework's I/O, decomposition or
! registry, and nothing in the library or the shims depends on it.
!
! The module and procedure names are the framework's, as the shims (and the reference sources they replace, e.g.
! src/shared/mpas_seaice_velocity_solver.F:17-22) import them.  The mini_* procedures are this file's own: they build
! the pools.

module mpass_derived_types
  implicit none
  integer, parameter :: RKIND = selected_real_kind(12)
  integer, parameter :: StrKIND = 512
  integer, parameter :: MPAS_LOG_OUT = 1, MPAS_LOG_WARN = 2, MPAS_LOG_ERR = 3, MPAS_LOG_CRIT = 4
  integer, parameter :: MINI_KEY = 64, MINI_CAPACITY = 256
  ! kinds of pool entries
  integer, parameter :: MINI_SUBPOOL = 1, MINI_DIM = 2, MINI_CFG_INT = 3, MINI_CFG_REAL = 4, MINI_CFG_LOGICAL = 5, &
       MINI_CFG_CHAR = 6, MINI_FIELD_REAL = 7, MINI_FIELD_INT = 8

  type MPAS_Clock_type
     integer :: direction = 0
  end type MPAS_Clock_type

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

  type field0DReal
     real(kind=RKIND), pointer :: scalar => null()
  end type field0DReal
  type field1DReal
     real(kind=RKIND), dimension(:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field1DReal
  type field2DReal
     real(kind=RKIND), dimension(:,:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field2DReal
  type field3DReal
     real(kind=RKIND), dimension(:,:,:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field3DReal
  type field1DInteger
     integer, dimension(:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field1DInteger
  type field2DInteger
     integer, dimension(:,:), pointer :: array => null()
     type(mpas_multihalo_exchange_list), pointer :: sendList => null(), recvList => null(), copyList => null()
  end type field2DInteger

  type MPAS_pool_field_info_type
     integer :: fieldType, nDims, nTimeLevels, nHaloLayers
     logical :: isActive
  end type MPAS_pool_field_info_type

  type mini_pool_entry
     character(len=MINI_KEY) :: key = ""
     integer :: kind = 0, nDims = 0, nTimeLevels = 1
     type(MPAS_pool_type), pointer :: sub => null()
     integer, pointer :: ival => null()
     real(kind=RKIND), pointer :: rval => null()
     logical, pointer :: lval => null()
     character(len=StrKIND), pointer :: cval => null()
     ! one element per time level
     type(field0DReal), dimension(:), pointer :: r0 => null()
     type(field1DReal), dimension(:), pointer :: r1 => null()
     type(field2DReal), dimension(:), pointer :: r2 => null()
     type(field3DReal), dimension(:), pointer :: r3 => null()
     type(field1DInteger), dimension(:), pointer :: i1 => null()
     type(field2DInteger), dimension(:), pointer :: i2 => null()
  end type mini_pool_entry

  type MPAS_pool_type
     integer :: size = 0
     character(len=MINI_KEY) :: name = ""
     type(mini_pool_entry), dimension(:), pointer :: entries => null()
  end type MPAS_pool_type

  type MPAS_streamManager_type
     integer :: numStreams = 0
  end type MPAS_streamManager_type
  type block_type
     type(MPAS_pool_type), pointer :: structs => null(), dimensions => null(), configs => null()
     type(block_type), pointer :: prev => null(), next => null()
     integer :: blockID = 0, localBlockID = 0
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

module mpass_log
  use mpass_derived_types
  implicit none
  integer, save :: mini_log_errors = 0, mini_log_warnings = 0
contains
  !> prints the message with its arguments substituted ($i, $r, $l in order) and stops the run on MPAS_LOG_CRIT
  subroutine mpas_log_write(message, messageType, masterOnly, flushNow, intArgs, realArgs, logicArgs, err)
    character(len=*), intent(in) :: message
    integer, intent(in), optional :: messageType
    logical, intent(in), optional :: masterOnly, flushNow
    integer, dimension(:), intent(in), optional :: intArgs
    real(kind=RKIND), dimension(:), intent(in), optional :: realArgs
    logical, dimension(:), intent(in), optional :: logicArgs
    integer, intent(out), optional :: err
    integer :: kind, i, ni, nr, nl
    character(len=2048) :: text
    character(len=32) :: piece
    kind = MPAS_LOG_OUT
    if (present(messageType)) kind = messageType
    text = ""
    ni = 0; nr = 0; nl = 0
    i = 1
    do while (i <= len_trim(message))
       if (message(i:i) == "$" .and. i < len(message)) then
          piece = ""
          select case (message(i+1:i+1))
          case ("i")
             ni = ni + 1
             if (present(intArgs)) then
                if (ni <= size(intArgs)) write(piece,'(i0)') intArgs(ni)
             endif
          case ("r")
             nr = nr + 1
             if (present(realArgs)) then
                if (nr <= size(realArgs)) write(piece,'(es23.15)') realArgs(nr)
             endif
          case ("l")
             nl = nl + 1
             if (present(logicArgs)) then
                if (nl <= size(logicArgs)) write(piece,'(l1)') logicArgs(nl)
             endif
          case default
             piece = message(i:i+1)
          end select
          text = trim(text)//" "//trim(adjustl(piece))
          i = i + 2
       else
          text = text(1:len_trim(text) + merge(1, 0, i > 1 .and. message(max(i-1,1):max(i-1,1)) == " "))//message(i:i)
          i = i + 1
       endif
    enddo
    write(*,'(a,i0,2a)') "mpas_log_write type ", kind, ": ", trim(text)
    if (kind == MPAS_LOG_WARN) mini_log_warnings = mini_log_warnings + 1
    if (kind == MPAS_LOG_ERR) mini_log_errors = mini_log_errors + 1
    if (present(err)) err = 0
    if (kind == MPAS_LOG_CRIT) stop 3   ! the framework aborts the run here
  end subroutine mpas_log_write
end module mpass_log

module mpass_pool_routines
  use mpass_derived_types
  use mpass_log, only: mpas_log_write
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
     module procedure fld_r1, fld_r2, fld_r3, fld_i1, fld_i2
  end interface MPAS_pool_get_field
contains

  ! ---- building pools (this file's own procedures) -------------------------------------------------------------

  subroutine mini_pool_create(pool, name)
    type(MPAS_pool_type), pointer :: pool
    character(len=*), intent(in) :: name
    allocate(pool)
    allocate(pool % entries(MINI_CAPACITY))
    pool % size = 0
    pool % name = name
  end subroutine mini_pool_create

  function mini_new_entry(pool, key, kind) result(e)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    integer, intent(in) :: kind
    type(mini_pool_entry), pointer :: e
    if (pool % size >= MINI_CAPACITY) call mpas_log_write("mini framework: pool "//trim(pool % name)//" is full", MPAS_LOG_CRIT)
    pool % size = pool % size + 1
    e => pool % entries(pool % size)
    e % key = key
    e % kind = kind
  end function mini_new_entry

  subroutine mini_pool_add_subpool(pool, key, sub)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    type(MPAS_pool_type), pointer :: sub
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_SUBPOOL)
    e % sub => sub
  end subroutine mini_pool_add_subpool

  subroutine mini_pool_add_dimension(pool, key, val)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    integer, intent(in) :: val
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_DIM)
    allocate(e % ival)
    e % ival = val
  end subroutine mini_pool_add_dimension

  subroutine mini_pool_add_config_int(pool, key, val)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    integer, intent(in) :: val
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_CFG_INT)
    allocate(e % ival)
    e % ival = val
  end subroutine mini_pool_add_config_int

  subroutine mini_pool_add_config_real(pool, key, val)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    real(kind=RKIND), intent(in) :: val
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_CFG_REAL)
    allocate(e % rval)
    e % rval = val
  end subroutine mini_pool_add_config_real

  subroutine mini_pool_add_config_logical(pool, key, val)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    logical, intent(in) :: val
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_CFG_LOGICAL)
    allocate(e % lval)
    e % lval = val
  end subroutine mini_pool_add_config_logical

  subroutine mini_pool_add_config_char(pool, key, val)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key, val
    type(mini_pool_entry), pointer :: e
    e => mini_new_entry(pool, key, MINI_CFG_CHAR)
    allocate(e % cval)
    e % cval = val
  end subroutine mini_pool_add_config_char

  !> a zero-filled field: nDims 0..3 (0: a scalar real), dims(1:nDims) its extents, one or two time levels
  subroutine mini_pool_add_field(pool, key, isInteger, nDims, dims, nTimeLevels)
    type(MPAS_pool_type), intent(inout) :: pool
    character(len=*), intent(in) :: key
    logical, intent(in) :: isInteger
    integer, intent(in) :: nDims, nTimeLevels
    integer, dimension(3), intent(in) :: dims
    type(mini_pool_entry), pointer :: e
    integer :: t
    e => mini_new_entry(pool, key, merge(MINI_FIELD_INT, MINI_FIELD_REAL, isInteger))
    e % nDims = nDims
    e % nTimeLevels = nTimeLevels
    if (isInteger) then
       select case (nDims)
       case (1)
          allocate(e % i1(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % i1(t) % array(dims(1)))
             e % i1(t) % array = 0
          enddo
       case (2)
          allocate(e % i2(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % i2(t) % array(dims(1), dims(2)))
             e % i2(t) % array = 0
          enddo
       case default
          call mpas_log_write("mini framework: integer field "//trim(key)//" with unsupported rank", MPAS_LOG_CRIT)
       end select
    else
       select case (nDims)
       case (0)
          allocate(e % r0(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % r0(t) % scalar)
             e % r0(t) % scalar = 0.0_RKIND
          enddo
       case (1)
          allocate(e % r1(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % r1(t) % array(dims(1)))
             e % r1(t) % array = 0.0_RKIND
          enddo
       case (2)
          allocate(e % r2(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % r2(t) % array(dims(1), dims(2)))
             e % r2(t) % array = 0.0_RKIND
          enddo
       case (3)
          allocate(e % r3(nTimeLevels))
          do t = 1, nTimeLevels
             allocate(e % r3(t) % array(dims(1), dims(2), dims(3)))
             e % r3(t) % array = 0.0_RKIND
          enddo
       case default
          call mpas_log_write("mini framework: real field "//trim(key)//" with unsupported rank", MPAS_LOG_CRIT)
       end select
    endif
  end subroutine mini_pool_add_field

  !> entry `key` of one of the kinds kindA / kindB; a missing entry is a critical error, as a missing Registry
  !> name would be in the framework
  function mini_find(pool, key, kindA, kindB, optional_) result(e)
    type(MPAS_pool_type), intent(in) :: pool
    character(len=*), intent(in) :: key
    integer, intent(in) :: kindA
    integer, intent(in), optional :: kindB
    logical, intent(in), optional :: optional_
    type(mini_pool_entry), pointer :: e
    integer :: i
    e => null()
    do i = 1, pool % size
       if (trim(pool % entries(i) % key) == trim(key)) then
          if (pool % entries(i) % kind == kindA) e => pool % entries(i)
          if (present(kindB)) then
             if (pool % entries(i) % kind == kindB) e => pool % entries(i)
          endif
          if (associated(e)) return
       endif
    enddo
    if (present(optional_)) then
       if (optional_) return
    endif
    call mpas_log_write("mini framework: pool '"//trim(pool % name)//"' has no entry '"//trim(key)//"' of the kind asked for", &
         MPAS_LOG_CRIT)
  end function mini_find

  integer function mini_level(e, timeLevel) result(t)
    type(mini_pool_entry), intent(in) :: e
    integer, intent(in), optional :: timeLevel
    t = 1
    if (present(timeLevel)) t = timeLevel
    if (t < 1 .or. t > e % nTimeLevels) call mpas_log_write("mini framework: field '"//trim(e % key)// &
         "' has no such time level", MPAS_LOG_CRIT)
  end function mini_level

  subroutine mini_wrong_rank(pool, key)
    type(MPAS_pool_type), intent(in) :: pool
    character(len=*), intent(in) :: key
    call mpas_log_write("mini framework: field '"//trim(key)//"' of pool '"//trim(pool % name)// &
         "' was asked for with another rank or type than it has", MPAS_LOG_CRIT)
  end subroutine mini_wrong_rank

  ! ---- the framework's pool interface ------------------------------------------------------------------------------

  subroutine get_r0(inPool, key, scalar, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), pointer :: scalar
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 0) call mini_wrong_rank(inPool, key)
    scalar => e % r0(mini_level(e, timeLevel)) % scalar
  end subroutine get_r0
  subroutine get_r1(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:), pointer :: array
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 1) call mini_wrong_rank(inPool, key)
    array => e % r1(mini_level(e, timeLevel)) % array
  end subroutine get_r1
  subroutine get_r2(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 2) call mini_wrong_rank(inPool, key)
    array => e % r2(mini_level(e, timeLevel)) % array
  end subroutine get_r2
  subroutine get_r3(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), dimension(:,:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 3) call mini_wrong_rank(inPool, key)
    array => e % r3(mini_level(e, timeLevel)) % array
  end subroutine get_r3
  subroutine get_i1(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, dimension(:), pointer :: array
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_INT)
    if (e % nDims /= 1) call mini_wrong_rank(inPool, key)
    array => e % i1(mini_level(e, timeLevel)) % array
  end subroutine get_i1
  subroutine get_i2(inPool, key, array, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, dimension(:,:), pointer :: array
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_INT)
    if (e % nDims /= 2) call mini_wrong_rank(inPool, key)
    array => e % i2(mini_level(e, timeLevel)) % array
  end subroutine get_i2

  subroutine cfg_r(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    real(kind=RKIND), pointer :: value
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_CFG_REAL)
    value => e % rval
  end subroutine cfg_r
  subroutine cfg_i(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, pointer :: value
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_CFG_INT)
    value => e % ival
  end subroutine cfg_i
  subroutine cfg_l(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    logical, pointer :: value
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_CFG_LOGICAL)
    value => e % lval
  end subroutine cfg_l
  subroutine cfg_c(inPool, key, value)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    character(len=StrKIND), pointer :: value
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_CFG_CHAR)
    value => e % cval
  end subroutine cfg_c

  subroutine dim_0(inPool, key, dim)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    integer, pointer :: dim
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_DIM)
    dim => e % ival
  end subroutine dim_0

  subroutine fld_r1(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field1DReal), pointer :: field
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 1) call mini_wrong_rank(inPool, key)
    field => e % r1(mini_level(e, timeLevel))
  end subroutine fld_r1
  subroutine fld_r2(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field2DReal), pointer :: field
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 2) call mini_wrong_rank(inPool, key)
    field => e % r2(mini_level(e, timeLevel))
  end subroutine fld_r2
  subroutine fld_r3(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field3DReal), pointer :: field
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL)
    if (e % nDims /= 3) call mini_wrong_rank(inPool, key)
    field => e % r3(mini_level(e, timeLevel))
  end subroutine fld_r3
  subroutine fld_i1(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field1DInteger), pointer :: field
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_INT)
    if (e % nDims /= 1) call mini_wrong_rank(inPool, key)
    field => e % i1(mini_level(e, timeLevel))
  end subroutine fld_i1
  subroutine fld_i2(inPool, key, field, timeLevel)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(field2DInteger), pointer :: field
    integer, intent(in), optional :: timeLevel
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_INT)
    if (e % nDims /= 2) call mini_wrong_rank(inPool, key)
    field => e % i2(mini_level(e, timeLevel))
  end subroutine fld_i2

  subroutine MPAS_pool_get_subpool(inPool, key, subPool)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_type), pointer :: subPool
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_SUBPOOL)
    subPool => e % sub
  end subroutine MPAS_pool_get_subpool

  subroutine MPAS_pool_get_field_info(inPool, key, info)
    type(MPAS_pool_type), intent(in) :: inPool
    character(len=*), intent(in) :: key
    type(MPAS_pool_field_info_type), intent(out) :: info
    type(mini_pool_entry), pointer :: e
    e => mini_find(inPool, key, MINI_FIELD_REAL, MINI_FIELD_INT)
    info % fieldType = merge(1, 0, e % kind == MINI_FIELD_INT)
    info % nDims = e % nDims
    info % nTimeLevels = e % nTimeLevels
    info % nHaloLayers = 0
    info % isActive = .true.
  end subroutine MPAS_pool_get_field_info

  !> fields with two time levels: the levels trade places (what the framework's shift does to two levels)
  subroutine MPAS_pool_shift_time_levels(inPool)
    type(MPAS_pool_type), intent(inout) :: inPool
    type(mini_pool_entry), pointer :: e
    real(kind=RKIND), pointer :: s0
    real(kind=RKIND), dimension(:), pointer :: s1
    real(kind=RKIND), dimension(:,:), pointer :: s2
    real(kind=RKIND), dimension(:,:,:), pointer :: s3
    integer, dimension(:), pointer :: t1
    integer, dimension(:,:), pointer :: t2
    integer :: i
    do i = 1, inPool % size
       e => inPool % entries(i)
       if (e % nTimeLevels /= 2) cycle
       if (e % kind == MINI_FIELD_REAL) then
          select case (e % nDims)
          case (0); s0 => e % r0(1) % scalar; e % r0(1) % scalar => e % r0(2) % scalar; e % r0(2) % scalar => s0
          case (1); s1 => e % r1(1) % array; e % r1(1) % array => e % r1(2) % array; e % r1(2) % array => s1
          case (2); s2 => e % r2(1) % array; e % r2(1) % array => e % r2(2) % array; e % r2(2) % array => s2
          case (3); s3 => e % r3(1) % array; e % r3(1) % array => e % r3(2) % array; e % r3(2) % array => s3
          end select
       else if (e % kind == MINI_FIELD_INT) then
          select case (e % nDims)
          case (1); t1 => e % i1(1) % array; e % i1(1) % array => e % i1(2) % array; e % i1(2) % array => t1
          case (2); t2 => e % i2(1) % array; e % i2(1) % array => e % i2(2) % array; e % i2(2) % array => t2
          end select
       endif
    enddo
  end subroutine MPAS_pool_shift_time_levels
end module mpass_pool_routines

module mpass_timer
  implicit none
  integer, parameter :: MINI_MAX_TIMERS = 64
  character(len=64), dimension(MINI_MAX_TIMERS), save :: mini_timer_names = ""
  integer(kind=8), dimension(MINI_MAX_TIMERS), save :: mini_timer_started = 0, mini_timer_ticks = 0
  integer, dimension(MINI_MAX_TIMERS), save :: mini_timer_calls = 0
  integer, save :: mini_timer_count = 0
contains
  integer function mini_timer_index(timer_name) result(i)
    character(len=*), intent(in) :: timer_name
    do i = 1, mini_timer_count
       if (trim(mini_timer_names(i)) == trim(timer_name)) return
    enddo
    if (mini_timer_count < MINI_MAX_TIMERS) mini_timer_count = mini_timer_count + 1
    i = mini_timer_count
    mini_timer_names(i) = timer_name
  end function mini_timer_index
  subroutine mpas_timer_start(timer_name)
    character(len=*), intent(in) :: timer_name
    call system_clock(mini_timer_started(mini_timer_index(timer_name)))
  end subroutine mpas_timer_start
  subroutine mpas_timer_stop(timer_name)
    character(len=*), intent(in) :: timer_name
    integer(kind=8) :: now
    integer :: i
    call system_clock(now)
    i = mini_timer_index(timer_name)
    mini_timer_ticks(i) = mini_timer_ticks(i) + now - mini_timer_started(i)
    mini_timer_calls(i) = mini_timer_calls(i) + 1
  end subroutine mpas_timer_stop
  !> the timers the shims started and stopped, in the order of their first start
  subroutine mini_timer_report()
    integer(kind=8) :: rate
    integer :: i
    call system_clock(count_rate=rate)
    do i = 1, mini_timer_count
       write(*,'(3a,i0,a,f12.3)') "timer '", trim(mini_timer_names(i)), "' calls ", mini_timer_calls(i), " ms ", &
            real(mini_timer_ticks(i), 8) / real(rate, 8) * 1.0e3_8
    enddo
  end subroutine mini_timer_report
  subroutine mini_timer_reset()
    mini_timer_ticks = 0
    mini_timer_calls = 0
  end subroutine mini_timer_reset
end module mpass_timer

module mpass_timekeeping
  use mpass_derived_types
  implicit none
end module mpass_timekeeping

!> one rank: a halo update has nothing to bring, a reduction returns its argument
module mpass_dmpar
  use mpass_derived_types
  implicit none
  integer, save :: mini_halo_calls = 0
  interface MPAS_dmpar_exch_halo_field
     module procedure exch_r3
  end interface MPAS_dmpar_exch_halo_field
contains
  subroutine MPAS_dmpar_field_halo_exch(domain, fieldName, timeLevel, haloLayers)
    type(domain_type), intent(inout) :: domain
    character(len=*), intent(in) :: fieldName
    integer, intent(in), optional :: timeLevel
    integer, dimension(:), intent(in), optional :: haloLayers
    mini_halo_calls = mini_halo_calls + 1
  end subroutine MPAS_dmpar_field_halo_exch
  subroutine mpas_dmpar_min_real(dminfo, r, global_min)
    type(dm_info), intent(in) :: dminfo
    real(kind=RKIND), intent(in) :: r
    real(kind=RKIND), intent(out) :: global_min
    global_min = r
  end subroutine mpas_dmpar_min_real
  subroutine mpas_dmpar_min_int(dminfo, i, global_min)
    type(dm_info), intent(in) :: dminfo
    integer, intent(in) :: i
    integer, intent(out) :: global_min
    global_min = i
  end subroutine mpas_dmpar_min_int
  subroutine mpas_dmpar_sum_real(dminfo, r, global_sum)
    type(dm_info), intent(in) :: dminfo
    real(kind=RKIND), intent(in) :: r
    real(kind=RKIND), intent(out) :: global_sum
    global_sum = r
  end subroutine mpas_dmpar_sum_real
  subroutine exch_r3(field, haloLayersIn)
    type(field3DReal), pointer :: field
    integer, dimension(:), intent(in), optional :: haloLayersIn
    mini_halo_calls = mini_halo_calls + 1
  end subroutine exch_r3
end module mpass_dmpar

! src/shared/mpas_seaice_error.F:44, :80 of the reference stays in an MPAS build and is called by the advection shim;
! here its two entry points say what they are asked to do and keep the reference's last step, the critical message
module seaice_error
  use mpass_derived_types
  implicit none
contains
  subroutine seaice_critical_error_write_block(domain, block, abortFlag)
    type(domain_type), intent(in) :: domain
    type(block_type), intent(in) :: block
    logical, intent(in) :: abortFlag
    if (abortFlag) write(*,'(a,l1)') "seaice_critical_error_write_block abortFlag=", abortFlag
  end subroutine seaice_critical_error_write_block
  subroutine seaice_check_critical_error(domain, abortFlag)
    use mpass_log, only: mpas_log_write
    type(domain_type), intent(in) :: domain
    logical, intent(in) :: abortFlag
    if (abortFlag) then
       write(*,'(a,l1)') "seaice_check_critical_error (vote of all ranks) abortFlag=", abortFlag
       call mpas_log_write("Runtime error", messageType=MPAS_LOG_CRIT)   ! mpas_seaice_error.F:111
    endif
  end subroutine seaice_check_critical_error
end module seaice_error

! coeffs_reconstruct is the output of these two framework operators (called at
! src/shared/mpas_seaice_advection_incremental_remap.F:744-746); the harness reads the array into the mesh pool with
! the rest of the case, so the calls have nothing left to compute
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

! The tracer list of the incremental remap (src/shared/mpas_seaice_advection_incremental_remap_tracers.F:29-139, :158,
! :936, :1131) stays in an MPAS build.  This stand-in offers the components and entry points the shims use: the list
! is built from (name, parent) pairs in the order the harness hands over (mini_set_tracer_table: the reference's order,
! parents before children), and the pointers are bound to the `tracers` pool as the reference binds them (a tracer
! with one layer as array2D => array(1,:,:), :1053).
module seaice_advection_incremental_remap_tracers
  use mpass_derived_types
  use mpass_pool_routines
  implicit none
  type tracer_type
     character(len=StrKIND) :: tracerName, parentName
     integer :: ndims
     real(kind=RKIND), dimension(:,:), pointer :: array2D => null()
     real(kind=RKIND), dimension(:,:,:), pointer :: array3D => null()
     type(tracer_type), pointer :: next => null()
  end type tracer_type
  type(tracer_type), pointer :: tracersHead => null()
  integer, parameter :: MINI_MAX_TRACERS = 64
  character(len=MINI_KEY), dimension(MINI_MAX_TRACERS), save :: mini_tracer_names = "", mini_tracer_parents = ""
  integer, save :: mini_n_tracers = 0
contains
  subroutine mini_set_tracer_table(n, names, parents)
    integer, intent(in) :: n
    character(len=*), dimension(:), intent(in) :: names, parents
    mini_n_tracers = n
    mini_tracer_names(1:n) = names(1:n)
    mini_tracer_parents(1:n) = parents(1:n)
  end subroutine mini_set_tracer_table
  subroutine seaice_add_tracers_to_linked_list(domain)
    type(domain_type), intent(in) :: domain
    type(tracer_type), pointer :: thisTracer, last
    integer :: i
    tracersHead => null()
    last => null()
    do i = 1, mini_n_tracers
       allocate(thisTracer)
       thisTracer % tracerName = mini_tracer_names(i)
       thisTracer % parentName = mini_tracer_parents(i)
       thisTracer % ndims = 3
       if (associated(last)) then
          last % next => thisTracer
       else
          tracersHead => thisTracer
       endif
       last => thisTracer
    enddo
  end subroutine seaice_add_tracers_to_linked_list
  subroutine seaice_set_tracer_array_pointers(tracersHead, block, timeLevelIn)
    type(tracer_type), pointer, intent(in) :: tracersHead
    type(block_type), intent(inout) :: block
    integer, intent(in), optional :: timeLevelIn
    type(tracer_type), pointer :: thisTracer
    type(MPAS_pool_type), pointer :: tracers
    real(kind=RKIND), dimension(:,:,:), pointer :: a
    integer :: tl
    tl = 1
    if (present(timeLevelIn)) tl = timeLevelIn
    call MPAS_pool_get_subpool(block % structs, "tracers", tracers)
    thisTracer => tracersHead
    do while (associated(thisTracer))
       call MPAS_pool_get_array(tracers, trim(thisTracer % tracerName), a, tl)
       if (size(a, 1) == 1) then
          thisTracer % ndims = 2
          thisTracer % array2D => a(1,:,:)
          thisTracer % array3D => null()
       else
          thisTracer % ndims = 3
          thisTracer % array3D => a
          thisTracer % array2D => null()
       endif
       thisTracer => thisTracer % next
    enddo
  end subroutine seaice_set_tracer_array_pointers
  subroutine seaice_update_tracer_halo(tracersHead, domain, timeLevelIn)
    type(tracer_type), pointer, intent(in) :: tracersHead
    type(domain_type), intent(inout) :: domain
    integer, intent(in), optional :: timeLevelIn
  end subroutine seaice_update_tracer_halo
end module seaice_advection_incremental_remap_tracers
