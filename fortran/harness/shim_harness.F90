!> Runs the MPAS-side shim modules (fortran/shim/*.F: seaice_mesh_pool, seaice_velocity_solver, seaice_advection, with
!> the names and signatures of the reference modules they replace) against libseaice_hip on the GPU, on pools built by
!> the mini framework of this directory from a case file, and writes every pool field back to a result file.
!>
!> usage: shim_harness <case file> <result file> [steps] [time]
!>
!> The call sequence is the reference's: seaice_init (src/shared/mpas_seaice_initialize.F:104, :121, :126) creates the
!> mesh pool and initialises the velocity solver and the advection; every dynamics step of seaice_timestep
!> (src/shared/mpas_seaice_time_integration.F:142-158) runs the velocity solver, then the advection, then -- with
!> config_use_column_package -- the column package's ridging and aggregation (seaice_column_dynamics_hip).  With a fourth
!> argument the steps are also timed (the shims' own timers, which name the transfers and the solver).
!>
!> Case file (little-endian stream, written by mpas_seaice_amd/fortran_case.py:write_pools): magic; configs (key, type,
!> value); dimensions (key, value); the ordered tracer list (name, parent); fields (pool, key, integer or real, rank,
!> extents, time levels, data of every level).  The result file holds the field records only.
program shim_harness

  use iso_c_binding
  use mpass_derived_types
  use mpass_pool_routines
  use mpass_log, only: mpas_log_write, mini_log_errors
  use mpass_timer, only: mini_timer_report, mini_timer_reset
  use seaice_advection_incremental_remap_tracers, only: mini_set_tracer_table
  use seaice_mesh_pool, only: seaice_mesh_pool_create, seaice_mesh_pool_destroy
  use seaice_velocity_solver, only: seaice_init_velocity_solver, seaice_run_velocity_solver, &
       seaice_velocity_solver_fetch_diagnostics, seaice_finalize_velocity_solver
  use seaice_advection, only: seaice_init_advection, seaice_run_advection
  use seaice_column_dynamics_hip, only: seaice_column_reinitialize_diagnostics_dynamics, &
       seaice_column_dynamics_time_integration

  implicit none

  integer(c_int32_t), parameter :: MAGIC = 1397311537   ! "SIH1"
  character(len=1024) :: caseFile, resultFile, arg
  type(domain_type) :: domain
  type(MPAS_Clock_type) :: clock
  type(block_type), pointer :: block
  integer :: nSteps, iStep, err, rep
  logical :: timed
  integer(kind=8) :: t0, t1, rate

  if (command_argument_count() < 2) then
     write(*,*) "usage: shim_harness <case file> <result file> [steps] [time]"
     stop 2
  endif
  call get_command_argument(1, caseFile)
  call get_command_argument(2, resultFile)
  nSteps = 1
  if (command_argument_count() >= 3) then
     call get_command_argument(3, arg)
     read(arg, *) nSteps
  endif
  timed = command_argument_count() >= 4

  call load_case(trim(caseFile))
  block => domain % blocklist

  ! seaice_init
  call seaice_mesh_pool_create(domain)
  call seaice_init_velocity_solver(domain)
  call seaice_init_advection(domain)

  ! seaice_timestep, dynamics part
  do iStep = 1, nSteps
     call seaice_column_reinitialize_diagnostics_dynamics(domain)   ! :142 (nothing without the column package)
     call seaice_run_velocity_solver(domain, clock)                 ! :147
     call seaice_run_advection(domain, clock)                       ! :153
     call seaice_column_dynamics_time_integration(domain, clock)    ! :158: ridging and aggregate state
  enddo
  call seaice_velocity_solver_fetch_diagnostics(domain)   ! nothing unless the diagnostics were left on the device

  call dump_pools(trim(resultFile))

  if (timed) then
     call system_clock(count_rate=rate)
     do rep = 0, 5   ! rep 0 warms up
        if (rep == 1) then
           call mini_timer_reset()
           call system_clock(t0)
        endif
        call seaice_run_velocity_solver(domain, clock)
     enddo
     call system_clock(t1)
     write(*,'(a,f10.3)') "shim step ms, seaice_run_velocity_solver: ", real(t1 - t0, 8) / real(rate, 8) * 1.0e3_8 / 5
     call mini_timer_report()
     do rep = 0, 5
        if (rep == 1) then
           call mini_timer_reset()
           call system_clock(t0)
        endif
        call seaice_run_velocity_solver(domain, clock)
        call seaice_run_advection(domain, clock)
     enddo
     call system_clock(t1)
     write(*,'(a,f10.3)') "shim step ms, seaice_run_velocity_solver + seaice_run_advection: ", &
          real(t1 - t0, 8) / real(rate, 8) * 1.0e3_8 / 5
     call mini_timer_report()
  endif

  call seaice_finalize_velocity_solver()
  call seaice_mesh_pool_destroy(err)
  if (err /= 0) then
     write(*,*) "seaice_mesh_pool_destroy failed"
     stop 1
  endif
  write(*,'(a,i0,a,i0)') "shim_harness ok, steps ", nSteps, ", logged errors ", mini_log_errors

contains

  function get_or_make_pool(name) result(pool)
    character(len=*), intent(in) :: name
    type(MPAS_pool_type), pointer :: pool
    type(mini_pool_entry), pointer :: e
    e => mini_find(domain % blocklist % structs, name, MINI_SUBPOOL, optional_=.true.)
    if (associated(e)) then
       pool => e % sub
    else
       call mini_pool_create(pool, name)
       call mini_pool_add_subpool(domain % blocklist % structs, name, pool)
    endif
  end function get_or_make_pool

  subroutine load_case(path)
    character(len=*), intent(in) :: path
    integer :: iu, i, t
    integer(c_int32_t) :: word, n, kind, ival, isInt, nDims, nLevels
    integer(c_int32_t), dimension(3) :: dims
    real(c_double) :: rval
    character(len=MINI_KEY) :: key, poolName, cval
    character(len=MINI_KEY), dimension(:), allocatable :: names, parents
    type(MPAS_pool_type), pointer :: pool, mesh
    type(mini_pool_entry), pointer :: e

    allocate(domain % blocklist, domain % dminfo, domain % streamManager)
    call mini_pool_create(domain % configs, "configs")
    call mini_pool_create(domain % blocklist % structs, "structs")
    call mini_pool_create(domain % blocklist % dimensions, "dimensions")
    domain % blocklist % configs => domain % configs
    domain % dminfo % nprocs = 1
    domain % dminfo % my_proc_id = 0
    mesh => get_or_make_pool("mesh")

    open(newunit=iu, file=path, access="stream", form="unformatted", status="old")
    read(iu) word
    if (word /= MAGIC) then
       write(*,*) "not a case file of mpas_seaice_amd/fortran_case.py:write_pools: ", path
       stop 2
    endif
    read(iu) n
    do i = 1, n
       read(iu) key, kind
       select case (kind)
       case (1)
          read(iu) ival
          call mini_pool_add_config_int(domain % configs, trim(key), int(ival))
       case (2)
          read(iu) rval
          call mini_pool_add_config_real(domain % configs, trim(key), real(rval, RKIND))
          ! attributes of the mesh, read by the shims from the mesh pool
          if (trim(key) == "sphere_radius") call mini_pool_add_config_real(mesh, trim(key), real(rval, RKIND))
       case (3)
          read(iu) ival
          call mini_pool_add_config_logical(domain % configs, trim(key), ival /= 0)
          if (trim(key) == "on_a_sphere") call mini_pool_add_config_logical(mesh, trim(key), ival /= 0)
       case default
          read(iu) cval
          call mini_pool_add_config_char(domain % configs, trim(key), trim(cval))
       end select
    enddo
    read(iu) n
    do i = 1, n
       read(iu) key, ival
       call mini_pool_add_dimension(domain % blocklist % dimensions, trim(key), int(ival))
       call mini_pool_add_dimension(mesh, trim(key), int(ival))
    enddo
    read(iu) n
    allocate(names(max(n, 1)), parents(max(n, 1)))
    do i = 1, n
       read(iu) names(i), parents(i)
    enddo
    call mini_set_tracer_table(int(n), names, parents)
    read(iu) n
    do i = 1, n
       read(iu) poolName, key, isInt, nDims, dims, nLevels
       pool => get_or_make_pool(trim(poolName))
       call mini_pool_add_field(pool, trim(key), isInt /= 0, int(nDims), int(dims), int(nLevels))
       e => pool % entries(pool % size)
       do t = 1, nLevels
          if (isInt /= 0) then
             if (nDims == 1) then
                read(iu) e % i1(t) % array
             else
                read(iu) e % i2(t) % array
             endif
          else
             select case (nDims)
             case (0); read(iu) e % r0(t) % scalar
             case (1); read(iu) e % r1(t) % array
             case (2); read(iu) e % r2(t) % array
             case (3); read(iu) e % r3(t) % array
             end select
          endif
       enddo
    enddo
    close(iu)
  end subroutine load_case

  subroutine dump_pools(path)
    character(len=*), intent(in) :: path
    integer :: ou, ip, i, t
    integer(c_int32_t) :: n
    integer(c_int32_t), dimension(3) :: dims
    type(MPAS_pool_type), pointer :: structs, pool
    type(mini_pool_entry), pointer :: e

    structs => domain % blocklist % structs
    n = 0
    do ip = 1, structs % size
       pool => structs % entries(ip) % sub
       do i = 1, pool % size
          if (pool % entries(i) % kind == MINI_FIELD_REAL .or. pool % entries(i) % kind == MINI_FIELD_INT) n = n + 1
       enddo
    enddo
    open(newunit=ou, file=path, access="stream", form="unformatted", status="replace")
    write(ou) MAGIC, n
    do ip = 1, structs % size
       pool => structs % entries(ip) % sub
       do i = 1, pool % size
          e => pool % entries(i)
          if (e % kind /= MINI_FIELD_REAL .and. e % kind /= MINI_FIELD_INT) cycle
          dims = 1
          if (e % kind == MINI_FIELD_INT) then
             if (e % nDims == 1) dims(1:1) = shape(e % i1(1) % array)
             if (e % nDims == 2) dims(1:2) = shape(e % i2(1) % array)
          else
             if (e % nDims == 1) dims(1:1) = shape(e % r1(1) % array)
             if (e % nDims == 2) dims(1:2) = shape(e % r2(1) % array)
             if (e % nDims == 3) dims(1:3) = shape(e % r3(1) % array)
          endif
          write(ou) structs % entries(ip) % key, e % key, int(merge(1, 0, e % kind == MINI_FIELD_INT), c_int32_t), &
               int(e % nDims, c_int32_t), dims, int(e % nTimeLevels, c_int32_t)
          do t = 1, e % nTimeLevels
             if (e % kind == MINI_FIELD_INT) then
                if (e % nDims == 1) write(ou) e % i1(t) % array
                if (e % nDims == 2) write(ou) e % i2(t) % array
             else
                select case (e % nDims)
                case (0); write(ou) e % r0(t) % scalar
                case (1); write(ou) e % r1(t) % array
                case (2); write(ou) e % r2(t) % array
                case (3); write(ou) e % r3(t) % array
                end select
             endif
          enddo
       enddo
    enddo
    close(ou)
  end subroutine dump_pools

end program shim_harness
