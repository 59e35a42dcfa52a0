!> Walks the error path of the advection shim on the CPU: seaice_run_advection with a library call that reports
!> the incremental remap's run-time abort (SIH_ERR_IR, provided by abort_path_mock.c in place of the library's
!> sih_ir_run).  What the reference does at that point (mpas_seaice_advection_incremental_remap.F:3182-3183,
!> mpas_seaice_error.F:44-115) must come out in this order: the message as an error, the abort_block write of the
!> failing block, the vote of all ranks, and only then the critical "Runtime error".
program abort_path_main
  use mpass_derived_types
  use seaice_advection, only: seaice_run_advection
  implicit none
  type(domain_type) :: domain
  type(MPAS_Clock_type) :: clock
  allocate(domain % blocklist, domain % configs, domain % dminfo, domain % streamManager)
  allocate(domain % blocklist % structs, domain % blocklist % dimensions, domain % blocklist % configs)
  domain % dminfo % nprocs = 1
  call seaice_run_advection(domain, clock)
  write(*,'(a)') "seaice_run_advection returned"
end program abort_path_main
