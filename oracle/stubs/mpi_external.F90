! TEST INFRASTRUCTURE -- not product code.
! zm_conv_readnl (zm_conv_intr.F90:211-238) broadcasts its namelist with the MPI library's external mpi_bcast; a one-process
! oracle has nothing to broadcast.  External procedure (no module), as an MPI-1 Fortran binding is.
subroutine mpi_bcast(buffer, count, datatype, root, comm, ierror)
  integer :: buffer(*)
  integer :: count, datatype, root, comm, ierror
  ierror = 0
end subroutine mpi_bcast
