! colphys_resident -- Fortran host side of the WHOLE-NODE dispatch (BASELINE north star: "host code stays Fortran").
!
! The per-chunk shim (zm_conv_hip.F90 ...) copies every argument to the device and back on each call.  This module is the
! other half of the design: a rank's physics_state / pbuf fields are uploaded ONCE per physics step into column-contiguous
! SoA device buffers ([column][level], include/colphys.h), the resident-data entry points of the C ABI (colphys_*_dev)
! run on them -- zm_conv_tend (zm_conv_intr.F90:441-905) is ONE call for all chunks of the rank -- and only what the host
! needs comes back.  Device memory and copies go through the HIP runtime's C entry points (bind(C) below): no wrapper
! library, no CUDA-compatibility layer.
!
! Layout contract.  Host arrays are CAM's: (pcols, nlev, nchunks), columns 1:ncol(c) of chunk c valid, the rest padding
! (+inf in CAM, physics_types.F90:2143-2178).  Device row r = (c-1)*pcols + i - 1 holds column i of chunk c; padding rows
! shadow column 1 of their chunk (finite, and they convect exactly when it does), so the kernels never see +inf and the
! gathered sets are trimmed back to the valid columns on download (colphys_download_gathered_sets).
module colphys_resident
  use iso_c_binding
  use colphys_c_api
  implicit none
  private
  save

  integer, parameter :: r8 = c_double
  integer(c_int), parameter :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2

  type, public :: colphys_field                     ! one device buffer [nrows][nlev]
     type(c_ptr) :: d = c_null_ptr
     integer :: nlev = 0
     integer :: nrows = 0
  end type colphys_field

  ! include/colphys.h: colphys_zm_conv_tend_io (field order = the header's)
  type, bind(C), public :: colphys_zm_conv_tend_io
     type(c_ptr) :: t, q, u, v, pmid, pint, pdel, rpdel, lnpint, lnpmid, zm, zi
     type(c_ptr) :: phis, ps
     type(c_ptr) :: rair, zvir
     type(c_ptr) :: pblh, tpert, landfrac
     type(c_ptr) :: cld
     type(c_ptr) :: ptend_s, ptend_q, ptend_u, ptend_v
     type(c_ptr) :: prec, snow, rliq, rice, jctop, jcbot, cape
     type(c_ptr) :: ql, rprd, evapcdp, dlf
     type(c_ptr) :: mconzm, flxprec, flxsnow
     type(c_ptr) :: mu, md, du, eu, ed, dp, dsubcld
     type(c_ptr) :: jt, maxg, ideep, lengath
     type(c_ptr) :: freqzm, pcont, pconb
     type(c_ptr) :: mu_out, md_out
     type(c_ptr) :: t1, q1, u1, v1, zi1, zm1
     type(c_ptr) :: status
     ! ABI 4: zm_conv_tend's remaining intent(out) arguments and its outfld-only quantities (null: kept in work arrays)
     type(c_ptr) :: cme = c_null_ptr, zdu = c_null_ptr, dif = c_null_ptr, eurt = c_null_ptr, pflx = c_null_ptr
     type(c_ptr) :: zmdt_s = c_null_ptr, zmdq = c_null_ptr
     type(c_ptr) :: evap_s = c_null_ptr, evap_s_snwprd = c_null_ptr, evap_s_snwevmlt = c_null_ptr
     type(c_ptr) :: ntprprd = c_null_ptr, ntsnprd = c_null_ptr, seten = c_null_ptr
     type(c_ptr) :: pguall = c_null_ptr, pgdall = c_null_ptr, icwu = c_null_ptr, icwd = c_null_ptr
  end type colphys_zm_conv_tend_io

  ! include/colphys.h: colphys_zm_conv_tend_2_io
  type, bind(C), public :: colphys_zm_conv_tend_2_io
     type(c_ptr) :: q, fracis, pdeldry, mu, md, du, eu, ed, dp
     type(c_ptr) :: jt, maxg, ideep, lq, isdry
     type(c_ptr) :: ptend_q
  end type colphys_zm_conv_tend_2_io

  interface
     integer(c_int) function hipMalloc(ptr, bytes) bind(C, name='hipMalloc')
       import
       type(c_ptr), intent(out) :: ptr
       integer(c_size_t), value :: bytes
     end function
     integer(c_int) function hipFree(ptr) bind(C, name='hipFree')
       import
       type(c_ptr), value :: ptr
     end function
     integer(c_int) function hipMemcpy(dst, src, bytes, kind) bind(C, name='hipMemcpy')
       import
       type(c_ptr), value :: dst, src
       integer(c_size_t), value :: bytes
       integer(c_int), value :: kind
     end function
     integer(c_int) function hipDeviceSynchronize() bind(C, name='hipDeviceSynchronize')
       import
     end function
     integer(c_int) function colphys_zm_conv_tend_dev(ncol, pver, pcols, ztodt, dycore_kind, qmin, io, stream) &
          bind(C, name='colphys_zm_conv_tend_dev')
       import
       integer(c_int), value :: ncol, pver, pcols, dycore_kind
       real(c_double), value :: ztodt, qmin
       type(colphys_zm_conv_tend_io), intent(in) :: io
       type(c_ptr), value :: stream
     end function
     integer(c_int) function colphys_zm_conv_tend_2_dev(ncol, pver, pcols, ncnst, ztodt, io, stream) &
          bind(C, name='colphys_zm_conv_tend_2_dev')
       import
       integer(c_int), value :: ncol, pver, pcols, ncnst
       real(c_double), value :: ztodt
       type(colphys_zm_conv_tend_2_io), intent(in) :: io
       type(c_ptr), value :: stream
     end function
     ! the optics on resident data (include/colphys.h); colphys_modal_aero_io is declared in colphys_c_api
     integer(c_int) function colphys_modal_aero_sw_dev(ncol, pver, io, stream) bind(C, name='colphys_modal_aero_sw_dev')
       import
       integer(c_int), value :: ncol, pver
       type(colphys_modal_aero_io), intent(in) :: io
       type(c_ptr), value :: stream
     end function
     integer(c_int) function colphys_modal_aero_lw_dev(ncol, pver, io, stream) bind(C, name='colphys_modal_aero_lw_dev')
       import
       integer(c_int), value :: ncol, pver
       type(colphys_modal_aero_io), intent(in) :: io
       type(c_ptr), value :: stream
     end function
     integer(c_int) function colphys_modal_aero_sw_diag_dev(ncol, pver, io, lev, col, stream) &
          bind(C, name='colphys_modal_aero_sw_diag_dev')
       import
       integer(c_int), value :: ncol, pver
       type(colphys_modal_aero_io), intent(in) :: io
       type(c_ptr), value :: lev, col, stream
     end function
  end interface

  public :: colphys_zm_conv_tend_dev, colphys_zm_conv_tend_2_dev
  public :: colphys_modal_aero_sw_dev, colphys_modal_aero_lw_dev, colphys_modal_aero_sw_diag_dev
  public :: colphys_alloc, colphys_alloc_int, colphys_release, colphys_upload, colphys_upload_multi, colphys_upload_int, &
            colphys_download, colphys_download_int, colphys_download_gathered_sets, colphys_sync
  public :: colphys_check_zm_status, colphys_check_optics_status

contains

  subroutine check(rc, what)
    use cam_abortutils, only: endrun
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    if (rc /= 0) call endrun('colphys_resident: '//what//' failed')
  end subroutine check

  ! ---- what the reference ends the run on, for a whole rank (the per-chunk shim does the same per call) ---------------------
  ! status(pcols, nchunks): the downloaded status words of colphys_zm_conv_tend_dev / colphys_zm_convr_dev
  subroutine colphys_check_zm_status(status, ncol)
    use cam_abortutils, only: endrun
    use cam_logfile,    only: iulog
    integer(c_int32_t), intent(in) :: status(:, :)
    integer, intent(in) :: ncol(:)
    integer :: c, i
    do c = 1, size(status, 2)
      do i = 1, ncol(c)
        if (iand(status(i, c), COLPHYS_STATUS_BRENT_NOT_CONVERGED) /= 0) then
          if (iand(status(i, c), COLPHYS_STATUS_BRENT_ENTHALPY) /= 0) then     ! the inversions a column meets first (zm_conv.F90:5102)
            write(iulog,*) '*** ZM_CONV: IENTHALPY: Failed and about to exit, chunk,icol= ', c, i
            call endrun('**** ZM_CONV IENTHALPY: Tmix did not converge ****')                 ! zm_conv.F90:5565
          end if
          write(iulog,*) '*** ZM_CONV: IENTROPY: Failed and about to exit, chunk,icol= ', c, i
          call endrun('**** ZM_CONV IENTROPY: Tmix did not converge ****')                    ! zm_conv.F90:5409
        end if
      end do
    end do
  end subroutine colphys_check_zm_status

  ! status words of colphys_modal_aero_sw_dev / _lw_dev.  nerr_dopaer: the caller's running count (the reference keeps one in
  ! modal_aero_sw and one in modal_aero_lw, SAVEd); nerrmax < 0: the count alone never halts (modal_aero_sw), 1000 in
  ! modal_aero_lw (modal_aer_opt.F90:619-620, 1024-1026, 1275-1276, 1432-1434)
  subroutine colphys_check_optics_status(subname, status, ncol, nerr_dopaer, nerrmax)
    use cam_abortutils, only: endrun
    use cam_logfile,    only: iulog
    character(len=*), intent(in) :: subname
    integer(c_int32_t), intent(in) :: status(:, :)
    integer, intent(in) :: ncol(:), nerrmax
    integer, intent(inout) :: nerr_dopaer
    integer :: c, i
    logical :: negative
    character(len=12) :: num
    negative = .false.
    do c = 1, size(status, 2)
      do i = 1, ncol(c)
        if (iand(status(i, c), COLPHYS_STATUS_DOPAER_RANGE) /= 0) then
          if (iand(status(i, c), COLPHYS_STATUS_DOPAER_NEGATIVE) /= 0) then
            write(iulog,*) 'ERROR: Negative aerosol optical depth in this layer.'
            negative = .true.
          else
            write(iulog,*) 'WARNING: Aerosol optical depth is unreasonably high in this layer.'
          end if
          write(iulog,*) subname//': chunk=', c, ' column=', i, ' layers out of range=', ishft(status(i, c), -COLPHYS_STATUS_COUNT_SHIFT)
          nerr_dopaer = nerr_dopaer + ishft(status(i, c), -COLPHYS_STATUS_COUNT_SHIFT)
        end if
      end do
    end do
    if (negative .or. (nerrmax >= 0 .and. nerr_dopaer >= nerrmax)) then
      write(num, '(i0)') nerr_dopaer
      write(iulog,*) '*** halting in ', subname, ' after nerr_dopaer =', nerr_dopaer
      call endrun('*** halting in '//subname//' after nerr_dopaer = '//trim(num))
    end if
  end subroutine colphys_check_optics_status

  subroutine colphys_alloc(f, nrows, nlev)          ! fp64 field
    type(colphys_field), intent(inout) :: f
    integer, intent(in) :: nrows, nlev
    call check(hipMalloc(f%d, int(nrows, c_size_t) * int(nlev, c_size_t) * 8_c_size_t), 'hipMalloc')
    f%nrows = nrows; f%nlev = nlev
  end subroutine colphys_alloc

  subroutine colphys_alloc_int(f, nrows)            ! int32 per-row field
    type(colphys_field), intent(inout) :: f
    integer, intent(in) :: nrows
    call check(hipMalloc(f%d, int(nrows, c_size_t) * 4_c_size_t), 'hipMalloc')
    f%nrows = nrows; f%nlev = 1
  end subroutine colphys_alloc_int

  subroutine colphys_release(f)
    type(colphys_field), intent(inout) :: f
    if (c_associated(f%d)) call check(hipFree(f%d), 'hipFree')
    f%d = c_null_ptr; f%nrows = 0; f%nlev = 0
  end subroutine colphys_release

  subroutine colphys_sync()
    call check(hipDeviceSynchronize(), 'hipDeviceSynchronize')
  end subroutine colphys_sync

  ! host (pcols, nlev, nchunks) -> device [nchunks*pcols][nlev]; padding rows shadow column 1 of their chunk
  subroutine colphys_upload(f, host, pcols, nlev, nchunks, ncol)
    type(colphys_field), intent(inout) :: f
    integer, intent(in) :: pcols, nlev, nchunks, ncol(nchunks)
    real(r8), intent(in) :: host(pcols, nlev, nchunks)
    real(r8), allocatable, target :: tmp(:,:)
    integer :: c, i, isrc
    if (.not. c_associated(f%d)) call colphys_alloc(f, nchunks*pcols, nlev)
    allocate(tmp(nlev, nchunks*pcols))
    do c = 1, nchunks
       do i = 1, pcols
          isrc = merge(i, 1, i <= ncol(c))
          tmp(:, (c-1)*pcols + i) = host(isrc, :, c)
       end do
    end do
    call check(hipMemcpy(f%d, c_loc(tmp), int(size(tmp), c_size_t) * 8_c_size_t, hipMemcpyHostToDevice), 'hipMemcpy (upload)')
  end subroutine colphys_upload

  ! nf fields at once, host (pcols, nlev, nf, nchunks) as CAM keeps constituents -> device [nf][nchunks*pcols][nlev]
  subroutine colphys_upload_multi(f, host, pcols, nlev, nf, nchunks, ncol)
    type(colphys_field), intent(inout) :: f
    integer, intent(in) :: pcols, nlev, nf, nchunks, ncol(nchunks)
    real(r8), intent(in) :: host(pcols, nlev, nf, nchunks)
    real(r8), allocatable, target :: tmp(:,:,:)
    integer :: c, i, m, isrc
    if (.not. c_associated(f%d)) call colphys_alloc(f, nf*nchunks*pcols, nlev)
    allocate(tmp(nlev, nchunks*pcols, nf))
    do c = 1, nchunks
       do m = 1, nf
          do i = 1, pcols
             isrc = merge(i, 1, i <= ncol(c))
             tmp(:, (c-1)*pcols + i, m) = host(isrc, :, m, c)
          end do
       end do
    end do
    call check(hipMemcpy(f%d, c_loc(tmp), int(size(tmp), c_size_t) * 8_c_size_t, hipMemcpyHostToDevice), 'hipMemcpy (upload)')
  end subroutine colphys_upload_multi

  ! per-column integers, host (pcols, nchunks) -> device [nchunks*pcols]; padding rows shadow column 1 of their chunk
  subroutine colphys_upload_int(f, host, pcols, nchunks, ncol)
    type(colphys_field), intent(inout) :: f
    integer, intent(in) :: pcols, nchunks, ncol(nchunks)
    integer, intent(in) :: host(pcols, nchunks)
    integer(c_int32_t), allocatable, target :: tmp(:)
    integer :: c, i
    if (.not. c_associated(f%d)) call colphys_alloc_int(f, nchunks*pcols)
    allocate(tmp(nchunks*pcols))
    do c = 1, nchunks
       do i = 1, pcols
          tmp((c-1)*pcols + i) = host(merge(i, 1, i <= ncol(c)), c)
       end do
    end do
    call check(hipMemcpy(f%d, c_loc(tmp), int(size(tmp), c_size_t) * 4_c_size_t, hipMemcpyHostToDevice), 'hipMemcpy (upload)')
  end subroutine colphys_upload_int

  ! device [nchunks*pcols][nlev] -> host (pcols, nlev, nchunks), the first nvalid(c) columns of each chunk
  subroutine colphys_download(f, host, pcols, nlev, nchunks, nvalid)
    type(colphys_field), intent(in) :: f
    integer, intent(in) :: pcols, nlev, nchunks, nvalid(nchunks)
    real(r8), intent(inout) :: host(pcols, nlev, nchunks)
    real(r8), allocatable, target :: tmp(:,:)
    integer :: c, i
    allocate(tmp(nlev, nchunks*pcols))
    call check(hipMemcpy(c_loc(tmp), f%d, int(size(tmp), c_size_t) * 8_c_size_t, hipMemcpyDeviceToHost), 'hipMemcpy (download)')
    do c = 1, nchunks
       do i = 1, nvalid(c)
          host(i, :, c) = tmp(:, (c-1)*pcols + i)
       end do
    end do
  end subroutine colphys_download

  subroutine colphys_download_int(f, host, pcols, nchunks, nvalid)
    type(colphys_field), intent(in) :: f
    integer, intent(in) :: pcols, nchunks, nvalid(nchunks)
    integer, intent(inout) :: host(pcols, nchunks)
    integer(c_int32_t), allocatable, target :: tmp(:)
    integer :: c, i
    allocate(tmp(nchunks*pcols))
    call check(hipMemcpy(c_loc(tmp), f%d, int(size(tmp), c_size_t) * 4_c_size_t, hipMemcpyDeviceToHost), 'hipMemcpy (download)')
    do c = 1, nchunks
       do i = 1, nvalid(c)
          host(i, c) = tmp((c-1)*pcols + i)
       end do
    end do
  end subroutine colphys_download_int

  ! lengath / ideep as zm_convr returns them per chunk (zm_conv.F90:905-940), with the shadow (padding) columns -- which
  ! sit at the tail of a chunk's ascending ideep list -- dropped: lengath(c) = count(ideep(1:len,c) <= ncol(c))
  subroutine colphys_download_gathered_sets(f_ideep, f_lengath, ideep, lengath, pcols, nchunks, ncol)
    type(colphys_field), intent(in) :: f_ideep, f_lengath
    integer, intent(in) :: pcols, nchunks, ncol(nchunks)
    integer, intent(out) :: ideep(pcols, nchunks), lengath(nchunks)
    integer(c_int32_t), allocatable, target :: tmp(:), len(:)
    integer :: c, j, n
    allocate(tmp(nchunks*pcols), len(nchunks))
    call check(hipMemcpy(c_loc(tmp), f_ideep%d, int(size(tmp), c_size_t) * 4_c_size_t, hipMemcpyDeviceToHost), 'hipMemcpy (ideep)')
    call check(hipMemcpy(c_loc(len), f_lengath%d, int(nchunks, c_size_t) * 4_c_size_t, hipMemcpyDeviceToHost), 'hipMemcpy (lengath)')
    ideep = 0
    do c = 1, nchunks
       n = 0
       do j = 1, len(c)
          if (tmp((c-1)*pcols + j) <= ncol(c)) then
             n = n + 1
             ideep(n, c) = tmp((c-1)*pcols + j)
          end if
       end do
       lengath(c) = n
    end do
  end subroutine colphys_download_gathered_sets

end module colphys_resident
