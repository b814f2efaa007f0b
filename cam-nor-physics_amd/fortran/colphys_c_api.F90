! colphys_c_api -- ISO_C_BINDING view of include/colphys.h (the C ABI of libcolphys_hip.so).
! Field order of every bind(C) type must match the header.
module colphys_c_api
  use iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: COLPHYS_OK = 0
  integer(c_int), parameter :: COLPHYS_DYCORE_EUL = 0, COLPHYS_DYCORE_FV = 1, COLPHYS_DYCORE_SE = 2
  integer(c_int), parameter :: COLPHYS_STATUS_BRENT_NOT_CONVERGED = 1, COLPHYS_STATUS_DOPAER_RANGE = 2
  integer(c_int), parameter :: COLPHYS_STATUS_DOPAER_NEGATIVE = 4, COLPHYS_STATUS_BRENT_ENTHALPY = 8, COLPHYS_STATUS_COUNT_SHIFT = 8

  type, bind(C) :: colphys_zm_params
    real(c_double) :: cpair, epsilo, gravit, latice, latvap, tmelt, rair, cpwv, cpliq, rh2o, cpvir, zvir
    real(c_double) :: c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, tiedke_add, capelmt, dmpdz_param, tau
    real(c_double) :: tiedke_lnd, entrmn, alfadet, tentrm, plclmin, cin_threshd, parcel_hscale
    integer(c_int32_t) :: limcnv, num_cin, zm_org, zmconv_microp, no_deep_pbl, lparcel_pbl, cam3, reserved
  end type colphys_zm_params

  type, bind(C) :: colphys_zm_convr_io
    type(c_ptr) :: t, qh, pap, dpp, zm, paph, zi, geos, pblh, tpert, landfrac
    type(c_ptr) :: qtnd, heat, cme, zdu, rprd, dlf, ql, dif, dnlf, dnif, eurt, mcon, pflx
    type(c_ptr) :: prec, jctop, jcbot, cape, rliq, rice
    type(c_ptr) :: mu, md, du, eu, ed, dp, dsubcld
    type(c_ptr) :: jt, maxg, ideep, lengath, status
  end type colphys_zm_convr_io

  type, bind(C) :: colphys_zm_conv_evap_io
    type(c_ptr) :: t, pmid, pdel, q, prdprec, cldfrc, landfrac, prec
    type(c_ptr) :: tend_s, tend_s_snwprd, tend_s_snwevmlt, tend_q, ntprprd, ntsnprd, flxprec, flxsnow, snow
  end type colphys_zm_conv_evap_io

  type, bind(C) :: colphys_momtran_io
    type(c_ptr) :: winds, mu, md, du, eu, ed, dp, dsubcld
    type(c_ptr) :: jt, mx, ideep, lengath
    type(c_ptr) :: dqdt, pguall, pgdall, icwu, icwd, seten
  end type colphys_momtran_io

  type, bind(C) :: colphys_convtran_io
    type(c_ptr) :: q, fracis, mu, md, du, eu, ed, dp, dpdry
    type(c_ptr) :: jt, mx, ideep, lengath, doconvtran, isdry
    type(c_ptr) :: dqdt
  end type colphys_convtran_io

  type, bind(C) :: colphys_aero_tables
    integer(c_int32_t) :: nmodes, nspec_max, nswbands, nlwbands
    integer(c_int32_t) :: top_lev
    integer(c_int32_t) :: idx_sw_diag, idx_uv_diag, idx_nir_diag
    real(c_double) :: rhoh2o, rga, rair
    real(c_double) :: xrmin, xrmax
    type(c_ptr) :: nspec, spectype
    type(c_ptr) :: sigmag, specdens, hygro
    type(c_ptr) :: specrefindex_sw, specrefindex_lw
    type(c_ptr) :: refrtabsw, refitabsw, refrtablw, refitablw
    type(c_ptr) :: extpsw, abspsw, asmpsw, absplw
    type(c_ptr) :: crefwsw, crefwlw
  end type colphys_aero_tables

  type, bind(C) :: colphys_modal_aero_io
    type(c_ptr) :: pdeldry, pmid, t
    type(c_ptr) :: specmmr
    type(c_ptr) :: dgnumwet, qaerwat
    type(c_ptr) :: troplevchem
    type(c_ptr) :: tauxar, wa, ga, fa
    type(c_ptr) :: status
    type(c_ptr) :: vis_optics = c_null_ptr
  end type colphys_modal_aero_io

  ! ---- host face (include/colphys.h "HOST FACE") ----
  integer(c_int), parameter :: COLPHYS_HF_ZM = 1, COLPHYS_HF_AERO = 2, COLPHYS_HF_SWDIAG = 4, COLPHYS_HF_HIST = 8
  type, bind(C) :: colphys_hostface_times
     real(c_double) :: wall_ms, host_cpu_ms, h2d_ms, relayout_ms, physics_ms, d2h_ms
     integer(c_int64_t) :: h2d_bytes, d2h_bytes
     integer(c_int32_t) :: nslices, nstreams
  end type colphys_hostface_times


  interface
    integer(c_int) function colphys_device_count() bind(C, name='colphys_device_count')
      import
    end function
    integer(c_int) function colphys_is_initialized() bind(C, name='colphys_is_initialized')
      import
    end function
    subroutine colphys_zm_params_defaults(p) bind(C, name='colphys_zm_params_defaults')
      import
      type(colphys_zm_params), intent(out) :: p
    end subroutine
    integer(c_int) function colphys_aero_tables_set(tb) bind(C, name='colphys_aero_tables_set')
      import
      type(colphys_aero_tables), intent(in) :: tb
    end function
    integer(c_int) function colphys_modal_aero_sw_chunks(nchunks, pcols, pver, ncol, io) &
         bind(C, name='colphys_modal_aero_sw_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      type(colphys_modal_aero_io), intent(in) :: io
    end function
    integer(c_int) function colphys_modal_aero_sw_diag_chunks(nchunks, pcols, pver, ncol, io, lev, col) &
         bind(C, name='colphys_modal_aero_sw_diag_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      type(colphys_modal_aero_io), intent(in) :: io
      type(c_ptr), value :: lev, col
    end function
    integer(c_int) function colphys_modal_aero_lw_chunks(nchunks, pcols, pver, ncol, io) &
         bind(C, name='colphys_modal_aero_lw_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      type(colphys_modal_aero_io), intent(in) :: io
    end function
    integer(c_int) function colphys_init(device, p) bind(C, name='colphys_init')
      import
      integer(c_int), value :: device
      type(colphys_zm_params), intent(in) :: p
    end function
    integer(c_int) function colphys_finalize() bind(C, name='colphys_finalize')
      import
    end function
    type(c_ptr) function colphys_last_error() bind(C, name='colphys_last_error')
      import
    end function
    integer(c_int) function colphys_geopotential_t_chunks(nchunks, pcols, pver, ncol, dycore_kind, nq, nspecies, &
         piln, pmln, pint, pmid, pdel, rpdel, t, q, rair, gravit, zvir, zi, zm) bind(C, name='colphys_geopotential_t_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver, dycore_kind, nq, nspecies
      integer(c_int32_t), intent(in) :: ncol(*)
      type(c_ptr), value :: piln, pmln, pint, pmid, pdel, rpdel, t, q, rair, zvir, zi, zm
      real(c_double), value :: gravit
    end function
    integer(c_int) function colphys_geopotential_dse_chunks(nchunks, pcols, pver, ncol, dycore_kind, nq, nspecies, &
         piln, pmln, pint, pmid, pdel, rpdel, dse, q, phis, rair, gravit, cpair, zvir, t, zi, zm) &
         bind(C, name='colphys_geopotential_dse_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver, dycore_kind, nq, nspecies
      integer(c_int32_t), intent(in) :: ncol(*)
      type(c_ptr), value :: piln, pmln, pint, pmid, pdel, rpdel, dse, q, phis, rair, cpair, zvir, t, zi, zm
      real(c_double), value :: gravit
    end function
    integer(c_int) function colphys_zm_convr_chunks(nchunks, pcols, pver, ncol, delt, io) bind(C, name='colphys_zm_convr_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      real(c_double), value :: delt
      type(colphys_zm_convr_io), intent(in) :: io
    end function
    integer(c_int) function colphys_zm_conv_evap_chunks(nchunks, pcols, pver, ncol, deltat, io) &
         bind(C, name='colphys_zm_conv_evap_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      real(c_double), value :: deltat
      type(colphys_zm_conv_evap_io), intent(in) :: io
    end function
    integer(c_int) function colphys_momtran_chunks(nchunks, pcols, pver, ncol, dt, io) bind(C, name='colphys_momtran_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver
      integer(c_int32_t), intent(in) :: ncol(*)
      real(c_double), value :: dt
      type(colphys_momtran_io), intent(in) :: io
    end function
    integer(c_int) function colphys_convtran_chunks(nchunks, pcols, pver, ncol, ncnst, dt, io) &
         bind(C, name='colphys_convtran_chunks')
      import
      integer(c_int), value :: nchunks, pcols, pver, ncnst
      integer(c_int32_t), intent(in) :: ncol(*)
      real(c_double), value :: dt
      type(colphys_convtran_io), intent(in) :: io
    end function
     integer(c_int) function colphys_hostface_create(nchunks, pcols, pver, ncol, what, nslices, hf) bind(C, name='colphys_hostface_create')
       import
       integer(c_int), value :: nchunks, pcols, pver, what, nslices
       integer(c_int32_t), intent(in) :: ncol(*)
       type(c_ptr), intent(out) :: hf
     end function
     integer(c_int) function colphys_hostface_destroy(hf) bind(C, name='colphys_hostface_destroy')
       import
       type(c_ptr), value :: hf
     end function
     integer(c_int) function colphys_hostface_geometry(hf, in_rec, out_rec, nfields) bind(C, name='colphys_hostface_geometry')
       import
       type(c_ptr), value :: hf
       integer(c_int64_t), intent(out) :: in_rec, out_rec
       integer(c_int32_t), intent(out) :: nfields
     end function
     integer(c_int) function colphys_hostface_field(hf, i, name, nlev, is_int, is_out, offset) bind(C, name='colphys_hostface_field')
       import
       type(c_ptr), value :: hf
       integer(c_int), value :: i
       character(kind=c_char), intent(out) :: name(32)
       integer(c_int32_t), intent(out) :: nlev, is_int, is_out
       integer(c_int64_t), intent(out) :: offset
     end function
     type(c_ptr) function colphys_hostface_in_base(hf) bind(C, name='colphys_hostface_in_base')
       import
       type(c_ptr), value :: hf
     end function
     type(c_ptr) function colphys_hostface_out_base(hf) bind(C, name='colphys_hostface_out_base')
       import
       type(c_ptr), value :: hf
     end function
     integer(c_int) function colphys_hostface_run(hf, ztodt, dycore_kind, qmin) bind(C, name='colphys_hostface_run')
       import
       type(c_ptr), value :: hf
       real(c_double), value :: ztodt, qmin
       integer(c_int), value :: dycore_kind
     end function
     integer(c_int) function colphys_hostface_last_times(hf, t) bind(C, name='colphys_hostface_last_times')
       import
       type(c_ptr), value :: hf
       type(colphys_hostface_times), intent(out) :: t
     end function
  end interface

contains

  ! C string of colphys_last_error() as a Fortran string
  function colphys_error_message() result(msg)
    character(len=256) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: i
    msg = ' '
    p = colphys_last_error()
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, (/ 256 /))
    do i = 1, 256
      if (s(i) == c_null_char) exit
      msg(i:i) = s(i)
    end do
  end function colphys_error_message

end module colphys_c_api
