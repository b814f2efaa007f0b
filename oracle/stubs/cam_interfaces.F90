! TEST INFRASTRUCTURE -- not product code.
!
! Interface-only stand-ins for the parent-model (ESCOMP/CAM, CESM share) modules that the reference's
! zm_conv.F90 / geopotential.F90 `use` but that are NOT part of NorESMhub/CAM-Nor-physics (SURVEY.md
! Appendix A).  Nothing in this file carries hot-path arithmetic: kinds, grid sizes, constants, logging,
! abort, history no-ops.  The four upstream modules that DO carry arithmetic live in l0_arithmetic.F90.
!
! Constants are the CESM share-code values (shr_const_mod); the product never hard-codes them -- they are
! handed to the C-ABI through colphys_params at init (include/colphys.h).
!
! PCOLS / PVER come from the compiler command line (oracle/Makefile), as CAM's ppgrid gets them from cpp.

module shr_kind_mod
  integer, parameter :: shr_kind_r8 = selected_real_kind(12)
  integer, parameter :: shr_kind_cl = 256
end module shr_kind_mod

module spmd_utils
  ! masterproc=.true.: zm_convi resets tentrm=-dmpdz_param only on masterproc (zm_conv.F90:185,213)
  logical :: masterproc = .true.
  integer :: mpicom = 0, masterprocid = 0, mpi_real8 = 0, mpi_integer = 0, mpi_logical = 0, mpi_character = 0
end module spmd_utils

module ppgrid
  integer, parameter :: pcols = PCOLS
  integer, parameter :: pver  = PVER
  integer, parameter :: pverp = PVER + 1
  integer :: begchunk = 1, endchunk = 1
end module ppgrid

module cam_logfile
  integer :: iulog = 0   ! stderr: keeps the zm_convi banner off stdout
end module cam_logfile

module cam_abortutils
  integer :: endrun_count = 0
contains
  subroutine endrun(msg)
    character(len=*), intent(in), optional :: msg
    if (present(msg)) write(0,*) 'ENDRUN: ', trim(msg)
    endrun_count = endrun_count + 1
    stop 1
  end subroutine endrun
end module cam_abortutils

module physconst
  use shr_kind_mod, only: r8 => shr_kind_r8
  real(r8), parameter :: pi      = 3.14159265358979323846_r8
  real(r8), parameter :: boltz   = 1.38065e-23_r8
  real(r8), parameter :: avogad  = 6.02214e26_r8
  real(r8), parameter :: rgas_u  = avogad*boltz
  real(r8), parameter :: mwdair  = 28.966_r8
  real(r8), parameter :: mwwv    = 18.016_r8
  real(r8), parameter :: rair    = rgas_u/mwdair
  real(r8), parameter :: rh2o    = rgas_u/mwwv
  real(r8), parameter :: cpair   = 1.00464e3_r8
  real(r8), parameter :: cpwv    = 1.810e3_r8
  real(r8), parameter :: cpliq   = 4.188e3_r8
  real(r8), parameter :: gravit  = 9.80616_r8
  real(r8), parameter :: latvap  = 2.501e6_r8
  real(r8), parameter :: latice  = 3.337e5_r8
  real(r8), parameter :: tmelt   = 273.15_r8
  real(r8), parameter :: epsilo  = mwwv/mwdair
  real(r8), parameter :: zvir    = rh2o/rair - 1._r8
  real(r8), parameter :: cpvir   = cpwv/cpair - 1._r8
  real(r8), parameter :: rhoh2o  = 1.000e3_r8
  real(r8), parameter :: rga     = 1._r8/gravit
end module physconst

module cam_history
  use shr_kind_mod, only: r8 => shr_kind_r8
  use ppgrid,       only: pcols, pverp
  integer, parameter :: horiz_only = 1
  ! History output is a no-op in the oracle build, except that the test driver may switch on a capture of what the
  ! reference hands to outfld (oracle/ref_wrap_aero.F90 reads modal_aero_sw's diagnostics back this way): the fields of
  ! the CURRENT chunk, by name, last call wins.
  integer, parameter :: cap_max = 128
  logical            :: cap_on = .false.
  integer            :: cap_n = 0
  character(len=32)  :: cap_name(cap_max)
  integer            :: cap_nlev(cap_max)
  real(r8)           :: cap_val(pcols, pverp, cap_max)
  interface addfld
    module procedure addfld_1d, addfld_nd
  end interface addfld
contains
  subroutine outfld(name, field, idim, c)
    character(len=*), intent(in) :: name
    real(r8), intent(in) :: field(..)
    integer, intent(in) :: idim, c
    integer :: s, j
    if (.not. cap_on) return
    s = 0
    do j = 1, cap_n
      if (trim(cap_name(j)) == trim(name)) s = j
    end do
    if (s == 0) then
      if (cap_n >= cap_max) return
      cap_n = cap_n + 1; s = cap_n; cap_name(s) = name
    end if
    select rank (field)
    rank (1)
      cap_nlev(s) = 1
      cap_val(1:size(field), 1, s) = field
    rank (2)
      cap_nlev(s) = size(field, 2)
      cap_val(1:size(field, 1), 1:size(field, 2), s) = field
    end select
  end subroutine outfld
  ! copies the captured field `name` of the current chunk to out(1:n, 1:nlev); returns its level count, 0 if absent
  integer function outfld_captured(name, n, out)
    character(len=*), intent(in) :: name
    integer, intent(in) :: n
    real(r8), intent(out) :: out(n, pverp)
    integer :: j
    outfld_captured = 0
    out = 0._r8
    do j = 1, cap_n
      if (trim(cap_name(j)) == trim(name)) then
        outfld_captured = cap_nlev(j)
        out(1:n, 1:cap_nlev(j)) = cap_val(1:n, 1:cap_nlev(j), j)
      end if
    end do
  end function outfld_captured
  subroutine addfld_1d(fname, vdim_name, avgflag, units, long_name, flag_xyfill, sampling_seq, standard_name, fill_value)
    character(len=*), intent(in) :: fname, avgflag, units, long_name
    integer, intent(in) :: vdim_name
    logical, intent(in), optional :: flag_xyfill
    character(len=*), intent(in), optional :: sampling_seq, standard_name
    real(r8), intent(in), optional :: fill_value
  end subroutine addfld_1d
  subroutine addfld_nd(fname, dimnames, avgflag, units, long_name, flag_xyfill, sampling_seq, standard_name, fill_value)
    character(len=*), intent(in) :: fname, avgflag, units, long_name
    character(len=*), intent(in) :: dimnames(:)
    logical, intent(in), optional :: flag_xyfill
    character(len=*), intent(in), optional :: sampling_seq, standard_name
    real(r8), intent(in), optional :: fill_value
  end subroutine addfld_nd
  subroutine add_default(name, tindex, flag)
    character(len=*), intent(in) :: name, flag
    integer, intent(in) :: tindex
  end subroutine add_default
end module cam_history

module constituents
  use shr_kind_mod, only: r8 => shr_kind_r8
  integer, parameter :: pcnst = 40
  logical :: cnst_is_dry(pcnst) = .false.      ! set by the test driver (oracle/ref_wrap_zm.F90)
  ! what the operator layer (physics_types.F90, zm_conv_intr.F90) reads; the test driver (oracle/ref_wrap_intr.F90) sets them
  real(r8)          :: qmin(pcnst) = 0._r8
  character(len=16) :: cnst_name(pcnst) = ' '
  character(len=3)  :: cnst_type(pcnst) = 'wet'
  logical :: cnst_is_convtran1(pcnst) = .false., cnst_is_convtran2(pcnst) = .false.
contains
  function cnst_get_type_byind(m) result(t)
    integer, intent(in) :: m
    character(len=3) :: t
    t = 'wet'
    if (cnst_is_dry(m)) t = 'dry'
  end function cnst_get_type_byind
  subroutine cnst_get_ind(name, ind, abort)
    character(len=*), intent(in) :: name
    integer, intent(out) :: ind
    logical, intent(in), optional :: abort
    integer :: m
    ind = -1
    do m = 1, pcnst
      if (trim(cnst_name(m)) == trim(name)) ind = m
    end do
  end subroutine cnst_get_ind
  subroutine cnst_add(name, mwc, cpc, qminc, ind, longname, readiv, mixtype, cam_outfld)
    character(len=*), intent(in) :: name
    real(r8), intent(in) :: mwc, cpc, qminc
    integer, intent(out) :: ind
    character(len=*), intent(in), optional :: longname, mixtype
    logical, intent(in), optional :: readiv, cam_outfld
    ind = -1
  end subroutine cnst_add
end module constituents

module phys_grid
  use shr_kind_mod, only: r8 => shr_kind_r8
  use ppgrid,       only: pcols
  integer :: ncols_of_chunk = pcols              ! the test driver sets the chunk's column count before physics_state_alloc
contains
  subroutine get_rlat_all_p(lchnk, n, rlat)
    integer, intent(in) :: lchnk, n
    real(r8), intent(out) :: rlat(n)
    rlat = 0._r8
  end subroutine get_rlat_all_p
  subroutine get_rlon_all_p(lchnk, n, rlon)
    integer, intent(in) :: lchnk, n
    real(r8), intent(out) :: rlon(n)
    rlon = 0._r8
  end subroutine get_rlon_all_p
  subroutine get_gcol_all_p(lchnk, n, gcols)
    integer, intent(in) :: lchnk, n
    integer, intent(out) :: gcols(n)
    integer :: i
    gcols = (/ (i, i = 1, n) /)
  end subroutine get_gcol_all_p
  integer function get_ncols_p(lchnk)
    integer, intent(in) :: lchnk
    get_ncols_p = ncols_of_chunk
  end function get_ncols_p
  real(r8) function get_rlat_p(lchnk, icol)
    integer, intent(in) :: lchnk, icol
    get_rlat_p = 0._r8
  end function get_rlat_p
  real(r8) function get_rlon_p(lchnk, icol)
    integer, intent(in) :: lchnk, icol
    get_rlon_p = 0._r8
  end function get_rlon_p
end module phys_grid

module phys_control
  ! Only the entry points the hot path and its operator layer touch (zm_conv.F90:263,871; zm_conv_intr.F90:268,318,375;
  ! physics_types.F90:16,283; convect_diagnostics.F90:57).  The reference's own phys_control.F90 needs four more absent
  ! modules plus mpi_bcast (SURVEY.md Appendix A).
  logical :: deepconv_pbl_off = .false.          ! phys_deepconv_pbl(): "no deep convection in the PBL"; set by the test driver
contains
  logical function cam_physpkg_is(name)
    character(len=*), intent(in) :: name
    cam_physpkg_is = (trim(name) == 'cam_dev')
  end function cam_physpkg_is
  logical function waccmx_is(name)
    character(len=*), intent(in) :: name
    waccmx_is = .false.
  end function waccmx_is
  logical function phys_deepconv_pbl()
    phys_deepconv_pbl = deepconv_pbl_off
  end function phys_deepconv_pbl
  subroutine phys_getopts(history_amwg_out, history_aero_optics_out, history_dust_out, history_budget_out, &
                          history_budget_histfile_num_out, state_debug_checks_out, shallow_scheme_out, dme_energy_adjust_out)
    logical, intent(out), optional :: history_amwg_out, history_aero_optics_out, history_dust_out, history_budget_out
    logical, intent(out), optional :: state_debug_checks_out, dme_energy_adjust_out
    integer, intent(out), optional :: history_budget_histfile_num_out
    character(len=16), intent(out), optional :: shallow_scheme_out
    if (present(history_amwg_out)) history_amwg_out = .false.
    if (present(history_aero_optics_out)) history_aero_optics_out = .false.
    if (present(history_dust_out)) history_dust_out = .false.
    if (present(history_budget_out)) history_budget_out = .false.
    if (present(history_budget_histfile_num_out)) history_budget_histfile_num_out = 1
    if (present(state_debug_checks_out)) state_debug_checks_out = .false.      ! CAM's default (phys_control.F90)
    if (present(shallow_scheme_out)) shallow_scheme_out = 'CLUBB_SGS'
    if (present(dme_energy_adjust_out)) dme_energy_adjust_out = .false.
  end subroutine phys_getopts
end module phys_control

module dycore
  ! run-time switchable so one oracle build covers all three geopotential_t branches
  ! (geopotential.F90:217,222): 'EUL', 'LR', 'SE'
  character(len=8) :: dycore_name = 'LR'
contains
  logical function dycore_is(name)
    character(len=*), intent(in) :: name
    dycore_is = (trim(name) == trim(dycore_name))
  end function dycore_is
end module dycore

module air_composition
  ! water vapour only unless the test driver widens it (geopotential.F90:256-274)
  use shr_kind_mod, only: r8 => shr_kind_r8
  integer :: thermodynamic_active_species_num = 1
  integer :: thermodynamic_active_species_idx(8) = (/1, 2, 3, 4, 5, 6, 7, 8/)
  integer :: dry_air_species_num = 0
  ! composition-dependent cp and R of dry air per chunk (physics_types.F90:12,393-409); allocated and filled by the test driver
  real(r8), allocatable :: cpairv(:,:,:), rairv(:,:,:)
end module air_composition

module zm_microphysics
  ! Types and the 70-argument signature only (zm_conv.F90:23,566-619,3782-3793).  zmconv_microp=.true.
  ! is out of scope (SURVEY.md 8c): the body aborts.
  use shr_kind_mod, only: r8 => shr_kind_r8
  type ptr2d
    real(r8), pointer :: val(:,:)
  end type ptr2d
  type zm_aero_t
    character(len=5) :: scheme = 'none'
    integer :: nmodes = 0, nbulk = 0
    integer, allocatable :: nspec(:)
    type(ptr2d), allocatable :: num_a(:), dgnum(:), mmr_bulk(:)
    type(ptr2d), allocatable :: mmr_a(:,:)
    real(r8), allocatable :: numg_a(:,:,:), dgnumg(:,:,:), mmrg_bulk(:,:,:)
    real(r8), allocatable :: mmrg_a(:,:,:,:)
    ! components zm_conv_intr.F90:1155-1290 (zm_aero_init, only reached under zmconv_microp) names
    integer :: mode_accum_idx = -1, mode_aitken_idx = -1, mode_coarse_idx = -1, coarse_dust_idx = -1, coarse_nacl_idx = -1
    integer :: idxsul = -1, idxdst1 = -1, idxdst2 = -1, idxdst3 = -1, idxdst4 = -1, idxbcphi = -1
    real(r8) :: sigmag_aitken = 0._r8
    real(r8), allocatable :: voltonumblo(:), voltonumbhi(:), specdens(:,:), spechygro(:,:), num_to_mass_aer(:)
  end type zm_aero_t
  type zm_conv_t
    real(r8), allocatable, dimension(:,:) :: frz, sprd, wu, qi, qliq, qice, qrain, qsnow, di, dnl, dni, &
      qnl, qni, qnr, qns, qide, qncde, qnide, autolm, accrlm, bergnm, fhtimm, fhtctm, fhmlm, hmpim, accslm, &
      dlfm, cmel, autoln, accrln, bergnn, fhtimn, fhtctn, fhmln, accsln, activn, dlfn, autoim, accsim, difm, &
      cmei, nuclin, autoin, accsin, hmpin, difn, trspcm, trspcn, trspim, trspin, lambdadpcu, mudpcu
    real(r8), allocatable :: dcape(:)
  end type zm_conv_t
contains
  subroutine zm_mphy(su, qu, mu, du, eu, cmel, cmei, zf, p, t, q, eps0, jb, jt, jlcl, msg, il2g, grav, cp, rd, aero, gamhat, &
       qliq, qice, qnl, qni, qcde, qide, qncde, qnide, rprd, sprd, frz, wu, qrain, qsnow, qnr, qns, &
       autolm, accrlm, bergnm, fhtimm, fhtctm, fhmlm, hmpim, accslm, dlfm, autoln, accrln, bergnn, fhtimn, fhtctn, &
       fhmln, accsln, activn, dlfn, autoim, accsim, difm, nuclin, autoin, accsin, hmpin, difn, trspcm, trspcn, &
       trspim, trspin, lambdadpcu, mudpcu)
    real(r8), dimension(:,:) :: su, qu, mu, du, eu, cmel, cmei, zf, p, t, q, gamhat, &
       qliq, qice, qnl, qni, qcde, qide, qncde, qnide, rprd, sprd, frz, wu, qrain, qsnow, qnr, qns, &
       autolm, accrlm, bergnm, fhtimm, fhtctm, fhmlm, hmpim, accslm, dlfm, autoln, accrln, bergnn, fhtimn, fhtctn, &
       fhmln, accsln, activn, dlfn, autoim, accsim, difm, nuclin, autoin, accsin, hmpin, difn, trspcm, trspcn, &
       trspim, trspin, lambdadpcu, mudpcu
    real(r8) :: eps0(:), grav, cp, rd
    integer :: jb(:), jt(:), jlcl(:), msg, il2g
    type(zm_aero_t) :: aero
    stop 'zm_mphy: convective microphysics (zmconv_microp) is outside the hot-path scope'
  end subroutine zm_mphy
  subroutine zm_mphyi()        ! zm_conv_intr.F90:1129 (zm_conv_micro_init, only under zmconv_microp)
    stop 'zm_mphyi: convective microphysics (zmconv_microp) is outside the hot-path scope'
  end subroutine zm_mphyi
end module zm_microphysics
