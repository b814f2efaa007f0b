! zm_conv_hip -- ISO_C_BINDING shim: the reference's ZM operator surface, bodies on the MI355X.
!
! Public procedures have the SAME names, argument lists, kinds and intents as the reference module zm_conv
! (NorESMhub/CAM-Nor-physics physics/zm_conv.F90):
!     zm_convi      :115-120      zm_convr   :231-244 (intents :362-423)
!     zm_conv_evap  :1712-1717    momtran    :2315-2319      convtran   :1976-1981
! so zm_conv_intr.F90 (`use zm_conv, only: zm_conv_evap, zm_convr, convtran, momtran`, :13; calls at :376,
! :662, :764, :822) compiles against it unchanged.  Each body marshals the caller's (pcols,pver) arrays to
! the C ABI's host-layout entry points (include/colphys.h, *_chunks), which pack exactly `ncol` columns, run
! the HIP kernels and copy the results back.  physics_state / physics_ptend / pbuf stay what CAM expects.
!
! Module name: `zm_conv_hip` by default (delegation: see INTEGRATION.md); compiled with
! -DCOLPHYS_DROPIN_NAMES it is `zm_conv`, a whole-module replacement -- that is how tests/ builds it against
! the very wrapper source that drives the reference (oracle/ref_wrap_zm.F90).
!
! convtran (zm_conv.F90:1976, SURVEY.md 8f-1) is offloaded too (non-microphysics path).
#ifdef COLPHYS_DROPIN_NAMES
#define ZM_MODULE zm_conv
#else
#define ZM_MODULE zm_conv_hip
#endif

module ZM_MODULE
  use iso_c_binding
  use shr_kind_mod,    only: r8 => shr_kind_r8
  use spmd_utils,      only: masterproc
  use ppgrid,          only: pcols, pver, pverp
  use physconst,       only: cpair, epsilo, gravit, latice, latvap, tmelt, rair, cpwv, cpliq, rh2o, cpvir, zvir
  use cam_abortutils,  only: endrun
  use cam_logfile,     only: iulog
  use zm_microphysics, only: zm_aero_t, zm_conv_t
  use colphys_c_api
  implicit none
  save
  private

  public zm_convi, zm_convr, zm_conv_evap, momtran, convtran

  logical :: zmconv_microp = .false.
  logical :: zm_org = .false.

contains

subroutine zm_convi(limcnv_in, zmconv_c0_lnd, zmconv_c0_ocn, zmconv_ke, zmconv_ke_lnd, &
                    zmconv_momcu, zmconv_momcd, zmconv_num_cin, zmconv_org, &
                    zmconv_microp_in, no_deep_pbl_in, zmconv_tiedke_add, &
                    zmconv_capelmt, zmconv_dmpdz, zmconv_parcel_pbl, zmconv_tau)
   use phys_control, only: cam_physpkg_is
   integer, intent(in)  :: limcnv_in
   integer, intent(in)  :: zmconv_num_cin
   real(r8),intent(in)  :: zmconv_c0_lnd
   real(r8),intent(in)  :: zmconv_c0_ocn
   real(r8),intent(in)  :: zmconv_ke
   real(r8),intent(in)  :: zmconv_ke_lnd
   real(r8),intent(in)  :: zmconv_momcu
   real(r8),intent(in)  :: zmconv_momcd
   logical              :: zmconv_org
   logical, intent(in)  :: zmconv_microp_in
   logical, intent(in)  :: no_deep_pbl_in
   real(r8),intent(in)  :: zmconv_tiedke_add
   real(r8),intent(in)  :: zmconv_capelmt
   real(r8),intent(in)  :: zmconv_dmpdz
   logical, intent(in)  :: zmconv_parcel_pbl
   real(r8),intent(in)  :: zmconv_tau

   type(colphys_zm_params) :: p
   integer(c_int) :: rc, device, ndev
   character(len=32) :: env
   integer :: stat

   if (zmconv_num_cin > 5) call endrun('**** ZM_CONVI : NUM_CIN must not exceeed 5 ****')   ! zm_conv.F90:200

   ! physconst values come from the host model, never from the kernels
   p%cpair = cpair; p%epsilo = epsilo; p%gravit = gravit; p%latice = latice; p%latvap = latvap; p%tmelt = tmelt
   p%rair = rair; p%cpwv = cpwv; p%cpliq = cpliq; p%rh2o = rh2o; p%cpvir = cpvir; p%zvir = zvir
   p%c0_lnd = zmconv_c0_lnd; p%c0_ocn = zmconv_c0_ocn; p%ke = zmconv_ke; p%ke_lnd = zmconv_ke_lnd
   p%momcu = zmconv_momcu; p%momcd = zmconv_momcd; p%tiedke_add = zmconv_tiedke_add
   p%capelmt = zmconv_capelmt; p%dmpdz_param = zmconv_dmpdz; p%tau = zmconv_tau
   ! module-level tunables of the reference and their defaults (zm_conv.F90:83-97,103,213)
   p%tiedke_lnd = 1.0_r8; p%entrmn = 2e-4_r8; p%alfadet = 0.1_r8; p%tentrm = -zmconv_dmpdz
   p%plclmin = 6.e2_r8; p%cin_threshd = 0.33_r8; p%parcel_hscale = 0.5_r8
   p%limcnv = limcnv_in; p%num_cin = zmconv_num_cin
   p%zm_org = merge(1, 0, zmconv_org); p%zmconv_microp = merge(1, 0, zmconv_microp_in)
   p%no_deep_pbl = merge(1, 0, no_deep_pbl_in); p%lparcel_pbl = merge(1, 0, zmconv_parcel_pbl)
   p%cam3 = merge(1, 0, cam_physpkg_is('cam3')); p%reserved = 0
   zmconv_microp = zmconv_microp_in
   zm_org = zmconv_org

   ! one process per GPU: COLPHYS_DEVICE selects it (a launcher sets it to the node-local MPI rank)
   device = 0
   call get_environment_variable('COLPHYS_DEVICE', env, status=stat)
   if (stat == 0) read(env, *, iostat=stat) device
   ndev = colphys_device_count()
   if (ndev > 0) device = mod(device, ndev)

   rc = colphys_init(device, p)
   if (rc /= COLPHYS_OK) call endrun('**** ZM_CONVI (colphys): '//trim(colphys_error_message())//' ****')
   if (masterproc) write(iulog,*) '**** ZM: DILUTE Buoyancy Calculation on HIP device ', device, ' ****'
end subroutine zm_convi


subroutine zm_convr(lchnk   ,ncol    , &
                    t       ,qh      ,prec    ,jctop   ,jcbot   , &
                    pblh    ,zm      ,geos    ,zi      ,qtnd    , &
                    heat    ,pap     ,paph    ,dpp     , &
                    delt    ,mcon    ,cme     ,cape    ,eurt    , &
                    tpert   ,dlf     ,pflx    ,zdu     ,rprd    , &
                    mu      ,md      ,du      ,eu      ,ed      , &
                    dp      ,dsubcld ,jt      ,maxg    ,ideep   , lengath , &
                    ql      ,rliq    ,landfrac,                   &
                    org     ,orgt    ,org2d   ,  &
                    dif     ,dnlf    ,dnif    ,conv    , &
                    aero    , rice)
   integer, intent(in) :: lchnk
   integer, intent(in) :: ncol
   real(r8), intent(in), target :: t(pcols,pver)
   real(r8), intent(in), target :: qh(pcols,pver)
   real(r8), intent(in), target :: pap(pcols,pver)
   real(r8), intent(in), target :: paph(pcols,pver+1)
   real(r8), intent(in), target :: dpp(pcols,pver)
   real(r8), intent(in), target :: zm(pcols,pver)
   real(r8), intent(in), target :: geos(pcols)
   real(r8), intent(in), target :: zi(pcols,pver+1)
   real(r8), intent(in), target :: pblh(pcols)
   real(r8), intent(in), target :: tpert(pcols)
   real(r8), intent(in), target :: landfrac(pcols)
   type(zm_conv_t), intent(inout) :: conv
   type(zm_aero_t), intent(inout) :: aero
   real(r8), intent(out), target :: qtnd(pcols,pver)
   real(r8), intent(out), target :: heat(pcols,pver)
   real(r8), intent(out), target :: mcon(pcols,pverp)
   real(r8), intent(out), target :: dlf(pcols,pver)
   real(r8), intent(out), target :: pflx(pcols,pverp)
   real(r8), intent(out), target :: cme(pcols,pver)
   real(r8), intent(out), target :: cape(pcols)
   real(r8), intent(out), target :: zdu(pcols,pver)
   real(r8), intent(out), target :: rprd(pcols,pver)
   real(r8), intent(out), target :: dif(pcols,pver)
   real(r8), intent(out), target :: dnlf(pcols,pver)
   real(r8), intent(out), target :: dnif(pcols,pver)
   real(r8), intent(out), target :: mu(pcols,pver)
   real(r8), intent(out), target :: eu(pcols,pver)
   real(r8), intent(out), target :: eurt(pcols,pver)
   real(r8), intent(out), target :: du(pcols,pver)
   real(r8), intent(out), target :: md(pcols,pver)
   real(r8), intent(out), target :: ed(pcols,pver)
   real(r8), intent(out), target :: dp(pcols,pver)
   real(r8), intent(out), target :: dsubcld(pcols)
   real(r8), intent(out), target :: jctop(pcols)
   real(r8), intent(out), target :: jcbot(pcols)
   real(r8), intent(out), target :: prec(pcols)
   real(r8), intent(out), target :: rliq(pcols)
   real(r8), intent(out), target :: rice(pcols)
   integer,  intent(out), target :: ideep(pcols)
   integer,  intent(out) :: lengath
   real(r8), pointer :: org(:,:)
   real(r8), pointer :: orgt(:,:)
   real(r8), pointer :: org2d(:,:)
   integer, target :: jt(pcols)
   integer, target :: maxg(pcols)
   real(r8), target :: ql(pcols,pver)
   real(r8) delt

   type(colphys_zm_convr_io) :: io
   integer(c_int32_t), target :: ncol_c(1), lengath_c(1), status_c(pcols)
   integer(c_int) :: rc
   integer :: i

   ! whole-array initialisations of the reference (zm_conv.F90:559-561); the device path defines 1:ncol
   qtnd(:,:) = 0._r8
   heat(:,:) = 0._r8
   mcon(:,:) = 0._r8

   io%t = c_loc(t); io%qh = c_loc(qh); io%pap = c_loc(pap); io%dpp = c_loc(dpp); io%zm = c_loc(zm)
   io%paph = c_loc(paph); io%zi = c_loc(zi); io%geos = c_loc(geos); io%pblh = c_loc(pblh)
   io%tpert = c_loc(tpert); io%landfrac = c_loc(landfrac)
   io%qtnd = c_loc(qtnd); io%heat = c_loc(heat); io%cme = c_loc(cme); io%zdu = c_loc(zdu); io%rprd = c_loc(rprd)
   io%dlf = c_loc(dlf); io%ql = c_loc(ql); io%dif = c_loc(dif); io%dnlf = c_loc(dnlf); io%dnif = c_loc(dnif)
   io%eurt = c_loc(eurt); io%mcon = c_loc(mcon); io%pflx = c_loc(pflx)
   io%prec = c_loc(prec); io%jctop = c_loc(jctop); io%jcbot = c_loc(jcbot); io%cape = c_loc(cape)
   io%rliq = c_loc(rliq); io%rice = c_loc(rice)
   io%mu = c_loc(mu); io%md = c_loc(md); io%du = c_loc(du); io%eu = c_loc(eu); io%ed = c_loc(ed); io%dp = c_loc(dp)
   io%dsubcld = c_loc(dsubcld); io%jt = c_loc(jt); io%maxg = c_loc(maxg); io%ideep = c_loc(ideep)
   io%lengath = c_loc(lengath_c); io%status = c_loc(status_c)
   ncol_c(1) = ncol
   status_c = 0
   ideep(:) = 0                                                         ! zm_conv.F90:906

   rc = colphys_zm_convr_chunks(1, pcols, pver, ncol_c, delt, io)
   if (rc /= COLPHYS_OK) call endrun('**** ZM_CONVR (colphys): '//trim(colphys_error_message())//' ****')
   lengath = lengath_c(1)
   do i = 1, ncol                                                       ! zm_conv.F90:5401-5410, 5557-5566
      if (iand(status_c(i), COLPHYS_STATUS_BRENT_NOT_CONVERGED) /= 0) then
         ! the inversions of a column run ienthalpy first (parcel_dilute's first loop, :5102, :5129), so that is the one the
         ! reference would have stopped in when both kinds failed
         if (iand(status_c(i), COLPHYS_STATUS_BRENT_ENTHALPY) /= 0) then
            write(iulog,*) '*** ZM_CONV: IENTHALPY: Failed and about to exit, lchnk,icol= ', lchnk, i
            call endrun('**** ZM_CONV IENTHALPY: Tmix did not converge ****')
         end if
         write(iulog,*) '*** ZM_CONV: IENTROPY: Failed and about to exit, lchnk,icol= ', lchnk, i
         call endrun('**** ZM_CONV IENTROPY: Tmix did not converge ****')
      end if
   end do
end subroutine zm_convr


subroutine zm_conv_evap(ncol,lchnk, &
     t,pmid,pdel,q, &
     landfrac, &
     tend_s, tend_s_snwprd, tend_s_snwevmlt, tend_q, &
     prdprec, cldfrc, deltat,  &
     prec, snow, ntprprd, ntsnprd, flxprec, flxsnow, prdsnow)
    integer,intent(in) :: ncol, lchnk
    real(r8),intent(in), dimension(pcols,pver), target :: t
    real(r8),intent(in), dimension(pcols,pver), target :: pmid
    real(r8),intent(in), dimension(pcols,pver), target :: pdel
    real(r8),intent(in), dimension(pcols,pver), target :: q
    real(r8),intent(in), dimension(pcols), target :: landfrac
    real(r8),intent(inout), dimension(pcols,pver), target :: tend_s
    real(r8),intent(inout), dimension(pcols,pver), target :: tend_q
    real(r8),intent(out  ), dimension(pcols,pver), target :: tend_s_snwprd
    real(r8),intent(out  ), dimension(pcols,pver), target :: tend_s_snwevmlt
    real(r8), intent(in   ), target :: prdprec(pcols,pver)
    real(r8), intent(in   ), target :: cldfrc(pcols,pver)
    real(r8), intent(in   ) :: deltat
    real(r8), intent(inout), target :: prec(pcols)
    real(r8), intent(out), target   :: snow(pcols)
    real(r8), optional, intent(in), allocatable  :: prdsnow(:,:)
    real(r8),intent(out), target :: flxprec(pcols,pverp)
    real(r8),intent(out), target :: flxsnow(pcols,pverp)
    real(r8),intent(out), target :: ntprprd(pcols,pver)
    real(r8),intent(out), target :: ntsnprd(pcols,pver)

    type(colphys_zm_conv_evap_io) :: io
    integer(c_int32_t) :: ncol_c(1)
    integer(c_int) :: rc

    if (present(prdsnow)) then                                          ! zm_conv.F90:1789-1794
       if (allocated(prdsnow)) call endrun('zm_conv_evap (colphys): prdsnow / convective microphysics is outside the offloaded path')
    end if
    io%t = c_loc(t); io%pmid = c_loc(pmid); io%pdel = c_loc(pdel); io%q = c_loc(q); io%prdprec = c_loc(prdprec)
    io%cldfrc = c_loc(cldfrc); io%landfrac = c_loc(landfrac); io%prec = c_loc(prec)
    io%tend_s = c_loc(tend_s); io%tend_s_snwprd = c_loc(tend_s_snwprd); io%tend_s_snwevmlt = c_loc(tend_s_snwevmlt)
    io%tend_q = c_loc(tend_q); io%ntprprd = c_loc(ntprprd); io%ntsnprd = c_loc(ntsnprd)
    io%flxprec = c_loc(flxprec); io%flxsnow = c_loc(flxsnow); io%snow = c_loc(snow)
    ncol_c(1) = ncol
    rc = colphys_zm_conv_evap_chunks(1, pcols, pver, ncol_c, deltat, io)
    if (rc /= COLPHYS_OK) call endrun('**** ZM_CONV_EVAP (colphys): '//trim(colphys_error_message())//' ****')
end subroutine zm_conv_evap


subroutine momtran(lchnk, ncol, &
                    domomtran,q       ,ncnst   ,mu      ,md    , &
                    du      ,eu      ,ed      ,dp      ,dsubcld , &
                    jt      ,mx      ,ideep   ,il1g    ,il2g    , &
                    nstep   ,dqdt    ,pguall     ,pgdall, icwu, icwd, dt, seten    )
   integer, intent(in) :: lchnk
   integer, intent(in) :: ncol
   integer, intent(in) :: ncnst
   logical, intent(in) :: domomtran(ncnst)
   real(r8), intent(in), target :: q(pcols,pver,ncnst)
   real(r8), intent(in), target :: mu(pcols,pver)
   real(r8), intent(in), target :: md(pcols,pver)
   real(r8), intent(in), target :: du(pcols,pver)
   real(r8), intent(in), target :: eu(pcols,pver)
   real(r8), intent(in), target :: ed(pcols,pver)
   real(r8), intent(in), target :: dp(pcols,pver)
   real(r8), intent(in), target :: dsubcld(pcols)
   real(r8), intent(in) :: dt
   integer, intent(in), target :: jt(pcols)
   integer, intent(in), target :: mx(pcols)
   integer, intent(in), target :: ideep(pcols)
   integer, intent(in) :: il1g
   integer, intent(in) :: il2g
   integer, intent(in) :: nstep
   real(r8), intent(out), target :: dqdt(pcols,pver,ncnst)
   real(r8),intent(out), target ::  pguall(pcols,pver,ncnst)
   real(r8),intent(out), target ::  pgdall(pcols,pver,ncnst)
   real(r8),intent(out), target ::  icwu(pcols,pver,ncnst)
   real(r8),intent(out), target ::  icwd(pcols,pver,ncnst)
   real(r8),intent(out), target ::  seten(pcols,pver)

   type(colphys_momtran_io) :: io
   integer(c_int32_t), target :: ncol_c(1), lengath_c(1)
   integer(c_int) :: rc

   if (ncnst /= 2 .or. il1g /= 1) call endrun('momtran (colphys): expects the two wind components and il1g=1 (zm_conv_intr.F90:822-826)')
   if (.not. all(domomtran)) call endrun('momtran (colphys): domomtran must be .true. for both components')

   ! whole-array initialisations of the reference (zm_conv.F90:2429-2443, 2630)
   pguall(:,:,:) = 0.0_r8
   pgdall(:,:,:) = 0.0_r8
   dqdt(:,:,:)   = 0.0_r8
   seten(:,:)    = 0.0_r8

   io%winds = c_loc(q); io%mu = c_loc(mu); io%md = c_loc(md); io%du = c_loc(du); io%eu = c_loc(eu); io%ed = c_loc(ed)
   io%dp = c_loc(dp); io%dsubcld = c_loc(dsubcld); io%jt = c_loc(jt); io%mx = c_loc(mx); io%ideep = c_loc(ideep)
   io%lengath = c_loc(lengath_c)
   io%dqdt = c_loc(dqdt); io%pguall = c_loc(pguall); io%pgdall = c_loc(pgdall); io%icwu = c_loc(icwu); io%icwd = c_loc(icwd)
   io%seten = c_loc(seten)
   ncol_c(1) = ncol
   lengath_c(1) = il2g
   rc = colphys_momtran_chunks(1, pcols, pver, ncol_c, dt, io)
   if (rc /= COLPHYS_OK) call endrun('**** MOMTRAN (colphys): '//trim(colphys_error_message())//' ****')
end subroutine momtran


subroutine convtran(lchnk   , &
                    doconvtran,q       ,ncnst   ,mu      ,md      , &
                    du      ,eu      ,ed      ,dp      ,dsubcld , &
                    jt      ,mx      ,ideep   ,il1g    ,il2g    , &
                    nstep   ,fracis  ,dqdt    ,dpdry   ,dt)
   ! Same interface as zm_conv.F90:1976-1981 (SURVEY.md 8f-1, the first "next" row).
   use constituents, only: cnst_get_type_byind
   integer, intent(in) :: lchnk
   integer, intent(in) :: ncnst
   logical, intent(in) :: doconvtran(ncnst)
   real(r8), intent(in), target :: q(pcols,pver,ncnst)
   real(r8), intent(in), target :: mu(pcols,pver)
   real(r8), intent(in), target :: md(pcols,pver)
   real(r8), intent(in), target :: du(pcols,pver)
   real(r8), intent(in), target :: eu(pcols,pver)
   real(r8), intent(in), target :: ed(pcols,pver)
   real(r8), intent(in), target :: dp(pcols,pver)
   real(r8), intent(in) :: dsubcld(pcols)
   real(r8), intent(in), target :: fracis(pcols,pver,ncnst)
   integer, intent(in), target :: jt(pcols)
   integer, intent(in), target :: mx(pcols)
   integer, intent(in), target :: ideep(pcols)
   integer, intent(in) :: il1g
   integer, intent(in) :: il2g
   integer, intent(in) :: nstep
   real(r8), intent(in), target :: dpdry(pcols,pver)
   real(r8), intent(in) :: dt
   real(r8), intent(out), target :: dqdt(pcols,pver,ncnst)

   type(colphys_convtran_io) :: io
   integer(c_int32_t), target :: ncol_c(1), lengath_c(1), do_c(ncnst), dry_c(ncnst)
   integer(c_int) :: rc
   integer :: m

   if (zmconv_microp) call endrun('convtran (colphys): convective microphysics is outside the offloaded path')
   if (il1g /= 1) call endrun('convtran (colphys): expects il1g=1 (zm_conv_intr.F90:875-879, 1020-1024)')
   do m = 1, ncnst
      do_c(m)  = merge(1, 0, doconvtran(m) .and. m >= 2)                 ! do m = 2, ncnst  (zm_conv.F90:2062)
      dry_c(m) = merge(1, 0, cnst_get_type_byind(m) .eq. 'dry')
      if (do_c(m) /= 0) dqdt(:,:,m) = 0._r8                              ! zm_conv.F90:2299
   end do
   io%q = c_loc(q); io%fracis = c_loc(fracis); io%mu = c_loc(mu); io%md = c_loc(md); io%du = c_loc(du)
   io%eu = c_loc(eu); io%ed = c_loc(ed); io%dp = c_loc(dp); io%dpdry = c_loc(dpdry)
   io%jt = c_loc(jt); io%mx = c_loc(mx); io%ideep = c_loc(ideep); io%lengath = c_loc(lengath_c)
   io%doconvtran = c_loc(do_c); io%isdry = c_loc(dry_c); io%dqdt = c_loc(dqdt)
   ! the gathered arrays are defined for 1:il2g only; ncol is not an argument of convtran: every column the
   ! gathered set points at lies within 1:pcols, so the chunk is handed over at full width
   ncol_c(1) = pcols
   lengath_c(1) = il2g
   rc = colphys_convtran_chunks(1, pcols, pver, ncol_c, ncnst, dt, io)
   if (rc /= COLPHYS_OK) call endrun('**** CONVTRAN (colphys): '//trim(colphys_error_message())//' ****')
end subroutine convtran

end module ZM_MODULE
