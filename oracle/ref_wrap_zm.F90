! TEST INFRASTRUCTURE -- not product code.
!
! bind(C) entry points around the UNMODIFIED reference modules zm_conv (physics/zm_conv.F90) and
! geopotential (physics/geopotential.F90), which oracle/Makefile compiles in place from /root/reference.
! The wrappers do what CAM's chunk loop does (physpkg.F90:1147-1161): cut a flat list of `n` columns into
! chunks of `pcols`, copy each chunk into (pcols,pver) arrays whose padding columns hold +inf exactly as
! physics_state_alloc leaves them (physics_types.F90:2143-2178), call the reference, copy results back.
!
! Interchange layout ("column-contiguous"): every profile array is a(pver,n) in Fortran order, i.e. a
! C / numpy array of shape [n][pver]; interface profiles are [n][pver+1].
! Gathered outputs keep the reference's per-chunk compaction: gathered slot j of chunk c is row
! c*pcols + j (0-based) of the flat array; ideep holds the 1-based in-chunk column index; lengath is per chunk.
! Slots beyond lengath are zero-filled by the wrapper before the call (the reference leaves them stale).

module ref_wrap_zm
  use iso_c_binding
  use shr_kind_mod,    only: r8 => shr_kind_r8
  use ppgrid,          only: pcols, pver, pverp
  use zm_conv,         only: zm_convi, zm_convr, zm_conv_evap, momtran, convtran
  use geopotential,    only: geopotential_t, geopotential_dse
  use zm_microphysics, only: zm_conv_t, zm_aero_t
  implicit none
  real(r8), parameter :: pad = huge(1._r8)
contains

  subroutine ref_dims(pc, pv) bind(C, name='ref_dims')
    integer(c_int), intent(out) :: pc, pv
    pc = pcols; pv = pver
  end subroutine ref_dims

  subroutine ref_set_dycore(kind) bind(C, name='ref_set_dycore')
    use dycore, only: dycore_name
    integer(c_int), value :: kind      ! 0 = EUL-like, 1 = LR/FV, 2 = SE/MPAS
    select case (kind)
    case (1);  dycore_name = 'LR'
    case (2);  dycore_name = 'SE'
    case default; dycore_name = 'EUL'
    end select
  end subroutine ref_set_dycore

  subroutine ref_set_thermo_species(num) bind(C, name='ref_set_thermo_species')
    use air_composition, only: thermodynamic_active_species_num
    integer(c_int), value :: num
    thermodynamic_active_species_num = num
  end subroutine ref_set_thermo_species

  subroutine ref_zm_convi(limcnv, c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, num_cin, org, microp, &
                          no_deep_pbl, tiedke_add, capelmt, dmpdz, parcel_pbl, tau) bind(C, name='ref_zm_convi')
    integer(c_int), value :: limcnv, num_cin, org, microp, no_deep_pbl, parcel_pbl
    real(c_double), value :: c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, tiedke_add, capelmt, dmpdz, tau
    call zm_convi(limcnv, c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, num_cin, org /= 0, microp /= 0, &
                  no_deep_pbl /= 0, tiedke_add, capelmt, dmpdz, parcel_pbl /= 0, tau)
  end subroutine ref_zm_convi

  ! ---- helpers ---------------------------------------------------------------------------------
  subroutine pack2(src, c0, nc, nlev, dst)
    integer,  intent(in)  :: c0, nc, nlev
    real(r8), intent(in)  :: src(nlev,*)
    real(r8), intent(out) :: dst(pcols,nlev)
    integer :: i, k
    dst = pad
    do i = 1, nc
      do k = 1, nlev
        dst(i,k) = src(k,c0+i)
      end do
    end do
  end subroutine pack2
  subroutine unpack2(src, c0, nc, nlev, dst)
    integer,  intent(in)    :: c0, nc, nlev
    real(r8), intent(in)    :: src(pcols,nlev)
    real(r8), intent(inout) :: dst(nlev,*)
    integer :: i, k
    do i = 1, nc
      do k = 1, nlev
        dst(k,c0+i) = src(i,k)
      end do
    end do
  end subroutine unpack2
  subroutine pack1(src, c0, nc, dst)
    integer,  intent(in)  :: c0, nc
    real(r8), intent(in)  :: src(*)
    real(r8), intent(out) :: dst(pcols)
    dst = pad
    dst(1:nc) = src(c0+1:c0+nc)
  end subroutine pack1

  ! ---- geopotential_t (geopotential.F90:153) -----------------------------------------------------
  subroutine ref_geopotential_t(n, nq, piln, pmln, pint, pmid, pdel, rpdel, t, q, rair, gravit, zvir, zi, zm) &
       bind(C, name='ref_geopotential_t')
    integer(c_int), value :: n, nq
    real(c_double), intent(in)  :: piln(pverp,n), pmln(pver,n), pint(pverp,n), pmid(pver,n), pdel(pver,n), rpdel(pver,n)
    real(c_double), intent(in)  :: t(pver,n), q(pver,n,nq), rair(pver,n), zvir(pver,n)
    real(c_double), value       :: gravit
    real(c_double), intent(out) :: zi(pverp,n), zm(pver,n)
    real(r8) :: piln_c(pcols,pverp), pmln_c(pcols,pver), pint_c(pcols,pverp), pmid_c(pcols,pver), pdel_c(pcols,pver), &
                rpdel_c(pcols,pver), t_c(pcols,pver), q_c(pcols,pver,nq), rair_c(pcols,pver), zvir_c(pcols,pver), &
                zi_c(pcols,pverp), zm_c(pcols,pver)
    integer :: c0, nc, m
    c0 = 0
    do while (c0 < n)
      nc = min(pcols, n - c0)
      call pack2(piln, c0, nc, pverp, piln_c); call pack2(pmln, c0, nc, pver, pmln_c)
      call pack2(pint, c0, nc, pverp, pint_c); call pack2(pmid, c0, nc, pver, pmid_c)
      call pack2(pdel, c0, nc, pver, pdel_c);  call pack2(rpdel, c0, nc, pver, rpdel_c)
      call pack2(t, c0, nc, pver, t_c)
      do m = 1, nq
        call pack2(q(:,:,m), c0, nc, pver, q_c(:,:,m))
      end do
      call pack2(rair, c0, nc, pver, rair_c);  call pack2(zvir, c0, nc, pver, zvir_c)
      call geopotential_t(piln_c, pmln_c, pint_c, pmid_c, pdel_c, rpdel_c, t_c, q_c, rair_c, gravit, zvir_c, &
                          zi_c, zm_c, nc)
      call unpack2(zi_c, c0, nc, pverp, zi); call unpack2(zm_c, c0, nc, pver, zm)
      c0 = c0 + nc
    end do
  end subroutine ref_geopotential_t

  ! ---- geopotential_dse (geopotential.F90:29) ----------------------------------------------------
  subroutine ref_geopotential_dse(n, nq, piln, pmln, pint, pmid, pdel, rpdel, dse, q, phis, rair, gravit, cpair, zvir, &
       t, zi, zm) bind(C, name='ref_geopotential_dse')
    integer(c_int), value :: n, nq
    real(c_double), intent(in)  :: piln(pverp,n), pmln(pver,n), pint(pverp,n), pmid(pver,n), pdel(pver,n), rpdel(pver,n)
    real(c_double), intent(in)  :: dse(pver,n), q(pver,n,nq), phis(n), rair(pver,n), cpair(pver,n), zvir(pver,n)
    real(c_double), value       :: gravit
    real(c_double), intent(out) :: t(pver,n), zi(pverp,n), zm(pver,n)
    real(r8) :: piln_c(pcols,pverp), pmln_c(pcols,pver), pint_c(pcols,pverp), pmid_c(pcols,pver), pdel_c(pcols,pver), &
                rpdel_c(pcols,pver), dse_c(pcols,pver), q_c(pcols,pver,nq), phis_c(pcols), rair_c(pcols,pver), &
                cpair_c(pcols,pver), zvir_c(pcols,pver), t_c(pcols,pver), zi_c(pcols,pverp), zm_c(pcols,pver)
    integer :: c0, nc, m
    c0 = 0
    do while (c0 < n)
      nc = min(pcols, n - c0)
      call pack2(piln, c0, nc, pverp, piln_c); call pack2(pmln, c0, nc, pver, pmln_c)
      call pack2(pint, c0, nc, pverp, pint_c); call pack2(pmid, c0, nc, pver, pmid_c)
      call pack2(pdel, c0, nc, pver, pdel_c);  call pack2(rpdel, c0, nc, pver, rpdel_c)
      call pack2(dse, c0, nc, pver, dse_c)
      do m = 1, nq
        call pack2(q(:,:,m), c0, nc, pver, q_c(:,:,m))
      end do
      call pack1(phis, c0, nc, phis_c)
      call pack2(rair, c0, nc, pver, rair_c);  call pack2(cpair, c0, nc, pver, cpair_c); call pack2(zvir, c0, nc, pver, zvir_c)
      call geopotential_dse(piln_c, pmln_c, pint_c, pmid_c, pdel_c, rpdel_c, dse_c, q_c, phis_c, rair_c, gravit, cpair_c, &
                            zvir_c, t_c, zi_c, zm_c, nc)
      call unpack2(t_c, c0, nc, pver, t); call unpack2(zi_c, c0, nc, pverp, zi); call unpack2(zm_c, c0, nc, pver, zm)
      c0 = c0 + nc
    end do
  end subroutine ref_geopotential_dse

  ! ---- zm_convr (zm_conv.F90:231) ----------------------------------------------------------------
  subroutine ref_zm_convr(n, t, qh, pblh, zm, geos, zi, pap, paph, dpp, delt, tpert, landfrac, &
       qtnd, heat, mcon, dlf, pflx, cme, cape, zdu, rprd, dif, dnlf, dnif, eurt, ql, &
       mu, md, du, eu, ed, dp, dsubcld, jctop, jcbot, prec, rliq, rice, jt, maxg, ideep, lengath) &
       bind(C, name='ref_zm_convr')
    integer(c_int), value :: n
    real(c_double), intent(in)  :: t(pver,n), qh(pver,n), zm(pver,n), zi(pverp,n), pap(pver,n), paph(pverp,n), dpp(pver,n)
    real(c_double), intent(in)  :: pblh(n), geos(n), tpert(n), landfrac(n)
    real(c_double), value       :: delt
    real(c_double), intent(out) :: qtnd(pver,n), heat(pver,n), mcon(pverp,n), dlf(pver,n), pflx(pverp,n), cme(pver,n), &
                                   zdu(pver,n), rprd(pver,n), dif(pver,n), dnlf(pver,n), dnif(pver,n), eurt(pver,n), &
                                   ql(pver,n), mu(pver,n), md(pver,n), du(pver,n), eu(pver,n), ed(pver,n), dp(pver,n)
    real(c_double), intent(out) :: cape(n), dsubcld(n), jctop(n), jcbot(n), prec(n), rliq(n), rice(n)
    integer(c_int), intent(out) :: jt(n), maxg(n), ideep(n), lengath(*)

    real(r8) :: t_c(pcols,pver), qh_c(pcols,pver), zm_c(pcols,pver), zi_c(pcols,pverp), pap_c(pcols,pver), &
                paph_c(pcols,pverp), dpp_c(pcols,pver), pblh_c(pcols), geos_c(pcols), tpert_c(pcols), landfrac_c(pcols)
    real(r8) :: qtnd_c(pcols,pver), heat_c(pcols,pver), mcon_c(pcols,pverp), dlf_c(pcols,pver), pflx_c(pcols,pverp), &
                cme_c(pcols,pver), zdu_c(pcols,pver), rprd_c(pcols,pver), dif_c(pcols,pver), dnlf_c(pcols,pver), &
                dnif_c(pcols,pver), eurt_c(pcols,pver), ql_c(pcols,pver), mu_c(pcols,pver), md_c(pcols,pver), &
                du_c(pcols,pver), eu_c(pcols,pver), ed_c(pcols,pver), dp_c(pcols,pver)
    real(r8) :: cape_c(pcols), dsubcld_c(pcols), jctop_c(pcols), jcbot_c(pcols), prec_c(pcols), rliq_c(pcols), rice_c(pcols)
    integer  :: jt_c(pcols), maxg_c(pcols), ideep_c(pcols), lengath_c
    real(r8), pointer :: org(:,:), orgt(:,:), org2d(:,:)
    type(zm_conv_t) :: conv
    type(zm_aero_t) :: aero
    integer :: c0, nc, ic

    nullify(org, orgt, org2d)
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      call pack2(t, c0, nc, pver, t_c);     call pack2(qh, c0, nc, pver, qh_c)
      call pack2(zm, c0, nc, pver, zm_c);   call pack2(zi, c0, nc, pverp, zi_c)
      call pack2(pap, c0, nc, pver, pap_c); call pack2(paph, c0, nc, pverp, paph_c)
      call pack2(dpp, c0, nc, pver, dpp_c)
      call pack1(pblh, c0, nc, pblh_c);     call pack1(geos, c0, nc, geos_c)
      call pack1(tpert, c0, nc, tpert_c);   call pack1(landfrac, c0, nc, landfrac_c)
      ! gathered outputs: deterministic content beyond lengath
      mu_c = 0._r8; md_c = 0._r8; du_c = 0._r8; eu_c = 0._r8; ed_c = 0._r8; dp_c = 0._r8
      jt_c = 0; maxg_c = 0; ideep_c = 0; dsubcld_c = 0._r8
      call zm_convr(ic, nc, t_c, qh_c, prec_c, jctop_c, jcbot_c, pblh_c, zm_c, geos_c, zi_c, qtnd_c, heat_c, &
                    pap_c, paph_c, dpp_c, delt, mcon_c, cme_c, cape_c, eurt_c, tpert_c, dlf_c, pflx_c, zdu_c, rprd_c, &
                    mu_c, md_c, du_c, eu_c, ed_c, dp_c, dsubcld_c, jt_c, maxg_c, ideep_c, lengath_c, ql_c, rliq_c, &
                    landfrac_c, org, orgt, org2d, dif_c, dnlf_c, dnif_c, conv, aero, rice_c)
      lengath(ic) = lengath_c
      if (lengath_c < pcols) then     ! normalise the don't-care tail (SURVEY.md 7.3-6)
        mu_c(lengath_c+1:,:) = 0._r8; md_c(lengath_c+1:,:) = 0._r8; du_c(lengath_c+1:,:) = 0._r8
        eu_c(lengath_c+1:,:) = 0._r8; ed_c(lengath_c+1:,:) = 0._r8; dp_c(lengath_c+1:,:) = 0._r8
        jt_c(lengath_c+1:) = 0; maxg_c(lengath_c+1:) = 0; dsubcld_c(lengath_c+1:) = 0._r8
      end if
      call unpack2(qtnd_c, c0, nc, pver, qtnd); call unpack2(heat_c, c0, nc, pver, heat)
      call unpack2(mcon_c, c0, nc, pverp, mcon); call unpack2(dlf_c, c0, nc, pver, dlf)
      call unpack2(pflx_c, c0, nc, pverp, pflx); call unpack2(cme_c, c0, nc, pver, cme)
      call unpack2(zdu_c, c0, nc, pver, zdu);   call unpack2(rprd_c, c0, nc, pver, rprd)
      call unpack2(dif_c, c0, nc, pver, dif);   call unpack2(dnlf_c, c0, nc, pver, dnlf)
      call unpack2(dnif_c, c0, nc, pver, dnif); call unpack2(eurt_c, c0, nc, pver, eurt)
      call unpack2(ql_c, c0, nc, pver, ql)
      call unpack2(mu_c, c0, nc, pver, mu); call unpack2(md_c, c0, nc, pver, md); call unpack2(du_c, c0, nc, pver, du)
      call unpack2(eu_c, c0, nc, pver, eu); call unpack2(ed_c, c0, nc, pver, ed); call unpack2(dp_c, c0, nc, pver, dp)
      cape(c0+1:c0+nc) = cape_c(1:nc);   dsubcld(c0+1:c0+nc) = dsubcld_c(1:nc)
      jctop(c0+1:c0+nc) = jctop_c(1:nc); jcbot(c0+1:c0+nc) = jcbot_c(1:nc)
      prec(c0+1:c0+nc) = prec_c(1:nc);   rliq(c0+1:c0+nc) = rliq_c(1:nc); rice(c0+1:c0+nc) = rice_c(1:nc)
      jt(c0+1:c0+nc) = jt_c(1:nc); maxg(c0+1:c0+nc) = maxg_c(1:nc); ideep(c0+1:c0+nc) = ideep_c(1:nc)
      c0 = c0 + nc
    end do
  end subroutine ref_zm_convr

  ! ---- zm_conv_evap (zm_conv.F90:1712) -----------------------------------------------------------
  subroutine ref_zm_conv_evap(n, t, pmid, pdel, q, landfrac, prdprec, cldfrc, deltat, prec, &
       tend_s, tend_s_snwprd, tend_s_snwevmlt, tend_q, snow, ntprprd, ntsnprd, flxprec, flxsnow) &
       bind(C, name='ref_zm_conv_evap')
    integer(c_int), value :: n
    real(c_double), intent(in)    :: t(pver,n), pmid(pver,n), pdel(pver,n), q(pver,n), prdprec(pver,n), cldfrc(pver,n)
    real(c_double), intent(in)    :: landfrac(n)
    real(c_double), value         :: deltat
    real(c_double), intent(inout) :: prec(n)
    real(c_double), intent(out)   :: tend_s(pver,n), tend_s_snwprd(pver,n), tend_s_snwevmlt(pver,n), tend_q(pver,n), &
                                     ntprprd(pver,n), ntsnprd(pver,n), flxprec(pverp,n), flxsnow(pverp,n), snow(n)
    real(r8) :: t_c(pcols,pver), pmid_c(pcols,pver), pdel_c(pcols,pver), q_c(pcols,pver), prd_c(pcols,pver), &
                cld_c(pcols,pver), lf_c(pcols), prec_c(pcols), snow_c(pcols)
    real(r8) :: ts_c(pcols,pver), tsp_c(pcols,pver), tsm_c(pcols,pver), tq_c(pcols,pver), np_c(pcols,pver), &
                ns_c(pcols,pver), fp_c(pcols,pverp), fs_c(pcols,pverp)
    integer :: c0, nc, ic
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      call pack2(t, c0, nc, pver, t_c);       call pack2(pmid, c0, nc, pver, pmid_c)
      call pack2(pdel, c0, nc, pver, pdel_c); call pack2(q, c0, nc, pver, q_c)
      call pack2(prdprec, c0, nc, pver, prd_c); call pack2(cldfrc, c0, nc, pver, cld_c)
      call pack1(landfrac, c0, nc, lf_c);     call pack1(prec, c0, nc, prec_c)
      ts_c = 0._r8; tq_c = 0._r8
      call zm_conv_evap(nc, ic, t_c, pmid_c, pdel_c, q_c, lf_c, ts_c, tsp_c, tsm_c, tq_c, prd_c, cld_c, deltat, &
                        prec_c, snow_c, np_c, ns_c, fp_c, fs_c)
      call unpack2(ts_c, c0, nc, pver, tend_s);  call unpack2(tsp_c, c0, nc, pver, tend_s_snwprd)
      call unpack2(tsm_c, c0, nc, pver, tend_s_snwevmlt); call unpack2(tq_c, c0, nc, pver, tend_q)
      call unpack2(np_c, c0, nc, pver, ntprprd); call unpack2(ns_c, c0, nc, pver, ntsnprd)
      call unpack2(fp_c, c0, nc, pverp, flxprec); call unpack2(fs_c, c0, nc, pverp, flxsnow)
      prec(c0+1:c0+nc) = prec_c(1:nc); snow(c0+1:c0+nc) = snow_c(1:nc)
      c0 = c0 + nc
    end do
  end subroutine ref_zm_conv_evap

  ! ---- momtran (zm_conv.F90:2315) ----------------------------------------------------------------
  subroutine ref_momtran(n, winds, mu, md, du, eu, ed, dp, dsubcld, jt, mx, ideep, lengath, dt, &
       dqdt, pguall, pgdall, icwu, icwd, seten) bind(C, name='ref_momtran')
    integer(c_int), value :: n
    real(c_double), intent(in)  :: winds(pver,n,2), mu(pver,n), md(pver,n), du(pver,n), eu(pver,n), ed(pver,n), dp(pver,n)
    real(c_double), intent(in)  :: dsubcld(n)
    integer(c_int), intent(in)  :: jt(n), mx(n), ideep(n), lengath(*)
    real(c_double), value       :: dt
    real(c_double), intent(out) :: dqdt(pver,n,2), pguall(pver,n,2), pgdall(pver,n,2), icwu(pver,n,2), icwd(pver,n,2), &
                                   seten(pver,n)
    real(r8) :: w_c(pcols,pver,2), mu_c(pcols,pver), md_c(pcols,pver), du_c(pcols,pver), eu_c(pcols,pver), &
                ed_c(pcols,pver), dp_c(pcols,pver), ds_c(pcols)
    real(r8) :: dq_c(pcols,pver,2), pgu_c(pcols,pver,2), pgd_c(pcols,pver,2), iu_c(pcols,pver,2), id_c(pcols,pver,2), &
                se_c(pcols,pver)
    integer  :: jt_c(pcols), mx_c(pcols), id_cc(pcols)
    logical  :: l_windt(2)
    integer :: c0, nc, ic, m
    l_windt = .true.
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      do m = 1, 2
        call pack2(winds(:,:,m), c0, nc, pver, w_c(:,:,m))
      end do
      call pack2(mu, c0, nc, pver, mu_c); call pack2(md, c0, nc, pver, md_c); call pack2(du, c0, nc, pver, du_c)
      call pack2(eu, c0, nc, pver, eu_c); call pack2(ed, c0, nc, pver, ed_c); call pack2(dp, c0, nc, pver, dp_c)
      call pack1(dsubcld, c0, nc, ds_c)
      jt_c = 0; mx_c = 0; id_cc = 0
      jt_c(1:nc) = jt(c0+1:c0+nc); mx_c(1:nc) = mx(c0+1:c0+nc); id_cc(1:nc) = ideep(c0+1:c0+nc)
      call momtran(ic, nc, l_windt, w_c, 2, mu_c, md_c, du_c, eu_c, ed_c, dp_c, ds_c, jt_c, mx_c, id_cc, 1, lengath(ic), &
                   0, dq_c, pgu_c, pgd_c, iu_c, id_c, dt, se_c)
      do m = 1, 2
        call unpack2(dq_c(:,:,m), c0, nc, pver, dqdt(:,:,m));   call unpack2(pgu_c(:,:,m), c0, nc, pver, pguall(:,:,m))
        call unpack2(pgd_c(:,:,m), c0, nc, pver, pgdall(:,:,m)); call unpack2(iu_c(:,:,m), c0, nc, pver, icwu(:,:,m))
        call unpack2(id_c(:,:,m), c0, nc, pver, icwd(:,:,m))
      end do
      call unpack2(se_c, c0, nc, pver, seten)
      c0 = c0 + nc
    end do
  end subroutine ref_momtran

  ! ---- convtran (zm_conv.F90:1976) ---------------------------------------------------------------
  subroutine ref_convtran(n, ncnst, doconvtran, isdry, q, mu, md, du, eu, ed, dp, dsubcld, jt, mx, ideep, lengath, &
       fracis, dpdry, dt, dqdt) bind(C, name='ref_convtran')
    use constituents, only: cnst_is_dry
    integer(c_int), value :: n, ncnst
    integer(c_int), intent(in)  :: doconvtran(ncnst), isdry(ncnst)
    real(c_double), intent(in)  :: q(pver,n,ncnst), fracis(pver,n,ncnst)
    real(c_double), intent(in)  :: mu(pver,n), md(pver,n), du(pver,n), eu(pver,n), ed(pver,n), dp(pver,n), dpdry(pver,n)
    real(c_double), intent(in)  :: dsubcld(n)
    integer(c_int), intent(in)  :: jt(n), mx(n), ideep(n), lengath(*)
    real(c_double), value       :: dt
    real(c_double), intent(out) :: dqdt(pver,n,ncnst)
    real(r8) :: q_c(pcols,pver,ncnst), f_c(pcols,pver,ncnst), dq_c(pcols,pver,ncnst)
    real(r8) :: mu_c(pcols,pver), md_c(pcols,pver), du_c(pcols,pver), eu_c(pcols,pver), ed_c(pcols,pver), dp_c(pcols,pver), &
                dpd_c(pcols,pver), ds_c(pcols)
    integer  :: jt_c(pcols), mx_c(pcols), id_cc(pcols)
    logical  :: doit(ncnst)
    integer :: c0, nc, ic, m
    doit = doconvtran /= 0
    cnst_is_dry = .false.
    cnst_is_dry(1:ncnst) = isdry /= 0
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      do m = 1, ncnst
        call pack2(q(:,:,m), c0, nc, pver, q_c(:,:,m)); call pack2(fracis(:,:,m), c0, nc, pver, f_c(:,:,m))
      end do
      call pack2(mu, c0, nc, pver, mu_c); call pack2(md, c0, nc, pver, md_c); call pack2(du, c0, nc, pver, du_c)
      call pack2(eu, c0, nc, pver, eu_c); call pack2(ed, c0, nc, pver, ed_c); call pack2(dp, c0, nc, pver, dp_c)
      call pack2(dpdry, c0, nc, pver, dpd_c); call pack1(dsubcld, c0, nc, ds_c)
      jt_c = 0; mx_c = 0; id_cc = 0
      jt_c(1:nc) = jt(c0+1:c0+nc); mx_c(1:nc) = mx(c0+1:c0+nc); id_cc(1:nc) = ideep(c0+1:c0+nc)
      dq_c = 0._r8
      call convtran(ic, doit, q_c, ncnst, mu_c, md_c, du_c, eu_c, ed_c, dp_c, ds_c, jt_c, mx_c, id_cc, 1, lengath(ic), &
                    0, f_c, dq_c, dpd_c, dt)
      do m = 1, ncnst
        call unpack2(dq_c(:,:,m), c0, nc, pver, dqdt(:,:,m))
      end do
      c0 = c0 + nc
    end do
  end subroutine ref_convtran

#ifndef COLPHYS_SHIM_BUILD
  ! ---- L0 seams (oracle/stubs/l0_arithmetic.F90), exported so the C restatement can be pinned to them ----
  ! (oracle builds only: the shim libraries compile this file with -DCOLPHYS_SHIM_BUILD and link nothing of the oracle's arithmetic)
  subroutine ref_qsat_water(n, t, p, es, qs) bind(C, name='ref_qsat_water')
    use wv_saturation, only: qsat_water
    integer(c_int), value :: n
    real(c_double), intent(in)  :: t(n), p(n)
    real(c_double), intent(out) :: es(n), qs(n)
    integer :: i
    do i = 1, n
      call qsat_water(t(i), p(i), es(i), qs(i))
    end do
  end subroutine ref_qsat_water

  subroutine ref_qsat_table(n, t, p, es, qs) bind(C, name='ref_qsat_table')
    use wv_saturation, only: qsat
    integer(c_int), value :: n
    real(c_double), intent(in)  :: t(n), p(n)
    real(c_double), intent(out) :: es(n), qs(n)
    call qsat(t, p, es, qs, n)
  end subroutine ref_qsat_table
#endif

end module ref_wrap_zm
