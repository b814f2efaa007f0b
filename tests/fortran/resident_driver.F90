! TEST DRIVER (tests/test_fortran_resident_gpu.py) -- a Fortran caller of the whole-node dispatch:
! physics_state fields of ALL chunks of a rank in CAM's (pcols,pver,nchunks) layout -> colphys_resident uploads them once ->
! ONE colphys_zm_conv_tend_dev call -> ptend_all and the surface / pbuf fields come back in CAM's layout.
! bind(C) so that the Python test can hand it numpy arrays; a CAM build would call colphys_resident from tphysbc's
! restructured chunk loop the same way (INTEGRATION.md 3, "Whole node").
subroutine res_zm_conv_tend(nchunks, ncol, ztodt, dycore_kind, qmin,                                  &
     t, q, u, v, pmid, pint, pdel, rpdel, lnpint, lnpmid, zm, zi, phis, ps, rair, zvir, pblh, tpert, landfrac, cld, &
     ptend_s, ptend_q, ptend_u, ptend_v, prec, snow, rliq, jctop, jcbot, mconzm, evapcdp, mu, jt, maxg, ideep, lengath, nbad) &
     bind(C, name='res_zm_conv_tend')
  use iso_c_binding
  use ppgrid, only: pcols, pver, pverp
  use colphys_c_api
  use colphys_resident
  implicit none
  integer(c_int), value :: nchunks, dycore_kind
  integer(c_int), intent(in) :: ncol(nchunks)
  real(c_double), value :: ztodt, qmin
  real(c_double), intent(in) :: t(pcols,pver,nchunks), q(pcols,pver,nchunks), u(pcols,pver,nchunks), v(pcols,pver,nchunks)
  real(c_double), intent(in) :: pmid(pcols,pver,nchunks), pint(pcols,pverp,nchunks), pdel(pcols,pver,nchunks), rpdel(pcols,pver,nchunks)
  real(c_double), intent(in) :: lnpint(pcols,pverp,nchunks), lnpmid(pcols,pver,nchunks), zm(pcols,pver,nchunks), zi(pcols,pverp,nchunks)
  real(c_double), intent(in) :: phis(pcols,1,nchunks), ps(pcols,1,nchunks), rair(pcols,pver,nchunks), zvir(pcols,pver,nchunks)
  real(c_double), intent(in) :: pblh(pcols,1,nchunks), tpert(pcols,1,nchunks), landfrac(pcols,1,nchunks), cld(pcols,pver,nchunks)
  real(c_double), intent(inout) :: ptend_s(pcols,pver,nchunks), ptend_q(pcols,pver,nchunks), ptend_u(pcols,pver,nchunks), &
                                   ptend_v(pcols,pver,nchunks), prec(pcols,1,nchunks), snow(pcols,1,nchunks), rliq(pcols,1,nchunks), &
                                   jctop(pcols,1,nchunks), jcbot(pcols,1,nchunks), mconzm(pcols,pverp,nchunks), &
                                   evapcdp(pcols,pver,nchunks), mu(pcols,pver,nchunks)
  integer(c_int), intent(inout) :: jt(pcols,nchunks), maxg(pcols,nchunks), ideep(pcols,nchunks), lengath(nchunks)
  integer(c_int), intent(out) :: nbad

  integer, parameter :: NIN = 20, NMID = 14, NINT = 3, NCOLF = 11
  type(colphys_field) :: f_in(NIN), f_mid(NMID), f_int(NINT), f_col(NCOLF), f_jt, f_maxg, f_ideep, f_len, f_status
  type(colphys_zm_conv_tend_io) :: io
  integer :: nrows, i, rc, st(pcols, nchunks)

  nrows = nchunks * pcols
  ! ---- physics_state of the rank, resident --------------------------------------------------------------------
  call colphys_upload(f_in(1), t, pcols, pver, nchunks, ncol);       call colphys_upload(f_in(2), q, pcols, pver, nchunks, ncol)
  call colphys_upload(f_in(3), u, pcols, pver, nchunks, ncol);       call colphys_upload(f_in(4), v, pcols, pver, nchunks, ncol)
  call colphys_upload(f_in(5), pmid, pcols, pver, nchunks, ncol);    call colphys_upload(f_in(6), pint, pcols, pverp, nchunks, ncol)
  call colphys_upload(f_in(7), pdel, pcols, pver, nchunks, ncol);    call colphys_upload(f_in(8), rpdel, pcols, pver, nchunks, ncol)
  call colphys_upload(f_in(9), lnpint, pcols, pverp, nchunks, ncol); call colphys_upload(f_in(10), lnpmid, pcols, pver, nchunks, ncol)
  call colphys_upload(f_in(11), zm, pcols, pver, nchunks, ncol);     call colphys_upload(f_in(12), zi, pcols, pverp, nchunks, ncol)
  call colphys_upload(f_in(13), phis, pcols, 1, nchunks, ncol);      call colphys_upload(f_in(14), ps, pcols, 1, nchunks, ncol)
  call colphys_upload(f_in(15), rair, pcols, pver, nchunks, ncol);   call colphys_upload(f_in(16), zvir, pcols, pver, nchunks, ncol)
  call colphys_upload(f_in(17), pblh, pcols, 1, nchunks, ncol);      call colphys_upload(f_in(18), tpert, pcols, 1, nchunks, ncol)
  call colphys_upload(f_in(19), landfrac, pcols, 1, nchunks, ncol);  call colphys_upload(f_in(20), cld, pcols, pver, nchunks, ncol)
  io%t = f_in(1)%d; io%q = f_in(2)%d; io%u = f_in(3)%d; io%v = f_in(4)%d; io%pmid = f_in(5)%d; io%pint = f_in(6)%d
  io%pdel = f_in(7)%d; io%rpdel = f_in(8)%d; io%lnpint = f_in(9)%d; io%lnpmid = f_in(10)%d; io%zm = f_in(11)%d; io%zi = f_in(12)%d
  io%phis = f_in(13)%d; io%ps = f_in(14)%d; io%rair = f_in(15)%d; io%zvir = f_in(16)%d; io%pblh = f_in(17)%d
  io%tpert = f_in(18)%d; io%landfrac = f_in(19)%d; io%cld = f_in(20)%d
  ! ---- outputs --------------------------------------------------------------------------------------------------
  do i = 1, NMID;  call colphys_alloc(f_mid(i), nrows, pver);  end do
  do i = 1, NINT;  call colphys_alloc(f_int(i), nrows, pverp); end do
  do i = 1, NCOLF; call colphys_alloc(f_col(i), nrows, 1);     end do
  call colphys_alloc_int(f_jt, nrows); call colphys_alloc_int(f_maxg, nrows); call colphys_alloc_int(f_ideep, nrows)
  call colphys_alloc_int(f_len, nchunks); call colphys_alloc_int(f_status, nrows)
  io%ptend_s = f_mid(1)%d; io%ptend_q = f_mid(2)%d; io%ptend_u = f_mid(3)%d; io%ptend_v = f_mid(4)%d
  io%ql = f_mid(5)%d; io%rprd = f_mid(6)%d; io%evapcdp = f_mid(7)%d; io%dlf = f_mid(8)%d
  io%mu = f_mid(9)%d; io%md = f_mid(10)%d; io%du = f_mid(11)%d; io%eu = f_mid(12)%d; io%ed = f_mid(13)%d; io%dp = f_mid(14)%d
  io%mconzm = f_int(1)%d; io%flxprec = f_int(2)%d; io%flxsnow = f_int(3)%d
  io%prec = f_col(1)%d; io%snow = f_col(2)%d; io%rliq = f_col(3)%d; io%rice = f_col(4)%d; io%jctop = f_col(5)%d
  io%jcbot = f_col(6)%d; io%cape = f_col(7)%d; io%dsubcld = f_col(8)%d
  io%freqzm = f_col(9)%d; io%pcont = f_col(10)%d; io%pconb = f_col(11)%d
  io%jt = f_jt%d; io%maxg = f_maxg%d; io%ideep = f_ideep%d; io%lengath = f_len%d; io%status = f_status%d
  io%mu_out = c_null_ptr; io%md_out = c_null_ptr
  io%t1 = c_null_ptr; io%q1 = c_null_ptr; io%u1 = c_null_ptr; io%v1 = c_null_ptr; io%zi1 = c_null_ptr; io%zm1 = c_null_ptr

  ! ---- zm_conv_tend for every chunk of the rank: one call ----------------------------------------------------------
  rc = colphys_zm_conv_tend_dev(nrows, pver, pcols, ztodt, dycore_kind, qmin, io, c_null_ptr)
  if (rc /= COLPHYS_OK) then
     nbad = -1
     return
  end if
  call colphys_sync()

  ! ---- what the host needs back: ptend_all, surface fluxes, the gathered sets -----------------------------------------
  call colphys_download(f_mid(1), ptend_s, pcols, pver, nchunks, ncol); call colphys_download(f_mid(2), ptend_q, pcols, pver, nchunks, ncol)
  call colphys_download(f_mid(3), ptend_u, pcols, pver, nchunks, ncol); call colphys_download(f_mid(4), ptend_v, pcols, pver, nchunks, ncol)
  call colphys_download(f_mid(7), evapcdp, pcols, pver, nchunks, ncol)
  call colphys_download(f_int(1), mconzm, pcols, pverp, nchunks, ncol)
  call colphys_download(f_col(1), prec, pcols, 1, nchunks, ncol); call colphys_download(f_col(2), snow, pcols, 1, nchunks, ncol)
  call colphys_download(f_col(3), rliq, pcols, 1, nchunks, ncol); call colphys_download(f_col(5), jctop, pcols, 1, nchunks, ncol)
  call colphys_download(f_col(6), jcbot, pcols, 1, nchunks, ncol)
  call colphys_download_gathered_sets(f_ideep, f_len, ideep, lengath, pcols, nchunks, ncol)
  call colphys_download(f_mid(9), mu, pcols, pver, nchunks, lengath)
  call colphys_download_int(f_jt, jt, pcols, nchunks, lengath); call colphys_download_int(f_maxg, maxg, pcols, nchunks, lengath)
  st = 0
  call colphys_download_int(f_status, st, pcols, nchunks, ncol)
  nbad = count(st /= 0)
  call colphys_check_zm_status(st, ncol)            ! ends the run where the reference's ientropy / ienthalpy would

  do i = 1, NIN;   call colphys_release(f_in(i));  end do
  do i = 1, NMID;  call colphys_release(f_mid(i)); end do
  do i = 1, NINT;  call colphys_release(f_int(i)); end do
  do i = 1, NCOLF; call colphys_release(f_col(i)); end do
  call colphys_release(f_jt); call colphys_release(f_maxg); call colphys_release(f_ideep); call colphys_release(f_len)
  call colphys_release(f_status)
end subroutine res_zm_conv_tend


! The optics of a rank on resident data from Fortran: inputs of ALL chunks uploaded once (species mixing ratios in CAM's
! (pcols,pver,species,chunk) order), modal_aero_sw with the hand-over buffer, its history diagnostics, modal_aero_lw; the band
! outputs come back as (pcols, level, band, chunk) -- the device's [column][band][level] is exactly that per column.
subroutine res_modal_aero(nchunks, ncol, nmodes, nspec_max, nsw, nlw, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, &
     tauxar, wa, ga, fa, lwtau, dlev, dcol, nbad) bind(C, name='res_modal_aero')
  use iso_c_binding
  use ppgrid, only: pcols, pver, pverp
  use colphys_c_api
  use colphys_resident
  implicit none
  integer(c_int), value :: nchunks, nmodes, nspec_max, nsw, nlw
  integer(c_int), intent(in) :: ncol(nchunks), troplev(pcols, nchunks)
  real(c_double), intent(in) :: pdeldry(pcols,pver,nchunks), pmid(pcols,pver,nchunks), t(pcols,pver,nchunks)
  real(c_double), intent(in) :: specmmr(pcols,pver,nspec_max*nmodes,nchunks), dgnumwet(pcols,pver,nmodes,nchunks), &
                                qaerwat(pcols,pver,nmodes,nchunks)
  real(c_double), intent(inout) :: tauxar(pcols,pverp*nsw,nchunks), wa(pcols,pverp*nsw,nchunks), ga(pcols,pverp*nsw,nchunks), &
                                   fa(pcols,pverp*nsw,nchunks), lwtau(pcols,pver*nlw,nchunks)
  real(c_double), intent(inout) :: dlev(pcols,pver,nchunks,5), dcol(pcols,1,nchunks,23+3*nmodes)
  integer(c_int), intent(out) :: nbad

  type(colphys_field) :: f_pd, f_pm, f_t, f_mmr, f_dg, f_qw, f_tl, f_out(5), f_vis, f_lev, f_col, f_st
  type(colphys_modal_aero_io) :: io
  integer :: nrows, i, rc, st(pcols, nchunks), f
  integer, save :: nerr_dopaer_sw = 0
  type(colphys_field) :: view

  nrows = nchunks * pcols
  call colphys_upload(f_pd, pdeldry, pcols, pver, nchunks, ncol); call colphys_upload(f_pm, pmid, pcols, pver, nchunks, ncol)
  call colphys_upload(f_t, t, pcols, pver, nchunks, ncol)
  call colphys_upload_multi(f_mmr, specmmr, pcols, pver, nspec_max*nmodes, nchunks, ncol)
  call colphys_upload_multi(f_dg, dgnumwet, pcols, pver, nmodes, nchunks, ncol)
  call colphys_upload_multi(f_qw, qaerwat, pcols, pver, nmodes, nchunks, ncol)
  call colphys_upload_int(f_tl, troplev, pcols, nchunks, ncol)
  do i = 1, 4; call colphys_alloc(f_out(i), nrows, pverp*nsw); end do
  call colphys_alloc(f_out(5), nrows, pver*nlw)
  call colphys_alloc(f_vis, 3*nmodes*nrows, pver); call colphys_alloc(f_lev, 5*nrows, pver); call colphys_alloc(f_col, (23+3*nmodes)*nrows, 1)
  call colphys_alloc_int(f_st, nrows)

  io%pdeldry = f_pd%d; io%pmid = f_pm%d; io%t = f_t%d; io%specmmr = f_mmr%d; io%dgnumwet = f_dg%d; io%qaerwat = f_qw%d
  io%troplevchem = f_tl%d; io%status = f_st%d
  io%tauxar = f_out(1)%d; io%wa = f_out(2)%d; io%ga = f_out(3)%d; io%fa = f_out(4)%d; io%vis_optics = f_vis%d
  nbad = -1
  rc = colphys_modal_aero_sw_dev(nrows, pver, io, c_null_ptr);                       if (rc /= COLPHYS_OK) return
  rc = colphys_modal_aero_sw_diag_dev(nrows, pver, io, f_lev%d, f_col%d, c_null_ptr); if (rc /= COLPHYS_OK) return
  call colphys_sync()
  st = 0
  call colphys_download_int(f_st, st, pcols, nchunks, ncol)
  nbad = count(st /= 0)
  call colphys_check_optics_status('modal_aero_sw', st, ncol, nerr_dopaer_sw, -1)   ! warns / counts / halts as modal_aero_sw does
  call colphys_download(f_out(1), tauxar, pcols, pverp*nsw, nchunks, ncol); call colphys_download(f_out(2), wa, pcols, pverp*nsw, nchunks, ncol)
  call colphys_download(f_out(3), ga, pcols, pverp*nsw, nchunks, ncol);     call colphys_download(f_out(4), fa, pcols, pverp*nsw, nchunks, ncol)
  ! the packed diagnostics are field-major on the device: one view per field
  do f = 1, 5
     view%d = c_ptr_plus(f_lev%d, int(f-1, c_size_t) * int(nrows, c_size_t) * int(pver, c_size_t) * 8_c_size_t)
     view%nrows = nrows; view%nlev = pver
     call colphys_download(view, dlev(:,:,:,f), pcols, pver, nchunks, ncol)
  end do
  do f = 1, 23 + 3*nmodes
     view%d = c_ptr_plus(f_col%d, int(f-1, c_size_t) * int(nrows, c_size_t) * 8_c_size_t)
     view%nrows = nrows; view%nlev = 1
     call colphys_download(view, dcol(:,:,:,f), pcols, 1, nchunks, ncol)
  end do
  ! modal_aero_lw: the same resident inputs, its own output
  io%tauxar = f_out(5)%d; io%wa = c_null_ptr; io%ga = c_null_ptr; io%fa = c_null_ptr; io%vis_optics = c_null_ptr
  rc = colphys_modal_aero_lw_dev(nrows, pver, io, c_null_ptr)
  if (rc /= COLPHYS_OK) then
     nbad = -1
     return
  end if
  call colphys_sync()
  call colphys_download_int(f_st, st, pcols, nchunks, ncol)
  nbad = nbad + count(st /= 0)
  call colphys_download(f_out(5), lwtau, pcols, pver*nlw, nchunks, ncol)

  call colphys_release(f_pd); call colphys_release(f_pm); call colphys_release(f_t); call colphys_release(f_mmr)
  call colphys_release(f_dg); call colphys_release(f_qw); call colphys_release(f_tl); call colphys_release(f_vis)
  call colphys_release(f_lev); call colphys_release(f_col); call colphys_release(f_st)
  do i = 1, 5; call colphys_release(f_out(i)); end do

contains
  function c_ptr_plus(p, bytes) result(q)
    type(c_ptr), intent(in) :: p
    integer(c_size_t), intent(in) :: bytes
    type(c_ptr) :: q
    q = transfer(transfer(p, 0_c_intptr_t) + int(bytes, c_intptr_t), q)
  end function c_ptr_plus
end subroutine res_modal_aero
