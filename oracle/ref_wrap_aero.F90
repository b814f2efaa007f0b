! TEST INFRASTRUCTURE -- not product code.
!
! bind(C) entry points around the UNMODIFIED reference module modal_aer_opt (physics/modal_aer_opt.F90),
! compiled in place by oracle/Makefile.  Tables arrive in the C layouts of colphys_aero_tables
! (include/colphys.h) and are stored in the accessor stubs (oracle/stubs/aero_interfaces.F90), from which the
! reference fetches them exactly as it does from rad_constituents / pbuf / pio in CAM.
! Column layout as in ref_wrap_zm.F90; optical depths come back as [n][nbands][levels].

module ref_wrap_aero
  use iso_c_binding
  use shr_kind_mod,     only: r8 => shr_kind_r8
  use ppgrid,           only: pcols, pver, pverp
  use radconstants,     only: nswbands, nlwbands
  use modal_aer_opt,    only: modal_aer_opt_init, modal_aero_sw, modal_aero_lw
  use physics_types,    only: physics_state
  use physics_buffer,   only: physics_buffer_desc, dgnumwet_store, qaerwat_store
  use rad_constituents
  implicit none
  character(len=32), parameter :: type_names(0:5) = (/ 'sulfate  ', 'p-organic', 's-organic', 'black-c  ', 'dust     ', 'seasalt  ' /)
contains

  subroutine ref_aero_set_tables(nmodes, nspec_max, top_lev, nspec, spectype, sigmag, specdens, hygro_in, sri_sw, sri_lw, &
       refrtabsw, refitabsw, refrtablw, refitablw, extpsw, abspsw, asmpsw, absplw, crefwsw, crefwlw) &
       bind(C, name='ref_aero_set_tables')
    use pio,      only: wat_r_sw, wat_i_sw, wat_r_lw, wat_i_lw
    use ref_pres, only: clim_modal_aero_top_lev
    integer(c_int), value :: nmodes, nspec_max, top_lev
    integer(c_int), intent(in) :: nspec(nmodes), spectype(nspec_max, nmodes)
    real(c_double), intent(in) :: sigmag(nmodes), specdens(nspec_max, nmodes), hygro_in(nspec_max, nmodes)
    real(c_double), intent(in) :: sri_sw(2, nswbands, nspec_max, nmodes), sri_lw(2, nlwbands, nspec_max, nmodes)
    real(c_double), intent(in) :: refrtabsw(7, nswbands, nmodes), refitabsw(10, nswbands, nmodes)
    real(c_double), intent(in) :: refrtablw(7, nlwbands, nmodes), refitablw(10, nlwbands, nmodes)
    real(c_double), intent(in) :: extpsw(5,7,10,nswbands,nmodes), abspsw(5,7,10,nswbands,nmodes), asmpsw(5,7,10,nswbands,nmodes)
    real(c_double), intent(in) :: absplw(5,7,10,nlwbands,nmodes)
    real(c_double), intent(in) :: crefwsw(2, nswbands), crefwlw(2, nlwbands)
    integer :: m, l
    nm = nmodes; nsmax = nspec_max
    clim_modal_aero_top_lev = top_lev
    if (allocated(nspec_m)) deallocate(nspec_m, mmr_store, dens, hygro, sigmag_m, ri_sw, ri_lw, stype, refrsw, refisw, &
                                       refrlw, refilw, ext, abss, asmp, abl, dgnumwet_store, qaerwat_store)
    allocate(nspec_m(nm), mmr_store(pcols,pver,nsmax,nm), dens(nsmax,nm), hygro(nsmax,nm), sigmag_m(nm), &
             ri_sw(nswbands,nsmax,nm), ri_lw(nlwbands,nsmax,nm), stype(nsmax,nm), refrsw(7,nswbands,nm), &
             refisw(10,nswbands,nm), refrlw(7,nlwbands,nm), refilw(10,nlwbands,nm), ext(5,7,10,nswbands,nm), &
             abss(5,7,10,nswbands,nm), asmp(5,7,10,nswbands,nm), abl(5,7,10,nlwbands,nm), &
             dgnumwet_store(pcols,pver,nm), qaerwat_store(pcols,pver,nm))
    nspec_m = nspec; sigmag_m = sigmag; dens = specdens; hygro = hygro_in
    do m = 1, nm
      do l = 1, nsmax
        if (spectype(l,m) >= 0 .and. spectype(l,m) <= 5) then
          stype(l,m) = type_names(spectype(l,m))
        else
          stype(l,m) = 'other'
        end if
        ri_sw(:,l,m) = cmplx(sri_sw(1,:,l,m), sri_sw(2,:,l,m), kind=r8)
        ri_lw(:,l,m) = cmplx(sri_lw(1,:,l,m), sri_lw(2,:,l,m), kind=r8)
      end do
    end do
    refrsw = refrtabsw; refisw = refitabsw; refrlw = refrtablw; refilw = refitablw
    ext = extpsw; abss = abspsw; asmp = asmpsw; abl = absplw
    wat_r_sw = crefwsw(1,:); wat_i_sw = crefwsw(2,:); wat_r_lw = crefwlw(1,:); wat_i_lw = crefwlw(2,:)
    call modal_aer_opt_init()
  end subroutine ref_aero_set_tables

  subroutine load_chunk(n, c0, nc, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, state)
    use tropopause, only: troplev_store
    integer, intent(in) :: n, c0, nc
    real(c_double), intent(in) :: pdeldry(pver,n), pmid(pver,n), t(pver,n)
    real(c_double), intent(in) :: specmmr(pver,n,nsmax,nm), dgnumwet(pver,n,nm), qaerwat(pver,n,nm)
    integer(c_int), intent(in) :: troplev(n)
    type(physics_state), intent(inout) :: state
    integer :: i, k, l, m
    state%ncol = nc
    state%t = 1._r8; state%pmid = 1._r8; state%pdeldry = 1._r8
    mmr_store = 0._r8; dgnumwet_store = 1.e-7_r8; qaerwat_store = 0._r8
    troplev_store = 1
    do i = 1, nc
      troplev_store(i) = troplev(c0+i)
      do k = 1, pver
        state%t(i,k) = t(k,c0+i); state%pmid(i,k) = pmid(k,c0+i); state%pdeldry(i,k) = pdeldry(k,c0+i)
        do m = 1, nm
          dgnumwet_store(i,k,m) = dgnumwet(k,c0+i,m); qaerwat_store(i,k,m) = qaerwat(k,c0+i,m)
          do l = 1, nsmax
            mmr_store(i,k,l,m) = specmmr(k,c0+i,l,m)
          end do
        end do
      end do
    end do
  end subroutine load_chunk

  subroutine ref_modal_aero_sw(n, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, tauxar, wa, ga, fa) &
       bind(C, name='ref_modal_aero_sw')
    integer(c_int), value :: n
    real(c_double), intent(in)  :: pdeldry(pver,n), pmid(pver,n), t(pver,n)
    real(c_double), intent(in)  :: specmmr(pver,n,nsmax,nm), dgnumwet(pver,n,nm), qaerwat(pver,n,nm)
    integer(c_int), intent(in)  :: troplev(n)
    real(c_double), intent(out) :: tauxar(0:pver,nswbands,n), wa(0:pver,nswbands,n), ga(0:pver,nswbands,n), fa(0:pver,nswbands,n)
    type(physics_state) :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8) :: tau_c(pcols,0:pver,nswbands), wa_c(pcols,0:pver,nswbands), ga_c(pcols,0:pver,nswbands), fa_c(pcols,0:pver,nswbands)
    integer :: idxnite(1), c0, nc, ic, i, k, b
    nullify(pbuf)
    allocate(state%t(pcols,pver), state%pmid(pcols,pver), state%pdeldry(pcols,pver))
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      state%lchnk = ic
      call load_chunk(n, c0, nc, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, state)
      call modal_aero_sw(0, state, pbuf, 0, idxnite, tau_c, wa_c, ga_c, fa_c)
      do i = 1, nc
        do b = 1, nswbands
          do k = 0, pver
            tauxar(k,b,c0+i) = tau_c(i,k,b); wa(k,b,c0+i) = wa_c(i,k,b); ga(k,b,c0+i) = ga_c(i,k,b); fa(k,b,c0+i) = fa_c(i,k,b)
          end do
        end do
      end do
      c0 = c0 + nc
    end do
  end subroutine ref_modal_aero_sw

  ! modal_aero_sw once more, this time for what it hands to outfld (captured by the cam_history stand-in): the "dn"
  ! fields in lev/col (layouts of colphys_modal_aero_sw_diag_dev, include/colphys.h) and their day-only twins, in
  ! which the columns flagged in nite carry fillvalue, in lev_f/col_f.
  subroutine ref_modal_aero_sw_diag(n, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, nite, lev, col, lev_f, col_f) &
       bind(C, name='ref_modal_aero_sw_diag')
    use cam_history, only: cap_on, cap_n, outfld_captured
    integer(c_int), value :: n
    real(c_double), intent(in)  :: pdeldry(pver,n), pmid(pver,n), t(pver,n)
    real(c_double), intent(in)  :: specmmr(pver,n,nsmax,nm), dgnumwet(pver,n,nm), qaerwat(pver,n,nm)
    integer(c_int), intent(in)  :: troplev(n), nite(n)
    real(c_double), intent(out) :: lev(pver,n,5), col(n,23+3*nm), lev_f(pver,n,5), col_f(n,23+3*nm)
    character(len=16), parameter :: lev_names(5) = (/ 'EXTINCT   ', 'ABSORB    ', 'EXTxASYM  ', 'EXTINCTUV ', 'EXTINCTNIR' /)
    character(len=16), parameter :: col_names(23) = (/ &
         'AODVIS       ', 'AODABS       ', 'AODTOT       ', 'AODVISst     ', 'SSAVIS       ', 'AODxASYM     ', 'AODUV        ', &
         'AODNIR       ', 'AODUVst      ', 'AODNIRst     ', 'BURDENSO4    ', 'BURDENPOM    ', 'BURDENSOA    ', 'BURDENBC     ', &
         'BURDENDUST   ', 'BURDENSEASALT', 'AODABSBC     ', 'AODSO4       ', 'AODPOM       ', 'AODSOA       ', 'AODBC        ', &
         'AODDUST      ', 'AODSS        ' /)
    type(physics_state) :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8) :: tau_c(pcols,0:pver,nswbands), wa_c(pcols,0:pver,nswbands), ga_c(pcols,0:pver,nswbands), fa_c(pcols,0:pver,nswbands)
    real(r8) :: buf(pcols,pverp)
    integer :: idxnite(pcols), nnite, c0, nc, ic, i, k, f, m, nl
    character(len=32) :: nm_dn, nm_day
    nullify(pbuf)
    allocate(state%t(pcols,pver), state%pmid(pcols,pver), state%pdeldry(pcols,pver))
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      state%lchnk = ic
      call load_chunk(n, c0, nc, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, state)
      nnite = 0
      do i = 1, nc
        if (nite(c0+i) /= 0) then
          nnite = nnite + 1; idxnite(nnite) = i
        end if
      end do
      cap_n = 0; cap_on = .true.
      call modal_aero_sw(0, state, pbuf, nnite, idxnite(1:nnite), tau_c, wa_c, ga_c, fa_c)
      cap_on = .false.
      do f = 1, 5
        nl = outfld_captured(trim(lev_names(f))//'dn', pcols, buf)
        do i = 1, nc
          lev(:,c0+i,f) = buf(i,1:pver)
        end do
        nl = outfld_captured(trim(lev_names(f)), pcols, buf)
        do i = 1, nc
          lev_f(:,c0+i,f) = buf(i,1:pver)
        end do
      end do
      do f = 1, 23 + 3*nm
        if (f <= 23) then
          nm_dn = trim(col_names(f))//'dn'; nm_day = col_names(f)
        else
          m = (f - 24)/3 + 1
          select case (mod(f - 24, 3))
          case (0)
            write(nm_dn,'(a,i2.2)') 'BURDENdn', m;  write(nm_day,'(a,i2.2)') 'BURDEN', m
          case (1)
            write(nm_dn,'(a,i1)') 'AODdn_mode', m;  write(nm_day,'(a,i1)') 'AOD_mode', m
          case (2)
            write(nm_dn,'(a,i2.2)') 'AODdnDUST', m; write(nm_day,'(a,i2.2)') 'AODDUST', m
          end select
        end if
        nl = outfld_captured(trim(nm_dn), pcols, buf)
        col(c0+1:c0+nc,f) = buf(1:nc,1)
        nl = outfld_captured(trim(nm_day), pcols, buf)
        col_f(c0+1:c0+nc,f) = buf(1:nc,1)
      end do
      c0 = c0 + nc
    end do
  end subroutine ref_modal_aero_sw_diag

  subroutine ref_modal_aero_lw(n, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, tauxar) &
       bind(C, name='ref_modal_aero_lw')
    integer(c_int), value :: n
    real(c_double), intent(in)  :: pdeldry(pver,n), pmid(pver,n), t(pver,n)
    real(c_double), intent(in)  :: specmmr(pver,n,nsmax,nm), dgnumwet(pver,n,nm), qaerwat(pver,n,nm)
    integer(c_int), intent(in)  :: troplev(n)
    real(c_double), intent(out) :: tauxar(pver,nlwbands,n)
    type(physics_state) :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8) :: tau_c(pcols,pver,nlwbands)
    integer :: c0, nc, ic, i, k, b
    nullify(pbuf)
    allocate(state%t(pcols,pver), state%pmid(pcols,pver), state%pdeldry(pcols,pver))
    c0 = 0; ic = 0
    do while (c0 < n)
      nc = min(pcols, n - c0); ic = ic + 1
      state%lchnk = ic
      call load_chunk(n, c0, nc, pdeldry, pmid, t, specmmr, dgnumwet, qaerwat, troplev, state)
      call modal_aero_lw(0, state, pbuf, tau_c)
      do i = 1, nc
        do b = 1, nlwbands
          do k = 1, pver
            tauxar(k,b,c0+i) = tau_c(i,k,b)
          end do
        end do
      end do
      c0 = c0 + nc
    end do
  end subroutine ref_modal_aero_lw

end module ref_wrap_aero
