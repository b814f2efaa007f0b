! TEST INFRASTRUCTURE -- not product code.
!
! SPECIFICATION OF THE PARITY-UNPINNED SEAMS  (SURVEY.md 8c, "third-party arithmetic").
!
! The reference calls four upstream ESCOMP/CAM routines whose source is NOT in NorESMhub/CAM-Nor-physics
! and whose version the reference does not pin (comments point at cam6_3_041 ... cam6_3_102):
!
!   wv_saturation : qsat_water   (zm_conv.F90:5423,5433 through qsat_hPa; 11 call sites)
!                   qsat         (zm_conv.F90:1729,1804)
!   cloud_fraction: cldfrc_fice  (zm_conv.F90:18,1809)
!   table_interp_mod: table_interp_calcwghts / table_interp  (modal_aer_opt.F90:37,857-862,1383-1386)
!
! They are restated here from the published formulas those routines implement (Goff & Gratch 1946 as
! tabulated in the Smithsonian Meteorological Tables; CAM's linear ice/snow ramps; bilinear interpolation
! with the lower-edge-extrapolating / upper-edge-clamping index search of the classic modal_aer_opt
! "binterp").  Because nothing in the reference pins them, THIS FILE IS THE SPECIFICATION shared by
!   - the oracle build of the reference Fortran      (oracle/_ref, this file linked in),
!   - the C restatement                               (oracle/colphys_oracle.c),
!   - the HIP kernels                                 (cam-nor-physics_amd/csrc/).
! Results at these seams are "parity unpinned" against real CAM builds; DESIGN.md says so.

module wv_saturation
  use shr_kind_mod, only: r8 => shr_kind_r8
  use physconst,    only: epsilo, tmelt
  implicit none
  private
  public :: qsat_water, qsat, wv_sat_table_init, svp_water, svp_ice, estblf
  public :: plenest, tmin_tbl, tmax_tbl, estbl

  real(r8), parameter :: tboil   = 373.16_r8      ! Goff-Gratch steam point
  real(r8), parameter :: h2otrip = 273.16_r8      ! Goff-Gratch ice point
  real(r8), parameter :: ttrice  = 20.00_r8       ! water<->ice transition range below tmelt
  real(r8), parameter :: tmin_tbl = 127.16_r8, tmax_tbl = 375.16_r8
  integer,  parameter :: plenest = 249            ! ceiling(tmax-tmin)+1
  real(r8), save :: estbl(plenest)
  logical,  save :: table_ready = .false.

  interface qsat_water
    module procedure qsat_water_scalar
  end interface qsat_water

contains

  elemental function svp_water(t) result(es)      ! Pa
    real(r8), intent(in) :: t
    real(r8) :: es
    es = 10._r8**(-7.90298_r8*(tboil/t-1._r8)+ &
         5.02808_r8*log10(tboil/t)- &
         1.3816e-7_r8*(10._r8**(11.344_r8*(1._r8-t/tboil))-1._r8)+ &
         8.1328e-3_r8*(10._r8**(-3.49149_r8*(tboil/t-1._r8))-1._r8)+ &
         log10(1013.246_r8))*100._r8
  end function svp_water

  elemental function svp_ice(t) result(es)        ! Pa
    real(r8), intent(in) :: t
    real(r8) :: es
    es = 10._r8**(-9.09718_r8*(h2otrip/t-1._r8)-3.56654_r8* &
         log10(h2otrip/t)+0.876793_r8*(1._r8-t/h2otrip)+ &
         log10(6.1071_r8))*100._r8
  end function svp_ice

  elemental function svp_trans(t) result(es)
    ! water above tmelt, ice below tmelt-ttrice, linear blend in between
    real(r8), intent(in) :: t
    real(r8) :: es, esice, weight
    es = 0._r8
    if (t >= (tmelt - ttrice)) es = svp_water(t)
    if (t < tmelt) then
      esice = svp_ice(t)
      if ( (tmelt - t) > ttrice ) then
        weight = 1.0_r8
      else
        weight = (tmelt - t)/ttrice
      end if
      es = weight*esice + (1.0_r8 - weight)*es
    end if
  end function svp_trans

  subroutine wv_sat_table_init()
    integer :: i
    do i = 1, plenest
      estbl(i) = svp_trans(tmin_tbl + real(i-1,r8))
    end do
    table_ready = .true.
  end subroutine wv_sat_table_init

  elemental function svp_to_qsat(es, p) result(qs)
    real(r8), intent(in) :: es, p
    real(r8) :: qs
    if ( (p - es) <= 0._r8 ) then
      qs = 1.0_r8
    else
      qs = epsilo*es / (p - (1._r8-epsilo)*es)
    end if
  end function svp_to_qsat

  function estblf(t) result(es)
    real(r8), intent(in) :: t
    real(r8) :: es, t_tmp, weight
    integer :: i
    t_tmp = max(min(t,tmax_tbl)-tmin_tbl, 0._r8)
    i = int(t_tmp) + 1
    if (i < plenest) then
      weight = min(t_tmp - aint(t_tmp), 1._r8)
      es = (1._r8 - weight)*estbl(i) + weight*estbl(i+1)
    else
      es = estbl(plenest)
    end if
  end function estblf

  subroutine qsat_water_scalar(t, p, es, qs)
    real(r8), intent(in)  :: t, p      ! K, Pa
    real(r8), intent(out) :: es, qs
    es = svp_water(t)
    qs = svp_to_qsat(es, p)
    es = min(es, p)
  end subroutine qsat_water_scalar

  subroutine qsat(t, p, es, qs, vlen)
    ! table look-up with water<->ice transition
    integer,  intent(in)  :: vlen
    real(r8), intent(in)  :: t(vlen), p(vlen)
    real(r8), intent(out) :: es(vlen), qs(vlen)
    integer :: i
    if (.not. table_ready) call wv_sat_table_init()
    do i = 1, vlen
      es(i) = estblf(t(i))
      qs(i) = svp_to_qsat(es(i), p(i))
      es(i) = min(es(i), p(i))
    end do
  end subroutine qsat

end module wv_saturation

module cloud_fraction
  use shr_kind_mod, only: r8 => shr_kind_r8
  use ppgrid,       only: pcols, pver
  implicit none
contains
  subroutine cldfrc_fice(ncol, t, fice, fsnow)
    use physconst, only: tmelt
    integer,  intent(in)  :: ncol
    real(r8), intent(in)  :: t(pcols,pver)
    real(r8), intent(out) :: fice(pcols,pver), fsnow(pcols,pver)
    real(r8) :: tmax_fice, tmin_fice, tmax_fsnow, tmin_fsnow
    integer :: i, k
    tmax_fice  = tmelt - 10._r8
    tmin_fice  = tmax_fice - 30._r8
    tmax_fsnow = tmelt
    tmin_fsnow = tmelt - 5._r8
    do k = 1, pver
      do i = 1, ncol
        if (t(i,k) > tmax_fice) then
          fice(i,k) = 0.0_r8
        else if (t(i,k) < tmin_fice) then
          fice(i,k) = 1.0_r8
        else
          fice(i,k) = (tmax_fice - t(i,k)) / (tmax_fice - tmin_fice)
        end if
        if (t(i,k) > tmax_fsnow) then
          fsnow(i,k) = 0.0_r8
        else if (t(i,k) < tmin_fsnow) then
          fsnow(i,k) = 1.0_r8
        else
          fsnow(i,k) = (tmax_fsnow - t(i,k)) / (tmax_fsnow - tmin_fsnow)
        end if
      end do
    end do
  end subroutine cldfrc_fice
end module cloud_fraction

module table_interp_mod
  use shr_kind_mod, only: r8 => shr_kind_r8
  implicit none
  private
  public :: table_interp_wghts, table_interp_calcwghts, table_interp

  type :: table_interp_wghts
    real(r8) :: wt1, wt2
    integer  :: ix1, ix2
  end type table_interp_wghts

  interface table_interp
    module procedure interp2d_mcoef
  end interface table_interp

contains

  pure function table_interp_calcwghts(ngrid, xgrid, ncols, xcols) result(w)
    ! first grid point strictly above x -> upper neighbour; linear extrapolation below the table,
    ! clamped to the last point above it
    integer,  intent(in) :: ngrid, ncols
    real(r8), intent(in) :: xgrid(ngrid), xcols(ncols)
    type(table_interp_wghts) :: w(ncols)
    integer  :: i, ii
    real(r8) :: dx, t
    do i = 1, ncols
      do ii = 1, ngrid
        if (xcols(i) < xgrid(ii)) exit
      end do
      w(i)%ix1 = max(ii-1, 1)
      w(i)%ix2 = min(w(i)%ix1+1, ngrid)
      dx = xgrid(w(i)%ix2) - xgrid(w(i)%ix1)
      if (abs(dx) > 1.e-20_r8) then
        t = (xcols(i) - xgrid(w(i)%ix1))/dx
      else
        t = 0._r8
      end if
      w(i)%wt1 = 1._r8 - t
      w(i)%wt2 = t
    end do
  end function table_interp_calcwghts

  pure function interp2d_mcoef(ncoef, ncol, nxs, nys, xw, yw, tbl) result(res)
    integer,  intent(in) :: ncoef, ncol, nxs, nys
    type(table_interp_wghts), intent(in) :: xw(ncol), yw(ncol)
    real(r8), intent(in) :: tbl(ncoef,nxs,nys)
    real(r8) :: res(ncoef,ncol)
    real(r8) :: fx1, fx2
    integer  :: i, c
    do i = 1, ncol
      do c = 1, ncoef
        fx1 = xw(i)%wt1*tbl(c,xw(i)%ix1,yw(i)%ix1) + xw(i)%wt2*tbl(c,xw(i)%ix2,yw(i)%ix1)
        fx2 = xw(i)%wt1*tbl(c,xw(i)%ix1,yw(i)%ix2) + xw(i)%wt2*tbl(c,xw(i)%ix2,yw(i)%ix2)
        res(c,i) = yw(i)%wt1*fx1 + yw(i)%wt2*fx2
      end do
    end do
  end function interp2d_mcoef

end module table_interp_mod
