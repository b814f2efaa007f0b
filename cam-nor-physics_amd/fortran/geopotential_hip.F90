! geopotential_hip -- ISO_C_BINDING shim of geopotential_t and geopotential_dse, bodies on the MI355X.
!
! Same names, argument lists (assumed-shape dummies), kinds and intents as the reference's
! physics/geopotential.F90:153-156 and :29-32, so geopotential_t's three callers -- physics_update (physics_types.F90:436),
! physics_dme_adjust_THT (physics_types.F90:1780), d_p_coupling (fv/dp_coupling.F90:617) -- compile unchanged.
! The branch the reference picks with dycore_is() / air_composition is picked the same way here and handed
! to the C ABI as dycore_kind / nspecies (include/colphys.h).
!
! Module name: `geopotential_hip` by default; `geopotential` with -DCOLPHYS_DROPIN_NAMES (see zm_conv_hip.F90).
! geopotential_dse (geopotential.F90:29) has no caller inside the reference (physics_types.F90:9 imports it); it is
! offloaded all the same so that the module is a whole replacement of `geopotential`.
#ifdef COLPHYS_DROPIN_NAMES
#define GEO_MODULE geopotential
#else
#define GEO_MODULE geopotential_hip
#endif

module GEO_MODULE
  use iso_c_binding
  use shr_kind_mod,    only: r8 => shr_kind_r8
  use ppgrid,          only: pver, pverp, pcols
  use dycore,          only: dycore_is
  use air_composition, only: thermodynamic_active_species_num, thermodynamic_active_species_idx
  use cam_abortutils,  only: endrun
  use colphys_c_api
  implicit none
  private
  save

  public geopotential_t
  public geopotential_dse

contains

  subroutine geopotential_t(                                 &
       piln   , pmln   , pint   , pmid   , pdel   , rpdel  , &
       t      , q      , rair   , gravit , zvir   ,          &
       zi     , zm     , ncol   )
    integer, intent(in) :: ncol
    real(r8), intent(in) :: piln (:,:)
    real(r8), intent(in) :: pmln (:,:)
    real(r8), intent(in) :: pint (:,:)
    real(r8), intent(in) :: pmid (:,:)
    real(r8), intent(in) :: pdel (:,:)
    real(r8), intent(in) :: rpdel(:,:)
    real(r8), intent(in) :: t    (:,:)
    real(r8), intent(in) :: q    (:,:,:)
    real(r8), intent(in) :: rair (:,:)
    real(r8), intent(in) :: gravit
    real(r8), intent(in) :: zvir (:,:)
    real(r8), intent(out) :: zi(:,:)
    real(r8), intent(out) :: zm(:,:)

    ! contiguous (np,nlev) copies: assumed-shape actuals may be strided sections
    integer :: np, kind, nsp, idx, rc
    integer(c_int32_t) :: ncol_c(1)
    real(r8), allocatable, target :: a_piln(:,:), a_pint(:,:), a_pmid(:,:), a_pdel(:,:), a_rpdel(:,:), a_t(:,:), &
                                     a_q(:,:,:), a_rair(:,:), a_zvir(:,:), a_zi(:,:), a_zm(:,:)

    np = size(t, 1)
    if (dycore_is('MPAS') .or. dycore_is('SE')) then                    ! geopotential.F90:217
       kind = COLPHYS_DYCORE_SE
       nsp = thermodynamic_active_species_num
       if (thermodynamic_active_species_idx(1) /= 1) &
          call endrun('geopotential_t (colphys): water vapour must be the first thermodynamically active species')
    else if (dycore_is('LR') .or. dycore_is('FV3')) then                ! :222
       kind = COLPHYS_DYCORE_FV
       nsp = 1
    else
       kind = COLPHYS_DYCORE_EUL
       nsp = 1
    end if
    allocate(a_t(np,pver), a_q(np,pver,nsp), a_rair(np,pver), a_zvir(np,pver), a_zi(np,pverp), a_zm(np,pver))
    allocate(a_piln(np,pverp), a_pint(np,pverp), a_pmid(np,pver), a_pdel(np,pver), a_rpdel(np,pver))
    a_piln = piln(:np,:pverp); a_pint = pint(:np,:pverp); a_pmid = pmid(:np,:pver); a_pdel = pdel(:np,:pver)
    a_rpdel = rpdel(:np,:pver); a_t = t(:np,:pver); a_rair = rair(:np,:pver); a_zvir = zvir(:np,:pver)
    if (kind == COLPHYS_DYCORE_SE) then
       do idx = 1, nsp
          a_q(:,:,idx) = q(:np,:pver,thermodynamic_active_species_idx(idx))
       end do
    else
       a_q(:,:,1) = q(:np,:pver,1)             ! the reference reads q(i,k,1) whatever the active-species list holds (:107, :241)
    end if
    ncol_c(1) = ncol
    rc = colphys_geopotential_t_chunks(1, np, pver, ncol_c, kind, nsp, nsp, c_loc(a_piln), c_null_ptr, c_loc(a_pint), &
                                       c_loc(a_pmid), c_loc(a_pdel), c_loc(a_rpdel), c_loc(a_t), c_loc(a_q), &
                                       c_loc(a_rair), gravit, c_loc(a_zvir), c_loc(a_zi), c_loc(a_zm))
    if (rc /= COLPHYS_OK) call endrun('geopotential_t (colphys): '//trim(colphys_error_message()))
    zi(:ncol,:pverp) = a_zi(:ncol,:)
    zm(:ncol,:pver)  = a_zm(:ncol,:)
  end subroutine geopotential_t

  subroutine geopotential_dse(                                &
       piln   , pmln   , pint   , pmid   , pdel   , rpdel  ,  &
       dse    , q      , phis   , rair   , gravit , cpair  ,  &
       zvir   , t      , zi     , zm     , ncol             )
    ! geopotential.F90:29-32
    integer, intent(in) :: ncol
    real(r8), intent(in) :: piln (:,:), pmln (:,:), pint (:,:), pmid (:,:), pdel (:,:), rpdel(:,:), dse  (:,:)
    real(r8), intent(in) :: q    (:,:,:), phis (:), rair (:,:), gravit, cpair(:,:), zvir (:,:)
    real(r8), intent(out) :: t(:,:), zi(:,:), zm(:,:)

    integer :: np, kind, nsp, idx, rc
    integer(c_int32_t) :: ncol_c(1)
    real(r8), allocatable, target :: a_piln(:,:), a_pint(:,:), a_pmid(:,:), a_pdel(:,:), a_rpdel(:,:), a_dse(:,:), a_q(:,:,:), &
                                     a_phis(:), a_rair(:,:), a_cpair(:,:), a_zvir(:,:), a_t(:,:), a_zi(:,:), a_zm(:,:)

    np = size(dse, 1)
    if (dycore_is('MPAS') .or. dycore_is('SE')) then                    ! geopotential.F90:89
       kind = COLPHYS_DYCORE_SE
       nsp = thermodynamic_active_species_num
       if (thermodynamic_active_species_idx(1) /= 1) &
          call endrun('geopotential_dse (colphys): water vapour must be the first thermodynamically active species')
    else if (dycore_is('LR') .or. dycore_is('FV3')) then                ! :92
       kind = COLPHYS_DYCORE_FV
       nsp = 1
    else
       kind = COLPHYS_DYCORE_EUL
       nsp = 1
    end if
    allocate(a_dse(np,pver), a_q(np,pver,nsp), a_phis(np), a_rair(np,pver), a_cpair(np,pver), a_zvir(np,pver))
    allocate(a_t(np,pver), a_zi(np,pverp), a_zm(np,pver))
    allocate(a_piln(np,pverp), a_pint(np,pverp), a_pmid(np,pver), a_pdel(np,pver), a_rpdel(np,pver))
    a_piln = piln(:np,:pverp); a_pint = pint(:np,:pverp); a_pmid = pmid(:np,:pver); a_pdel = pdel(:np,:pver)
    a_rpdel = rpdel(:np,:pver); a_dse = dse(:np,:pver); a_rair = rair(:np,:pver); a_cpair = cpair(:np,:pver)
    a_zvir = zvir(:np,:pver); a_phis = phis(:np)
    if (kind == COLPHYS_DYCORE_SE) then
       do idx = 1, nsp
          a_q(:,:,idx) = q(:np,:pver,thermodynamic_active_species_idx(idx))
       end do
    else
       a_q(:,:,1) = q(:np,:pver,1)             ! the reference reads q(i,k,1) whatever the active-species list holds (:107, :241)
    end if
    ncol_c(1) = ncol
    rc = colphys_geopotential_dse_chunks(1, np, pver, ncol_c, kind, nsp, nsp, c_loc(a_piln), c_null_ptr, c_loc(a_pint), &
                                         c_loc(a_pmid), c_loc(a_pdel), c_loc(a_rpdel), c_loc(a_dse), c_loc(a_q), c_loc(a_phis), &
                                         c_loc(a_rair), gravit, c_loc(a_cpair), c_loc(a_zvir), c_loc(a_t), c_loc(a_zi), c_loc(a_zm))
    if (rc /= COLPHYS_OK) call endrun('geopotential_dse (colphys): '//trim(colphys_error_message()))
    t (:ncol,:pver)  = a_t (:ncol,:)
    zi(:ncol,:pverp) = a_zi(:ncol,:)
    zm(:ncol,:pver)  = a_zm(:ncol,:)
  end subroutine geopotential_dse

end module GEO_MODULE
