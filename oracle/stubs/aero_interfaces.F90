! TEST INFRASTRUCTURE -- not product code.
!
! Interface stand-ins for the parent-model modules that the reference's modal_aer_opt.F90 `use`s but that
! are NOT part of NorESMhub/CAM-Nor-physics (SURVEY.md Appendix A).  No hot-path arithmetic here: the
! accessor modules (rad_constituents, physics_buffer, pio, tropopause) hand back synthetic DATA that the test
! driver stored in module arrays through oracle/ref_wrap_aero.F90; everything else is a signature.
! (table_interp_mod, the one arithmetic seam of this file's client, lives in l0_arithmetic.F90.)

module string_utils
contains
  function int2str(n) result(s)
    integer, intent(in) :: n
    character(len=12) :: s
    write(s,'(i0)') n
  end function int2str
end module string_utils

module ref_pres
  integer :: clim_modal_aero_top_lev = 1
end module ref_pres

module radconstants
  use shr_kind_mod, only: r8 => shr_kind_r8
  integer, parameter :: nswbands = 14, nlwbands = 16
  integer, parameter :: idx_sw_diag = 10, idx_nir_diag = 8, idx_uv_diag = 11
contains
  subroutine get_lw_spectral_boundaries(lo, hi, units)
    real(r8), intent(out) :: lo(nlwbands), hi(nlwbands)
    character(len=*), intent(in) :: units
    integer :: i
    do i = 1, nlwbands           ! descending wavelength bands in um, RRTMG-like; band 7 brackets 10 um
      hi(i) = 1000._r8/real(10+60*(i-1), r8)*10._r8
      lo(i) = 1000._r8/real(10+60*i, r8)*10._r8
    end do
  end subroutine get_lw_spectral_boundaries
end module radconstants

#ifndef COLPHYS_REAL_PHYSICS_TYPES
! (the "intr" oracle library compiles the reference's own physics_types.F90 and the physics_buffer of host_interfaces.F90)
module physics_types
  ! the three components of physics_state (physics/physics_types.F90:62-121) modal_aer_opt reads
  use shr_kind_mod, only: r8 => shr_kind_r8
  type physics_state
    integer :: lchnk, ncol
    real(r8), allocatable :: t(:,:), pmid(:,:), pdeldry(:,:)
  end type physics_state
end module physics_types

module physics_buffer
  use shr_kind_mod, only: r8 => shr_kind_r8
  type physics_buffer_desc
    integer :: dummy
  end type physics_buffer_desc
  real(r8), allocatable, target, save :: dgnumwet_store(:,:,:), qaerwat_store(:,:,:)   ! (pcols,pver,nmodes)
  interface pbuf_get_field
    module procedure get3d
  end interface pbuf_get_field
contains
  integer function pbuf_get_index(name, errcode)
    character(len=*), intent(in) :: name
    integer, intent(out), optional :: errcode
    if (present(errcode)) errcode = 0
    pbuf_get_index = 1
    if (trim(name) == 'QAERWAT') pbuf_get_index = 2
  end function pbuf_get_index
  subroutine get3d(pbuf, idx, fld)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    real(r8), pointer :: fld(:,:,:)
    if (idx == 1) then
      fld => dgnumwet_store
    else
      fld => qaerwat_store
    end if
  end subroutine get3d
end module physics_buffer
#endif

module rad_constituents
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use physics_types,  only: physics_state
  use physics_buffer, only: physics_buffer_desc
  use radconstants,   only: nswbands, nlwbands
  integer, parameter :: n_diag = 10
  integer, save :: nm = 0, nsmax = 0
  integer,  allocatable, save :: nspec_m(:)
  real(r8), allocatable, target, save :: mmr_store(:,:,:,:)                 ! (pcols,pver,nsmax,nm)
  real(r8), allocatable, save :: dens(:,:), hygro(:,:), sigmag_m(:)         ! (nsmax,nm)
  complex(r8), allocatable, target, save :: ri_sw(:,:,:), ri_lw(:,:,:)      ! (nbands,nsmax,nm)
  character(len=32), allocatable, save :: stype(:,:)
  real(r8), allocatable, target, save :: refrsw(:,:,:), refisw(:,:,:), refrlw(:,:,:), refilw(:,:,:)
  real(r8), allocatable, target, save :: ext(:,:,:,:,:), abss(:,:,:,:,:), asmp(:,:,:,:,:), abl(:,:,:,:,:)
  interface rad_cnst_get_info
    module procedure info_list, info_mode, info_spec
  end interface rad_cnst_get_info
  interface rad_cnst_get_aer_mmr
    module procedure mmr_mode, mmr_bulk
  end interface rad_cnst_get_aer_mmr
  interface rad_cnst_get_aer_props
    module procedure props_mode, props_bulk
  end interface rad_cnst_get_aer_props
contains
  subroutine rad_cnst_get_call_list(call_list)
    logical, intent(out) :: call_list(0:n_diag)
    call_list = .false.; call_list(0) = .true.
  end subroutine rad_cnst_get_call_list
  subroutine info_list(list_idx, nmodes, naero)
    integer, intent(in) :: list_idx
    integer, intent(out), optional :: nmodes, naero
    if (present(nmodes)) nmodes = nm
    if (present(naero)) naero = 0
  end subroutine info_list
  subroutine info_mode(list_idx, m, nspec, mode_type)
    integer, intent(in) :: list_idx, m
    integer, intent(out), optional :: nspec
    character(len=*), intent(out), optional :: mode_type
    if (present(nspec)) nspec = nspec_m(m)
    if (present(mode_type)) write(mode_type,'(a,i1)') 'mode', m
  end subroutine info_mode
  subroutine info_spec(list_idx, m, l, spec_type)
    integer, intent(in) :: list_idx, m, l
    character(len=*), intent(out), optional :: spec_type
    if (present(spec_type)) spec_type = stype(l,m)
  end subroutine info_spec
  subroutine mmr_mode(list_idx, m, l, phase, state, pbuf, mmr)
    integer, intent(in) :: list_idx, m, l
    character(len=1), intent(in) :: phase
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8), pointer :: mmr(:,:)
    mmr => mmr_store(:,:,l,m)
  end subroutine mmr_mode
  ! bulk-aerosol and mode-number accessors: named by zm_conv_intr.F90:15-16 and used only under zmconv_microp (out of scope)
  subroutine mmr_bulk(list_idx, m, state, pbuf, mmr)
    integer, intent(in) :: list_idx, m
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8), pointer :: mmr(:,:)
    nullify(mmr)
  end subroutine mmr_bulk
  subroutine props_bulk(list_idx, iaer, aername, num_to_mass_aer)
    integer, intent(in) :: list_idx, iaer
    character(len=*), intent(out), optional :: aername
    real(r8), intent(out), optional :: num_to_mass_aer
    if (present(aername)) aername = ' '
    if (present(num_to_mass_aer)) num_to_mass_aer = 0._r8
  end subroutine props_bulk
  subroutine rad_cnst_get_mode_num(list_idx, m, phase, state, pbuf, num)
    integer, intent(in) :: list_idx, m
    character(len=1), intent(in) :: phase
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8), pointer :: num(:,:)
    nullify(num)
  end subroutine rad_cnst_get_mode_num
  subroutine props_mode(list_idx, m, l, density_aer, refindex_aer_sw, refindex_aer_lw, spectype, hygro_aer)
    integer, intent(in) :: list_idx, m, l
    real(r8), intent(out), optional :: density_aer, hygro_aer
    complex(r8), pointer, optional :: refindex_aer_sw(:), refindex_aer_lw(:)
    character(len=32), intent(out), optional :: spectype
    if (present(density_aer)) density_aer = dens(l,m)
    if (present(hygro_aer)) hygro_aer = hygro(l,m)
    if (present(refindex_aer_sw)) refindex_aer_sw => ri_sw(:,l,m)
    if (present(refindex_aer_lw)) refindex_aer_lw => ri_lw(:,l,m)
    if (present(spectype)) spectype = stype(l,m)
  end subroutine props_mode
  subroutine rad_cnst_get_mode_props(list_idx, m, sigmag, refrtabsw, refitabsw, extpsw, abspsw, asmpsw, &
                                     refrtablw, refitablw, absplw, ncoef, prefr, prefi, dgnumlo, dgnumhi)
    integer, intent(in) :: list_idx, m
    real(r8), intent(out), optional :: sigmag, dgnumlo, dgnumhi
    real(r8), pointer, optional :: refrtabsw(:,:), refitabsw(:,:), refrtablw(:,:), refitablw(:,:)
    real(r8), pointer, optional :: extpsw(:,:,:,:), abspsw(:,:,:,:), asmpsw(:,:,:,:), absplw(:,:,:,:)
    integer, intent(out), optional :: ncoef, prefr, prefi
    if (present(sigmag)) sigmag = sigmag_m(m)
    if (present(refrtabsw)) refrtabsw => refrsw(:,:,m)
    if (present(refitabsw)) refitabsw => refisw(:,:,m)
    if (present(refrtablw)) refrtablw => refrlw(:,:,m)
    if (present(refitablw)) refitablw => refilw(:,:,m)
    if (present(extpsw)) extpsw => ext(:,:,:,:,m)
    if (present(abspsw)) abspsw => abss(:,:,:,:,m)
    if (present(asmpsw)) asmpsw => asmp(:,:,:,:,m)
    if (present(absplw)) absplw => abl(:,:,:,:,m)
    if (present(ncoef)) ncoef = 5
    if (present(prefr)) prefr = 7
    if (present(prefi)) prefi = 10
    if (present(dgnumlo)) dgnumlo = 0._r8
    if (present(dgnumhi)) dgnumhi = 1._r8
  end subroutine rad_cnst_get_mode_props
end module rad_constituents

module pio
  ! serves the water refractive index that read_water_refindex (modal_aer_opt.F90:1475-1534) asks for
  use shr_kind_mod, only: r8 => shr_kind_r8
  use radconstants, only: nswbands, nlwbands
  type file_desc_t
    integer :: i = 0
  end type file_desc_t
  type var_desc_t
    integer :: which = 0
  end type var_desc_t
  integer, parameter :: pio_nowrite = 0
  real(r8), save :: wat_r_sw(nswbands), wat_i_sw(nswbands), wat_r_lw(nlwbands), wat_i_lw(nlwbands)
contains
  integer function pio_inq_dimid(f, name, did)
    type(file_desc_t) :: f
    character(len=*) :: name
    integer :: did
    did = 1
    if (trim(name) == 'sw_band') did = 2
    pio_inq_dimid = 0
  end function pio_inq_dimid
  integer function pio_inq_dimlen(f, did, dimlen)
    type(file_desc_t) :: f
    integer :: did, dimlen
    dimlen = nlwbands
    if (did == 2) dimlen = nswbands
    pio_inq_dimlen = 0
  end function pio_inq_dimlen
  integer function pio_inq_varid(f, name, vid)
    type(file_desc_t) :: f
    character(len=*) :: name
    type(var_desc_t) :: vid
    select case (trim(name))
    case ('refindex_real_water_sw'); vid%which = 1
    case ('refindex_im_water_sw');   vid%which = 2
    case ('refindex_real_water_lw'); vid%which = 3
    case ('refindex_im_water_lw');   vid%which = 4
    end select
    pio_inq_varid = 0
  end function pio_inq_varid
  integer function pio_get_var(f, vid, x)
    type(file_desc_t) :: f
    type(var_desc_t) :: vid
    real(r8) :: x(:)
    select case (vid%which)
    case (1); x = wat_r_sw
    case (2); x = wat_i_sw
    case (3); x = wat_r_lw
    case (4); x = wat_i_lw
    end select
    pio_get_var = 0
  end function pio_get_var
  subroutine pio_closefile(f)
    type(file_desc_t) :: f
  end subroutine pio_closefile
end module pio

module cam_pio_utils
  use pio, only: file_desc_t
contains
  subroutine cam_pio_openfile(f, name, mode)
    type(file_desc_t) :: f
    character(len=*) :: name
    integer :: mode
  end subroutine cam_pio_openfile
end module cam_pio_utils

module cam_history_support
  use shr_kind_mod, only: r8 => shr_kind_r8
  real(r8), parameter :: fillvalue = 1.e36_r8
end module cam_history_support

module perf_mod
contains
  subroutine t_startf(n)
    character(len=*) :: n
  end subroutine t_startf
  subroutine t_stopf(n)
    character(len=*) :: n
  end subroutine t_stopf
end module perf_mod

module modal_aero_wateruptake
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use physics_types,  only: physics_state
  use physics_buffer, only: physics_buffer_desc
contains
  subroutine modal_aero_wateruptake_dr(state, pbuf, list_idx, dgnumdry_m, dgnumwet_m, qaerwat_m, wetdens_m, hygro_m, &
                                       dryvol_m, dryrad_m, drymass_m, so4dryvol_m, naer_m)
    type(physics_state), target, intent(in) :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, optional, intent(in) :: list_idx
    real(r8), optional, pointer, dimension(:,:,:) :: dgnumdry_m, dgnumwet_m, qaerwat_m, wetdens_m, hygro_m, dryvol_m, &
                                                    dryrad_m, drymass_m, so4dryvol_m, naer_m
    stop 'modal_aero_wateruptake_dr: diagnostic lists (list_idx > 0) are outside the hot-path scope'
  end subroutine modal_aero_wateruptake_dr
end module modal_aero_wateruptake

module modal_aero_calcsize
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use physics_types,  only: physics_state
  use physics_buffer, only: physics_buffer_desc
contains
  subroutine modal_aero_calcsize_diag(state, pbuf, list_idx, dgnum_m, hygro_m, dryvol_m, dryrad_m, drymass_m, so4dryvol_m, naer_m)
    type(physics_state), target, intent(in) :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, optional, intent(in) :: list_idx
    real(r8), optional, pointer, dimension(:,:,:) :: dgnum_m, hygro_m, dryvol_m, dryrad_m, drymass_m, so4dryvol_m, naer_m
    stop 'modal_aero_calcsize_diag: diagnostic lists (list_idx > 0) are outside the hot-path scope'
  end subroutine modal_aero_calcsize_diag
end module modal_aero_calcsize

module namelist_utils
contains
  ! positions unit u at the line that opens namelist group g (CAM's find_group_name); status 0 = found
  subroutine find_group_name(u, g, status)
    integer :: u, status
    character(len=*) :: g
    character(len=256) :: line
    integer :: ios
    status = 1
    rewind(u, iostat=ios)
    if (ios /= 0) return
    do
      read(u, '(a)', iostat=ios) line
      if (ios /= 0) return
      line = adjustl(line)
      if (line(1:1) == '&' .and. trim(line(2:len_trim(g)+1)) == trim(g)) then
        backspace(u)
        status = 0
        return
      end if
    end do
  end subroutine find_group_name
end module namelist_utils

module units
contains
  integer function getunit()
    getunit = 99
  end function getunit
  subroutine freeunit(u)
    integer :: u
  end subroutine freeunit
end module units

module mpishorthand
  integer :: mpichar = 0, mpicom = 0
end module mpishorthand

module iofilemod
contains
  subroutine getfil(a, b)
    character(len=*) :: a, b
    b = a
  end subroutine getfil
end module iofilemod

module tropopause
  use physics_types, only: physics_state
  use ppgrid,        only: pcols
  integer, save :: troplev_store(pcols) = 1
contains
  subroutine tropopause_findChemTrop(state, troplev)
    type(physics_state), intent(in) :: state
    integer, intent(out) :: troplev(pcols)
    troplev = troplev_store
  end subroutine tropopause_findChemTrop
end module tropopause
