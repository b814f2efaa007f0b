! TEST INFRASTRUCTURE -- not product code.
!
! Stand-ins for the parent-model (ESCOMP/CAM, CESM share) modules that the reference's OPERATOR layer `use`s --
! physics/physics_types.F90, physics/zm_conv_intr.F90, physics/convect_diagnostics.F90 -- and that are not part of
! NorESMhub/CAM-Nor-physics, beyond what cam_interfaces.F90 already supplies.  Compiled only into the "intr" oracle library
! (oracle/Makefile, target ref_intr) and its shim twin (cam-nor-physics_amd/fortran/Makefile, target intr).
!
! Interface-only, with ONE exception that carries arithmetic and is therefore a fifth parity-unpinned L0 seam (DESIGN.md 2):
!   qneg_module:qneg3 -- CAM's negative-constituent clip: q < qmin  ->  q = qmin (plus a log message, omitted here).
! Everything else is kinds, switches, registries and data holders:
!   physics_buffer is a small working implementation (named fields, one flat store per field and chunk, pointer views by
!   rank) because zm_conv_tend / convect_diagnostics_calc fetch every in/out array through it.

module shr_const_mod
  use shr_kind_mod, only: r8 => shr_kind_r8
  real(r8), parameter :: shr_const_boltz = 1.38065e-23_r8, shr_const_avogad = 6.02214e26_r8, shr_const_mwwv = 18.016_r8
  real(r8), parameter :: shr_const_rwv = shr_const_avogad*shr_const_boltz/shr_const_mwwv
end module shr_const_mod

module shr_infnan_mod
  use shr_kind_mod, only: r8 => shr_kind_r8
  use, intrinsic :: ieee_arithmetic
  type shr_infnan_inf_type
    logical :: positive = .true.
  end type shr_infnan_inf_type
  type(shr_infnan_inf_type), parameter :: shr_infnan_posinf = shr_infnan_inf_type(.true.)
  type(shr_infnan_inf_type), parameter :: shr_infnan_neginf = shr_infnan_inf_type(.false.)
  interface assignment(=)
    module procedure set_inf_r8
  end interface
contains
  elemental subroutine set_inf_r8(x, v)
    real(r8), intent(out) :: x
    type(shr_infnan_inf_type), intent(in) :: v
    if (v%positive) then
      x = ieee_value(x, ieee_positive_inf)
    else
      x = ieee_value(x, ieee_negative_inf)
    end if
  end subroutine set_inf_r8
end module shr_infnan_mod

module infnan
  ! CAM's older spelling of the same thing: `state%t(:,:) = inf` (physics_types.F90:2143-2178)
  use shr_kind_mod, only: r8 => shr_kind_r8
  use, intrinsic :: ieee_arithmetic
  type inf_t
    integer :: dummy = 0
  end type inf_t
  type(inf_t), parameter :: inf = inf_t(0)
  interface assignment(=)
    module procedure inf_to_r8
  end interface
contains
  elemental subroutine inf_to_r8(x, v)
    real(r8), intent(out) :: x
    type(inf_t), intent(in) :: v
    x = ieee_value(x, ieee_positive_inf)
  end subroutine inf_to_r8
end module infnan

module shr_assert_mod
  ! shr_assert_in_domain (physics_state_check): the checks physics_types.F90:547-691 asks for, abort through endrun
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use cam_abortutils, only: endrun
  use, intrinsic :: ieee_arithmetic
  interface shr_assert_in_domain
    module procedure dom1, dom2, dom3
  end interface
contains
  subroutine dom_flat(n, v, varname, msg, is_nan, lt, gt)
    integer, intent(in) :: n
    real(r8), intent(in) :: v(n)
    character(len=*), intent(in), optional :: varname, msg
    logical, intent(in), optional :: is_nan
    real(r8), intent(in), optional :: lt, gt
    logical :: bad
    bad = .false.
    if (present(is_nan)) then
      if (.not. is_nan) bad = bad .or. any(ieee_is_nan(v))
    end if
    if (present(lt)) bad = bad .or. any(.not. (v < lt))
    if (present(gt)) bad = bad .or. any(.not. (v > gt))
    if (bad) then
      if (present(varname) .and. present(msg)) then
        call endrun(trim(msg)//' ['//trim(varname)//']')
      else
        call endrun('shr_assert_in_domain failed')
      end if
    end if
  end subroutine dom_flat
  subroutine dom1(v, varname, msg, is_nan, lt, gt)
    real(r8), intent(in) :: v(:)
    character(len=*), intent(in), optional :: varname, msg
    logical, intent(in), optional :: is_nan
    real(r8), intent(in), optional :: lt, gt
    call dom_flat(size(v), v, varname, msg, is_nan, lt, gt)
  end subroutine dom1
  subroutine dom2(v, varname, msg, is_nan, lt, gt)
    real(r8), intent(in) :: v(:,:)
    character(len=*), intent(in), optional :: varname, msg
    logical, intent(in), optional :: is_nan
    real(r8), intent(in), optional :: lt, gt
    call dom_flat(size(v), reshape(v, (/size(v)/)), varname, msg, is_nan, lt, gt)
  end subroutine dom2
  subroutine dom3(v, varname, msg, is_nan, lt, gt)
    real(r8), intent(in) :: v(:,:,:)
    character(len=*), intent(in), optional :: varname, msg
    logical, intent(in), optional :: is_nan
    real(r8), intent(in), optional :: lt, gt
    call dom_flat(size(v), reshape(v, (/size(v)/)), varname, msg, is_nan, lt, gt)
  end subroutine dom3
end module shr_assert_mod

module scammod
  logical :: single_column = .false., scm_crm_mode = .false.
end module scammod

module hycoef
  ! hybrid coefficients: only physics_dme_adjust_THT reads them (out of scope; never called by the oracle drivers)
  use shr_kind_mod, only: r8 => shr_kind_r8
  use ppgrid,       only: pver, pverp
  real(r8) :: hyai(pverp) = 0._r8, hybi(pverp) = 0._r8, hyam(pver) = 0._r8, hybm(pver) = 0._r8, ps0 = 1.e5_r8
end module hycoef

module cam_thermo
  use shr_kind_mod, only: r8 => shr_kind_r8
contains
  subroutine cam_thermo_dry_air_update(mmr, t, lchnk, ncol, to_dry_factor)
    real(r8), intent(in) :: mmr(:,:,:), t(:,:)
    integer, intent(in) :: lchnk, ncol
    real(r8), intent(in), optional :: to_dry_factor(:,:)
    ! WACCM-X only (variable composition): nothing to update with constant dry air
  end subroutine cam_thermo_dry_air_update
end module cam_thermo

module qneg_module
  ! *** arithmetic *** CAM's qneg3: every value below the constituent's minimum is set to that minimum
  ! (upstream also counts them and writes a log line; neither affects the state)
  use shr_kind_mod, only: r8 => shr_kind_r8
contains
  subroutine qneg3(subnam, idx, ncol, ncold, lver, lconst_beg, lconst_end, qmin, q)
    character(len=*), intent(in) :: subnam
    integer, intent(in) :: idx, ncol, ncold, lver, lconst_beg, lconst_end
    real(r8), intent(in) :: qmin(lconst_beg:lconst_end)
    real(r8), intent(inout) :: q(ncold, lver, lconst_beg:lconst_end)
    integer :: i, k, m
    do m = lconst_beg, lconst_end
      do k = 1, lver
        do i = 1, ncol
          if (q(i,k,m) < qmin(m)) q(i,k,m) = qmin(m)
        end do
      end do
    end do
  end subroutine qneg3
end module qneg_module

module pmgrid
  use ppgrid, only: pver, pverp
  integer, parameter :: plev = pver, plevp = pverp
end module pmgrid

module time_manager
  integer :: nstep_now = 1
contains
  integer function get_nstep()
    get_nstep = nstep_now
  end function get_nstep
  logical function is_first_step()
    is_first_step = (nstep_now == 0)
  end function is_first_step
end module time_manager

module ndrop_bam
contains
  subroutine ndrop_bam_init()
  end subroutine ndrop_bam_init
end module ndrop_bam

! -----------------------------------------------------------------------------------------------------------------
! physics_buffer: named fields with per-chunk storage.  A chunk's buffer is pbuf(:) -- one descriptor per registered
! field, each owning a flat store; pbuf_get_field hands out pointer views of the rank the caller's pointer has, whole
! field or the (start, kount) box CAM's interface allows (used for the time level of CLD and for FRACIS).
! -----------------------------------------------------------------------------------------------------------------
module physics_buffer
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use cam_abortutils, only: endrun
  implicit none
  integer, parameter :: dtype_r8 = 1, dtype_i4 = 2
  integer, parameter :: max_fields = 96
  integer, parameter :: dyn_time_lvls = 2
  type physics_buffer_desc
    integer :: dtype = 0, ndims = 0, dims(4) = 1
    real(r8), pointer :: r(:) => null()
    integer,  pointer :: i(:) => null()
  end type physics_buffer_desc
  integer, save :: nfields = 0
  character(len=32), save :: fname(max_fields)
  integer, save :: fdtype(max_fields), fndims(max_fields), fdims(4, max_fields)
  integer, save :: old_tim_idx = 1
  interface pbuf_get_field
    module procedure get_r1, get_r2, get_r3, get_i1
  end interface pbuf_get_field
  interface pbuf_set_field
    module procedure set_const_2d, set_r2
  end interface pbuf_set_field
contains
  subroutine pbuf_add_field(name, scope, dtype, dims, idx)
    character(len=*), intent(in) :: name, scope
    integer, intent(in) :: dtype, dims(:)
    integer, intent(out) :: idx
    integer :: j
    do j = 1, nfields                      ! registering twice returns the same slot (drivers may re-initialise)
      if (trim(fname(j)) == trim(name)) then
        idx = j
        return
      end if
    end do
    if (nfields >= max_fields) call endrun('pbuf_add_field: registry full')
    nfields = nfields + 1
    idx = nfields
    fname(idx) = name; fdtype(idx) = dtype; fndims(idx) = size(dims)
    fdims(:, idx) = 1
    fdims(1:size(dims), idx) = dims
  end subroutine pbuf_add_field
  integer function pbuf_get_index(name, errcode)
    character(len=*), intent(in) :: name
    integer, intent(out), optional :: errcode
    integer :: j
    pbuf_get_index = -1
    do j = 1, nfields
      if (trim(fname(j)) == trim(name)) pbuf_get_index = j
    end do
    if (present(errcode)) then
      errcode = merge(0, 1, pbuf_get_index > 0)
    else if (pbuf_get_index < 0) then
      call endrun('pbuf_get_index: no field '//trim(name))
    end if
  end function pbuf_get_index
  integer function pbuf_old_tim_idx()
    pbuf_old_tim_idx = old_tim_idx
  end function pbuf_old_tim_idx
  ! allocates one chunk's buffer for every field registered so far (CAM: pbuf_initialize / pbuf_allocate)
  subroutine pbuf_allocate_chunk(pbuf)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer :: j, n
    allocate(pbuf(nfields))
    do j = 1, nfields
      pbuf(j)%dtype = fdtype(j); pbuf(j)%ndims = fndims(j); pbuf(j)%dims = fdims(:, j)
      n = product(fdims(:, j))
      if (fdtype(j) == dtype_r8) then
        allocate(pbuf(j)%r(n)); pbuf(j)%r = 0._r8
      else
        allocate(pbuf(j)%i(n)); pbuf(j)%i = 0
      end if
    end do
  end subroutine pbuf_allocate_chunk
  ! ... and the model's whole buffer pbuf2d(field, chunk), of which a chunk's pbuf(:) is a column
  subroutine pbuf_allocate_2d(pbuf2d, nchunks)
    type(physics_buffer_desc), pointer :: pbuf2d(:,:)
    integer, intent(in) :: nchunks
    integer :: j, c, n
    allocate(pbuf2d(nfields, nchunks))
    do c = 1, nchunks
      do j = 1, nfields
        pbuf2d(j,c)%dtype = fdtype(j); pbuf2d(j,c)%ndims = fndims(j); pbuf2d(j,c)%dims = fdims(:, j)
        n = product(fdims(:, j))
        if (fdtype(j) == dtype_r8) then
          allocate(pbuf2d(j,c)%r(n)); pbuf2d(j,c)%r = 0._r8
        else
          allocate(pbuf2d(j,c)%i(n)); pbuf2d(j,c)%i = 0
        end if
      end do
    end do
  end subroutine pbuf_allocate_2d
  subroutine pbuf_deallocate_chunk(pbuf)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer :: j
    do j = 1, size(pbuf)
      if (associated(pbuf(j)%r)) deallocate(pbuf(j)%r)
      if (associated(pbuf(j)%i)) deallocate(pbuf(j)%i)
    end do
    deallocate(pbuf)
  end subroutine pbuf_deallocate_chunk
  ! offset (0-based, in elements) and extents of the box (start, kount) inside field idx; the box must be contiguous in
  ! memory, which holds for the uses in the reference: whole leading dimensions, one index or a range of the last one
  subroutine box(p, start, kount, off, ext)
    type(physics_buffer_desc), intent(in) :: p
    integer, intent(in), optional :: start(:), kount(:)
    integer, intent(out) :: off, ext(4)
    integer :: d, stride
    off = 0; ext = p%dims
    if (present(start) .and. present(kount)) then
      stride = 1
      ext = 1
      do d = 1, size(start)
        off = off + (start(d) - 1)*stride
        ext(d) = kount(d)
        stride = stride*p%dims(d)
      end do
    end if
  end subroutine box
  subroutine get_r1(pbuf, idx, fld, start, kount)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    real(r8), pointer :: fld(:)
    integer, intent(in), optional :: start(:), kount(:)
    integer :: off, ext(4)
    call box(pbuf(idx), start, kount, off, ext)
    fld(1:ext(1)) => pbuf(idx)%r(off+1:off+ext(1))
  end subroutine get_r1
  subroutine get_r2(pbuf, idx, fld, start, kount)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    real(r8), pointer :: fld(:,:)
    integer, intent(in), optional :: start(:), kount(:)
    integer :: off, ext(4)
    call box(pbuf(idx), start, kount, off, ext)
    fld(1:ext(1), 1:ext(2)) => pbuf(idx)%r(off+1:off+ext(1)*ext(2))
  end subroutine get_r2
  subroutine get_r3(pbuf, idx, fld, start, kount)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    real(r8), pointer :: fld(:,:,:)
    integer, intent(in), optional :: start(:), kount(:)
    integer :: off, ext(4)
    call box(pbuf(idx), start, kount, off, ext)
    fld(1:ext(1), 1:ext(2), 1:ext(3)) => pbuf(idx)%r(off+1:off+ext(1)*ext(2)*ext(3))
  end subroutine get_r3
  subroutine get_i1(pbuf, idx, fld, start, kount)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    integer, pointer :: fld(:)
    integer, intent(in), optional :: start(:), kount(:)
    integer :: off, ext(4)
    call box(pbuf(idx), start, kount, off, ext)
    fld(1:ext(1)) => pbuf(idx)%i(off+1:off+ext(1))
  end subroutine get_i1
  ! pbuf_set_field(pbuf2d, idx, const): every chunk (convect_diagnostics_init)
  subroutine set_const_2d(pbuf2d, idx, val)
    type(physics_buffer_desc), pointer :: pbuf2d(:,:)
    integer, intent(in) :: idx
    real(r8), intent(in) :: val
    integer :: c
    if (.not. associated(pbuf2d)) return
    do c = lbound(pbuf2d, 2), ubound(pbuf2d, 2)
      pbuf2d(idx, c)%r = val
    end do
  end subroutine set_const_2d
  ! pbuf_set_field(pbuf, idx, array, start, kount): a box that is NOT contiguous when kount(1) < dims(1) (RPRDTOT, :ncol)
  subroutine set_r2(pbuf, idx, val, start, kount)
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer, intent(in) :: idx
    real(r8), intent(in) :: val(:,:)
    integer, intent(in), optional :: start(:), kount(:)
    integer :: i, k, s1, s2, n1, n2, ld
    s1 = 1; s2 = 1; n1 = size(val, 1); n2 = size(val, 2)
    if (present(start)) then
      s1 = start(1); s2 = start(2)
    end if
    if (present(kount)) then
      n1 = kount(1); n2 = kount(2)
    end if
    ld = pbuf(idx)%dims(1)
    do k = 1, n2
      do i = 1, n1
        pbuf(idx)%r((s2 + k - 2)*ld + s1 + i - 1) = val(i, k)
      end do
    end do
  end subroutine set_r2
end module physics_buffer

module check_energy
  ! zm_conv_intr.F90:406 imports the name; zm_conv_tend never calls it (the energy check of deep convection is tphysbc's,
  ! physpkg.F90:2867)
contains
  subroutine check_energy_chng()
  end subroutine check_energy_chng
end module check_energy
