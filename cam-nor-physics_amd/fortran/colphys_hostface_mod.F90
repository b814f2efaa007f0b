! colphys_hostface_mod -- the Fortran face of the whole-rank dispatch: physics_state in, physics_ptend and pbuf fields out.
!
! What replaces the chunk loop of phys_run1 for the deep-convection step (physpkg.F90:1147-1161 -> tphysbc :2820-2838 ->
! zm_conv_tend, zm_conv_intr.F90:390):
!
!     !$omp parallel do                                 ! the loop stays, its body shrinks to copies
!     do c = begchunk, endchunk
!        call colphys_rank_push(c - begchunk + 1, state(c), pblh, tpert, landfrac, cld)        ! state(c)%t ... -> chunk record
!     end do
!     call colphys_rank_run_zm_conv_tend(ztodt)          ! ONE dispatch for the rank: upload, device re-layout, every column's
!                                                        ! zm_conv_tend, re-layout back, download; the thread sleeps meanwhile
!     !$omp parallel do
!     do c = begchunk, endchunk
!        call colphys_rank_pull(c - begchunk + 1, state(c), ptend_all, ...pbuf fields...)      ! chunk record -> ptend_all, pbuf
!     end do
!
! The records are the C ABI's host face (include/colphys.h, colphys_hostface_*): per chunk, every field as the (pcols, nlev)
! array physics_state / physics_ptend / the physics buffer hold it, in pinned memory owned by the library.  Nothing is
! transposed or packed on the host; push and pull are whole-array copies, safe to run concurrently for distinct chunks.
! physics_state and physics_ptend are the host model's own types (physics_types.F90:62-173), unchanged.
module colphys_hostface_mod
  use iso_c_binding
  use shr_kind_mod,   only: r8 => shr_kind_r8
  use ppgrid,         only: pcols, pver, pverp
  use constituents,   only: pcnst
  use physics_types,  only: physics_state, physics_ptend, physics_ptend_init
  use cam_abortutils, only: endrun
  use cam_logfile,    only: iulog
  use colphys_c_api,  only: COLPHYS_HF_ZM, colphys_hostface_times, colphys_hostface_create, colphys_hostface_destroy, colphys_hostface_geometry, &
                            colphys_hostface_field, colphys_hostface_in_base, colphys_hostface_out_base, colphys_hostface_run, &
                            colphys_hostface_last_times, colphys_last_error
  implicit none
  private
  save

  public :: colphys_rank_init, colphys_rank_final, colphys_rank_push, colphys_rank_run_zm_conv_tend, colphys_rank_pull, &
            colphys_rank_last_times

  type(c_ptr) :: hf = c_null_ptr
  integer :: nchunks_rank = 0
  integer, allocatable :: ncol_rank(:)
  integer(c_intptr_t) :: in_base = 0, out_base = 0
  integer(c_int64_t)  :: in_rec = 0, out_rec = 0
  ! the fields this face moves, resolved to record offsets once (colphys_rank_init)
  integer, parameter :: n_in = 18, n_out = 33
  character(len=*), parameter :: in_names(n_in) = &
       [character(len=8) :: 't', 'q', 'u', 'v', 'pmid', 'pdel', 'rpdel', 'lnpmid', 'zm', 'cld', 'pint', 'lnpint', 'zi', 'phis', 'ps', 'pblh', 'tpert', 'landfrac']
  character(len=*), parameter :: out_names(n_out) = &
       [character(len=7) :: 'ptend_s', 'ptend_q', 'ptend_u', 'ptend_v', 'mconzm', 'flxprec', 'flxsnow', 'ql', 'rprd', 'evapcdp', 'dlf', &
       'prec', 'snow', 'rliq', 'rice', 'jctop', 'jcbot', 'cape', 'freqzm', 'pcont', 'pconb', 'lengath', 'mu', 'md', 'du', 'eu', 'ed', 'dp', 'dsubcld', 'jt', 'maxg', 'ideep', 'status']
  integer(c_int64_t) :: off_in(n_in) = -1, off_out(n_out) = -1
  integer, parameter :: i_t = 1, i_q = 2, i_u = 3, i_v = 4, i_pmid = 5, i_pdel = 6, i_rpdel = 7, i_lnpmid = 8, i_zm = 9, i_cld = 10, i_pint = 11, i_lnpint = 12, i_zi = 13, i_phis = 14, i_ps = 15, i_pblh = 16, i_tpert = 17, i_landfrac = 18
  integer, parameter :: o_ptend_s = 1, o_ptend_q = 2, o_ptend_u = 3, o_ptend_v = 4, o_mconzm = 5, o_flxprec = 6, o_flxsnow = 7, o_ql = 8, o_rprd = 9, o_evapcdp = 10, o_dlf = 11, o_prec = 12, o_snow = 13, o_rliq = 14, o_rice = 15, o_jctop = 16, o_jcbot = 17
  integer, parameter :: o_cape = 18, o_freqzm = 19, o_pcont = 20, o_pconb = 21, o_lengath = 22, o_mu = 23, o_md = 24, o_du = 25, o_eu = 26, o_ed = 27, o_dp = 28, o_dsubcld = 29, o_jt = 30, o_maxg = 31, o_ideep = 32, o_status = 33

contains

  subroutine fail(what)
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    character(len=256) :: text
    integer :: i
    text = ' '
    call c_f_pointer(colphys_last_error(), msg, [256])
    do i = 1, 256
      if (msg(i) == c_null_char) exit
      text(i:i) = msg(i)
    end do
    call endrun('colphys_hostface_mod: '//what//': '//trim(text))
  end subroutine fail

  ! ncol(1:nchunks): active columns of the rank's chunks, in the order begchunk .. endchunk.  zm_convi (the shim's, or
  ! colphys_init) must have run: the engine's parameters and device are those.
  subroutine colphys_rank_init(ncol)
    integer, intent(in) :: ncol(:)
    integer(c_int32_t) :: nc(size(ncol)), nf, nlev, is_int, is_out
    integer(c_int64_t) :: off
    character(kind=c_char) :: cname(32)
    character(len=32) :: name
    integer :: i, j
    if (c_associated(hf)) call colphys_rank_final()
    nc = ncol
    nchunks_rank = size(ncol)
    allocate(ncol_rank(nchunks_rank)); ncol_rank = ncol
    if (colphys_hostface_create(nchunks_rank, pcols, pver, nc, COLPHYS_HF_ZM, 0, hf) /= 0) call fail('colphys_hostface_create')
    if (colphys_hostface_geometry(hf, in_rec, out_rec, nf) /= 0) call fail('colphys_hostface_geometry')
    do i = 1, nf
      if (colphys_hostface_field(hf, i - 1, cname, nlev, is_int, is_out, off) /= 0) call fail('colphys_hostface_field')
      name = ' '
      do j = 1, 32
        if (cname(j) == c_null_char) exit
        name(j:j) = cname(j)
      end do
      if (is_out /= 0) then
        do j = 1, n_out
          if (trim(out_names(j)) == trim(name)) off_out(j) = off
        end do
      else
        do j = 1, n_in
          if (trim(in_names(j)) == trim(name)) off_in(j) = off
        end do
      end if
    end do
    if (any(off_in < 0) .or. any(off_out < 0)) call endrun('colphys_hostface_mod: the library does not offer every field this face moves')
    in_base = transfer(colphys_hostface_in_base(hf), in_base)
    out_base = transfer(colphys_hostface_out_base(hf), out_base)
  end subroutine colphys_rank_init

  subroutine colphys_rank_final()
    integer(c_int) :: rc
    if (c_associated(hf)) rc = colphys_hostface_destroy(hf)
    hf = c_null_ptr
    if (allocated(ncol_rank)) deallocate(ncol_rank)
    nchunks_rank = 0
  end subroutine colphys_rank_final

  ! address of input field k (i_*) / output field k (o_*) in chunk c's record
  function addr_in(k, c) result(addr)
    integer, intent(in) :: k, c
    type(c_ptr) :: addr
    addr = transfer(in_base + int(c - 1, c_intptr_t) * in_rec + off_in(k), addr)
  end function addr_in
  function addr_out(k, c) result(addr)
    integer, intent(in) :: k, c
    type(c_ptr) :: addr
    addr = transfer(out_base + int(c - 1, c_intptr_t) * out_rec + off_out(k), addr)
  end function addr_out

  subroutine rec2(addr, nlev, p)
    type(c_ptr), intent(in) :: addr
    integer, intent(in) :: nlev
    real(r8), pointer :: p(:,:)
    call c_f_pointer(addr, p, [pcols, nlev])
  end subroutine rec2
  subroutine rec1(addr, p)
    type(c_ptr), intent(in) :: addr
    real(r8), pointer :: p(:)
    call c_f_pointer(addr, p, [pcols])
  end subroutine rec1
  subroutine reci(addr, n, p)
    type(c_ptr), intent(in) :: addr
    integer, intent(in) :: n
    integer(c_int32_t), pointer :: p(:)
    call c_f_pointer(addr, p, [n])
  end subroutine reci

  ! ---- chunk loop, first half: state(c) and the pbuf / surface inputs of zm_conv_tend into the chunk's record -------------
  ! (what zm_conv_tend reads: zm_conv_intr.F90:662-673 the state fields, :585 CLD at the old time level; whole (pcols, nlev)
  ! arrays are copied, padding and all -- the device shadows the padding)
  subroutine colphys_rank_push(c, state, pblh, tpert, landfrac, cld)
    integer, intent(in) :: c
    type(physics_state), intent(in) :: state
    real(r8), intent(in) :: pblh(pcols), tpert(pcols), landfrac(pcols), cld(pcols, pver)
    real(r8), pointer :: p2(:,:), p1(:)
    if (state%ncol /= ncol_rank(c)) call endrun('colphys_rank_push: state%ncol differs from the count given to colphys_rank_init')
    call rec2(addr_in(i_t, c), pver, p2);       p2 = state%t(:pcols, :pver)
    call rec2(addr_in(i_q, c), pver, p2);       p2 = state%q(:pcols, :pver, 1)
    call rec2(addr_in(i_u, c), pver, p2);       p2 = state%u(:pcols, :pver)
    call rec2(addr_in(i_v, c), pver, p2);       p2 = state%v(:pcols, :pver)
    call rec2(addr_in(i_pmid, c), pver, p2);    p2 = state%pmid(:pcols, :pver)
    call rec2(addr_in(i_pdel, c), pver, p2);    p2 = state%pdel(:pcols, :pver)
    call rec2(addr_in(i_rpdel, c), pver, p2);   p2 = state%rpdel(:pcols, :pver)
    call rec2(addr_in(i_lnpmid, c), pver, p2);  p2 = state%lnpmid(:pcols, :pver)
    call rec2(addr_in(i_zm, c), pver, p2);      p2 = state%zm(:pcols, :pver)
    call rec2(addr_in(i_cld, c), pver, p2);     p2 = cld
    call rec2(addr_in(i_pint, c), pverp, p2);   p2 = state%pint(:pcols, :pverp)
    call rec2(addr_in(i_lnpint, c), pverp, p2); p2 = state%lnpint(:pcols, :pverp)
    call rec2(addr_in(i_zi, c), pverp, p2);     p2 = state%zi(:pcols, :pverp)
    call rec1(addr_in(i_phis, c), p1);          p1 = state%phis(:pcols)
    call rec1(addr_in(i_ps, c), p1);            p1 = state%ps(:pcols)
    call rec1(addr_in(i_pblh, c), p1);          p1 = pblh
    call rec1(addr_in(i_tpert, c), p1);         p1 = tpert
    call rec1(addr_in(i_landfrac, c), p1);      p1 = landfrac
  end subroutine colphys_rank_push

  ! ---- ONE dispatch for the rank ---------------------------------------------------------------------------------------
  subroutine colphys_rank_run_zm_conv_tend(ztodt, dycore_kind, qmin)
    real(r8), intent(in) :: ztodt
    integer, intent(in), optional :: dycore_kind          ! COLPHYS_DYCORE_* of include/colphys.h; default 1 = FV ('LR')
    real(r8), intent(in), optional :: qmin                ! constituents' qmin(1); default 1e-12
    integer(c_int) :: dk
    real(c_double) :: qm
    integer :: c, i
    integer(c_int32_t), pointer :: st(:)
    dk = 1; qm = 1.e-12_r8
    if (present(dycore_kind)) dk = dycore_kind
    if (present(qmin)) qm = qmin
    if (colphys_hostface_run(hf, ztodt, dk, qm) /= 0) call fail('colphys_hostface_run')
    ! what the reference ends the run on (zm_conv.F90:5401-5410, 5557-5566): an inversion that did not converge
    do c = 1, nchunks_rank
      call reci(addr_out(o_status, c), pcols, st)
      do i = 1, ncol_rank(c)
        if (iand(st(i), 1) /= 0) then
          if (iand(st(i), 8) /= 0) then
            write(iulog,*) '*** ZM_CONV: IENTHALPY: Failed and about to exit, chunk,icol= ', c, i
            call endrun('**** ZM_CONV IENTHALPY: Tmix did not converge ****')
          end if
          write(iulog,*) '*** ZM_CONV: IENTROPY: Failed and about to exit, chunk,icol= ', c, i
          call endrun('**** ZM_CONV IENTROPY: Tmix did not converge ****')
        end if
      end do
    end do
  end subroutine colphys_rank_run_zm_conv_tend

  ! ---- chunk loop, second half: the chunk's results into ptend_all, zm_conv_tend's arguments and the pbuf fields ---------
  ! ptend_all is initialised as zm_conv_tend leaves it before convtran1 (:731-859): s, q(:,:,1), u, v.  mcon is the mass flux in
  ! kg/m2/s (:691); the pbuf arrays are what pbuf_get_field returned for the fields of the same comment.
  subroutine colphys_rank_pull(c, state, ptend_all, mcon, rliq, rice, jctop, jcbot, prec, snow, ql, rprd, evapcdp, dlf, mconzm, &
                               flxprec, flxsnow, mu, md, du, eu, ed, dp, dsubcld, jt, maxg, ideep, lengath, cape, freqzm, pcont, pconb)
    integer, intent(in) :: c
    type(physics_state), intent(in) :: state
    type(physics_ptend), intent(out) :: ptend_all
    real(r8), intent(out) :: mcon(pcols, pverp), rliq(pcols), rice(pcols), jctop(pcols), jcbot(pcols)
    real(r8), intent(inout) :: prec(pcols), snow(pcols)                                   ! PREC_DP, SNOW_DP
    real(r8), intent(inout) :: ql(pcols, pver), rprd(pcols, pver), evapcdp(pcols, pver), dlf(pcols, pver)   ! ICWMRDP, RPRDDP, NEVAPR_DPCU, DLFZM
    real(r8), intent(inout) :: mconzm(pcols, pverp), flxprec(pcols, pverp), flxsnow(pcols, pverp)           ! CMFMC_DP, DP_FLXPRC, DP_FLXSNW
    real(r8), intent(inout) :: mu(pcols, pver), md(pcols, pver), du(pcols, pver), eu(pcols, pver), ed(pcols, pver), dp(pcols, pver)
    real(r8), intent(inout) :: dsubcld(pcols)
    integer, intent(inout) :: jt(pcols), maxg(pcols), ideep(pcols)                         ! ZM_JT, ZM_MAXG, ZM_IDEEP
    integer, intent(out) :: lengath
    real(r8), intent(out) :: cape(pcols), freqzm(pcols), pcont(pcols), pconb(pcols)        ! history: CAPE, FREQZM, PCONVT, PCONVB
    real(r8), pointer :: p2(:,:), p1(:)
    integer(c_int32_t), pointer :: pi(:)
    logical :: lq(pcnst)
    integer :: n
    n = state%ncol
    lq = .false.; lq(1) = .true.
    call physics_ptend_init(ptend_all, state%psetcols, 'zm_conv_tend', ls=.true., lu=.true., lv=.true., lq=lq)
    call rec2(addr_out(o_ptend_s, c), pver, p2); ptend_all%s(:n, :pver) = p2(:n, :)
    call rec2(addr_out(o_ptend_q, c), pver, p2); ptend_all%q(:n, :pver, 1) = p2(:n, :)
    call rec2(addr_out(o_ptend_u, c), pver, p2); ptend_all%u(:n, :pver) = p2(:n, :)
    call rec2(addr_out(o_ptend_v, c), pver, p2); ptend_all%v(:n, :pver) = p2(:n, :)
    call rec2(addr_out(o_mconzm, c), pverp, p2);  mcon(:n, :) = p2(:n, :); mconzm(:n, :) = p2(:n, :)
    call rec2(addr_out(o_flxprec, c), pverp, p2); flxprec(:n, :) = p2(:n, :)
    call rec2(addr_out(o_flxsnow, c), pverp, p2); flxsnow(:n, :) = p2(:n, :)
    call rec2(addr_out(o_ql, c), pver, p2);       ql(:n, :) = p2(:n, :)
    call rec2(addr_out(o_rprd, c), pver, p2);     rprd(:n, :) = p2(:n, :)
    call rec2(addr_out(o_evapcdp, c), pver, p2);  evapcdp(:n, :) = p2(:n, :)
    call rec2(addr_out(o_dlf, c), pver, p2);      dlf(:n, :) = p2(:n, :)
    call rec1(addr_out(o_prec, c), p1);  prec(:n) = p1(:n)
    call rec1(addr_out(o_snow, c), p1);  snow(:n) = p1(:n)
    call rec1(addr_out(o_rliq, c), p1);  rliq(:n) = p1(:n)
    call rec1(addr_out(o_rice, c), p1);  rice(:n) = p1(:n)
    call rec1(addr_out(o_jctop, c), p1); jctop(:n) = p1(:n)
    call rec1(addr_out(o_jcbot, c), p1); jcbot(:n) = p1(:n)
    call rec1(addr_out(o_cape, c), p1);  cape(:n) = p1(:n)
    call rec1(addr_out(o_freqzm, c), p1); freqzm(:n) = p1(:n)
    call rec1(addr_out(o_pcont, c), p1); pcont(:n) = p1(:n)
    call rec1(addr_out(o_pconb, c), p1); pconb(:n) = p1(:n)
    ! the gathered set: slots 1:lengath hold the deep columns in ascending order (zm_conv.F90:926-940); the rest is zero
    call reci(addr_out(o_lengath, c), 1, pi); lengath = pi(1)
    call rec2(addr_out(o_mu, c), pver, p2); mu = p2
    call rec2(addr_out(o_md, c), pver, p2); md = p2
    call rec2(addr_out(o_du, c), pver, p2); du = p2
    call rec2(addr_out(o_eu, c), pver, p2); eu = p2
    call rec2(addr_out(o_ed, c), pver, p2); ed = p2
    call rec2(addr_out(o_dp, c), pver, p2); dp = p2
    call rec1(addr_out(o_dsubcld, c), p1);  dsubcld = p1
    call reci(addr_out(o_jt, c), pcols, pi);    jt = pi
    call reci(addr_out(o_maxg, c), pcols, pi);  maxg = pi
    call reci(addr_out(o_ideep, c), pcols, pi); ideep = pi
  end subroutine colphys_rank_pull

  ! wall time of the last dispatch, the calling thread's CPU time in it (enqueueing; it sleeps while the device works), bytes moved
  subroutine colphys_rank_last_times(wall_ms, host_cpu_ms, h2d_bytes, d2h_bytes)
    real(r8), intent(out) :: wall_ms, host_cpu_ms
    integer(c_int64_t), intent(out) :: h2d_bytes, d2h_bytes
    type(colphys_hostface_times) :: t
    if (colphys_hostface_last_times(hf, t) /= 0) call fail('colphys_hostface_last_times')
    wall_ms = t%wall_ms; host_cpu_ms = t%host_cpu_ms; h2d_bytes = t%h2d_bytes; d2h_bytes = t%d2h_bytes
  end subroutine colphys_rank_last_times

end module colphys_hostface_mod
