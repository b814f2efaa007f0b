! modal_aer_opt_hip -- ISO_C_BINDING shim: the reference's modal aerosol optics surface, bodies on the MI355X.
!
! Public procedures have the SAME names, argument lists, kinds and intents as the reference module
! modal_aer_opt (NorESMhub/CAM-Nor-physics physics/modal_aer_opt.F90):
!     modal_aer_opt_readnl :74      modal_aer_opt_init :113      modal_aero_sw :485-486      modal_aero_lw :1216
! so the radiation-side caller (aer_rad_props: `call modal_aero_sw(list_idx, state, pbuf, nnite, idxnite, tauxar, wa,
! ga, fa)`, `call modal_aero_lw(list_idx, state, pbuf, tauxar)`) compiles against it unchanged.
!
! What stays on the host, exactly where the reference has it: the accessors.  Mode / species properties come from
! rad_constituents (rad_cnst_get_mode_props, rad_cnst_get_aer_props, rad_cnst_get_info), the mixing ratios from
! rad_cnst_get_aer_mmr, wet size and aerosol water from the physics buffer (DGNUMWET, QAERWAT), the water refractive
! index from the file named in the namelist through pio -- the same calls as modal_aer_opt.F90:153, 167-172,
! 679-681, 717-724, 759-762, 1334-1355, 1485-1530.  The shim hands the tables to the engine once
! (colphys_aero_tables_set) and per call marshals the chunk's fields to colphys_modal_aero_{sw,lw}_chunks.
!
! History diagnostics of modal_aero_sw (:896-993, 1046-1209): the values come from the engine
! (colphys_modal_aero_sw_diag_chunks, one upload shared with the radiative kernel); the night-column fill and the
! outfld calls -- same names, same order, "dn" field before its day-only twin -- are here.  addfld registration
! (:173-480) is the host model's history setup and is not repeated.
! Not offloaded (DESIGN.md "out of scope"): the diagnostic radiation lists (list_idx > 0, :683-699), which need the absent
! modal_aero_calcsize / modal_aero_wateruptake; the shim aborts on list_idx > 0 instead of returning wrong numbers.
!
! Module name: `modal_aer_opt_hip` by default; with -DCOLPHYS_DROPIN_NAMES it is `modal_aer_opt`, a whole-module
! replacement -- tests/ build it that way against the very wrapper source that drives the reference
! (oracle/ref_wrap_aero.F90).
#ifdef COLPHYS_DROPIN_NAMES
#define AER_MODULE modal_aer_opt
#else
#define AER_MODULE modal_aer_opt_hip
#endif

module AER_MODULE
  use iso_c_binding
  use shr_kind_mod,      only: r8 => shr_kind_r8, shr_kind_cl
  use ppgrid,            only: pcols, pver, pverp
  use spmd_utils,        only: masterproc
  use ref_pres,          only: top_lev => clim_modal_aero_top_lev
  use physconst,         only: rhoh2o, rga, rair, cpair, epsilo, gravit, latice, latvap, tmelt, cpwv, cpliq, rh2o, &
                               cpvir, zvir
  use radconstants,      only: nswbands, nlwbands, idx_sw_diag, idx_uv_diag, idx_nir_diag
  use rad_constituents,  only: n_diag, rad_cnst_get_call_list, rad_cnst_get_info, rad_cnst_get_aer_mmr, &
                               rad_cnst_get_aer_props, rad_cnst_get_mode_props
  use physics_types,     only: physics_state
  use physics_buffer,    only: pbuf_get_index, physics_buffer_desc, pbuf_get_field
  use pio,               only: file_desc_t, var_desc_t, pio_inq_dimlen, pio_inq_dimid, pio_inq_varid, &
                               pio_get_var, pio_nowrite, pio_closefile
  use cam_pio_utils,     only: cam_pio_openfile
  use cam_logfile,       only: iulog
  use cam_abortutils,    only: endrun
  use cam_history,       only: outfld
  use cam_history_support, only: fillvalue
  use colphys_c_api
  implicit none
  private
  save

  public :: modal_aer_opt_readnl, modal_aer_opt_init, modal_aero_sw, modal_aero_lw
  ! whole-rank dispatch (not part of the reference's surface; the optics' half of the host face, colphys_hostface_mod.F90
  ! being the convection step's): every chunk pushed, ONE run, every chunk pulled
  public :: modal_aero_rank_init, modal_aero_rank_final, modal_aero_rank_push, modal_aero_rank_run, modal_aero_rank_pull_sw, &
            modal_aero_rank_pull_lw

  integer, parameter :: ncoef = 5, prefr = 7, prefi = 10            ! modal_aer_opt.F90:56
  character(len=*), parameter :: unset_str = 'UNSET'
  character(shr_kind_cl) :: water_refindex_file = unset_str         ! namelist modal_aer_opt_nl (:51, :84)

  integer :: dgnumwet_idx = -1, qaerwat_idx = -1                    ! physics buffer indices (:63-64)
  integer :: nmodes_clim = 0, nspec_max = 0
  integer, parameter :: nerrmax_dopaer = 1000                       ! :619, :1275
  integer :: nerr_dopaer_sw = 0, nerr_dopaer_lw = 0                 ! :620 (modal_aero_sw's), :1276 (modal_aero_lw's)
  integer(c_int32_t), allocatable, target :: nspec_m(:)

  ! field order of the engine's packed SW diagnostics (include/colphys.h COLPHYS_SWD_*)
  integer, parameter :: nlevf = 5, ncolf0 = 23

  ! ---- whole-rank dispatch: the host face's handle and the record offsets of the fields this module moves ----------------
  type(c_ptr) :: hf = c_null_ptr
  integer :: nchunks_rank = 0
  integer, allocatable :: ncol_rank(:)
  integer(c_intptr_t) :: in_base = 0, out_base = 0
  integer(c_int64_t)  :: in_rec = 0, out_rec = 0
  integer(c_int64_t)  :: off_t = -1, off_pmid = -1, off_pdeldry = -1, off_troplev = -1
  integer(c_int64_t), allocatable :: off_dgn(:), off_qaw(:), off_mmr(:,:)              ! (mode), (mode), (species, mode)
  integer(c_int64_t)  :: off_sw(4) = -1, off_lw = -1, off_swst = -1, off_lwst = -1, off_swdlev = -1, off_swdcol = -1

contains

  !---------------------------------------------------------------------------------------------------------
  subroutine modal_aer_opt_readnl(nlfile)
    ! modal_aer_opt.F90:74-109: the one namelist variable of group modal_aer_opt_nl, read on the master task
    ! and broadcast
    use namelist_utils, only: find_group_name
    use units,          only: getunit, freeunit
    use mpishorthand
    character(len=*), intent(in) :: nlfile
    integer :: u, ios
    namelist /modal_aer_opt_nl/ water_refindex_file

    if (masterproc) then
      u = getunit()
      open(u, file=trim(nlfile), status='old')
      call find_group_name(u, 'modal_aer_opt_nl', status=ios)
      if (ios == 0) then
        read(u, modal_aer_opt_nl, iostat=ios)
        if (ios /= 0) call endrun('modal_aer_opt_readnl:: ERROR reading namelist')
      end if
      close(u)
      call freeunit(u)
    end if
#ifdef SPMD
    call mpibcast(water_refindex_file, len(water_refindex_file), mpichar, 0, mpicom)
#endif
  end subroutine modal_aer_opt_readnl

  !---------------------------------------------------------------------------------------------------------
  subroutine modal_aer_opt_init()
    ! modal_aer_opt.F90:113-480 without the history registration: size-range constants, consistency of the
    ! coefficient-table dimensions, physics-buffer indices, water refractive index; then every table of the climate
    ! list goes to the device once.
    use ioFileMod, only: getfil
    logical :: call_list(0:n_diag)
    integer :: ilist, nmodes, m, l, ib, m_ncoef, m_prefr, m_prefi, errcode, nspec, rc
    character(len=256) :: locfile
    character(len=32)  :: spectype
    real(r8) :: crefwsw(2, nswbands), crefwlw(2, nlwbands)
    real(r8) :: specdens_1, hygro_1, sigma_1
    complex(r8), pointer :: ri_sw(:), ri_lw(:)
    real(r8), pointer :: p_refrsw(:,:), p_refisw(:,:), p_refrlw(:,:), p_refilw(:,:)
    real(r8), pointer :: p_ext(:,:,:,:), p_abs(:,:,:,:), p_asm(:,:,:,:), p_abl(:,:,:,:)
    integer(c_int32_t), allocatable, target :: spectype_c(:,:)
    real(c_double), allocatable, target :: sigmag(:), specdens(:,:), hygro(:,:), sri_sw(:,:,:,:), sri_lw(:,:,:,:)
    real(c_double), allocatable, target :: refrsw(:,:,:), refisw(:,:,:), refrlw(:,:,:), refilw(:,:,:)
    real(c_double), allocatable, target :: ext(:,:,:,:,:), abss(:,:,:,:,:), asmp(:,:,:,:,:), abl(:,:,:,:,:)
    real(c_double), target :: cw_sw(2, nswbands), cw_lw(2, nlwbands)
    type(colphys_aero_tables) :: tb
    type(colphys_zm_params) :: zp

    call rad_cnst_get_call_list(call_list)                                          ! :148-163
    do ilist = 0, n_diag
      if (call_list(ilist)) then
        call rad_cnst_get_info(ilist, nmodes=nmodes)
        do m = 1, nmodes
          call rad_cnst_get_mode_props(ilist, m, ncoef=m_ncoef, prefr=m_prefr, prefi=m_prefi)
          if (m_ncoef /= ncoef .or. m_prefr /= prefr .or. m_prefi /= prefi) then
            write(iulog,*) 'modal_aer_opt_init: ERROR - file and module values do not match:'
            write(iulog,*) '   ncoef:', ncoef, m_ncoef, '   prefr:', prefr, m_prefr, '   prefi:', prefi, m_prefi
            call endrun('modal_aer_opt_init: ERROR - file and module values do not match')
          end if
        end do
      end if
    end do

    dgnumwet_idx = pbuf_get_index('DGNUMWET', errcode)                              ! :168-175
    if (errcode < 0) call endrun('modal_aer_opt_init ERROR: cannot find physics buffer field DGNUMWET')
    qaerwat_idx = pbuf_get_index('QAERWAT', errcode)
    if (errcode < 0) call endrun('modal_aer_opt_init ERROR: cannot find physics buffer field QAERWAT')

    call getfil(water_refindex_file, locfile)                                       ! :177-179
    call read_water_refindex(locfile, cw_sw, cw_lw)
    if (masterproc) write(iulog,*) 'modal_aer_opt_init: read water refractive index file:', trim(locfile)

    ! ---- tables of the climate list (list_idx = 0), laid out as include/colphys.h describes ---------------
    call rad_cnst_get_info(0, nmodes=nmodes_clim)
    nmodes = nmodes_clim
    if (allocated(nspec_m)) deallocate(nspec_m)
    allocate(nspec_m(nmodes))
    do m = 1, nmodes
      call rad_cnst_get_info(0, m, nspec=nspec)
      nspec_m(m) = nspec
    end do
    nspec_max = maxval(nspec_m)
    allocate(spectype_c(nspec_max, nmodes), sigmag(nmodes), specdens(nspec_max, nmodes), hygro(nspec_max, nmodes), &
             sri_sw(2, nswbands, nspec_max, nmodes), sri_lw(2, nlwbands, nspec_max, nmodes), &
             refrsw(prefr, nswbands, nmodes), refisw(prefi, nswbands, nmodes), &
             refrlw(prefr, nlwbands, nmodes), refilw(prefi, nlwbands, nmodes), &
             ext(ncoef, prefr, prefi, nswbands, nmodes), abss(ncoef, prefr, prefi, nswbands, nmodes), &
             asmp(ncoef, prefr, prefi, nswbands, nmodes), abl(ncoef, prefr, prefi, nlwbands, nmodes))
    spectype_c = -1; specdens = 1._r8; hygro = 0._r8; sri_sw = 0._r8; sri_lw = 0._r8
    do m = 1, nmodes
      call rad_cnst_get_mode_props(0, m, sigmag=sigma_1, refrtabsw=p_refrsw, refitabsw=p_refisw, extpsw=p_ext, &
                                   abspsw=p_abs, asmpsw=p_asm)                      ! :717-718
      call rad_cnst_get_mode_props(0, m, refrtablw=p_refrlw, refitablw=p_refilw, absplw=p_abl)   ! :1334-1335
      sigmag(m) = sigma_1
      refrsw(:,:,m) = p_refrsw; refisw(:,:,m) = p_refisw; refrlw(:,:,m) = p_refrlw; refilw(:,:,m) = p_refilw
      ext(:,:,:,:,m) = p_ext; abss(:,:,:,:,m) = p_abs; asmp(:,:,:,:,m) = p_asm; abl(:,:,:,:,m) = p_abl
      do l = 1, nspec_m(m)
        call rad_cnst_get_aer_props(0, m, l, density_aer=specdens_1, refindex_aer_sw=ri_sw, spectype=spectype, &
                                    hygro_aer=hygro_1)                              ! :760-762
        call rad_cnst_get_aer_props(0, m, l, refindex_aer_lw=ri_lw)                 ! :1354-1355
        specdens(l, m) = specdens_1; hygro(l, m) = hygro_1
        do ib = 1, nswbands
          sri_sw(1, ib, l, m) = real(ri_sw(ib)); sri_sw(2, ib, l, m) = aimag(ri_sw(ib))
        end do
        do ib = 1, nlwbands
          sri_lw(1, ib, l, m) = real(ri_lw(ib)); sri_lw(2, ib, l, m) = aimag(ri_lw(ib))
        end do
        select case (trim(spectype))                                                ! :783-833 (diagnostics only)
        case ('sulfate');   spectype_c(l, m) = 0
        case ('p-organic'); spectype_c(l, m) = 1
        case ('s-organic'); spectype_c(l, m) = 2
        case ('black-c');   spectype_c(l, m) = 3
        case ('dust');      spectype_c(l, m) = 4
        case ('seasalt');   spectype_c(l, m) = 5
        end select
      end do
    end do

    ! the engine must be up before tables can be uploaded; zm_convi normally did that already (re-initialising
    ! later with the ZM namelist keeps the tables)
    if (colphys_is_initialized() == 0) then
      zp%cpair = cpair; zp%epsilo = epsilo; zp%gravit = gravit; zp%latice = latice; zp%latvap = latvap; zp%tmelt = tmelt
      zp%rair = rair; zp%cpwv = cpwv; zp%cpliq = cpliq; zp%rh2o = rh2o; zp%cpvir = cpvir; zp%zvir = zvir
      zp%c0_lnd = 0._r8; zp%c0_ocn = 0._r8; zp%ke = 0._r8; zp%ke_lnd = 0._r8; zp%momcu = 0._r8; zp%momcd = 0._r8
      zp%tiedke_add = 0._r8; zp%capelmt = 70._r8; zp%dmpdz_param = -1.e-3_r8; zp%tau = 3600._r8
      zp%limcnv = 1; zp%num_cin = 1; zp%zm_org = 0; zp%zmconv_microp = 0; zp%no_deep_pbl = 0; zp%lparcel_pbl = 0
      call colphys_zm_params_defaults(zp)
      rc = colphys_init(aero_device(), zp)
      if (rc /= COLPHYS_OK) call endrun('modal_aer_opt_init (colphys): '//trim(colphys_error_message()))
    end if

    tb%nmodes = nmodes; tb%nspec_max = nspec_max; tb%nswbands = nswbands; tb%nlwbands = nlwbands
    tb%top_lev = top_lev
    tb%idx_sw_diag = idx_sw_diag; tb%idx_uv_diag = idx_uv_diag; tb%idx_nir_diag = idx_nir_diag
    tb%rhoh2o = rhoh2o; tb%rga = rga; tb%rair = rair
    tb%xrmin = log(0.01e-6_r8); tb%xrmax = log(25.e-6_r8)                            ! :140-143
    tb%nspec = c_loc(nspec_m); tb%spectype = c_loc(spectype_c)
    tb%sigmag = c_loc(sigmag); tb%specdens = c_loc(specdens); tb%hygro = c_loc(hygro)
    tb%specrefindex_sw = c_loc(sri_sw); tb%specrefindex_lw = c_loc(sri_lw)
    tb%refrtabsw = c_loc(refrsw); tb%refitabsw = c_loc(refisw); tb%refrtablw = c_loc(refrlw); tb%refitablw = c_loc(refilw)
    tb%extpsw = c_loc(ext); tb%abspsw = c_loc(abss); tb%asmpsw = c_loc(asmp); tb%absplw = c_loc(abl)
    tb%crefwsw = c_loc(cw_sw); tb%crefwlw = c_loc(cw_lw)
    rc = colphys_aero_tables_set(tb)
    if (rc /= COLPHYS_OK) call endrun('modal_aer_opt_init (colphys): '//trim(colphys_error_message()))
  end subroutine modal_aer_opt_init

  integer function aero_device()
    ! one process per GPU: COLPHYS_DEVICE selects it (a launcher sets it to the node-local MPI rank)
    character(len=16) :: env
    integer :: stat, ndev
    aero_device = 0
    call get_environment_variable('COLPHYS_DEVICE', env, status=stat)
    if (stat == 0) read(env, *, iostat=stat) aero_device
    ndev = colphys_device_count()
    if (ndev > 0) aero_device = mod(aero_device, ndev)
  end function aero_device

  !---------------------------------------------------------------------------------------------------------
  subroutine read_water_refindex(infilename, cw_sw, cw_lw)
    ! modal_aer_opt.F90:1475-1534: band counts checked against radconstants, four variables read; the imaginary
    ! part enters with its absolute value (:1525-1530)
    character(len=*), intent(in) :: infilename
    real(r8), intent(out) :: cw_sw(2, nswbands), cw_lw(2, nlwbands)
    type(file_desc_t) :: ncid
    type(var_desc_t)  :: vid
    integer :: ierr, did, dimlen
    real(r8) :: re_sw(nswbands), im_sw(nswbands), re_lw(nlwbands), im_lw(nlwbands)

    call cam_pio_openfile(ncid, infilename, PIO_NOWRITE)
    ierr = pio_inq_dimid(ncid, 'lw_band', did)
    ierr = pio_inq_dimlen(ncid, did, dimlen)
    if (dimlen /= nlwbands) then
      write(iulog,*) 'lw_band len=', dimlen, ' from ', infilename, ' ne nlwbands=', nlwbands
      call endrun('read_modal_optics: bad lw_band value')
    end if
    ierr = pio_inq_dimid(ncid, 'sw_band', did)
    ierr = pio_inq_dimlen(ncid, did, dimlen)
    if (dimlen /= nswbands) then
      write(iulog,*) 'sw_band len=', dimlen, ' from ', infilename, ' ne nswbands=', nswbands
      call endrun('read_modal_optics: bad sw_band value')
    end if
    ierr = pio_inq_varid(ncid, 'refindex_real_water_sw', vid); ierr = pio_get_var(ncid, vid, re_sw)
    ierr = pio_inq_varid(ncid, 'refindex_im_water_sw', vid);   ierr = pio_get_var(ncid, vid, im_sw)
    ierr = pio_inq_varid(ncid, 'refindex_real_water_lw', vid); ierr = pio_get_var(ncid, vid, re_lw)
    ierr = pio_inq_varid(ncid, 'refindex_im_water_lw', vid);   ierr = pio_get_var(ncid, vid, im_lw)
    cw_sw(1,:) = re_sw; cw_sw(2,:) = abs(im_sw)
    cw_lw(1,:) = re_lw; cw_lw(2,:) = abs(im_lw)
    call pio_closefile(ncid)
  end subroutine read_water_refindex

  !---------------------------------------------------------------------------------------------------------
  ! the chunk's inputs in the host layout of the *_chunks entry points (one chunk: (pcols,pver[,nspec_max],nmodes))
  subroutine marshal_inputs(routine, list_idx, state, pbuf, pdeldry, specmmr, dgnumwet, qaerwat)
    character(len=*), intent(in) :: routine
    integer, intent(in) :: list_idx
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(c_double), intent(out) :: pdeldry(pcols, pver), specmmr(pcols, pver, nspec_max, nmodes_clim)
    real(c_double), intent(out) :: dgnumwet(pcols, pver, nmodes_clim), qaerwat(pcols, pver, nmodes_clim)
    real(r8), pointer :: dgnumwet_m(:,:,:), qaerwat_m(:,:,:), mmr(:,:)
    integer :: ncol, m, l

    if (list_idx /= 0) call endrun(routine//' (colphys): diagnostic radiation lists (list_idx > 0) are not offloaded')
    if (nmodes_clim <= 0) call endrun(routine//' (colphys): modal_aer_opt_init has not been called')
    ncol = state%ncol
    call pbuf_get_field(pbuf, dgnumwet_idx, dgnumwet_m)                              ! :679-681 / :1320-1322
    call pbuf_get_field(pbuf, qaerwat_idx,  qaerwat_m)
    pdeldry = 1._r8; specmmr = 0._r8; dgnumwet = 1.e-7_r8; qaerwat = 0._r8           ! padding is never read back
    pdeldry(:ncol, :) = state%pdeldry(:ncol, :)
    do m = 1, nmodes_clim
      dgnumwet(:ncol, :, m) = dgnumwet_m(:ncol, :, m)
      qaerwat(:ncol, :, m)  = qaerwat_m(:ncol, :, m)
      do l = 1, nspec_m(m)
        call rad_cnst_get_aer_mmr(list_idx, m, l, 'a', state, pbuf, mmr)            ! :759 / :1353
        specmmr(:ncol, :, l, m) = mmr(:ncol, :)
      end do
    end do
  end subroutine marshal_inputs

  subroutine check_status(subname, lchnk, ncol, status, nerr_dopaer, nerrmax)
    ! The reference's sanity check of the layer optical depths (:997-1029 SW, :1402-1437 LW): every value outside the range is
    ! logged and counted in nerr_dopaer (SAVEd in each routine, so counted over the run); the run halts on a negative value
    ! -- a large one is a warning -- and, in modal_aero_lw only, once the count reaches nerrmax_dopaer.  The engine reports
    ! per column what the reference tests per (mode, band, level): the count in the chunk is exact, the count quoted in the
    ! halting message is the one at the end of the chunk's call, not at the first offending layer.
    use string_utils, only: int2str
    character(len=*), intent(in) :: subname
    integer, intent(in) :: lchnk, ncol
    integer(c_int32_t), intent(in) :: status(pcols)
    integer, intent(inout) :: nerr_dopaer
    integer, intent(in) :: nerrmax                  ! < 0: the count alone never halts (modal_aero_sw)
    integer :: i
    logical :: negative
    negative = .false.
    do i = 1, ncol
      if (iand(status(i), COLPHYS_STATUS_DOPAER_RANGE) /= 0) then
        if (iand(status(i), COLPHYS_STATUS_DOPAER_NEGATIVE) /= 0) then
          write(iulog,*) "ERROR: Negative aerosol optical depth in this layer."
          negative = .true.
        else
          write(iulog,*) "WARNING: Aerosol optical depth is unreasonably high in this layer."
        end if
        write(iulog,*) subname//': lchnk=', lchnk, ' column=', i, ' layers out of range=', ishft(status(i), -COLPHYS_STATUS_COUNT_SHIFT)
        nerr_dopaer = nerr_dopaer + ishft(status(i), -COLPHYS_STATUS_COUNT_SHIFT)
      end if
    end do
    if (negative .or. (nerrmax >= 0 .and. nerr_dopaer >= nerrmax)) then
      write(iulog,*) '*** halting in ', subname, ' after nerr_dopaer =', nerr_dopaer
      call endrun('*** halting in '//subname//' after nerr_dopaer = '//int2str(nerr_dopaer))
    end if
  end subroutine check_status

  !---------------------------------------------------------------------------------------------------------
  subroutine modal_aero_sw(list_idx, state, pbuf, nnite, idxnite, tauxar, wa, ga, fa)
    ! modal_aer_opt.F90:485-1212
    use tropopause, only: tropopause_findChemTrop
    integer,             intent(in) :: list_idx
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    integer,             intent(in) :: nnite
    integer,             intent(in) :: idxnite(nnite)
    real(r8), intent(out), target :: tauxar(pcols,0:pver,nswbands)
    real(r8), intent(out), target :: wa(pcols,0:pver,nswbands)
    real(r8), intent(out), target :: ga(pcols,0:pver,nswbands)
    real(r8), intent(out), target :: fa(pcols,0:pver,nswbands)

    real(c_double), allocatable, target :: pdeldry(:,:), specmmr(:,:,:,:), dgnumwet(:,:,:), qaerwat(:,:,:)
    real(c_double), allocatable, target :: pmid(:,:), t(:,:), dlev(:,:,:), dcol(:,:)
    integer(c_int32_t), target :: status(pcols), troplevchem(pcols)
    integer(c_int32_t) :: ncol_c(1)
    type(colphys_modal_aero_io) :: io
    integer :: rc, ncol, lchnk

    ncol = state%ncol; lchnk = state%lchnk
    allocate(pdeldry(pcols,pver), specmmr(pcols,pver,max(nspec_max,1),max(nmodes_clim,1)), &
             dgnumwet(pcols,pver,max(nmodes_clim,1)), qaerwat(pcols,pver,max(nmodes_clim,1)), &
             pmid(pcols,pver), t(pcols,pver), dlev(pcols,pver,nlevf), dcol(pcols,ncolf0+3*max(nmodes_clim,1)))
    call marshal_inputs('modal_aero_sw', list_idx, state, pbuf, pdeldry, specmmr, dgnumwet, qaerwat)
    pmid = 1._r8; t = 1._r8; troplevchem = 1
    pmid(:ncol,:) = state%pmid(:ncol,:); t(:ncol,:) = state%t(:ncol,:)
    call tropopause_findChemTrop(state, troplevchem)                                ! :679
    status = 0
    io%pdeldry = c_loc(pdeldry); io%pmid = c_loc(pmid); io%t = c_loc(t)
    io%specmmr = c_loc(specmmr); io%dgnumwet = c_loc(dgnumwet); io%qaerwat = c_loc(qaerwat)
    io%troplevchem = c_loc(troplevchem)
    io%tauxar = c_loc(tauxar); io%wa = c_loc(wa); io%ga = c_loc(ga); io%fa = c_loc(fa)
    io%status = c_loc(status)
    ncol_c(1) = ncol
    rc = colphys_modal_aero_sw_diag_chunks(1, pcols, pver, ncol_c, io, c_loc(dlev), c_loc(dcol))
    if (rc /= COLPHYS_OK) call endrun('modal_aero_sw (colphys): '//trim(colphys_error_message()))
    call check_status('modal_aero_sw', lchnk, ncol, status, nerr_dopaer_sw, -1)

    call sw_history_out(lchnk, ncol, nnite, idxnite, dlev, dcol)
  end subroutine modal_aero_sw

  !---------------------------------------------------------------------------------------------------------
  ! What modal_aero_sw hands to outfld (:1046-1209), from the engine's packed diagnostics: dlev (pcols,pver,COLPHYS_SWD_NLEV),
  ! dcol (pcols, COLPHYS_SWD_NCOL + 3 nmodes).  Night-column fill and the order of the calls are the reference's.
  subroutine sw_history_out(lchnk, ncol, nnite, idxnite, dlev, dcol)
    integer, intent(in) :: lchnk, ncol, nnite
    integer, intent(in) :: idxnite(nnite)
    real(r8), intent(inout) :: dlev(pcols,pver,nlevf), dcol(pcols,ncolf0+3*max(nmodes_clim,1))
    character(len=16), parameter :: lev_names(nlevf) = (/ 'EXTINCT   ', 'ABSORB    ', 'EXTxASYM  ', 'EXTINCTUV ', 'EXTINCTNIR' /)
    character(len=16), parameter :: col_names(ncolf0) = (/ &
         'AODVIS       ', 'AODABS       ', 'AODTOT       ', 'AODVISst     ', 'SSAVIS       ', 'AODxASYM     ', 'AODUV        ', &
         'AODNIR       ', 'AODUVst      ', 'AODNIRst     ', 'BURDENSO4    ', 'BURDENPOM    ', 'BURDENSOA    ', 'BURDENBC     ', &
         'BURDENDUST   ', 'BURDENSEASALT', 'AODABSBC     ', 'AODSO4       ', 'AODPOM       ', 'AODSOA       ', 'AODBC        ', &
         'AODDUST      ', 'AODSS        ' /)
    ! outfld order of the reference: the seven summed visible-band fields (:1094-1118), then the climate-list set (:1130-1209)
    integer, parameter :: first_set(7)  = (/ 101, 102, 1, 2, 3, 4, 103 /)            ! 100 + level field, else column field
    integer, parameter :: first_set_day(7) = (/ 101, 102, 1, 3, 2, 4, 103 /)         ! the day-only set swaps AODTOT and AODABS (:1112-1118)
    integer, parameter :: second_set(23) = (/ 5, 6, 104, 105, 7, 8, 9, 10, 15, 11, 12, 13, 14, 16, 17, 22, 18, 19, 20, 21, 23, 0, 0 /)

    integer :: m, i, j
    character(len=32) :: outname, modename

    do m = 1, nmodes_clim                                                           ! :1046-1074
      call rad_cnst_get_info(0, m, mode_type=modename)
      j = ncolf0 + 3*(m-1)
      write(outname,'(a,i2.2)') 'BURDENdn', m;  call outfld(trim(outname), dcol(:,j+1), pcols, lchnk)
      outname = 'AODdn_'//trim(modename);       call outfld(trim(outname), dcol(:,j+2), pcols, lchnk)
      write(outname,'(a,i2.2)') 'AODdnDUST', m; call outfld(trim(outname), dcol(:,j+3), pcols, lchnk)
      do i = 1, nnite
        dcol(idxnite(i), j+1:j+3) = fillvalue
      end do
      write(outname,'(a,i2.2)') 'BURDEN', m;    call outfld(trim(outname), dcol(:,j+1), pcols, lchnk)
      outname = 'AOD_'//trim(modename);         call outfld(trim(outname), dcol(:,j+2), pcols, lchnk)
      write(outname,'(a,i2.2)') 'AODDUST', m;   call outfld(trim(outname), dcol(:,j+3), pcols, lchnk)
    end do

    call put_set(first_set, 7, 'dn')                                                ! :1094-1100
    do i = 1, nnite                                                                 ! :1102-1110
      dlev(idxnite(i),:,1:3) = fillvalue
      dcol(idxnite(i),1:4)   = fillvalue
    end do
    call put_set(first_set_day, 7, '')                                              ! :1112-1118

    do i = 1, ncol                                                                  ! :1122-1128: aodvis is night-filled by now
      if (dcol(i,1) > 1.e-10_r8) then
        dcol(i,5) = dcol(i,5)/dcol(i,1)
      else
        dcol(i,5) = 0.925_r8
      end if
    end do
    call put_set(second_set, 21, 'dn')                                              ! :1130-1154
    do i = 1, nnite                                                                 ! :1157-1183
      dlev(idxnite(i),:,4:5) = fillvalue
      dcol(idxnite(i),5:ncolf0) = fillvalue
    end do
    call put_set(second_set, 21, '')                                                ! :1185-1209

  contains
    subroutine put_set(set, n, suffix)
      integer, intent(in) :: set(:), n
      character(len=*), intent(in) :: suffix
      integer :: q
      do q = 1, n
        if (set(q) > 100) then
          call outfld(trim(lev_names(set(q)-100))//suffix, dlev(:,:,set(q)-100), pcols, lchnk)
        else
          call outfld(trim(col_names(set(q)))//suffix, dcol(:,set(q)), pcols, lchnk)
        end if
      end do
    end subroutine put_set
  end subroutine sw_history_out

  !---------------------------------------------------------------------------------------------------------
  subroutine modal_aero_lw(list_idx, state, pbuf, tauxar)
    ! modal_aer_opt.F90:1216-1469
    integer,             intent(in)  :: list_idx
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8), intent(out), target :: tauxar(pcols,pver,nlwbands)

    real(c_double), allocatable, target :: pdeldry(:,:), specmmr(:,:,:,:), dgnumwet(:,:,:), qaerwat(:,:,:)
    integer(c_int32_t), target :: status(pcols)
    integer(c_int32_t) :: ncol_c(1)
    type(colphys_modal_aero_io) :: io
    integer :: rc

    allocate(pdeldry(pcols,pver), specmmr(pcols,pver,max(nspec_max,1),max(nmodes_clim,1)), &
             dgnumwet(pcols,pver,max(nmodes_clim,1)), qaerwat(pcols,pver,max(nmodes_clim,1)))
    call marshal_inputs('modal_aero_lw', list_idx, state, pbuf, pdeldry, specmmr, dgnumwet, qaerwat)
    status = 0
    io%pdeldry = c_loc(pdeldry); io%pmid = c_null_ptr; io%t = c_null_ptr
    io%specmmr = c_loc(specmmr); io%dgnumwet = c_loc(dgnumwet); io%qaerwat = c_loc(qaerwat)
    io%troplevchem = c_null_ptr
    io%tauxar = c_loc(tauxar); io%wa = c_null_ptr; io%ga = c_null_ptr; io%fa = c_null_ptr
    io%status = c_loc(status)
    ncol_c(1) = state%ncol
    rc = colphys_modal_aero_lw_chunks(1, pcols, pver, ncol_c, io)
    if (rc /= COLPHYS_OK) call endrun('modal_aero_lw (colphys): '//trim(colphys_error_message()))
    call check_status('modal_aero_lw', state%lchnk, state%ncol, status, nerr_dopaer_lw, nerrmax_dopaer)
  end subroutine modal_aero_lw

  !=========================================================================================================
  ! Whole-rank dispatch of the optics (the host face, include/colphys.h colphys_hostface_*; DESIGN.md 3).
  !
  ! What the radiation driver's chunk loop becomes (upstream aer_rad_props calls modal_aero_sw / modal_aero_lw per chunk,
  ! modal_aer_opt.F90:485, 1216):
  !     call modal_aero_rank_init(ncol)                                   ! once, after modal_aer_opt_init
  !     do c: call modal_aero_rank_push(c, state(c), pbuf_c)              ! the accessors of :679-681, 717-762 per chunk; copies only
  !     call modal_aero_rank_run()                                        ! ONE dispatch: SW, its history diagnostics, LW, all chunks
  !     do c: call modal_aero_rank_pull_sw(c, state(c), nnite, idxnite, tauxar, wa, ga, fa)   ! + the outfld calls of :1046-1209
  !           call modal_aero_rank_pull_lw(c, state(c), tauxar_lw)
  ! Records are the (pcols, nlev) arrays of the chunk in pinned memory the library owns; pushes and pulls of distinct chunks
  ! may run concurrently (the counters nerr_dopaer_* are shared, as the reference's SAVEd ones are).
  !=========================================================================================================
  subroutine rank_fail(what)
    character(len=*), intent(in) :: what
    call endrun('modal_aero_rank (colphys): '//what//': '//trim(colphys_error_message()))
  end subroutine rank_fail

  subroutine modal_aero_rank_init(ncol)
    integer, intent(in) :: ncol(:)                       ! active columns of the rank's chunks, begchunk .. endchunk
    integer(c_int32_t) :: nc(size(ncol)), nf, nlev, is_int, is_out
    integer(c_int64_t) :: off
    character(kind=c_char) :: cname(32)
    character(len=32) :: name
    integer :: i, j, m, l
    if (nmodes_clim <= 0) call endrun('modal_aero_rank_init (colphys): modal_aer_opt_init has not been called')
    if (c_associated(hf)) call modal_aero_rank_final()
    nc = ncol
    nchunks_rank = size(ncol)
    allocate(ncol_rank(nchunks_rank)); ncol_rank = ncol
    allocate(off_dgn(nmodes_clim), off_qaw(nmodes_clim), off_mmr(max(nspec_max,1), nmodes_clim))
    off_dgn = -1; off_qaw = -1; off_mmr = -1
    if (colphys_hostface_create(nchunks_rank, pcols, pver, nc, COLPHYS_HF_AERO + COLPHYS_HF_SWDIAG, 0, hf) /= 0) call rank_fail('colphys_hostface_create')
    if (colphys_hostface_geometry(hf, in_rec, out_rec, nf) /= 0) call rank_fail('colphys_hostface_geometry')
    do i = 1, nf
      if (colphys_hostface_field(hf, i - 1, cname, nlev, is_int, is_out, off) /= 0) call rank_fail('colphys_hostface_field')
      name = ' '
      do j = 1, 32
        if (cname(j) == c_null_char) exit
        name(j:j) = cname(j)
      end do
      select case (trim(name))
      case ('t');           off_t = off
      case ('pmid');        off_pmid = off
      case ('pdeldry');     off_pdeldry = off
      case ('troplevchem'); off_troplev = off
      case ('sw_tauxar');   off_sw(1) = off
      case ('sw_wa');       off_sw(2) = off
      case ('sw_ga');       off_sw(3) = off
      case ('sw_fa');       off_sw(4) = off
      case ('lw_tauxar');   off_lw = off
      case ('sw_status');   off_swst = off
      case ('lw_status');   off_lwst = off
      case ('swd_lev');     off_swdlev = off
      case ('swd_col');     off_swdcol = off
      case default
        do m = 1, nmodes_clim
          if (trim(name) == 'dgnumwet_'//trim(istr(m))) off_dgn(m) = off
          if (trim(name) == 'qaerwat_'//trim(istr(m)))  off_qaw(m) = off
          do l = 1, nspec_m(m)
            if (trim(name) == 'specmmr_'//trim(istr(m))//'_'//trim(istr(l))) off_mmr(l, m) = off
          end do
        end do
      end select
    end do
    if (min(off_t, off_pmid, off_pdeldry, off_troplev, off_lw, off_swst, off_lwst, off_swdlev, off_swdcol) < 0 .or. any(off_sw < 0) .or. &
        any(off_dgn < 0) .or. any(off_qaw < 0)) call endrun('modal_aero_rank_init (colphys): the library does not offer every field this face moves')
    do m = 1, nmodes_clim
      if (any(off_mmr(:nspec_m(m), m) < 0)) call endrun('modal_aero_rank_init (colphys): a species record is missing')
    end do
    in_base = transfer(colphys_hostface_in_base(hf), in_base)
    out_base = transfer(colphys_hostface_out_base(hf), out_base)
  contains
    function istr(n) result(t)
      integer, intent(in) :: n
      character(len=8) :: t
      write(t, '(i0)') n
    end function istr
  end subroutine modal_aero_rank_init

  subroutine modal_aero_rank_final()
    integer(c_int) :: rc
    if (c_associated(hf)) rc = colphys_hostface_destroy(hf)
    hf = c_null_ptr
    if (allocated(ncol_rank)) deallocate(ncol_rank)
    if (allocated(off_dgn)) deallocate(off_dgn, off_qaw, off_mmr)
    nchunks_rank = 0
  end subroutine modal_aero_rank_final

  function rec_in(off, c) result(addr)
    integer(c_int64_t), intent(in) :: off
    integer, intent(in) :: c
    type(c_ptr) :: addr
    addr = transfer(in_base + int(c - 1, c_intptr_t) * in_rec + off, addr)
  end function rec_in
  function rec_out(off, c) result(addr)
    integer(c_int64_t), intent(in) :: off
    integer, intent(in) :: c
    type(c_ptr) :: addr
    addr = transfer(out_base + int(c - 1, c_intptr_t) * out_rec + off, addr)
  end function rec_out

  ! chunk c's optics inputs into its record: the same accessor calls as marshal_inputs, whole (pcols,pver) arrays copied as
  ! they are (padding columns travel and are never read on the device)
  subroutine modal_aero_rank_push(c, state, pbuf)
    use tropopause, only: tropopause_findChemTrop
    integer, intent(in) :: c
    type(physics_state), intent(in), target :: state
    type(physics_buffer_desc), pointer :: pbuf(:)
    real(r8), pointer :: p2(:,:), dgnumwet_m(:,:,:), qaerwat_m(:,:,:), mmr(:,:)
    integer(c_int32_t), pointer :: pi(:)
    integer(c_int32_t) :: troplevchem(pcols)
    integer :: m, l
    if (.not. c_associated(hf)) call endrun('modal_aero_rank_push (colphys): modal_aero_rank_init has not been called')
    if (state%ncol /= ncol_rank(c)) call endrun('modal_aero_rank_push (colphys): state%ncol differs from the count given to modal_aero_rank_init')
    call c_f_pointer(rec_in(off_pdeldry, c), p2, [pcols, pver]); p2 = state%pdeldry(:pcols, :pver)
    call c_f_pointer(rec_in(off_pmid, c), p2, [pcols, pver]);    p2 = state%pmid(:pcols, :pver)
    call c_f_pointer(rec_in(off_t, c), p2, [pcols, pver]);       p2 = state%t(:pcols, :pver)
    call pbuf_get_field(pbuf, dgnumwet_idx, dgnumwet_m)                              ! :679-681 / :1320-1322
    call pbuf_get_field(pbuf, qaerwat_idx,  qaerwat_m)
    do m = 1, nmodes_clim
      call c_f_pointer(rec_in(off_dgn(m), c), p2, [pcols, pver]); p2 = dgnumwet_m(:pcols, :pver, m)
      call c_f_pointer(rec_in(off_qaw(m), c), p2, [pcols, pver]); p2 = qaerwat_m(:pcols, :pver, m)
      do l = 1, nspec_m(m)
        call rad_cnst_get_aer_mmr(0, m, l, 'a', state, pbuf, mmr)                   ! :759 / :1353
        call c_f_pointer(rec_in(off_mmr(l, m), c), p2, [pcols, pver]); p2 = mmr(:pcols, :pver)
      end do
    end do
    troplevchem = 1
    call tropopause_findChemTrop(state, troplevchem)                                ! :679
    call c_f_pointer(rec_in(off_troplev, c), pi, [pcols]); pi = troplevchem
  end subroutine modal_aero_rank_push

  subroutine modal_aero_rank_run()
    if (.not. c_associated(hf)) call endrun('modal_aero_rank_run (colphys): modal_aero_rank_init has not been called')
    if (colphys_hostface_run(hf, 0._c_double, COLPHYS_DYCORE_FV, 0._c_double) /= 0) call rank_fail('colphys_hostface_run')
  end subroutine modal_aero_rank_run

  ! chunk c's modal_aero_sw results: the four radiative arrays for 1:ncol, the sanity check of :997-1029, the history fields
  subroutine modal_aero_rank_pull_sw(c, state, nnite, idxnite, tauxar, wa, ga, fa)
    integer, intent(in) :: c
    type(physics_state), intent(in) :: state
    integer, intent(in) :: nnite
    integer, intent(in) :: idxnite(nnite)
    real(r8), intent(out) :: tauxar(pcols,0:pver,nswbands), wa(pcols,0:pver,nswbands), ga(pcols,0:pver,nswbands), fa(pcols,0:pver,nswbands)
    real(r8), pointer :: p3(:,:,:), p2(:,:)
    integer(c_int32_t), pointer :: st(:)
    real(r8), allocatable :: dlev(:,:,:), dcol(:,:)
    integer :: n
    n = state%ncol
    call c_f_pointer(rec_out(off_sw(1), c), p3, [pcols, pver + 1, nswbands]); tauxar(:n,:,:) = p3(:n,:,:)
    call c_f_pointer(rec_out(off_sw(2), c), p3, [pcols, pver + 1, nswbands]); wa(:n,:,:) = p3(:n,:,:)
    call c_f_pointer(rec_out(off_sw(3), c), p3, [pcols, pver + 1, nswbands]); ga(:n,:,:) = p3(:n,:,:)
    call c_f_pointer(rec_out(off_sw(4), c), p3, [pcols, pver + 1, nswbands]); fa(:n,:,:) = p3(:n,:,:)
    call c_f_pointer(rec_out(off_swst, c), st, [pcols])
    call check_status('modal_aero_sw', state%lchnk, n, st, nerr_dopaer_sw, -1)
    allocate(dlev(pcols,pver,nlevf), dcol(pcols,ncolf0+3*nmodes_clim))
    call c_f_pointer(rec_out(off_swdlev, c), p3, [pcols, pver, nlevf]);       dlev = p3
    call c_f_pointer(rec_out(off_swdcol, c), p2, [pcols, ncolf0+3*nmodes_clim]); dcol = p2
    call sw_history_out(state%lchnk, n, nnite, idxnite, dlev, dcol)
  end subroutine modal_aero_rank_pull_sw

  subroutine modal_aero_rank_pull_lw(c, state, tauxar)
    integer, intent(in) :: c
    type(physics_state), intent(in) :: state
    real(r8), intent(out) :: tauxar(pcols,pver,nlwbands)
    real(r8), pointer :: p3(:,:,:)
    integer(c_int32_t), pointer :: st(:)
    integer :: n
    n = state%ncol
    call c_f_pointer(rec_out(off_lw, c), p3, [pcols, pver, nlwbands]); tauxar(:n,:,:) = p3(:n,:,:)
    call c_f_pointer(rec_out(off_lwst, c), st, [pcols])
    call check_status('modal_aero_lw', state%lchnk, n, st, nerr_dopaer_lw, nerrmax_dopaer)
  end subroutine modal_aero_rank_pull_lw

end module AER_MODULE
