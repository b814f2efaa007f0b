! oslo_aero_microp_hip -- the OSLO-aerosol microphysics hook of BASELINE config 4 ("tphysbc-lite: ZM + modal_aer_opt +
! oslo_aero_microp") on the engine's side of the boundary.
!
! In NorESMhub/CAM-Nor-physics the hook is a placeholder for the external OSLO_AERO package: three public procedures with
! empty bodies (physics/oslo_aero_microp.F90:10-25) behind the two switches of physics/oslo_aero_share.F90:8-9
! (`use_oslo_aero = .false.`, `nbmodes = 0`); tphysac would call the run procedure inside its macro/microphysics sub-stepping
! only `if (use_oslo_aero)` (physpkg.F90:1801-1804).  It contributes no arithmetic, so there is nothing to put on the device:
! the engine's counterpart is the same surface -- same module and procedure names (with -DCOLPHYS_DROPIN_NAMES), same dummy
! arguments, kinds and intents -- doing the same nothing, so that a physpkg.F90 that `use`s it links unchanged whichever
! side provides the module.  ptend_all is intent(out) and is left default-initialised, as in the reference (no field
! flagged, nothing allocated).
#ifdef COLPHYS_DROPIN_NAMES
#define MICROP_MODULE oslo_aero_microp
#define SHARE_MODULE  oslo_aero_share
#else
#define MICROP_MODULE oslo_aero_microp_hip
#define SHARE_MODULE  oslo_aero_share_hip
#endif

module SHARE_MODULE
  implicit none
  public
  integer :: nbmodes       = 0           ! oslo_aero_share.F90:8
  logical :: use_oslo_aero = .false.     ! oslo_aero_share.F90:9
end module SHARE_MODULE

module MICROP_MODULE
  implicit none
  private
  public :: oslo_aero_microp_register, oslo_aero_microp_init, oslo_aero_microp_run
contains

  subroutine oslo_aero_microp_register()
    ! nothing to register: no pbuf field belongs to the hook
  end subroutine oslo_aero_microp_register

  subroutine oslo_aero_microp_init()
  end subroutine oslo_aero_microp_init

  subroutine oslo_aero_microp_run(state, ptend_all, deltatin, pbuf)
    use shr_kind_mod,   only: r8 => shr_kind_r8
    use physics_types,  only: physics_state, physics_ptend
    use physics_buffer, only: physics_buffer_desc
    type(physics_state),       intent(in)  :: state
    type(physics_ptend),       intent(out) :: ptend_all
    real(r8),                  intent(in)  :: deltatin
    type(physics_buffer_desc), pointer     :: pbuf(:)
    ! no tendency: the whole-node dispatch has nothing to enqueue for this hook
  end subroutine oslo_aero_microp_run

end module MICROP_MODULE
