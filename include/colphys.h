/* colphys.h -- C ABI of the MI355X column-physics engine (libcolphys_hip.so).
 *
 * Drop-in boundary for ONE hot path of NorESMhub/CAM-Nor-physics:
 *
 *   entry point                      replaces (reference file:line)
 *   -------------------------------  -------------------------------------------------------------
 *   colphys_init                     zm_convi                 physics/zm_conv.F90:115-227
 *                                    (+ the physconst values zm_conv.F90:19-20 `use`s, + xrmin/xrmax
 *                                     of modal_aer_opt_init   physics/modal_aer_opt.F90:140-143)
 *   colphys_geopotential_t*          geopotential_t           physics/geopotential.F90:153-310
 *   colphys_geopotential_dse*        geopotential_dse         physics/geopotential.F90:29-150
 *   colphys_zm_convr*                zm_convr                 physics/zm_conv.F90:231-1709
 *                                    (buoyan_dilute :4425, parcel_dilute :4824, cldprp :3024,
 *                                     closure :4028, q1q2_pjr :4262, entropy/enthalpy/Brent :5280-5570)
 *   colphys_zm_conv_evap*            zm_conv_evap             physics/zm_conv.F90:1712-1972
 *   colphys_momtran*                 momtran                  physics/zm_conv.F90:2315-2715
 *   colphys_convtran*                convtran                 physics/zm_conv.F90:1976-2311
 *   colphys_convect_diagnostics_dev  convect_diagnostics_calc physics/convect_diagnostics.F90:115-249
 *   colphys_derived_fields_dev       d_p_coupling, 'derived_fields' block   fv/dp_coupling.F90:550-629
 *   colphys_dry_mass_fields_dev      ... its tail: set_state_pdry, set_wet_to_dry, qneg3   fv/dp_coupling.F90:630-651,
 *                                    physics/physics_types.F90:1925-1984
 *   colphys_zm_conv_tend_2_dev       zm_conv_tend_2           physics/zm_conv_intr.F90:955-1028 (and the convtran1
 *                                    step of zm_conv_tend, :860-884) on resident data
 *   colphys_aero_tables_set          rad_cnst_get_mode_props / rad_cnst_get_aer_props data consumed at
 *                                    physics/modal_aer_opt.F90:717-721,759-762,1334-1355 and the water
 *                                    refractive indices crefwsw/crefwlw :59-60,1525-1530
 *   colphys_modal_aero_sw*           modal_aero_sw            physics/modal_aer_opt.F90:485-1212
 *   colphys_modal_aero_lw*           modal_aero_lw            physics/modal_aer_opt.F90:1216-1469
 *   colphys_modal_aero_sw_diag*      modal_aero_sw's history diagnostics   physics/modal_aer_opt.F90:647-678, 770-831, 896-993, 1046-1154
 *
 * Two families of entry points:
 *
 *   *_dev   : every pointer is a DEVICE pointer, data already resident in HBM, layout "column-contiguous
 *             SoA": one buffer per field, shape [ncol][pver] (interfaces [ncol][pver+1]), the column's
 *             profile contiguous.  `stream` is a hipStream_t passed as void*.  Asynchronous: returns after
 *             enqueueing.  This is the whole-node dispatch (every grid column one work item).
 *
 *   *_chunks: HOST pointers in the reference's own chunk layout -- Fortran (pcols,pver[,..],nchunks),
 *             column index fastest -- as the ISO_C_BINDING shim (cam-nor-physics_amd/fortran/) passes the
 *             physics_state / ptend / pbuf arrays of one or many chunks.  Packs exactly ncol(c) columns
 *             per chunk (padding holds +inf in CAM: physics_types.F90:2143-2178), copies to the device,
 *             runs the *_dev path, copies back, synchronises.
 *
 * Conventions
 *   - fp64 throughout; level indices are int32, 1-based as in the reference.
 *   - Gathered outputs (mu, md, du, eu, ed, dp, dsubcld, jt, maxg, ideep) keep the reference's PER-CHUNK
 *     compaction (zm_conv.F90:926-940): gathered slot j (0-based) of chunk c lives in row c*pcols + j,
 *     ideep holds the 1-based in-chunk column index in ascending order, lengath[c] the count.
 *     Slots >= lengath[c] are zero-filled (the reference leaves them stale; nothing reads them:
 *     zm_conv_intr.F90:700-706,1006-1017).
 *   - Return value 0 = ok, < 0 = argument/runtime error (colphys_last_error()), and the per-column
 *     status[] array carries the conditions the reference turns into endrun():
 *       bit 0 (1): ientropy/ienthalpy did not converge  (zm_conv.F90:5401-5410, 5557-5566); bit 3 (8) with it when an
 *                  ienthalpy inversion was among them (the reference meets those first and names the routine in its message)
 *       bit 1 (2): aerosol optical depth of a layer outside its sanity range (modal_aer_opt.F90:997-1029 SW, 1402-1437 LW);
 *                  bit 2 (4) with it when one was negative (< -1e-10); status >> 8 counts the (mode, band, level) units of the
 *                  column outside the range.  The reference only WARNS about a large value and adds it to nerr_dopaer; it halts
 *                  on a negative one (SW, LW) and once nerr_dopaer reaches 1000 (LW) -- the shim's decision, from these fields.
 *   - No entry point retains caller pointers past its return (the *_dev ones: past stream completion).
 *   - There is no CPU fallback: every compute entry point fails with COLPHYS_ERR_NODEVICE when no HIP
 *     device is present.
 *   - Threads.  The *_chunks entry points may be called concurrently from several host threads (the reference calls the path
 *     from inside its OpenMP loop over chunks, physpkg.F90:1147): they share staging buffers and device work arrays, so the
 *     library runs such calls one after another.  The *_dev entry points only enqueue; calls that use the library's grow-only
 *     work arrays (zm_convr, zm_conv_tend, convtran / zm_conv_tend_2) must be ordered by the caller -- one stream, or events
 *     between streams -- the others (geopotential, zm_conv_evap, momtran, the optics, the updates) are free of shared state.
 *     colphys_init / colphys_aero_tables_set / colphys_finalize are not to overlap with compute calls.  Every call leaves
 *     the calling thread on the engine's HIP device; colphys_last_error() is per calling thread.  The per-kernel timers
 *     (colphys_profile_enable) are a measurement mode for one calling thread.
 */
#ifndef COLPHYS_H
#define COLPHYS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COLPHYS_ABI_VERSION 4

#define COLPHYS_OK             0
#define COLPHYS_ERR_ARG      (-1)
#define COLPHYS_ERR_NODEVICE (-2)
#define COLPHYS_ERR_HIP      (-3)
#define COLPHYS_ERR_NOTINIT  (-4)
#define COLPHYS_ERR_UNSUPPORTED (-5)

#define COLPHYS_STATUS_BRENT_NOT_CONVERGED 1
#define COLPHYS_STATUS_DOPAER_RANGE        2
#define COLPHYS_STATUS_DOPAER_NEGATIVE     4      /* set with DOPAER_RANGE: an optical depth < -1e-10 (the reference halts on these) */
#define COLPHYS_STATUS_BRENT_ENTHALPY      8      /* set with BRENT_NOT_CONVERGED: an ienthalpy inversion failed (they come first) */
#define COLPHYS_STATUS_COUNT_SHIFT         8      /* status >> 8: number of out-of-range optical depths of the column (nerr_dopaer) */

#define COLPHYS_MAX_PVER 128

/* dycore_kind for geopotential_t (geopotential.F90:217,222,227) */
#define COLPHYS_DYCORE_EUL 0   /* hkl = pdel/pmid,            tvfac = 1 + zvir*q                    */
#define COLPHYS_DYCORE_FV  1   /* 'LR'/'FV3': hkl = dlnp,     tvfac = 1 + zvir*q                    */
#define COLPHYS_DYCORE_SE  2   /* 'SE'/'MPAS': hkl = pdel/pmid, generalised virtual temperature     */

/* Everything zm_convi copies into SAVEd module variables (zm_conv.F90:42-108,145-183), the physconst
 * values the module `use`s (:19-20), and the module-level tunables with their defaults (:83-97,103).
 * Filled by the host from ITS physconst / namelist -- nothing here is hard-coded in the kernels. */
typedef struct colphys_zm_params {
    /* physconst */
    double cpair, epsilo, gravit, latice, latvap, tmelt, rair, cpwv, cpliq, rh2o, cpvir, zvir;
    /* zm_convi arguments */
    double c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, tiedke_add, capelmt, dmpdz_param, tau;
    /* module tunables (defaults in colphys_zm_params_defaults) */
    double tiedke_lnd, entrmn, alfadet, tentrm, plclmin, cin_threshd, parcel_hscale;
    int32_t limcnv, num_cin, zm_org, zmconv_microp, no_deep_pbl, lparcel_pbl;
    int32_t cam3;           /* cam_physpkg_is('cam3'): non-dilute buoyan; unsupported (zm_conv.F90:871) */
    int32_t reserved;
} colphys_zm_params;

/* module-level defaults of zm_conv.F90:83-97,103; tentrm = -dmpdz_param as zm_convi resets it (:213) */
void colphys_zm_params_defaults(colphys_zm_params *p);

int  colphys_abi_version(void);
const char *colphys_last_error(void);
int  colphys_device_count(void);
int  colphys_is_initialized(void);        /* 1 after a successful colphys_init, 0 before / after colphys_finalize */

/* Select the HIP device of this process (one process per GPU) and upload parameters; builds the
 * 1-K saturation table zm_conv_evap's qsat uses.  May be called again to change parameters. */
int  colphys_init(int device, const colphys_zm_params *p);
int  colphys_finalize(void);

/* ---------------------------------------------------------------------------------------------------
 * geopotential_t  (geopotential.F90:153-156)
 *   q is [nq][ncol][pver]; species 0 is water vapour; for COLPHYS_DYCORE_SE the first `nspecies`
 *   entries are the thermodynamically active species (air_composition).
 *   rair, zvir ([ncol][pver]: rairv, zvirv of physics_types.F90:393-415) may BOTH be NULL in the _dev entry: the engine then uses
 *   the rair / zvir of colphys_zm_params for every cell -- what the two fields hold everywhere unless WACCM-X varies the
 *   composition -- and the kernel reads two streams less (same bits as fields filled with those constants).
 * ------------------------------------------------------------------------------------------------- */
int colphys_geopotential_t_dev(int ncol, int pver, int dycore_kind, int nq, int nspecies,
                               const double *piln, const double *pmln, const double *pint,
                               const double *pmid, const double *pdel, const double *rpdel,
                               const double *t, const double *q, const double *rair, double gravit,
                               const double *zvir, double *zi, double *zm, void *stream);

int colphys_geopotential_t_chunks(int nchunks, int pcols, int pver, const int32_t *ncol,
                                  int dycore_kind, int nq, int nspecies,
                                  const double *piln, const double *pmln, const double *pint,
                                  const double *pmid, const double *pdel, const double *rpdel,
                                  const double *t, const double *q, const double *rair, double gravit,
                                  const double *zvir, double *zi, double *zm);

/* ---------------------------------------------------------------------------------------------------
 * geopotential_dse  (geopotential.F90:29-32): temperature and heights from dry static energy.
 *   Same conventions as geopotential_t; dse, cpair [ncol][pver], phis [ncol]; outputs t [ncol][pver], zi, zm.
 * ------------------------------------------------------------------------------------------------- */
int colphys_geopotential_dse_dev(int ncol, int pver, int dycore_kind, int nq, int nspecies,
                                 const double *piln, const double *pmln, const double *pint,
                                 const double *pmid, const double *pdel, const double *rpdel,
                                 const double *dse, const double *q, const double *phis, const double *rair, double gravit,
                                 const double *cpair, const double *zvir, double *t, double *zi, double *zm, void *stream);

int colphys_geopotential_dse_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, int dycore_kind, int nq, int nspecies,
                                    const double *piln, const double *pmln, const double *pint,
                                    const double *pmid, const double *pdel, const double *rpdel,
                                    const double *dse, const double *q, const double *phis, const double *rair, double gravit,
                                    const double *cpair, const double *zvir, double *t, double *zi, double *zm);

/* ---------------------------------------------------------------------------------------------------
 * zm_convr  (zm_conv.F90:231-244; intents :362-423)
 *   full-layout outputs  [ncol][pver]  : qtnd heat cme zdu rprd dlf ql dif dnlf dnif eurt
 *                        [ncol][pver+1]: mcon pflx
 *                        [ncol]        : prec jctop jcbot cape rliq rice
 *   gathered-layout outputs            : mu md du eu ed dp [ncol][pver]; dsubcld [ncol];
 *                                        jt maxg ideep [ncol] int32; lengath [nchunks]
 *   pcols = chunk width used for the gathered layout (ncol need not be a multiple of it).
 *   delt = 0.5*ztodt exactly as zm_conv_tend passes it (zm_conv_intr.F90:666).
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_zm_convr_io {
    /* in */
    const double *t, *qh, *pap, *dpp, *zm;        /* [ncol][pver]   */
    const double *paph, *zi;                      /* [ncol][pver+1] */
    const double *geos, *pblh, *tpert, *landfrac; /* [ncol]         */
    /* out, full layout */
    double *qtnd, *heat, *cme, *zdu, *rprd, *dlf, *ql, *dif, *dnlf, *dnif, *eurt; /* [ncol][pver] */
    double *mcon, *pflx;                          /* [ncol][pver+1] */
    double *prec, *jctop, *jcbot, *cape, *rliq, *rice; /* [ncol]    */
    /* out, gathered layout */
    double *mu, *md, *du, *eu, *ed, *dp;          /* [ncol][pver]   */
    double *dsubcld;                              /* [ncol]         */
    int32_t *jt, *maxg, *ideep;                   /* [ncol]         */
    int32_t *lengath;                             /* [nchunks]      */
    int32_t *status;                              /* [ncol]         */
} colphys_zm_convr_io;

int colphys_zm_convr_dev(int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io *io, void *stream);
int colphys_zm_convr_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double delt,
                            const colphys_zm_convr_io *io);

/* ---------------------------------------------------------------------------------------------------
 * zm_conv_evap  (zm_conv.F90:1712-1717); prdsnow absent => old_snow (:1789-1794)
 *   prec is in/out (m/s).
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_zm_conv_evap_io {
    const double *t, *pmid, *pdel, *q, *prdprec, *cldfrc;   /* [ncol][pver] */
    const double *landfrac;                                 /* [ncol]       */
    double *prec;                                           /* [ncol] inout */
    double *tend_s, *tend_s_snwprd, *tend_s_snwevmlt, *tend_q, *ntprprd, *ntsnprd; /* [ncol][pver] */
    double *flxprec, *flxsnow;                              /* [ncol][pver+1] */
    double *snow;                                           /* [ncol]       */
} colphys_zm_conv_evap_io;

int colphys_zm_conv_evap_dev(int ncol, int pver, double deltat, const colphys_zm_conv_evap_io *io, void *stream);
int colphys_zm_conv_evap_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double deltat,
                                const colphys_zm_conv_evap_io *io);

/* ---------------------------------------------------------------------------------------------------
 * momtran  (zm_conv.F90:2315-2319) for the two wind components (ncnst = 2, domomtran = .true.)
 *   winds and the 3-D outputs are [2][ncol][pver]; mass fluxes etc. are the gathered outputs of zm_convr.
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_momtran_io {
    const double *winds;                             /* [2][ncol][pver] */
    const double *mu, *md, *du, *eu, *ed, *dp;       /* gathered [ncol][pver] */
    const double *dsubcld;                           /* gathered [ncol] */
    const int32_t *jt, *mx, *ideep;                  /* gathered [ncol] */
    const int32_t *lengath;                          /* [nchunks] */
    double *dqdt, *pguall, *pgdall, *icwu, *icwd;    /* [2][ncol][pver] */
    double *seten;                                   /* [ncol][pver] */
} colphys_momtran_io;

int colphys_momtran_dev(int ncol, int pver, int pcols, double dt, const colphys_momtran_io *io, void *stream);
int colphys_momtran_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double dt,
                           const colphys_momtran_io *io);

/* ---------------------------------------------------------------------------------------------------
 * convtran  (zm_conv.F90:1976-1981): convective transport of tracers m = 2..ncnst on the gathered columns.
 *   q, fracis, dqdt are [ncnst][ncol][pver] (tracer 1 = water vapour is never transported, :2062);
 *   doconvtran[m], isdry[m] (cnst_get_type_byind(m) == 'dry', :2065) are HOST arrays of ncnst flags;
 *   mass fluxes etc. are the gathered outputs of zm_convr; dpdry is gathered too (zm_conv_intr.F90:1010-1017).
 *   dqdt(:,:,m) is defined (zero outside the gathered columns) for every tracer with doconvtran[m] != 0 and
 *   left untouched for the others, as in the reference (:2299).
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_convtran_io {
    const double *q, *fracis;                        /* [ncnst][ncol][pver] */
    const double *mu, *md, *du, *eu, *ed, *dp, *dpdry; /* gathered [ncol][pver] */
    const int32_t *jt, *mx, *ideep;                  /* gathered [ncol] */
    const int32_t *lengath;                          /* [nchunks] */
    const int32_t *doconvtran, *isdry;               /* HOST [ncnst] */
    double *dqdt;                                    /* [ncnst][ncol][pver] */
} colphys_convtran_io;

int colphys_convtran_dev(int ncol, int pver, int pcols, int ncnst, double dt, const colphys_convtran_io *io, void *stream);
int colphys_convtran_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, int ncnst, double dt,
                            const colphys_convtran_io *io);

/* ---------------------------------------------------------------------------------------------------
 * zm_conv_tend_2 on resident data  (zm_conv_intr.F90:955-1028): convective transport of every constituent with
 * lq[m] != 0 (cnst_is_convtran2) on the mass fluxes zm_conv_tend left in the physics buffer, ONE launch for all of them:
 *     lengath = count(ideep > 0) per chunk                                        :993
 *     dpdry(i,:) = state%pdeldry(ideep(i),:)/100 for the gathered columns         :1010-1017 (done on the fly)
 *     convtran(lq, state%q, ..., fracis, ptend%q, dpdry, ztodt)                   :1020-1024
 * The same call with lq = cnst_is_convtran1 is the "convtran1" step that closes zm_conv_tend (:860-884: cloud liquid
 * and ice, moist mixing ratios; the reference passes a zero dpdry there, so pdeldry may be NULL when no selected
 * constituent is of type 'dry').  q, fracis, ptend_q are [ncnst][ncol][pver] with water vapour at index 0 (never
 * transported, zm_conv.F90:2062); ptend_q(:,:,m) is defined for every selected m (zero outside the gathered columns) and
 * left untouched for the others.  lq, isdry are HOST arrays of ncnst flags.
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_zm_conv_tend_2_io {
    const double *q, *fracis;                        /* [ncnst][ncol][pver] state%q, pbuf FRACIS */
    const double *pdeldry;                           /* [ncol][pver] state%pdeldry [Pa] */
    const double *mu, *md, *du, *eu, *ed, *dp;       /* pbuf ZM_MU ... ZM_DP, gathered [ncol][pver] */
    const int32_t *jt, *maxg, *ideep;                /* pbuf ZM_JT, ZM_MAXG, ZM_IDEEP, gathered [ncol] */
    const int32_t *lq, *isdry;                       /* HOST [ncnst] */
    double *ptend_q;                                 /* [ncnst][ncol][pver] */
} colphys_zm_conv_tend_2_io;

int colphys_zm_conv_tend_2_dev(int ncol, int pver, int pcols, int ncnst, double ztodt, const colphys_zm_conv_tend_2_io *io,
                               void *stream);

/* ---------------------------------------------------------------------------------------------------
 * d_p_coupling's derived fields  (fv/dp_coupling.F90:550-625): what the dynamics-physics coupler derives
 * from pint / T / q before physics runs, on resident data, so that the fields the ZM path consumes are produced on the
 * device:
 *     pdel = pint(k+1) - pint(k), rpdel = 1/pdel, pmid = (pint(k) + pint(k+1))/2, lnpmid = log(pmid)      :552-559
 *     bottom-layer fix of negative constituents, every m: if q(pver,m) < qmin(m) and the level above can pay,
 *         q(pver,m) = qmin(m), q(pver-1,m) -= (qmin(m) - q(pver,m)) pdel(pver)/pdel(pver-1)               :565-590
 *     geopotential_t(lnpint, lnpmid, pint, pmid, pdel, rpdel, t, q, rairv, gravit, zvirv) -> zi, zm        :617-621
 *     s = cpairv t + gravit zm + phis                                                                      :624-629
 * q is [ncnst][ncol][pver], in/out; qmin a HOST array of ncnst lower bounds (constituents' qmin).
 * The block's tail -- set_state_pdry, set_wet_to_dry, qneg3 (:630-651) -- is colphys_dry_mass_fields_dev below.
 * Not included (host / upstream code): physics_cnst_limit, cam_thermo_dry_air_update (WACCM-X only), check_energy_timestep_init.
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_derived_fields_io {
    const double *pint, *lnpint;                     /* [ncol][pver+1] */
    const double *t;                                 /* [ncol][pver]   */
    double *q;                                       /* [ncnst][ncol][pver] in/out */
    const double *qmin;                              /* HOST [ncnst]   */
    const double *phis;                              /* [ncol]         */
    const double *cpair, *rair, *zvir;               /* [ncol][pver]: cpairv, rairv, zvirv */
    double *pdel, *rpdel, *pmid, *lnpmid;            /* [ncol][pver]   out */
    double *zi;                                      /* [ncol][pver+1] out */
    double *zm, *s;                                  /* [ncol][pver]   out */
} colphys_derived_fields_io;

int colphys_derived_fields_dev(int ncol, int pver, int ncnst, int dycore_kind, int nspecies, double gravit,
                               const colphys_derived_fields_io *io, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * The dry-mass tail of d_p_coupling's derived fields  (fv/dp_coupling.F90:630-651) on resident data:
 *     set_state_pdry  (physics_types.F90:1925-1964): pdeldry = pdel (1 - q(:,:,1)); pintdry(1) = pint(1),
 *         pintdry(k+1) = pintdry(k) + pdeldry(k); pmiddry = (pintdry(k+1) + pintdry(k))/2; psdry = pint(1) + sum pdeldry;
 *         rpdeldry = 1/pdeldry; lnpmiddry = log(pmiddry); lnpintdry = log(pintdry)
 *     set_wet_to_dry  (:1968-1984): q(:,:,m) = q(:,:,m) pdel/pdeldry for every constituent of type 'dry'
 *     qneg3 over 1..pcnst, then once more for cloud liquid and ice (:644-651; idempotent): q < qmin(m) -> qmin(m)
 * q is [ncnst][ncol][pver], in/out, water vapour first; isdry, qmin are HOST arrays of ncnst.
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_dry_mass_io {
    const double *pint;                              /* [ncol][pver+1] */
    const double *pdel;                              /* [ncol][pver]   */
    double *q;                                       /* [ncnst][ncol][pver] in/out */
    const int32_t *isdry;                            /* HOST [ncnst]: cnst_type(m) == 'dry' */
    const double *qmin;                              /* HOST [ncnst]   */
    double *pdeldry, *rpdeldry, *pmiddry, *lnpmiddry; /* [ncol][pver]   out */
    double *pintdry, *lnpintdry;                     /* [ncol][pver+1] out */
    double *psdry;                                   /* [ncol]         out */
} colphys_dry_mass_io;

int colphys_dry_mass_fields_dev(int ncol, int pver, int ncnst, const colphys_dry_mass_io *io, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * convect_diagnostics_calc  (convect_diagnostics.F90:115-249): merges the deep scheme's outputs with the shallow scheme's
 * on resident data, element-wise.
 *     shallow_is_clubb != 0 ('CLUBB_SGS', :178-189): cmfmc2, rprdsh, precc, icwmr, rliq2, qc2, evapcsh, snow are ZEROED
 *         first (each may be NULL), cnt2 = pver, cnb2 = 1 (the cnt2 / cnb2 arguments are ignored);
 *     cmfmc += cmfmc2 (:199); cnt = min(cnt, cnt2), cnb = max(cnb, cnb2), cnb == 1 -> cnb = cnt,
 *     pcnt = pmid(int(cnt)), pcnb = pmid(int(cnb)) (:210-216); rprdtot = rprdsh + rprddp (:223);
 *     qc += qc2, rliq += rliq2 (:230-231).
 * cnt / cnb are the real-valued level indices zm_convr returns as jctop / jcbot (pbuf CLDTOP / CLDBOT).
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_convect_diag_io {
    double *cmfmc;                                   /* [ncol][pver+1] in/out */
    double *qc;                                      /* [ncol][pver]   in/out */
    double *rliq, *cnt, *cnb;                        /* [ncol]         in/out */
    double *cmfmc2;                                  /* [ncol][pver+1] shallow mass flux (zeroed under CLUBB) */
    double *qc2, *rprdsh;                            /* [ncol][pver] */
    double *rliq2;                                   /* [ncol] */
    const double *cnt2, *cnb2;                       /* [ncol], shallow top / base (ignored under CLUBB) */
    double *precc, *snow;                            /* [ncol]         zeroed under CLUBB, untouched otherwise; may be NULL */
    double *icwmr, *evapcsh;                         /* [ncol][pver]   zeroed under CLUBB, untouched otherwise; may be NULL */
    const double *rprddp, *pmid;                     /* [ncol][pver] */
    double *rprdtot;                                 /* [ncol][pver] out */
    double *pcnt, *pcnb;                             /* [ncol] out */
} colphys_convect_diag_io;

int colphys_convect_diagnostics_dev(int ncol, int pver, int shallow_is_clubb, const colphys_convect_diag_io *io, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * physics_update, the part the path needs between its three kernels (physics_types.F90:292-340, 425-432;
 * called at zm_conv_intr.F90:739,806,836): apply tendencies over dt, out of place (state -> state1 is
 * physics_state_copy, zm_conv_intr.F90:579):
 *     t_out = t_in + s_tend*dt/cpair            (:427)
 *     q_out = max(q_in + q_tend*dt, qmin)       (:321 + qneg3 clip, :329; qneg3 itself is upstream code)
 *     u_out = u_in + u_tend*dt, v likewise      (:295, :303)
 * Any (in, tend, out) triple may be all-NULL to skip that field; in == out is allowed.
 * The caller follows with colphys_geopotential_t_dev, as physics_update does (:435-441).
 * ------------------------------------------------------------------------------------------------- */
int colphys_physics_update_dev(int ncol, int pver, double dt, double cpair, double qmin,
                               const double *t_in, const double *s_tend, double *t_out,
                               const double *q_in, const double *q_tend, double *q_out,
                               const double *u_in, const double *u_tend, double *u_out,
                               const double *v_in, const double *v_tend, double *v_out, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * modal aerosol optics  (modal_aer_opt.F90)
 *   Tables (host pointers, copied to the device once):
 *     sigmag[nmodes]; nspec[nmodes]; specdens, hygro [nmodes][nspec_max]; spectype [nmodes][nspec_max]
 *       (0 sulfate,1 p-organic,2 s-organic,3 black-c,4 dust,5 seasalt, -1 other);
 *     specrefindex_sw [nmodes][nspec_max][nswbands][2] (re,im); specrefindex_lw likewise;
 *     refrtabsw [nmodes][nswbands][prefr]; refitabsw [nmodes][nswbands][prefi]; (lw likewise)
 *     extpsw/abspsw/asmpsw [nmodes][nswbands][prefi][prefr][ncoef] (= Fortran (ncoef,prefr,prefi,nsw));
 *     absplw [nmodes][nlwbands][prefi][prefr][ncoef]; crefwsw [nswbands][2]; crefwlw [nlwbands][2].
 * ------------------------------------------------------------------------------------------------- */
#define COLPHYS_NCOEF 5
#define COLPHYS_PREFR 7
#define COLPHYS_PREFI 10

typedef struct colphys_aero_tables {
    int32_t nmodes, nspec_max, nswbands, nlwbands;
    int32_t top_lev;                 /* clim_modal_aero_top_lev, 1-based */
    int32_t idx_sw_diag, idx_uv_diag, idx_nir_diag;   /* radconstants, 1-based */
    double  rhoh2o, rga, rair;       /* physconst values modal_aer_opt uses (:18) */
    double  xrmin, xrmax;            /* log(rmmin), log(rmmax) (:140-143) */
    const int32_t *nspec, *spectype;
    const double *sigmag, *specdens, *hygro;
    const double *specrefindex_sw, *specrefindex_lw;
    const double *refrtabsw, *refitabsw, *refrtablw, *refitablw;
    const double *extpsw, *abspsw, *asmpsw, *absplw;
    const double *crefwsw, *crefwlw;
} colphys_aero_tables;

int colphys_aero_tables_set(const colphys_aero_tables *tb);

typedef struct colphys_modal_aero_io {
    const double *pdeldry, *pmid, *t;     /* [ncol][pver] (pmid, t only used by the SW diagnostics) */
    const double *specmmr;                /* [nmodes][nspec_max][ncol][pver] */
    const double *dgnumwet, *qaerwat;     /* [nmodes][ncol][pver] */
    const int32_t *troplevchem;           /* [ncol] (SW diagnostics only; may be NULL) */
    /* SW out: [ncol][nswbands][pver+1] (level index 0..pver as in the reference's (pcols,0:pver,nswbands)) */
    double *tauxar, *wa, *ga, *fa;
    /* LW out: [ncol][nlwbands][pver]; pass through tauxar (wa/ga/fa NULL) */
    int32_t *status;                      /* [ncol] */
    /* optional (NULL: not used): [3][nmodes][ncol][pver] -- per mode, at the visible band idx_sw_diag, the specific
     * extinction and absorption as they enter dopaer (modal_aer_opt.F90:881, 888-890: scaled by wetvol*rhoh2o, absorption
     * clamped to [0, extinction]) and the asymmetry factor (:883-887).  Written by colphys_modal_aero_sw_dev, read by
     * colphys_modal_aero_sw_diag_dev, which then does not evaluate the tables a second time. */
    double *vis_optics;
} colphys_modal_aero_io;

/* Which of the two kernel families serves modal_aero_sw / modal_aero_lw (same results to rounding; both are tested against
 * the same goldens).  AUTO = BANDS_OUTER when the tables have at most four modes, else GROUPED.  BANDS_OUTER keeps every
 * mode's size / volume state of a cell and walks the bands once: fastest alone (the call pattern of the radiation driver).
 * GROUPED walks modes outer / a band group inner with LDS accumulators: for LW it is 15 % slower alone but leaves the CUs'
 * LDS and wave slots freer, which can pay when the optics run BESIDE the ZM chain on a second stream and the chain is the
 * longer of the two (bench.py --no-sw-diag: 0.6 % per substep; with the SW diagnostics the optics stream is the longer one
 * and BANDS_OUTER wins by 1.1 %). */
#define COLPHYS_AERO_KERNEL_AUTO        0
#define COLPHYS_AERO_KERNEL_BANDS_OUTER 1
#define COLPHYS_AERO_KERNEL_GROUPED     2
int colphys_aero_kernel_select(int sw_kind, int lw_kind);

int colphys_modal_aero_sw_dev(int ncol, int pver, const colphys_modal_aero_io *io, void *stream);
int colphys_modal_aero_lw_dev(int ncol, int pver, const colphys_modal_aero_io *io, void *stream);
int colphys_modal_aero_sw_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io);
int colphys_modal_aero_lw_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io);

/* ---------------------------------------------------------------------------------------------------
 * modal_aero_sw history diagnostics  (modal_aer_opt.F90:647-678 initialisation, :770-831, :896-993 accumulation,
 * :1046-1154 the values handed to outfld; climate list, list_idx = 0)
 *
 * A second output set of modal_aero_sw, computed from the same inputs AFTER colphys_modal_aero_sw_dev has filled
 * io->tauxar (the all-band / UV / near-IR totals are read from it: tauxar(:,k,isw) is the sum over modes of dopaer).
 * io->pmid, io->t and io->troplevchem are required here.  The quantities that need the per-mode pabs / pasm / palb and the
 * species partition take each mode's visible-band optics from io->vis_optics when the SW call was given that buffer (the
 * fast path: no table is evaluated twice); with io->vis_optics NULL the visible band (idx_sw_diag) is re-evaluated per mode.
 *   lev  [COLPHYS_SWD_NLEV][ncol][pver]            level fields (1/m), zero above top_lev
 *   col  [COLPHYS_SWD_NCOL + 3*nmodes][ncol]       column fields; per mode m (0-based) at COLPHYS_SWD_NCOL + 3*m:
 *                                                   burden (BURDENdn<mm>), aodmode (AODdn_<mode>), dustaodmode (AODdnDUST<mm>)
 * These are the reference's "dn" (day+night) values; the night-column fill of the day-only twins (:1059-1063,
 * :1102-1110, :1157-1183) and the outfld calls stay in the Fortran shim.  SSAVIS is the sum of dopaer*palb (:929): the
 * reference divides it by aodvis AFTER aodvis' night columns have been overwritten with fillvalue (:1106, :1122-1128),
 * so that step belongs to the caller that owns the night-column list.
 * ------------------------------------------------------------------------------------------------- */
#define COLPHYS_SWD_LEV_EXTINCT     0   /* EXTINCTdn     */
#define COLPHYS_SWD_LEV_ABSORB      1   /* ABSORBdn      */
#define COLPHYS_SWD_LEV_ASYMEXT     2   /* EXTxASYMdn    */
#define COLPHYS_SWD_LEV_EXTINCTUV   3   /* EXTINCTUVdn   */
#define COLPHYS_SWD_LEV_EXTINCTNIR  4   /* EXTINCTNIRdn  */
#define COLPHYS_SWD_NLEV            5
#define COLPHYS_SWD_COL_AODVIS      0   /* AODVISdn      */
#define COLPHYS_SWD_COL_AODABS      1   /* AODABSdn      */
#define COLPHYS_SWD_COL_AODTOT      2   /* AODTOTdn      */
#define COLPHYS_SWD_COL_AODVISST    3   /* AODVISstdn    */
#define COLPHYS_SWD_COL_SSAVIS      4   /* SSAVISdn before its normalisation */
#define COLPHYS_SWD_COL_ASYMVIS     5   /* AODxASYMdn    */
#define COLPHYS_SWD_COL_AODUV       6   /* AODUVdn       */
#define COLPHYS_SWD_COL_AODNIR      7   /* AODNIRdn      */
#define COLPHYS_SWD_COL_AODUVST     8   /* AODUVstdn     */
#define COLPHYS_SWD_COL_AODNIRST    9   /* AODNIRstdn    */
#define COLPHYS_SWD_COL_BURDEN0     10  /* BURDEN{SO4,POM,SOA,BC,DUST,SEASALT}dn in the spectype order 0..5 */
#define COLPHYS_SWD_COL_AODABSBC    16  /* AODABSBCdn    */
#define COLPHYS_SWD_COL_AOD0        17  /* AOD{SO4,POM,SOA,BC,DUST,SS}dn in the spectype order 0..5 */
#define COLPHYS_SWD_NCOL            23
#define COLPHYS_SWD_NMODEF          3   /* per-mode column fields */

int colphys_modal_aero_sw_diag_dev(int ncol, int pver, const colphys_modal_aero_io *io, double *lev, double *col, void *stream);
/* host (chunk) form: runs modal_aero_sw AND its diagnostics on one upload.  io as for colphys_modal_aero_sw_chunks plus
 * pmid, t (pcols,pver,nchunks) and troplevchem (pcols,nchunks); lev (pcols,pver,nchunks,COLPHYS_SWD_NLEV);
 * col (pcols,nchunks,COLPHYS_SWD_NCOL + 3*nmodes) */
int colphys_modal_aero_sw_diag_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io,
                                      double *lev, double *col);

/* ---------------------------------------------------------------------------------------------------
 * zm_conv_tend on resident data  (zm_conv_intr.F90:441-905, the part between the pbuf lookups and convtran1)
 *
 * The whole deep-convection tendency of a rank in ONE call, state1 never leaving the device:
 *     zm_convr(state, delt = ztodt/2)                                            :662-675
 *     mcon -> kg/m2/s, FREQZM, ZMMU/ZMMD scatter, PCONVT/PCONVB                  :685-731
 *     ptend_all = ptend_loc; state1 = physics_update(state1, ptend_loc) + geopotential_t   :737-739
 *     zm_conv_evap(state1, rprd, cld, ztodt; prec inout, snow)                   :764-771
 *     evapcdp = dq/dt of evaporation; ptend_all += ptend_loc; physics_update + geopotential_t     :773, :802-806
 *     momtran(winds = state1 u, v; ztodt)                                        :822-827
 *     ptend_all += (wind tendencies, seten); physics_update + geopotential_t     :829-836
 * All arrays are device pointers in the engine's layout ([ncol][pver], interfaces [ncol][pver+1], [ncol]); q is
 * water vapour only.  Outputs under "diagnostics" and "state1" may be NULL.  Work arrays (ptend_loc, the kernels'
 * diagnostic outputs that zm_conv_tend only hands to outfld) live in an internal grow-only device arena.
 * The caller applies ptend_all to its own state afterwards, as tphysbc does (physpkg.F90:2838), and runs
 * convtran on the gathered mass fluxes returned here (colphys_convtran_dev; zm_conv_intr.F90:875).
 * ------------------------------------------------------------------------------------------------- */
typedef struct colphys_zm_conv_tend_io {
    /* physics_state */
    const double *t, *q, *u, *v, *pmid, *pint, *pdel, *rpdel, *lnpint, *lnpmid, *zm, *zi;
    const double *phis, *ps;                 /* [ncol] */
    const double *rair, *zvir;               /* [ncol][pver], as geopotential_t takes them; or both NULL: the engine's constants */
    const double *pblh, *tpert, *landfrac;   /* [ncol] */
    const double *cld;                       /* [ncol][pver], pbuf CLD at the old time level (:585) */
    /* ptend_all */
    double *ptend_s, *ptend_q, *ptend_u, *ptend_v;
    /* surface / pbuf fields */
    double *prec, *snow, *rliq, *rice, *jctop, *jcbot, *cape;      /* [ncol] */
    double *ql, *rprd, *evapcdp, *dlf;                             /* ICWMRDP, RPRDDP, NEVAPR_DPCU, DLFZM */
    double *mconzm, *flxprec, *flxsnow;                            /* [ncol][pver+1]: CMFMC_DP (kg/m2/s), DP_FLXPRC, DP_FLXSNW */
    double *mu, *md, *du, *eu, *ed, *dp, *dsubcld;                 /* ZM_* gathered layout */
    int32_t *jt, *maxg, *ideep, *lengath;
    /* diagnostics (any may be NULL) */
    double *freqzm, *pcont, *pconb;                                /* [ncol] */
    double *mu_out, *md_out;                                       /* [ncol][pver], kg/m2/s */
    /* state1 after the three updates (any may be NULL) */
    double *t1, *q1, *u1, *v1, *zi1, *zm1;
    int32_t *status;                                               /* [ncol] */
    /* ABI 4 -- zm_conv_tend's own intent(out) arguments besides mcon (:420-427) and everything else it hands to outfld
     * (:677-731, :781-792, :845-857).  Any may be NULL: the quantity then stays in the engine's work arrays.  The heating
     * rates are J/kg/s as the kernels produce them; the history fields of the same name divided by cpair are the caller's
     * (ZMDT = zmdt_s/cpair, EVAPTZM = evap_s/cpair, FZSNTZM, EVSNTZM, ZMMTT = seten/cpair), exactly as in the reference. */
    double *cme, *zdu, *dif, *eurt;                                /* [ncol][pver]: cme, zdu; DIFZM; eurt (EURT = eurt(:,pver)) */
    double *pflx;                                                  /* [ncol][pver+1] */
    double *zmdt_s, *zmdq;                                         /* zm_convr's ptend_loc%s, %q(:,:,1): ZMDT x cpair, ZMDQ */
    double *evap_s, *evap_s_snwprd, *evap_s_snwevmlt;              /* zm_conv_evap: ZMEIHEAT, FZSNTZM x cpair, EVSNTZM x cpair */
    double *ntprprd, *ntsnprd;                                     /* ZMNTPRPD, ZMNTSNPD */
    double *seten;                                                 /* momtran's dissipative heating: ZMMTT x cpair */
    double *pguall, *pgdall, *icwu, *icwd;                         /* [2][ncol][pver] (u, v): ZM{U,V}PGU, ZM{U,V}PGD, ZMIC{U,V}U, ZMIC{U,V}D */
} colphys_zm_conv_tend_io;

/* qmin: lower bound physics_update's qneg3 applies to water vapour (constituents' qmin(1), 1e-12 in CAM) */
int colphys_zm_conv_tend_dev(int ncol, int pver, int pcols, double ztodt, int dycore_kind, double qmin,
                             const colphys_zm_conv_tend_io *io, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole-rank HOST FACE: physics_state / pbuf / ptend arrays of every chunk of a rank, in the reference's own
 * per-chunk layout -- Fortran (pcols, nlev), column index fastest, exactly what state(c)%t, ptend(c)%s or a pbuf field
 * is (physics_types.F90:62-121, 137-173) -- in PINNED host memory that the caller fills and reads in place.
 *
 * This is what replaces the chunk loop of phys_run1 (physpkg.F90:1147-1161) for the path: the loop body shrinks to
 * copying state(c) into the chunk's record (thread-safe for distinct chunks: colphys_hostface_base + chunk * record
 * size + field offset), ONE colphys_hostface_run for the rank, and a second loop copying the chunk's results out.
 * Nothing is transposed on the host: a record goes to the device as it is, a HIP kernel re-lays it out to the engine's
 * column-contiguous buffers at HBM speed, the kernels run, a second kernel re-lays the results out into output
 * records, which come back as they are.  The rank is cut into `nslices` chunk ranges that are pipelined over four
 * streams by role (upload, two for kernels, download), so that the upload of one slice, the kernels of another and the
 * download of a third overlap (PCIe is full duplex): the wall time of a run tends to the larger of the two transfer times.
 *
 * Padding columns (i > ncol(chunk)) hold +inf in CAM: the device shadows them with the chunk's first column and
 * drops them from the gathered sets again; their output slots are written but meaningless.
 *
 * Field names: inputs  t q u v pmid pdel rpdel lnpmid zm cld (pcols,pver); pint lnpint zi (pcols,pver+1); phis ps pblh tpert
 *                      landfrac (pcols); with COLPHYS_HF_AERO: pdeldry, dgnumwet_<m>, qaerwat_<m>, specmmr_<m>_<l> (pcols,pver),
 *                      m = 1..nmodes, l = 1..nspec(m); with COLPHYS_HF_SWDIAG: troplevchem (pcols) int32
 *              outputs ptend_s ptend_q ptend_u ptend_v ql rprd evapcdp dlf mu md du eu ed dp (pcols,pver); mconzm flxprec flxsnow
 *                      (pcols,pver+1); prec snow rliq rice jctop jcbot cape dsubcld freqzm pcont pconb (pcols); jt maxg ideep
 *                      status (pcols) int32; lengath (1) int32; with COLPHYS_HF_HIST the ABI-4 history outputs of
 *                      colphys_zm_conv_tend_io under their names there (pguall ... icwd as <name>_u, <name>_v); with
 *                      COLPHYS_HF_AERO: sw_tauxar sw_wa sw_ga sw_fa (pcols, 0:pver, nswbands), lw_tauxar (pcols,pver,nlwbands),
 *                      sw_status lw_status (pcols) int32; with COLPHYS_HF_SWDIAG: swd_lev (pcols,pver,COLPHYS_SWD_NLEV),
 *                      swd_col (pcols, COLPHYS_SWD_NCOL + 3 nmodes)
 * mu ... ideep keep the reference's per-chunk gathered layout (slot j of the chunk in row j).
 * ------------------------------------------------------------------------------------------------- */
#define COLPHYS_HF_ZM     1   /* zm_conv_tend */
#define COLPHYS_HF_AERO   2   /* modal_aero_sw, modal_aero_lw (colphys_aero_tables_set first).  Without COLPHYS_HF_ZM the face moves the
                               optics' fields only: inputs t pmid (read by the SW diagnostics) + the COLPHYS_HF_AERO inputs, no ZM outputs,
                               no lengath -- the radiation driver's call, away from the convection step */
#define COLPHYS_HF_SWDIAG 4   /* + modal_aero_sw's history diagnostics (needs COLPHYS_HF_AERO) */
#define COLPHYS_HF_HIST   8   /* + zm_conv_tend's history / argument outputs (ABI 4; needs COLPHYS_HF_ZM) */

typedef struct colphys_hostface colphys_hostface;

/* ncol[nchunks]: active columns per chunk (1..pcols).  nslices >= 1 (0: chosen by the library). */
int colphys_hostface_create(int nchunks, int pcols, int pver, const int32_t *ncol, int what, int nslices, colphys_hostface **hf);
int colphys_hostface_destroy(colphys_hostface *hf);
/* record geometry: bytes per chunk of the input / output records, number of fields */
int colphys_hostface_geometry(const colphys_hostface *hf, int64_t *in_record_bytes, int64_t *out_record_bytes, int32_t *nfields);
/* field i (0-based): name (<= 31 chars + NUL into name[32]), nlev (second dimension of the (pcols, nlev) array), is_int
 * (int32 instead of fp64), is_out, byte offset inside its record */
int colphys_hostface_field(const colphys_hostface *hf, int i, char *name, int32_t *nlev, int32_t *is_int, int32_t *is_out, int64_t *offset);
/* pinned base addresses: chunk c's input record starts at in_base + c * in_record_bytes */
void *colphys_hostface_in_base(colphys_hostface *hf);
void *colphys_hostface_out_base(colphys_hostface *hf);
/* H2D + device re-layout + zm_conv_tend (+ optics) + device re-layout + D2H for the whole rank; returns when the output
 * records are complete.  dycore_kind / qmin as colphys_zm_conv_tend_dev. */
int colphys_hostface_run(colphys_hostface *hf, double ztodt, int dycore_kind, double qmin);
/* of the last run: wall time of the call; CPU time the calling thread spent in it; summed device time of the uploads,
 * the two re-layout kernels, the physics kernels and the downloads (they overlap across slices); bytes moved each way */
typedef struct colphys_hostface_times {
    double wall_ms, host_cpu_ms, h2d_ms, relayout_ms, physics_ms, d2h_ms;
    int64_t h2d_bytes, d2h_bytes;
    int32_t nslices, nstreams;
} colphys_hostface_times;
int colphys_hostface_last_times(const colphys_hostface *hf, colphys_hostface_times *t);

/* ---------------------------------------------------------------------------------------------------
 * Measurement helper: average device time (ms) of each kernel family since the last reset, measured
 * with HIP events on the launch stream when profiling is switched on (bench.py's roofline leg).
 * ------------------------------------------------------------------------------------------------- */
#define COLPHYS_K_GEOPOTENTIAL 0
#define COLPHYS_K_ZM_TRIGGER   1
#define COLPHYS_K_ZM_PLUME1    2   /* first cloud-ensemble pass of the triggered columns */
#define COLPHYS_K_ZM_EVAP      3
#define COLPHYS_K_MOMTRAN      4
#define COLPHYS_K_AERO_SW      5
#define COLPHYS_K_AERO_LW      6
#define COLPHYS_K_UPDATE       7
#define COLPHYS_K_CONVTRAN     8
#define COLPHYS_K_ZM_ASCENT2   9   /* second dilute ascent of the triggered columns */
#define COLPHYS_K_ZM_PLUME2    10  /* second ensemble pass, closure, tendencies, gather */
#define COLPHYS_K_AERO_SW_DIAG 11  /* modal_aero_sw history diagnostics */
#define COLPHYS_K_COUNT        12
int colphys_profile_enable(int on);
int colphys_profile_reset(void);
int colphys_profile_get(int kernel, double *avg_ms, int64_t *launches);

/* ---------------------------------------------------------------------------------------------------
 * Diagnostic: evaluates, element-wise on device arrays, the scalar device functions the kernels use where the
 * reference calls its libm / its L0 saturation seam, so that their accuracy can be tested on its own:
 *   COLPHYS_PROBE_LOG    y = log(x)      (entropy, zm_conv.F90:5297; CAPE integral :4790)
 *   COLPHYS_PROBE_LOG10  y = log10(x)    (Goff-Gratch svp_water, oracle/stubs/l0_arithmetic.F90)
 *   COLPHYS_PROBE_SVP    y = svp_water(x) [Pa], x = temperature [K]
 *   COLPHYS_PROBE_EXP    y = exp(x)
 * ------------------------------------------------------------------------------------------------- */
#define COLPHYS_PROBE_LOG   0
#define COLPHYS_PROBE_LOG10 1
#define COLPHYS_PROBE_SVP   2
#define COLPHYS_PROBE_EXP   3   /* y = exp(x), |x| < 700  (specific extinction of the SW optics, modal_aer_opt.F90:874) */
int colphys_math_probe_dev(int kind, int64_t n, const double *x, double *y, void *stream);

/* Diagnostic: table_interp's interval search (table_interp_calcwghts; modal_aer_opt.F90:857-858, 1383-1384) of one grid of
 * one (mode, band) table set -- which = 0 the real-index grid, 1 the imaginary-index grid -- evaluated on device values x:
 * 1-based ix1, ix2 and the weight t of ix2.  use_lut = 0: the scan over the grid points (the reference's loop);
 * 1: the bucket lookup the optics kernels use when *lut_available is 1 (only then meaningful).  Both must agree. */
int colphys_aero_grid_probe_dev(int sw, int mode, int band, int which, int use_lut, int64_t n, const double *x, int32_t *ix1,
                                int32_t *ix2, double *t, int32_t *lut_available, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COLPHYS_H */
