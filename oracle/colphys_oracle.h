/* TEST INFRASTRUCTURE -- CPU restatement (plain C) of the reference hot path.  NOT product code:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Pinning: every function here is checked against the reference's own Fortran compiled in this
 * container (oracle/_ref, tests/test_oracle_vs_golden.py + tests/golden/).  The reference holds no tests
 * or golden vectors of its own.  The upstream L0 arithmetic the reference calls but does not
 * contain (qsat_water, qsat, cldfrc_fice, table_interp) is restated in oracle/stubs/l0_arithmetic.F90 and
 * here identically: parity at those seams is UNPINNED against real CAM builds.
 *
 * Array layout: column-contiguous [ncol][pver] (interfaces [ncol][pver+1]); see include/colphys.h.
 */
#ifndef COLPHYS_ORACLE_H
#define COLPHYS_ORACLE_H
#include "../include/colphys.h"   /* parameter / io structs only (the ABI definition) */

#ifdef __cplusplus
extern "C" {
#endif

void oracle_init(const colphys_zm_params *p);

void oracle_derived_fields(int ncol, int pver, int ncnst, int dycore_kind, int nspecies, double gravit,
                           const colphys_derived_fields_io *io);
void oracle_convect_diagnostics(int ncol, int pver, int clubb, const colphys_convect_diag_io *io);
void oracle_geopotential_dse(int ncol, int pver, int dycore_kind, int nq, int nspecies, const double *piln,
                             const double *pmln, const double *pint, const double *pmid, const double *pdel,
                             const double *rpdel, const double *dse, const double *q, const double *phis, const double *rair,
                             double gravit, const double *cpair, const double *zvir, double *t, double *zi, double *zm);
void oracle_geopotential_t(int ncol, int pver, int dycore_kind, int nq, int nspecies,
                           const double *piln, const double *pmln, const double *pint, const double *pmid,
                           const double *pdel, const double *rpdel, const double *t, const double *q,
                           const double *rair, double gravit, const double *zvir, double *zi, double *zm);

void oracle_zm_convr(int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io *io);
void oracle_zm_conv_evap(int ncol, int pver, double deltat, const colphys_zm_conv_evap_io *io);
void oracle_momtran(int ncol, int pver, int pcols, double dt, const colphys_momtran_io *io);
void oracle_convtran(int ncol, int pver, int pcols, int ncnst, double dt, const colphys_convtran_io *io);

void oracle_aero_tables_set(const colphys_aero_tables *tb);
void oracle_modal_aero_sw(int ncol, int pver, const colphys_modal_aero_io *io);
void oracle_modal_aero_lw(int ncol, int pver, const colphys_modal_aero_io *io);
void oracle_modal_aero_sw_diag(int ncol, int pver, const colphys_modal_aero_io *io, double *lev, double *col);

/* L0 seams, exported so tests can pin them against the Fortran stubs */
void   oracle_qsat_water(double t, double p, double *es, double *qs);
void   oracle_qsat_table(double t, double p, double *es, double *qs);
double oracle_entropy(double tk, double p, double qtot);
double oracle_enthalpy(double tk, double p, double qtot, double z);

#ifdef __cplusplus
}
#endif
#endif
