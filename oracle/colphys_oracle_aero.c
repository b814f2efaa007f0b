/* TEST INFRASTRUCTURE -- CPU restatement (plain C) of the reference's modal aerosol optics,
 * /root/reference/physics/modal_aer_opt.F90 (modal_aero_sw :485-1212, modal_aero_lw :1216-1469,
 * modal_size_parameters :1538-1574).  See colphys_oracle.h for who may use it and how it is pinned.
 *
 * Radiative outputs (tauxar, wa, ga, fa / LW tauxar) and, in oracle_modal_aero_sw_diag, the history diagnostics of
 * modal_aero_sw (:647-678, :770-831, :896-993, :1046-1154) in the reference's own loop order (modes, bands, levels, species).
 * table_interp semantics are those of oracle/stubs/l0_arithmetic.F90 (parity-unpinned seam).
 */
#include "colphys_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static colphys_aero_tables T;

void oracle_aero_tables_set(const colphys_aero_tables *tb) { T = *tb; }

typedef struct { double wt1, wt2; int ix1, ix2; } wght;   /* 0-based indices */

static wght calcwghts(int ngrid, const double *xgrid, double x)
{
    wght w;
    int ii;
    for (ii = 1; ii <= ngrid; ++ii)
        if (x < xgrid[ii - 1]) break;
    int ix1 = (ii - 1 > 1) ? ii - 1 : 1;
    int ix2 = (ix1 + 1 < ngrid) ? ix1 + 1 : ngrid;
    double dx = xgrid[ix2 - 1] - xgrid[ix1 - 1], t;
    if (fabs(dx) > 1.e-20) t = (x - xgrid[ix1 - 1]) / dx;
    else t = 0.0;
    w.wt1 = 1.0 - t; w.wt2 = t; w.ix1 = ix1 - 1; w.ix2 = ix2 - 1;
    return w;
}

/* tbl: [prefi][prefr][ncoef] */
static void interp(const double *tbl, wght wr, wght wi, double *res)
{
    for (int c = 0; c < COLPHYS_NCOEF; ++c) {
#define TB(ir, ii) tbl[((size_t)(ii) * COLPHYS_PREFR + (ir)) * COLPHYS_NCOEF + c]
        double fx1 = wr.wt1 * TB(wr.ix1, wi.ix1) + wr.wt2 * TB(wr.ix2, wi.ix1);
        double fx2 = wr.wt1 * TB(wr.ix1, wi.ix2) + wr.wt2 * TB(wr.ix2, wi.ix2);
        res[c] = wi.wt1 * fx1 + wi.wt2 * fx2;
#undef TB
    }
}

/* modal_size_parameters :1538-1574, one cell */
static void size_parameters(double sigma_logr_aer, double dgnumwet, double *radsurf, double *logradsurf, double *cheb)
{
    double alnsg_amode = log(sigma_logr_aer);
    double explnsigma = exp(2.0 * alnsg_amode * alnsg_amode);
    *radsurf = 0.5 * dgnumwet * explnsigma;
    *logradsurf = log(*radsurf);
    double xrad = fmax(*logradsurf, T.xrmin);
    xrad = fmin(xrad, T.xrmax);
    xrad = (2.0 * xrad - T.xrmax - T.xrmin) / (T.xrmax - T.xrmin);
    cheb[0] = 1.0;
    cheb[1] = xrad;
    for (int nc = 2; nc < COLPHYS_NCOEF; ++nc) cheb[nc] = 2.0 * xrad * cheb[nc - 1] - cheb[nc - 2];
}

void oracle_modal_aero_sw(int ncol, int pver, const colphys_modal_aero_io *io)
{
    const int nm = T.nmodes, ns = T.nspec_max, nsw = T.nswbands;
    const size_t L = (size_t)pver, LP = (size_t)pver + 1, F = (size_t)ncol * L;
    const double rh2odens = 1.0 / T.rhoh2o;
    const size_t tblsz = (size_t)COLPHYS_PREFI * COLPHYS_PREFR * COLPHYS_NCOEF;

    for (int i = 0; i < ncol; ++i) {
        io->status[i] = 0;
        for (int isw = 0; isw < nsw; ++isw) {
            double *tau = io->tauxar + ((size_t)i * nsw + isw) * LP, *wa = io->wa + ((size_t)i * nsw + isw) * LP,
                   *ga = io->ga + ((size_t)i * nsw + isw) * LP, *fa = io->fa + ((size_t)i * nsw + isw) * LP;
            for (int k = 0; k <= pver; ++k) { tau[k] = 0.0; wa[k] = 0.0; ga[k] = 0.0; fa[k] = 0.0; }
            tau[0] = 0.0; wa[0] = 0.925; ga[0] = 0.850; fa[0] = 0.7225;          /* :639-642 */
        }
        for (int m = 0; m < nm; ++m) {                                           /* :706 */
            const int nspec = T.nspec[m];
            for (int isw = 0; isw < nsw; ++isw) {                                /* :726 */
                double *tau = io->tauxar + ((size_t)i * nsw + isw) * LP, *wa = io->wa + ((size_t)i * nsw + isw) * LP,
                       *ga = io->ga + ((size_t)i * nsw + isw) * LP, *fa = io->fa + ((size_t)i * nsw + isw) * LP;
                const double *refrtab = T.refrtabsw + ((size_t)m * nsw + isw) * COLPHYS_PREFR;
                const double *refitab = T.refitabsw + ((size_t)m * nsw + isw) * COLPHYS_PREFI;
                const double *ext = T.extpsw + ((size_t)m * nsw + isw) * tblsz;
                const double *abs_ = T.abspsw + ((size_t)m * nsw + isw) * tblsz;
                const double *asm_ = T.asmpsw + ((size_t)m * nsw + isw) * tblsz;
                for (int k = T.top_lev; k <= pver; ++k) {                        /* :731 */
                    const size_t cell = (size_t)i * L + (k - 1);
                    const double mass = io->pdeldry[cell] * T.rga;               /* :644 */
                    double radsurf, logradsurf, cheb[COLPHYS_NCOEF];
                    size_parameters(T.sigmag[m], io->dgnumwet[m * F + cell], &radsurf, &logradsurf, cheb);
                    double cre = 0.0, cim = 0.0, dryvol = 0.0;
                    for (int l = 0; l < nspec; ++l) {                            /* :758-768 */
                        const double specmmr = io->specmmr[((size_t)m * ns + l) * F + cell];
                        const double *ri = T.specrefindex_sw + (((size_t)m * ns + l) * nsw + isw) * 2;
                        double vol = specmmr / T.specdens[m * ns + l];
                        dryvol = dryvol + vol;
                        cre = cre + vol * ri[0];
                        cim = cim + vol * ri[1];
                    }
                    double watervol = io->qaerwat[m * F + cell] * rh2odens;      /* :835-851 */
                    double wetvol = watervol + dryvol;
                    if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
                    cre = cre + watervol * T.crefwsw[isw * 2];
                    cim = cim + watervol * T.crefwsw[isw * 2 + 1];
                    cre = cre / fmax(wetvol, 1.e-60);
                    cim = cim / fmax(wetvol, 1.e-60);
                    double refr = cre, refi = fabs(cim);
                    wght wr = calcwghts(COLPHYS_PREFR, refrtab, refr);           /* :857-862 */
                    wght wi = calcwghts(COLPHYS_PREFI, refitab, refi);
                    double cext[COLPHYS_NCOEF], cabs[COLPHYS_NCOEF], casm[COLPHYS_NCOEF];
                    interp(ext, wr, wi, cext);
                    interp(abs_, wr, wi, cabs);
                    interp(asm_, wr, wi, casm);
                    double pext, pabs, pasm, palb, dopaer;
                    if (logradsurf <= T.xrmax) {                                 /* :869-877 */
                        pext = 0.5 * cext[0];
                        for (int nc = 1; nc < COLPHYS_NCOEF; ++nc) pext = pext + cheb[nc] * cext[nc];
                        pext = exp(pext);
                    } else {
                        pext = 1.5 / (radsurf * T.rhoh2o);
                    }
                    pext = pext * wetvol * T.rhoh2o;                             /* :881-894 */
                    pabs = 0.5 * cabs[0];
                    pasm = 0.5 * casm[0];
                    for (int nc = 1; nc < COLPHYS_NCOEF; ++nc) {
                        pabs = pabs + cheb[nc] * cabs[nc];
                        pasm = pasm + cheb[nc] * casm[nc];
                    }
                    pabs = pabs * wetvol * T.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    pabs = fmin(pext, pabs);
                    palb = 1.0 - pabs / fmax(pext, 1.e-40);
                    dopaer = pext * mass;
                    if (dopaer <= -1.e-10 || dopaer >= 30.0) io->status[i] = (io->status[i] | COLPHYS_STATUS_DOPAER_RANGE | (dopaer < -1.e-10 ? COLPHYS_STATUS_DOPAER_NEGATIVE : 0)) + (1 << COLPHYS_STATUS_COUNT_SHIFT);   /* :997 */
                    tau[k] = tau[k] + dopaer;                                    /* :1032-1037 */
                    wa[k] = wa[k] + dopaer * palb;
                    ga[k] = ga[k] + dopaer * palb * pasm;
                    fa[k] = fa[k] + dopaer * palb * pasm * pasm;
                }
            }
        }
    }
}

/* History diagnostics of modal_aero_sw, climate list: the values the reference hands to outfld as the "dn" fields
 * (:1050-1057, :1094-1100, :1130-1154).  Loop order and association as in the reference: modes (:706), bands (:726),
 * levels (:731), species (:758).  Layouts: include/colphys.h, colphys_modal_aero_sw_diag_dev. */
void oracle_modal_aero_sw_diag(int ncol, int pver, const colphys_modal_aero_io *io, double *lev, double *col)
{
    const int nm = T.nmodes, ns = T.nspec_max, nsw = T.nswbands;
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const double rh2odens = 1.0 / T.rhoh2o;
    const size_t tblsz = (size_t)COLPHYS_PREFI * COLPHYS_PREFR * COLPHYS_NCOEF;
    const int ncolf = COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * nm;
    for (size_t j = 0; j < (size_t)COLPHYS_SWD_NLEV * F; ++j) lev[j] = 0.0;
    for (size_t j = 0; j < (size_t)ncolf * ncol; ++j) col[j] = 0.0;
#define LEV(f, k) lev[(size_t)(f) * F + (size_t)i * L + ((k) - 1)]
#define COL(f) col[(size_t)(f) * ncol + i]

    for (int i = 0; i < ncol; ++i) {
        const int trop = io->troplevchem[i];
        double ssavis = 0.0;
        for (int m = 0; m < nm; ++m) {                                           /* :706 */
            const int nspec = T.nspec[m];
            double burden = 0.0, aodmode = 0.0, dustaodmode = 0.0;               /* :709-711 */
            for (int isw = 0; isw < nsw; ++isw) {                                /* :726 */
                const int savaervis = (isw + 1 == T.idx_sw_diag), savaeruv = (isw + 1 == T.idx_uv_diag),
                          savaernir = (isw + 1 == T.idx_nir_diag);
                const double *refrtab = T.refrtabsw + ((size_t)m * nsw + isw) * COLPHYS_PREFR;
                const double *refitab = T.refitabsw + ((size_t)m * nsw + isw) * COLPHYS_PREFI;
                const double *ext = T.extpsw + ((size_t)m * nsw + isw) * tblsz;
                const double *abs_ = T.abspsw + ((size_t)m * nsw + isw) * tblsz;
                const double *asm_ = T.asmpsw + ((size_t)m * nsw + isw) * tblsz;
                for (int k = T.top_lev; k <= pver; ++k) {                        /* :731 */
                    const size_t cell = (size_t)i * L + (k - 1);
                    const double mass = io->pdeldry[cell] * T.rga;               /* :644 */
                    const double air_density = io->pmid[cell] / (T.rair * io->t[cell]);   /* :645 */
                    double radsurf, logradsurf, cheb[COLPHYS_NCOEF];
                    size_parameters(T.sigmag[m], io->dgnumwet[m * F + cell], &radsurf, &logradsurf, cheb);
                    double cre = 0.0, cim = 0.0, dryvol = 0.0, dustvol = 0.0;
                    double scat[6] = {0, 0, 0, 0, 0, 0}, absb[6] = {0, 0, 0, 0, 0, 0}, hyg[6] = {0, 0, 0, 0, 0, 0};   /* :738-755 */
                    for (int l = 0; l < nspec; ++l) {                            /* :758-832 */
                        const double specmmr = io->specmmr[((size_t)m * ns + l) * F + cell];
                        const double *ri = T.specrefindex_sw + (((size_t)m * ns + l) * nsw + isw) * 2;
                        double vol = specmmr / T.specdens[m * ns + l];
                        dryvol = dryvol + vol;
                        cre = cre + vol * ri[0];
                        cim = cim + vol * ri[1];
                        if (savaervis) {
                            const int ty = T.spectype[m * ns + l];
                            burden = burden + specmmr * mass;                    /* :777 */
                            if (ty >= 0 && ty < 6) {                             /* :780-829: assignments, the last species of a type stays */
                                COL(COLPHYS_SWD_COL_BURDEN0 + ty) = COL(COLPHYS_SWD_COL_BURDEN0 + ty) + specmmr * mass;
                                if (ty == 4) dustvol = vol;
                                scat[ty] = vol * ri[0];
                                absb[ty] = -vol * ri[1];
                                hyg[ty] = vol * T.hygro[m * ns + l];
                            }
                        }
                    }
                    double watervol = io->qaerwat[m * F + cell] * rh2odens;      /* :835-851 */
                    double wetvol = watervol + dryvol;
                    if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
                    cre = cre + watervol * T.crefwsw[isw * 2];
                    cim = cim + watervol * T.crefwsw[isw * 2 + 1];
                    cre = cre / fmax(wetvol, 1.e-60);
                    cim = cim / fmax(wetvol, 1.e-60);
                    double refr = cre, refi = fabs(cim);
                    wght wr = calcwghts(COLPHYS_PREFR, refrtab, refr);           /* :857-862 */
                    wght wi = calcwghts(COLPHYS_PREFI, refitab, refi);
                    double cext[COLPHYS_NCOEF], cabs[COLPHYS_NCOEF], casm[COLPHYS_NCOEF];
                    interp(ext, wr, wi, cext);
                    interp(abs_, wr, wi, cabs);
                    interp(asm_, wr, wi, casm);
                    double pext, pabs, pasm, palb, dopaer;
                    if (logradsurf <= T.xrmax) {                                 /* :869-877 */
                        pext = 0.5 * cext[0];
                        for (int nc = 1; nc < COLPHYS_NCOEF; ++nc) pext = pext + cheb[nc] * cext[nc];
                        pext = exp(pext);
                    } else {
                        pext = 1.5 / (radsurf * T.rhoh2o);
                    }
                    pext = pext * wetvol * T.rhoh2o;                             /* :881-894 */
                    pabs = 0.5 * cabs[0];
                    pasm = 0.5 * casm[0];
                    for (int nc = 1; nc < COLPHYS_NCOEF; ++nc) {
                        pabs = pabs + cheb[nc] * cabs[nc];
                        pasm = pasm + cheb[nc] * casm[nc];
                    }
                    pabs = pabs * wetvol * T.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    pabs = fmin(pext, pabs);
                    palb = 1.0 - pabs / fmax(pext, 1.e-40);
                    dopaer = pext * mass;
                    COL(COLPHYS_SWD_COL_AODTOT) = COL(COLPHYS_SWD_COL_AODTOT) + dopaer;             /* :896 */
                    if (savaeruv) {                                              /* :899-907 */
                        LEV(COLPHYS_SWD_LEV_EXTINCTUV, k) = LEV(COLPHYS_SWD_LEV_EXTINCTUV, k) + dopaer * air_density / mass;
                        COL(COLPHYS_SWD_COL_AODUV) = COL(COLPHYS_SWD_COL_AODUV) + dopaer;
                        if (k <= trop) COL(COLPHYS_SWD_COL_AODUVST) = COL(COLPHYS_SWD_COL_AODUVST) + dopaer;
                    }
                    if (savaernir) {                                             /* :909-917 */
                        LEV(COLPHYS_SWD_LEV_EXTINCTNIR, k) = LEV(COLPHYS_SWD_LEV_EXTINCTNIR, k) + dopaer * air_density / mass;
                        COL(COLPHYS_SWD_COL_AODNIR) = COL(COLPHYS_SWD_COL_AODNIR) + dopaer;
                        if (k <= trop) COL(COLPHYS_SWD_COL_AODNIRST) = COL(COLPHYS_SWD_COL_AODNIRST) + dopaer;
                    }
                    if (savaervis) {                                             /* :921-993 */
                        LEV(COLPHYS_SWD_LEV_EXTINCT, k) = LEV(COLPHYS_SWD_LEV_EXTINCT, k) + dopaer * air_density / mass;
                        LEV(COLPHYS_SWD_LEV_ABSORB, k) = LEV(COLPHYS_SWD_LEV_ABSORB, k) + pabs * air_density;
                        COL(COLPHYS_SWD_COL_AODVIS) = COL(COLPHYS_SWD_COL_AODVIS) + dopaer;
                        COL(COLPHYS_SWD_COL_AODABS) = COL(COLPHYS_SWD_COL_AODABS) + pabs * mass;
                        aodmode = aodmode + dopaer;
                        ssavis = ssavis + dopaer * palb;
                        COL(COLPHYS_SWD_COL_ASYMVIS) = COL(COLPHYS_SWD_COL_ASYMVIS) + dopaer * pasm;
                        LEV(COLPHYS_SWD_LEV_ASYMEXT, k) = LEV(COLPHYS_SWD_LEV_ASYMEXT, k) + dopaer * pasm * air_density / mass;
                        if (k <= trop) COL(COLPHYS_SWD_COL_AODVISST) = COL(COLPHYS_SWD_COL_AODVISST) + dopaer;
                        if (wetvol > 1.e-40) {
                            dustaodmode = dustaodmode + dopaer * dustvol / wetvol;
                            const double scath2o = watervol * T.crefwsw[isw * 2];            /* :943-950 */
                            const double absh2o = -watervol * T.crefwsw[isw * 2 + 1];
                            /* sums in the reference's order: so4, pom, soa, bc, dust, seasalt (= spectype 0..5) */
                            const double sumscat = scat[0] + scat[1] + scat[2] + scat[3] + scat[4] + scat[5] + scath2o;
                            const double sumabs = absb[0] + absb[1] + absb[2] + absb[3] + absb[4] + absb[5] + absh2o;
                            const double sumhygro = hyg[0] + hyg[1] + hyg[2] + hyg[3] + hyg[4] + hyg[5];
                            for (int ty = 0; ty < 6; ++ty) {                     /* :952-968 */
                                scat[ty] = (scat[ty] + scath2o * hyg[ty] / sumhygro) / sumscat;
                                absb[ty] = (absb[ty] + absh2o * hyg[ty] / sumhygro) / sumabs;
                            }
                            COL(COLPHYS_SWD_COL_AODABSBC) = COL(COLPHYS_SWD_COL_AODABSBC) + absb[3] * dopaer * (1.0 - palb);   /* :970 */
                            for (int ty = 0; ty < 6; ++ty) {                     /* :972-988 */
                                const double aodc = (absb[ty] * (1.0 - palb) + palb * scat[ty]) * dopaer;
                                COL(COLPHYS_SWD_COL_AOD0 + ty) = COL(COLPHYS_SWD_COL_AOD0 + ty) + aodc;
                            }
                        }
                    }
                }
            }
            COL(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 0) = burden;         /* :1050-1057 */
            COL(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 1) = aodmode;
            COL(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 2) = dustaodmode;
        }
        COL(COLPHYS_SWD_COL_SSAVIS) = ssavis;     /* the sum of :929; normalised by the caller (:1122-1128, AFTER the night fill of aodvis) */
    }
#undef LEV
#undef COL
}

void oracle_modal_aero_lw(int ncol, int pver, const colphys_modal_aero_io *io)
{
    const int nm = T.nmodes, ns = T.nspec_max, nlw = T.nlwbands;
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const double rh2odens = 1.0 / T.rhoh2o;
    const size_t tblsz = (size_t)COLPHYS_PREFI * COLPHYS_PREFR * COLPHYS_NCOEF;

    for (int i = 0; i < ncol; ++i) {
        io->status[i] = 0;
        for (size_t j = 0; j < (size_t)nlw * L; ++j) io->tauxar[(size_t)i * nlw * L + j] = 0.0;   /* :1297 */
        for (int m = 0; m < nm; ++m) {                                           /* :1328 */
            const int nspec = T.nspec[m];
            for (int ilw = 0; ilw < nlw; ++ilw) {                                /* :1343 */
                double *tau = io->tauxar + ((size_t)i * nlw + ilw) * L - 1;
                const double *refrtab = T.refrtablw + ((size_t)m * nlw + ilw) * COLPHYS_PREFR;
                const double *refitab = T.refitablw + ((size_t)m * nlw + ilw) * COLPHYS_PREFI;
                const double *abl = T.absplw + ((size_t)m * nlw + ilw) * tblsz;
                for (int k = T.top_lev; k <= pver; ++k) {                        /* :1345 */
                    const size_t cell = (size_t)i * L + (k - 1);
                    const double mass = io->pdeldry[cell] * T.rga;               /* :1300 */
                    double radsurf, logradsurf, cheby[COLPHYS_NCOEF];
                    size_parameters(T.sigmag[m], io->dgnumwet[m * F + cell], &radsurf, &logradsurf, cheby);
                    double cre = 0.0, cim = 0.0, dryvol = 0.0;
                    for (int l = 0; l < nspec; ++l) {                            /* :1352-1362 */
                        const double specmmr = io->specmmr[((size_t)m * ns + l) * F + cell];
                        const double *ri = T.specrefindex_lw + (((size_t)m * ns + l) * nlw + ilw) * 2;
                        double vol = specmmr / T.specdens[m * ns + l];
                        dryvol = dryvol + vol;
                        cre = cre + vol * ri[0];
                        cim = cim + vol * ri[1];
                    }
                    double watervol = io->qaerwat[m * F + cell] * rh2odens;      /* :1365-1379 */
                    double wetvol = watervol + dryvol;
                    if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
                    cre = cre + watervol * T.crefwlw[ilw * 2];
                    cim = cim + watervol * T.crefwlw[ilw * 2 + 1];
                    if (wetvol > 1.e-40) { cre = cre / wetvol; cim = cim / wetvol; }
                    double refr = cre, refi = cim;
                    wght wr = calcwghts(COLPHYS_PREFR, refrtab, refr);           /* :1383-1386 */
                    wght wi = calcwghts(COLPHYS_PREFI, refitab, refi);
                    double cabs[COLPHYS_NCOEF];
                    interp(abl, wr, wi, cabs);
                    double pabs = 0.5 * cabs[0];                                 /* :1390-1396 */
                    for (int nc = 1; nc < COLPHYS_NCOEF; ++nc) pabs = pabs + cheby[nc] * cabs[nc];
                    pabs = pabs * wetvol * T.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    double dopaer = pabs * mass;
                    if (dopaer <= -1.e-10 || dopaer >= 20.0) io->status[i] = (io->status[i] | COLPHYS_STATUS_DOPAER_RANGE | (dopaer < -1.e-10 ? COLPHYS_STATUS_DOPAER_NEGATIVE : 0)) + (1 << COLPHYS_STATUS_COUNT_SHIFT);   /* :1402 */
                    tau[k] = tau[k] + dopaer;                                    /* :1441 */
                }
            }
        }
    }
}
