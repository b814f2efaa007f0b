// k_modal_aero.hip -- modal aerosol optics (Ghan & Zaveri 2007 Chebyshev parameterisation) for gfx950.
//
// Replaces the compute part of /root/reference/physics/modal_aer_opt.F90:
//   modal_aero_sw :485-1212, modal_aero_lw :1216-1469, modal_size_parameters :1538-1574.
// (addfld/outfld history diagnostics stay on the host side; DESIGN.md "out of scope".)
//
// Every (column, level) cell is independent and modes only accumulate, so:
//   work item  = one cell; a block is one wave of 64 consecutive cells of the [ncol][pver] layout
//                (coalesced reads of pdeldry / dgnumwet / qaerwat / specmmr);
//   LDS        = (a) the (mode, band) slice of the coefficient tables -- ext/abs/asm(5,7,10) + the two
//                refractive-index grids, 8.5 KB -- staged once per (mode, band) for the whole block, because
//                each cell gathers 60 table entries per (mode, band) at data-dependent indices;
//                (b) the block's band accumulators tau, tau*w, tau*w*g, tau*w*g^2 [band][4][64 cells],
//                so the 4 x 14 running sums stay on chip across the mode loop and leave with one
//                coalesced store per (column, band) row;
//   order      = modes outer, bands inner, exactly the reference's accumulation order per output element.
// table_interp semantics: oracle/stubs/l0_arithmetic.F90 (parity-unpinned seam).
#include "colphys_common.h"

namespace {

constexpr int NT = 64;
constexpr int NSMAX = 8;
constexpr int NC = COLPHYS_NCOEF, NR = COLPHYS_PREFR, NI = COLPHYS_PREFI;
constexpr int TBL = NI * NR * NC;   // 350

struct Wght { double wt1, wt2; int ix1, ix2; };

__device__ __forceinline__ Wght calcwghts(int ngrid, const double *xgrid, double x)
{
    int ii;
    for (ii = 1; ii <= ngrid; ++ii)
        if (x < xgrid[ii - 1]) break;
    const int ix1 = max(ii - 1, 1), ix2 = min(ix1 + 1, ngrid);
    const double dx = xgrid[ix2 - 1] - xgrid[ix1 - 1];
    double t;
    if (fabs(dx) > 1.e-20) t = (x - xgrid[ix1 - 1]) / dx;
    else t = 0.0;
    Wght w;
    w.wt1 = 1.0 - t; w.wt2 = t; w.ix1 = ix1 - 1; w.ix2 = ix2 - 1;
    return w;
}

__device__ __forceinline__ void interp(const double *tbl, const Wght &wr, const Wght &wi, double *res)
{
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double fx1 = wr.wt1 * tbl[(wi.ix1 * NR + wr.ix1) * NC + c] + wr.wt2 * tbl[(wi.ix1 * NR + wr.ix2) * NC + c];
        const double fx2 = wr.wt1 * tbl[(wi.ix2 * NR + wr.ix1) * NC + c] + wr.wt2 * tbl[(wi.ix2 * NR + wr.ix2) * NC + c];
        res[c] = wi.wt1 * fx1 + wi.wt2 * fx2;
    }
}

__device__ __forceinline__ void size_parameters(const AeroDev &A, double sigma_logr_aer, double dgnumwet, double &radsurf,
                                                double &logradsurf, double *cheb)
{   // :1553-1570
    const double alnsg_amode = log(sigma_logr_aer);
    const double explnsigma = exp(2.0 * alnsg_amode * alnsg_amode);
    radsurf = 0.5 * dgnumwet * explnsigma;
    logradsurf = log(radsurf);
    double xrad = fmax(logradsurf, A.xrmin);
    xrad = fmin(xrad, A.xrmax);
    xrad = (2.0 * xrad - A.xrmax - A.xrmin) / (A.xrmax - A.xrmin);
    cheb[0] = 1.0;
    cheb[1] = xrad;
#pragma unroll
    for (int nc = 2; nc < NC; ++nc) cheb[nc] = 2.0 * xrad * cheb[nc - 1] - cheb[nc - 2];
}

template <bool SW>
__global__ __launch_bounds__(NT) void modal_aero_kernel(AeroDev A, int ncol, int pver, colphys_modal_aero_io io)
{
    extern __shared__ double smem[];
    const int nb = SW ? A.nswbands : A.nlwbands;
    const int nout = SW ? 4 : 1;
    double *s_tbl = smem;                              // SW: ext, abs, asm [TBL] each; LW: abs
    double *s_refr = s_tbl + (SW ? 3 : 1) * TBL;       // [NR]
    double *s_refi = s_refr + NR;                      // [NI]  (+1 pad)
    double *s_acc = s_refi + NI + 1;                   // [nb][nout][NT]

    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const size_t cell0 = (size_t)blockIdx.x * NT;
    const size_t cell = cell0 + threadIdx.x;
    const bool live = cell < F;
    const int col = live ? (int)(cell / L) : 0;
    const int k = live ? (int)(cell - (size_t)col * L) + 1 : 0;         // 1-based level
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;

    for (int i = threadIdx.x; i < nb * nout * NT; i += NT) s_acc[i] = 0.0;

    double mass = 0.0;
    if (active) mass = io.pdeldry[cell] * A.rga;                        // :644 / :1300
    int bad = 0;

    for (int m = 0; m < A.nmodes; ++m) {                                // :706 / :1328
        const int nspec = A.nspec[m];
        double vol[NSMAX], dryvol = 0.0, watervol = 0.0, wetvol = 0.0, radsurf = 1.0, logradsurf = 0.0, cheb[NC];
        if (active) {
            size_parameters(A, A.sigmag[m], io.dgnumwet[m * F + cell], radsurf, logradsurf, cheb);
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {                           // :764-766
                vol[l] = 0.0;
                if (l < nspec) {
                    vol[l] = io.specmmr[((size_t)m * A.nspec_max + l) * F + cell] / A.specdens[m * A.nspec_max + l];
                    dryvol = dryvol + vol[l];
                }
            }
            watervol = io.qaerwat[m * F + cell] * rh2odens;             // :835-844
            wetvol = watervol + dryvol;
            if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
        }
        for (int ib = 0; ib < nb; ++ib) {                               // :726 / :1343
            __syncthreads();
            {   // stage the (mode, band) table slice
                const size_t tb = ((size_t)m * nb + ib) * TBL;
                if (SW) {
                    for (int i = threadIdx.x; i < TBL; i += NT) {
                        s_tbl[i] = A.extpsw[tb + i]; s_tbl[TBL + i] = A.abspsw[tb + i]; s_tbl[2 * TBL + i] = A.asmpsw[tb + i];
                    }
                    if (threadIdx.x < NR) s_refr[threadIdx.x] = A.refrtabsw[((size_t)m * nb + ib) * NR + threadIdx.x];
                    if (threadIdx.x < NI) s_refi[threadIdx.x] = A.refitabsw[((size_t)m * nb + ib) * NI + threadIdx.x];
                } else {
                    for (int i = threadIdx.x; i < TBL; i += NT) s_tbl[i] = A.absplw[tb + i];
                    if (threadIdx.x < NR) s_refr[threadIdx.x] = A.refrtablw[((size_t)m * nb + ib) * NR + threadIdx.x];
                    if (threadIdx.x < NI) s_refi[threadIdx.x] = A.refitablw[((size_t)m * nb + ib) * NI + threadIdx.x];
                }
            }
            __syncthreads();
            if (!active) continue;

            // volume-mixed complex refractive index :767, 846-850 / :1360, 1375-1378
            const double *ri = (SW ? A.specrefindex_sw : A.specrefindex_lw) + ((size_t)m * A.nspec_max * nb + ib) * 2;
            double cre = 0.0, cim = 0.0;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec) {
                    cre = cre + vol[l] * ri[(size_t)l * nb * 2];
                    cim = cim + vol[l] * ri[(size_t)l * nb * 2 + 1];
                }
            }
            const double *crefw = (SW ? A.crefwsw : A.crefwlw) + ib * 2;
            cre = cre + watervol * crefw[0];
            cim = cim + watervol * crefw[1];
            double refr, refi;
            if (SW) {
                cre = cre / fmax(wetvol, 1.e-60);
                cim = cim / fmax(wetvol, 1.e-60);
                refr = cre; refi = fabs(cim);
            } else {
                if (wetvol > 1.e-40) { cre = cre / wetvol; cim = cim / wetvol; }
                refr = cre; refi = cim;
            }
            const Wght wr = calcwghts(NR, s_refr, refr);                // :857-858 / :1383-1384
            const Wght wi = calcwghts(NI, s_refi, refi);

            if (SW) {
                double cext[NC], cabs[NC], casm[NC];
                interp(s_tbl, wr, wi, cext);                            // :860-862
                interp(s_tbl + TBL, wr, wi, cabs);
                interp(s_tbl + 2 * TBL, wr, wi, casm);
                double pext, pabs, pasm;
                if (logradsurf <= A.xrmax) {                            // :869-877
                    pext = 0.5 * cext[0];
#pragma unroll
                    for (int nc = 1; nc < NC; ++nc) pext = pext + cheb[nc] * cext[nc];
                    pext = exp(pext);
                } else {
                    pext = 1.5 / (radsurf * A.rhoh2o);
                }
                pext = pext * wetvol * A.rhoh2o;                        // :881-894
                pabs = 0.5 * cabs[0];
                pasm = 0.5 * casm[0];
#pragma unroll
                for (int nc = 1; nc < NC; ++nc) {
                    pabs = pabs + cheb[nc] * cabs[nc];
                    pasm = pasm + cheb[nc] * casm[nc];
                }
                pabs = pabs * wetvol * A.rhoh2o;
                pabs = fmax(0.0, pabs);
                pabs = fmin(pext, pabs);
                const double palb = 1.0 - pabs / fmax(pext, 1.e-40);
                const double dopaer = pext * mass;
                if (dopaer <= -1.e-10 || dopaer >= 30.0) bad = 1;       // :997
                double *acc = s_acc + (size_t)ib * 4 * NT + threadIdx.x;
                acc[0] = acc[0] + dopaer;                               // :1032-1037
                acc[NT] = acc[NT] + dopaer * palb;
                acc[2 * NT] = acc[2 * NT] + dopaer * palb * pasm;
                acc[3 * NT] = acc[3 * NT] + dopaer * palb * pasm * pasm;
            } else {
                double cabs[NC];
                interp(s_tbl, wr, wi, cabs);                            // :1386
                double pabs = 0.5 * cabs[0];                            // :1390-1396
#pragma unroll
                for (int nc = 1; nc < NC; ++nc) pabs = pabs + cheb[nc] * cabs[nc];
                pabs = pabs * wetvol * A.rhoh2o;
                pabs = fmax(0.0, pabs);
                const double dopaer = pabs * mass;
                if (dopaer <= -1.e-10 || dopaer >= 20.0) bad = 1;       // :1402
                s_acc[(size_t)ib * NT + threadIdx.x] = s_acc[(size_t)ib * NT + threadIdx.x] + dopaer;   // :1441
            }
        }
    }
    __syncthreads();

    // write-out: lanes are consecutive levels of a column => contiguous runs per (column, band)
    if (live) {
        if (SW) {
            const size_t LP = L + 1;
            for (int ib = 0; ib < nb; ++ib) {
                const size_t o = ((size_t)col * nb + ib) * LP + k;
                io.tauxar[o] = s_acc[((size_t)ib * 4 + 0) * NT + threadIdx.x];
                io.wa[o] = s_acc[((size_t)ib * 4 + 1) * NT + threadIdx.x];
                io.ga[o] = s_acc[((size_t)ib * 4 + 2) * NT + threadIdx.x];
                io.fa[o] = s_acc[((size_t)ib * 4 + 3) * NT + threadIdx.x];
                if (k == 1) {                                           // zero'th layer :639-642
                    io.tauxar[o - 1] = 0.0; io.wa[o - 1] = 0.925; io.ga[o - 1] = 0.850; io.fa[o - 1] = 0.7225;
                }
            }
        } else {
            for (int ib = 0; ib < nb; ++ib)
                io.tauxar[((size_t)col * nb + ib) * L + (k - 1)] = s_acc[(size_t)ib * NT + threadIdx.x];
        }
        if (bad) atomicOr(io.status + col, COLPHYS_STATUS_DOPAER_RANGE);
    }
}

}  // namespace

static int launch_aero(bool sw, const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st)
{
    if (A.nspec_max > NSMAX) return COLPHYS_ERR_UNSUPPORTED;
    const int nb = sw ? A.nswbands : A.nlwbands;
    const size_t lds = ((sw ? 3 : 1) * TBL + NR + NI + 1 + (size_t)nb * (sw ? 4 : 1) * NT) * sizeof(double);
    const size_t cells = (size_t)ncol * pver;
    const int nblk = (int)((cells + NT - 1) / NT);
    if (hipMemsetAsync(io.status, 0, sizeof(int32_t) * ncol, st) != hipSuccess) return COLPHYS_ERR_HIP;
    if (sw) hipLaunchKernelGGL(modal_aero_kernel<true>, dim3(nblk), dim3(NT), lds, st, A, ncol, pver, io);
    else hipLaunchKernelGGL(modal_aero_kernel<false>, dim3(nblk), dim3(NT), lds, st, A, ncol, pver, io);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_modal_aero_sw(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st)
{
    return launch_aero(true, A, ncol, pver, io, st);
}
int launch_modal_aero_lw(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st)
{
    return launch_aero(false, A, ncol, pver, io, st);
}
