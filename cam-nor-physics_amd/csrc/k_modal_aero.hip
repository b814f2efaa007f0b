// k_modal_aero.hip -- modal aerosol optics (Ghan & Zaveri 2007 Chebyshev parameterisation) for gfx950.
//
// Replaces the compute part of /root/reference/physics/modal_aer_opt.F90:
//   modal_aero_sw :485-1212, modal_aero_lw :1216-1469, modal_size_parameters :1538-1574.
// (addfld/outfld history diagnostics stay on the host side; DESIGN.md "out of scope".)
//
// Every (column, level) cell is independent and modes only accumulate, so:
//   work item  = one cell; a block is 4 waves = 256 consecutive cells of the [ncol][pver] layout
//                (coalesced reads of pdeldry / dgnumwet / qaerwat / specmmr);
//   grid.y     = band group: the SW bands are split into groups so that the group's running sums
//                tau, tau*w, tau*w*g, tau*w*g^2 for 256 cells fit in LDS next to the tables and two blocks
//                share a CU (the per-(cell, mode) size/volume prologue is recomputed per group: ~10 % work);
//   LDS        = (a) the (mode, band) slice of the coefficient tables -- ext/abs/asm(5,7,10) + the two
//                refractive-index grids -- because each cell gathers 4 corners x 5 coefficients x 3 tables
//                per (mode, band) at data-dependent indices; coefficients padded 5 -> 6 so a corner is
//                three 16-byte LDS reads; DOUBLE-BUFFERED: the next slice is fetched into registers
//                before the current one is used and stored after it, one barrier per (mode, band);
//                (b) the block's band accumulators [band][4][256 cells] (lane-contiguous, conflict-free);
//   order      = modes outer, bands inner: exactly the reference's accumulation order per output element.
// table_interp semantics: oracle/stubs/l0_arithmetic.F90 (parity-unpinned seam).
#include "colphys_common.h"

#ifndef AERO_MIN_WAVES
#define AERO_MIN_WAVES 2
#endif
#ifndef AERO_LDS_CAP_KB
#define AERO_LDS_CAP_KB 78
#endif
#ifdef AERO_CONTRACT
#pragma clang fp contract(fast)
#endif

namespace {

constexpr int NT = 256;
constexpr int NSMAX = 8;
constexpr int NC = COLPHYS_NCOEF, NR = COLPHYS_PREFR, NI = COLPHYS_PREFI;
constexpr int NCP = 6;                      // coefficients padded to 6 doubles (48 B) per table corner
constexpr int TBL = NI * NR * NC;           // 350 doubles per table in HBM
constexpr int TBLP = NI * NR * NCP;         // 420 doubles per table in LDS
constexpr int GRIDS = 24;                   // refr[7] + refi[10], padded

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

// Bilinear table interpolation (table_interp) fused with the Chebyshev sum that consumes it:
//     p = 0.5*c(1) + sum_{nc=2..5} cheb(nc)*c(nc)        (modal_aer_opt.F90:870-873, 882-887, 1390-1393)
// with c(nc) the interpolated coefficient, evaluated in the reference's order.  tbl: LDS [NI][NR][NCP]; the
// four corners are read as 16-byte pairs, one coefficient pair at a time, to keep the live registers low.
__device__ __forceinline__ double interp_cheb(const double *tbl, const Wght &wr, const Wght &wi, const double *cheb)
{
    const double2 *p11 = reinterpret_cast<const double2 *>(tbl + (wi.ix1 * NR + wr.ix1) * NCP);
    const double2 *p21 = reinterpret_cast<const double2 *>(tbl + (wi.ix1 * NR + wr.ix2) * NCP);
    const double2 *p12 = reinterpret_cast<const double2 *>(tbl + (wi.ix2 * NR + wr.ix1) * NCP);
    const double2 *p22 = reinterpret_cast<const double2 *>(tbl + (wi.ix2 * NR + wr.ix2) * NCP);
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < NCP / 2; ++j) {
        const double2 a = p11[j], b = p21[j], c = p12[j], d = p22[j];
        {
            const double fx1 = wr.wt1 * a.x + wr.wt2 * b.x;
            const double fx2 = wr.wt1 * c.x + wr.wt2 * d.x;
            const double r = wi.wt1 * fx1 + wi.wt2 * fx2;
            if (j == 0) p = 0.5 * r;
            else p = p + cheb[2 * j] * r;
        }
        if (2 * j + 1 < NC) {
            const double fx1 = wr.wt1 * a.y + wr.wt2 * b.y;
            const double fx2 = wr.wt1 * c.y + wr.wt2 * d.y;
            const double r = wi.wt1 * fx1 + wi.wt2 * fx2;
            p = p + cheb[2 * j + 1] * r;
        }
    }
    return p;
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

// One table slice = NTAB tables of TBL doubles + refr[NR] + refi[NI] in HBM; NTAB*TBLP + GRIDS doubles in LDS.
// Fetch: each thread pulls its share into registers (issued early); store: writes them padded (issued late).
template <int NTAB>
struct SliceRegs {
    static constexpr int NELEM = NTAB * TBL + NR + NI;
    static constexpr int PER = (NELEM + NT - 1) / NT;
    double v[PER];
};

template <bool SW>
__device__ __forceinline__ void slice_fetch(const AeroDev &A, int m, int ib, int nb, SliceRegs<SW ? 3 : 1> &r)
{
    constexpr int NTAB = SW ? 3 : 1;
    const size_t tb = ((size_t)m * nb + ib) * TBL;
    const size_t gr = ((size_t)m * nb + ib);
#pragma unroll
    for (int j = 0; j < SliceRegs<NTAB>::PER; ++j) {
        const int e = threadIdx.x + j * NT;
        double x = 0.0;
        if (e < NTAB * TBL) {
            const int tsel = e / TBL, off = e - tsel * TBL;
            const double *src = SW ? (tsel == 0 ? A.extpsw : (tsel == 1 ? A.abspsw : A.asmpsw)) : A.absplw;
            x = src[tb + off];
        } else if (e < NTAB * TBL + NR) {
            x = (SW ? A.refrtabsw : A.refrtablw)[gr * NR + (e - NTAB * TBL)];
        } else if (e < NTAB * TBL + NR + NI) {
            x = (SW ? A.refitabsw : A.refitablw)[gr * NI + (e - NTAB * TBL - NR)];
        }
        r.v[j] = x;
    }
}

template <bool SW>
__device__ __forceinline__ void slice_store(double *s_slice, const SliceRegs<SW ? 3 : 1> &r)
{
    constexpr int NTAB = SW ? 3 : 1;
#pragma unroll
    for (int j = 0; j < SliceRegs<NTAB>::PER; ++j) {
        const int e = threadIdx.x + j * NT;
        if (e < NTAB * TBL) {
            const int tsel = e / TBL, off = e - tsel * TBL;
            const int corner = off / NC, c = off - corner * NC;
            s_slice[tsel * TBLP + corner * NCP + c] = r.v[j];
        } else if (e < NTAB * TBL + NR + NI) {
            s_slice[NTAB * TBLP + (e - NTAB * TBL)] = r.v[j];
        }
    }
}

template <bool SW>
__global__ __launch_bounds__(NT, AERO_MIN_WAVES) void modal_aero_kernel(AeroDev A, int ncol, int pver, int bands_per_group,
                                                       colphys_modal_aero_io io)
{
    constexpr int NTAB = SW ? 3 : 1;
    constexpr int NOUT = SW ? 4 : 1;
    constexpr int SLICE = NTAB * TBLP + GRIDS;
    extern __shared__ double smem[];
    double *s_slice0 = smem;                           // two slices (double buffer)
    double *s_acc = smem + 2 * SLICE;                  // [bands_per_group][NOUT][NT]

    const int nb = SW ? A.nswbands : A.nlwbands;
    const int b0 = blockIdx.y * bands_per_group;
    const int nbg = min(bands_per_group, nb - b0);     // bands of this group
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const size_t cell = (size_t)blockIdx.x * NT + threadIdx.x;
    const bool live = cell < F;
    const int col = live ? (int)(cell / L) : 0;
    const int k = live ? (int)(cell - (size_t)col * L) + 1 : 0;         // 1-based level
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;

    for (int i = threadIdx.x; i < nbg * NOUT * NT; i += NT) s_acc[i] = 0.0;

    double mass = 0.0;
    if (active) mass = io.pdeldry[cell] * A.rga;                        // :644 / :1300
    int bad = 0;

    SliceRegs<NTAB> nxt;
    slice_fetch<SW>(A, 0, b0, nb, nxt);
    slice_store<SW>(s_slice0, nxt);
    __syncthreads();
    int cur = 0;

    const int nunits = A.nmodes * nbg;
    int unit = 0;
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
        for (int jb = 0; jb < nbg; ++jb, ++unit) {                      // :726 / :1343
            const int ib = b0 + jb;
            const double *s_tbl = s_slice0 + cur * SLICE;
            const double *s_refr = s_tbl + NTAB * TBLP;
            const double *s_refi = s_refr + NR;
            // prefetch the next (mode, band) slice into registers; it lands in the other buffer after the compute
            const bool more = unit + 1 < nunits;
            if (more) {
                const int jn = (jb + 1 < nbg) ? jb + 1 : 0;
                const int mn = (jb + 1 < nbg) ? m : m + 1;
                slice_fetch<SW>(A, mn, b0 + jn, nb, nxt);
            }
            if (active) {
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
                const Wght wr = calcwghts(NR, s_refr, refr);            // :857-858 / :1383-1384
                const Wght wi = calcwghts(NI, s_refi, refi);

                if (SW) {
                    double pext = interp_cheb(s_tbl, wr, wi, cheb);         // :860, 870-873
                    double pabs = interp_cheb(s_tbl + TBLP, wr, wi, cheb);  // :861, 882-887
                    const double pasm = interp_cheb(s_tbl + 2 * TBLP, wr, wi, cheb);   // :862
                    if (logradsurf <= A.xrmax) pext = exp(pext);        // :869-877
                    else pext = 1.5 / (radsurf * A.rhoh2o);
                    pext = pext * wetvol * A.rhoh2o;                    // :881-894
                    pabs = pabs * wetvol * A.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    pabs = fmin(pext, pabs);
                    const double palb = 1.0 - pabs / fmax(pext, 1.e-40);
                    const double dopaer = pext * mass;
                    if (dopaer <= -1.e-10 || dopaer >= 30.0) bad = 1;   // :997
                    double *acc = s_acc + (size_t)jb * 4 * NT + threadIdx.x;
                    acc[0] = acc[0] + dopaer;                           // :1032-1037
                    acc[NT] = acc[NT] + dopaer * palb;
                    acc[2 * NT] = acc[2 * NT] + dopaer * palb * pasm;
                    acc[3 * NT] = acc[3 * NT] + dopaer * palb * pasm * pasm;
                } else {
                    double pabs = interp_cheb(s_tbl, wr, wi, cheb);         // :1386, 1390-1393
                    pabs = pabs * wetvol * A.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    const double dopaer = pabs * mass;
                    if (dopaer <= -1.e-10 || dopaer >= 20.0) bad = 1;   // :1402
                    s_acc[(size_t)jb * NT + threadIdx.x] = s_acc[(size_t)jb * NT + threadIdx.x] + dopaer;   // :1441
                }
            }
            if (more) slice_store<SW>(s_slice0 + (cur ^ 1) * SLICE, nxt);
            __syncthreads();
            cur ^= 1;
        }
    }

    // write-out: lanes are consecutive levels of a column => contiguous runs per (column, band)
    if (live) {
        if (SW) {
            const size_t LP = L + 1;
            for (int jb = 0; jb < nbg; ++jb) {
                const size_t o = ((size_t)col * nb + (b0 + jb)) * LP + k;
                io.tauxar[o] = s_acc[((size_t)jb * 4 + 0) * NT + threadIdx.x];
                io.wa[o] = s_acc[((size_t)jb * 4 + 1) * NT + threadIdx.x];
                io.ga[o] = s_acc[((size_t)jb * 4 + 2) * NT + threadIdx.x];
                io.fa[o] = s_acc[((size_t)jb * 4 + 3) * NT + threadIdx.x];
                if (k == 1) {                                           // zero'th layer :639-642
                    io.tauxar[o - 1] = 0.0; io.wa[o - 1] = 0.925; io.ga[o - 1] = 0.850; io.fa[o - 1] = 0.7225;
                }
            }
        } else {
            for (int jb = 0; jb < nbg; ++jb)
                io.tauxar[((size_t)col * nb + (b0 + jb)) * L + (k - 1)] = s_acc[(size_t)jb * NT + threadIdx.x];
        }
        if (bad) atomicOr(io.status + col, COLPHYS_STATUS_DOPAER_RANGE);
    }
}

}  // namespace

static int launch_aero(bool sw, const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st)
{
    if (A.nspec_max > NSMAX) return COLPHYS_ERR_UNSUPPORTED;
    const int nb = sw ? A.nswbands : A.nlwbands;
    const int ntab = sw ? 3 : 1, nout = sw ? 4 : 1;
    const size_t slice = (size_t)ntab * TBLP + GRIDS;
    // band groups: largest group whose accumulators + 2 slices stay within ~76 KB (two blocks per CU)
    int ngroups = 1;
    for (;; ++ngroups) {
        const int bpg = (nb + ngroups - 1) / ngroups;
        if ((2 * slice + (size_t)bpg * nout * NT) * sizeof(double) <= AERO_LDS_CAP_KB * 1024 || bpg == 1) break;
    }
    const int bpg = (nb + ngroups - 1) / ngroups;
    ngroups = (nb + bpg - 1) / bpg;
    const size_t lds = (2 * slice + (size_t)bpg * nout * NT) * sizeof(double);
    const size_t cells = (size_t)ncol * pver;
    const dim3 grid((unsigned)((cells + NT - 1) / NT), (unsigned)ngroups);
    if (hipMemsetAsync(io.status, 0, sizeof(int32_t) * ncol, st) != hipSuccess) return COLPHYS_ERR_HIP;
    if (sw) {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void *)modal_aero_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(modal_aero_kernel<true>, grid, dim3(NT), lds, st, A, ncol, pver, bpg, io);
    } else {
        if (lds > 64 * 1024)
            hipFuncSetAttribute((const void *)modal_aero_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(modal_aero_kernel<false>, grid, dim3(NT), lds, st, A, ncol, pver, bpg, io);
    }
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
