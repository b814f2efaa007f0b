// k_modal_aero.hip -- modal aerosol optics (Ghan & Zaveri 2007 Chebyshev parameterisation) for gfx950.
//
// Replaces the compute part of /root/reference/physics/modal_aer_opt.F90:
//   modal_aero_sw :485-1212, modal_aero_lw :1216-1469, modal_size_parameters :1538-1574.
// The values of modal_aero_sw's history diagnostics are computed by modal_aero_sw_diag_kernel below; addfld / outfld and
// the night-column fill stay in the Fortran shim.
//
// Every (column, level) cell is independent and modes only accumulate, so in both kernels here
//   work item  = one cell; a block is 4 waves = 256 consecutive cells of the [ncol][pver] layout
//                (coalesced reads of pdeldry / dgnumwet / qaerwat / specmmr);
//   LDS        = the (mode, band) slice of the coefficient tables -- ext/abs/asm(5,7,10) + the two
//                refractive-index grids, their reciprocal interval widths and bucket lookups -- because each cell gathers
//                4 corners x 5 coefficients x 3 tables per (mode, band) at data-dependent indices; coefficients
//                padded 5 -> 6 so a corner is three 16-byte LDS reads.  The slices are packed ONCE on the host in
//                exactly this image (colphys_aero_tables_set), so staging is a straight copy done by the LDS-DMA
//                path (global_load_lds_dwordx4: no staging registers), DOUBLE-BUFFERED: the next slice is in
//                flight while the current one is used, one barrier per (mode, band);
//   unit body  = one (mode, band) step: the species mixing dispatched once on the species count to straight-line code,
//                the grid search of table_interp through the slice's bucket lookup (colphys_common.h AERO_LUTB; a
//                branch-free scan against the grid in SGPRs where a table admits no lookup), three corner-first
//                Chebyshev / bilinear interpolations.
// Two kernel families per routine (colphys_aero_kernel_select):
// modal_aero_reg_kernel<SW|LW> (up to four modes): bands outer, modes inner, every mode's state kept; see there.
// modal_aero_kernel<SW|LW> ("grouped"; any mode count): modes outer, bands inner, the block's band
//                accumulators [band][NOUT][256 cells] in LDS (lane-contiguous, conflict-free); grid.y = band group:
//                the bands are split so that accumulators + slices fit AERO_LDS_CAP_KB and three blocks share a CU
//                (the per-(cell, mode) size / volume prologue is recomputed per group).
// Both add a band's modes in the reference's order.
// table_interp semantics: oracle/stubs/l0_arithmetic.F90 (parity-unpinned seam).
#include "colphys_common.h"

#include <algorithm>

#ifndef AERO_MIN_WAVES
#define AERO_MIN_WAVES 3                   // 3 waves / SIMD: the kernel needs ~136 VGPRs
#endif
#ifndef AERO_SW_WAVES
#define AERO_SW_WAVES 3                    // bands-outer SW kernel: waves per SIMD the register budget is set for
#endif
#ifndef AERO_BATCH_SW
#define AERO_BATCH_SW 1
#endif
#ifndef AERO_BATCH_LW
#define AERO_BATCH_LW 0
#endif
#ifndef AERO_LW_WAVES
#define AERO_LW_WAVES 4                    // bands-outer LW kernel
#endif
#ifndef AERO_PRIO
#define AERO_PRIO 0                       // issue priority of the optics waves (the ZM chain runs at COLPHYS_CHAIN_PRIO)
#endif
#ifndef AERO_LW_LDS_CAP_KB
#define AERO_LW_LDS_CAP_KB 30              // LW: two band groups of 24 KB instead of one of 40 KB: 7 % slower alone, but four
#endif                                    // blocks then leave 64 KB of a CU's LDS to the ZM chain running beside it (substep -1 %)
#ifndef AERO_LDS_CAP_KB
#define AERO_LDS_CAP_KB 53                 // three blocks per CU (A/B on MI355X: 53 KB x 3 beats 78 KB x 2)
#endif
// Unlike the ZM kernels (Brent iterations with discrete stopping tests) the optics are straight polynomial /
// bilinear arithmetic, so FMA contraction is allowed here: -8 % time; measured against the reference goldens the
// worst relative error is 4.9e-15 with (and with the corner-first interpolation below) and 3.2e-15 without (scripts/parity_report.py).  -DAERO_NO_CONTRACT undoes it.
#ifndef AERO_NO_CONTRACT
#pragma clang fp contract(fast)
#endif

namespace {

constexpr int NT = 256;
constexpr int NSMAX = 8;
constexpr int NC = COLPHYS_NCOEF, NR = COLPHYS_PREFR, NI = COLPHYS_PREFI;
constexpr int NCP = AERO_NCP;               // coefficients padded to 6 doubles (48 B) per table corner
constexpr int TBLP = AERO_TBLP;             // 420 doubles per table in the packed slice

struct Wght { double wt1, wt2; int ix1, ix2; };

// Wave-uniform table constants are read through the constant address space: such loads stay scalar (SGPR results,
// scalar cache) even though the kernel also issues LDS-DMA copies, which the compiler otherwise treats as stores
// that could clobber any global load and turns every uniform load into a vector load waiting on vmcnt with the DMA.
typedef const __attribute__((address_space(4))) double *cptr_t;
__device__ __forceinline__ cptr_t as_const(const double *p) { return (cptr_t)(const void *)p; }
typedef const __attribute__((address_space(4))) int *ciptr_t;
__device__ __forceinline__ ciptr_t as_const(const int *p) { return (ciptr_t)(const void *)p; }

// Volume mixing of the species' refractive indices (:767 / :1360) for a mode of exactly NS species, straight-line: the
// NS scalar loads of (re, im) go out as one batch and the FMAs follow without branches.  (A loop over l < nspec with a
// uniform branch per species made every load a round trip of its own and left phi copies behind every branch.)
// V(l): the species' volume of this lane.
template <int NS, class V>
__device__ __forceinline__ void mix_ns(V vol, cptr_t ri, size_t stride, double &cre, double &cim)
{
    double re[NS], im[NS];
#pragma unroll
    for (int l = 0; l < NS; ++l) { re[l] = ri[l * stride]; im[l] = ri[l * stride + 1]; }
#pragma unroll
    for (int l = 0; l < NS; ++l) {
        const double vl = vol(l);
        cre = cre + vl * re[l];
        cim = cim + vl * im[l];
    }
}
template <class V>
__device__ __forceinline__ void mix(int nspec, V vol, cptr_t ri, size_t stride, double &cre, double &cim)
{
    switch (nspec) {                                    // wave-uniform
    case 1: mix_ns<1>(vol, ri, stride, cre, cim); break;
    case 2: mix_ns<2>(vol, ri, stride, cre, cim); break;
    case 3: mix_ns<3>(vol, ri, stride, cre, cim); break;
    case 4: mix_ns<4>(vol, ri, stride, cre, cim); break;
    case 5: mix_ns<5>(vol, ri, stride, cre, cim); break;
    case 6: mix_ns<6>(vol, ri, stride, cre, cim); break;
    case 7: mix_ns<7>(vol, ri, stride, cre, cim); break;
    case 8: mix_ns<8>(vol, ri, stride, cre, cim); break;
    default: break;
    }
}

// calcwghts of table_interp: first grid point above x (the reference's loop with exit), extrapolating below the
// grid and clamping above it.  g: the grid through a wave-uniform global pointer (scalar loads -> SGPR operands);
// s_g / s_rdx: the same grid and the reciprocal widths 1/(g[j+1]-g[j]) of its intervals in LDS (packed with the
// slice on the host; 0 for a degenerate interval and past the last point, which reproduces the reference's
// `t = 0 when |dx| <= 1e-20`) for the two data-dependent reads.  t = (x-x1)*rdx differs from the reference's
// (x-x1)/dx in the last bit at most.
// The reference's sanity check of the layer optical depth (:997-1029, :1402-1437) as per-column status: the condition bits and,
// above COLPHYS_STATUS_COUNT_SHIFT, the number of (mode, band, level) units outside the range -- what the reference adds to
// nerr_dopaer and decides its abort on.
__device__ __forceinline__ void status_report(int32_t *st, int bad, unsigned long long negm)
{
    const bool neg = (negm >> (threadIdx.x & 63)) & 1ull;
    atomicOr(st, COLPHYS_STATUS_DOPAER_RANGE | (neg ? COLPHYS_STATUS_DOPAER_NEGATIVE : 0));
    atomicAdd(st, bad << COLPHYS_STATUS_COUNT_SHIFT);
}

template <int N>
__device__ __forceinline__ Wght calcwghts(cptr_t g, const double *s_g, const double *s_rdx, double x)
{
    int ii = N + 1;
#pragma unroll
    for (int j = N; j >= 1; --j) ii = (x < g[j - 1]) ? j : ii;
    const int ix1 = max(ii - 1, 1), ix2 = min(ix1 + 1, N);
    const double x1 = s_g[ix1 - 1], rdx = s_rdx[ix1 - 1];
    const double t = (x - x1) * rdx;
    Wght w;
    w.wt1 = 1.0 - t; w.wt2 = t; w.ix1 = ix1 - 1; w.ix2 = ix2 - 1;
    return w;
}

// The same interval through the slice's bucket lookup (colphys_common.h, AERO_LUTB): the count of grid points <= x is
// cnt[b] + (x >= gin[b]) with b cut from the high word of x.  Identical ix1 / ix2 / t to the scan above for every
// x that is not a NaN.  s_gin / s_cnt: this grid's lookup arrays in LDS; base / shift: wave-uniform.
template <int N>
__device__ __forceinline__ Wght calcwghts_lut(const double *s_gin, const unsigned char *s_cnt, int base, int shift,
                                              const double *s_g, const double *s_rdx, double x)
{
    const int hi = max(__double2hiint(x), 0);
    const int b = min(max((hi - base) >> shift, 0), AERO_LUTB - 1);
    const double gin = s_gin[b];
    int cnt = s_cnt[b];
    cnt += (x >= gin) ? 1 : 0;
    const int ix1 = max(cnt, 1), ix2 = min(ix1 + 1, N);
    const double x1 = s_g[ix1 - 1], rdx = s_rdx[ix1 - 1];
    const double t = (x - x1) * rdx;
    Wght w;
    w.wt1 = 1.0 - t; w.wt2 = t; w.ix1 = ix1 - 1; w.ix2 = ix2 - 1;
    return w;
}

// both weights of a (mode, band) unit.  LUT: through the lookup in the LDS slice; else: the scan against the grid read by
// scalar loads from the slice's global image (gslice)
template <int NTAB, bool LUT, bool PAIR = false>
__device__ __forceinline__ void unit_weights(const double *s_tbl, const double *gslice, double refr, double refi, Wght &wr, Wght &wi)
{
    const double *s_refr = s_tbl + NTAB * TBLP, *s_refi = s_refr + NR;
    const double *s_rdxr = s_refi + NI, *s_rdxi = s_rdxr + NR;
    if (LUT) {
        const double *s_lut = s_tbl + aero_lut_offset(NTAB);
        const unsigned char *s_cnt = reinterpret_cast<const unsigned char *>(s_lut + 2 * AERO_LUTB);
        ciptr_t par = as_const(reinterpret_cast<const int *>(gslice + aero_lut_offset(NTAB) + 2 * AERO_LUTB + 2 * AERO_LUTB / 8));
        if (PAIR) {
        // both grids side by side: two dependent LDS round trips (bucket entries, then the interval's x1 and 1/dx) instead of
        // four.  Same operations per grid as calcwghts_lut.
        const int br = min(max((max(__double2hiint(refr), 0) - par[0]) >> par[1], 0), AERO_LUTB - 1);
        const int bi = min(max((max(__double2hiint(refi), 0) - par[2]) >> par[3], 0), AERO_LUTB - 1);
        const double ginr = s_lut[br], gini = s_lut[AERO_LUTB + bi];
        int cr = s_cnt[br], ci = s_cnt[AERO_LUTB + bi];
        __builtin_amdgcn_sched_barrier(0);
        cr += (refr >= ginr) ? 1 : 0;
        ci += (refi >= gini) ? 1 : 0;
        const int r1 = max(cr, 1), r2 = min(r1 + 1, NR), i1 = max(ci, 1), i2 = min(i1 + 1, NI);
        const double x1r = s_refr[r1 - 1], rdxr = s_rdxr[r1 - 1], x1i = s_refi[i1 - 1], rdxi = s_rdxi[i1 - 1];
        __builtin_amdgcn_sched_barrier(0);
        const double tr = (refr - x1r) * rdxr, ti = (refi - x1i) * rdxi;
        wr.wt1 = 1.0 - tr; wr.wt2 = tr; wr.ix1 = r1 - 1; wr.ix2 = r2 - 1;
        wi.wt1 = 1.0 - ti; wi.wt2 = ti; wi.ix1 = i1 - 1; wi.ix2 = i2 - 1;
        } else {
        wr = calcwghts_lut<NR>(s_lut, s_cnt, par[0], par[1], s_refr, s_rdxr, refr);
        wi = calcwghts_lut<NI>(s_lut + AERO_LUTB, s_cnt + AERO_LUTB, par[2], par[3], s_refi, s_rdxi, refi);
        }
    } else {
        cptr_t g_grid = as_const(gslice + NTAB * TBLP);                  // uniform -> scalar loads
        wr = calcwghts<NR>(g_grid, s_refr, s_rdxr, refr);
        wi = calcwghts<NI>(g_grid + NR, s_refi, s_rdxi, refi);
    }
}

// Bilinear table interpolation (table_interp) fused with the Chebyshev sum that consumes it:
//     p = 0.5*c(1) + sum_{nc=2..5} cheb(nc)*c(nc)        (modal_aer_opt.F90:870-873, 882-887, 1390-1393)
// with c(nc) the bilinearly interpolated coefficient.  Both operations are linear, so the Chebyshev sum is taken at
// each of the four corners first and the four sums are interpolated once (26 FMA-class instructions instead of 35;
// a reassociation of the reference's order, covered by the optics' measured 4.9e-15 -- see the contraction note above).
// tbl: LDS [NI][NR][NCP]; the four corners are read as 16-byte pairs.
__device__ __forceinline__ double interp_cheb(const double *tbl, const Wght &wr, const Wght &wi, const double *cheb)
{
    const double2 *p11 = reinterpret_cast<const double2 *>(tbl + (wi.ix1 * NR + wr.ix1) * NCP);
    const double2 *p21 = reinterpret_cast<const double2 *>(tbl + (wi.ix1 * NR + wr.ix2) * NCP);
    const double2 *p12 = reinterpret_cast<const double2 *>(tbl + (wi.ix2 * NR + wr.ix1) * NCP);
    const double2 *p22 = reinterpret_cast<const double2 *>(tbl + (wi.ix2 * NR + wr.ix2) * NCP);
    auto corner = [&](const double2 *c) -> double {
        const double2 a = c[0], b = c[1];
        const double e = c[2].x;
        double sum = 0.5 * a.x;
        sum = sum + cheb[1] * a.y;
        sum = sum + cheb[2] * b.x;
        sum = sum + cheb[3] * b.y;
        sum = sum + cheb[4] * e;
        return sum;
    };
    const double s11 = corner(p11), s21 = corner(p21), s12 = corner(p12), s22 = corner(p22);
    const double fx1 = wr.wt1 * s11 + wr.wt2 * s21;
    const double fx2 = wr.wt1 * s12 + wr.wt2 * s22;
    return wi.wt1 * fx1 + wi.wt2 * fx2;
}

// The same for all NTAB tables of a slice at once, every corner of every table read BEFORE any arithmetic: the gathers of a
// unit are one LDS round trip instead of 4 x NTAB dependent ones.  (Left to itself the compiler keeps one corner in flight --
// read three words, wait, five multiply-adds, next corner -- to hold the register count; a wave then spends most of a unit
// waiting on LDS latency.)  Costs 10 VGPRs per corner in flight; same operations and order per table as interp_cheb.
template <int NTAB>
__device__ __forceinline__ void interp_cheb_all(const double *tbl, const Wght &wr, const Wght &wi, const double *cheb, double *out)
{
    const int o11 = (wi.ix1 * NR + wr.ix1) * NCP, o21 = (wi.ix1 * NR + wr.ix2) * NCP;
    const int o12 = (wi.ix2 * NR + wr.ix1) * NCP, o22 = (wi.ix2 * NR + wr.ix2) * NCP;
    double2 a[NTAB][4], b[NTAB][4];
    double e[NTAB][4];
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        const int off[4] = {o11, o21, o12, o22};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double2 *p = reinterpret_cast<const double2 *>(tbl + t * TBLP + off[c]);
            a[t][c] = p[0]; b[t][c] = p[1]; e[t][c] = p[2].x;
        }
    }
    __builtin_amdgcn_sched_barrier(0);                  // nothing moves across: the reads stay together, ahead of the sums
#pragma unroll
    for (int t = 0; t < NTAB; ++t) {
        double sc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double sum = 0.5 * a[t][c].x;
            sum = sum + cheb[1] * a[t][c].y;
            sum = sum + cheb[2] * b[t][c].x;
            sum = sum + cheb[3] * b[t][c].y;
            sum = sum + cheb[4] * e[t][c];
            sc[c] = sum;
        }
        const double fx1 = wr.wt1 * sc[0] + wr.wt2 * sc[1];
        const double fx2 = wr.wt1 * sc[2] + wr.wt2 * sc[3];
        out[t] = wi.wt1 * fx1 + wi.wt2 * fx2;
    }
}

__device__ __forceinline__ void size_parameters(const AeroDev &A, double explnsigma, double dgnumwet, double &radsurf,
                                                double &logradsurf, double *cheb)
{   // :1553-1570.  explnsigma = exp(2 ln(sigma_g)^2) is a per-mode constant (:1553-1554): evaluated once on the host
    // when the tables are set, not per cell
    radsurf = 0.5 * dgnumwet * explnsigma;
    logradsurf = cp_log(radsurf);
    double xrad = fmax(logradsurf, A.xrmin);
    xrad = fmin(xrad, A.xrmax);
    xrad = (2.0 * xrad - A.xrmax - A.xrmin) * A.rxrange;           // /(xrmax - xrmin), :1563, as a product
    cheb[0] = 1.0;
    cheb[1] = xrad;
#pragma unroll
    for (int nc = 2; nc < NC; ++nc) cheb[nc] = 2.0 * xrad * cheb[nc - 1] - cheb[nc - 2];
}

// Start the copy of one packed slice (SLICE doubles, a multiple of 1 KiB) into LDS: each wave-instruction moves
// 64 x 16 B to a wave-uniform LDS base + lane*16.  Completion: vmcnt, drained by the __syncthreads() that
// precedes the first read.
template <int SLICE>
__device__ __forceinline__ void slice_issue(const double *src, double *s_dst)
{
    constexpr int PIECES = SLICE / 2;                       // 16-byte pieces
    // the wave's number as a SCALAR: the "does this wave-instruction exist" test below is then a scalar branch instead of a
    // v_cmp + saveexec per piece, and the LDS base needs no v_readfirstlane
    const int wbase = __builtin_amdgcn_readfirstlane(threadIdx.x & ~(WAVE - 1));
    // per-lane source offset in bytes as an unsigned 32-bit value: uniform base + zero-extended offset selects the
    // saddr + voffset form of the load (no 64-bit vector add per piece)
    const unsigned loff = (unsigned)threadIdx.x * 16u;
    const char *base = (const char *)src;
#pragma unroll
    for (int j = 0; j * NT < PIECES; ++j) {
        const int e0 = j * NT + wbase;                      // first piece of this wave-instruction (uniform)
        if (e0 < PIECES)
            __builtin_amdgcn_global_load_lds((const void *)(base + (size_t)j * NT * 16 + loff),
                                             (__attribute__((address_space(3))) void *)(s_dst + 2 * e0), 16, 0, 0);
    }
}

template <bool SW, bool LUT, bool VIS>
__global__ __launch_bounds__(NT, AERO_MIN_WAVES) void modal_aero_kernel(AeroDev A, int ncol, int pver, int bands_per_group,
                                                       colphys_modal_aero_io io)
{
    constexpr int NTAB = SW ? 3 : 1;
    constexpr int NOUT = SW ? 4 : 1;
    constexpr int SLICE = aero_slice_doubles(NTAB);
    static_assert(SLICE % 128 == 0, "slice must be whole 1 KiB pieces");
    __builtin_amdgcn_s_setprio(AERO_PRIO);
    extern __shared__ double smem[];
    double *s_slice0 = smem;                           // two slices (double buffer)
    double *s_acc = smem + 2 * SLICE;                  // [bands_per_group][NOUT][NT]

    const int nb = SW ? A.nswbands : A.nlwbands;
    const int b0 = blockIdx.y * bands_per_group;
    const int nbg = min(bands_per_group, nb - b0);     // bands of this group
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const size_t cell = (size_t)blockIdx.x * NT + threadIdx.x;
    const bool live = cell < F;
    const size_t cc = live ? cell : F - 1;             // lanes past the end shadow the last cell (never stored)
    const int col = (int)(cc / L);
    const int k = (int)(cc - (size_t)col * L) + 1;     // 1-based level
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;
    const double *slices = SW ? A.slices_sw : A.slices_lw;

    slice_issue<SLICE>(slices + ((size_t)0 * nb + b0) * SLICE, s_slice0);
    for (int i = threadIdx.x; i < nbg * NOUT * NT; i += NT) s_acc[i] = 0.0;

    const double mass = io.pdeldry[cc] * A.rga;        // :644 / :1300
    int bad = 0;                                       // units of this cell outside the sanity range of the optical depth (nerr_dopaer)
    unsigned long long negm = 0;                       // lanes that met one < -1e-10, where the reference halts (:1024, :1431; exactly -1e-10 is only counted, :997): wave-uniform, in SGPRs
    __syncthreads();
    int cur = 0;

    const int nunits = A.nmodes * nbg;
    int unit = 0;
    for (int m = 0; m < A.nmodes; ++m) {                                // :706 / :1328
        const int nspec = as_const(A.nspec)[m];
        double vol[NSMAX], dryvol = 0.0, radsurf, logradsurf, cheb[NC];
        const double dgn = io.dgnumwet[m * F + cc];
        const double qw = io.qaerwat[m * F + cc];
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {                               // loads first, arithmetic after
            vol[l] = 0.0;
            if (l < nspec) vol[l] = io.specmmr[((size_t)m * A.nspec_max + l) * F + cc];
        }
        size_parameters(A, as_const(A.explnsigma)[m], dgn, radsurf, logradsurf, cheb);
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {                               // :764-766
            if (l < nspec) {
                vol[l] = vol[l] * as_const(A.rspecdens)[m * A.nspec_max + l];     // specmmr/specdens (:765) as a product
                dryvol = dryvol + vol[l];
            }
        }
        double watervol = qw * rh2odens;                                // :835-844
        double wetvol = watervol + dryvol;
        if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
        const bool small = logradsurf <= A.xrmax;
        // cre/wetvol, cim/wetvol (:850 / :1377) as products with one reciprocal per (cell, mode): last-bit difference
        const double wetinv = 1.0 / (SW ? fmax(wetvol, 1.e-60) : wetvol);

        for (int jb = 0; jb < nbg; ++jb, ++unit) {                      // :726 / :1343
            const int ib = b0 + jb;
            const double *s_tbl = s_slice0 + cur * SLICE;
            const size_t gsl = (size_t)m * nb + ib;
            // the next (mode, band) slice starts its way into the other buffer
            if (unit + 1 < nunits) {
                const int jn = (jb + 1 < nbg) ? jb + 1 : 0;
                const int mn = (jb + 1 < nbg) ? m : m + 1;
                slice_issue<SLICE>(slices + ((size_t)mn * nb + b0 + jn) * SLICE, s_slice0 + (cur ^ 1) * SLICE);
            }
            // volume-mixed complex refractive index :767, 846-850 / :1360, 1375-1378
            cptr_t ri = as_const((SW ? A.specrefindex_sw : A.specrefindex_lw) + ((size_t)m * A.nspec_max * nb + ib) * 2);
            double cre = 0.0, cim = 0.0;
            mix(nspec, [&](int l) { return vol[l]; }, ri, (size_t)nb * 2, cre, cim);
            cptr_t crefw = as_const((SW ? A.crefwsw : A.crefwlw) + ib * 2);
            cre = cre + watervol * crefw[0];
            cim = cim + watervol * crefw[1];
            double refr, refi;
            if (SW) {
                cre = cre * wetinv;
                cim = cim * wetinv;
                refr = cre; refi = fabs(cim);
            } else {
                const double qr = cre * wetinv, qi = cim * wetinv;
                const bool dv = wetvol > 1.e-40;
                refr = dv ? qr : cre; refi = dv ? qi : cim;
            }
            Wght wr, wi;
            unit_weights<NTAB, LUT>(s_tbl, slices + gsl * SLICE, refr, refi, wr, wi);   // :857-858 / :1383-1384

            if (SW) {
                double pext = interp_cheb(s_tbl, wr, wi, cheb);         // :860, 870-873
                double pabs = interp_cheb(s_tbl + TBLP, wr, wi, cheb);  // :861, 882-887
                const double pasm = interp_cheb(s_tbl + 2 * TBLP, wr, wi, cheb);   // :862
                const double pe = cp_exp(pext);                            // :869-877
                pext = small ? pe : 1.5 / (radsurf * A.rhoh2o);         // :875 (particles beyond the table: rare)
                pext = pext * wetvol * A.rhoh2o;                        // :881-894
                pabs = pabs * wetvol * A.rhoh2o;
                pabs = fmax(0.0, pabs);
                pabs = fmin(pext, pabs);
                const double dopaer = pext * mass;
                const double dw = (pext - pabs) * mass;                 // dopaer*palb without the quotient (see the bands-outer kernel)
                { const bool lo = dopaer <= -1.e-10, hi = dopaer >= 30.0; negm |= __ballot(dopaer < -1.e-10); bad += (lo | hi) ? 1 : 0; }       // :997
                if (VIS && ib == A.idx_sw_diag - 1 && live) {                     // hand-over to the diagnostics kernel
                    double *vo = io.vis_optics + (size_t)m * F + cc;
                    vo[0] = pext; vo[(size_t)A.nmodes * F] = pabs; vo[2 * (size_t)A.nmodes * F] = pasm;
                }
                double *acc = s_acc + (size_t)jb * 4 * NT + threadIdx.x;
                acc[0] = acc[0] + dopaer;                               // :1032-1037
                acc[NT] = acc[NT] + dw;
                acc[2 * NT] = acc[2 * NT] + dw * pasm;
                acc[3 * NT] = acc[3 * NT] + dw * pasm * pasm;
            } else {
                double pabs = interp_cheb(s_tbl, wr, wi, cheb);         // :1386, 1390-1393
                pabs = pabs * wetvol * A.rhoh2o;
                pabs = fmax(0.0, pabs);
                const double dopaer = pabs * mass;
                { const bool lo = dopaer <= -1.e-10, hi = dopaer >= 20.0; negm |= __ballot(dopaer < -1.e-10); bad += (lo | hi) ? 1 : 0; }       // :1402
                s_acc[(size_t)jb * NT + threadIdx.x] = s_acc[(size_t)jb * NT + threadIdx.x] + dopaer;   // :1441
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    // write-out: lanes are consecutive levels of a column => contiguous runs per (column, band); levels above
    // top_lev are not touched by the reference's mode loop and keep the initial zeros (:639-655 / :1312)
    if (live) {
        if (SW) {
            const size_t LP = L + 1;
            for (int jb = 0; jb < nbg; ++jb) {
                const size_t o = ((size_t)col * nb + (b0 + jb)) * LP + k;
                const double *acc = s_acc + (size_t)jb * 4 * NT + threadIdx.x;
                io.tauxar[o] = active ? acc[0] : 0.0;
                io.wa[o] = active ? acc[NT] : 0.0;
                io.ga[o] = active ? acc[2 * NT] : 0.0;
                io.fa[o] = active ? acc[3 * NT] : 0.0;
                if (k == 1) {                                           // zero'th layer :639-642
                    io.tauxar[o - 1] = 0.0; io.wa[o - 1] = 0.925; io.ga[o - 1] = 0.850; io.fa[o - 1] = 0.7225;
                }
            }
        } else {
            for (int jb = 0; jb < nbg; ++jb)
                io.tauxar[((size_t)col * nb + (b0 + jb)) * L + (k - 1)] = active ? s_acc[(size_t)jb * NT + threadIdx.x] : 0.0;
        }
        if (bad && active) status_report(io.status + col, bad, negm);
    }
}


// ------------------------------------------------------------------------------------------------------
// "Bands outer, modes inner", SW and LW (the default for up to 4 modes; colphys_aero_kernel_select): the
// per-(cell, mode) state of ALL modes is computed once and kept -- Chebyshev terms, water / wet volume in registers, the
// species volumes in a lane-private LDS array [species slot][cell] (packed through the modes: sum of nspec slots) -- the
// four running sums of one band are four registers, and a band is written out as soon as its last mode has been added
// (same order of additions per output element as the reference's modes-outer loop).  No LDS accumulators, no band
// groups, no recomputed prologue: 16 % fewer VALU instructions than five band groups.  With the volumes in registers
// the SW kernel needed 247 VGPRs (two waves per SIMD, 1.32 ms); with them in LDS it fits 168 (three waves, 1.24 ms; 0.955 ms
// after the lookup / mixing / division work of round 2).  LW: one table, one running sum, 109 VGPRs, four waves per SIMD.
// ------------------------------------------------------------------------------------------------------
constexpr int NMMAX = 4;

template <bool SW, bool LUT, bool VIS>
__global__ __launch_bounds__(NT, SW ? AERO_SW_WAVES : AERO_LW_WAVES) void modal_aero_reg_kernel(AeroDev A, int ncol, int pver, colphys_modal_aero_io io)
{
    constexpr int NTAB = SW ? 3 : 1;
    constexpr int SLICE = aero_slice_doubles(NTAB);
    // BATCH: fewer dependent LDS round trips per unit -- both grid lookups side by side, the four corners of a table read
    // before their sums (17 -> 7 round trips), paid for with registers freed by recomputing the Chebyshev terms per unit.
    // SW: -2.5 % (0.98 -> 0.96 ms; the kernel holds VALU and LDS pipe ~55 % busy each, latency is not what bounds it).
    // LW, whose unit has one table: +1..2 %, so not there.
    constexpr bool BATCH = SW ? (AERO_BATCH_SW != 0) : (AERO_BATCH_LW != 0);
    __builtin_amdgcn_s_setprio(AERO_PRIO);
    extern __shared__ double smem[];
    double *s_slice0 = smem;                           // two slices (double buffer)
    double *s_vol = smem + 2 * SLICE + threadIdx.x;    // species volumes [slot][NT], slot = species counted through the modes

    const int nb = SW ? A.nswbands : A.nlwbands;
    const size_t L = (size_t)pver, F = (size_t)ncol * L, LP = L + 1;
    const size_t cell = (size_t)blockIdx.x * NT + threadIdx.x;
    const bool live = cell < F;
    const size_t cc = live ? cell : F - 1;
    const int col = (int)(cc / L);
    const int k = (int)(cc - (size_t)col * L) + 1;
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;
    const double *slices = SW ? A.slices_sw : A.slices_lw;

    slice_issue<SLICE>(slices, s_slice0);              // (mode 0, band 0)

    const double mass = io.pdeldry[cc] * A.rga;        // :644 / :1300
    int bad = 0;                                       // units of this cell outside the sanity range of the optical depth (nerr_dopaer)
    unsigned long long negm = 0;                       // lanes that met one < -1e-10, where the reference halts (:1024, :1431; exactly -1e-10 is only counted, :997): wave-uniform, in SGPRs

    double cheb[NMMAX][NC], watervol[NMMAX], wetvol[NMMAX], wetinv[NMMAX], radsurf[NMMAX];
    // species counts: scalar loads, once.  (Read as A.nspec[m] inside the band loop they were VECTOR loads of a uniform
    // value -- global_load + s_waitcnt vmcnt(0) + v_readfirstlane at the head of every (band, mode) unit.)
    int nsp[NMMAX];
#pragma unroll
    for (int m = 0; m < NMMAX; ++m) nsp[m] = (m < A.nmodes) ? as_const(A.nspec)[m] : 0;
    int voff[NMMAX + 1];
    voff[0] = 0;
    bool small[NMMAX];
#pragma unroll
    for (int m = 0; m < NMMAX; ++m) {
        watervol[m] = 0.0; wetvol[m] = 0.0; wetinv[m] = 0.0; radsurf[m] = 1.0; small[m] = true;
        voff[m + 1] = voff[m];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) cheb[m][nc] = 0.0;
        if (m < A.nmodes) {
            const int nspec = nsp[m];
            voff[m + 1] = voff[m] + nspec;
            const double dgn = io.dgnumwet[m * F + cc];
            const double qw = io.qaerwat[m * F + cc];
            double v[NSMAX];
#pragma unroll
            for (int l = 0; l < NSMAX; ++l)
                if (l < nspec) v[l] = io.specmmr[((size_t)m * A.nspec_max + l) * F + cc];
            double logradsurf;
            size_parameters(A, as_const(A.explnsigma)[m], dgn, radsurf[m], logradsurf, cheb[m]);
            double dryvol = 0.0;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec) {
                    const double vl = v[l] * as_const(A.rspecdens)[m * A.nspec_max + l];
                    s_vol[(voff[m] + l) * NT] = vl;             // own lane's column of the LDS array: no barrier needed
                    dryvol = dryvol + vl;
                }
            }
            watervol[m] = qw * rh2odens;
            wetvol[m] = watervol[m] + dryvol;
            if (watervol[m] < 0.0) { watervol[m] = 0.0; wetvol[m] = dryvol; }
            small[m] = logradsurf <= A.xrmax;
            wetinv[m] = 1.0 / (SW ? fmax(wetvol[m], 1.e-60) : wetvol[m]);       // :848 / :1377
        }
    }
    __syncthreads();
    // Hand-over of the visible-band optics (VIS): the three values of a unit are stored at the START of the next unit, i.e.
    // right behind the barrier, not in front of it (stores count in vmcnt like the slice DMA, and a barrier waits for
    // vmcnt(0)).  The variant measures 3-5 % slower than the plain kernel alone, whatever the placement of the stores.
    double hv0 = 0.0, hv1 = 0.0, hv2 = 0.0;
    int hm = -1;                                        // mode whose values are pending (wave-uniform)
    auto hand_over_store = [&]() {
        if (live) {
            // uniform plane pointer + 32-bit byte offset of the cell (recomputed from col, k: nothing extra stays live)
            unsigned boff = ((unsigned)col * (unsigned)pver + (unsigned)(k - 1)) * 8u;
            asm volatile("" : "+v"(boff));              // opaque: or the store addresses are hoisted out of the band loop (24 VGPRs)
            char *vo = (char *)(io.vis_optics + (size_t)hm * F);
            const size_t VS = (size_t)A.nmodes * F * sizeof(double);
            *(double *)(vo + boff) = hv0; *(double *)(vo + VS + boff) = hv1; *(double *)(vo + 2 * VS + boff) = hv2;
        }
        hm = -1;
    };
    int cur = 0;
    for (int ib = 0; ib < nb; ++ib) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int m = 0; m < NMMAX; ++m) {
            if (m < A.nmodes) {
                const int nspec = nsp[m];
                const double *s_tbl = s_slice0 + cur * SLICE;
                const size_t gsl = (size_t)m * nb + ib;
                if (VIS && hm >= 0) hand_over_store();
                {   // next slice: (m+1, ib) or (0, ib+1)
                    const int mn = (m + 1 < A.nmodes) ? m + 1 : 0;
                    const int ibn = (m + 1 < A.nmodes) ? ib : ib + 1;
                    if (ibn < nb) slice_issue<SLICE>(slices + ((size_t)mn * nb + ibn) * SLICE, s_slice0 + (cur ^ 1) * SLICE);
                }
                cptr_t ri = as_const((SW ? A.specrefindex_sw : A.specrefindex_lw) + ((size_t)m * A.nspec_max * nb + ib) * 2);
                double cre = 0.0, cim = 0.0;
                const double *s_vm = s_vol + voff[m] * NT;
                mix(nspec, [&](int l) { return s_vm[l * NT]; }, ri, (size_t)nb * 2, cre, cim);
                cptr_t crefw = as_const((SW ? A.crefwsw : A.crefwlw) + ib * 2);
                cre = cre + watervol[m] * crefw[0];
                cim = cim + watervol[m] * crefw[1];
                double refr, refi;
                if (SW) {
                    refr = cre * wetinv[m]; refi = fabs(cim * wetinv[m]);
                } else {
                    const bool dv = wetvol[m] > 1.e-40;                 // :1376
                    refr = dv ? cre * wetinv[m] : cre; refi = dv ? cim * wetinv[m] : cim;
                }
                Wght wr, wi;
                unit_weights<NTAB, LUT, BATCH>(s_tbl, slices + gsl * SLICE, refr, refi, wr, wi);
                double chm[NC];
                if (BATCH) {
                    // the Chebyshev terms of the mode from its x: three recurrences per unit instead of twelve registers held
                    // (the registers pay for the four corners in flight)
                    chm[0] = 1.0; chm[1] = cheb[m][1];
                    asm volatile("" : "+v"(chm[1]));    // opaque: or the terms are hoisted out of the band loop and held again
#pragma unroll
                    for (int nc = 2; nc < NC; ++nc) chm[nc] = 2.0 * chm[1] * chm[nc - 1] - chm[nc - 2];
                } else {
#pragma unroll
                    for (int nc = 0; nc < NC; ++nc) chm[nc] = cheb[m][nc];
                }
                if (SW) {
                    double p3[3];
                    if (BATCH) {                        // a table's four corners in one LDS round trip
                        interp_cheb_all<1>(s_tbl, wr, wi, chm, p3);
                        interp_cheb_all<1>(s_tbl + TBLP, wr, wi, chm, p3 + 1);
                        interp_cheb_all<1>(s_tbl + 2 * TBLP, wr, wi, chm, p3 + 2);
                    } else {
                        p3[0] = interp_cheb(s_tbl, wr, wi, chm);
                        p3[1] = interp_cheb(s_tbl + TBLP, wr, wi, chm);
                        p3[2] = interp_cheb(s_tbl + 2 * TBLP, wr, wi, chm);
                    }
                    double pext = p3[0], pabs = p3[1];
                    const double pasm = p3[2];
                    const double pe = cp_exp(pext);
                    pext = small[m] ? pe : 1.5 / (radsurf[m] * A.rhoh2o);
                    pext = pext * wetvol[m] * A.rhoh2o;
                    pabs = pabs * wetvol[m] * A.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    pabs = fmin(pext, pabs);
                    // dopaer*palb = pext*mass*(1 - pabs/max(pext, 1e-40)) (:892-894, 1034) without the quotient: mass*(pext - pabs).
                    // Equal to rounding for pext > 1e-40 (0 <= pabs <= pext, so neither form cancels worse than the other);
                    // below that both are < 1e-40 * mass.  A division sequence less per (band, mode) unit.
                    const double dopaer = pext * mass;
                    const double dw = (pext - pabs) * mass;
                    { const bool lo = dopaer <= -1.e-10, hi = dopaer >= 30.0; negm |= __ballot(dopaer < -1.e-10); bad += (lo | hi) ? 1 : 0; }
                    if (VIS && ib == A.idx_sw_diag - 1) {                         // hand-over to the diagnostics kernel (uniform test):
                        hv0 = pext; hv1 = pabs; hv2 = pasm; hm = m;               // stored behind this unit's barrier, see above
                    }
                    a0 = a0 + dopaer;
                    a1 = a1 + dw;
                    a2 = a2 + dw * pasm;
                    a3 = a3 + dw * pasm * pasm;
                } else {
                    double pabs;
                    if (BATCH) interp_cheb_all<1>(s_tbl, wr, wi, chm, &pabs);
                    else pabs = interp_cheb(s_tbl, wr, wi, chm);        // :1386, 1390-1393
                    pabs = pabs * wetvol[m] * A.rhoh2o;
                    pabs = fmax(0.0, pabs);
                    const double dopaer = pabs * mass;
                    { const bool lo = dopaer <= -1.e-10, hi = dopaer >= 20.0; negm |= __ballot(dopaer < -1.e-10); bad += (lo | hi) ? 1 : 0; }   // :1402
                    a0 = a0 + dopaer;                                   // :1441
                }
                __syncthreads();
                cur ^= 1;
            }
        }
        if (live) {
            if (SW) {
                const size_t o = ((size_t)col * nb + ib) * LP + k;
                io.tauxar[o] = active ? a0 : 0.0;
                io.wa[o] = active ? a1 : 0.0;
                io.ga[o] = active ? a2 : 0.0;
                io.fa[o] = active ? a3 : 0.0;
                if (k == 1) { io.tauxar[o - 1] = 0.0; io.wa[o - 1] = 0.925; io.ga[o - 1] = 0.850; io.fa[o - 1] = 0.7225; }
            } else {
                io.tauxar[((size_t)col * nb + ib) * L + (k - 1)] = active ? a0 : 0.0;
            }
        }
    }
    if (VIS && hm >= 0) hand_over_store();              // the visible band was the last one
    if (live && bad && active) status_report(io.status + col, bad, negm);
}

// ------------------------------------------------------------------------------------------------------
// History diagnostics of modal_aero_sw (modal_aer_opt.F90:647-678, 770-831, 896-993, 1046-1154; include/colphys.h
// colphys_modal_aero_sw_diag_dev).  A second, small kernel behind the radiative one -- a fourteenth of its table work:
//   * the visible band (idx_sw_diag) is evaluated again per mode, because the species partition (:936-990) and the
//     absorption / asymmetry sums need the per-mode pabs, pasm, palb that the radiative sums tau, tau*w, tau*w*g fold;
//   * AODTOT and the UV / near-IR fields are sums over modes of dopaer, i.e. the radiative kernel's tauxar: read back
//     from it (sum over modes first, then over bands / levels: a reassociation of the reference's mode-outer sums of
//     positive terms, at rounding level);
//   * a block owns whole columns (NT / pver of them; thread = (column, level)), so the column sums are ordered sums
//     over the levels of per-cell terms staged through LDS, eight fields per round, one summing thread per
//     (column, field); the per-cell terms add the modes in the reference's order.
// One slice (mode, visible band) in LDS at a time, copied by the LDS-DMA path like in the radiative kernels.
// ------------------------------------------------------------------------------------------------------
constexpr int DIAG_ROUND = 8;               // column fields per summing pass
constexpr int DIAG_NSLOT = 3 * DIAG_ROUND;  // per-cell terms staged per mode
constexpr int DIAG_CPB = 8;                 // at most this many columns per block: CPB x ROUND x PARTS = NT summing threads
constexpr int DIAG_PARTS = 4;               // threads per (column, field) sum
constexpr int DIAG_STRIDE = NT + 4;         // slot stride in doubles: slots land on different LDS banks for the summing reads
constexpr int NTYPE = 6;                    // sulfate, p-organic, s-organic, black-c, dust, seasalt (include/colphys.h)
#ifndef DIAG_WAVES
#define DIAG_WAVES 2                        // waves per SIMD the register budget is set for
#endif

// Column sums of one pass: every thread has stored its cell's terms in s[slot][DIAG_STRIDE]; thread (column rc, slot j of
// the pass, part q) adds the levels q, q+4, ... of its column in order and the four parts are added as (p0 + p1) + (p2 + p3).
// Every thread of a group returns the sum.  No barrier inside.
__device__ __forceinline__ double level_sums(const double *s, int nred, int pver)
{
    const int tid = threadIdx.x;
    const int q = tid & (DIAG_PARTS - 1), j = (tid >> 2) & (DIAG_ROUND - 1), rc = tid >> 5;
    double a = 0.0;
    if (rc < nred) {
        const double *p = s + j * DIAG_STRIDE + rc * pver;
        for (int k = q; k < pver; k += DIAG_PARTS) a = a + p[k];
    }
    a = a + __shfl_xor(a, 1);
    a = a + __shfl_xor(a, 2);
    return a;
}

// slots of the three per-mode rounds -> column field (-1: a per-mode field or unused)
__device__ __constant__ int diag_slot_field[3 * DIAG_ROUND] = {
    COLPHYS_SWD_COL_AODVIS, COLPHYS_SWD_COL_AODABS, COLPHYS_SWD_COL_SSAVIS, COLPHYS_SWD_COL_ASYMVIS, COLPHYS_SWD_COL_AODABSBC,
    COLPHYS_SWD_COL_AODVISST, -1, -1,
    COLPHYS_SWD_COL_BURDEN0, COLPHYS_SWD_COL_BURDEN0 + 1, COLPHYS_SWD_COL_BURDEN0 + 2, COLPHYS_SWD_COL_BURDEN0 + 3,
    COLPHYS_SWD_COL_BURDEN0 + 4, COLPHYS_SWD_COL_BURDEN0 + 5, -1, -1,
    COLPHYS_SWD_COL_AOD0, COLPHYS_SWD_COL_AOD0 + 1, COLPHYS_SWD_COL_AOD0 + 2, COLPHYS_SWD_COL_AOD0 + 3, COLPHYS_SWD_COL_AOD0 + 4,
    COLPHYS_SWD_COL_AOD0 + 5, -1, -1};

template <bool LUT>
__global__ __launch_bounds__(NT, DIAG_WAVES) void modal_aero_sw_diag_kernel(AeroDev A, int ncol, int pver, int cpb, colphys_modal_aero_io io,
                                                                             double *__restrict__ lev, double *__restrict__ colf)
{
    constexpr int NTAB = 3;
    constexpr int SLICE = aero_slice_doubles(NTAB);
    extern __shared__ double smem[];
    double *s_slice0 = smem;                           // the (mode, visible band) slices, double-buffered
    double *s_red = smem + 2 * SLICE;                  // [DIAG_NSLOT][DIAG_STRIDE] per-cell terms of the fields being summed

    const int nb = A.nswbands, ivis = A.idx_sw_diag - 1;
    const size_t L = (size_t)pver, F = (size_t)ncol * L, LP = L + 1;
    const int tid = threadIdx.x;
    const int cl = tid / pver;                         // column of the block
    const int k = tid - cl * pver + 1;                 // 1-based level
    const int col = blockIdx.x * cpb + cl;
    const bool live = cl < cpb && col < ncol;
    const size_t cc = live ? (size_t)col * L + (k - 1) : F - 1;       // idle lanes shadow the last cell (never stored)
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;
    const double *slices = A.slices_sw;
    // summing role of this thread (level_sums): column rc of the block, slot j of a round, part q
    const int sq = tid & (DIAG_PARTS - 1), sj = (tid >> 2) & (DIAG_ROUND - 1), src = tid >> 5;
    const int nred = min(cpb, ncol - (int)blockIdx.x * cpb);            // columns of this block that exist
    const bool writer = sq == 0 && src < nred;
    const size_t wcol = (size_t)blockIdx.x * cpb + src;

    slice_issue<SLICE>(slices + (size_t)ivis * SLICE, s_slice0);       // mode 0

    const double mass = io.pdeldry[cc] * A.rga;                         // :644
    const double air_density = io.pmid[cc] / (A.rair * io.t[cc]);       // :645
    const double ad_mass = air_density / mass;                          // x*air_density/mass (:901, 911, 924, 931) as one product
    const bool strat = live && k <= io.troplevchem[live ? col : 0];     // :903, 913, 932

    double extinct = 0.0, absorb = 0.0, asymext = 0.0;                  // level fields: the cell's sums over the modes
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;                          // summing threads: column sums over the modes, one per round

    // A mode's cell inputs are loaded one mode ahead, right AFTER the barrier that waits for the slice: LDS-DMA and global
    // loads share vmcnt, so anything issued before that barrier is waited for at it; issued after it, the loads have the
    // whole table / partition phase of the current mode to land (two waves per SIMD hide little by themselves).
    double n_v[NSMAX], n_dgn, n_qw;
    auto load_mode = [&](int m) {
        const int nspec = as_const(A.nspec)[m];
        n_dgn = io.dgnumwet[m * F + cc];
        n_qw = io.qaerwat[m * F + cc];
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {
            n_v[l] = 0.0;
            if (l < nspec) n_v[l] = io.specmmr[((size_t)m * A.nspec_max + l) * F + cc];
        }
    };
    load_mode(0);

    int cur = 0;
    for (int m = 0; m < A.nmodes; ++m) {                                // :706
        const int nspec = as_const(A.nspec)[m];
        const double *s_tbl = s_slice0 + cur * SLICE;

        double v[NSMAX], radsurf, logradsurf, cheb[NC];
        const double dgn = n_dgn, qw = n_qw;
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) v[l] = n_v[l];
        size_parameters(A, as_const(A.explnsigma)[m], dgn, radsurf, logradsurf, cheb);

        // species loop :758-768 at the visible band; v[l] becomes the species volume
        cptr_t ri = as_const(A.specrefindex_sw + ((size_t)m * A.nspec_max * nb + ivis) * 2);
        double cre = 0.0, cim = 0.0, dryvol = 0.0, burden = 0.0;
        double bur[NTYPE];
#pragma unroll
        for (int ty = 0; ty < NTYPE; ++ty) bur[ty] = 0.0;
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {
            if (l < nspec) {
                const double bm = active ? v[l] * mass : 0.0;           // :777
                burden = burden + bm;
                const int sty = as_const(A.spectype)[m * A.nspec_max + l];
#pragma unroll
                for (int ty = 0; ty < NTYPE; ++ty) bur[ty] = bur[ty] + ((sty == ty) ? 1.0 : 0.0) * bm;   // :782-824, scalar weights
                v[l] = v[l] * as_const(A.rspecdens)[m * A.nspec_max + l];     // :765 as a product, like the radiative kernel
                dryvol = dryvol + v[l];
                cre = cre + v[l] * ri[(size_t)l * nb * 2];
                cim = cim + v[l] * ri[(size_t)l * nb * 2 + 1];
            }
        }
        double watervol = qw * rh2odens;                                // :835-844
        double wetvol = watervol + dryvol;
        if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
        const bool small = logradsurf <= A.xrmax;
        const double wetinv = 1.0 / fmax(wetvol, 1.e-60);
        cptr_t crefw = as_const(A.crefwsw + ivis * 2);
        const double cwr = crefw[0], cwi = crefw[1];
        cre = (cre + watervol * cwr) * wetinv;                          // :846-850
        cim = (cim + watervol * cwi) * wetinv;
        const double refr = cre, refi = fabs(cim);

        __syncthreads();                                                // this mode's slice has landed; the other buffer is free
        if (m + 1 < A.nmodes) {
            slice_issue<SLICE>(slices + ((size_t)(m + 1) * nb + ivis) * SLICE, s_slice0 + (cur ^ 1) * SLICE);
            load_mode(m + 1);
        }
        Wght wr, wi;
        unit_weights<NTAB, LUT>(s_tbl, slices + ((size_t)m * nb + ivis) * SLICE, refr, refi, wr, wi);    // :857-862
        double pext = interp_cheb(s_tbl, wr, wi, cheb);
        double pabs = interp_cheb(s_tbl + TBLP, wr, wi, cheb);
        const double pasm = interp_cheb(s_tbl + 2 * TBLP, wr, wi, cheb);
        const double pe = cp_exp(pext);                                    // :869-877
        if (!small) pext = 1.5 / (radsurf * A.rhoh2o); else pext = pe;
        pext = pext * wetvol * A.rhoh2o;                                // :881-894
        pabs = pabs * wetvol * A.rhoh2o;
        pabs = fmax(0.0, pabs);
        pabs = fmin(pext, pabs);
        const double palb = 1.0 - pabs / fmax(pext, 1.e-40);
        const double dopaer = active ? pext * mass : 0.0;

        extinct = extinct + dopaer * ad_mass;                           // :921-934
        absorb = absorb + pabs * air_density;
        asymext = asymext + dopaer * pasm * ad_mass;

        // per-cell terms of this mode: slots 0..5 plain visible-band sums, 6, 7 two of the three per-mode fields
        // (:1050-1057; the third, aodmode, is slot 0), 8..13 burdens by type, 16..21 optical depths by type
        s_red[0 * DIAG_STRIDE + tid] = dopaer;                          // aodvis, aodmode
        s_red[1 * DIAG_STRIDE + tid] = active ? pabs * mass : 0.0;      // aodabs
        s_red[2 * DIAG_STRIDE + tid] = dopaer * palb;                   // ssavis
        s_red[3 * DIAG_STRIDE + tid] = dopaer * pasm;                   // asymvis
        s_red[5 * DIAG_STRIDE + tid] = strat ? dopaer : 0.0;            // aodvisst
        s_red[6 * DIAG_STRIDE + tid] = burden;
#pragma unroll
        for (int ty = 0; ty < NTYPE; ++ty) s_red[(DIAG_ROUND + ty) * DIAG_STRIDE + tid] = bur[ty];

        // species partition :936-990.  Only the LAST species of a type enters (the reference assigns, :783-827).
        // sums first, then each kept species' share straight into its type's term (scalar 0/1 weights: sty is uniform)
        double aod[NTYPE], absbc = 0.0, dustaod_m = 0.0;
#pragma unroll
        for (int ty = 0; ty < NTYPE; ++ty) aod[ty] = 0.0;
        if (wetvol > 1.e-40) {
            const int lastmask = as_const(A.lastmask)[m];
            const double scath2o = watervol * cwr, absh2o = -(watervol * cwi);
            double sumscat = 0.0, sumabs = 0.0, sumhygro = 0.0;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec && ((lastmask >> l) & 1)) {
                    sumscat = sumscat + v[l] * ri[(size_t)l * nb * 2];
                    sumabs = sumabs - v[l] * ri[(size_t)l * nb * 2 + 1];
                    sumhygro = sumhygro + v[l] * as_const(A.hygro)[m * A.nspec_max + l];
                }
            }
            sumscat = sumscat + scath2o;
            sumabs = sumabs + absh2o;
            // the eighteen quotients of :952-968 as products with three reciprocals
            const double rscat = 1.0 / sumscat, rabs = 1.0 / sumabs, rhyg = 1.0 / sumhygro;
            const double wsc = scath2o * rhyg, wab = absh2o * rhyg;
            const double ssa_d = palb * dopaer * rscat, abs_d = (1.0 - palb) * dopaer * rabs;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec && ((lastmask >> l) & 1)) {
                    const int sty = as_const(A.spectype)[m * A.nspec_max + l];
                    const double hy = as_const(A.hygro)[m * A.nspec_max + l];
                    const double ab = (wab * hy - ri[(size_t)l * nb * 2 + 1]) * v[l] * abs_d;       // absX*(1-palb)*dopaer, :953-972
                    const double sc = (wsc * hy + ri[(size_t)l * nb * 2]) * v[l] * ssa_d;           // scatX*palb*dopaer
#pragma unroll
                    for (int ty = 0; ty < NTYPE; ++ty) aod[ty] = aod[ty] + ((sty == ty) ? 1.0 : 0.0) * (ab + sc);
                    absbc = absbc + ((sty == 3) ? 1.0 : 0.0) * ab;      // :970
                    dustaod_m = dustaod_m + ((sty == 4) ? 1.0 : 0.0) * (dopaer * v[l] * wetinv);   // :938, /wetvol = *wetinv here
                }
            }
        }
        s_red[4 * DIAG_STRIDE + tid] = absbc;
        s_red[7 * DIAG_STRIDE + tid] = dustaod_m;
#pragma unroll
        for (int ty = 0; ty < NTYPE; ++ty) s_red[(2 * DIAG_ROUND + ty) * DIAG_STRIDE + tid] = aod[ty];
        __syncthreads();
        {
            const double a = level_sums(s_red, nred, pver);
            if (sj < 6) acc0 = acc0 + a;
            if (writer && sj == 0) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 1) * ncol + wcol] = a;    // aodmode
            if (writer && sj == 6) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 0) * ncol + wcol] = a;    // burden
            if (writer && sj == 7) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 2) * ncol + wcol] = a;    // dustaodmode
            if (sj < NTYPE) {
                acc1 = acc1 + level_sums(s_red + DIAG_ROUND * DIAG_STRIDE, nred, pver);
                acc2 = acc2 + level_sums(s_red + 2 * DIAG_ROUND * DIAG_STRIDE, nred, pver);
            }
        }
        // (the next mode's first barrier -- its slice -- also orders these reads before the next stores)
        cur ^= 1;
    }

    if (live) {
        const size_t o = (size_t)col * L + (k - 1);
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCT * F + o] = active ? extinct : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_ABSORB * F + o] = active ? absorb : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_ASYMEXT * F + o] = active ? asymext : 0.0;
    }
    if (writer) {
        const int f0 = diag_slot_field[sj], f1 = diag_slot_field[DIAG_ROUND + sj], f2 = diag_slot_field[2 * DIAG_ROUND + sj];
        if (f0 >= 0) colf[(size_t)f0 * ncol + wcol] = acc0;
        if (f1 >= 0) colf[(size_t)f1 * ncol + wcol] = acc1;
        if (f2 >= 0) colf[(size_t)f2 * ncol + wcol] = acc2;
    }
}

// The same diagnostics from the visible-band optics the radiative kernel left in io.vis_optics (pext, pabs, pasm per mode):
// no tables, no slices, no size parameters -- what remains per mode is the species bookkeeping, the partition and the
// column sums, at a third of the registers (five waves per SIMD instead of two).
__global__ __launch_bounds__(NT, 4) void modal_aero_sw_diag_vis_kernel(AeroDev A, int ncol, int pver, int cpb, colphys_modal_aero_io io,
                                                                        double *__restrict__ lev, double *__restrict__ colf)
{
    extern __shared__ double smem[];
    double *s_red = smem;                              // [DIAG_ROUND][DIAG_STRIDE]: 17 KB, occupancy is then the registers'
    const int nb = A.nswbands, ivis = A.idx_sw_diag - 1;
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const int tid = threadIdx.x;
    const int cl = tid / pver, k = tid - cl * pver + 1;
    const int col = blockIdx.x * cpb + cl;
    const bool live = cl < cpb && col < ncol;
    const size_t cc = live ? (size_t)col * L + (k - 1) : F - 1;
    const bool active = live && k >= A.top_lev;
    const double rh2odens = 1.0 / A.rhoh2o;
    const int sq = tid & (DIAG_PARTS - 1), sj = (tid >> 2) & (DIAG_ROUND - 1), src = tid >> 5;
    const int nred = min(cpb, ncol - (int)blockIdx.x * cpb);
    const bool writer = sq == 0 && src < nred;
    const size_t wcol = (size_t)blockIdx.x * cpb + src;

    const double mass = io.pdeldry[cc] * A.rga;                         // :644
    const double air_density = io.pmid[cc] / (A.rair * io.t[cc]);       // :645
    const double ad_mass = air_density / mass;
    const bool strat = live && k <= io.troplevchem[live ? col : 0];
    // AODTOT and the UV / near-IR fields (:896-917) are sums over the modes of dopaer, i.e. the radiative kernel's tauxar
    double t_tot = 0.0, t_uv = 0.0, t_nir = 0.0;
    if (active) {
        const double *tau = io.tauxar + (size_t)col * nb * (L + 1) + k;
        for (int ib = 0; ib < nb; ++ib) t_tot = t_tot + tau[(size_t)ib * (L + 1)];
        t_uv = tau[(size_t)(A.idx_uv_diag - 1) * (L + 1)];
        t_nir = tau[(size_t)(A.idx_nir_diag - 1) * (L + 1)];
    }
    double extinct = 0.0, absorb = 0.0, asymext = 0.0;
    double acc0 = 0.0;
    // burdens and optical depths by species type: the cell's sums over the modes stay in registers and the column sums
    // are taken once after the mode loop (levels outer, modes inner: a reassociation of the reference's mode-outer sums
    // of non-negative terms); the plain visible-band sums and the per-mode fields are summed mode by mode
    double c_bur[NTYPE], c_aod[NTYPE];
#pragma unroll
    for (int ty = 0; ty < NTYPE; ++ty) { c_bur[ty] = 0.0; c_aod[ty] = 0.0; }
    const size_t VS = (size_t)A.nmodes * F;

    for (int m = 0; m < A.nmodes; ++m) {                                // :706
        const int nspec = as_const(A.nspec)[m];
        const double qw = io.qaerwat[m * F + cc];
        const double pext = io.vis_optics[m * F + cc], pabs = io.vis_optics[VS + m * F + cc], pasm = io.vis_optics[2 * VS + m * F + cc];
        double v[NSMAX];
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {
            v[l] = 0.0;
            if (l < nspec) v[l] = io.specmmr[((size_t)m * A.nspec_max + l) * F + cc];
        }
        cptr_t ri = as_const(A.specrefindex_sw + ((size_t)m * A.nspec_max * nb + ivis) * 2);
        double dryvol = 0.0, burden = 0.0;
#pragma unroll
        for (int l = 0; l < NSMAX; ++l) {
            if (l < nspec) {
                const double bm = active ? v[l] * mass : 0.0;           // :777
                burden = burden + bm;
                const int sty = as_const(A.spectype)[m * A.nspec_max + l];
#pragma unroll
                for (int ty = 0; ty < NTYPE; ++ty) c_bur[ty] = c_bur[ty] + ((sty == ty) ? 1.0 : 0.0) * bm;
                v[l] = v[l] * as_const(A.rspecdens)[m * A.nspec_max + l];
                dryvol = dryvol + v[l];
            }
        }
        double watervol = qw * rh2odens;                                // :835-844
        double wetvol = watervol + dryvol;
        if (watervol < 0.0) { watervol = 0.0; wetvol = dryvol; }
        const double wetinv = 1.0 / fmax(wetvol, 1.e-60);
        cptr_t crefw = as_const(A.crefwsw + ivis * 2);
        const double cwr = crefw[0], cwi = crefw[1];
        const double palb = 1.0 - pabs / fmax(pext, 1.e-40);
        const double dopaer = active ? pext * mass : 0.0;

        extinct = extinct + dopaer * ad_mass;                           // :921-934
        absorb = absorb + pabs * air_density;
        asymext = asymext + dopaer * pasm * ad_mass;
        if (m > 0) __syncthreads();                                     // the previous mode's sums have been taken
        s_red[0 * DIAG_STRIDE + tid] = dopaer;
        s_red[1 * DIAG_STRIDE + tid] = active ? pabs * mass : 0.0;
        s_red[2 * DIAG_STRIDE + tid] = dopaer * palb;
        s_red[3 * DIAG_STRIDE + tid] = dopaer * pasm;
        s_red[5 * DIAG_STRIDE + tid] = strat ? dopaer : 0.0;
        s_red[6 * DIAG_STRIDE + tid] = burden;
        double absbc = 0.0, dustaod_m = 0.0;
        if (wetvol > 1.e-40) {                                          // :936-990, as in the kernel above
            const int lastmask = as_const(A.lastmask)[m];
            const double scath2o = watervol * cwr, absh2o = -(watervol * cwi);
            double sumscat = 0.0, sumabs = 0.0, sumhygro = 0.0;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec && ((lastmask >> l) & 1)) {
                    sumscat = sumscat + v[l] * ri[(size_t)l * nb * 2];
                    sumabs = sumabs - v[l] * ri[(size_t)l * nb * 2 + 1];
                    sumhygro = sumhygro + v[l] * as_const(A.hygro)[m * A.nspec_max + l];
                }
            }
            sumscat = sumscat + scath2o;
            sumabs = sumabs + absh2o;
            const double rscat = 1.0 / sumscat, rabs = 1.0 / sumabs, rhyg = 1.0 / sumhygro;
            const double wsc = scath2o * rhyg, wab = absh2o * rhyg;
            const double ssa_d = palb * dopaer * rscat, abs_d = (1.0 - palb) * dopaer * rabs;
#pragma unroll
            for (int l = 0; l < NSMAX; ++l) {
                if (l < nspec && ((lastmask >> l) & 1)) {
                    const int sty = as_const(A.spectype)[m * A.nspec_max + l];
                    const double hy = as_const(A.hygro)[m * A.nspec_max + l];
                    const double ab = (wab * hy - ri[(size_t)l * nb * 2 + 1]) * v[l] * abs_d;
                    const double sc = (wsc * hy + ri[(size_t)l * nb * 2]) * v[l] * ssa_d;
#pragma unroll
                    for (int ty = 0; ty < NTYPE; ++ty) c_aod[ty] = c_aod[ty] + ((sty == ty) ? 1.0 : 0.0) * (ab + sc);
                    absbc = absbc + ((sty == 3) ? 1.0 : 0.0) * ab;
                    dustaod_m = dustaod_m + ((sty == 4) ? 1.0 : 0.0) * (dopaer * v[l] * wetinv);
                }
            }
        }
        s_red[4 * DIAG_STRIDE + tid] = absbc;
        s_red[7 * DIAG_STRIDE + tid] = dustaod_m;
        __syncthreads();
        {
            const double a = level_sums(s_red, nred, pver);
            if (sj < 6) acc0 = acc0 + a;
            if (writer && sj == 0) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 1) * ncol + wcol] = a;
            if (writer && sj == 6) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 0) * ncol + wcol] = a;
            if (writer && sj == 7) colf[(size_t)(COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * m + 2) * ncol + wcol] = a;
        }
    }
    double acc1, acc2;
    __syncthreads();
#pragma unroll
    for (int ty = 0; ty < NTYPE; ++ty) s_red[ty * DIAG_STRIDE + tid] = c_bur[ty];
    __syncthreads();
    acc1 = (sj < NTYPE) ? level_sums(s_red, nred, pver) : 0.0;
    __syncthreads();
#pragma unroll
    for (int ty = 0; ty < NTYPE; ++ty) s_red[ty * DIAG_STRIDE + tid] = c_aod[ty];
    __syncthreads();
    acc2 = (sj < NTYPE) ? level_sums(s_red, nred, pver) : 0.0;
    __syncthreads();
    s_red[0 * DIAG_STRIDE + tid] = t_tot;
    s_red[1 * DIAG_STRIDE + tid] = t_uv;
    s_red[2 * DIAG_STRIDE + tid] = t_nir;
    s_red[3 * DIAG_STRIDE + tid] = strat ? t_uv : 0.0;
    s_red[4 * DIAG_STRIDE + tid] = strat ? t_nir : 0.0;
    __syncthreads();
    const double acc3 = (sj < 5) ? level_sums(s_red, nred, pver) : 0.0;
    if (live) {
        const size_t o = (size_t)col * L + (k - 1);
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCT * F + o] = active ? extinct : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_ABSORB * F + o] = active ? absorb : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_ASYMEXT * F + o] = active ? asymext : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTUV * F + o] = active ? t_uv * ad_mass : 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTNIR * F + o] = active ? t_nir * ad_mass : 0.0;
    }
    if (writer && sj < 5) {
        const int f3 = sj == 0 ? COLPHYS_SWD_COL_AODTOT : sj == 1 ? COLPHYS_SWD_COL_AODUV : sj == 2 ? COLPHYS_SWD_COL_AODNIR
                     : sj == 3 ? COLPHYS_SWD_COL_AODUVST : COLPHYS_SWD_COL_AODNIRST;
        colf[(size_t)f3 * ncol + wcol] = acc3;
    }
    if (writer) {
        const int f0 = diag_slot_field[sj], f1 = diag_slot_field[DIAG_ROUND + sj], f2 = diag_slot_field[2 * DIAG_ROUND + sj];
        if (f0 >= 0) colf[(size_t)f0 * ncol + wcol] = acc0;
        if (f1 >= 0) colf[(size_t)f1 * ncol + wcol] = acc1;
        if (f2 >= 0) colf[(size_t)f2 * ncol + wcol] = acc2;
    }
}

// AODTOT and the UV / near-IR fields (:896-917): sums over the modes of dopaer, i.e. the radiative kernel's tauxar.  A
// streaming kernel of its own (same block shape and column sums as above) so that its fourteen loads per cell do not
// sit behind the table work in the two-waves-per-SIMD kernel above.
__global__ __launch_bounds__(NT) void modal_aero_sw_diag_tau_kernel(AeroDev A, int ncol, int pver, int cpb, colphys_modal_aero_io io,
                                                                     double *__restrict__ lev, double *__restrict__ colf)
{
    __shared__ double s_red[5 * DIAG_STRIDE];
    const int nb = A.nswbands;
    const size_t L = (size_t)pver, F = (size_t)ncol * L, LP = L + 1;
    const int tid = threadIdx.x;
    const int cl = tid / pver, k = tid - cl * pver + 1;
    const int col = blockIdx.x * cpb + cl;
    const bool live = cl < cpb && col < ncol;
    const bool active = live && k >= A.top_lev;
    const int sq = tid & (DIAG_PARTS - 1), sj = (tid >> 2) & (DIAG_ROUND - 1), src = tid >> 5;
    const int nred = min(cpb, ncol - (int)blockIdx.x * cpb);
    double t_tot = 0.0, t_uv = 0.0, t_nir = 0.0;
    bool strat = false;
    if (active) {
        const size_t cc = (size_t)col * L + (k - 1);
        const double *tau = io.tauxar + (size_t)col * nb * LP + k;
        for (int ib = 0; ib < nb; ++ib) t_tot = t_tot + tau[(size_t)ib * LP];
        t_uv = tau[(size_t)(A.idx_uv_diag - 1) * LP];
        t_nir = tau[(size_t)(A.idx_nir_diag - 1) * LP];
        strat = k <= io.troplevchem[col];                                // :903, 913
        const double ad_mass = io.pmid[cc] / (A.rair * io.t[cc]) / (io.pdeldry[cc] * A.rga);   // air_density/mass :644-645, 901, 911
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTUV * F + cc] = t_uv * ad_mass;
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTNIR * F + cc] = t_nir * ad_mass;
    } else if (live) {
        const size_t cc = (size_t)col * L + (k - 1);
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTUV * F + cc] = 0.0;
        lev[(size_t)COLPHYS_SWD_LEV_EXTINCTNIR * F + cc] = 0.0;
    }
    s_red[0 * DIAG_STRIDE + tid] = t_tot;
    s_red[1 * DIAG_STRIDE + tid] = t_uv;
    s_red[2 * DIAG_STRIDE + tid] = t_nir;
    s_red[3 * DIAG_STRIDE + tid] = strat ? t_uv : 0.0;
    s_red[4 * DIAG_STRIDE + tid] = strat ? t_nir : 0.0;
    __syncthreads();
    if (sj < 5) {
        const double a = level_sums(s_red, nred, pver);
        const int f = sj == 0 ? COLPHYS_SWD_COL_AODTOT : sj == 1 ? COLPHYS_SWD_COL_AODUV : sj == 2 ? COLPHYS_SWD_COL_AODNIR
                    : sj == 3 ? COLPHYS_SWD_COL_AODUVST : COLPHYS_SWD_COL_AODNIRST;
        if (sq == 0 && src < nred) colf[(size_t)f * ncol + (size_t)blockIdx.x * cpb + src] = a;
    }
}

// Diagnostic: table_interp's interval search of one grid of one slice, by the scan or by the bucket lookup, on arbitrary x
// (colphys_aero_grid_probe_dev; the tests walk every grid point and its neighbours in floating point through both).
__global__ void aero_grid_probe_kernel(const double *slice, int ntab, int which, int use_lut, long long n, const double *x,
                                       int *ix1, int *ix2, double *t)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *g_refr = slice + ntab * TBLP, *g_refi = g_refr + NR, *g_rdxr = g_refi + NI, *g_rdxi = g_rdxr + NR;
    const double *lut = slice + aero_lut_offset(ntab);
    const unsigned char *cnt = reinterpret_cast<const unsigned char *>(lut + 2 * AERO_LUTB);
    const int *par = reinterpret_cast<const int *>(lut + 2 * AERO_LUTB + 2 * AERO_LUTB / 8);
    Wght w;
    if (which == 0)
        w = use_lut ? calcwghts_lut<NR>(lut, cnt, par[0], par[1], g_refr, g_rdxr, x[i]) : calcwghts<NR>(as_const(g_refr), g_refr, g_rdxr, x[i]);
    else
        w = use_lut ? calcwghts_lut<NI>(lut + AERO_LUTB, cnt + AERO_LUTB, par[2], par[3], g_refi, g_rdxi, x[i])
                    : calcwghts<NI>(as_const(g_refi), g_refi, g_rdxi, x[i]);
    ix1[i] = w.ix1 + 1; ix2[i] = w.ix2 + 1; t[i] = w.wt2;
}

}  // namespace

int launch_aero_grid_probe(const AeroDev &A, int sw, int mode, int band, int which, int use_lut, long long n, const double *x,
                           int *ix1, int *ix2, double *t, hipStream_t st)
{
    const int ntab = sw ? 3 : 1, nb = sw ? A.nswbands : A.nlwbands;
    const double *slice = (sw ? A.slices_sw : A.slices_lw) + ((size_t)mode * nb + band) * aero_slice_doubles(ntab);
    if (n > 0) hipLaunchKernelGGL(aero_grid_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slice, ntab, which, use_lut, n, x, ix1, ix2, t);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

// kernel family per routine (colphys_aero_kernel_select): 0 automatic = bands-outer when the mode count allows
static int g_aero_kind[2] = {COLPHYS_AERO_KERNEL_AUTO, COLPHYS_AERO_KERNEL_AUTO};      // [0] SW, [1] LW
void aero_kernel_select(int sw_kind, int lw_kind) { g_aero_kind[0] = sw_kind; g_aero_kind[1] = lw_kind; }

static int launch_aero(bool sw, const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st)
{
    if (A.nspec_max > NSMAX) return COLPHYS_ERR_UNSUPPORTED;
    const int nb = sw ? A.nswbands : A.nlwbands;
    const int ntab = sw ? 3 : 1, nout = sw ? 4 : 1;
    const size_t slice = (size_t)aero_slice_doubles(ntab);
    // band groups: largest group whose accumulators + 2 slices stay within AERO_LDS_CAP_KB (three blocks per CU)
    int ngroups = 1;
    for (;; ++ngroups) {
        const int bpg = (nb + ngroups - 1) / ngroups;
        if ((2 * slice + (size_t)bpg * nout * NT) * sizeof(double) <= (size_t)(sw ? AERO_LDS_CAP_KB : AERO_LW_LDS_CAP_KB) * 1024 || bpg == 1) break;
    }
    const int bpg = (nb + ngroups - 1) / ngroups;
    ngroups = (nb + bpg - 1) / bpg;
    const size_t lds = (2 * slice + (size_t)bpg * nout * NT) * sizeof(double);
    const size_t cells = (size_t)ncol * pver;
    const dim3 grid((unsigned)((cells + NT - 1) / NT), (unsigned)ngroups);
    if (hipMemsetAsync(io.status, 0, sizeof(int32_t) * ncol, st) != hipSuccess) return COLPHYS_ERR_HIP;
    const char *force_grouped = getenv(sw ? "COLPHYS_SW_GROUPED" : "COLPHYS_LW_GROUPED");      // tests: the kernel that serves more than NMMAX modes
    const bool reg = !(force_grouped && atoi(force_grouped)) && g_aero_kind[sw ? 0 : 1] != COLPHYS_AERO_KERNEL_GROUPED;
    if (A.nmodes <= NMMAX && reg) {
        const dim3 g1((unsigned)((cells + NT - 1) / NT));
        // LDS: the slice double buffer + the species volumes of all modes, [sum of nspec][NT] (nspec lives on the device:
        // size for the largest table the engine accepts would waste LDS, so the host mirror of nspec is used)
        const size_t lds_reg = (2 * slice + (size_t)A.nspec_total * NT) * sizeof(double);
        const bool vis = sw && io.vis_optics != nullptr;     // template parameter: the hand-over stores cost the plain kernel registers
        auto kern = sw ? (A.lut_ok_sw ? (vis ? modal_aero_reg_kernel<true, true, true> : modal_aero_reg_kernel<true, true, false>)
                                      : (vis ? modal_aero_reg_kernel<true, false, true> : modal_aero_reg_kernel<true, false, false>))
                       : (A.lut_ok_lw ? modal_aero_reg_kernel<false, true, false> : modal_aero_reg_kernel<false, false, false>);
        if (lds_reg > 64 * 1024)
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_reg);
        hipLaunchKernelGGL(kern, g1, dim3(NT), lds_reg, st, A, ncol, pver, io);
        return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
    }
    const bool vis = sw && io.vis_optics != nullptr;
    auto kern = sw ? (A.lut_ok_sw ? (vis ? modal_aero_kernel<true, true, true> : modal_aero_kernel<true, true, false>)
                                  : (vis ? modal_aero_kernel<true, false, true> : modal_aero_kernel<true, false, false>))
                   : (A.lut_ok_lw ? modal_aero_kernel<false, true, false> : modal_aero_kernel<false, false, false>);
    if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, A, ncol, pver, bpg, io);
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

int launch_modal_aero_sw_diag(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, double *lev, double *col,
                              hipStream_t st)
{
    if (A.nspec_max > NSMAX || pver > NT) return COLPHYS_ERR_UNSUPPORTED;
    const int cpb = std::min(NT / pver, DIAG_CPB);
    const size_t lds = (2 * (size_t)aero_slice_doubles(3) + (size_t)DIAG_NSLOT * DIAG_STRIDE) * sizeof(double);
    if (lds > 160 * 1024) return COLPHYS_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((ncol + cpb - 1) / cpb));
    if (io.vis_optics) {                    // the radiative kernel left the visible-band optics: no second table evaluation
        const size_t lds_v = (size_t)DIAG_ROUND * DIAG_STRIDE * sizeof(double);
        hipLaunchKernelGGL(modal_aero_sw_diag_vis_kernel, grid, dim3(NT), lds_v, st, A, ncol, pver, cpb, io, lev, col);
    } else {
        auto kern = A.lut_ok_sw ? modal_aero_sw_diag_kernel<true> : modal_aero_sw_diag_kernel<false>;
        if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, A, ncol, pver, cpb, io, lev, col);
        hipLaunchKernelGGL(modal_aero_sw_diag_tau_kernel, grid, dim3(NT), 0, st, A, ncol, pver, cpb, io, lev, col);
    }
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}
