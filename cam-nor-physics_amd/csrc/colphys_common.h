// colphys_common.h -- shared by the HIP translation units of libcolphys_hip.so (gfx950 only).
//
// Device data layout (DESIGN.md "Data layout in HBM"): one buffer per field, column-contiguous
// [ncol][pver] (interfaces [ncol][pver+1]); a work item is one grid column.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/colphys.h"

#define WAVE 64

// The ZM chain (trigger -> plume -> evaporation -> momentum -> update -> geopotential) is a sequence of
// latency-bound launches with few waves; the aerosol optics on the second stream are throughput work that shares
// the SIMDs with it.  Chain kernels raise their waves' issue priority so that the optics fill the chain's stalls
// instead of time-slicing with it (s_setprio: wave priority at the SIMD's instruction arbiter).
#ifndef COLPHYS_CHAIN_PRIO
#define COLPHYS_CHAIN_PRIO 3
#endif
#define CHAIN_PRIO() __builtin_amdgcn_s_setprio(COLPHYS_CHAIN_PRIO)
// serial level loops of the scan kernels (one lane = one column, profiles in LDS): iterations the compiler may keep in flight
#ifndef SCAN_UNROLL
#define SCAN_UNROLL 2
#endif

// Synchronisation of ONE wave with itself: LDS traffic of a wave executes in order, so lanes of the same wave see each other's
// earlier LDS stores without a hardware barrier; what is needed is that the COMPILER neither moves LDS accesses across this point
// nor keeps a value in a register that another lane has overwritten in LDS.  A wavefront-scope release / acquire fence pair is
// that (and the s_waitcnt lgkmcnt(0) the memory model asks for); wave_barrier keeps the scheduler from reordering around it.
// Unlike __syncthreads() it is legal under divergent control flow (no s_barrier is executed) and costs no barrier slot.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Parameters as the kernels see them: colphys_zm_params plus the values zm_convi derives
// (zm_conv.F90:145-155).  Passed BY VALUE as a kernel argument (wave-uniform => SGPRs / scalar loads).
struct ZmDev {
    // physconst
    double cpair, epsilo, gravit, latice, latvap, tmelt, rair, cpwv, cpliq, rh2o, cpvir, zvir;
    // tunables
    double c0_lnd, c0_ocn, ke, ke_lnd, momcu, momcd, tiedke_add, capelmt, tau;
    double tiedke_lnd, entrmn, alfadet, tentrm, plclmin, cin_threshd, parcel_hscale;
    // derived
    double rl, cpres, rgrav, rgas, grav, cp, tfreez, eps1, dcol, omeps;
    double ln_tfreez;               // ln(tfreez), host libm at init
    int limcnv, num_cin, zm_org, no_deep_pbl, lparcel_pbl;
};

// 1-K saturation table zm_conv_evap's qsat interpolates in (spec: oracle/stubs/l0_arithmetic.F90)
#define COLPHYS_PLENEST 249
#define COLPHYS_TMIN_TBL 127.16
#define COLPHYS_TMAX_TBL 375.16

struct AeroDev {
    int nmodes, nspec_max, nswbands, nlwbands, top_lev;
    double rhoh2o, rga, rair, xrmin, xrmax;
    const int *nspec, *spectype;
    const double *sigmag, *specdens, *hygro, *specrefindex_sw, *specrefindex_lw;
    const double *refrtabsw, *refitabsw, *refrtablw, *refitablw, *extpsw, *abspsw, *asmpsw, *absplw, *crefwsw, *crefwlw;
    // packed per-(mode, band) table slices = the LDS image the optics kernels stage (built by colphys_aero_tables_set):
    // ntab x [PREFI*PREFR corners][AERO_NCP] coefficients (5 + zero pad), refr[PREFR], refi[PREFI], the reciprocal
    // interval widths rdxr[PREFR], rdxi[PREFI] (0 where the interval is degenerate and for the last point), zero fill to 1 KiB
    const double *slices_sw, *slices_lw;
    const double *explnsigma;       // [nmodes] exp(2 ln(sigmag)^2), modal_aer_opt.F90:1553-1554
    const double *rspecdens;        // [nmodes][nspec_max] 1/specdens
    double rxrange;                 // 1/(xrmax - xrmin)
    int nspec_total;                // sum of nspec over the modes (host mirror)
    int idx_sw_diag, idx_uv_diag, idx_nir_diag;   // radconstants' diagnostic bands, 1-based (SW history diagnostics)
    int lut_ok_sw, lut_ok_lw;       // every SW / LW slice has an interval lookup (host mirror)
    const int *lastmask;            // [nmodes] bit l: species l is the LAST of its type (0..5) in the mode -- the one whose
                                    // refractive index / hygroscopicity the species partition keeps (modal_aer_opt.F90:780-829)
};
constexpr int AERO_NCP = 6;
constexpr int AERO_TBLP = COLPHYS_PREFI * COLPHYS_PREFR * AERO_NCP;
// Interval lookup of table_interp's grid search (first grid point above x), per grid of a slice: a bucket index is cut
// from the HIGH WORD of x -- b = clamp((max(hi32(x), 0) - base) >> shift, 0, AERO_LUTB-1), monotone in x, roughly
// logarithmic across binades and linear inside one, which suits both the linear real-index grids and the geometric
// imaginary-index grids of the modal optics files -- chosen on the host so that no bucket holds more than one grid point.
// Then  #{j : g_j <= x} = cnt[b] + (x >= gin[b]),  cnt[b] = grid points in lower buckets, gin[b] = the grid point inside
// bucket b (NaN if none: the compare is then false): one compare instead of one per grid point, and the grid leaves the SGPRs.
// Slice image after the tables and the four grid arrays: gin_r[LUTB], gin_i[LUTB] doubles; cnt_r[LUTB], cnt_i[LUTB]
// bytes; int32 base_r, shift_r, base_i, shift_i, ok (ok = 0: some bucket would hold two points -> the kernels' linear
// search is used for the whole table set).
constexpr int AERO_LUTB = 32;
constexpr int AERO_LUT_DOUBLES = 2 * AERO_LUTB + 2 * AERO_LUTB / 8 + 3;
constexpr int aero_lut_offset(int ntab) { return ntab * AERO_TBLP + 2 * (COLPHYS_PREFR + COLPHYS_PREFI); }
constexpr int aero_slice_doubles(int ntab)
{
    return ((aero_lut_offset(ntab) + AERO_LUT_DOUBLES) * 8 + 1023) / 1024 * 128;
}

// ---- division ---------------------------------------------------------------------------------------------
// a / b as hipcc's own fp64 division computes it -- hardware reciprocal, two Newton steps, quotient, one residual
// correction -- WITHOUT the operand pre-scaling and the special-case fix-up around it (v_div_scale x2, v_div_fmas,
// v_div_fixup): 8 instructions instead of 11 on the ascent's dependent chain, where a third of all instructions are
// division sequences.  Bit-identical to a / b whenever neither operand, the reciprocal nor the quotient leaves the
// normal range (the pre-scaling is the identity there); used only where that holds by construction: b is a
// temperature, a pressure, a mixing ratio, Brent's function values (non-zero where divided by), or 2 + f inside a
// logarithm.  NaN operands give NaN as before; b = 0 or inf gives NaN instead of inf / 0.
__device__ __forceinline__ double fdiv(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

// ---- logarithms -----------------------------------------------------------------------------------------
// The dilute-parcel ascent spends ~45 % of its instructions in log / log10 (three per entropy evaluation inside
// Brent's iteration).  ROCm's OCML log is ~90 VALU instructions; this is the classical argument-reduction form
//     x = 2^k m,  m in [sqrt(1/2), sqrt(2)),  f = m - 1,  s = f/(2+f),  log m = f - (f^2/2 - s (f^2/2 + R(s^2)))
// with the degree-7 minimax R of that scheme (Remez coefficients as published with Sun's fdlibm): < 1 ulp, the same
// accuracy class as OCML and as the host libm the reference links, in ~45 instructions.  Neither is bit-identical
// to the reference's libm; parity is 1e-10 relative either way ("transcendentals").
// Zero, negative, infinite and NaN arguments take OCML's path, so exceptional values propagate as before.
__device__ __forceinline__ void cp_log_core(double x, double &dk, double &lm)
{
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    int k = __builtin_amdgcn_frexp_exp(x);            // x = mant * 2^k, mant in [0.5, 1); denormals included
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = fdiv(f, 2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, L6, L4), L2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    dk = (double)k;
    lm = f - (hfsq - s * (hfsq + R));
}
__device__ __forceinline__ double cp_log(double x)
{
    if (!(x > 0.0 && x < __builtin_inf())) return log(x);
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    int k = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = fdiv(f, 2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, L6, L4), L2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}
__device__ __forceinline__ double cp_log10(double x)
{
    if (!(x > 0.0 && x < __builtin_inf())) return log10(x);
    const double ivln10_hi = 4.34294481903251816668e-01, ivln10_lo = 1.09831965021676507e-17;
    const double log10_2hi = 0.3010299955494702, log10_2lo = 1.1451100898021838e-10;   // hi: 32 significant bits
    double dk, lm;
    cp_log_core(x, dk, lm);
    return dk * log10_2hi + (dk * log10_2lo + __builtin_fma(lm, ivln10_lo, lm * ivln10_hi));
}

// ---- exp10 ------------------------------------------------------------------------------------------------
// ROCm OCML's fp64 exp10, operation for operation (constants read from its gfx950 code: n = rint(x log2(10)),
// r = x - n log10(2) in two pieces, u = r ln(10) in two pieces, degree-11 polynomial, ldexp), WITHOUT its two range
// guards (x > 1024 -> inf, x < -1075 -> 0: two compares and four selects per call).  Bit-identical to exp10() for
// |x| < 1024 -- ldexp alone saturates correctly there -- and NaN in, NaN out; the Goff-Gratch exponents it is used for
// lie in [-12, 9] for any temperature a sounding can hold (T > 100 K).  Larger |x| (T -> 0, garbage input) differ from
// OCML only in HOW the result is wrong (inf/0 vs a saturated ldexp).
__device__ __forceinline__ double cp_exp10(double x)
{
    const double n = __builtin_rint(x * __longlong_as_double(0x400a934f0979a371LL));
    double r = __builtin_fma(__longlong_as_double(0xbfd34413509f79ffLL), n, x);
    r = __builtin_fma(__longlong_as_double(0x3c49dc1da994fd21LL), n, r);
    double u = r * __longlong_as_double(0xbcaf48ad494ea3e9LL);
    u = __builtin_fma(__longlong_as_double(0x40026bb1bbb55516LL), r, u);
    double p = __builtin_fma(u, __longlong_as_double(0x3e5ade156a5dcb37LL), __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f81111111122322LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fc5555555555511LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(u, p, 1.0);
    p = __builtin_fma(u, p, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// exp(x) the same way (OCML's reduction x = n ln2 + u with ln2 in two pieces, the same polynomial in u, ldexp), without
// OCML's overflow / underflow guards: for |x| < 700.  Used by the optics for the specific extinction exp(sum of Chebyshev
// terms), whose argument is a logarithm of m2/kg values (|x| < 40).  < 1 ulp.
__device__ __forceinline__ double cp_exp(double x)
{
    const double n = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82feLL));       // log2(e)
    double u = __builtin_fma(__longlong_as_double(0xbfe62e42fefa39efLL), n, x);            // -ln2 (high part)
    u = __builtin_fma(__longlong_as_double(0xbc7abc9e3b39803fLL), n, u);                   // -ln2 (low part)
    double p = __builtin_fma(u, __longlong_as_double(0x3e5ade156a5dcb37LL), __longlong_as_double(0x3e928af3fca7ab0cLL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3ec71dee623fde64LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3efa01997c89e6b0LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f2a01a014761f6eLL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3f81111111122322LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fa55555555502a1LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fc5555555555511LL));
    p = __builtin_fma(u, p, __longlong_as_double(0x3fe000000000000bLL));
    p = __builtin_fma(u, p, 1.0);
    p = __builtin_fma(u, p, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}

// want_ln ? cp_log(x) : cp_log10(x), bit for bit, from ONE reduction + polynomial: the two lanes of a lane pair
// (below) evaluate a natural and a decimal logarithm of different arguments in the same instruction stream.
__device__ __forceinline__ double cp_log_sel(double x, bool want_ln)
{
    if (!(x > 0.0 && x < __builtin_inf())) return want_ln ? log(x) : log10(x);
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double ivln10_hi = 4.34294481903251816668e-01, ivln10_lo = 1.09831965021676507e-17;
    const double log10_2hi = 0.3010299955494702, log10_2lo = 1.1451100898021838e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01,
                 L4 = 2.222219843214978396e-01, L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                 L7 = 1.479819860511658591e-01;
    int k = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double s = fdiv(f, 2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, L6, L4), L2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double sR = s * (hfsq + R);
    const double ln = dk * ln2_hi - ((hfsq - (sR + dk * ln2_lo)) - f);                    // cp_log's tail
    const double lm = f - (hfsq - sR);                                                     // cp_log_core's
    const double l10 = dk * log10_2hi + (dk * log10_2lo + __builtin_fma(lm, ivln10_lo, lm * ivln10_hi));   // cp_log10's tail
    return want_ln ? ln : l10;
}

// ---- lane pairs -----------------------------------------------------------------------------------------
// The dilute-parcel ascent is one long dependent chain per column (a lone wave retires one dependent fp64
// instruction per ~8.6 cycles, scripts/ubench), and most of it is the saturation-pressure / entropy evaluation inside
// Brent's iteration: 3 exp10 + 3 logarithms of which two exp10 and two logarithms are mutually independent.  With
// LPC = 2 a column owns TWO adjacent lanes that hold bit-identical state and run the same instruction stream; inside an
// evaluation the odd lane ("half") takes the second member of each independent pair and the values are exchanged by DPP
// (quad_perm [1,0,3,2]: no LDS, no exec change).  Every value is produced by the same operations in the same order as
// with one lane per column, so results are bit-identical by construction; the chain per evaluation is ~0.62x as long.
__device__ __forceinline__ double pair_swap(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// LPC = 4, lane quads: a column owns FOUR adjacent lanes = two lane pairs ("sides").  Everywhere a quad behaves as two pairs
// doing the same thing (the DPP swap of pair_swap is per pair); the one place the second pair earns its keep is the opening of
// Brent's iteration, whose two bracket evaluations f(Tfg - 10), f(Tfg + 10) are independent: side 0 evaluates the first, side 1
// the second, at the same time, and quad_swap2 (quad_perm [2,3,0,1]) hands each side the other's value -- one function
// evaluation less on the chain of every inversion (~8.5 -> ~7.5), same bits.  Pays only while the twice-as-many waves still
// find a SIMD each: a rank's share of a strong-scaled grid, a chunk-sized call.
__device__ __forceinline__ double quad_swap2(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ---- saturation vapour pressure: Goff-Gratch over water / ice (spec: oracle/stubs/l0_arithmetic.F90) ----
// u = Tboil/T and v = T/Tboil as the caller formed them (quotients in svp_water_gg, products with a shared
// reciprocal inside Brent's function evaluation).  WANT_LN: also return lnval = cp_log(lnarg) (the entropy's
// log(T/Tfreez), which a lane pair evaluates beside log10(u)).
template <int LPC, bool WANT_LN>
__device__ __forceinline__ double svp_gg_core(double u, double v, bool half, double lnarg, double &lnval)
{
    const double um1 = u - 1.0;
    double l10u, e2, e3;
    if (LPC >= 2) {
        if (WANT_LN) {
            const double lg = cp_log_sel(half ? lnarg : u, half);
            const double lgo = pair_swap(lg);
            l10u = half ? lgo : lg;
            lnval = half ? lg : lgo;
        } else {
            l10u = cp_log10(u);
        }
        const double ee = cp_exp10(half ? -3.49149 * um1 : 11.344 * (1.0 - v));
        const double eo = pair_swap(ee);
        e2 = half ? eo : ee;
        e3 = half ? ee : eo;
    } else {
        l10u = cp_log10(u);
        if (WANT_LN) lnval = cp_log(lnarg);
        e2 = cp_exp10(11.344 * (1.0 - v));
        e3 = cp_exp10(-3.49149 * um1);
    }
    return cp_exp10(-7.90298 * um1 + 5.02808 * l10u - 1.3816e-7 * (e2 - 1.0) + 8.1328e-3 * (e3 - 1.0)
                 + 3.0057148979490314) * 100.0;        // log10(1013.246), correctly rounded
}
__device__ inline double svp_water_gg(double t)
{
    const double tboil = 373.16;
    double unused;
    return svp_gg_core<1, false>(tboil / t, t / tboil, false, 0.0, unused);
}

// ---- launchers implemented in the k_*.hip files ------------------------------------------------------
int launch_geopotential_t(int ncol, int pver, int dycore_kind, int nq, int nspecies, const double *piln,
                          const double *pint, const double *pmid, const double *pdel, const double *rpdel,
                          const double *t, const double *q, const double *rair, double gravit, const double *zvir,
                          double *zi, double *zm, hipStream_t st, int rz_uniform = 0);

int launch_geopotential_dse(int ncol, int pver, int dycore_kind, int nspecies, const double *piln, const double *pint,
                            const double *pmid, const double *pdel, const double *rpdel, const double *dse, const double *q,
                            const double *phis, const double *rair, double gravit, const double *cpair, const double *zvir,
                            double *t, double *zi, double *zm, hipStream_t st);

struct ZmWorkspace {            // device scratch owned by the API layer (grow-only)
    int *count;                 // [1]   number of first-pass triggered columns
    int *list;                  // [ncol] their column indices (order irrelevant)
    double *cape1, *cin1;       // [ncol]
    int *lel1, *mx1;            // [ncol]
    int *deep;                  // [ncol] 1 if the column ends in the gathered set
    int *jt_f, *maxg_f;         // [ncol] full-layout copies
    double *dsub_f;             // [ncol]
    double *mu_f, *md_f, *du_f, *eu_f, *ed_f, *dp_f;  // [ncol][pver] full-layout plume profiles
    // second ascent of the first-pass triggered columns (zm_ascent2_kernel -> zm_plume2_kernel)
    double *dmpdz2, *cape2, *tl2;   // [ncol]
    int *lcl2, *lel2, *mx2, *retrig; // [ncol]
    double *tp2, *qstp2;            // [ncol][pver]
};
// zm_conv_tend's host-side post-processing of zm_convr (zm_conv_intr.F90:685-731), done by the gather kernel when asked:
// mass flux to kg/m2/s, FREQZM, ZMMU/ZMMD, PCONVT/PCONVB (diagnostic pointers may be null)
struct ZmTendDiag {
    int enabled;
    double gravit;
    double *mconzm;
    const double *ps, *pmid;
    double *freqzm, *pcont, *pconb, *mu_out, *md_out;
};
int launch_zm_convr(const ZmDev &P, int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io &io,
                    const ZmWorkspace &ws, hipStream_t st, hipEvent_t *ev /* 8 events or null */,
                    const ZmTendDiag *diag = nullptr);

int launch_zm_conv_evap(const ZmDev &P, const double *estbl_dev, int ncol, int pver, double deltat,
                        const colphys_zm_conv_evap_io &io, hipStream_t st);
// winds_v: v in its own [ncol][pver] array (then io.winds holds u only), or null for the stacked [2][ncol][pver] layout
int launch_momtran(const ZmDev &P, int ncol, int pver, int pcols, double dt, const colphys_momtran_io &io, hipStream_t st,
                   const double *winds_v = nullptr);
// The tail of zm_conv_tend for a SMALL rank (a strong-scaled share, a chunk-sized call: every launch is latency, not work), after
// zm_convr and its first physics_update: geopotential_t (state1 after zm_convr), zm_conv_evap and momtran -- independent of each
// other (zm_conv_intr.F90:739-826: momtran reads winds and mass fluxes the first two never touch) -- as ONE launch of
// heterogeneous blocks; then the second and third physics_update as one element-wise launch, the second and third
// geopotential_t as one batched launch.  Same kernels' bodies, same results bit for bit; 5 launches instead of 9.
struct ZmTailSmall {
    int ncol, pver, pcols, dycore_kind;
    int rz_uniform;                                   // rair / zvir point at ONE value each (stride 0) instead of fields
    double ztodt, qmin;
    // geopotential_t inputs of the state (pressure fields are the caller's, unchanged by the path)
    const double *lnpint, *pint, *pmid, *pdel, *rpdel, *rair, *zvir;
    double *t1, *q1, *u1, *v1, *zi1, *zm1;           // state1 (t1, q1 hold the state after the first update on entry)
    double *t2, *zi_scratch, *zm_scratch;            // work: temperature after the second update, the second geopotential_t's heights
    const double *u, *v;                             // state winds
    colphys_zm_conv_evap_io evap;
    colphys_momtran_io mom;                          // mom.winds = u, winds_v = v
    double *ptend_s, *ptend_q, *ptend_u, *ptend_v;   // ptend_all (s, q hold the zm_convr contribution on entry)
};
int launch_zm_tail_small(const ZmDev &P, const double *estbl_dev, const ZmTailSmall &T, hipStream_t st);
int launch_modal_aero_sw(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st);
int launch_modal_aero_lw(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, hipStream_t st);
void aero_kernel_select(int sw_kind, int lw_kind);
int launch_aero_grid_probe(const AeroDev &A, int sw, int mode, int band, int which, int use_lut, long long n, const double *x,
                           int *ix1, int *ix2, double *t, hipStream_t st);
int launch_modal_aero_sw_diag(const AeroDev &A, int ncol, int pver, const colphys_modal_aero_io &io, double *lev, double *col,
                              hipStream_t st);
int launch_physics_update(int ncol, int pver, double dt, double cpair, double qmin, const double *t_in,
                          const double *s_tend, double *t_out, const double *q_in, const double *q_tend, double *q_out,
                          const double *u_in, const double *u_tend, double *u_out, const double *v_in,
                          const double *v_tend, double *v_out, hipStream_t st, double *sum_s = nullptr,
                          double *sum_q = nullptr, double *sum_u = nullptr, double *sum_v = nullptr, int first = 0);
int launch_math_probe(int kind, int64_t n, const double *x, double *y, hipStream_t st);
int launch_convect_diag(int ncol, int pver, int clubb, const colphys_convect_diag_io &io, hipStream_t st);
int launch_derived_fields(int ncol, int pver, int ncnst, int dycore_kind, int nspecies, double gravit,
                          const colphys_derived_fields_io &io, hipStream_t st);
int launch_dry_mass_fields(int ncol, int pver, int ncnst, const colphys_dry_mass_io &io, hipStream_t st);
// pdeldry: null (io.dpdry is the gathered dry pressure thickness) or state%pdeldry [ncol][pver] in Pa, gathered through ideep on the fly
// ws_rows [ncol], ws_count [1], ws_il2g [nchunks]: device scratch (gathered-row list of the launch)
int launch_convtran(int ncol, int pver, int pcols, int ncnst, const colphys_convtran_io &io, hipStream_t st, const double *pdeldry,
                    int *ws_rows, int *ws_count, int *ws_il2g);
