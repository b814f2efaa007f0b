// k_zm_convr.hip -- Zhang-McFarlane deep convection (zm_convr) for gfx950.
//
// Replaces /root/reference/physics/zm_conv.F90:231-1709 and its callees buoyan_dilute :4425, parcel_dilute
// :4824, entropy/enthalpy/ientropy/ienthalpy :5280-5570, cldprp :3024, closure :4028, q1q2_pjr :4262.
// Line numbers in comments are zm_conv.F90.
//
// Whole-node dispatch: one work item = one grid column, independent of how the host groups columns into
// chunks.  Five launches per call:
//
//   zm_trigger_kernel  every column : unit conversion, PBL-top index, first dilute-parcel ascent
//                                     (buoyan_dilute/parcel_dilute), CAPE/CIN, trigger test; zero-fills the
//                                     outputs; appends triggered columns to a device-side list.
//   zm_plume1_kernel   listed columns, one wave each: cloud ensemble (cldprp) -> the ensemble's mean entrainment
//   zm_ascent2_kernel  listed columns: second parcel ascent with that entrainment, re-trigger test
//   zm_plume2_kernel   re-triggered columns, one wave each: cldprp again, closure, tendencies (q1q2_pjr),
//                                     precipitation integrals; writes full-layout results.
//   zm_gather_kernel   one block per host chunk: ascending-order compaction of the deep columns into the
//                                     reference's per-chunk gathered layout (mu md du eu ed dp dsubcld jt maxg
//                                     ideep lengath), zm_conv.F90:905-940,1096-1174.
//
// The dilute parcel is the cost (>= 84 % of the CPU time).  The reference makes two sweeps over
// the levels (entrainment loop :4997-5148, then precipitation/freezing loop :5175-5273) through seven work
// profiles; here the two sweeps run as a two-wave pipeline over the same 64 columns (buoyan_dilute_2w below), each
// wave carrying the (k+1) values of its sweep in registers -- same arithmetic in the same order per level, no work
// profiles.  The only profile the trigger needs afterwards is the parcel virtual temperature (for the
// neutral-buoyancy search and the CAPE/CIN sums, :4773-4810); it is kept in LDS as [level][lane].
//
// Arithmetic: fp64, no FMA contraction (-ffp-contract=off), operation order of the reference.
// Transcendentals are ROCm OCML (exp10/exp/pow) and the < 1 ulp logarithms of colphys_common.h: not
// bit-identical to the host libm the reference Fortran links, which is why parity is stated as 1e-10
// relative, not bitwise.
#include "colphys_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// L0 seams (spec: oracle/stubs/l0_arithmetic.F90)
// ------------------------------------------------------------------------------------------------------
// LPC / half: lane pairs (colphys_common.h); LPC = 1 is one lane per column, half is then false.
template <int LPC = 1>
__device__ __forceinline__ void qsat_from_es(const ZmDev &P, double es, double p_hPa, double &es_hPa, double &qs)
{   // qsat_hPa :5421-5437 around qsat_water(t, p[Pa]), given es = svp_water(t) [Pa]
    const double p = p_hPa * 100.0;
    if ((p - es) <= 0.0) qs = 1.0;
    else qs = P.epsilo * es / (p - P.omeps * es);
    es = fmin(es, p);
    es_hPa = es * 0.01;
}
template <int LPC = 1>
__device__ __forceinline__ void qsat_hPa(const ZmDev &P, double t, double p_hPa, double &es_hPa, double &qs, bool half = false)
{
    double unused;
    const double es = svp_gg_core<LPC, false>(373.16 / t, t / 373.16, half, 0.0, unused);
    qsat_from_es(P, es, p_hPa, es_hPa, qs);
}

template <int LPC = 1>
__device__ __forceinline__ double entropy_d(const ZmDev &P, double TK, double p, double qtot, bool half = false)
{   // :5280-5300
    const double pref = 1000.0;
    double L = P.rl - (P.cpliq - P.cpwv) * (TK - P.tfreez);
    double est, qst, lnT;
    const double es = svp_gg_core<LPC, true>(373.16 / TK, TK / 373.16, half, TK / P.tfreez, lnT);
    qsat_from_es(P, es, p, est, qst);
    double qv = fmin(qtot, qst);
    double e = qv * p / (P.eps1 + qv);
    double lgp, lgq;
    if (LPC >= 2) {                         // the pair splits log((p - e)/pref) and log(qv/qst)
        const double lg = cp_log((half ? qv : (p - e)) / (half ? qst : pref));
        const double lgo = pair_swap(lg);
        lgp = half ? lgo : lg;
        lgq = half ? lg : lgo;
    } else {
        lgp = cp_log((p - e) / pref);
        lgq = cp_log(qv / qst);
    }
    return (P.cpres + qtot * P.cpliq) * lnT - P.rgas * lgp + L * qv / TK - qv * P.rh2o * lgq;
}

template <int LPC = 1>
__device__ __forceinline__ double enthalpy_d(const ZmDev &P, double TK, double p, double qtot, double z, bool half = false)
{   // :5440-5457
    double L = P.rl - (P.cpliq - P.cpwv) * (TK - P.tfreez);
    double est, qst;
    qsat_hPa<LPC>(P, TK, p, est, qst, half);
    double qv = fmin(qtot, qst);
    return (P.cpres + qtot * P.cpliq) * TK + L * qv + (1.0 + qtot) * P.grav * z;
}

// ientropy :5304-5414 (KIND 0) / ienthalpy :5460-5570 (KIND 1): Brent's method, bracket Tfg +- 10 K.
//
// Same iteration as the reference.  The function evaluation is where the ascent's instructions go (~9 evaluations
// per inversion, 3-4 inversions per level), so it is written for instruction count:
//  (1) qst(T) is carried with every bracket point, so the reference's closing `qsat_hPa(T)` (:5399, :5555) --
//      a re-evaluation at a temperature that was already evaluated -- is a register move (identical bits);
//  (2) entropy: while the parcel is unsaturated at the trial temperature (qv = qt) the term
//      log((p - e)/pref) does not depend on T: it is computed once per inversion and reused (identical bits);
//  (3) entropy: when saturated, qv = qst and log(qv/qst) = log(1) = 0 exactly: the logarithm is skipped;
//  (4) ONE reciprocal of the trial temperature serves the whole evaluation, and divisions by constants are
//      products: Tboil/T = Tboil*(1/T), L qv/T = L qv (1/T), T/Tboil = T*(1/Tboil), T/Tfreez = T*(1/Tfreez),
//      x/1000 = x*0.001.  Each differs from the reference's quotient by at most one unit in the last place of
//      that intermediate (|error| of the entropy ~1e-13 J/kg/K) -- the same class of difference as OCML vs the
//      host libm.  (Sharing one ln T between log(T/Tfreez) and log10(Tboil/T) was tried and rejected: it costs
//      ~5e-13 in the entropy, enough to flip Brent's sign tests near the root on ~1 column in 1000 and move CAPE
//      by 1.5e-10.)
#ifndef BRENT_INLINE
#define BRENT_INLINE __forceinline__
#endif
//  (5) LPC = 2: the function evaluation runs on a lane pair (colphys_common.h): same values, shorter chain.
//  (6) the quotients of the evaluation and of Brent's interpolation step use fdiv (colphys_common.h): the compiler's
//      own division sequence without its range pre-scaling, bit-identical for these operands, 3 instructions shorter.
//  (7) LPC = 4: the two bracket evaluations that open the iteration run side by side on the two pairs of a lane quad
//      (`side`; colphys_common.h quad_swap2): one evaluation less on the chain, same bits.
template <int KIND, int LPC = 1>
__device__ BRENT_INLINE int invert_brent(const ZmDev &P, double s, double p, double z, double qt, double &T, double &qst,
                                         double Tfg, bool half = false, bool side = false)
{
    const int LOOPMAX = 100;
    const double EPS = 3.e-8;
    double a, b, c, d = 0.0, ebr = 0.0, fa, fb, fc, pbr, qbr, rbr, sbr, tol1, xm;
    double qa, qb, qc;                         // qst at a, b, c
    const double tol = 0.001;
    bool converged = false;

    // T-independent pieces of entropy(T, p, qt) for the unsaturated branch (:5295-5298)
    double lg_unsat = 0.0;
    if (KIND == 0) {
        const double e_unsat = qt * p / (P.eps1 + qt);
        lg_unsat = cp_log((p - e_unsat) / 1000.0);
    }
    const double pPa = p * 100.0;              // qsat_hPa :5430
    const double RTBOIL = 1.0 / 373.16, rtfreez = 1.0 / P.tfreez;

    auto feval = [&](double TK, double &qs_out) -> double {
        const double rt = fdiv(1.0, TK);
        // Goff-Gratch over water (svp_gg_core, colphys_common.h) on the shared 1/T; the entropy's log(T/Tfreez) rides
        // beside its log10 on the second lane of a pair
        double lnT = 0.0;
        const double es = svp_gg_core<LPC, KIND == 0>(373.16 * rt, TK * RTBOIL, half, TK * rtfreez, lnT);
        double qs;                             // qsat_water: qs = eps es / (p - (1-eps) es), 1 where p <= es
        if ((pPa - es) <= 0.0) qs = 1.0;
        else qs = fdiv(P.epsilo * es, pPa - P.omeps * es);
        qs_out = qs;
        const double L = P.rl - (P.cpliq - P.cpwv) * (TK - P.tfreez);
        const double qv = fmin(qt, qs);
        if (KIND == 1) return ((P.cpres + qt * P.cpliq) * TK + L * qv + (1.0 + qt) * P.grav * z) - s;       // :5454
        // unsaturated (or exactly saturated, qv = qt): e depends on qt only, lgp = lg_unsat, lgq = log(qv/qs);
        // saturated (qv = qs): lgp = log((p - e)/pref) with e = qv p/(eps1 + qv), lgq = log(qs/qs) = 0.
        // One quotient and one logarithm serve both cases (lanes of a wave differ near the condensation level).
        const bool unsat = (qv == qt);
        const double quo = fdiv(unsat ? qv : qv * p, unsat ? qs : P.eps1 + qv);
        const double lg = cp_log(unsat ? quo : (p - quo) * 0.001);
        const double lgp = unsat ? lg_unsat : lg, lgq = unsat ? lg : 0.0;
        return ((P.cpres + qt * P.cpliq) * lnT - P.rgas * lgp + L * qv * rt - qv * P.rh2o * lgq) - s;   // :5297
    };

    a = Tfg - 10;
    b = Tfg + 10;
    if (LPC == 4) {
        double q0;
        const double f0 = feval(side ? b : a, q0);
        const double fo = quad_swap2(f0), qo = quad_swap2(q0);
        fa = side ? fo : f0; qa = side ? qo : q0;
        fb = side ? f0 : fo; qb = side ? q0 : qo;
    } else {
        fa = feval(a, qa);
        fb = feval(b, qb);
    }
    c = b;
    fc = fb;
    qc = qb;

    for (int i = 0; i <= LOOPMAX; ++i) {
        if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) {
            c = a; fc = fa; qc = qa; d = b - a; ebr = d;
        }
        if (fabs(fc) < fabs(fb)) {
            a = b; b = c; c = a; fa = fb; fb = fc; fc = fa;
            const double qt_ = qb; qa = qb; qb = qc; qc = qt_;
        }
        tol1 = 2.0 * EPS * fabs(b) + 0.5 * tol;
        xm = 0.5 * (c - b);
        converged = (fabs(xm) <= tol1 || fb == 0.0);
        if (converged) break;

        if (fabs(ebr) >= tol1 && fabs(fa) > fabs(fb)) {
            sbr = fdiv(fb, fa);
            if (a == c) {
                pbr = 2.0 * xm * sbr;
                qbr = 1.0 - sbr;
            } else {
                qbr = fdiv(fa, fc);
                rbr = fdiv(fb, fc);
                pbr = sbr * (2.0 * xm * qbr * (qbr - rbr) - (b - a) * (rbr - 1.0));
                qbr = (qbr - 1.0) * (rbr - 1.0) * (sbr - 1.0);
            }
            if (pbr > 0.0) qbr = -qbr;
            pbr = fabs(pbr);
            if (2.0 * pbr < fmin(3.0 * xm * qbr - fabs(tol1 * qbr), fabs(ebr * qbr))) {
                ebr = d; d = fdiv(pbr, qbr);
            } else {
                d = xm; ebr = d;
            }
        } else {
            d = xm; ebr = d;
        }
        a = b;
        fa = fb;
        qa = qb;
        b = b + ((fabs(d) > tol1) ? d : copysign(tol1, xm));
        fb = feval(b, qb);
    }
    T = b;
    qst = qb;                                  // = qsat_hPa(T, p) of :5399 / :5555, already evaluated at b
    return converged ? 0 : (COLPHYS_STATUS_BRENT_NOT_CONVERGED | (KIND == 1 ? COLPHYS_STATUS_BRENT_ENTHALPY : 0));
}

// ------------------------------------------------------------------------------------------------------
// One column's inputs, 1-based level index.  p, z, pf, zf are the unit-converted fields of :825-837.
// ------------------------------------------------------------------------------------------------------
struct ColIn {
    const double *t, *q, *pap, *paph, *zm, *zi, *dpp;
    double zs, tpert, landfrac, pblh;
    int pver, msg;
    __device__ __forceinline__ double T(int k) const { return t[k]; }
    __device__ __forceinline__ double Q(int k) const { return q[k]; }
    __device__ __forceinline__ double Pm(int k) const { return pap[k] * 0.01; }
    __device__ __forceinline__ double Pf(int k) const { return paph[k] * 0.01; }
    __device__ __forceinline__ double Z(int k) const { return zm[k] + zs; }
    __device__ __forceinline__ double Zf(int k) const { return zi[k] + zs; }
};

struct BuoyRes {
    double cape, cin, tl;
    int lcl, lel, mx, lon, status;
};

__device__ __forceinline__ double moist_static_energy_total(const ZmDev &P, double t, double q, double z)
{   // :4681-4682
    return (P.cp + q * P.cpliq) * t / (1.0 + q) + (1.0 + q / P.eps1) / (1.0 + q) * P.grav * z
           + (P.rl - (P.cpliq - P.cpwv) * (t - P.tfreez)) * q;
}

// buoyan_dilute :4425-4819 with parcel_dilute :4824-5277, the two sweeps of the reference run as a TWO-WAVE PIPELINE.
//
// The ascent of a column is a chain that is strictly serial in the level index, and at ne30 size there are fewer
// columns (48 602) than lanes on the chip, so a column's own latency is what the trigger costs.  The reference's two
// sweeps over the levels -- sweep 1 (:4997-5148: entrainment, enthalpy inversion, LCL) and sweep 2 (:5175-5273: rain-out
// and freezing, two entropy inversions per level) -- are coupled only one way: sweep 2 at level k needs sweep 1's
// (tmix, qsmix, qtmix) of level k.  A block is therefore two waves over the same 64 columns (lane = column):
//     wave A (role 0): sweep 1 of level k     + the entropy of the mixture that sweep 2 starts from     (1/3 of the work)
//     wave B (role 1): sweep 2 of level k + 1 + parcel virtual temperature -> LDS profile               (2/3 of the work)
// exchanging four doubles per (column, level) through a double-buffered LDS mailbox, one barrier per level.  Each wave
// carries the (k+1) state of its own sweep in registers (no work profiles).  Same arithmetic per level as the fused
// single-wave pass; the level loop is uniform over the block (lanes below their launch level are predicated off).
// After the last level wave A hands its scalars (LCL index and pressure, parcel temperature at the LCL, status) to
// wave B, which does buoyancy, neutral levels and the CAPE / CIN sums (:4742-4816).  Results are valid on wave B.
//   dmpdz    : parcel entrainment rate (/m); level-uniform in both calls of the reference (:544, :1074)
//   s_tpv    : LDS, element for level k at s_tpv[k*WAVE]  (caller passes base + lane)
//   s_mail   : LDS, MAIL_DOUBLES doubles for the block
//   tp, qstp : optional 1-based outputs (parcel temperature / saturation mixing ratio), written where store_ok
#ifndef CHAIN_WAVES_PER_SIMD
#define CHAIN_WAVES_PER_SIMD 3
#endif
// Issue priority inside a wave pair.  The two waves of a pair share a SIMD (trigger) and VALU issue is arbitrated by
// priority, then age (MI355X_MICROARCH.md, "Two waves per SIMD"): wave B carries 2/3 of the work and is the pair's
// critical path, wave A (the OLDER wave of the block, which would win every tie) only has to stay one level ahead.
// s_setprio ignores EXEC, so the role test must be wave-uniform by construction (readfirstlane).
#ifndef PIPE_PRIO_A
#define PIPE_PRIO_A 1
#endif
#ifndef PIPE_PRIO_B
#define PIPE_PRIO_B 3
#endif
#define PIPE_PRIO(role)                                                                   \
    do {                                                                                  \
        if (__builtin_amdgcn_readfirstlane(role) == 0) __builtin_amdgcn_s_setprio(PIPE_PRIO_A); \
        else __builtin_amdgcn_s_setprio(PIPE_PRIO_B);                                     \
    } while (0)
constexpr int MAIL_FIELDS = 4;
constexpr int MAIL_DOUBLES = (2 * MAIL_FIELDS + 4) * WAVE;

template <bool STORE_TP, bool PIPE, int LPC>
__device__ void buoyan_dilute_2w(const ZmDev &P, const ColIn &c, double pblt, double dmpdz, double cin_in, int role, int lane,
                                 double *s_tpv, double *s_mail, double *tp, double *qstp, bool store_ok, BuoyRes &r)
{
    const int pver = c.pver, msg = c.msg;
    const int npblt = (int)lrint(pblt);
    const bool half = (LPC >= 2) && (lane & 1);          // second lane of a pair (colphys_common.h)
    const bool side = (LPC == 4) && (lane & 2);          // second pair of this column's lane quad
    const double lwmax = 1.e-3, tscool = 0.0;
    double hmax = 0.0, pl, ql;
    int status = 0;

    r.lon = min(pver, npblt + 2);                                   // :4578
    r.lel = pver;
    r.mx = r.lon;
    r.cape = 0.0;
    r.cin = cin_in;

    // launch level and launch parcel: both waves need them and both compute them (a few hundred instructions)
    double parcel_ztop = 0.0, hpar = 0.0, qpar = 0.0;
    if (P.lparcel_pbl) {                                            // :4586-4588, 4638-4673
        double pbl_dz = c.Z(npblt) - c.zs;
        double parcel_dz = fmax(c.zi[pver], P.parcel_hscale * pbl_dz);
        parcel_ztop = parcel_dz + c.zs;
        double parcel_hdp = 0.0, parcel_dp = 0.0, parcel_qdp = 0.0;
        int ipar = pver;
        for (int k = pver; k >= msg + 1; --k) {
            if (c.zi[k + 1] <= parcel_dz) {
                double hmn_lev = moist_static_energy_total(P, c.T(k), c.Q(k), c.Z(k));
                double dp_lev = c.Pf(k + 1) - c.Pf(k);
                double hmn_zdp = hmn_lev * dp_lev;
                double q_zdp = c.Q(k) * dp_lev;
                double dp_zfrac;
                ipar = k;
                if (k == pver) dp_zfrac = 1.0;
                else dp_zfrac = fmin(1.0, (parcel_dz - c.zi[k + 1]) / (c.zi[k] - c.zi[k + 1]));
                parcel_hdp = parcel_hdp + hmn_zdp * dp_zfrac;
                parcel_qdp = parcel_qdp + q_zdp * dp_zfrac;
                parcel_dp = parcel_dp + dp_lev * dp_zfrac;
            }
        }
        hpar = parcel_hdp / parcel_dp;
        qpar = parcel_qdp / parcel_dp;
        r.mx = ipar;
    } else {                                                        // :4677-4689
        for (int k = r.lon; k >= max(npblt, msg + 1); --k) {
            double hmn = moist_static_energy_total(P, c.T(k), c.Q(k), c.Z(k));
            if (hmn > hmax) { hmax = hmn; r.mx = k; }
        }
    }
    const int klaunch = r.mx;
    r.lcl = klaunch;                                                // :4703-4717
    if (P.lparcel_pbl) {
        r.tl = (hpar - P.rl * qpar - P.grav * parcel_ztop) / P.cp;
        ql = qpar;
    } else {
        r.tl = c.T(klaunch);
        ql = c.Q(klaunch);
    }
    pl = c.Pm(klaunch);

    // ---- sweep 1 (:4997-5148): state and one level ------------------------------------------------------------
    double zl = 0.0, mp = 0.0, qtp = 0.0, sp = 0.0, sp0 = 0.0, qtp0 = 0.0, mp0 = 1.0;
    double t_kp1 = 0.0, q_kp1 = 0.0, p_kp1 = 0.0, z_kp1 = 0.0;                       // environment at k+1
    double tmix1_kp1 = 0.0, qsmix1_kp1 = 0.0, qtmix_kp1 = 0.0, smix1_kp1 = 0.0;       // sweep-1 values at k+1
    double t_nx = 0.0, q_nx = 0.0, pap_nx = 0.0, zm_nx = 0.0;                         // inputs of the next level
    auto initA = [&]() {
        const int k = klaunch;                                      // :5002-5038
        const double tk = c.T(k), qk = c.Q(k), pk = c.Pm(k), zk = c.Z(k);
        double est;
        if (P.lparcel_pbl) {
            qtp0 = ql;
            sp0 = enthalpy_d<LPC>(P, r.tl, pl, qtp0, zl, half);
        } else {
            qtp0 = qk;
            sp0 = enthalpy_d<LPC>(P, tk, pk, qtp0, zk, half);
        }
        mp0 = 1.0;
        smix1_kp1 = sp0;
        qtmix_kp1 = qtp0;
        tmix1_kp1 = tk;
        qsat_hPa<LPC>(P, tmix1_kp1, pk, est, qsmix1_kp1, half);
        t_kp1 = tk; q_kp1 = qk; p_kp1 = pk; z_kp1 = zk;
        // the four input values of the NEXT level are requested one level ahead: their (strided, usually L2-missing)
        // loads complete under the Brent solves
        const int kn = klaunch - 1;
        if (kn >= msg + 1) { t_nx = c.t[kn]; q_nx = c.q[kn]; pap_nx = c.pap[kn]; zm_nx = c.zm[kn]; }
    };
    auto levelA = [&](int k, double &pk, double &tmix, double &qsmix, double &qtmix, double &smix2) {
        const double tk = t_nx, qk = q_nx, zk = zm_nx + c.zs;
        pk = pap_nx * 0.01;
        if (k - 1 >= msg + 1) { t_nx = c.t[k - 1]; q_nx = c.q[k - 1]; pap_nx = c.pap[k - 1]; zm_nx = c.zm[k - 1]; }
        const double dp = (pk - p_kp1);                             // :5042-5106
        const double qtenv = 0.5 * (qk + q_kp1);
        const double tenv = 0.5 * (tk + t_kp1);
        const double penv = 0.5 * (pk + p_kp1);
        const double zenv = 0.5 * (zk + z_kp1);
        const double senv = enthalpy_d<LPC>(P, tenv, penv, qtenv, zenv, half);
        const double dpdz = -(penv * P.grav) / (P.rgas * tenv);
        const double dzdp = 1.0 / dpdz;
        const double dmpdp = dmpdz * dzdp;
        sp = sp - dmpdp * dp * senv;
        qtp = qtp - dmpdp * dp * qtenv;
        mp = mp - dmpdp * dp;
        const double smix1 = (sp0 + sp) / (mp0 + mp);
        qtmix = (qtp0 + qtp) / (mp0 + mp);
        status |= invert_brent<1, LPC>(P, smix1, pk, zk, qtmix, tmix, qsmix, tmix1_kp1, half, side);

        if (qsmix <= qtmix && qsmix1_kp1 > qtmix_kp1) {             // LCL :5111-5133
            r.lcl = k;
            const double qxsk = qtmix - qsmix;
            const double qxskp1 = qtmix_kp1 - qsmix1_kp1;
            const double dqxsdp = (qxsk - qxskp1) / dp;
            pl = p_kp1 - qxskp1 / dqxsdp;
            zl = z_kp1 - qxskp1 / dqxsdp * dzdp;
            const double dsdp = (smix1 - smix1_kp1) / dp;
            const double dqtdp = (qtmix - qtmix_kp1) / dp;
            const double slcl = smix1_kp1 + dsdp * (pl - p_kp1);
            const double qtlcl = qtmix_kp1 + dqtdp * (pl - p_kp1);
            double qslcl;
            status |= invert_brent<1, LPC>(P, slcl, pl, zl, qtlcl, r.tl, qslcl, tmix, half, side);
        }
        tmix1_kp1 = tmix; qsmix1_kp1 = qsmix; qtmix_kp1 = qtmix; smix1_kp1 = smix1;
        t_kp1 = tk; q_kp1 = qk; p_kp1 = pk; z_kp1 = zk;
        smix2 = entropy_d<LPC>(P, tmix, pk, qtmix, half);                      // :5204, what sweep 2 starts from
    };

    // ---- sweep 2 (:5175-5273): state and one level ------------------------------------------------------------
    double xsh2o_kp1 = 0.0, ds_xsh2o_kp1 = 0.0, ds_freeze_kp1 = 0.0, qsmix2_kp1 = 0.0;   // sweep-2 values at k+1
    auto initB = [&]() {
        const int k = klaunch;                                      // :5180-5200
        const double tk = c.T(k), qk = c.Q(k), pk = c.Pm(k);
        double est;
        qsat_hPa<LPC>(P, tk, pk, est, qsmix2_kp1, half);
        const double tpk = tk, qstpk = qk;
        s_tpv[k * WAVE] = (tpk + c.tpert) * (1.0 + qstpk / P.eps1) / (1.0 + qstpk);   // :5194
        if (STORE_TP && store_ok) { tp[k] = tpk; qstp[k] = qstpk; }
    };
    auto levelB = [&](int k, double pk, double tmix, double qsmix, double qtmix, double smix2) {
        double xsh2o = 0.0, ds_xsh2o = 0.0, ds_freeze = 0.0, new_s = 0.0, new_q = 0.0;
#pragma unroll 1
        for (int ii = 0; ii < 2; ++ii) {                            // :5202-5267
            xsh2o = fmax(0.0, qtmix - qsmix - lwmax);
            ds_xsh2o = ds_xsh2o_kp1 - P.cpliq * cp_log(tmix / P.tfreez) * fmax(0.0, (xsh2o - xsh2o_kp1));
            if (tmix <= P.tfreez + tscool && ds_freeze_kp1 == 0.0)
                ds_freeze = (P.latice / tmix) * fmax(0.0, qtmix - qsmix - xsh2o);
            if (tmix <= P.tfreez + tscool && ds_freeze_kp1 != 0.0)
                ds_freeze = ds_freeze_kp1 + (P.latice / tmix) * fmax(0.0, (qsmix2_kp1 - qsmix));
            new_s = smix2 + ds_xsh2o + ds_freeze;
            new_q = qtmix - xsh2o;
            const double tfguess = tmix;
            status |= invert_brent<0, LPC>(P, new_s, pk, 0.0, new_q, tmix, qsmix, tfguess, half, side);
        }
        xsh2o_kp1 = xsh2o; ds_xsh2o_kp1 = ds_xsh2o; ds_freeze_kp1 = ds_freeze; qsmix2_kp1 = qsmix;

        const double tpk = tmix;
        const double qstpk = (new_q > qsmix) ? qsmix : new_q;       // :5249-5253
        s_tpv[k * WAVE] = (tpk + c.tpert) * (1.0 + qstpk / P.eps1) / (1.0 + new_q);     // :5263
        if (STORE_TP && store_ok) { tp[k] = tpk; qstp[k] = qstpk; }
    };

    if (!PIPE) {
        // one wave does both sweeps of a level back to back (the trigger at ne30 size: 760 blocks already cover three
        // quarters of the SIMDs, a second wave per block would share SIMDs with other blocks' waves and gain nothing)
        initA();
        initB();
        for (int k = klaunch - 1; k >= msg + 1; --k) {
            double pk, tmix, qsmix, qtmix, smix2;
            levelA(k, pk, tmix, qsmix, qtmix, smix2);
            levelB(k, pk, tmix, qsmix, qtmix, smix2);
        }
    } else {
        // uniform level schedule: at step i wave A works on level pver-1-i, wave B on the level A mailed at step i-1;
        // levels at or below a lane's launch level are skipped by that lane
        double *s_fin = s_mail + 2 * MAIL_FIELDS * WAVE + lane;     // [4][WAVE]: lcl, pl, tl, status of wave A
        const int nsteps = (pver - 1) - (msg + 1) + 2;
        if (role == 0) {
            initA();
            for (int i = 0; i < nsteps; ++i) {
                const int k = pver - 1 - i;
                if (k >= msg + 1 && k < klaunch) {
                    double pk, tmix, qsmix, qtmix, smix2;
                    levelA(k, pk, tmix, qsmix, qtmix, smix2);
                    double *m = s_mail + (i & 1) * MAIL_FIELDS * WAVE + lane;
                    m[0] = tmix; m[WAVE] = qsmix; m[2 * WAVE] = qtmix; m[3 * WAVE] = smix2;
                }
                __syncthreads();
            }
            s_fin[0] = (double)r.lcl; s_fin[WAVE] = pl; s_fin[2 * WAVE] = r.tl; s_fin[3 * WAVE] = (double)status;
            __syncthreads();
            r.status = status;
            return;
        }
        initB();
        double papb_nx = 0.0;
        if (klaunch - 1 >= msg + 1) papb_nx = c.pap[klaunch - 1];
        for (int i = 0; i < nsteps; ++i) {
            const int k = pver - i;
            if (i >= 1 && k >= msg + 1 && k < klaunch) {
                const double pk = papb_nx * 0.01;
                if (k - 1 >= msg + 1) papb_nx = c.pap[k - 1];
                const double *m = s_mail + ((i - 1) & 1) * MAIL_FIELDS * WAVE + lane;
                levelB(k, pk, m[0], m[WAVE], m[2 * WAVE], m[3 * WAVE]);
            }
            __syncthreads();
        }
        __syncthreads();                                            // wave A's scalars are in s_fin
        r.lcl = (int)s_fin[0]; pl = s_fin[WAVE]; r.tl = s_fin[2 * WAVE];
        status |= (int)s_fin[3 * WAVE];
    }
    if (STORE_TP && store_ok) {
        for (int k = pver; k > klaunch; --k) { tp[k] = c.T(k); qstp[k] = c.Q(k); }     // :4597-4598
        for (int k = msg; k >= 1; --k) { tp[k] = c.T(k); qstp[k] = c.Q(k); }
    }

    // ---- buoyancy, neutral levels, CAPE / CIN (:4742-4816) -----------------------------------------
    const bool plge600 = (pl >= P.plclmin);
    if (STORE_TP && store_ok && !plge600) {
        for (int k = pver; k >= msg + 1; --k) { tp[k] = c.T(k); qstp[k] = c.Q(k); }     // :4760-4761
    }
    // buoy(k) overwrites tpv(k) in LDS; zero outside [msg+1, mx]
    for (int k = pver; k >= msg + 1; --k) {
        double b = 0.0;
        if (k <= klaunch && plge600) {
            const double tk = c.T(k), qk = c.Q(k);
            const double tv = tk * (1.0 + qk / P.eps1) / (1.0 + qk);
            b = s_tpv[k * WAVE] - tv + P.tiedke_add;
        }
        s_tpv[k * WAVE] = b;
    }
    int lelten[6];
    double capeten[6], cinten[6];
#pragma unroll
    for (int n = 1; n <= 5; ++n) { lelten[n] = pver; capeten[n] = 0.0; cinten[n] = 0.0; }
    int knt = 0;
    if (plge600) {
        for (int k = msg + 2; k <= pver; ++k) {                     // :4773-4782
            if (k < r.lcl) {
                const double bk = s_tpv[k * WAVE];
                const double bkp1 = (k + 1 <= pver) ? s_tpv[(k + 1) * WAVE] : 0.0;
                if (bkp1 > 0.0 && bk <= 0.0) {
                    knt = min(P.num_cin, knt + 1);
                    lelten[knt] = k;
                }
            }
        }
        for (int n = 1; n <= P.num_cin; ++n) {                      // :4785-4796
            double ct = 0.0, ci = 0.0;
            for (int k = max(msg + 1, lelten[n] + 1); k <= klaunch; ++k) {
                const double bk = s_tpv[k * WAVE];
                const double lg = cp_log(c.Pf(k + 1) / c.Pf(k));
                ct = ct + P.rgas * bk * lg;
                ci = ci - P.rgas * fmin(bk, 0.0) * lg;
            }
            capeten[n] = ct;
            cinten[n] = ci;
        }
    }
    for (int n = 1; n <= P.num_cin; ++n) {                          // :4802-4810
        if (capeten[n] > r.cape) {
            r.cape = capeten[n];
            r.cin = cinten[n];
            r.lel = lelten[n];
        }
    }
    r.cape = fmax(r.cape, 0.0);
    r.status = status;
}

__device__ __forceinline__ double pbl_top_index(const ColIn &c)
{   // :777, 839-843
    double pblt = c.pver;
    for (int k = c.pver - 1; k >= c.msg + 1; --k)
        if (fabs(c.Z(k) - c.zs - c.pblh) < (c.Zf(k) - c.Zf(k + 1)) * 0.5) pblt = k;
    return pblt;
}

__device__ __forceinline__ ColIn make_colin(const ZmDev &P, const colphys_zm_convr_io &io, int col, int pver)
{
    ColIn c;
    const size_t L = (size_t)pver, LP = (size_t)pver + 1;
    c.t = io.t + col * L - 1;      c.q = io.qh + col * L - 1;   c.pap = io.pap + col * L - 1;
    c.zm = io.zm + col * L - 1;    c.dpp = io.dpp + col * L - 1;
    c.paph = io.paph + col * LP - 1; c.zi = io.zi + col * LP - 1;
    c.zs = io.geos[col] * P.rgrav;  // :826
    c.tpert = io.tpert[col]; c.landfrac = io.landfrac[col]; c.pblh = io.pblh[col];
    c.pver = pver; c.msg = P.limcnv - 1;
    return c;
}

// coalesced fill of n doubles starting at p by the whole block
__device__ __forceinline__ void block_fill(double *p, size_t n, double v)
{
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------------
// Kernel 1: trigger
// ------------------------------------------------------------------------------------------------------
// TRIG_GROUPS column groups of 64 per block, each served by a wave pair (sweep 1 / sweep 2, buoyan_dilute_2w).  The
// waves are ordered [A0 .. A(G-1), B0 .. B(G-1)] with G = 4: a workgroup's waves are placed on the CU's four SIMDs in a
// fixed cyclic order, so waves i and i+4 -- the two waves of one column group -- share a SIMD.  That is the point: at
// ne30 size there are only 760 column groups for 1 024 SIMDs, a second wave per group on ANOTHER SIMD would just
// collide with other groups' waves, while on the SAME SIMD the pair fills each other's dependent-issue bubbles (a single
// ascent wave keeps its SIMD 62 % busy) and the per-level latency drops to the longer sweep's.
// (G < 4 when the per-group LDS does not allow four groups, e.g. beyond ~L75: the pairs then sit on different SIMDs.)
#ifndef TRIG_GROUPS_MAX
#define TRIG_GROUPS_MAX 4
#endif
// LPC = 2: a column owns a lane pair (colphys_common.h), a wave serves 32 columns: twice the waves, each ~0.6x as long.
template <int LPC>
__global__ __launch_bounds__(2 * TRIG_GROUPS_MAX * WAVE) void zm_trigger_kernel(ZmDev P, int ncol, int pver, int TRIG_GROUPS,
                                                                               colphys_zm_convr_io io, ZmWorkspace ws)
{
    constexpr int CPW = WAVE / LPC;             // columns per wave
    extern __shared__ double smem[];            // per group: [(pver+2)][WAVE] parcel virtual temperature / buoyancy + mailbox
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (WAVE - 1);
    const int role = wave / TRIG_GROUPS, group = wave % TRIG_GROUPS;
    PIPE_PRIO(role);
    const int bcol0 = blockIdx.x * TRIG_GROUPS * CPW;
    const int bnb = min(TRIG_GROUPS * CPW, ncol - bcol0);           // columns of this block
    const int col = bcol0 + group * CPW + lane / LPC;
    const bool valid = col < ncol;
    const bool first = (lane & (LPC - 1)) == 0;                     // the lane of a pair / quad that stores the column's results
    const size_t L = (size_t)pver, LP = (size_t)pver + 1;
    const int rows = pver + 2 - (P.limcnv - 1);               // the profile is only touched at levels > msg = limcnv - 1
    double *gmem = smem + (size_t)group * (rows * WAVE + MAIL_DOUBLES);

    // zero-fill every output of this block's columns, coalesced (:559-563, 625-650, 771-784 and the
    // gathered arrays, whose slots beyond lengath the reference leaves stale)
    {
        double *mid[] = {io.qtnd, io.heat, io.cme, io.zdu, io.rprd, io.dlf, io.ql, io.dif, io.dnlf, io.dnif, io.eurt,
                         io.mu, io.md, io.du, io.eu, io.ed, io.dp};
        for (double *p : mid) block_fill(p + bcol0 * L, bnb * L, 0.0);
        block_fill(io.mcon + bcol0 * LP, bnb * LP, 0.0);
        block_fill(io.pflx + bcol0 * LP, bnb * LP, 0.0);
    }
    if (role == 1 && valid && first) {
        io.prec[col] = 0.0; io.rliq[col] = 0.0; io.rice[col] = 0.0;
        io.jctop[col] = pver; io.jcbot[col] = 1;
        io.dsubcld[col] = 0.0; io.jt[col] = 0; io.maxg[col] = 0; io.ideep[col] = 0;
        ws.deep[col] = 0;
    }

    // lanes past the last column shadow it (the pipeline's barriers need every lane); nothing of theirs is stored
    const ColIn c = make_colin(P, io, valid ? col : ncol - 1, pver);
    const double pblt = pbl_top_index(c);
    BuoyRes r;
    buoyan_dilute_2w<false, true, LPC>(P, c, pblt, -P.tentrm, 0.0, role, lane, gmem + lane - (P.limcnv - 1) * WAVE, gmem + rows * WAVE, nullptr,
                                       nullptr, false, r);                                        // :544, 886
    if (role == 0 || !valid || !first) return;

    io.cape[col] = r.cape;
    io.status[col] = r.status;
    ws.cape1[col] = r.cape; ws.cin1[col] = r.cin; ws.lel1[col] = r.lel; ws.mx1[col] = r.mx;
    if (r.cape > P.capelmt && r.cin < r.cape * P.cin_threshd) {     // :908-915
        const int slot = atomicAdd(ws.count, 1);
        ws.list[slot] = col;
    }
}

// The same trigger with ONE wave per 64 columns doing both sweeps of a level back to back (buoyan_dilute_2w<.., PIPE =
// false>): for column counts that fill the SIMDs on their own (ne120: 12 150 column groups) and for level counts
// where the pipeline's barriers and its second copy of the launch-level search only cost (measured at ne120: 7.2 ms
// fused, 9.2 ms paired).
__global__ __launch_bounds__(WAVE) void zm_trigger_fused_kernel(ZmDev P, int ncol, int pver, colphys_zm_convr_io io,
                                                               ZmWorkspace ws)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];            // [(pver+2)][WAVE] parcel virtual temperature / buoyancy
    const int col0 = blockIdx.x * WAVE;
    const int nb = min(WAVE, ncol - col0);
    const int lane = threadIdx.x;
    const int col = col0 + lane;
    const size_t L = (size_t)pver, LP = (size_t)pver + 1;
    {
        double *mid[] = {io.qtnd, io.heat, io.cme, io.zdu, io.rprd, io.dlf, io.ql, io.dif, io.dnlf, io.dnif, io.eurt,
                         io.mu, io.md, io.du, io.eu, io.ed, io.dp};
        for (double *p : mid) block_fill(p + col0 * L, nb * L, 0.0);
        block_fill(io.mcon + col0 * LP, nb * LP, 0.0);
        block_fill(io.pflx + col0 * LP, nb * LP, 0.0);
    }
    if (lane >= nb) return;
    io.prec[col] = 0.0; io.rliq[col] = 0.0; io.rice[col] = 0.0;
    io.jctop[col] = pver; io.jcbot[col] = 1;
    io.dsubcld[col] = 0.0; io.jt[col] = 0; io.maxg[col] = 0; io.ideep[col] = 0;
    ws.deep[col] = 0;

    const ColIn c = make_colin(P, io, col, pver);
    const double pblt = pbl_top_index(c);
    BuoyRes r;
    buoyan_dilute_2w<false, false, 1>(P, c, pblt, -P.tentrm, 0.0, 1, lane, smem + lane - (P.limcnv - 1) * WAVE, nullptr, nullptr, nullptr, false, r);   // :544, 886

    io.cape[col] = r.cape;
    io.status[col] = r.status;
    ws.cape1[col] = r.cape; ws.cin1[col] = r.cin; ws.lel1[col] = r.lel; ws.mx1[col] = r.mx;
    if (r.cape > P.capelmt && r.cin < r.cape * P.cin_threshd) {     // :908-915
        const int slot = atomicAdd(ws.count, 1);
        ws.list[slot] = col;
    }
}

// ------------------------------------------------------------------------------------------------------
// Plume ensemble on ONE WAVE PER COLUMN (cldprp :3024-4026, closure :4028-4260, q1q2_pjr :4262-4421).
//
// Only the triggered columns (~10 %) get here, so a lane-per-column mapping leaves the machine almost empty
// (80 waves) and each lane drags ~50 private work profiles through scratch.  Instead every triggered column
// gets its own wave, the work profiles live in LDS (49 names on 36 slots of pver+2 doubles: 10 KB at L32, see PlumeLds), and
//   * loops that are element-wise in k run one level per lane            (PAR),
//   * the genuine recurrences / searches / running sums run in the reference's order, redundantly on all
//     lanes (uniform control flow: in SIMT that costs exactly what one lane would), reading and writing LDS.
// A barrier separates a PAR phase from whatever reads its neighbours' levels.  Same arithmetic, same order.
// PSTEP = lanes per column: 64, or 32 in the two-columns-per-wave mode (pver <= 32; not selected by default any more, see
// launch_zm_convr) -- then a wave carries TWO columns, lanes 0..31 the first and 32..63 the second, each half with its own LDS
// profiles (otherwise half the lanes of every PAR loop idle and twice the waves run the serial loops).  Nothing else changes: a lane belongs to a column and does what that column's control flow says; where
// the two columns of a wave disagree (loop bounds, `eps0 > 0`, the break of :3690) the wave runs both sides under EXEC
// masks.  A block is one wave, so SYNC() inside such a branch is safe: LDS operations of a wave complete in order.
// ------------------------------------------------------------------------------------------------------
#define PAR(k, lo, hi) for (int k = (lo) + lane; k <= (hi); k += PSTEP)      // PSTEP: lanes per column (template parameter in scope)
// A plume block is ONE wave: SYNC() orders the wave's own LDS traffic (colphys_common.h wave_sync).  It used to be __syncthreads(),
// i.e. an s_barrier executed under whatever EXEC mask a divergent branch of the two-columns-per-wave mode had left -- harmless
// for a one-wave workgroup in practice, but outside what HIP defines; -DPLUME_SYNC_BARRIER restores it (scripts/plume_hazard.sh).
#ifdef PLUME_SYNC_BARRIER
#define SYNC() __syncthreads()
#else
#define SYNC() wave_sync()
#endif

// cldprp's work profiles: 49 names on 36 LDS slots (round 3: the waves resident per CU are set by this footprint -- 11 at L32
// with 49 slots, 15 with 36 -- and the plume kernels' time at scale by the waves resident).  cldprp runs in three phases --
// entrainment (k1 .. i4, f, eps, gamma, mcp, mrl, hsat, z: last read at :3598), updraft thermodynamics, downdraft (md, ed, sd, qd,
// mc, evp, cmeg, hd, td, qds, pflx: first touched at :3880) -- and a downdraft profile takes the slot of an entrainment profile;
// its initial value (:3245-3314) is set when the downdraft phase begins instead of at the top, to the same bits.  tp / qstp
// (closure only) take the slots of tu / hu (updraft phase only) and are loaded after cldprp.
constexpr int PLUME_NARR = 36;
#ifndef SERIAL_UNROLL
#define SERIAL_UNROLL 4                    // serial level loops of the plume kernels: four iterations of LDS reads in flight (plume2 0.21 -> 0.19 ms)
#endif

struct PlumeLds {
    double *q, *t, *p, *z, *s, *zf, *shat, *qhat, *dp, *tp, *qstp;
    double *mu, *eu, *du, *md, *ed, *sd, *qd, *mc, *qu, *su, *qst, *ql, *cmeg, *pflx, *evp, *cu, *rprd, *qcde;
    double *gamma, *dz, *hu, *hd, *eps, *f, *k1, *i2, *i3, *i4, *qsthat, *hsthat, *gamhat, *qds, *mcp, *mrl, *tu, *td, *hmn,
        *hsat;
    __device__ void carve(double *base, int LD)
    {
        double **slots[] = {&q, &t, &p, &z, &s, &zf, &shat, &qhat, &dp, &mu, &eu, &du, &qu, &su, &qst, &ql, &cu, &rprd, &qcde,
                            &gamma, &dz, &hu, &eps, &f, &k1, &i2, &i3, &i4, &qsthat, &hsthat, &gamhat, &mcp, &mrl, &tu, &hmn, &hsat};
        static_assert(sizeof(slots) / sizeof(slots[0]) == PLUME_NARR, "PLUME_NARR");
        int n = 0;
        for (double **sl : slots) { *sl = base + (size_t)n * LD; ++n; }
        // downdraft phase over entrainment phase
        md = k1; ed = i2; sd = i3; qd = i4; mc = f; evp = eps; cmeg = gamma; hd = mcp; td = mrl; qds = hsat; pflx = z;
        // closure over updraft phase
        tp = tu; qstp = hu;
    }
};

struct PlumeIdx { int jt, jlcl, j0, jd; };

// column inputs -> LDS (:825-860, 926-940) and the interface values shat / qhat (:1007-1027)
template <int PSTEP>
__device__ void plume_load(const ZmDev &P, const ColIn &c, const PlumeLds &A, int lane)
{
    const int pver = c.pver, msg = c.msg;
    PAR(k, 1, pver) {
        const double qk = c.Q(k), tk = c.T(k), zk = c.Z(k);
        A.q[k] = qk; A.t[k] = tk; A.p[k] = c.Pm(k); A.z[k] = zk; A.zf[k] = c.Zf(k);
        const double sk = tk + (P.grav / ((1.0 + P.zvir * qk) * P.cpres)) * zk;
        A.s[k] = sk; A.shat[k] = sk; A.qhat[k] = qk;
        A.dp[k] = 0.01 * c.dpp[k];
    }
    if (lane == 0) A.zf[pver + 1] = c.Zf(pver + 1);
    SYNC();
    PAR(k, msg + 2, pver) {
        const double sg = A.s[k], sgm = A.s[k - 1], qg = A.q[k], qgm = A.q[k - 1];
        double sdifr = 0.0, qdifr = 0.0;
        if (sg > 0.0 || sgm > 0.0) sdifr = fabs((sg - sgm) / fmax(sgm, sg));
        if (qg > 0.0 || qgm > 0.0) qdifr = fabs((qg - qgm) / fmax(qgm, qg));
        if (sdifr > 1.E-6) A.shat[k] = log(sgm / sg) * sgm * sg / (sgm - sg);
        else A.shat[k] = 0.5 * (sg + sgm);
        if (qdifr > 1.E-6) A.qhat[k] = log(qgm / qg) * qgm * qg / (qgm - qg);
        else A.qhat[k] = 0.5 * (qg + qgm);
    }
    SYNC();
}

// cldprp :3024-4026 (non-microphysics path)
// ENTRAINMENT_ONLY: the first call of the reference (:1032) is made for one thing, the entrainment profile eu that
// sets the second ascent's rate (:1046-1078); everything cldprp computes after eu is final is overwritten by the second
// call, so the first pass stops there.
template <bool ENTRAINMENT_ONLY, int PSTEP>
__device__ void cldprp_w(const ZmDev &P, int pver, int msg, const PlumeLds &A, int jb, int lel, double landfrac, int lane,
                         PlumeIdx &o)
{
    const int mx = jb;
    const double rd = P.rgas, grav = P.grav, cp = P.cp, rl = P.rl, dcol = P.dcol, eps1 = P.eps1;
    const double zvir = P.zvir, cpvir = P.cpvir, tmelt = P.tmelt;
    const double small = 1.e-20;
    double hmin, expdif = 0.0, eps0, epsm = 0.0, totpcp = 0.0, totevp = 0.0, mdt;

    const double c0mask = P.c0_ocn * (1.0 - landfrac) + P.c0_lnd * landfrac;                 // :3239-3240
    const double tiedke_msk = P.tiedke_add * (1.0 - landfrac) + P.tiedke_lnd * landfrac;

    PAR(k, 1, pver) {                                                                        // :3245-3314
        const double qk = A.q[k], tk = A.t[k], pk = A.p[k], sk = A.s[k];
        A.dz[k] = A.zf[k] - A.zf[k + 1];
        A.k1[k] = 0.0; A.i2[k] = 0.0; A.i3[k] = 0.0; A.i4[k] = 0.0;
        A.mu[k] = 0.0; A.f[k] = 0.0; A.eps[k] = 0.0; A.eu[k] = 0.0; A.du[k] = 0.0; A.ql[k] = 0.0; A.cu[k] = 0.0;
        A.qu[k] = qk; A.su[k] = sk;          // (evp cmeg qds md ed sd qd mc hd td: set where the downdraft phase begins, see PlumeLds)
        double est, qst;
        qsat_hPa(P, tk, pk, est, qst);
        if (pk - est <= 0.0) qst = 1.0;
        A.qst[k] = qst;
        const double mrd = (1.0 + zvir * qk) * rd;
        const double mcp = (1.0 + cpvir * qk) * cp;
        const double mrl = (1.0 - dcol * (tk - tmelt)) * rl;
        A.mcp[k] = mcp; A.mrl[k] = mrl;
        A.gamma[k] = qst * (1.0 + qst / eps1) * eps1 * mrl / (mrd * (tk * tk)) * mrl / mcp;
        const double hmn = mcp * tk + grav * A.z[k] + mrl * qk;
        A.hmn[k] = hmn;
        A.hsat[k] = mcp * tk + grav * A.z[k] + mrl * qst;
        A.hu[k] = hmn;
        A.rprd[k] = 0.0;
        A.qcde[k] = 0.0;
        A.tu[k] = 0.0;
    }
    SYNC();
    PAR(k, 1, pver) {                                                                        // :3349-3378
        if (k <= msg + 1) {
            A.hsthat[k] = A.hsat[k]; A.qsthat[k] = A.qst[k]; A.gamhat[k] = A.gamma[k];
        } else {
            const double qa = A.qst[k - 1], qb = A.qst[k], ga = A.gamma[k - 1], gb = A.gamma[k];
            double qsthat;
            if (fabs(qa - qb) > 1.E-6) qsthat = log(qa / qb) * qa * qb / (qa - qb);
            else qsthat = qb;
            A.qsthat[k] = qsthat;
            A.hsthat[k] = A.mcp[k] * A.shat[k] + A.mrl[k] * qsthat;
            if (fabs(ga - gb) > 1.E-6) A.gamhat[k] = log(ga / gb) * ga * gb / (ga - gb);
            else A.gamhat[k] = gb;
        }
    }
    SYNC();

    o.jt = max(lel, P.limcnv + 1);                                                           // :3383-3390
    o.jt = min(o.jt, pver);
    o.jd = pver;
    o.jlcl = lel;
    hmin = 1.E6;
    o.j0 = pver;
#pragma unroll SERIAL_UNROLL
    for (int k = msg + 1; k <= pver; ++k) {                                                  // :3395-3402
        const double h = A.hsat[k];
        if (h <= hmin && k >= o.jt && k <= jb) { hmin = h; o.j0 = k; }
    }
    o.j0 = min(o.j0, jb - 2);                                                                // :3404-3409
    o.j0 = max(o.j0, o.jt + 2);
    o.j0 = min(o.j0, pver);

    const double hmn_mx = A.hmn[mx], s_mx = A.s[mx];
    PAR(k, msg + 1, pver) {                                                                  // :3414-3424
        if (k >= o.jt && k <= jb) {
            A.hu[k] = hmn_mx + cp * tiedke_msk;
            A.su[k] = s_mx + tiedke_msk / (1.0 + cpvir * A.qu[k]);
        }
    }
    {                                                                                        // :3430-3442 prefix sums
        double k1p = 0.0, i2p = 0.0, i3p = 0.0, i4p = 0.0;     // values at k+1 (all zero at pver)
#pragma unroll SERIAL_UNROLL
        for (int k = pver - 1; k >= msg + 1; --k) {
            if (k < jb && k >= o.jt) {
                const double dzk = A.dz[k];
                const double k1k = k1p + (hmn_mx - A.hmn[k]) * dzk;
                const double ihat = 0.5 * (k1p + k1k);
                const double i2k = i2p + ihat * dzk;
                const double idag = 0.5 * (i2p + i2k);
                const double i3k = i3p + idag * dzk;
                const double iprm = 0.5 * (i3p + i3k);
                const double i4k = i4p + iprm * dzk;
                A.k1[k] = k1k; A.i2[k] = i2k; A.i3[k] = i3k; A.i4[k] = i4k;
                k1p = k1k; i2p = i2k; i3p = i3k; i4p = i4k;
            } else {
                k1p = 0.0; i2p = 0.0; i3p = 0.0; i4p = 0.0;     // untouched levels hold their initial zeros
            }
        }
    }
    hmin = 1.E6;                                                                             // :3446-3456
#pragma unroll SERIAL_UNROLL
    for (int k = msg + 1; k <= pver; ++k) {
        const double h = A.hmn[k];
        if (k >= o.j0 && k <= jb && h <= hmin) {
            hmin = h;
            expdif = hmn_mx - hmin;
        }
    }
    SYNC();
    PAR(k, msg + 2, pver) {                                                                  // :3462-3487
        double expnum = 0.0;
        double k1k = A.k1[k];
        if (k < o.jt || k >= jb) {
            k1k = 0.0;
            A.k1[k] = 0.0;
        } else {
            expnum = hmn_mx - (A.hsat[k - 1] * (A.zf[k] - A.z[k]) + A.hsat[k] * (A.z[k - 1] - A.zf[k])) / (A.z[k - 1] - A.z[k]);
        }
        if ((expdif > 100.0 && expnum > 0.0) && k1k > expnum * A.dz[k]) {
            const double i2k = A.i2[k], i3k = A.i3[k], i4k = A.i4[k];
            const double ftemp = expnum / k1k;
            const double ft2 = ftemp * ftemp, ft3 = ft2 * ftemp, ft4 = ft2 * ft2;
            const double i2sq = i2k * i2k, i2cu = i2sq * i2k;
            const double k1sq = k1k * k1k, k1cu = k1sq * k1k;
            double fk = ftemp + i2k / k1k * ft2 + (2.0 * i2sq - k1k * i3k) / k1sq * ft3
                        + (-5.0 * k1k * i2k * i3k + 5.0 * i2cu + k1sq * i4k) / k1cu * ft4;
            fk = fmax(fk, 0.0);
            fk = fmin(fk, P.entrmn);
            A.f[k] = fk;
        }
    }
    SYNC();
    if (o.j0 < jb) {                                                                         // :3488-3492
        if (A.f[o.j0] < 1.E-6 && A.f[o.j0 + 1] > A.f[o.j0]) o.j0 = o.j0 + 1;
    }
    {
        double f_c = 0.0;
        bool carried = false;
#pragma unroll SERIAL_UNROLL
        for (int k = msg + 2; k <= pver; ++k) {                                              // :3493-3499 running max, carried in a register
            if (k >= o.jt && k <= o.j0) {
                const double fk = fmax(A.f[k], carried ? f_c : A.f[k - 1]);
                A.f[k] = fk;
                f_c = fk; carried = true;
            } else {
                carried = false;
            }
        }
    }
    eps0 = A.f[o.j0];                                                                        // :3500-3503
    const double f_j0 = eps0;
    PAR(k, msg + 1, pver) {                                                                  // :3502-3518
        if (k >= o.j0 && k <= jb) A.eps[k] = f_j0;
        else if (k < o.j0 && k >= o.jt) A.eps[k] = A.f[k];
        else if (k == jb) A.eps[k] = eps0;
    }
    SYNC();

    const int tmplel = o.jt;                                                                 // :3558
    const double zf_jb = A.zf[jb];
    PAR(k, msg + 1, pver) {                                                                  // :3550-3571, mu first
        if (eps0 > 0.0) {
            if (k == jb) A.mu[k] = 1.0;
            else if (k >= tmplel && k < jb) {
                const double zuef = A.zf[k] - zf_jb;
                A.mu[k] = (1.0 / eps0) * (exp(A.eps[k] * zuef) - 1.0) / zuef;
            }
        }
    }
    SYNC();
    PAR(k, msg + 1, pver) {                                                                  // ... then eu, du
        if (eps0 > 0.0) {
            if (k == jb) A.eu[k] = A.mu[k] / A.dz[k];
            else if (k >= tmplel && k < jb) {
                const double zuef = A.zf[k] - zf_jb;
                const double rmue = (1.0 / eps0) * (exp(A.eps[k + 1] * zuef) - 1.0) / zuef;
                A.eu[k] = (rmue - A.mu[k + 1]) / A.dz[k];
                A.du[k] = (rmue - A.mu[k]) / A.dz[k];
            }
        }
    }
    SYNC();
    if (eps0 > 0.0) {                                                                        // :3579-3598
        double hu_kp1 = A.hu[jb], mu_kp1 = A.mu[jb];            // carried in registers down the recurrence
#pragma unroll SERIAL_UNROLL
        for (int k = jb - 1; k >= lel; --k) {
            const double muk = A.mu[k];
            if (muk < 0.02) {
                hu_kp1 = A.hmn[k];
                A.hu[k] = hu_kp1;
                A.mu[k] = 0.0;
                A.eu[k] = 0.0;
                A.du[k] = mu_kp1 / A.dz[k];
                mu_kp1 = 0.0;
            } else {
                hu_kp1 = mu_kp1 / muk * hu_kp1 + A.dz[k] / muk * (A.eu[k] * A.hmn[k] - A.du[k] * A.hsat[k]);
                A.hu[k] = hu_kp1;
                mu_kp1 = muk;
            }
        }
    }
    {
        bool doit = true;                                                                    // :3603-3628
        const double hu_jb = A.hu[jb];
#pragma unroll SERIAL_UNROLL
        for (int k = jb - 2; k >= lel - 1; --k) {
            if (doit && k >= 1) {
                const double huk = A.hu[k], muk = A.mu[k];
                if (huk <= A.hsthat[k] && A.hu[k + 1] > A.hsthat[k + 1] && muk >= 0.02) {
                    if (huk - A.hsthat[k] < -2000.0) { o.jt = k + 1; doit = false; }
                    else { o.jt = k; doit = false; }
                } else if ((huk > hu_jb) || muk < 0.02) {          // totfrz == 0 without microphysics
                    o.jt = k + 1;
                    doit = false;
                }
            }
        }
    }
    {                                                                                        // :3632-3646
        const double mu_jtp1 = (o.jt + 1 <= pver) ? A.mu[o.jt + 1] : 0.0;
        SYNC();
        PAR(k, msg + 1, pver) {
            if (k >= lel && k <= o.jt && eps0 > 0.0) { A.mu[k] = 0.0; A.eu[k] = 0.0; A.du[k] = 0.0; A.hu[k] = A.hmn[k]; }
            if (k == o.jt && eps0 > 0.0) {
                A.du[k] = mu_jtp1 / A.dz[k];
                A.eu[k] = 0.0;
                A.mu[k] = 0.0;
            }
        }
    }
    SYNC();
    if (ENTRAINMENT_ONLY) return;                                                            // eu is final (:3632-3646)
    PAR(k, msg + 2, pver) {                                                                  // :3649-3654
        A.tu[k] = (A.hu[k] - grav * A.zf[k] - (1.0 + dcol * tmelt) * rl * A.qu[k]) / (cp * (1.0 + (cpvir - dcol * (rl / cp)) * A.qu[k]));
    }
    SYNC();
    // su / qu of the level below ride in registers (k = jb sets them, jt < k < jb continues them; never read across a gap)
    double su_c = 0.0, qu_c = 0.0;
    bool uc = false;
#pragma unroll SERIAL_UNROLL
    for (int k = pver; k >= msg + 2; --k) {                                                  // :3656-3692
        bool wrote = false;
        if (k == jb && eps0 > 0.0) {
            const double quk = A.q[mx];
            A.qu[k] = quk;
            const double tuk = (A.hu[k] - grav * A.zf[k] - (1.0 + dcol * tmelt) * rl * quk) / (cp * (1.0 + (cpvir - dcol * (rl / cp)) * quk));
            A.tu[k] = tuk;
            const double suk = (A.hu[k] - (1.0 - dcol * (tuk - tmelt)) * rl * quk) / ((1.0 + cpvir * quk) * cp);
            A.su[k] = suk;
            su_c = suk; qu_c = quk; wrote = true;
        }
        if ((k > o.jt && k < jb) && eps0 > 0.0) {
            const double muk = A.mu[k], mukp1 = A.mu[k + 1], dzk = A.dz[k], euk = A.eu[k], duk = A.du[k];
            const double sukp1 = uc ? su_c : A.su[k + 1], qukp1 = uc ? qu_c : A.qu[k + 1];
            const double suk = mukp1 / muk * sukp1 + dzk / muk * (euk - duk) * A.s[k];
            const double quk = mukp1 / muk * qukp1 + dzk / muk * (euk * A.q[k] - duk * A.qst[k]);
            A.su[k] = suk; A.qu[k] = quk;
            su_c = suk; qu_c = quk; wrote = true;
            // 0.85 is a default-REAL literal in the reference (:3680): single precision, then promoted
            const double tuk = suk - grav / ((1.0 + (double)0.85f * quk) * cp) * A.zf[k];
            A.tu[k] = tuk;
            double estu, qstu;
            qsat_hPa(P, tuk, (A.p[k] + A.p[k - 1]) / 2.0, estu, qstu);
            if (quk >= qstu) {
                o.jlcl = k;
                break;                                                                       // :3690
            }
        }
        uc = wrote;
    }
    SYNC();
    PAR(k, msg + 2, pver) {                                                                  // :3693-3707
        if ((k > o.jt && k <= o.jlcl) && eps0 > 0.0) {
            const double gh = A.gamhat[k], dh = A.hu[k] - A.hsthat[k];
            const double quk = A.qsthat[k] + gh * dh / ((1.0 - dcol * (A.tu[k] - tmelt)) * rl * (1.0 + gh));
            const double suk = A.shat[k] + dh / ((1.0 + cpvir * quk) * cp * (1.0 + gh));
            A.qu[k] = quk; A.su[k] = suk;
            A.tu[k] = suk - grav / ((1.0 + cpvir * quk) * cp) * A.zf[k];
        }
    }
    SYNC();
    PAR(k, msg + 2, pver) {                                                                  // :3716-3739
        if (k >= o.jt && k < jb && eps0 > 0.0) {
            double cuk = ((A.mu[k] * A.su[k] - A.mu[k + 1] * A.su[k + 1]) / A.dz[k] - (A.eu[k] - A.du[k]) * A.s[k]) / (rl / cp)
                         * ((1.0 + cpvir * A.qu[k]) / (1.0 - dcol * (A.tu[k] - tmelt)));
            if (k == o.jt) cuk = 0.0;
            A.cu[k] = fmax(0.0, cuk);
        }
    }
    SYNC();
    double ql_c = 0.0;                                    // ql of the level below in a register
    bool qc = false;
#pragma unroll SERIAL_UNROLL
    for (int k = pver; k >= msg + 2; --k) {                                                  // :3846-3870
        double rp = 0.0;
        const double muk = A.mu[k];
        bool wrote = false;
        if (k >= o.jt && k < jb && eps0 > 0.0 && muk >= 0.0) {
            const double dzk = A.dz[k], duk = A.du[k], cuk = A.cu[k], qlkp1 = qc ? ql_c : A.ql[k + 1];
            double qlk;
            if (muk > 0.0) {
                const double ql1 = 1.0 / muk * (A.mu[k + 1] * qlkp1 - dzk * duk * qlkp1 + dzk * cuk);
                qlk = ql1 / (1.0 + dzk * c0mask);
            } else {
                qlk = 0.0;
            }
            A.ql[k] = qlk;
            totpcp = totpcp + dzk * (cuk - duk * qlkp1);
            rp = c0mask * muk * qlk;
            A.qcde[k] = qlk;
            ql_c = qlk; wrote = true;
        }
        A.rprd[k] = rp;
        qc = wrote;
    }

    // downdraft :3880-3975.  Its profiles take the LDS slots of the entrainment phase's (PlumeLds): their initial values of
    // :3245-3314, which that loop no longer sets, are set here -- same expressions, same operands (hmn as stored)
    SYNC();
    PAR(k, 1, pver) {
        const double qk = A.q[k], hmn = A.hmn[k];
        A.evp[k] = 0.0; A.cmeg[k] = 0.0; A.qds[k] = qk; A.md[k] = 0.0; A.ed[k] = 0.0; A.sd[k] = A.s[k];
        A.qd[k] = qk; A.mc[k] = 0.0;
        A.hd[k] = hmn;
        A.td[k] = (hmn - grav * A.zf[k] - (1.0 + dcol * tmelt) * rl * qk) / (cp * (1.0 + (cpvir - dcol * (rl / cp)) * qk));
    }
    SYNC();
    const double alfa = P.alfadet;
    o.jt = min(o.jt, jb - 1);
    o.jd = max(o.j0, o.jt + 1);
    o.jd = min(o.jd, jb);
    const int jd = o.jd, jt = o.jt;
    A.hd[jd] = A.hmn[jd - 1];
    if (jd < jb && eps0 > 0.0) {
        epsm = eps0;
        A.md[jd] = -alfa * epsm / eps0;
    }
    SYNC();
    const double zf_jd = A.zf[jd];
    PAR(k, msg + 1, pver) {                                                                  // :3896-3904
        if ((k > jd && k <= jb) && eps0 > 0.0) {
            const double zdef = zf_jd - A.zf[k];
            A.md[k] = -alfa / (2.0 * eps0) * (exp(2.0 * epsm * zdef) - 1.0) / zdef;
        }
    }
    SYNC();
    if (eps0 > 0.0 && jd < jb) {                                                             // :3906-3913
        const double ratmjb = fmin(fabs(A.mu[jb] / A.md[jb]), 1.0);   // md(jb) as it is before its own scaling
        SYNC();
        PAR(k, msg + 1, pver) {
            if (k >= jt && k <= jb) A.md[k] = A.md[k] * ratmjb;
        }
    }
    SYNC();
    {   // hd's recurrence in a register (levels k >= jt are contiguous; k = jt reads hd(jt-1) as stored)
        double hd_c = 0.0;
        bool carried = false;
#pragma unroll SERIAL_UNROLL
        for (int k = msg + 1; k <= pver; ++k) {                                              // :3916-3924
            if ((k >= jt && k <= pver) && eps0 > 0.0) {
                const double edkm1 = (A.md[k - 1] - A.md[k]) / A.dz[k - 1];
                A.ed[k - 1] = edkm1;
                mdt = fmin(A.md[k], -small);
                const double hdkm1 = carried ? hd_c : A.hd[k - 1];
                const double hdk = (A.md[k - 1] * hdkm1 - A.dz[k - 1] * edkm1 * A.hmn[k - 1]) / mdt;
                A.hd[k] = hdk;
                hd_c = hdk; carried = true;
            } else {
                carried = false;
            }
        }
    }
    SYNC();
    PAR(k, msg + 2, pver) {                                                                  // :3928-3941
        if ((k >= jd && k <= jb) && eps0 > 0.0 && jd < jb) {
            const double gh = A.gamhat[k], dh = A.hd[k] - A.hsthat[k];
            double qds = A.qsthat[k] + gh * dh / (rl * (1.0 + gh));
            const double tdk = (A.hd[k] - grav * A.zf[k] - (1.0 + dcol * tmelt) * rl * qds) / (cp * (1.0 + (cpvir - dcol * (rl / cp)) * qds));
            A.td[k] = tdk;
            qds = A.qsthat[k] + gh * dh / ((1.0 - dcol * (tdk - tmelt)) * rl * (1.0 + gh));
            A.qds[k] = qds;
        }
    }
    SYNC();
    {                                                                                        // :3943-3951
        const int k = jd;
        const double qdk = A.qds[k];
        A.qd[k] = qdk;
        const double sdk = (A.hd[k] - (1.0 - dcol * (A.td[k] - tmelt)) * rl * qdk) / ((1.0 + cpvir * qdk) * cp);
        A.sd[k] = sdk;
        A.td[k] = sdk - grav / ((1.0 + cpvir * qdk) * cp) * A.zf[k];
    }
    {   // the recurrences of qd and sd ride in registers: the levels k in [jd, jb) are contiguous, so what level k reads at k
        // is what level k-1 stored at (k-1)+1 -- except at k = jd, which reads the values set just above
        double qd_c = 0.0, sd_c = 0.0;
        bool carried = false;
#pragma unroll SERIAL_UNROLL
        for (int k = msg + 2; k <= pver; ++k) {                                              // :3953-3971
            if (k >= jd && k < jb && eps0 > 0.0) {
                const double qdkp1 = A.qds[k + 1];
                A.qd[k + 1] = qdkp1;
                const double edk = A.ed[k], qk = A.q[k], dzk = A.dz[k], mdk = A.md[k], mdkp1 = A.md[k + 1];
#ifdef PLUME_QDSD_LDS                    // experiment (scripts/plume_hazard.sh): round 2's form, the recurrence read back through LDS.
                                         // It is as correct as the register carry; it merely was the allocation under which hipcc
                                         // miscompiled zm_plume2_kernel<2> (isa_check.py), which round 2 took for an LDS hazard.
                const double qdk = A.qd[k], sdk = A.sd[k];
#else
                const double qdk = carried ? qd_c : A.qd[k], sdk = carried ? sd_c : A.sd[k];
#endif
                double evpk = -edk * qk + (mdk * qdk - mdkp1 * qdkp1) / dzk;
                evpk = fmax(evpk, 0.0);
                A.evp[k] = evpk;
                mdt = fmin(mdkp1, -small);
                const double sdkp1 = (((1.0 - dcol * (A.td[k] - tmelt)) * rl / ((1.0 + cpvir * qdk) * cp) * evpk - edk * A.s[k]) * dzk + mdk * sdk) / mdt;
                A.sd[k + 1] = sdkp1;
                totevp = totevp - dzk * edk * qk;
                qd_c = qdkp1; sd_c = sdkp1; carried = true;
            } else {
                carried = false;
            }
#ifdef PLUME_QDSD_SYNC
            SYNC();
#endif
        }
    }
    totevp = totevp + A.md[jd] * A.qd[jd] - A.md[jb] * A.qd[jb];                             // :3974
    totpcp = fmax(totpcp, 0.0);
    totevp = fmax(totevp, 0.0);
    SYNC();
    PAR(k, msg + 2, pver) {                                                                  // :3993-4009
        double evpk;
        if (totevp > 0.0 && totpcp > 0.0) {
            A.md[k] = A.md[k] * fmin(1.0, totpcp / (totevp + totpcp));
            A.ed[k] = A.ed[k] * fmin(1.0, totpcp / (totevp + totpcp));
            evpk = A.evp[k] * fmin(1.0, totpcp / (totevp + totpcp));
        } else {
            A.md[k] = 0.0; A.ed[k] = 0.0; evpk = 0.0;
        }
        A.evp[k] = evpk;
        A.cmeg[k] = A.cu[k] - evpk;
        A.rprd[k] = A.rprd[k] - evpk;
    }
    SYNC();
    {                                                                                        // :4012-4017 prefix sum
        double pf = 0.0;
        A.pflx[1] = 0.0;
#pragma unroll SERIAL_UNROLL
        for (int k = 2; k <= pver + 1; ++k) {
            pf = pf + A.rprd[k - 1] * A.dz[k - 1];
            A.pflx[k] = pf;
        }
    }
    PAR(k, msg + 1, pver) A.mc[k] = A.mu[k] + A.md[k];                                       // :4019-4023
    SYNC();
}

// closure :4028-4260: the per-level integrand one level per lane, the sum in the reference's order
template <int PSTEP>
__device__ double closure_w(const ZmDev &P, int pver, int msg, const PlumeLds &A, double *s_dboy, double dsubcld, double cape,
                            double tl, int lcl, int lel, int jt, int mx, int lane)
{
    const double rd = P.rgas, cp = P.cp, rl = P.rl, eps1 = P.eps1, grav = P.grav;
    const double *q = A.q, *t = A.t, *p = A.p, *s = A.s, *tp = A.tp, *qstp = A.qstp, *qhat = A.qhat, *shat = A.shat, *dp = A.dp;
    const double *mu = A.mu, *md = A.md, *su = A.su, *sd = A.sd, *qu = A.qu, *qd = A.qd, *mc = A.mc, *du = A.du, *ql = A.ql;
    double mb = 0.0;
    const double eb = p[mx] * q[mx] / (eps1 + q[mx]);
    const double dtbdt = (1.0 / dsubcld) * (mu[mx] * (shat[mx] - su[mx]) + md[mx] * (shat[mx] - sd[mx]));
    const double dqbdt = (1.0 / dsubcld) * (mu[mx] * (qhat[mx] - qu[mx]) + md[mx] * (qhat[mx] - qd[mx]));
    const double debdt = eps1 * p[mx] / ((eps1 + q[mx]) * (eps1 + q[mx])) * dqbdt;
    const double lg = (3.5 * log(t[mx]) - log(eb) - 4.805);
    const double dtldt = -2840.0 * (3.5 / t[mx] * dtbdt - debdt / eb) / (lg * lg);
    const double beta = 0.0;

    PAR(k, lel, mx - 1) {
        double dtmdt = 0.0, dqmdt = 0.0, dboydt;
        if (k == jt && k <= pver - 1) {                                                      // :4154-4169
            dqmdt = (1.0 / dp[k]) * (mu[k + 1] * (qu[k + 1] - qhat[k + 1] + ql[k + 1]) + md[k + 1] * (qd[k + 1] - qhat[k + 1]));
            dtmdt = (1.0 / dp[k]) * (mu[k + 1] * (su[k + 1] - shat[k + 1] - rl / cp * ql[k + 1]) + md[k + 1] * (sd[k + 1] - shat[k + 1]));
        }
        if (k > jt && k < mx) {                                                              // :4172-4189
            dtmdt = (mc[k] * (shat[k] - s[k]) - mc[k + 1] * (shat[k + 1] - s[k])) / dp[k]
                    - rl / cp * du[k] * (beta * ql[k] + (1 - beta) * ql[k + 1]);
            dqmdt = (mu[k + 1] * (qu[k + 1] - qhat[k + 1] + cp / rl * (su[k + 1] - s[k]))
                     - mu[k] * (qu[k] - qhat[k] + cp / rl * (su[k] - s[k]))
                     + md[k + 1] * (qd[k + 1] - qhat[k + 1] + cp / rl * (sd[k + 1] - s[k]))
                     - md[k] * (qd[k] - qhat[k] + cp / rl * (sd[k] - s[k]))) / dp[k]
                    + du[k] * (beta * ql[k] + (1 - beta) * ql[k + 1]);
        }
        if (k <= lcl) {                                                                      // :4191-4222
            const double thetavp = tp[k] * pow(1000.0 / p[k], rd / cp) * (1.0 + 1.608 * qstp[k] - q[mx]);
            const double thetavm = t[k] * pow(1000.0 / p[k], rd / cp) * (1.0 + 0.608 * q[k]);
            const double dqsdtp = qstp[k] * (1.0 + qstp[k] / eps1) * eps1 * rl / (rd * (tp[k] * tp[k]));
            const double dtpdt = tp[k] / (1.0 + rl / cp * (dqsdtp - qstp[k] / tp[k]))
                                 * (dtbdt / t[mx] + rl / cp * (dqbdt / tl - q[mx] / (tl * tl) * dtldt));
            dboydt = ((dtpdt / tp[k] + 1.0 / (1.0 + 1.608 * qstp[k] - q[mx]) * (1.608 * dqsdtp * dtpdt - dqbdt))
                      - (dtmdt / t[k] + 0.608 / (1.0 + 0.608 * q[k]) * dqmdt)) * grav * thetavp / thetavm;
        } else {                                                                             // :4224-4239
            const double thetavp = tp[k] * pow(1000.0 / p[k], rd / cp) * (1.0 + 0.608 * q[mx]);
            const double thetavm = t[k] * pow(1000.0 / p[k], rd / cp) * (1.0 + 0.608 * q[k]);
            dboydt = (dtbdt / t[mx] + 0.608 / (1.0 + 0.608 * q[mx]) * dqbdt - dtmdt / t[k]
                      - 0.608 / (1.0 + 0.608 * q[k]) * dqmdt) * grav * thetavp / thetavm;
        }
        s_dboy[k] = dboydt;
    }
    SYNC();
    double dadt = 0.0;                                                                       // :4244-4253
#pragma unroll SERIAL_UNROLL
    for (int k = lel; k <= mx - 1; ++k) dadt = dadt + s_dboy[k] * (A.zf[k] - A.zf[k + 1]);
    const double dltaa = -1.0 * (cape - P.capelmt);
    if (dadt != 0.0) mb = fmax(dltaa / P.tau / dadt, 0.0);
    return mb;
}

// ------------------------------------------------------------------------------------------------------
// Kernel 2a: first cloud-ensemble pass -> the test parcel's entrainment rate for the second ascent (:1032-1078)
// ------------------------------------------------------------------------------------------------------
template <int CPWV>
__global__ __launch_bounds__(WAVE) void zm_plume1_kernel(ZmDev P, int ncol, int pver, colphys_zm_convr_io io, ZmWorkspace ws)
{
    constexpr int PSTEP = WAVE / CPWV;          // lanes per column
    CHAIN_PRIO();
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (PSTEP - 1), half = CPWV == 1 ? 0 : threadIdx.x / PSTEP;
    const int msg = P.limcnv - 1;
    const int LD = pver + 2;
    PlumeLds A;
    A.carve(smem + (size_t)half * (PLUME_NARR + 2) * LD, LD);
    const int count = *ws.count;
    for (int b0 = blockIdx.x * CPWV; b0 < count; b0 += gridDim.x * CPWV) {      // persistent waves over the triggered list
    const bool valid = b0 + half < count;
    const int col = ws.list[valid ? b0 + half : b0];           // an idle half shadows the wave's first column
    const ColIn c = make_colin(P, io, col, pver);
    plume_load<PSTEP>(P, c, A, lane);
    PlumeIdx ix;
    cldprp_w<true, PSTEP>(P, pver, msg, A, ws.mx1[col], ws.lel1[col], c.landfrac, lane, ix);        // :1032
    double dmpdz = -1.0;                                                                     // :1046-1078
    {
        double hk = 0.0, dmmx = 0.0, dmsm = 0.0;
        const double orgc = 1.0;
#pragma unroll SERIAL_UNROLL
        for (int k = pver; k >= msg + 1; --k) {
            const double e = A.eu[k];
            if (e > 0) {
                dmmx = -fmax(-dmmx, e);
                dmsm = dmsm - e;
                hk = hk + 1.0;
            }
        }
        if (hk > 0) {
            dmsm = dmsm / hk;
            dmpdz = dmsm * orgc + dmmx * (1.0 - orgc);
        }
    }
    if (lane == 0 && valid) ws.dmpdz2[col] = dmpdz;
    SYNC();
    }
}

// ------------------------------------------------------------------------------------------------------
// Kernel 2b: second dilute ascent of the listed columns with that rate, re-trigger (:1080-1107)
// ------------------------------------------------------------------------------------------------------
// (register budget of three waves per SIMD, 168 VGPRs: beside the optics -- three waves of 168 -- a wave of this kernel
// then fits as soon as ONE optics block retires from its SIMD; at 180 it needed two)
template <int LPC>
__global__ __launch_bounds__(2 * WAVE, CHAIN_WAVES_PER_SIMD) void zm_ascent2_kernel(ZmDev P, int ncol, int pver, colphys_zm_convr_io io, ZmWorkspace ws)
{
    constexpr int CPW = WAVE / LPC;             // listed columns per wave pair
    extern __shared__ double smem[];            // as zm_trigger_kernel
    const int lane = threadIdx.x & (WAVE - 1), role = threadIdx.x >> 6;
    PIPE_PRIO(role);
    const int count = *ws.count;
    if ((int)blockIdx.x * CPW >= count) return;                   // block-uniform
    const int gid = blockIdx.x * CPW + lane / LPC;
    const bool valid = gid < count;
    const bool first = (lane & (LPC - 1)) == 0;
    const int col = ws.list[valid ? gid : count - 1];             // lanes past the list shadow its last column
    const size_t L = (size_t)pver;
    const ColIn c = make_colin(P, io, col, pver);
    const double pblt = pbl_top_index(c);
    BuoyRes r;
    const int rows = pver + 2 - (P.limcnv - 1);
    buoyan_dilute_2w<true, true, LPC>(P, c, pblt, ws.dmpdz2[col], ws.cin1[col], role, lane, smem + lane - (P.limcnv - 1) * WAVE, smem + rows * WAVE,
                                      ws.tp2 + col * L - 1, ws.qstp2 + col * L - 1, valid && first, r);
    if (role == 0 || !valid || !first) return;
    io.cape[col] = r.cape;
    io.status[col] |= r.status;
    ws.cape2[col] = r.cape; ws.tl2[col] = r.tl; ws.lcl2[col] = r.lcl; ws.lel2[col] = r.lel; ws.mx2[col] = r.mx;
    ws.retrig[col] = (r.cape > P.capelmt && r.cin < r.cape * P.cin_threshd) ? 1 : 0;        // :1099-1107
}

// ------------------------------------------------------------------------------------------------------
// Kernel 2c: second ensemble pass, closure, tendencies, precipitation integrals of the re-triggered columns
// ------------------------------------------------------------------------------------------------------
template <int CPWV>
#ifndef PLUME_WAVES_PER_SIMD
#define PLUME_WAVES_PER_SIMD CHAIN_WAVES_PER_SIMD
#endif
__global__ __launch_bounds__(WAVE, PLUME_WAVES_PER_SIMD) void zm_plume2_kernel(ZmDev P, int ncol, int pver, double delt, colphys_zm_convr_io io,
                                                        ZmWorkspace ws)
{
    constexpr int PSTEP = WAVE / CPWV;          // lanes per column
    CHAIN_PRIO();
    extern __shared__ double smem[];
    const int lane = threadIdx.x & (PSTEP - 1), half = CPWV == 1 ? 0 : threadIdx.x / PSTEP;
    const int msg = P.limcnv - 1;
    const int LD = pver + 2;
    const size_t L = (size_t)pver, LP = (size_t)pver + 1;
    PlumeLds A;
    double *base = smem + (size_t)half * (PLUME_NARR + 2) * LD;
    A.carve(base, LD);
    double *s_w1 = base + (size_t)PLUME_NARR * LD, *s_w2 = s_w1 + LD;       // dboydt / dqdt, dl
    const int count = *ws.count;
    for (int b0 = blockIdx.x * CPWV; b0 < count; b0 += gridDim.x * CPWV) {      // persistent waves over the triggered list
    int col = ws.list[b0 + half < count ? b0 + half : b0];
    const bool valid = b0 + half < count && ws.retrig[col];
    if (CPWV == 1) {
        if (!valid) continue;
    } else {
        if (!__any(valid)) continue;
        const int other = __shfl_xor(col, PSTEP);              // a half without a column of its own shadows the other half's
        if (!valid) col = other;
    }
    const ColIn c = make_colin(P, io, col, pver);
    plume_load<PSTEP>(P, c, A, lane);

    const int maxg = ws.mx2[col], lelg = ws.lel2[col], lcl = ws.lcl2[col];                   // :1156-1174
    const double cape = ws.cape2[col], tl = ws.tl2[col], dmpdz = ws.dmpdz2[col];
    double dsubcld = 0.0;
#pragma unroll SERIAL_UNROLL
    for (int k = msg + 1; k <= pver; ++k)
        if (k >= maxg) dsubcld = dsubcld + A.dp[k];

    PlumeIdx ix;
    cldprp_w<false, PSTEP>(P, pver, msg, A, maxg, lelg, c.landfrac, lane, ix);               // :1216
    const int jt = ix.jt, mx = maxg;
    // the second ascent's parcel profiles, for the closure: into the slots cldprp's updraft phase has left (PlumeLds)
    PAR(k, 1, pver) { A.tp[k] = ws.tp2[col * L + k - 1]; A.qstp[k] = ws.qstp2[col * L + k - 1]; }

    PAR(k, msg + 1, pver) {                                                                  // :1252-1262
        const double dzk = (A.zf[k] - A.zf[k + 1]), dpk = A.dp[k];
        A.du[k] = A.du[k] * dzk / dpk;   A.eu[k] = A.eu[k] * dzk / dpk;   A.ed[k] = A.ed[k] * dzk / dpk;
        A.cu[k] = A.cu[k] * dzk / dpk;   A.cmeg[k] = A.cmeg[k] * dzk / dpk; A.rprd[k] = A.rprd[k] * dzk / dpk;
        A.evp[k] = A.evp[k] * dzk / dpk;
    }
    SYNC();
    double mb = closure_w<PSTEP>(P, pver, msg, A, s_w1, dsubcld, cape, tl, lcl, lelg, jt, mx, lane);         // :1273
    double mumax = 0;                                                                        // :1285-1300
#pragma unroll SERIAL_UNROLL
    for (int k = msg + 2; k <= pver; ++k) mumax = fmax(mumax, A.mu[k] / A.dp[k]);
    if (mumax > 0.0) mb = fmin(mb, 0.5 / (delt * mumax));
    else mb = 0.0;
    if (P.no_deep_pbl) {                                                                     // :1304-1308
        if (c.zm[jt] < c.pblh) mb = 0;
    }
    SYNC();
    PAR(k, msg + 1, pver) {                                                                  // :1319-1331
        A.mu[k] = A.mu[k] * mb; A.md[k] = A.md[k] * mb; A.mc[k] = A.mc[k] * mb; A.du[k] = A.du[k] * mb;
        A.eu[k] = A.eu[k] * mb; A.ed[k] = A.ed[k] * mb; A.cmeg[k] = A.cmeg[k] * mb; A.rprd[k] = A.rprd[k] * mb;
        A.cu[k] = A.cu[k] * mb; A.evp[k] = A.evp[k] * mb;
        A.pflx[k + 1] = A.pflx[k + 1] * mb * 100.0 / P.grav;
    }
    SYNC();

    // q1q2_pjr :4262-4421: the flux-form tendencies are element-wise; levels below the launch level copy it
    double *dqdt = s_w1, *dlg = s_w2;
    const double rl = P.rl, cp = P.cp;
    double *o_qtnd = io.qtnd + col * L - 1, *o_heat = io.heat + col * L - 1, *o_cme = io.cme + col * L - 1,
           *o_zdu = io.zdu + col * L - 1, *o_rprd = io.rprd + col * L - 1, *o_dlf = io.dlf + col * L - 1,
           *o_ql = io.ql + col * L - 1, *o_eurt = io.eurt + col * L - 1, *o_mcon = io.mcon + col * LP - 1,
           *o_pflx = io.pflx + col * LP - 1;
    double dsdt_mx = 0.0, dqdt_mx = 0.0;
    if (mx >= msg + 1 && mx <= pver) {                                                       // :4402-4412
        dsdt_mx = (1.0 / dsubcld) * (-A.mu[mx] * (A.su[mx] - A.shat[mx]) - A.md[mx] * (A.sd[mx] - A.shat[mx]));
        dqdt_mx = (1.0 / dsubcld) * (-A.mu[mx] * (A.qu[mx] - A.qhat[mx]) - A.md[mx] * (A.qd[mx] - A.qhat[mx]));
    }
    PAR(k, msg + 1, pver) {
        double dsdt = 0.0, dq = 0.0, dl = 0.0;
        if (k >= jt && k <= pver - 1) {                                                      // :4364-4399
            const double emc = -A.cu[k] + A.evp[k];
            dsdt = -rl / cp * emc + (A.mu[k + 1] * (A.su[k + 1] - A.shat[k + 1]) - A.mu[k] * (A.su[k] - A.shat[k])
                                     + A.md[k + 1] * (A.sd[k + 1] - A.shat[k + 1]) - A.md[k] * (A.sd[k] - A.shat[k])) / A.dp[k];
            dq = emc + (A.mu[k + 1] * (A.qu[k + 1] - A.qhat[k + 1]) - A.mu[k] * (A.qu[k] - A.qhat[k])
                        + A.md[k + 1] * (A.qd[k + 1] - A.qhat[k + 1]) - A.md[k] * (A.qd[k] - A.qhat[k])) / A.dp[k];
            dl = A.du[k] * A.qcde[k + 1];
        }
        if (k >= mx) { dsdt = dsdt_mx; dq = dqdt_mx; }                                       // :4404-4416
        dqdt[k] = dq; dlg[k] = dl;
        if (valid) {
            o_qtnd[k] = dq;                                                                  // :1495-1511
            o_cme[k] = A.cmeg[k];
            o_rprd[k] = A.rprd[k];
            o_zdu[k] = A.du[k];
            o_mcon[k] = A.mc[k];
            o_heat[k] = dsdt * P.cpres;
            o_dlf[k] = dl;
            o_pflx[k] = A.pflx[k];
            o_ql[k] = A.ql[k];
            o_eurt[k] = -dmpdz;                                                              // :1233-1237
        }
    }
    // full-layout plume profiles; zm_gather_kernel compacts them per chunk
    PAR(k, 1, pver) {
        const size_t g = col * L + k - 1;
        if (valid) {
            ws.mu_f[g] = A.mu[k]; ws.md_f[g] = A.md[k]; ws.du_f[g] = A.du[k]; ws.eu_f[g] = A.eu[k]; ws.ed_f[g] = A.ed[k];
            ws.dp_f[g] = A.dp[k];
        }
    }
    SYNC();
    double prec = 0.0, rliq = 0.0;
#pragma unroll SERIAL_UNROLL
    for (int k = pver; k >= msg + 1; --k) {                                                  // :1629-1633 (dif == 0)
        const double qk = A.q[k], dppk = c.dpp[k];
        const double qnew = qk + 2.0 * delt * dqdt[k];
        prec = prec - dppk * (qnew - qk) - dppk * (dlg[k] + 0.0) * 2.0 * delt;
    }
    prec = P.rgrav * fmax(prec, 0.0) / (2.0 * delt) / 1000.0;                                // :1637
#pragma unroll SERIAL_UNROLL
    for (int k = 1; k <= pver; ++k) {                                                        // :1642-1649
        const double dlfk = (k >= msg + 1) ? dlg[k] : 0.0;
        rliq = rliq + (dlfk + 0.0) * c.dpp[k] / P.gravit;
    }
    if (lane == 0 && valid) {
        io.jctop[col] = jt;                                                                  // :1616-1620
        io.jcbot[col] = maxg;
        o_pflx[pver + 1] = A.pflx[pver + 1];
        io.prec[col] = prec;
        io.rliq[col] = rliq / 1000.0;
        io.rice[col] = 0.0 / 1000.0;
        ws.dsub_f[col] = dsubcld;
        ws.jt_f[col] = jt;
        ws.maxg_f[col] = maxg;
        ws.deep[col] = 1;
    }
    SYNC();
    }
}
#undef PAR
#undef SYNC

// ------------------------------------------------------------------------------------------------------
// Kernel 3: per-chunk ascending compaction into the reference's gathered layout
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVE) void zm_gather_kernel(int ncol, int pver, int pcols, colphys_zm_convr_io io, ZmWorkspace ws,
                                                        ZmTendDiag dg)
{
    CHAIN_PRIO();
    __shared__ int s_src[1024];
    const int c0 = blockIdx.x * pcols;
    const int nc = min(pcols, ncol - c0);
    // ascending compaction by ballot: the block is one wave (round 3: lane 0 used to walk the chunk's columns one dependent
    // load after the other -- 20 us of a latency-bound chain)
    int n = 0;                                          // wave-uniform running count
    for (int base = 0; base < nc; base += WAVE) {
        const int i = base + (int)threadIdx.x;
        const bool d = i < nc && ws.deep[c0 + i] != 0;
        const unsigned long long m = __ballot(d);
        if (d) s_src[n + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = i;
        n += __popcll(m);
    }
    if (threadIdx.x == 0) io.lengath[blockIdx.x] = n;
    __syncthreads();
    const size_t L = (size_t)pver;
    for (int idx = threadIdx.x; idx < n * pver; idx += blockDim.x) {
        const int j = idx / pver, k = idx - j * pver;
        const size_t src = (size_t)(c0 + s_src[j]) * L + k, dst = (size_t)(c0 + j) * L + k;
        io.mu[dst] = ws.mu_f[src]; io.md[dst] = ws.md_f[src]; io.du[dst] = ws.du_f[src];
        io.eu[dst] = ws.eu_f[src]; io.ed[dst] = ws.ed_f[src]; io.dp[dst] = ws.dp_f[src];
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const int src = c0 + s_src[j];
        io.dsubcld[c0 + j] = ws.dsub_f[src];
        io.jt[c0 + j] = ws.jt_f[src];
        io.maxg[c0 + j] = ws.maxg_f[src];
        io.ideep[c0 + j] = s_src[j] + 1;
    }
    if (!dg.enabled) return;
    // zm_conv_tend's post-processing (zm_conv_intr.F90:685-731) for the chunk's columns.  The scatters through ideep are
    // reads of the full-layout copies here: a deep column's own mu / md / jt / maxg.
    const size_t LP = L + 1;
    for (size_t i = threadIdx.x; i < (size_t)nc * LP; i += blockDim.x)
        dg.mconzm[(size_t)c0 * LP + i] = io.mcon[(size_t)c0 * LP + i] * 100.0 / dg.gravit;          // :691-692
    if (dg.mu_out || dg.md_out) {
        for (size_t i = threadIdx.x; i < (size_t)nc * L; i += blockDim.x) {                          // :575-576, 700-706
            const size_t g = (size_t)c0 * L + i;
            const bool deep = ws.deep[c0 + (int)(i / L)] != 0;
            if (dg.mu_out) dg.mu_out[g] = deep ? ws.mu_f[g] * 100.0 / dg.gravit : 0.0;
            if (dg.md_out) dg.md_out[g] = deep ? ws.md_f[g] * 100.0 / dg.gravit : 0.0;
        }
    }
    for (int i = threadIdx.x; i < nc; i += blockDim.x) {
        const int col = c0 + i;
        const bool deep = ws.deep[col] != 0;
        if (dg.freqzm) dg.freqzm[col] = deep ? 1.0 : 0.0;                                            // :685-688
        double pt = dg.ps[col], pb = dg.ps[col];                                                     // :722-729
        if (deep && ws.maxg_f[col] > ws.jt_f[col]) {
            pt = dg.pmid[(size_t)col * L + (ws.jt_f[col] - 1)];
            pb = dg.pmid[(size_t)col * L + (ws.maxg_f[col] - 1)];
        }
        if (dg.pcont) dg.pcont[col] = pt;
        if (dg.pconb) dg.pconb[col] = pb;
    }
}

}  // namespace

int launch_zm_convr(const ZmDev &P, int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io &io,
                    const ZmWorkspace &ws, hipStream_t st, hipEvent_t *ev, const ZmTendDiag *diag)
{
    if (pcols > 1024) return COLPHYS_ERR_ARG;
    const int nblk = (ncol + WAVE - 1) / WAVE;
    const size_t lds = (size_t)(pver + 2 - (P.limcnv - 1)) * WAVE * sizeof(double);   // parcel profile, levels above limcnv only
    const size_t lds2 = lds + MAIL_DOUBLES * sizeof(double);       // second ascent: two-wave pipeline with its mailbox
    const int nchunks = (ncol + pcols - 1) / pcols;
    if (hipMemsetAsync(ws.count, 0, sizeof(int), st) != hipSuccess) return COLPHYS_ERR_HIP;
    if (ev) hipEventRecord(ev[0], st);
    // trigger: wave pairs while the column groups alone leave SIMDs idle (four groups per block so that a pair shares a
    // SIMD; fewer when the LDS profile is long -- two pairs per block were still 9 % better than one wave per group in
    // the overlapped substep at L70, before the profile was trimmed to the levels above limcnv and four fitted);
    // one wave per group doing both sweeps once the groups fill the SIMDs on their own (measured with the optics
    // running beside it: 48 602 columns 2.93 vs 3.01 ms, 97 202 columns 5.21 vs 5.26, 194 402 columns 10.24 vs 10.02,
    // 777 602 columns 41.7 vs 39.0).  COLPHYS_TRIG_MODE = 1 / 2 forces paired / fused (experiments).
    // SIMD count of the CURRENT device (cached per device index: the engine may be re-initialised on another one)
    static int simd_of_dev[64] = {0};
    int n_simd = 1024;
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
            if (!simd_of_dev[dev]) {
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                simd_of_dev[dev] = 4 * cus;
            }
            n_simd = simd_of_dev[dev];
        }
    }
    bool paired = nblk <= n_simd;
    if (const char *e = getenv("COLPHYS_TRIG_MODE")) {
        if (atoi(e) == 1) paired = true;
        if (atoi(e) == 2) paired = false;
    }
    // lane pairs (two lanes per column, colphys_common.h): the second ascent always (it covers the ~10 % triggered
    // columns, the SIMDs are empty: 0.57 -> 0.49 ms); the trigger only while its twice-as-many waves still find a SIMD each
    // (up to 16 384 columns: 0.58 -> 0.42 ms, a chunk-sized call 1.06 -> 0.96 ms, a 6 075-column share of a strong-scaled grid
    // 1.28 -> 1.12 ms per ZM substep; at 20 000 columns it is already 0.64 against 0.58, at 48 602 1.03 against 0.76).
    // COLPHYS_LPC_TRIG / COLPHYS_LPC_ASC = 1 / 2 force the choice (experiments).
    int lpc_trig_env = 0, lpc_asc_env = 0;             // read per call: tests switch them between calls
    if (const char *e = getenv("COLPHYS_LPC_TRIG")) lpc_trig_env = atoi(e);
    if (const char *e = getenv("COLPHYS_LPC_ASC")) lpc_asc_env = atoi(e);
    const int nblk32 = (ncol + 31) / 32;
    // ... unless its lane pairs still leave SIMDs idle: a chunk-sized call, or a rank's share of a strong-scaled grid
    // lane quads (four lanes per column: the two bracket evaluations of every inversion side by side) while even four times the
    // waves find a SIMD each: the trigger up to 8 192 columns, the second ascent -- ~20 % of the columns -- up to 24 000
    // (round 3; a 6 075-column share: trigger 0.42 -> , second ascent 0.42 -> ms)
    const int nblk16 = (ncol + 15) / 16;
    int lpc_trig = (2 * nblk16 <= n_simd) ? 4 : ((2 * nblk32 <= n_simd) ? 2 : 1);
    if (lpc_trig_env == 1 || lpc_trig_env == 2 || lpc_trig_env == 4) lpc_trig = lpc_trig_env;
    int lpc_asc = ncol <= 24000 ? 4 : 2;
    if (lpc_asc_env == 1 || lpc_asc_env == 2 || lpc_asc_env == 4) lpc_asc = lpc_asc_env;
    int groups = TRIG_GROUPS_MAX;
    if (lpc_trig >= 2) groups = 2;              // blocks of two pairs: three of them share a CU at three waves per SIMD
    // one lane per column and at most 1 024 waves (16 385 .. 32 768 columns): blocks of two pairs [A0 A1 B0 B1] put every wave on
    // a SIMD of its own (a block's waves go round the four SIMDs) instead of pairing A and B on one -- the pairing is for the
    // sizes where two waves per SIMD are unavoidable.  Trigger 0.59 -> 0.49 ms at 18 000 / 24 301 columns (substep 1.40 -> 1.27,
    // 1.555 -> 1.54 ms), 1.99 -> 1.87 ms at 32 768; beyond, two-pair blocks lose (40 000 columns: 2.25 -> 2.40 ms).
    else if (2 * nblk <= n_simd) groups = 2;
    if (const char *e = getenv("COLPHYS_TRIG_GROUPS")) { const int g = atoi(e); if (g == 1 || g == 2 || g == 4) groups = g; }   // tests / A-B
    while (groups > 1 && lds2 * groups > 158 * 1024) groups /= 2;
    if (lds2 * groups > 160 * 1024) paired = false;
    if (paired) {
        const int cpb = groups * (WAVE / lpc_trig);
        const int ntrig = (ncol + cpb - 1) / cpb;
        const size_t ldst = lds2 * groups;
        const void *fn = lpc_trig == 4 ? (const void *)zm_trigger_kernel<4> : (lpc_trig == 2 ? (const void *)zm_trigger_kernel<2> : (const void *)zm_trigger_kernel<1>);
        if (ldst > 64 * 1024) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldst);
        if (lpc_trig == 4) hipLaunchKernelGGL(zm_trigger_kernel<4>, dim3(ntrig), dim3(2 * groups * WAVE), ldst, st, P, ncol, pver, groups, io, ws);
        else if (lpc_trig == 2) hipLaunchKernelGGL(zm_trigger_kernel<2>, dim3(ntrig), dim3(2 * groups * WAVE), ldst, st, P, ncol, pver, groups, io, ws);
        else hipLaunchKernelGGL(zm_trigger_kernel<1>, dim3(ntrig), dim3(2 * groups * WAVE), ldst, st, P, ncol, pver, groups, io, ws);
    } else {
        if (lds > 160 * 1024) return COLPHYS_ERR_UNSUPPORTED;
        if (lds > 64 * 1024) hipFuncSetAttribute((const void *)zm_trigger_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(zm_trigger_fused_kernel, dim3(nblk), dim3(WAVE), lds, st, P, ncol, pver, io, ws);
    }
    if (ev) { hipEventRecord(ev[1], st); hipEventRecord(ev[2], st); }
    // plume kernels: one column per wave.  Two columns per wave (pver <= 32; COLPHYS_PLUME_CPW=2, kept for tests / A-B) paid
    // as throughput at ne120 while a wave's 49-profile footprint allowed 11 waves per CU (16.75 -> 16.03 ms in round 2); with
    // the profiles on 36 slots 12 single-column waves are resident (the VGPR budget's limit), 7 pairs, and a pair still takes as
    // long as its slower member plus what the two disagree on: ne120 plume2 3.71 ms single, 4.14 paired; 48 602 columns 0.28 / 0.32.
    int cpwv = 1;
    if (const char *e = getenv("COLPHYS_PLUME_CPW")) cpwv = (atoi(e) == 2 && pver <= WAVE / 2) ? 2 : 1;
    const size_t lds_w = (size_t)cpwv * (PLUME_NARR + 2) * (pver + 2) * sizeof(double);
    // Persistent waves over the device-side list: a multiple of what is RESIDENT at once (LDS: 11 waves per CU at L32, 5 with two
    // columns per wave or at L70).  Round 3: the grid used to be 12 waves per CU whatever fitted; the waves beyond the resident
    // set then ran as a second, nearly empty round -- ne120 plume2 5.83 ms with 12 per CU, 7.3 with 6, 5.15-5.28 with 5 or 10;
    // 48 602 columns 0.333 -> 0.295 ms.  Twice the resident set: finer grain beside the optics' blocks (substep 2.705 -> 2.68 ms).
    static int occ_cache[2][COLPHYS_MAX_PVER + 1] = {};
    int &occ = occ_cache[cpwv - 1][pver];
    if (occ == 0) {
        int nb = 0;
        const void *fn = cpwv == 2 ? (const void *)zm_plume2_kernel<2> : (const void *)zm_plume2_kernel<1>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, WAVE, lds_w) != hipSuccess || nb < 1) nb = 6;
        occ = nb;
        if (getenv("COLPHYS_DEBUG_OCC")) fprintf(stderr, "plume kernels: %d resident waves per CU (cpw %d, pver %d, %zu B of LDS each)\n", nb, cpwv, pver, lds_w);
    }
    int wpc = 2 * occ;
    if (const char *e = getenv("COLPHYS_PLUME_WPC")) wpc = atoi(e) > 0 ? atoi(e) : wpc;      // tests / A-B
    const int n_cu = n_simd / 4;
    const int nper = ncol < n_cu * wpc ? ncol : n_cu * wpc;
    if (cpwv == 2) hipLaunchKernelGGL(zm_plume1_kernel<2>, dim3(nper), dim3(WAVE), lds_w, st, P, ncol, pver, io, ws);
    else hipLaunchKernelGGL(zm_plume1_kernel<1>, dim3(nper), dim3(WAVE), lds_w, st, P, ncol, pver, io, ws);
    if (ev) { hipEventRecord(ev[3], st); hipEventRecord(ev[4], st); }
    if (lpc_asc == 4) {
        if (lds2 > 64 * 1024) hipFuncSetAttribute((const void *)zm_ascent2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(zm_ascent2_kernel<4>, dim3(nblk16), dim3(2 * WAVE), lds2, st, P, ncol, pver, io, ws);
    } else if (lpc_asc == 2) {
        if (lds2 > 64 * 1024) hipFuncSetAttribute((const void *)zm_ascent2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(zm_ascent2_kernel<2>, dim3(nblk32), dim3(2 * WAVE), lds2, st, P, ncol, pver, io, ws);
    } else {
        if (lds2 > 64 * 1024) hipFuncSetAttribute((const void *)zm_ascent2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(zm_ascent2_kernel<1>, dim3(nblk), dim3(2 * WAVE), lds2, st, P, ncol, pver, io, ws);
    }
    if (ev) { hipEventRecord(ev[5], st); hipEventRecord(ev[6], st); }
    if (cpwv == 2) hipLaunchKernelGGL(zm_plume2_kernel<2>, dim3(nper), dim3(WAVE), lds_w, st, P, ncol, pver, delt, io, ws);
    else hipLaunchKernelGGL(zm_plume2_kernel<1>, dim3(nper), dim3(WAVE), lds_w, st, P, ncol, pver, delt, io, ws);
    ZmTendDiag dg{};
    if (diag) dg = *diag;
    hipLaunchKernelGGL(zm_gather_kernel, dim3(nchunks), dim3(WAVE), 0, st, ncol, pver, pcols, io, ws, dg);
    if (ev) hipEventRecord(ev[7], st);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}
