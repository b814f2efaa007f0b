// k_scans.hip -- the three bandwidth-bound vertical scans of the path, for gfx950:
//
//   geopotential_t   /root/reference/physics/geopotential.F90:153-310   bottom-up hydrostatic integration
//   geopotential_dse /root/reference/physics/geopotential.F90:29-150    the same from dry static energy (T is an output)
//   zm_conv_evap     /root/reference/physics/zm_conv.F90:1712-1972      top-down precipitation flux / evaporation
//   convtran         /root/reference/physics/zm_conv.F90:1976-2311      tracer transport on the gathered columns
//   momtran          /root/reference/physics/zm_conv.F90:2315-2715      momentum transport on the gathered columns
//
// HBM layout is column-contiguous [ncol][pver]; the scans are serial in k and independent across columns, so
// the mapping is one lane per column with the block's columns staged through LDS:
//   - the block's slab of a field ([TILE columns][nlev], contiguous in HBM) is read with fully coalesced
//     loads by all lanes and written to LDS transposed-by-stride (row stride odd => ds_read_b64 of one level
//     by 32 consecutive columns is bank-conflict free);
//   - each lane then walks its own column in LDS, keeping the reference's operation order exactly
//     (no tree scans: the recurrences are evaluated serially per column, so results do not depend on the
//     mapping);
//   - outputs go back through the same LDS tiles and leave with coalesced stores.
// Algorithmic bytes per column (fp64): DESIGN.md "Kernels".
#include "colphys_common.h"

namespace {

#ifndef EVAP_TILE
#define EVAP_TILE 16     // 34 KB of LDS per block at L32: fits the room ONE retiring optics block leaves (A/B: 32 -> 16 columns, substep 2.66 -> 2.58 ms)
#endif
constexpr int ETILE = EVAP_TILE;   // zm_conv_evap: columns per block (8 LDS tiles of [ETILE][pver])
constexpr int NT = 64;        // threads per block (one wave): all lanes move data, lanes < TILE scan

__device__ __forceinline__ int ldk_of(int nlev) { return (nlev & 1) ? nlev : nlev + 1; }

// global [nc][nlev] contiguous slab -> LDS tile s[c*ldk + k]
__device__ __forceinline__ void tile_in(double *s, const double *g, int nc, int nlev, int ldk)
{
    const int n = nc * nlev;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = i / nlev, k = i - c * nlev;
        s[c * ldk + k] = g[i];
    }
}
// NF tiles of the same shape in ONE pass: the NF loads of a cell are in flight together (NF separate tile_in loops are NF
// times as many dependent global round trips: each loop's LDS stores wait for its own loads)
template <int NF>
__device__ __forceinline__ void tile_in_n(double *const (&s)[NF], const double *const (&g)[NF], int nc, int nlev, int ldk)
{
    const int n = nc * nlev;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = i / nlev, k = i - c * nlev;
        double v[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f] = g[f][i];
#pragma unroll
        for (int f = 0; f < NF; ++f) s[f][c * ldk + k] = v[f];
    }
}
__device__ __forceinline__ void tile_out(double *g, const double *s, int nc, int nlev, int ldk)
{
    const int n = nc * nlev;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int c = i / nlev, k = i - c * nlev;
        g[i] = s[c * ldk + k];
    }
}

// ------------------------------------------------------------------------------------------------------
// geopotential_t
//   Phase 1 (element-wise, every lane, flat index over the block's [GT columns][pver] slab => every HBM
//            access coalesced): the two per-level increments  dzi = rog*tv*hkl,  dzm = rog*tv*hkk
//            (geopotential.F90:222-245 / :282-305) go to two LDS tiles.
//   Phase 2 (one lane per column): the bottom-up sums zi(k) = zi(k+1) + dzi(k), zm(k) = zi(k+1) + dzm(k), in the
//            reference's order, in place in LDS (odd row stride: conflict-free).
//   Phase 3: coalesced store of the two tiles.
//   LDS = 2 tiles: 34 KB for 64 columns x L32 => 4 blocks per CU; all 64 lanes scan.
// ------------------------------------------------------------------------------------------------------
#ifndef GEO_TILE
#define GEO_TILE 32      // 17 KB of LDS per block at L32 (A/B 64 -> 32 columns: 0.031 -> 0.028 ms alone, substep -0.5 %)
#endif
constexpr int GT = GEO_TILE;  // columns per block
#ifndef GEO_THREADS
#define GEO_THREADS 128
#endif
constexpr int GNT = GEO_THREADS;   // threads per block: two waves move data (8 flat-index steps of a 32-column tile at L32), 32 lanes scan

// body shared by geopotential_kernel and the fused tail launch of a small rank (zm_tail3_kernel): block `bid` of `nt` threads
__device__ __forceinline__ void geopotential_body(int bid, int tid, int nt, double *smem, int ncol, int pver, int dycore_kind, int nspecies,
                                                  const double *__restrict__ piln, const double *__restrict__ pint,
                                                  const double *__restrict__ pmid, const double *__restrict__ pdel,
                                                  const double *__restrict__ rpdel, const double *__restrict__ t,
                                                  const double *__restrict__ q, const double *__restrict__ rair,
                                                  double gravit, const double *__restrict__ zvir,
                                                  double *__restrict__ zi, double *__restrict__ zm, size_t rzs)
{
    // rzs: stride of rair / zvir, 1 = fields [ncol][pver] (rairv, zvirv of physics_types.F90:393-415), 0 = ONE value each --
    // what they hold without WACCM-X; the two reads then cost no HBM traffic (2 of the FV branch's 9 streams)
    const int col0 = bid * GT;
    const int nc = min(GT, ncol - col0);
    const int pverp = pver + 1;
    const int ldp = ldk_of(pverp);
    const size_t L = (size_t)pver, LP = (size_t)pverp, F = (size_t)ncol * L;
    double *sZI = smem;                   // [GT][ldp]: dzi(k) at [k], then zi(k); zi(pver+1) = 0 at [pver]
    double *sZM = smem + GT * ldp;        // [GT][ldp]: dzm(k), then zm(k)

    const int n = nc * pver;
    for (int i = tid; i < n; i += nt) {
        const int c = i / pver, k = i - c * pver;              // 0-based level
        const size_t g = col0 * L + i;                         // flat index, mid-level fields
        const size_t gi = (size_t)(col0 + c) * LP + k;         // flat index, interface fields
        double hkl, hkk, tvfac;
        if (dycore_kind == COLPHYS_DYCORE_FV) {                // :222-226
            hkl = piln[gi + 1] - piln[gi];
            hkk = 1.0 - pint[gi] * hkl * rpdel[g];
        } else {                                               // :232-235
            hkl = pdel[g] / pmid[g];
            hkk = 0.5 * hkl;
        }
        const double q1 = q[g];
        if (dycore_kind == COLPHYS_DYCORE_SE) {                // generalised virtual temperature :255-275, 301
            double qfac = 1.0, sum_dry = 1.0;
            for (int idx = 0; idx < nspecies; ++idx) qfac = qfac - q[idx * F + g];
            qfac = 1.0 / qfac;
            for (int idx = 0; idx < nspecies; ++idx) sum_dry = sum_dry + q[idx * F + g] * qfac;
            sum_dry = 1.0 / sum_dry;
            tvfac = (1.0 + (zvir[g * rzs] + 1.0) * q1 * qfac) * sum_dry;
        } else {
            tvfac = 1.0 + zvir[g * rzs] * q1;                  // :241
        }
        const double rog = rair[g * rzs] / gravit;             // :203
        const double tv = t[g] * tvfac;
        sZM[c * ldp + k] = rog * tv * hkk;                     // :244
        sZI[c * ldp + k] = rog * tv * hkl;                     // :245
    }
    __syncthreads();

    const int c = tid;
    if (c < nc) {
        double zi_kp1 = 0.0;                                   // :207-209
        sZI[c * ldp + pver] = 0.0;
#pragma unroll SCAN_UNROLL
        for (int k = pver - 1; k >= 0; --k) {
            sZM[c * ldp + k] = zi_kp1 + sZM[c * ldp + k];
            zi_kp1 = zi_kp1 + sZI[c * ldp + k];
            sZI[c * ldp + k] = zi_kp1;
        }
    }
    __syncthreads();
    for (int i = tid; i < nc * pverp; i += nt) {
        const int cc = i / pverp, k = i - cc * pverp;
        zi[col0 * LP + i] = sZI[cc * ldp + k];
    }
    for (int i = tid; i < n; i += nt) {
        const int cc = i / pver, k = i - cc * pver;
        zm[col0 * L + i] = sZM[cc * ldp + k];
    }
}

__global__ __launch_bounds__(GNT) void geopotential_kernel(int ncol, int pver, int dycore_kind, int nspecies,
                                                         const double *__restrict__ piln, const double *__restrict__ pint,
                                                         const double *__restrict__ pmid, const double *__restrict__ pdel,
                                                         const double *__restrict__ rpdel, const double *__restrict__ t,
                                                         const double *__restrict__ q, const double *__restrict__ rair,
                                                         double gravit, const double *__restrict__ zvir,
                                                         double *__restrict__ zi, double *__restrict__ zm, size_t rzs)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    geopotential_body(blockIdx.x, threadIdx.x, GNT, smem, ncol, pver, dycore_kind, nspecies, piln, pint, pmid, pdel, rpdel, t, q, rair, gravit,
                      zvir, zi, zm, rzs);
}

// ------------------------------------------------------------------------------------------------------
// geopotential_dse (geopotential.F90:29-150): temperature and geopotential height from dry static energy.
//   Same tiling as geopotential_t.  The recurrence carries a division here,
//       tv = (dse - phis - gravit*zi(k+1)) / (cpair/tvfac + rair*hkk),   zi(k) = zi(k+1) + rog*tv*hkl,
//   so phase 1 puts everything that does not depend on zi(k+1) into LDS tiles (numerator part dse - phis, denominator,
//   rog, hkk, hkl), phase 2 walks the column, phase 3 turns tv into t = tv/tvfac element-wise (tvfac recomputed from the
//   same expression) and stores t, zi, zm coalesced.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double geo_tvfac(int dycore_kind, int nspecies, size_t g, size_t F, const double *__restrict__ q,
                                            const double *__restrict__ zvir)
{
    const double q1 = q[g];
    if (dycore_kind == COLPHYS_DYCORE_SE) {                    // :118-133, 140
        double qfac = 1.0, sum_dry = 1.0;
        for (int idx = 0; idx < nspecies; ++idx) qfac = qfac - q[idx * F + g];
        qfac = 1.0 / qfac;
        for (int idx = 0; idx < nspecies; ++idx) sum_dry = sum_dry + q[idx * F + g] * qfac;
        sum_dry = 1.0 / sum_dry;
        return (1.0 + (zvir[g] + 1.0) * q1 * qfac) * sum_dry;
    }
    return 1.0 + zvir[g] * q1;                                 // :107
}

__global__ __launch_bounds__(GNT) void geopotential_dse_kernel(int ncol, int pver, int dycore_kind, int nspecies,
                                                             const double *__restrict__ piln, const double *__restrict__ pint,
                                                             const double *__restrict__ pmid, const double *__restrict__ pdel,
                                                             const double *__restrict__ rpdel, const double *__restrict__ dse,
                                                             const double *__restrict__ q, const double *__restrict__ phis,
                                                             const double *__restrict__ rair, double gravit,
                                                             const double *__restrict__ cpair, const double *__restrict__ zvir,
                                                             double *__restrict__ t, double *__restrict__ zi, double *__restrict__ zm)
{
    extern __shared__ double smem[];
    const int col0 = blockIdx.x * GT;
    const int nc = min(GT, ncol - col0);
    const int pverp = pver + 1;
    const int ldp = ldk_of(pverp);
    const size_t L = (size_t)pver, LP = (size_t)pverp, F = (size_t)ncol * L;
    double *sZI = smem;                   // [GT][ldp]: hkl(k), then zi(k); zi(pver+1) = 0 at [pver]
    double *sZM = sZI + GT * ldp;         // hkk(k), then zm(k)
    double *sA = sZM + GT * ldp;          // dse - phis, then tv
    double *sD = sA + GT * ldp;           // cpair/tvfac + rair*hkk
    double *sR = sD + GT * ldp;           // rog

    const int n = nc * pver;
    for (int i = threadIdx.x; i < n; i += GNT) {
        const int c = i / pver, k = i - c * pver;              // 0-based level
        const size_t g = col0 * L + i, gi = (size_t)(col0 + c) * LP + k;
        double hkl, hkk;
        if (dycore_kind == COLPHYS_DYCORE_FV) {                // :93-97
            hkl = piln[gi + 1] - piln[gi];
            hkk = 1.0 - pint[gi] * hkl * rpdel[g];
        } else {                                               // :99-102, 135-138
            hkl = pdel[g] / pmid[g];
            hkk = 0.5 * hkl;
        }
        const double tvfac = geo_tvfac(dycore_kind, nspecies, g, F, q, zvir);
        sZI[c * ldp + k] = hkl;
        sZM[c * ldp + k] = hkk;
        sA[c * ldp + k] = dse[g] - phis[col0 + c];             // (dse - phis) - gravit*zi : left to right, :108
        sD[c * ldp + k] = (cpair[g] / tvfac) + rair[g] * hkk;
        sR[c * ldp + k] = rair[g] / gravit;                    // :78
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < nc) {
        double zi_kp1 = 0.0;                                   // :81-83
        sZI[c * ldp + pver] = 0.0;
#pragma unroll SCAN_UNROLL
        for (int k = pver - 1; k >= 0; --k) {
            const double tv = (sA[c * ldp + k] - gravit * zi_kp1) / sD[c * ldp + k];
            const double rt = sR[c * ldp + k] * tv;
            sA[c * ldp + k] = tv;
            sZM[c * ldp + k] = zi_kp1 + rt * sZM[c * ldp + k];   // :111
            zi_kp1 = zi_kp1 + rt * sZI[c * ldp + k];             // :112
            sZI[c * ldp + k] = zi_kp1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc * pverp; i += GNT) {
        const int cc = i / pverp, k = i - cc * pverp;
        zi[col0 * LP + i] = sZI[cc * ldp + k];
    }
    for (int i = threadIdx.x; i < n; i += GNT) {
        const int cc = i / pver, k = i - cc * pver;
        const size_t g = col0 * L + i;
        zm[g] = sZM[cc * ldp + k];
        t[g] = sA[cc * ldp + k] / geo_tvfac(dycore_kind, nspecies, g, F, q, zvir);   // :110
    }
}

// ------------------------------------------------------------------------------------------------------
// zm_conv_evap (old_snow path; kemask = ke unless zm_org)
// ------------------------------------------------------------------------------------------------------
#ifndef EVAP_THREADS
#define EVAP_THREADS 256
#endif
constexpr int ENT = EVAP_THREADS;
__device__ __forceinline__ double kemask_of(const ZmDev &P, double landfrac)
{   // :1860-1866: ke, or the land/ocean blend when zm_org
    return P.zm_org ? (P.ke * (1.0 - landfrac) + P.ke_lnd * landfrac) : P.ke;
}      // threads per block: four waves move the tiles (4 flat-index steps per tile at L32), 32 lanes scan
// body shared by zm_conv_evap_kernel and zm_tail3_kernel (blocks of ENT threads either way)
__device__ __forceinline__ void zm_conv_evap_body(int bid, double *smem, double *s_estbl, const ZmDev &P, const double *__restrict__ estbl,
                                                  int ncol, int pver, double deltat, const colphys_zm_conv_evap_io &io)
{
    const int col0 = bid * ETILE;
    const int nc = min(ETILE, ncol - col0);
    const int pverp = pver + 1;
    const int ld = ldk_of(pver), ldp = ldk_of(pverp);
    const size_t L = (size_t)pver, LP = (size_t)pverp;

    double *sT = smem;                 // t        -> tend_s
    double *sPM = sT + ETILE * ld;      // pmid     -> tend_s_snwprd
    double *sPD = sPM + ETILE * ld;     // pdel     -> tend_s_snwevmlt
    double *sQ = sPD + ETILE * ld;      // q        -> tend_q
    double *sPR = sQ + ETILE * ld;      // prdprec  -> ntprprd
    double *sCL = sPR + ETILE * ld;     // cldfrc   -> ntsnprd
    double *sFP = sCL + ETILE * ld;     // flxprec [ldp]
    double *sFS = sFP + ETILE * ldp;    // flxsnow [ldp]

    for (int i = threadIdx.x; i <= COLPHYS_PLENEST; i += ENT) s_estbl[i] = estbl[i];
    {
        double *const st[6] = {sT, sPM, sPD, sQ, sPR, sCL};
        const double *const gt[6] = {io.t + col0 * L, io.pmid + col0 * L, io.pdel + col0 * L, io.q + col0 * L, io.prdprec + col0 * L,
                                     io.cldfrc + col0 * L};
        tile_in_n<6>(st, gt, nc, pver, ld);
    }
    __syncthreads();

    // Phase A, element-wise over the tile (all threads): everything of a level that does not depend on the fluxes coming
    // down from above -- saturation (1-K table), the evaporation limit's flux-free factor, the snow fraction -- so that
    // the serial top-down scan below is left with the recurrence itself.  Results replace t / q / cldfrc in their tiles:
    //   sT  <- kemask*(1-cldfrc)*max(1 - q/(1+q)/qs, 0)   (the factor of sqrt(flxprec) in :1868, same operation order)
    //   sQ  <- max(1 - q/(1+q)/qs, 0)                      (evplimit before the flux limiters, :1854)
    //   sCL <- fsnow_conv, or -1 where t > tmelt            (cldfrc_fice :1809 and the melt test :1821)
    {
        const double tmelt = P.tmelt;
        for (int i = threadIdx.x; i < nc * pver; i += ENT) {
            const int cc = i / pver, k = i - cc * pver;
            const double tk = sT[cc * ld + k], pmid = sPM[cc * ld + k], qk = sQ[cc * ld + k], cldfrc = sCL[cc * ld + k];
            double es, qs;                                                       // qsat :1804 (spec oracle/stubs/l0_arithmetic.F90)
            {
                const double t_tmp = fmax(fmin(tk, COLPHYS_TMAX_TBL) - COLPHYS_TMIN_TBL, 0.0);
                const int it = (int)t_tmp + 1;
                if (it < COLPHYS_PLENEST) {
                    const double weight = fmin(t_tmp - trunc(t_tmp), 1.0);
                    es = (1.0 - weight) * s_estbl[it] + weight * s_estbl[it + 1];
                } else {
                    es = s_estbl[COLPHYS_PLENEST];
                }
                if ((pmid - es) <= 0.0) qs = 1.0;
                else qs = P.epsilo * es / (pmid - P.omeps * es);
            }
            double fsnow_conv;                                                   // cldfrc_fice :1809
            {
                const double tmax_fsnow = tmelt, tmin_fsnow = tmelt - 5.0;
                if (tk > tmax_fsnow) fsnow_conv = 0.0;
                else if (tk < tmin_fsnow) fsnow_conv = 1.0;
                else fsnow_conv = (tmax_fsnow - tk) / (tmax_fsnow - tmin_fsnow);
            }
            const double evplimit0 = fmax(1.0 - qk / (1.0 + qk) / qs, 0.0);      // :1854
            sT[cc * ld + k] = kemask_of(P, io.landfrac[col0 + cc]) * (1.0 - cldfrc) * evplimit0;
            sQ[cc * ld + k] = evplimit0;
            sCL[cc * ld + k] = (tk > tmelt) ? -1.0 : fsnow_conv;
        }
    }
    __syncthreads();

    const int c = threadIdx.x;
    if (c < nc) {
        const int col = col0 + c;
        const double gravit = P.gravit, latice = P.latice, latvap = P.latvap;
        const double prec = io.prec[col] * 1000.0;                               // :1797
        double evpvint = 0.0, flxprec = 0.0, flxsnow = 0.0;
        sFP[c * ldp] = 0.0;                                                      // :1812-1813
        sFS[c * ldp] = 0.0;
#pragma unroll SCAN_UNROLL
        for (int k = 0; k < pver; ++k) {
            const double kfac = sT[c * ld + k], pdel = sPD[c * ld + k], prdprec = sPR[c * ld + k], fs = sCL[c * ld + k];
            const bool warm = fs < 0.0;
            const double fsnow_conv = warm ? 0.0 : fs;
            double flxsntm, snowmlt;                                             // :1821-1828
            if (warm) { flxsntm = 0.0; snowmlt = flxsnow * gravit / pdel; }
            else { flxsntm = flxsnow; snowmlt = 0.0; }
            double evplimit = sQ[c * ld + k];                                    // :1854
            double evpprec = kfac * sqrt(flxprec);                               // :1868
            evplimit = fmin(evplimit, flxprec * gravit / pdel);                  // :1885 (:1875-1878 disabled by tht_tweaks)
            evplimit = fmin(evplimit, (prec - evpvint) * gravit / pdel);         // :1888
            evpprec = fmin(evplimit, evpprec);
            double evpsnow, work1, work2;
            if (flxprec > 0.0) {                                                 // :1898-1905
                work1 = fmin(fmax(0.0, flxsntm / flxprec), 1.0);
                evpsnow = evpprec * work1;
            } else {
                evpsnow = 0.0;
            }
            evpvint = evpvint + evpprec * pdel / gravit;
            const double ntprprd = prdprec - evpprec;
            if (flxprec > 0.0) work1 = fmin(fmax(0.0, flxsnow / flxprec), 1.0);  // :1924-1935
            else work1 = 0.0;
            work2 = fmax(fsnow_conv, work1);
            if (snowmlt > 0.0) work2 = 0.0;
            const double ntsnprd = prdprec * work2 - evpsnow - snowmlt;
            sPM[c * ld + k] = prdprec * work2 * latice;                          // tend_s_snwprd
            sPD[c * ld + k] = -(evpsnow + snowmlt) * latice;                     // tend_s_snwevmlt
            flxprec = flxprec + ntprprd * pdel / gravit;                         // :1943-1948
            flxsnow = flxsnow + ntsnprd * pdel / gravit;
            flxprec = fmax(flxprec, 0.0);
            flxsnow = fmax(flxsnow, 0.0);
            sFP[c * ldp + k + 1] = flxprec;
            sFS[c * ldp + k + 1] = flxsnow;
            sT[c * ld + k] = -evpprec * latvap + ntsnprd * latice;               // tend_s :1956
            sQ[c * ld + k] = evpprec;                                            // tend_q
            sPR[c * ld + k] = ntprprd;
            sCL[c * ld + k] = ntsnprd;
        }
        io.prec[col] = flxprec / 1000.0;                                         // :1965-1966
        io.snow[col] = flxsnow / 1000.0;
    }
    __syncthreads();
    tile_out(io.tend_s + col0 * L, sT, nc, pver, ld);
    tile_out(io.tend_s_snwprd + col0 * L, sPM, nc, pver, ld);
    tile_out(io.tend_s_snwevmlt + col0 * L, sPD, nc, pver, ld);
    tile_out(io.tend_q + col0 * L, sQ, nc, pver, ld);
    tile_out(io.ntprprd + col0 * L, sPR, nc, pver, ld);
    tile_out(io.ntsnprd + col0 * L, sCL, nc, pver, ld);
    tile_out(io.flxprec + col0 * LP, sFP, nc, pverp, ldp);
    tile_out(io.flxsnow + col0 * LP, sFS, nc, pverp, ldp);
}

__global__ __launch_bounds__(ENT) void zm_conv_evap_kernel(ZmDev P, const double *__restrict__ estbl, int ncol, int pver,
                                                         double deltat, colphys_zm_conv_evap_io io)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    __shared__ double s_estbl[COLPHYS_PLENEST + 1];
    zm_conv_evap_body(blockIdx.x, smem, s_estbl, P, estbl, ncol, pver, deltat, io);
}

// ------------------------------------------------------------------------------------------------------
// convtran :1976-2311 (non-microphysics path): two launches for ALL transported tracers.
//
// convtran_prep_kernel (one block per host chunk): zero-fills dqdt(:,:,m) of every transported tracer (:2299), coalesced,
//   and appends the chunk's gathered rows (row = chunk*pcols + slot, slot < il2g) to a device-side list.
//   il2g is lengath[chunk], or count(ideep > 0) when the caller has none (zm_conv_tend_2, zm_conv_intr.F90:993).
// convtran_wave_kernel (persistent waves over that list): one wave = one gathered row, lane = one transported tracer
//   (groups of 64 when there are more).  The row's mass-flux profiles (mu md du eu ed dp dpdry) are the same for every
//   lane: staged once in LDS, read as broadcasts.  Per lane the reference keeps five work profiles; two sweeps need two:
//     sweep 1, bottom-up : interface value chat(k) (:2117-2141) and the updraft recurrence conu(k) (:2154-2176)
//                          -> LDS [level][lane] (conflict-free: lanes are contiguous);
//     sweep 2, top-down  : the downdraft recurrence cond (:2160-2162, 2179-2187) one level ahead in registers, the
//                          flux-limited tendency (:2190-2262) from chat / conu of levels k, k+1, stored to dqdt.
//   The constituent and its insoluble fraction are read straight from the full-layout arrays through ideep (:2110-2113):
//   each lane walks its own contiguous profile, requested two levels ahead of use so that the (strided, one cache line
//   per 16 levels) misses complete under the recurrence's divisions and the interface logarithm.
//   dpdry: the gathered array (convtran's argument), or, for zm_conv_tend_2 on resident data, state%pdeldry through ideep,
//   /100 (zm_conv_intr.F90:1010-1017).
// ------------------------------------------------------------------------------------------------------
constexpr int CT_MAX = 120;             // transported tracers per launch (kernel-argument list)
struct ConvtranList {
    int n;
    int code[CT_MAX];                   // tracer index (0-based) | dry flag << 30
};

__global__ __launch_bounds__(256) void convtran_prep_kernel(int ncol, int pver, int pcols, ConvtranList tl, colphys_convtran_io io,
                                                            int *__restrict__ rows, int *__restrict__ nrows, int *__restrict__ il2g_out)
{
    const int ic = blockIdx.x, c0 = ic * pcols;
    const int ncc = min(pcols, ncol - c0);
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int il2g;
    if (io.lengath) {
        il2g = io.lengath[ic];
    } else {
        int c = 0;
        for (int j = threadIdx.x; j < ncc; j += blockDim.x) c += io.ideep[c0 + j] > 0 ? 1 : 0;
        if (c) atomicAdd(&s_cnt, c);
        __syncthreads();
        il2g = s_cnt;
    }
    if (threadIdx.x == 0) il2g_out[ic] = il2g;
    for (int j = threadIdx.x; j < il2g; j += blockDim.x) rows[atomicAdd(nrows, 1)] = c0 + j;
    const size_t n = (size_t)ncc * L;                            // the chunk's slab of one tracer is contiguous
    for (int t = 0; t < tl.n; ++t) {
        double *z = io.dqdt + (size_t)(tl.code[t] & 0xffffff) * F + (size_t)c0 * L;
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) z[i] = 0.0;
    }
}

__global__ __launch_bounds__(64) void convtran_wave_kernel(int ncol, int pver, int pcols, int ntp, ConvtranList tl, colphys_convtran_io io,
                                                          const double *__restrict__ pdeldry, const int *__restrict__ rows,
                                                          const int *__restrict__ nrows, const int *__restrict__ il2g_chunk)
{
    // ntp: lanes per row, a power of two >= the tracer count (64 when there are more than 32 tracers: groups of 64);
    // R = 64 / ntp rows share a wave, lane = (row-in-wave, tracer)
    extern __shared__ double smem[];
    const int lane = threadIdx.x;
    const int R = 64 / ntp, rw = lane / ntp, tw = lane - rw * ntp;
    double *s_chat = smem + lane - 64;                         // element (k, lane) at s_chat[k * 64], k = 1..pver
    double *s_conu = s_chat + (size_t)pver * 64;
    double *s_prof = smem + 2 * (size_t)pver * 64 - 1 + (size_t)rw * 7 * pver;   // this lane's row: 7 profiles [pver], 1-based
    double *s_mu = s_prof, *s_md = s_mu + pver, *s_du = s_md + pver, *s_eu = s_du + pver, *s_ed = s_eu + pver, *s_dp = s_ed + pver,
           *s_dpd = s_dp + pver;
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const double small = 1.e-36, mbsth = 1.e-15;
    const int count = *nrows;

    for (int b = blockIdx.x * R; b < count; b += gridDim.x * R) {
        const bool rvalid = b + rw < count;
        const int row = rows[rvalid ? b + rw : b];             // lanes past the list shadow the wave's first row (never stored)
        const int ic = row / pcols, c0 = ic * pcols;
        const int il2g = il2g_chunk[ic];
        int ktm = pver, kbm = pver;                                              // :2055-2060 (per chunk)
        for (int i = 0; i < il2g; ++i) {
            ktm = min(ktm, io.jt[c0 + i]);
            kbm = min(kbm, io.mx[c0 + i]);
        }
        const size_t ii = (size_t)(c0 + io.ideep[row] - 1);
        const int mx = io.mx[row];
        __syncthreads();                                                         // previous rows' readers are done
        for (int k = tw + 1; k <= pver; k += ntp) {                              // the ntp lanes of a row stage its profiles
            const size_t g = (size_t)row * L + k - 1;
            s_mu[k] = io.mu[g]; s_md[k] = io.md[g]; s_du[k] = io.du[g]; s_eu[k] = io.eu[g]; s_ed[k] = io.ed[g]; s_dp[k] = io.dp[g];
            s_dpd[k] = pdeldry ? pdeldry[ii * L + k - 1] / 100.0 : io.dpdry[g];   // zm_conv_intr.F90:1015
        }
        __syncthreads();

        for (int g = 0; g < tl.n; g += ntp) {
            const int t = g + tw;
            const bool on = rvalid && t < tl.n;
            const int code = tl.code[t < tl.n ? t : g];
            const int m = code & 0xffffff;
            const bool dry = (code >> 30) & 1;
            const double *qm = io.q + (size_t)m * F + ii * L - 1, *fr = io.fracis + (size_t)m * F + ii * L - 1;
            double *dq = io.dqdt + (size_t)m * F + ii * L - 1;

            // the four transport profiles of level k as this lane's tracer sees them (:2065-2094)
            auto fluxes = [&](int k, double &dptmp, double &dutmp, double &eutmp, double &edtmp) {
                const double dpk = s_dp[k], duk = s_du[k], euk = s_eu[k], edk = s_ed[k];
                if (dry) {
                    const double dd = s_dpd[k];
                    dptmp = dd;
                    dutmp = fdiv(duk * dpk, dd);
                    eutmp = fdiv(euk * dpk, dd);
                    edtmp = fdiv(edk * dpk, dd);
                } else {
                    dptmp = dpk; dutmp = duk; eutmp = euk; edtmp = edk;
                }
            };

            // ---- sweep 1, bottom-up: chat(k), conu(k) ----
            {
                double c_k = qm[pver], conu_kp1 = 0.0;
                double f_k = fr[pver];
                // two levels ahead: q(kk-1), q(kk-2) and fracis(kk-1), fracis(kk-2) are in flight while level kk computes
                double c_n1 = qm[max(1, pver - 1)], c_n2 = qm[max(1, pver - 2)];
                double f_n1 = fr[max(1, pver - 1)], f_n2 = fr[max(1, pver - 2)];
#pragma unroll SCAN_UNROLL
                for (int kk = pver; kk >= 1; --kk) {
                    const double c_km1 = c_n1;                                   // q(max(1, kk-1))
                    c_n1 = c_n2;
                    c_n2 = qm[max(1, kk - 3)];
                    const double minc = fmin(c_km1, c_k), maxc = fmax(c_km1, c_k);   // :2117-2141
                    double cdifr;
                    if (minc < 0) cdifr = 0.0;
                    else cdifr = fabs(c_k - c_km1) / fmax(maxc, small);
                    double chat;
                    if (cdifr > 1.E-6) {
                        const double cabv = fmax(c_km1, maxc * 1.e-12), cbel = fmax(c_k, maxc * 1.e-12);
                        chat = log(cabv / cbel) / (cabv - cbel) * cabv * cbel;
                    } else {
                        chat = 0.5 * (c_k + c_km1);
                    }
                    double dptmp, dutmp, eutmp, edtmp;
                    fluxes(kk, dptmp, dutmp, eutmp, edtmp);
                    const double efd = eutmp * f_k * c_k * dptmp;                // as grouped at :2157, :2171
                    const double mupdudp = s_mu[kk] + dutmp * dptmp;
                    double cu = chat;
                    if (mupdudp > mbsth) {                                       // > 1e-15: fdiv's operand range
                        if (kk == pver) cu = fdiv(+efd, mupdudp);
                        else cu = fdiv(s_mu[kk + 1] * conu_kp1 + efd, mupdudp);
                    }
                    s_chat[kk * 64] = chat;
                    s_conu[kk * 64] = cu;
                    conu_kp1 = cu;
                    c_k = c_km1;
                    f_k = f_n1; f_n1 = f_n2; f_n2 = fr[max(1, kk - 3)];
                }
            }
            // ---- sweep 2, top-down: cond one level ahead, tendency ----
            {
                double c_km1 = qm[1], c_k = qm[1];
                double c_n1 = qm[min(pver, 2)], c_n2 = qm[min(pver, 3)];
                double f_k = fr[1], f_n1 = fr[min(pver, 2)];
                double cond_k = s_chat[64];                                      // cond(1) = chat(1)
#pragma unroll SCAN_UNROLL
                for (int k = 1; k <= pver; ++k) {
                    const int kp1 = min(pver, k + 1);
                    const double c_kp1 = c_n1;                                   // q(min(pver, k+1))
                    c_n1 = c_n2;
                    c_n2 = qm[min(pver, k + 3)];
                    double dptmp, dutmp, eutmp, edtmp;
                    fluxes(k, dptmp, dutmp, eutmp, edtmp);
                    const double chat_k = s_chat[k * 64], chat_kp1 = s_chat[kp1 * 64];
                    const double conu_k = s_conu[k * 64], conu_kp1 = s_conu[kp1 * 64];
                    const double md_k = s_md[k], md_kp1 = s_md[kp1], mu_k = s_mu[k], mu_kp1 = s_mu[kp1];
                    // cond(k+1) (:2160-2162 for k+1 = 2, :2179-2187 beyond): provisional chat(k+1) unless the downdraft is active
                    double cond_kp1 = chat_kp1;
                    if (k + 1 <= pver && md_kp1 < -mbsth) {
                        const double edd = edtmp * f_k * c_k * dptmp;
                        if (k + 1 == 2) cond_kp1 = fdiv(-edd, md_kp1);
                        else cond_kp1 = fdiv(md_k * cond_k - edd, md_kp1);
                    }
                    if (k + 1 > pver) cond_kp1 = cond_k;                         // kp1 == k at the bottom
                    double dcondt = 0.0;
                    if (k >= ktm) {                                              // :2190-2220
                        const double fluxin = mu_kp1 * conu_kp1 + mu_k * fmin(chat_k, c_km1)
                                              - (md_k * cond_k + md_kp1 * fmin(chat_kp1, c_kp1));
                        const double fluxout = mu_k * conu_k + mu_kp1 * fmin(chat_kp1, c_k)
                                               - (md_kp1 * cond_kp1 + md_k * fmin(chat_k, c_k));
                        double netflux = fluxin - fluxout;
                        if (fabs(netflux) < fmax(fluxin, fluxout) * 1.e-12) netflux = 0.0;
                        dcondt = netflux / dptmp;
                    }
                    if (k >= kbm) {                                              // :2223-2250
                        if (k == mx) {
                            const double fluxin = mu_k * fmin(chat_k, c_km1) - md_k * cond_k;
                            const double fluxout = mu_k * conu_k - md_k * fmin(chat_k, c_k);
                            double netflux = fluxin - fluxout;
                            if (fabs(netflux) < fmax(fluxin, fluxout) * 1.e-12) netflux = 0.0;
                            dcondt = netflux / dptmp;
                        } else if (k > mx) {
                            dcondt = 0.0;
                        }
                    }
                    if (on) dq[k] = dcondt;                                      // :2301-2306
                    c_km1 = c_k; c_k = c_kp1; cond_k = cond_kp1;
                    f_k = f_n1; f_n1 = fr[min(pver, k + 2)];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// momtran :2315-2715: convective momentum transport of u, v on the gathered (deep) columns.
//
// Only the gathered columns do work -- a few thousand of 48 602 -- so the mapping is ONE WAVE PER GATHERED COLUMN with the
// lanes along the vertical.  A block serves one host chunk: its waves first write the initial values of the chunk's
// full-layout outputs (:2429-2443), then share the chunk's gathered rows among themselves (wave w takes rows w, w + nw, ...;
// round 3: a chunk with eight convecting columns used to walk them one after the other on a single wave, and the kernel took as
// long as the fullest chunk).  Per row:
//   - the column's profiles (winds, mass fluxes, layer thickness) are read as contiguous rows into the wave's LDS;
//   - everything that is element-wise in k (interface winds, pressure-gradient terms, tendencies, momentum flux,
//     end-of-step winds, KE dissipation) runs one level per lane;
//   - the two genuine recurrences (in-cloud updraft wind, bottom-up :2563-2575; downdraft wind, top-down :2579-2589) are
//     walked serially in the reference's order, in LDS, by lane 0 for u and lane 1 for v;
//   - results leave as contiguous rows.
// Waves synchronise with themselves only (wave_sync): rows of a chunk are independent.
// ------------------------------------------------------------------------------------------------------
constexpr int MOM_WAVES = 4;
__host__ __device__ inline size_t momtran_lds_doubles_per_wave(int pver) { return (size_t)(6 + 2 * 8) * (pver + 2); }

__device__ __forceinline__ void momtran_body(int bid, int tid, int nt, double *smem_block, const ZmDev &P, int ncol, int pver, int pcols, double dt,
                                             const colphys_momtran_io &io, const double *winds_v)
{
    const int ic = bid, c0 = ic * pcols;
    const int il2g = io.lengath[ic];
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const double *wind[2] = {io.winds, winds_v ? winds_v : io.winds + F};   // u, v (v may live in its own array)
    {   // initialisation of the chunk's full-layout outputs (:2429-2443): contiguous [nc][pver] runs, coalesced
        const int nc = min(pcols, ncol - c0);
        const size_t base = (size_t)c0 * L, n = (size_t)nc * L;
        for (size_t i = tid; i < n; i += nt) {
            for (int m = 0; m < 2; ++m) {
                const size_t o = m * F + base + i;
                const double w = wind[m][base + i];
                io.pguall[o] = 0.0; io.pgdall[o] = 0.0; io.dqdt[o] = 0.0; io.icwu[o] = w; io.icwd[o] = w;
            }
            io.seten[base + i] = 0.0;
        }
    }
    if (il2g <= 0) return;                                    // block-uniform
    __syncthreads();                                          // the rows below overwrite parts of what was just initialised
    const int lane = tid & (WAVE - 1), wave = tid >> 6, nw = nt >> 6;
    const int LD = pver + 2;                                  // 1-based profiles, [0] and [pver+1] as guards
    const double mbsth = 1.e-15;
    double *smem = smem_block + (size_t)wave * momtran_lds_doubles_per_wave(pver);

    double *s_mu = smem, *s_md = s_mu + LD, *s_du = s_md + LD, *s_eu = s_du + LD, *s_ed = s_eu + LD, *s_dp = s_ed + LD;
    double *s_c = s_dp + LD;          // const   [2][LD]
    double *s_chat = s_c + 2 * LD;    // chat    [2][LD]
    double *s_conu = s_chat + 2 * LD; // conu    [2][LD]
    double *s_cond = s_conu + 2 * LD; // cond    [2][LD]
    double *s_pgu = s_cond + 2 * LD;  // pgu     [2][LD]
    double *s_pgd = s_pgu + 2 * LD;   // pgd     [2][LD]
    double *s_mf = s_pgd + 2 * LD;    // mflux   [2][LD]  (index pver+1 stays 0)
    double *s_wf = s_mf + 2 * LD;     // windf   [2][LD]

    int ktm = pver, kbm = pver;                                                  // :2449-2454 (per chunk)
    for (int i = 0; i < il2g; ++i) {
        ktm = min(ktm, io.jt[c0 + i]);
        kbm = min(kbm, io.mx[c0 + i]);
    }
    for (int slot = wave; slot < il2g; slot += nw) {          // wave-uniform
    const int row = c0 + slot;                                // gathered row of the flat layout
    const size_t ii = (size_t)(c0 + io.ideep[row] - 1);
    const int mx = io.mx[row];

    for (int k = lane + 1; k <= pver; k += WAVE) {
        const size_t g = (size_t)row * L + (k - 1);
        s_mu[k] = io.mu[g]; s_md[k] = io.md[g]; s_du[k] = io.du[g]; s_eu[k] = io.eu[g]; s_ed[k] = io.ed[g]; s_dp[k] = io.dp[g];
        for (int m = 0; m < 2; ++m) {
            s_c[m * LD + k] = wind[m][ii * L + (k - 1)];                         // :2461-2466
            s_mf[m * LD + k] = 0.0; s_wf[m * LD + k] = 0.0;                      // :2436-2438
        }
    }
    if (lane < 2) { s_mf[lane * LD + pver + 1] = 0.0; s_mf[lane * LD] = 0.0; }
    wave_sync();

    for (int m = 0; m < 2; ++m) {
        const double *cn = s_c + m * LD;
        double *chat = s_chat + m * LD, *conu = s_conu + m * LD, *cond = s_cond + m * LD, *pgu = s_pgu + m * LD,
               *pgd = s_pgd + m * LD;
        for (int k = lane + 1; k <= pver; k += WAVE) {
            const int km1 = max(1, k - 1);
            const double ch = 0.5 * (cn[k] + cn[km1]);                           // :2473-2488
            chat[k] = ch; conu[k] = ch; cond[k] = ch;
            double gu, gd;
            if (k == 1) {                                                        // :2497-2499
                gu = 0.0; gd = 0.0;
            } else if (k <= pver - 1) {                                          // :2501-2521
                const int kp1 = k + 1;
                const double mududp = (s_mu[k] * (cn[k] - cn[km1]) / s_dp[km1] + s_mu[kp1] * (cn[kp1] - cn[k]) / s_dp[k]);
                gu = -P.momcu * 0.5 * mududp;
                const double mddudp = (s_md[k] * (cn[k] - cn[km1]) / s_dp[km1] + s_md[kp1] * (cn[kp1] - cn[k]) / s_dp[k]);
                gd = -P.momcd * 0.5 * mddudp;
            } else {                                                             // :2524-2535
                const double mududp = s_mu[k] * (cn[k] - cn[km1]) / s_dp[km1];
                gu = -P.momcu * mududp;
                const double mddudp = s_md[k] * (cn[k] - cn[km1]) / s_dp[km1];
                gd = -P.momcd * mddudp;
            }
            pgu[k] = gu; pgd[k] = gd;
        }
    }
    wave_sync();

    // the two recurrences, serial in k: lane 0 walks them for u, lane 1 for v
    if (lane < 2) {
        const int m = lane;                                   // one component per lane: u on lane 0, v on lane 1
        const double *cn = s_c + m * LD, *pgu = s_pgu + m * LD, *pgd = s_pgd + m * LD;
        double *conu = s_conu + m * LD, *cond = s_cond + m * LD;
        {                                                                        // :2543-2558
            const int k = 2, km1 = 1, kk = pver;
            const double mupdudp = s_mu[kk] + s_du[kk] * s_dp[kk];
            if (mupdudp > mbsth) conu[kk] = (+s_eu[kk] * cn[kk] * s_dp[kk] + pgu[kk] * s_dp[kk]) / mupdudp;
            // operator precedence of the reference kept as is (:2554)
            if (s_md[k] < -mbsth) cond[k] = (-s_ed[km1] * cn[km1] * s_dp[km1]) - pgd[km1] * s_dp[km1] / s_md[k];
        }
#pragma unroll SCAN_UNROLL
        for (int kk = pver - 1; kk >= 1; --kk) {                                 // :2563-2575
            const double mupdudp = s_mu[kk] + s_du[kk] * s_dp[kk];
            if (mupdudp > mbsth)
                conu[kk] = (s_mu[kk + 1] * conu[kk + 1] + s_eu[kk] * cn[kk] * s_dp[kk] + pgu[kk] * s_dp[kk]) / mupdudp;
        }
#pragma unroll SCAN_UNROLL
        for (int k = 3; k <= pver; ++k) {                                        // :2579-2589
            const int km1 = k - 1;
            if (s_md[k] < -mbsth)
                cond[k] = (s_md[km1] * cond[km1] - s_ed[km1] * cn[km1] * s_dp[km1] - pgd[km1] * s_dp[km1]) / s_md[k];
        }
    }
    wave_sync();

    for (int m = 0; m < 2; ++m) {
        const double *chat = s_chat + m * LD, *conu = s_conu + m * LD, *cond = s_cond + m * LD;
        double *mf = s_mf + m * LD;
        for (int k = lane + 1; k <= pver; k += WAVE) {
            double dcondt = 0.0;
            if (k >= ktm) {                                                      // :2596-2611
                const int kp1 = min(pver, k + 1);
                dcondt = +(s_mu[kp1] * (conu[kp1] - chat[kp1]) - s_mu[k] * (conu[k] - chat[k])
                           + s_md[kp1] * (cond[kp1] - chat[kp1]) - s_md[k] * (cond[k] - chat[k])) / s_dp[k];
                mf[k] = -s_mu[k] * (conu[k] - chat[k]) - s_md[k] * (cond[k] - chat[k]);       // :2646-2653
            }
            if (k >= kbm && k == mx)                                             // :2615-2627
                dcondt = (1.0 / s_dp[k]) * (-s_mu[k] * (conu[k] - chat[k]) - s_md[k] * (cond[k] - chat[k]));
            const size_t o = m * F + ii * L + (k - 1);                           // :2632-2642
            io.dqdt[o] = dcondt; io.pguall[o] = -s_pgu[m * LD + k]; io.pgdall[o] = -s_pgd[m * LD + k];
            io.icwu[o] = conu[k]; io.icwd[o] = cond[k];
        }
    }
    wave_sync();
    for (int m = 0; m < 2; ++m) {                                                // :2658-2666
        const double *cn = s_c + m * LD, *mf = s_mf + m * LD;
        for (int k = lane + 1; k <= pver; k += WAVE)
            if (k >= ktm) s_wf[m * LD + k] = cn[k] - (mf[k + 1] - mf[k]) * dt / s_dp[k];
    }
    wave_sync();
    for (int k = lane + 1; k <= pver; k += WAVE) {                               // :2675-2712
        double gset = 0.0;
        if (k >= ktm) {
            const int km1 = max(1, k - 1), kp1 = min(pver, k + 1);
            const double *w0u = s_c, *w0v = s_c + LD, *mfu = s_mf, *mfv = s_mf + LD, *wfu = s_wf, *wfv = s_wf + LD;
            const double utop = (w0u[k] + w0u[km1]) / 2.0, vtop = (w0v[k] + w0v[km1]) / 2.0;
            const double ubot = (w0u[kp1] + w0u[k]) / 2.0, vbot = (w0v[kp1] + w0v[k]) / 2.0;
            const double fket = utop * mfu[k] + vtop * mfv[k];
            const double fkeb = ubot * mfu[k + 1] + vbot * mfv[k + 1];
            const double ketend_cons = (fket - fkeb) / s_dp[k];
            const double ketend = ((wfu[k] * wfu[k] + wfv[k] * wfv[k]) - (w0u[k] * w0u[k] + w0v[k] * w0v[k])) * 0.5 / dt;
            gset = ketend_cons - ketend;
        }
        io.seten[ii * L + (k - 1)] = gset;
    }
    wave_sync();
    }
}

__global__ __launch_bounds__(MOM_WAVES * WAVE) void momtran_wave_kernel(ZmDev P, int ncol, int pver, int pcols, double dt,
                                                                       colphys_momtran_io io, const double *winds_v)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    momtran_body(blockIdx.x, threadIdx.x, MOM_WAVES * WAVE, smem, P, ncol, pver, pcols, dt, io, winds_v);
}

// ------------------------------------------------------------------------------------------------------
// The tail of zm_conv_tend for a small rank (colphys_common.h ZmTailSmall): one launch of heterogeneous blocks --
// [0, ng) geopotential_t tiles, [ng, ng + ne) zm_conv_evap tiles, the rest one momtran block per host chunk -- each running
// the body of the stand-alone kernel.  ENT threads per block; the dynamic LDS is the largest of the three.
// ------------------------------------------------------------------------------------------------------
struct Tail3Geo { int dycore_kind; const double *piln, *pint, *pmid, *pdel, *rpdel, *t, *q, *rair, *zvir; double gravit; double *zi, *zm; size_t rzs; };
__global__ __launch_bounds__(ENT) void zm_tail3_kernel(ZmDev P, const double *__restrict__ estbl, int ncol, int pver, int pcols, int ng, int ne,
                                                     Tail3Geo G, double ztodt, colphys_zm_conv_evap_io E, colphys_momtran_io M,
                                                     const double *winds_v)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    __shared__ double s_estbl[COLPHYS_PLENEST + 1];
    const int b = blockIdx.x;                                   // block-uniform dispatch
    if (b < ng)
        geopotential_body(b, threadIdx.x, ENT, smem, ncol, pver, G.dycore_kind, 1, G.piln, G.pint, G.pmid, G.pdel, G.rpdel, G.t, G.q, G.rair,
                          G.gravit, G.zvir, G.zi, G.zm, G.rzs);
    else if (b < ng + ne)
        zm_conv_evap_body(b - ng, smem, s_estbl, P, estbl, ncol, pver, ztodt, E);
    else
        momtran_body(b - ng - ne, threadIdx.x, ENT, smem, P, ncol, pver, pcols, ztodt, M, winds_v);
}

// second and third physics_update of zm_conv_tend in one pass (zm_conv_intr.F90:802-806, 829-836), element-wise:
//   t2 = t1 + evap_s dt/cp (kept: the second geopotential_t reads it),  q1 <- max(q1 + evap_q dt, qmin),
//   t1 <- t2 + seten dt/cp,  u1 = u + du dt,  v1 = v + dv dt;  ptend_all += in the reference's order
__global__ void physics_update23_kernel(size_t n, double dt, double cpair, double qmin, double *t1, double *t2, double *q1,
                                        const double *__restrict__ evap_s, const double *__restrict__ evap_q,
                                        const double *__restrict__ seten, const double *__restrict__ u, const double *__restrict__ v,
                                        const double *__restrict__ du, const double *__restrict__ dv, double *u1, double *v1,
                                        double *sum_s, double *sum_q, double *sum_u, double *sum_v)
{
    CHAIN_PRIO();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double es = evap_s[i], eq = evap_q[i], se = seten[i];
        const double ta = t1[i] + es * dt / cpair;                         // physics_types.F90:427, second update
        t2[i] = ta;
        q1[i] = fmax(q1[i] + eq * dt, qmin);                               // :321, qneg3 :329
        t1[i] = ta + se * dt / cpair;                                      // third update
        u1[i] = u[i] + du[i] * dt;                                         // :295
        v1[i] = v[i] + dv[i] * dt;                                         // :303
        sum_s[i] = (sum_s[i] + es) + se;                                   // physics_ptend_sum, twice (zm_conv_intr.F90:802, 829)
        sum_q[i] = sum_q[i] + eq;
        sum_u[i] = du[i];
        sum_v[i] = dv[i];
    }
}

// second and third geopotential_t in one launch: blockIdx.y picks (t2 -> scratch heights) or (t1 -> zi1, zm1)
__global__ __launch_bounds__(GNT) void geopotential_pair_kernel(int ncol, int pver, int dycore_kind, const double *__restrict__ piln,
                                                              const double *__restrict__ pint, const double *__restrict__ pmid,
                                                              const double *__restrict__ pdel, const double *__restrict__ rpdel,
                                                              const double *__restrict__ t_a, const double *__restrict__ t_b,
                                                              const double *__restrict__ q, const double *__restrict__ rair, double gravit,
                                                              const double *__restrict__ zvir, double *zi_a, double *zm_a, double *zi_b,
                                                              double *zm_b, size_t rzs)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    const bool second = blockIdx.y != 0;
    geopotential_body(blockIdx.x, threadIdx.x, GNT, smem, ncol, pver, dycore_kind, 1, piln, pint, pmid, pdel, rpdel, second ? t_b : t_a, q, rair,
                      gravit, zvir, second ? zi_b : zi_a, second ? zm_b : zm_a, rzs);
}

// ------------------------------------------------------------------------------------------------------
// physics_update (tendency application part), element-wise, grid-stride
// ------------------------------------------------------------------------------------------------------
__global__ void physics_update_kernel(size_t n, double dt, double cpair, double qmin, const double *__restrict__ t_in,
                                      const double *__restrict__ s_tend, double *t_out, const double *__restrict__ q_in,
                                      const double *__restrict__ q_tend, double *q_out, const double *__restrict__ u_in,
                                      const double *__restrict__ u_tend, double *u_out, const double *__restrict__ v_in,
                                      const double *__restrict__ v_tend, double *v_out, double *sum_s, double *sum_q,
                                      double *sum_u, double *sum_v, int first)
{
    CHAIN_PRIO();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (t_out) t_out[i] = t_in[i] + s_tend[i] * dt / cpair;            // physics_types.F90:427
        if (q_out) q_out[i] = fmax(q_in[i] + q_tend[i] * dt, qmin);       // :321, qneg3 :329
        if (u_out) u_out[i] = u_in[i] + u_tend[i] * dt;                   // :295
        if (v_out) v_out[i] = v_in[i] + v_tend[i] * dt;                   // :303
        // physics_ptend_sum (zm_conv_intr.F90:737, 802, 858): ptend_all = ptend_all + ptend_loc
        // (`first`: bit per field -- s 1, q 2, u 4, v 8 -- set where this is the first contribution)
        if (sum_s && s_tend) sum_s[i] = (first & 1) ? s_tend[i] : sum_s[i] + s_tend[i];
        if (sum_q && q_tend) sum_q[i] = (first & 2) ? q_tend[i] : sum_q[i] + q_tend[i];
        if (sum_u && u_tend) sum_u[i] = (first & 4) ? u_tend[i] : sum_u[i] + u_tend[i];
        if (sum_v && v_tend) sum_v[i] = (first & 8) ? v_tend[i] : sum_v[i] + v_tend[i];
    }
}

inline int set_lds(const void *fn, size_t bytes)
{
    if (bytes > 64 * 1024)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return -1;
    return 0;
}

}  // namespace

int launch_geopotential_t(int ncol, int pver, int dycore_kind, int nq, int nspecies, const double *piln,
                          const double *pint, const double *pmid, const double *pdel, const double *rpdel,
                          const double *t, const double *q, const double *rair, double gravit, const double *zvir,
                          double *zi, double *zm, hipStream_t st, int rz_uniform)
{
    (void)nq;
    const int ldp = ((pver + 1) & 1) ? pver + 1 : pver + 2;
    const size_t lds = (size_t)GT * 2 * ldp * sizeof(double);
    if (set_lds((const void *)geopotential_kernel, lds)) return COLPHYS_ERR_HIP;
    const int nblk = (ncol + GT - 1) / GT;
    hipLaunchKernelGGL(geopotential_kernel, dim3(nblk), dim3(GNT), lds, st, ncol, pver, dycore_kind, nspecies, piln, pint,
                       pmid, pdel, rpdel, t, q, rair, gravit, zvir, zi, zm, (size_t)(rz_uniform ? 0 : 1));
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_geopotential_dse(int ncol, int pver, int dycore_kind, int nspecies, const double *piln, const double *pint,
                            const double *pmid, const double *pdel, const double *rpdel, const double *dse, const double *q,
                            const double *phis, const double *rair, double gravit, const double *cpair, const double *zvir,
                            double *t, double *zi, double *zm, hipStream_t st)
{
    const int ldp = ((pver + 1) & 1) ? pver + 1 : pver + 2;
    const size_t lds = (size_t)GT * 5 * ldp * sizeof(double);
    if (set_lds((const void *)geopotential_dse_kernel, lds)) return COLPHYS_ERR_HIP;
    const int nblk = (ncol + GT - 1) / GT;
    hipLaunchKernelGGL(geopotential_dse_kernel, dim3(nblk), dim3(GNT), lds, st, ncol, pver, dycore_kind, nspecies, piln, pint, pmid,
                       pdel, rpdel, dse, q, phis, rair, gravit, cpair, zvir, t, zi, zm);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_zm_conv_evap(const ZmDev &P, const double *estbl_dev, int ncol, int pver, double deltat,
                        const colphys_zm_conv_evap_io &io, hipStream_t st)
{
    const int ld = (pver & 1) ? pver : pver + 1, ldp = ((pver + 1) & 1) ? pver + 1 : pver + 2;
    const size_t lds = (size_t)ETILE * (6 * ld + 2 * ldp) * sizeof(double);
    if (lds > 150 * 1024) return COLPHYS_ERR_UNSUPPORTED;
    if (set_lds((const void *)zm_conv_evap_kernel, lds)) return COLPHYS_ERR_HIP;
    const int nblk = (ncol + ETILE - 1) / ETILE;
    hipLaunchKernelGGL(zm_conv_evap_kernel, dim3(nblk), dim3(ENT), lds, st, P, estbl_dev, ncol, pver, deltat, io);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_momtran(const ZmDev &P, int ncol, int pver, int pcols, double dt, const colphys_momtran_io &io, hipStream_t st,
                   const double *winds_v)
{
    const size_t lds = MOM_WAVES * momtran_lds_doubles_per_wave(pver) * sizeof(double);
    if (set_lds((const void *)momtran_wave_kernel, lds)) return COLPHYS_ERR_HIP;
    const int nchunks = (ncol + pcols - 1) / pcols;
    hipLaunchKernelGGL(momtran_wave_kernel, dim3(nchunks), dim3(MOM_WAVES * WAVE), lds, st, P, ncol, pver, pcols, dt, io, winds_v);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_zm_tail_small(const ZmDev &P, const double *estbl_dev, const ZmTailSmall &T, hipStream_t st)
{
    const int ncol = T.ncol, pver = T.pver;
    const int ld = (pver & 1) ? pver : pver + 1, ldp = ((pver + 1) & 1) ? pver + 1 : pver + 2;
    const size_t lds_geo = (size_t)GT * 2 * ldp * sizeof(double), lds_evap = (size_t)ETILE * (6 * ld + 2 * ldp) * sizeof(double);
    const size_t lds_mom = (ENT / WAVE) * momtran_lds_doubles_per_wave(pver) * sizeof(double);
    const size_t lds = lds_geo > lds_evap ? (lds_geo > lds_mom ? lds_geo : lds_mom) : (lds_evap > lds_mom ? lds_evap : lds_mom);
    if (lds > 150 * 1024) return COLPHYS_ERR_UNSUPPORTED;
    if (set_lds((const void *)zm_tail3_kernel, lds)) return COLPHYS_ERR_HIP;
    const int ng = (ncol + GT - 1) / GT, ne = (ncol + ETILE - 1) / ETILE, nm = (ncol + T.pcols - 1) / T.pcols;
    const size_t rzs = T.rz_uniform ? 0 : 1;
    Tail3Geo G{T.dycore_kind, T.lnpint, T.pint, T.pmid, T.pdel, T.rpdel, T.t1, T.q1, T.rair, T.zvir, P.gravit, T.zi1, T.zm1, rzs};
    hipLaunchKernelGGL(zm_tail3_kernel, dim3(ng + ne + nm), dim3(ENT), lds, st, P, estbl_dev, ncol, pver, T.pcols, ng, ne, G, T.ztodt, T.evap,
                       T.mom, T.v);
    const size_t n = (size_t)ncol * pver;
    const int nblk = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(physics_update23_kernel, dim3(nblk), dim3(256), 0, st, n, T.ztodt, P.cpair, T.qmin, T.t1, T.t2, T.q1, T.evap.tend_s,
                       T.evap.tend_q, T.mom.seten, T.u, T.v, T.mom.dqdt, T.mom.dqdt + n, T.u1, T.v1, T.ptend_s, T.ptend_q, T.ptend_u, T.ptend_v);
    if (set_lds((const void *)geopotential_pair_kernel, lds_geo)) return COLPHYS_ERR_HIP;
    hipLaunchKernelGGL(geopotential_pair_kernel, dim3(ng, 2), dim3(GNT), lds_geo, st, ncol, pver, T.dycore_kind, T.lnpint, T.pint, T.pmid, T.pdel,
                       T.rpdel, T.t2, T.t1, T.q1, T.rair, P.gravit, T.zvir, T.zi_scratch, T.zm_scratch, T.zi1, T.zm1, rzs);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_physics_update(int ncol, int pver, double dt, double cpair, double qmin, const double *t_in,
                          const double *s_tend, double *t_out, const double *q_in, const double *q_tend, double *q_out,
                          const double *u_in, const double *u_tend, double *u_out, const double *v_in,
                          const double *v_tend, double *v_out, hipStream_t st, double *sum_s, double *sum_q, double *sum_u,
                          double *sum_v, int first)
{
    const size_t n = (size_t)ncol * pver;
    const int nblk = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(physics_update_kernel, dim3(nblk), dim3(256), 0, st, n, dt, cpair, qmin, t_in, s_tend, t_out, q_in,
                       q_tend, q_out, u_in, u_tend, u_out, v_in, v_tend, v_out, sum_s, sum_q, sum_u, sum_v, first);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

int launch_convtran(int ncol, int pver, int pcols, int ncnst, const colphys_convtran_io &io, hipStream_t st, const double *pdeldry,
                    int *ws_rows, int *ws_count, int *ws_il2g)
{
    // lanes per row: the tracer count rounded up to a power of two, at least 8 (several rows share a wave when few
    // tracers are transported: cloud liquid + ice alone would leave 62 lanes idle), 64 beyond 32 tracers
    auto lanes_per_row = [](int n) { int p = 8; while (p < n) p <<= 1; return p > 64 ? 64 : p; };
    auto lds_bytes = [&](int ntp) { return (2 * (size_t)pver * 64 + (size_t)(64 / ntp) * 7 * pver) * sizeof(double); };
    if (lds_bytes(64) > 160 * 1024) return COLPHYS_ERR_UNSUPPORTED;
    const int nchunks = (ncol + pcols - 1) / pcols;
    ConvtranList tl;
    tl.n = 0;
    for (int m = 1; m <= ncnst; ++m) {                      // do m = 2, ncnst (:2062); m == ncnst flushes the last group
        if (m < ncnst && io.doconvtran[m]) tl.code[tl.n++] = m | (io.isdry[m] ? (1 << 30) : 0);
        if (tl.n == CT_MAX || (m == ncnst && tl.n > 0)) {
            if (hipMemsetAsync(ws_count, 0, sizeof(int), st) != hipSuccess) return COLPHYS_ERR_HIP;
            hipLaunchKernelGGL(convtran_prep_kernel, dim3(nchunks), dim3(256), 0, st, ncol, pver, pcols, tl, io, ws_rows, ws_count, ws_il2g);
            int ntp = lanes_per_row(tl.n);
            while (ntp < 64 && lds_bytes(ntp) > 160 * 1024) ntp <<= 1;          // long profiles: fewer rows per wave
            const size_t lds = lds_bytes(ntp);
            // per launch, not once per process: the attribute belongs to the CURRENT device's copy of the kernel, and
            // colphys_finalize / colphys_init may have moved the engine to another one
            if (set_lds((const void *)convtran_wave_kernel, lds)) return COLPHYS_ERR_HIP;
            int per_cu = (int)((160 * 1024) / (lds + 256));
            per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
            const int nper = ncol < 256 * per_cu ? ncol : 256 * per_cu;         // persistent waves over the gathered rows
            hipLaunchKernelGGL(convtran_wave_kernel, dim3(nper), dim3(64), lds, st, ncol, pver, pcols, ntp, tl, io, pdeldry, ws_rows,
                               ws_count, ws_il2g);
            tl.n = 0;
        }
    }
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

// ---- d_p_coupling derived fields (fv/dp_coupling.F90:550-629), the element-wise parts around geopotential_t ----------
namespace {
constexpr int DF_MAXC = 256;
struct DfQmin { double v[DF_MAXC]; };

__global__ void derived_pressure_kernel(int ncol, int pver, const double *__restrict__ pint, double *__restrict__ pdel,
                                        double *__restrict__ rpdel, double *__restrict__ pmid, double *__restrict__ lnpmid)
{
    const size_t L = (size_t)pver, LP = L + 1, n = (size_t)ncol * L;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t c = i / L, k = i - c * L;
        const double p0 = pint[c * LP + k], p1 = pint[c * LP + k + 1];
        const double dp = p1 - p0;                                               // :554
        pdel[i] = dp;
        rpdel[i] = 1.0 / dp;                                                     // :555
        const double pm = 0.5 * (p0 + p1);                                       // :556
        pmid[i] = pm;
        lnpmid[i] = cp_log(pm);                                                  // :557
    }
}

// bottom-layer fix (:565-590): one thread per (constituent, column); pdel from pint so that the kernel does not depend on
// the kernel above having finished for other columns
__global__ void derived_qfix_kernel(int ncol, int pver, int m0, int nm, DfQmin qmin, const double *__restrict__ pint, double *q)
{
    const size_t L = (size_t)pver, LP = L + 1, F = (size_t)ncol * L;
    const size_t n = (size_t)nm * ncol, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t mm = i / ncol, c = i - mm * ncol;
        double *qc = q + (m0 + mm) * F + c * L;
        const double qm = qmin.v[mm];
        const double qbot = qc[pver - 1];
        if (qbot < qm) {
            const double *pc = pint + c * LP;
            const double pdel_bot = pc[pver] - pc[pver - 1], pdel_m1 = pc[pver - 1] - pc[pver - 2];
            const double qmavl = qc[pver - 2] - qm;                              // :569
            const double dqreq = (qm - qbot) * pdel_bot / pdel_m1;               // :571-572
            if (dqreq < qmavl) {                                                 // :575-577
                qc[pver - 1] = qm;
                qc[pver - 2] = qc[pver - 2] - dqreq;
            }
        }
    }
}

__global__ void derived_dse_kernel(int ncol, int pver, double gravit, const double *__restrict__ cpair, const double *__restrict__ t,
                                   const double *__restrict__ zm, const double *__restrict__ phis, double *__restrict__ s)
{
    const size_t L = (size_t)pver, n = (size_t)ncol * L, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        s[i] = cpair[i] * t[i] + gravit * zm[i] + phis[i / L];                   // :626-627
}
}  // namespace

int launch_derived_fields(int ncol, int pver, int ncnst, int dycore_kind, int nspecies, double gravit,
                          const colphys_derived_fields_io &io, hipStream_t st)
{
    const size_t n = (size_t)ncol * pver;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(derived_pressure_kernel, dim3(blocks), dim3(256), 0, st, ncol, pver, io.pint, io.pdel, io.rpdel, io.pmid, io.lnpmid);
    for (int m0 = 0; m0 < ncnst; m0 += DF_MAXC) {
        DfQmin qm;
        const int nm = ncnst - m0 < DF_MAXC ? ncnst - m0 : DF_MAXC;
        for (int m = 0; m < nm; ++m) qm.v[m] = io.qmin[m0 + m];
        const size_t nq = (size_t)nm * ncol;
        const int bq = (int)((nq + 255) / 256 < 4096 ? (nq + 255) / 256 : 4096);
        hipLaunchKernelGGL(derived_qfix_kernel, dim3(bq), dim3(256), 0, st, ncol, pver, m0, nm, qm, io.pint, io.q);
    }
    int rc = launch_geopotential_t(ncol, pver, dycore_kind, ncnst, nspecies, io.lnpint, io.pint, io.pmid, io.pdel, io.rpdel, io.t, io.q,
                                   io.rair, gravit, io.zvir, io.zi, io.zm, st);
    if (rc) return rc;
    hipLaunchKernelGGL(derived_dse_kernel, dim3(blocks), dim3(256), 0, st, ncol, pver, gravit, io.cpair, io.t, io.zm, io.phis, io.s);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

// ---- the dry-mass tail of d_p_coupling's derived fields (fv/dp_coupling.F90:630-651) ---------------------------------
// set_state_pdry (physics_types.F90:1925-1964): a top-down running sum per column.  Same tiling as geopotential_t: the
// element-wise part (pdeldry, its reciprocal) by all threads over the block's [GT columns][pver] slab, the sum by one lane per
// column in LDS (odd row stride), the interface / mid-point fields and their logarithms stored coalesced.
// set_wet_to_dry (:1968-1984) + qneg3 (dp_coupling.F90:644-651): element-wise over every constituent, grid-stride.
namespace {
__global__ __launch_bounds__(GNT) void dry_mass_scan_kernel(int ncol, int pver, const double *__restrict__ pint, const double *__restrict__ pdel,
                                                          const double *__restrict__ qv, double *__restrict__ pdeldry,
                                                          double *__restrict__ rpdeldry, double *__restrict__ pmiddry,
                                                          double *__restrict__ lnpmiddry, double *__restrict__ pintdry,
                                                          double *__restrict__ lnpintdry, double *__restrict__ psdry)
{
    extern __shared__ double smem[];
    const int col0 = blockIdx.x * GT;
    const int nc = min(GT, ncol - col0);
    const int pverp = pver + 1, ldp = ldk_of(pverp);
    const size_t L = (size_t)pver, LP = (size_t)pverp;
    double *sI = smem;                    // [GT][ldp]: pintdry(k)
    double *sD = smem + GT * ldp;         // [GT][ldp]: pdeldry(k), then pmiddry(k)
    const int n = nc * pver;
    for (int i = threadIdx.x; i < n; i += GNT) {
        const int c = i / pver, k = i - c * pver;
        const size_t g = col0 * L + i;
        const double d = pdel[g] * (1.0 - qv[g]);                              // :1948
        pdeldry[g] = d;
        rpdeldry[g] = 1.0 / d;                                                 // :1956
        sD[c * ldp + k] = d;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < nc) {
        double p = pint[(size_t)(col0 + c) * LP];                              // :1945-1946
        double ps = p;
        sI[c * ldp] = p;
#pragma unroll SCAN_UNROLL
        for (int k = 0; k < pver; ++k) {
            const double d = sD[c * ldp + k];
            const double pn = p + d;                                           // :1952
            sI[c * ldp + k + 1] = pn;
            sD[c * ldp + k] = (pn + p) / 2.0;                                  // :1953
            ps = ps + d;                                                       // :1954
            p = pn;
        }
        psdry[col0 + c] = ps;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc * pverp; i += GNT) {
        const int cc = i / pverp, k = i - cc * pverp;
        const double v = sI[cc * ldp + k];
        pintdry[col0 * LP + i] = v;
        lnpintdry[col0 * LP + i] = cp_log(v);                                  // :1958
    }
    for (int i = threadIdx.x; i < n; i += GNT) {
        const int cc = i / pver, k = i - cc * pver;
        const double v = sD[cc * ldp + k];
        pmiddry[col0 * L + i] = v;
        lnpmiddry[col0 * L + i] = cp_log(v);                                   // :1957
    }
}

struct DryTab { double qmin[DF_MAXC]; unsigned char dry[DF_MAXC]; };
__global__ void wet_to_dry_qneg_kernel(int ncol, int pver, int m0, int nm, DryTab tab, const double *__restrict__ pdel,
                                       const double *__restrict__ pdeldry, double *q)
{
    const size_t F = (size_t)ncol * pver, n = (size_t)nm * F, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t mm = i / F, g = i - mm * F;
        double v = q[(size_t)m0 * F + i];
        if (tab.dry[mm]) v = v * pdel[g] / pdeldry[g];                          // physics_types.F90:1979-1981
        if (v < tab.qmin[mm]) v = tab.qmin[mm];                                // qneg3
        q[(size_t)m0 * F + i] = v;
    }
}
}  // namespace

int launch_dry_mass_fields(int ncol, int pver, int ncnst, const colphys_dry_mass_io &io, hipStream_t st)
{
    const int ldp = ((pver + 1) & 1) ? pver + 1 : pver + 2;
    const size_t lds = (size_t)GT * 2 * ldp * sizeof(double);
    if (set_lds((const void *)dry_mass_scan_kernel, lds)) return COLPHYS_ERR_HIP;
    hipLaunchKernelGGL(dry_mass_scan_kernel, dim3((ncol + GT - 1) / GT), dim3(GNT), lds, st, ncol, pver, io.pint, io.pdel, io.q, io.pdeldry,
                       io.rpdeldry, io.pmiddry, io.lnpmiddry, io.pintdry, io.lnpintdry, io.psdry);
    for (int m0 = 0; m0 < ncnst; m0 += DF_MAXC) {
        DryTab tab;
        const int nm = ncnst - m0 < DF_MAXC ? ncnst - m0 : DF_MAXC;
        for (int m = 0; m < nm; ++m) { tab.qmin[m] = io.qmin[m0 + m]; tab.dry[m] = io.isdry[m0 + m] ? 1 : 0; }
        const size_t n = (size_t)nm * ncol * pver;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(wet_to_dry_qneg_kernel, dim3(blocks), dim3(256), 0, st, ncol, pver, m0, nm, tab, io.pdel, io.pdeldry, io.q);
    }
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

// ---- convect_diagnostics_calc (convect_diagnostics.F90:115-249): element-wise merge, grid-stride ------------------
namespace {
__global__ void convect_diag_kernel(int ncol, int pver, int clubb, colphys_convect_diag_io io)
{
    const size_t L = (size_t)pver, LP = L + 1, nM = (size_t)ncol * L, nI = (size_t)ncol * LP;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = i0; i < nI; i += stride) {
        if (clubb) { if (io.cmfmc2) io.cmfmc2[i] = 0.0; }                        // :179
        else io.cmfmc[i] = io.cmfmc[i] + io.cmfmc2[i];                           // :199 (+0 under CLUBB: unchanged bits)
    }
    for (size_t i = i0; i < nM; i += stride) {
        double rsh, q2;
        if (clubb) {                                                             // :180-188
            rsh = 0.0; q2 = 0.0;
            if (io.rprdsh) io.rprdsh[i] = 0.0;
            if (io.qc2) io.qc2[i] = 0.0;
            if (io.icwmr) io.icwmr[i] = 0.0;
            if (io.evapcsh) io.evapcsh[i] = 0.0;
        } else {
            rsh = io.rprdsh[i]; q2 = io.qc2[i];
        }
        io.rprdtot[i] = rsh + io.rprddp[i];                                      // :223
        io.qc[i] = io.qc[i] + q2;                                                // :230
    }
    for (size_t c = i0; c < (size_t)ncol; c += stride) {
        double t2, b2, l2;
        if (clubb) {
            t2 = (double)pver; b2 = 1.0; l2 = 0.0;                               // :185-186, 183
            if (io.rliq2) io.rliq2[c] = 0.0;
            if (io.precc) io.precc[c] = 0.0;
            if (io.snow) io.snow[c] = 0.0;
        } else {
            t2 = io.cnt2[c]; b2 = io.cnb2[c]; l2 = io.rliq2[c];
        }
        double cnt = io.cnt[c], cnb = io.cnb[c];
        if (t2 < cnt) cnt = t2;                                                  // :210-216
        if (b2 > cnb) cnb = b2;
        if (cnb == 1.0) cnb = cnt;
        io.cnt[c] = cnt; io.cnb[c] = cnb;
        io.pcnt[c] = io.pmid[c * L + ((int)cnt - 1)];
        io.pcnb[c] = io.pmid[c * L + ((int)cnb - 1)];
        io.rliq[c] = io.rliq[c] + l2;                                            // :231
    }
}
}  // namespace

int launch_convect_diag(int ncol, int pver, int clubb, const colphys_convect_diag_io &io, hipStream_t st)
{
    const size_t n = (size_t)ncol * (pver + 1);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(convect_diag_kernel, dim3(blocks), dim3(256), 0, st, ncol, pver, clubb, io);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}

// ---- diagnostic probe of the scalar device functions (colphys_math_probe_dev) ----------------------------
namespace {
__global__ void math_probe_kernel(int kind, int64_t n, const double *__restrict__ x, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = x[i];
        y[i] = kind == COLPHYS_PROBE_LOG ? cp_log(v) : kind == COLPHYS_PROBE_LOG10 ? cp_log10(v) : kind == COLPHYS_PROBE_EXP ? cp_exp(v) : svp_water_gg(v);
    }
}
}  // namespace

int launch_math_probe(int kind, int64_t n, const double *x, double *y, hipStream_t st)
{
    if (n <= 0) return COLPHYS_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(math_probe_kernel, dim3(blocks), dim3(256), 0, st, kind, n, x, y);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}
