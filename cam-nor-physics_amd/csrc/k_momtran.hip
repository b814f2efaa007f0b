// k_momtran.hip -- convective momentum transport, /root/reference/physics/zm_conv.F90:2315-2715, for gfx950.
//
// Only the gathered (deep) columns do work -- a few thousand of 48 602 -- so the mapping is ONE WAVE PER
// GATHERED COLUMN with the lanes along the vertical (one block per host chunk walks the chunk's gathered rows):
//   - the column's profiles (winds, mass fluxes, layer thickness) are read as contiguous rows into LDS;
//   - everything that is element-wise in k (interface winds, pressure-gradient terms, tendencies, momentum
//     flux, end-of-step winds, KE dissipation) runs one level per lane;
//   - the two genuine recurrences (in-cloud updraft wind, bottom-up :2563-2575; downdraft wind, top-down
//     :2579-2589) are walked serially by lane 0 in the reference's order, in LDS;
//   - results leave as contiguous rows.
// The chunk's block first writes the initial values of the chunk's full-layout outputs (:2429-2443).
// A thread-per-column mapping of the same work (round-1 first version) kept 13 private profiles in scratch and
// ran 80 waves for 0.33 ms.
#include "colphys_common.h"

namespace {

constexpr int NT = WAVE;

__global__ __launch_bounds__(NT) void momtran_wave_kernel(ZmDev P, int ncol, int pver, int pcols, double dt,
                                                         colphys_momtran_io io, const double *winds_v)
{
    CHAIN_PRIO();
    extern __shared__ double smem[];
    const int ic = blockIdx.x, c0 = ic * pcols;               // one block per host chunk, looping over its gathered rows
    const int il2g = io.lengath[ic];
    const size_t L = (size_t)pver, F = (size_t)ncol * L;
    const double *wind[2] = {io.winds, winds_v ? winds_v : io.winds + F};   // u, v (v may live in its own array)
    {   // initialisation of the chunk's full-layout outputs (:2429-2443): contiguous [nc][pver] runs, coalesced
        const int nc = min(pcols, ncol - c0);
        const size_t base = (size_t)c0 * L, n = (size_t)nc * L;
        for (size_t i = threadIdx.x; i < n; i += NT) {
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
    const int lane = threadIdx.x;
    const int LD = pver + 2;                                  // 1-based profiles, [0] and [pver+1] as guards
    const double mbsth = 1.e-15;

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
    for (int slot = 0; slot < il2g; ++slot) {
    const int row = c0 + slot;                                // gathered row of the flat layout
    const size_t ii = (size_t)(c0 + io.ideep[row] - 1);
    const int mx = io.mx[row];

    for (int k = lane + 1; k <= pver; k += NT) {
        const size_t g = (size_t)row * L + (k - 1);
        s_mu[k] = io.mu[g]; s_md[k] = io.md[g]; s_du[k] = io.du[g]; s_eu[k] = io.eu[g]; s_ed[k] = io.ed[g]; s_dp[k] = io.dp[g];
        for (int m = 0; m < 2; ++m) {
            s_c[m * LD + k] = wind[m][ii * L + (k - 1)];                         // :2461-2466
            s_mf[m * LD + k] = 0.0; s_wf[m * LD + k] = 0.0;                      // :2436-2438
        }
    }
    if (lane < 2) { s_mf[lane * LD + pver + 1] = 0.0; s_mf[lane * LD] = 0.0; }
    __syncthreads();

    for (int m = 0; m < 2; ++m) {
        const double *cn = s_c + m * LD;
        double *chat = s_chat + m * LD, *conu = s_conu + m * LD, *cond = s_cond + m * LD, *pgu = s_pgu + m * LD,
               *pgd = s_pgd + m * LD;
        for (int k = lane + 1; k <= pver; k += NT) {
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
    __syncthreads();

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
    __syncthreads();

    for (int m = 0; m < 2; ++m) {
        const double *chat = s_chat + m * LD, *conu = s_conu + m * LD, *cond = s_cond + m * LD;
        double *mf = s_mf + m * LD;
        for (int k = lane + 1; k <= pver; k += NT) {
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
    __syncthreads();
    for (int m = 0; m < 2; ++m) {                                                // :2658-2666
        const double *cn = s_c + m * LD, *mf = s_mf + m * LD;
        for (int k = lane + 1; k <= pver; k += NT)
            if (k >= ktm) s_wf[m * LD + k] = cn[k] - (mf[k + 1] - mf[k]) * dt / s_dp[k];
    }
    __syncthreads();
    for (int k = lane + 1; k <= pver; k += NT) {                                 // :2675-2712
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
    __syncthreads();
    }
}

}  // namespace

int launch_momtran(const ZmDev &P, int ncol, int pver, int pcols, double dt, const colphys_momtran_io &io, hipStream_t st,
                   const double *winds_v)
{
    const size_t lds = (size_t)(6 + 2 * 8) * (pver + 2) * sizeof(double);
    const int nchunks = (ncol + pcols - 1) / pcols;
    hipLaunchKernelGGL(momtran_wave_kernel, dim3(nchunks), dim3(NT), lds, st, P, ncol, pver, pcols, dt, io, winds_v);
    return (hipGetLastError() == hipSuccess) ? COLPHYS_OK : COLPHYS_ERR_HIP;
}
