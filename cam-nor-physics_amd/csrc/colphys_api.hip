// colphys_api.hip -- the C ABI of include/colphys.h: argument checks, device state, workspaces, and the
// host-layout (chunk) entry points the ISO_C_BINDING shim calls.  No CPU fallback anywhere: without a HIP
// device every compute entry point returns COLPHYS_ERR_NODEVICE.
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <limits>
#include <vector>

#include "colphys_common.h"

namespace {

std::mutex g_mu;                  // the library's state: parameters, tables, grow-only device work arrays
// Serialises the host-layout (*_chunks) calls from end to end: they share the staging arenas, the host thread pool and --
// through the *_dev calls they make on the null stream -- the device work arrays.  A threaded chunk loop (the reference's
// OpenMP loop over chunks, physpkg.F90:1147) may call them concurrently; the calls then run one after another.
std::mutex g_stage_mu;
thread_local std::string g_err;    // per calling thread: a concurrent caller's failure must not replace this thread's message
bool g_init = false;
int g_device = -1;
ZmDev g_P;
double *g_estbl = nullptr;          // device, [PLENEST+1], 1-based
double *g_rz = nullptr;             // device, {rair, zvir}: what rairv / zvirv hold everywhere without WACCM-X (a caller that passes
                                    // NULL for both fields of geopotential_t / zm_conv_tend gets these; colphys.h)
AeroDev g_A;
bool g_aero_set = false;
std::vector<void *> g_aero_bufs;
std::vector<int> g_hf_nspec;        // host copy of the tables' species counts per mode (the host face names its fields by them)

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(COLPHYS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// grow-only device / pinned-host arenas
struct Arena {
    void *p = nullptr;
    size_t cap = 0;
    bool host = false;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) { if (host) hipHostFree(p); else hipFree(p); p = nullptr; cap = 0; }
        const size_t want = bytes + bytes / 8;
        hipError_t e = host ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return -1; }
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) { if (host) hipHostFree(p); else hipFree(p); }
        p = nullptr; cap = 0;
    }
};
Arena g_ws_arena;        // ZmWorkspace
Arena g_dev_arena;       // device staging for *_chunks
Arena g_pin_arena;       // pinned host staging for *_chunks
Arena g_tend_arena;      // work arrays of colphys_zm_conv_tend_dev
Arena g_ct_arena;        // convtran: gathered-row list, its count, il2g per chunk

// carve helper
struct Carver {
    char *base; size_t off = 0;
    explicit Carver(void *b) : base((char *)b) {}
    template <class T> T *take(size_t n)
    {
        off = (off + 255) & ~size_t(255);
        T *r = (T *)(base + off);
        off += n * sizeof(T);
        return r;
    }
};

size_t zm_ws_bytes(int ncol, int pver)
{
    return 256 * 32 + sizeof(int) * (size_t)ncol * 10 + sizeof(double) * (size_t)ncol * 6 + sizeof(double) * (size_t)ncol * pver * 8 + 4096;
}
ZmWorkspace zm_ws_carve(void *base, int ncol, int pver)
{
    Carver c(base);
    ZmWorkspace w;
    w.count = c.take<int>(64);
    w.list = c.take<int>(ncol);
    w.cape1 = c.take<double>(ncol); w.cin1 = c.take<double>(ncol);
    w.lel1 = c.take<int>(ncol); w.mx1 = c.take<int>(ncol); w.deep = c.take<int>(ncol);
    w.jt_f = c.take<int>(ncol); w.maxg_f = c.take<int>(ncol); w.dsub_f = c.take<double>(ncol);
    const size_t n = (size_t)ncol * pver;
    w.mu_f = c.take<double>(n); w.md_f = c.take<double>(n); w.du_f = c.take<double>(n);
    w.eu_f = c.take<double>(n); w.ed_f = c.take<double>(n); w.dp_f = c.take<double>(n);
    w.dmpdz2 = c.take<double>(ncol); w.cape2 = c.take<double>(ncol); w.tl2 = c.take<double>(ncol);
    w.lcl2 = c.take<int>(ncol); w.lel2 = c.take<int>(ncol); w.mx2 = c.take<int>(ncol); w.retrig = c.take<int>(ncol);
    w.tp2 = c.take<double>(n); w.qstp2 = c.take<double>(n);
    return w;
}

// ---- profiling with HIP events on the launch stream ---------------------------------------------------
bool g_prof = false;
struct Pending { int kernel; hipEvent_t a, b; };
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_evpool;
double g_ms[COLPHYS_K_COUNT];
int64_t g_n[COLPHYS_K_COUNT];

hipEvent_t ev_get()
{
    if (!g_evpool.empty()) { hipEvent_t e = g_evpool.back(); g_evpool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
void prof_drain()
{
    for (auto &p : g_pending) {
        float ms = 0.f;
        hipEventSynchronize(p.b);
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_ms[p.kernel] += ms; g_n[p.kernel] += 1; }
        g_evpool.push_back(p.a);
        g_evpool.push_back(p.b);
    }
    g_pending.clear();
}
struct ProfScope {
    int kernel; hipStream_t st; hipEvent_t a{}, b{}; bool on;
    ProfScope(int k, hipStream_t s) : kernel(k), st(s), on(g_prof)
    {
        if (on) { a = ev_get(); b = ev_get(); hipEventRecord(a, st); }
    }
    ~ProfScope()
    {
        if (on) { hipEventRecord(b, st); g_pending.push_back({kernel, a, b}); }
    }
};

int check_ready()
{
    if (!g_init) return fail(COLPHYS_ERR_NOTINIT, "colphys_init has not been called");
    // HIP's current device is per host thread: a thread other than the one that called colphys_init (an OpenMP worker of the
    // reference's chunk loop) starts on device 0.  The calling thread is left on the engine's device.
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_device)
        if (hipSetDevice(g_device) != hipSuccess) return fail(COLPHYS_ERR_HIP, "hipSetDevice(engine device) failed on the calling thread");
    return COLPHYS_OK;
}
int check_dims(int ncol, int pver)
{
    if (ncol <= 0) return fail(COLPHYS_ERR_ARG, "ncol must be > 0");
    if (pver < 4 || pver > COLPHYS_MAX_PVER) return fail(COLPHYS_ERR_ARG, "pver out of range [4, COLPHYS_MAX_PVER]");
    return COLPHYS_OK;
}

// host restatement of the table construction (spec: oracle/stubs/l0_arithmetic.F90), pow() as the Fortran `**`
double h_svp_water(double t)
{
    const double tboil = 373.16;
    return std::pow(10.0, -7.90298 * (tboil / t - 1.0) + 5.02808 * std::log10(tboil / t)
                              - 1.3816e-7 * (std::pow(10.0, 11.344 * (1.0 - t / tboil)) - 1.0)
                              + 8.1328e-3 * (std::pow(10.0, -3.49149 * (tboil / t - 1.0)) - 1.0)
                              + std::log10(1013.246)) * 100.0;
}
double h_svp_ice(double t)
{
    const double h2otrip = 273.16;
    return std::pow(10.0, -9.09718 * (h2otrip / t - 1.0) - 3.56654 * std::log10(h2otrip / t)
                              + 0.876793 * (1.0 - t / h2otrip) + std::log10(6.1071)) * 100.0;
}
double h_svp_trans(double t, double tmelt)
{
    const double ttrice = 20.0;
    double es = 0.0;
    if (t >= (tmelt - ttrice)) es = h_svp_water(t);
    if (t < tmelt) {
        const double esice = h_svp_ice(t);
        const double weight = ((tmelt - t) > ttrice) ? 1.0 : (tmelt - t) / ttrice;
        es = weight * esice + (1.0 - weight) * es;
    }
    return es;
}

}  // namespace

extern "C" {

int colphys_abi_version(void) { return COLPHYS_ABI_VERSION; }
const char *colphys_last_error(void) { return g_err.c_str(); }

int colphys_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int colphys_is_initialized(void) { return g_init ? 1 : 0; }

void colphys_zm_params_defaults(colphys_zm_params *p)
{
    p->tiedke_lnd = 1.0; p->entrmn = 2e-4; p->alfadet = 0.1; p->plclmin = 6.e2; p->cin_threshd = 0.33;
    p->parcel_hscale = 0.5; p->tentrm = -p->dmpdz_param; p->cam3 = 0; p->reserved = 0;
}

int colphys_init(int device, const colphys_zm_params *p)
{
    std::lock_guard<std::mutex> slk(g_stage_mu);        // not while a *_chunks call is in flight (order: stage, then state)
    std::lock_guard<std::mutex> lk(g_mu);
    if (!p) return fail(COLPHYS_ERR_ARG, "null params");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(COLPHYS_ERR_NODEVICE, "no HIP device (the engine has no CPU path)");
    if (device < 0 || device >= n) return fail(COLPHYS_ERR_ARG, "device index out of range");
    if (p->zmconv_microp) return fail(COLPHYS_ERR_UNSUPPORTED, "zmconv_microp=.true. (zm_mphy) is outside the hot-path scope");
    if (p->zm_org) return fail(COLPHYS_ERR_UNSUPPORTED, "zmconv_org=.true. is outside the hot-path scope");
    if (p->cam3) return fail(COLPHYS_ERR_UNSUPPORTED, "cam3 non-dilute buoyan is outside the hot-path scope");
    if (p->num_cin < 1 || p->num_cin > 5) return fail(COLPHYS_ERR_ARG, "NUM_CIN must be in 1..5 (zm_conv.F90:200)");
    if (p->limcnv < 1) return fail(COLPHYS_ERR_ARG, "limcnv must be >= 1");
    HIPCHK(hipSetDevice(device));
    g_device = device;

    ZmDev &P = g_P;
    P.cpair = p->cpair; P.epsilo = p->epsilo; P.gravit = p->gravit; P.latice = p->latice; P.latvap = p->latvap;
    P.tmelt = p->tmelt; P.rair = p->rair; P.cpwv = p->cpwv; P.cpliq = p->cpliq; P.rh2o = p->rh2o; P.cpvir = p->cpvir;
    P.zvir = p->zvir;
    P.c0_lnd = p->c0_lnd; P.c0_ocn = p->c0_ocn; P.ke = p->ke; P.ke_lnd = p->ke_lnd; P.momcu = p->momcu; P.momcd = p->momcd;
    P.tiedke_add = p->tiedke_add; P.capelmt = p->capelmt; P.tau = p->tau;
    P.tiedke_lnd = p->tiedke_lnd; P.entrmn = p->entrmn; P.alfadet = p->alfadet; P.tentrm = p->tentrm;
    P.plclmin = p->plclmin; P.cin_threshd = p->cin_threshd; P.parcel_hscale = p->parcel_hscale;
    // zm_convi :145-155
    P.tfreez = p->tmelt; P.eps1 = p->epsilo; P.rl = p->latvap; P.cpres = p->cpair; P.rgrav = 1.0 / p->gravit;
    P.rgas = p->rair; P.grav = p->gravit; P.cp = P.cpres; P.dcol = (p->cpliq - p->cpwv) / p->latvap;
    P.omeps = 1.0 - p->epsilo;
    P.ln_tfreez = std::log(P.tfreez);
    P.limcnv = p->limcnv; P.num_cin = p->num_cin; P.zm_org = p->zm_org; P.no_deep_pbl = p->no_deep_pbl;
    P.lparcel_pbl = p->lparcel_pbl;

    std::vector<double> tbl(COLPHYS_PLENEST + 1, 0.0);
    for (int i = 1; i <= COLPHYS_PLENEST; ++i) tbl[i] = h_svp_trans(COLPHYS_TMIN_TBL + (double)(i - 1), p->tmelt);
    if (!g_estbl) HIPCHK(hipMalloc((void **)&g_estbl, sizeof(double) * (COLPHYS_PLENEST + 1)));
    HIPCHK(hipMemcpy(g_estbl, tbl.data(), sizeof(double) * (COLPHYS_PLENEST + 1), hipMemcpyHostToDevice));
    {
        const double rz[2] = {p->rair, p->zvir};
        if (!g_rz) HIPCHK(hipMalloc((void **)&g_rz, sizeof(rz)));
        HIPCHK(hipMemcpy(g_rz, rz, sizeof(rz), hipMemcpyHostToDevice));
    }
    g_pin_arena.host = true;
    g_init = true;
    return COLPHYS_OK;
}

int colphys_finalize(void)
{
    std::lock_guard<std::mutex> slk(g_stage_mu);        // not while a *_chunks call is in flight (order: stage, then state)
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init) {
        hipDeviceSynchronize();
        prof_drain();
        for (auto e : g_evpool) hipEventDestroy(e);
        g_evpool.clear();
        if (g_estbl) hipFree(g_estbl);
        g_estbl = nullptr;
        if (g_rz) hipFree(g_rz);
        g_rz = nullptr;
        for (void *b : g_aero_bufs) hipFree(b);
        g_aero_bufs.clear();
        g_aero_set = false;
        g_ws_arena.release(); g_dev_arena.release(); g_pin_arena.release(); g_tend_arena.release(); g_ct_arena.release();
    }
    g_init = false;
    return COLPHYS_OK;
}

int colphys_profile_enable(int on)
{
    g_prof = on != 0;
    return COLPHYS_OK;
}
int colphys_profile_reset(void)
{
    if (g_init) { hipDeviceSynchronize(); prof_drain(); }
    for (int i = 0; i < COLPHYS_K_COUNT; ++i) { g_ms[i] = 0.0; g_n[i] = 0; }
    return COLPHYS_OK;
}
int colphys_profile_get(int kernel, double *avg_ms, int64_t *launches)
{
    if (kernel < 0 || kernel >= COLPHYS_K_COUNT) return fail(COLPHYS_ERR_ARG, "bad kernel id");
    prof_drain();
    if (avg_ms) *avg_ms = g_n[kernel] ? g_ms[kernel] / (double)g_n[kernel] : 0.0;
    if (launches) *launches = g_n[kernel];
    return COLPHYS_OK;
}

// =======================================================================================================
// device-resident entry points
// =======================================================================================================
int colphys_geopotential_t_dev(int ncol, int pver, int dycore_kind, int nq, int nspecies, const double *piln,
                               const double *pmln, const double *pint, const double *pmid, const double *pdel,
                               const double *rpdel, const double *t, const double *q, const double *rair, double gravit,
                               const double *zvir, double *zi, double *zm, void *stream)
{
    (void)pmln;
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (dycore_kind < 0 || dycore_kind > 2) return fail(COLPHYS_ERR_ARG, "bad dycore_kind");
    if (nq < 1 || nspecies < 1 || nspecies > nq) return fail(COLPHYS_ERR_ARG, "need 1 <= nspecies <= nq");
    if (!t || !q || !zi || !zm) return fail(COLPHYS_ERR_ARG, "null pointer");
    if ((rair == nullptr) != (zvir == nullptr)) return fail(COLPHYS_ERR_ARG, "rair and zvir: both fields, or both NULL (the engine's constants)");
    if (dycore_kind == COLPHYS_DYCORE_FV ? (!piln || !pint || !rpdel) : (!pdel || !pmid))
        return fail(COLPHYS_ERR_ARG, "null pressure field for this dycore_kind");
    ProfScope ps(COLPHYS_K_GEOPOTENTIAL, (hipStream_t)stream);
    const int uni = rair == nullptr;
    rc = launch_geopotential_t(ncol, pver, dycore_kind, nq, nspecies, piln, pint, pmid, pdel, rpdel, t, q, uni ? g_rz : rair, gravit,
                               uni ? g_rz + 1 : zvir, zi, zm, (hipStream_t)stream, uni);
    return rc ? fail(rc, "geopotential_t launch failed") : COLPHYS_OK;
}

int colphys_geopotential_dse_dev(int ncol, int pver, int dycore_kind, int nq, int nspecies, const double *piln, const double *pmln,
                                 const double *pint, const double *pmid, const double *pdel, const double *rpdel, const double *dse,
                                 const double *q, const double *phis, const double *rair, double gravit, const double *cpair,
                                 const double *zvir, double *t, double *zi, double *zm, void *stream)
{
    (void)pmln;
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (dycore_kind < 0 || dycore_kind > 2) return fail(COLPHYS_ERR_ARG, "bad dycore_kind");
    if (nq < 1 || nspecies < 1 || nspecies > nq) return fail(COLPHYS_ERR_ARG, "need 1 <= nspecies <= nq");
    if (!dse || !q || !phis || !rair || !cpair || !zvir || !t || !zi || !zm) return fail(COLPHYS_ERR_ARG, "null pointer");
    if (dycore_kind == COLPHYS_DYCORE_FV ? (!piln || !pint || !rpdel) : (!pdel || !pmid))
        return fail(COLPHYS_ERR_ARG, "null pressure field for this dycore_kind");
    ProfScope ps(COLPHYS_K_GEOPOTENTIAL, (hipStream_t)stream);
    rc = launch_geopotential_dse(ncol, pver, dycore_kind, nspecies, piln, pint, pmid, pdel, rpdel, dse, q, phis, rair, gravit, cpair,
                                 zvir, t, zi, zm, (hipStream_t)stream);
    return rc ? fail(rc, "geopotential_dse launch failed") : COLPHYS_OK;
}

static int zm_convr_impl(int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io *io, void *stream,
                         const ZmTendDiag *diag, Arena *ws_arena = &g_ws_arena)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pcols < 1 || pcols > 1024) return fail(COLPHYS_ERR_ARG, "pcols must be in 1..1024");
    if (g_P.limcnv + 1 > pver) return fail(COLPHYS_ERR_ARG, "limcnv must be < pver");
    const void *ptrs[] = {io->t, io->qh, io->pap, io->dpp, io->zm, io->paph, io->zi, io->geos, io->pblh, io->tpert,
                          io->landfrac, io->qtnd, io->heat, io->cme, io->zdu, io->rprd, io->dlf, io->ql, io->dif, io->dnlf,
                          io->dnif, io->eurt, io->mcon, io->pflx, io->prec, io->jctop, io->jcbot, io->cape, io->rliq,
                          io->rice, io->mu, io->md, io->du, io->eu, io->ed, io->dp, io->dsubcld, io->jt, io->maxg,
                          io->ideep, io->lengath, io->status};
    for (const void *p : ptrs)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_zm_convr_io");
    hipStream_t st = (hipStream_t)stream;
    ZmWorkspace ws;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (ws_arena->reserve(zm_ws_bytes(ncol, pver))) return fail(COLPHYS_ERR_HIP, "workspace allocation failed");
        ws = zm_ws_carve(ws_arena->p, ncol, pver);
    }
    hipEvent_t ev[8];
    if (g_prof) for (auto &e : ev) e = ev_get();
    rc = launch_zm_convr(g_P, ncol, pver, pcols, delt, *io, ws, st, g_prof ? ev : nullptr, diag);
    if (g_prof) {
        g_pending.push_back({COLPHYS_K_ZM_TRIGGER, ev[0], ev[1]});
        g_pending.push_back({COLPHYS_K_ZM_PLUME1, ev[2], ev[3]});
        g_pending.push_back({COLPHYS_K_ZM_ASCENT2, ev[4], ev[5]});
        g_pending.push_back({COLPHYS_K_ZM_PLUME2, ev[6], ev[7]});
    }
    return rc ? fail(rc, "zm_convr launch failed") : COLPHYS_OK;
}

int colphys_zm_convr_dev(int ncol, int pver, int pcols, double delt, const colphys_zm_convr_io *io, void *stream)
{
    return zm_convr_impl(ncol, pver, pcols, delt, io, stream, nullptr);
}

int colphys_zm_conv_evap_dev(int ncol, int pver, double deltat, const colphys_zm_conv_evap_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    const void *ptrs[] = {io->t, io->pmid, io->pdel, io->q, io->prdprec, io->cldfrc, io->landfrac, io->prec, io->tend_s,
                          io->tend_s_snwprd, io->tend_s_snwevmlt, io->tend_q, io->ntprprd, io->ntsnprd, io->flxprec,
                          io->flxsnow, io->snow};
    for (const void *p : ptrs)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_zm_conv_evap_io");
    ProfScope ps(COLPHYS_K_ZM_EVAP, (hipStream_t)stream);
    rc = launch_zm_conv_evap(g_P, g_estbl, ncol, pver, deltat, *io, (hipStream_t)stream);
    return rc ? fail(rc, "zm_conv_evap launch failed") : COLPHYS_OK;
}

int colphys_momtran_dev(int ncol, int pver, int pcols, double dt, const colphys_momtran_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pcols < 1 || pcols > 1024) return fail(COLPHYS_ERR_ARG, "pcols must be in 1..1024");
    const void *ptrs[] = {io->winds, io->mu, io->md, io->du, io->eu, io->ed, io->dp, io->dsubcld, io->jt, io->mx, io->ideep,
                          io->lengath, io->dqdt, io->pguall, io->pgdall, io->icwu, io->icwd, io->seten};
    for (const void *p : ptrs)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_momtran_io");
    ProfScope ps(COLPHYS_K_MOMTRAN, (hipStream_t)stream);
    rc = launch_momtran(g_P, ncol, pver, pcols, dt, *io, (hipStream_t)stream);
    return rc ? fail(rc, "momtran launch failed") : COLPHYS_OK;
}

static std::mutex g_ct_mu;       // the convtran row-list workspace
static int convtran_ws(int ncol, int pcols, int **rows, int **count, int **il2g)
{
    std::lock_guard<std::mutex> lk(g_ct_mu);
    const size_t nchunks = ((size_t)ncol + pcols - 1) / pcols;
    if (g_ct_arena.reserve(sizeof(int) * ((size_t)ncol + nchunks + 64))) return fail(COLPHYS_ERR_HIP, "convtran workspace allocation failed");
    *count = (int *)g_ct_arena.p;
    *rows = *count + 16;
    *il2g = *rows + ncol;
    return COLPHYS_OK;
}

int colphys_convtran_dev(int ncol, int pver, int pcols, int ncnst, double dt, const colphys_convtran_io *io, void *stream)
{
    (void)dt;
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pcols < 1 || pcols > 1024 || ncnst < 1) return fail(COLPHYS_ERR_ARG, "bad pcols / ncnst");
    const void *ptrs[] = {io->q, io->fracis, io->mu, io->md, io->du, io->eu, io->ed, io->dp, io->dpdry, io->jt, io->mx, io->ideep,
                          io->lengath, io->doconvtran, io->isdry, io->dqdt};
    for (const void *p : ptrs)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_convtran_io");
    int *rows, *count, *il2g;
    if ((rc = convtran_ws(ncol, pcols, &rows, &count, &il2g))) return rc;
    ProfScope ps(COLPHYS_K_CONVTRAN, (hipStream_t)stream);
    rc = launch_convtran(ncol, pver, pcols, ncnst, *io, (hipStream_t)stream, nullptr, rows, count, il2g);
    return rc ? fail(rc, "convtran launch failed") : COLPHYS_OK;
}

int colphys_zm_conv_tend_2_dev(int ncol, int pver, int pcols, int ncnst, double ztodt, const colphys_zm_conv_tend_2_io *io,
                               void *stream)
{
    (void)ztodt;                                    // convtran's dt only enters the microphysics branch (zm_conv.F90:2254)
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pcols < 1 || pcols > 1024 || ncnst < 1) return fail(COLPHYS_ERR_ARG, "bad pcols / ncnst");
    const void *ptrs[] = {io->q, io->fracis, io->mu, io->md, io->du, io->eu, io->ed, io->dp, io->jt, io->maxg, io->ideep, io->lq,
                          io->isdry, io->ptend_q};
    for (const void *p : ptrs)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_zm_conv_tend_2_io");
    bool any = false, anydry = false;
    for (int m = 1; m < ncnst; ++m) {
        any |= io->lq[m] != 0;
        anydry |= io->lq[m] != 0 && io->isdry[m] != 0;
    }
    if (!any) return COLPHYS_OK;                                                 // zm_conv_intr.F90:1005
    if (anydry && !io->pdeldry) return fail(COLPHYS_ERR_ARG, "a 'dry' constituent is transported but pdeldry is null");
    colphys_convtran_io c{};
    c.q = io->q; c.fracis = io->fracis; c.mu = io->mu; c.md = io->md; c.du = io->du; c.eu = io->eu; c.ed = io->ed; c.dp = io->dp;
    c.dpdry = io->dp;                               // placeholder, only read for 'dry' constituents, which use pdeldry
    c.jt = io->jt; c.mx = io->maxg; c.ideep = io->ideep; c.lengath = nullptr;    // count(ideep > 0), :993
    c.doconvtran = io->lq; c.isdry = io->isdry; c.dqdt = io->ptend_q;
    int *rows, *count, *il2g;
    if ((rc = convtran_ws(ncol, pcols, &rows, &count, &il2g))) return rc;
    ProfScope ps(COLPHYS_K_CONVTRAN, (hipStream_t)stream);
    rc = launch_convtran(ncol, pver, pcols, ncnst, c, (hipStream_t)stream, anydry ? io->pdeldry : nullptr, rows, count, il2g);
    return rc ? fail(rc, "convtran launch failed") : COLPHYS_OK;
}

int colphys_derived_fields_dev(int ncol, int pver, int ncnst, int dycore_kind, int nspecies, double gravit,
                               const colphys_derived_fields_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pver < 2) return fail(COLPHYS_ERR_ARG, "the bottom-layer fix needs two levels");
    if (dycore_kind < 0 || dycore_kind > 2) return fail(COLPHYS_ERR_ARG, "bad dycore_kind");
    if (ncnst < 1 || nspecies < 1 || nspecies > ncnst) return fail(COLPHYS_ERR_ARG, "need 1 <= nspecies <= ncnst");
    const void *need[] = {io->pint, io->lnpint, io->t, io->q, io->qmin, io->phis, io->cpair, io->rair, io->zvir, io->pdel, io->rpdel,
                          io->pmid, io->lnpmid, io->zi, io->zm, io->s};
    for (const void *p : need)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_derived_fields_io");
    rc = launch_derived_fields(ncol, pver, ncnst, dycore_kind, nspecies, gravit, *io, (hipStream_t)stream);
    return rc ? fail(rc, "derived_fields launch failed") : COLPHYS_OK;
}

int colphys_dry_mass_fields_dev(int ncol, int pver, int ncnst, const colphys_dry_mass_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (ncnst < 1) return fail(COLPHYS_ERR_ARG, "need ncnst >= 1 (water vapour first)");
    const void *need[] = {io->pint, io->pdel, io->q, io->isdry, io->qmin, io->pdeldry, io->rpdeldry, io->pmiddry, io->lnpmiddry,
                          io->pintdry, io->lnpintdry, io->psdry};
    for (const void *p : need)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_dry_mass_io");
    rc = launch_dry_mass_fields(ncol, pver, ncnst, *io, (hipStream_t)stream);
    return rc ? fail(rc, "dry_mass_fields launch failed") : COLPHYS_OK;
}

int colphys_convect_diagnostics_dev(int ncol, int pver, int shallow_is_clubb, const colphys_convect_diag_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    const void *need[] = {io->cmfmc, io->qc, io->rliq, io->cnt, io->cnb, io->rprddp, io->pmid, io->rprdtot, io->pcnt, io->pcnb};
    for (const void *p : need)
        if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_convect_diag_io");
    if (!shallow_is_clubb) {
        const void *sh[] = {io->cmfmc2, io->qc2, io->rprdsh, io->rliq2, io->cnt2, io->cnb2};
        for (const void *p : sh)
            if (!p) return fail(COLPHYS_ERR_ARG, "a shallow-scheme field is null and shallow_is_clubb is 0");
    }
    rc = launch_convect_diag(ncol, pver, shallow_is_clubb ? 1 : 0, *io, (hipStream_t)stream);
    return rc ? fail(rc, "convect_diagnostics launch failed") : COLPHYS_OK;
}

int colphys_physics_update_dev(int ncol, int pver, double dt, double cpair, double qmin, const double *t_in,
                               const double *s_tend, double *t_out, const double *q_in, const double *q_tend, double *q_out,
                               const double *u_in, const double *u_tend, double *u_out, const double *v_in,
                               const double *v_tend, double *v_out, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    auto triple_ok = [](const void *a, const void *b, const void *c) { return (a && b && c) || (!a && !b && !c); };
    if (!triple_ok(t_in, s_tend, t_out) || !triple_ok(q_in, q_tend, q_out) || !triple_ok(u_in, u_tend, u_out) ||
        !triple_ok(v_in, v_tend, v_out))
        return fail(COLPHYS_ERR_ARG, "each (in, tend, out) triple must be all set or all NULL");
    if (t_out && !(cpair > 0.0)) return fail(COLPHYS_ERR_ARG, "cpair must be > 0");
    ProfScope ps(COLPHYS_K_UPDATE, (hipStream_t)stream);
    rc = launch_physics_update(ncol, pver, dt, cpair, qmin, t_in, s_tend, t_out, q_in, q_tend, q_out, u_in, u_tend, u_out,
                               v_in, v_tend, v_out, (hipStream_t)stream);
    return rc ? fail(rc, "physics_update launch failed") : COLPHYS_OK;
}

// ---- zm_conv_tend on resident data (zm_conv_intr.F90:441-905) ---------------------------------------------------
// ws_arena / tend_arena: the work arrays of this call.  The public entry uses the library's own pair (calls ordered by the
// caller); the host face runs several column slices on several streams at once and gives each stream a pair of its own.
static int zm_conv_tend_impl(int ncol, int pver, int pcols, double ztodt, int dycore_kind, double qmin,
                             const colphys_zm_conv_tend_io *io, void *stream, Arena *ws_arena, Arena *tend_arena)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    if (pcols < 1 || pcols > 1024) return fail(COLPHYS_ERR_ARG, "pcols must be in 1..1024");
    {
        const void *need[] = {io->t, io->q, io->u, io->v, io->pmid, io->pint, io->pdel, io->rpdel, io->lnpint, io->lnpmid, io->zm,
                              io->zi, io->phis, io->ps, io->pblh, io->tpert, io->landfrac, io->cld,
                              io->ptend_s, io->ptend_q, io->ptend_u, io->ptend_v, io->prec, io->snow, io->rliq, io->rice,
                              io->jctop, io->jcbot, io->cape, io->ql, io->rprd, io->evapcdp, io->dlf, io->mconzm, io->flxprec,
                              io->flxsnow, io->mu, io->md, io->du, io->eu, io->ed, io->dp, io->dsubcld, io->jt, io->maxg,
                              io->ideep, io->lengath, io->status};
        for (const void *p : need)
            if (!p) return fail(COLPHYS_ERR_ARG, "null pointer in colphys_zm_conv_tend_io");
        if ((io->rair == nullptr) != (io->zvir == nullptr))
            return fail(COLPHYS_ERR_ARG, "colphys_zm_conv_tend_io: rair and zvir are both fields, or both NULL (the engine's constants)");
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t nM = (size_t)ncol * pver, nI = (size_t)ncol * (pver + 1);

    // work arrays: ptend_loc of the three stages, what zm_conv_tend only hands to outfld, and state1
    std::unique_lock<std::mutex> lk(g_mu);
    const size_t bytes = 8 * (33 * nM + 4 * nI + 8 * (size_t)ncol) + 64 * 256;
    if (tend_arena->reserve(bytes)) return fail(COLPHYS_ERR_HIP, "work-array allocation failed");
    Carver w(tend_arena->p);
    lk.unlock();
    // (a work array the caller asked for by name -- ABI 4 history outputs -- lives in the caller's buffer instead)
    auto mid = [&](double *user, size_t n) { return user ? user : w.take<double>(n); };
    double *heat = mid(io->zmdt_s, nM), *qtnd = mid(io->zmdq, nM), *cme = mid(io->cme, nM), *zdu = mid(io->zdu, nM),
           *dif = mid(io->dif, nM), *dnlf = w.take<double>(nM), *dnif = w.take<double>(nM), *eurt = mid(io->eurt, nM),
           *mcon = w.take<double>(nI), *pflx = mid(io->pflx, nI);
    double *tend_s = mid(io->evap_s, nM), *snwprd = mid(io->evap_s_snwprd, nM), *snwevmlt = mid(io->evap_s_snwevmlt, nM),
           *tend_q = io->evapcdp, *ntprprd = mid(io->ntprprd, nM), *ntsnprd = mid(io->ntsnprd, nM);
    double *wtend = w.take<double>(2 * nM), *pguall = mid(io->pguall, 2 * nM), *pgdall = mid(io->pgdall, 2 * nM),
           *icwu = mid(io->icwu, 2 * nM), *icwd = mid(io->icwd, 2 * nM), *seten = mid(io->seten, nM);
    double *t1 = io->t1 ? io->t1 : w.take<double>(nM), *q1 = io->q1 ? io->q1 : w.take<double>(nM);
    double *u1 = io->u1 ? io->u1 : w.take<double>(nM), *v1 = io->v1 ? io->v1 : w.take<double>(nM);
    double *zi1 = io->zi1 ? io->zi1 : w.take<double>(nI), *zm1 = io->zm1 ? io->zm1 : w.take<double>(nM);

    // ---- zm_convr on the incoming state, delt = ztodt/2 (:662-675) ------------------------------------------
    colphys_zm_convr_io c{};
    c.t = io->t; c.qh = io->q; c.pap = io->pmid; c.dpp = io->pdel; c.zm = io->zm; c.paph = io->pint; c.zi = io->zi;
    c.geos = io->phis; c.pblh = io->pblh; c.tpert = io->tpert; c.landfrac = io->landfrac;
    c.qtnd = qtnd; c.heat = heat; c.cme = cme; c.zdu = zdu; c.rprd = io->rprd; c.dlf = io->dlf; c.ql = io->ql; c.dif = dif;
    c.dnlf = dnlf; c.dnif = dnif; c.eurt = eurt; c.mcon = mcon; c.pflx = pflx;
    c.prec = io->prec; c.jctop = io->jctop; c.jcbot = io->jcbot; c.cape = io->cape; c.rliq = io->rliq; c.rice = io->rice;
    c.mu = io->mu; c.md = io->md; c.du = io->du; c.eu = io->eu; c.ed = io->ed; c.dp = io->dp; c.dsubcld = io->dsubcld;
    c.jt = io->jt; c.maxg = io->maxg; c.ideep = io->ideep; c.lengath = io->lengath; c.status = io->status;
    // ... with the mass-flux units and the gathered diagnostics (:685-731) done by its gather kernel
    ZmTendDiag dg{};
    dg.enabled = 1; dg.gravit = g_P.gravit; dg.mconzm = io->mconzm; dg.ps = io->ps; dg.pmid = io->pmid;
    dg.freqzm = io->freqzm; dg.pcont = io->pcont; dg.pconb = io->pconb; dg.mu_out = io->mu_out; dg.md_out = io->md_out;
    if ((rc = zm_convr_impl(ncol, pver, pcols, 0.5 * ztodt, &c, stream, &dg, ws_arena))) return rc;

    auto update = [&](const double *ti, const double *s, const double *qi, const double *qt, const double *ui, const double *ut,
                      const double *vi, const double *vt, int first) -> int {
        // physics_ptend_sum + physics_update (tendency application) + geopotential_t, physics_types.F90:292-340, 425-441
        ProfScope ps(COLPHYS_K_UPDATE, st);
        int r = launch_physics_update(ncol, pver, ztodt, g_P.cpair, qmin, ti, s, t1, qi, qt, qt ? q1 : nullptr, ui, ut,
                                      ut ? u1 : nullptr, vi, vt, vt ? v1 : nullptr, st, io->ptend_s, io->ptend_q, io->ptend_u,
                                      io->ptend_v, first);
        return r;
    };
    auto geopotential = [&]() -> int {
        return colphys_geopotential_t_dev(ncol, pver, dycore_kind, 1, 1, io->lnpint, io->lnpmid, io->pint, io->pmid, io->pdel,
                                          io->rpdel, t1, q1, io->rair, g_P.gravit, io->zvir, zi1, zm1, stream);
    };

    // ---- ptend_all = ptend_loc; state1 = state + ptend_loc dt (:737-739) -----------------------------------
    if ((rc = update(io->t, heat, io->q, qtnd, nullptr, nullptr, nullptr, nullptr, 1 | 2))) return fail(rc, "physics_update launch failed");

    colphys_zm_conv_evap_io e{};
    e.t = t1; e.pmid = io->pmid; e.pdel = io->pdel; e.q = q1; e.prdprec = io->rprd; e.cldfrc = io->cld; e.landfrac = io->landfrac;
    e.prec = io->prec; e.tend_s = tend_s; e.tend_s_snwprd = snwprd; e.tend_s_snwevmlt = snwevmlt; e.tend_q = tend_q;
    e.ntprprd = ntprprd; e.ntsnprd = ntsnprd; e.flxprec = io->flxprec; e.flxsnow = io->flxsnow; e.snow = io->snow;
    colphys_momtran_io m{};
    m.winds = io->u; m.mu = io->mu; m.md = io->md; m.du = io->du; m.eu = io->eu; m.ed = io->ed; m.dp = io->dp;
    m.dsubcld = io->dsubcld; m.jt = io->jt; m.mx = io->maxg; m.ideep = io->ideep; m.lengath = io->lengath;
    m.dqdt = wtend; m.pguall = pguall; m.pgdall = pgdall; m.icwu = icwu; m.icwd = icwd; m.seten = seten;

    // A small rank -- a strong-scaled share of the grid, a chunk-sized call -- is all launch latency from here on: the three
    // stages that do not depend on each other go out as ONE launch, the last two updates and the last two geopotential_t as one
    // each (colphys_common.h ZmTailSmall; bit-identical).  COLPHYS_TAIL_FUSED = 0 / 1 forces the choice (tests, A/B); the
    // per-kernel timers of the measurement mode need the separate launches.
    bool fused_tail = ncol <= 20000;
    if (const char *ev = getenv("COLPHYS_TAIL_FUSED")) fused_tail = atoi(ev) != 0;
    if (g_prof) fused_tail = false;
    if (fused_tail) {
        double *t2 = w.take<double>(nM), *zi_s = w.take<double>(nI), *zm_s = w.take<double>(nM);
        ZmTailSmall T{};
        T.ncol = ncol; T.pver = pver; T.pcols = pcols; T.dycore_kind = dycore_kind; T.ztodt = ztodt; T.qmin = qmin;
        T.lnpint = io->lnpint; T.pint = io->pint; T.pmid = io->pmid; T.pdel = io->pdel; T.rpdel = io->rpdel;
        T.rz_uniform = io->rair == nullptr;
        T.rair = T.rz_uniform ? g_rz : io->rair; T.zvir = T.rz_uniform ? g_rz + 1 : io->zvir;
        T.t1 = t1; T.q1 = q1; T.u1 = u1; T.v1 = v1; T.zi1 = zi1; T.zm1 = zm1; T.t2 = t2; T.zi_scratch = zi_s; T.zm_scratch = zm_s;
        T.u = io->u; T.v = io->v; T.evap = e; T.mom = m;
        T.ptend_s = io->ptend_s; T.ptend_q = io->ptend_q; T.ptend_u = io->ptend_u; T.ptend_v = io->ptend_v;
        rc = launch_zm_tail_small(g_P, g_estbl, T, st);
        return rc ? fail(rc, "fused tail launch failed") : COLPHYS_OK;
    }
    if ((rc = geopotential())) return rc;

    // ---- zm_conv_evap on state1 (:764-771); evapcdp = its dq/dt (:773) ----------------------------------------
    if ((rc = colphys_zm_conv_evap_dev(ncol, pver, ztodt, &e, stream))) return rc;
    if ((rc = update(t1, tend_s, q1, tend_q, nullptr, nullptr, nullptr, nullptr, 0))) return fail(rc, "physics_update launch failed");
    if ((rc = geopotential())) return rc;

    // ---- momtran on state1's winds (:814-836); u, v have not been touched yet, so state1 u, v = state u, v -------
    {
        ProfScope ps(COLPHYS_K_MOMTRAN, st);
        if ((rc = launch_momtran(g_P, ncol, pver, pcols, ztodt, m, st, io->v))) return fail(rc, "momtran launch failed");
    }
    // ptend_all%u, %v start here (nothing wrote them before); %s accumulates
    if ((rc = update(t1, seten, nullptr, nullptr, io->u, wtend, io->v, wtend + nM, 4 | 8))) return fail(rc, "physics_update launch failed");
    return geopotential();
}

int colphys_zm_conv_tend_dev(int ncol, int pver, int pcols, double ztodt, int dycore_kind, double qmin,
                             const colphys_zm_conv_tend_io *io, void *stream)
{
    return zm_conv_tend_impl(ncol, pver, pcols, ztodt, dycore_kind, qmin, io, stream, &g_ws_arena, &g_tend_arena);
}

int colphys_math_probe_dev(int kind, int64_t n, const double *x, double *y, void *stream)
{
    int rc;
    if ((rc = check_ready())) return rc;
    if (kind < COLPHYS_PROBE_LOG || kind > COLPHYS_PROBE_EXP) return fail(COLPHYS_ERR_ARG, "unknown probe kind");
    if (n < 0 || (n > 0 && (!x || !y))) return fail(COLPHYS_ERR_ARG, "bad probe arguments");
    rc = launch_math_probe(kind, n, x, y, (hipStream_t)stream);
    return rc ? fail(rc, "math probe launch failed") : COLPHYS_OK;
}

// ---- aerosol tables -------------------------------------------------------------------------------------
int colphys_aero_tables_set(const colphys_aero_tables *tb)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int rc;
    if ((rc = check_ready())) return rc;
    if (!tb) return fail(COLPHYS_ERR_ARG, "null tables");
    if (tb->nmodes < 1 || tb->nspec_max < 1 || tb->nspec_max > 8) return fail(COLPHYS_ERR_UNSUPPORTED, "need 1 <= nspec_max <= 8");
    for (void *b : g_aero_bufs) hipFree(b);
    g_aero_bufs.clear();
    auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
        void *d = nullptr;
        if (!src) return -1;
        if (hipMalloc(&d, bytes) != hipSuccess) return -1;
        g_aero_bufs.push_back(d);
        if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return -1;
        *dst = d;
        return 0;
    };
    AeroDev &A = g_A;
    A.nmodes = tb->nmodes; A.nspec_max = tb->nspec_max; A.nswbands = tb->nswbands; A.nlwbands = tb->nlwbands;
    A.idx_sw_diag = tb->idx_sw_diag; A.idx_uv_diag = tb->idx_uv_diag; A.idx_nir_diag = tb->idx_nir_diag;
    A.top_lev = tb->top_lev; A.rhoh2o = tb->rhoh2o; A.rga = tb->rga; A.rair = tb->rair; A.xrmin = tb->xrmin; A.xrmax = tb->xrmax;
    const size_t nm = tb->nmodes, ns = tb->nspec_max, nsw = tb->nswbands, nlw = tb->nlwbands;
    const size_t tblsz = (size_t)COLPHYS_PREFI * COLPHYS_PREFR * COLPHYS_NCOEF;
    int bad = 0;
    bad |= up(tb->nspec, sizeof(int32_t) * nm, (const void **)&A.nspec);
    bad |= up(tb->spectype, sizeof(int32_t) * nm * ns, (const void **)&A.spectype);
    bad |= up(tb->sigmag, 8 * nm, (const void **)&A.sigmag);
    bad |= up(tb->specdens, 8 * nm * ns, (const void **)&A.specdens);
    bad |= up(tb->hygro, 8 * nm * ns, (const void **)&A.hygro);
    bad |= up(tb->specrefindex_sw, 8 * nm * ns * nsw * 2, (const void **)&A.specrefindex_sw);
    bad |= up(tb->specrefindex_lw, 8 * nm * ns * nlw * 2, (const void **)&A.specrefindex_lw);
    bad |= up(tb->refrtabsw, 8 * nm * nsw * COLPHYS_PREFR, (const void **)&A.refrtabsw);
    bad |= up(tb->refitabsw, 8 * nm * nsw * COLPHYS_PREFI, (const void **)&A.refitabsw);
    bad |= up(tb->refrtablw, 8 * nm * nlw * COLPHYS_PREFR, (const void **)&A.refrtablw);
    bad |= up(tb->refitablw, 8 * nm * nlw * COLPHYS_PREFI, (const void **)&A.refitablw);
    bad |= up(tb->extpsw, 8 * nm * nsw * tblsz, (const void **)&A.extpsw);
    bad |= up(tb->abspsw, 8 * nm * nsw * tblsz, (const void **)&A.abspsw);
    bad |= up(tb->asmpsw, 8 * nm * nsw * tblsz, (const void **)&A.asmpsw);
    bad |= up(tb->absplw, 8 * nm * nlw * tblsz, (const void **)&A.absplw);
    bad |= up(tb->crefwsw, 8 * nsw * 2, (const void **)&A.crefwsw);
    bad |= up(tb->crefwlw, 8 * nlw * 2, (const void **)&A.crefwlw);
    if (bad) return fail(COLPHYS_ERR_HIP, "uploading aerosol tables failed (null table or hipMalloc)");
    // packed slices (see AeroDev): table element (m, ib, ii, ir, nc) -> slice[t][(ii*PREFR + ir)*AERO_NCP + nc]
    // interval lookup of one grid (colphys_common.h, AERO_LUTB): false if two grid points would share a bucket
    auto build_lut = [](const double *g, int n, double *gin, unsigned char *cnt, int32_t *base, int32_t *shift) -> bool {
        auto hi32 = [](double x) -> int64_t { int64_t b; std::memcpy(&b, &x, 8); return b >> 32; };
        auto bucket = [&](double x, int64_t bs, int sh) -> int {
            int64_t h = hi32(x);
            if (h < 0) h = 0;
            int64_t b = (h - bs) >> sh;
            return (int)std::min<int64_t>(std::max<int64_t>(b, 0), AERO_LUTB - 1);
        };
        for (int j = 0; j < n; ++j)
            if (!(g[j] == g[j]) || g[j] < 0.0 || (j > 0 && !(g[j] > g[j - 1]))) return false;     // NaN, negative, not ascending
        // the first point may sit arbitrarily far below the rest (an imaginary-index grid starting at 0): it takes bucket 0
        // alone, the second point is put into bucket 1 and the last one must not pass the top bucket
        const int64_t h2 = hi32(g[n > 1 ? 1 : 0]), hn = hi32(g[n - 1]);
        for (int sh = 0; sh <= 30; ++sh) {
            const int64_t bs = h2 - ((int64_t)1 << sh);
            if (bs < 0) break;                                   // the kernels subtract in 32 bits from a non-negative word
            if (((hn - bs) >> sh) > AERO_LUTB - 1) continue;
            bool ok = true;
            for (int j = 1; j < n && ok; ++j) ok = bucket(g[j], bs, sh) != bucket(g[j - 1], bs, sh);
            if (!ok) return false;                               // coarser buckets only merge more
            for (int b = 0; b < AERO_LUTB; ++b) {
                gin[b] = std::numeric_limits<double>::quiet_NaN();      // empty bucket: x >= NaN is false for every x, +inf included
                int below = 0;
                for (int j = 0; j < n; ++j) {
                    const int bj = bucket(g[j], bs, sh);
                    if (bj < b) ++below;
                    if (bj == b) gin[b] = g[j];
                }
                cnt[b] = (unsigned char)below;
            }
            *base = (int32_t)bs; *shift = sh;
            return true;
        }
        return false;
    };
    bool lut_all_ok = true;
    auto pack = [&](int ntab, const double *const *tabs, const double *refr, const double *refi, size_t nbnd,
                    const double **dst) -> int {
        const size_t sl = (size_t)aero_slice_doubles(ntab), corners = (size_t)COLPHYS_PREFI * COLPHYS_PREFR;
        std::vector<double> h(nm * nbnd * sl, 0.0);
        for (size_t u = 0; u < nm * nbnd; ++u) {
            double *o = h.data() + u * sl;
            for (int t = 0; t < ntab; ++t)
                for (size_t c = 0; c < corners; ++c)
                    for (int nc = 0; nc < COLPHYS_NCOEF; ++nc)
                        o[(size_t)t * AERO_TBLP + c * AERO_NCP + nc] = tabs[t][u * tblsz + c * COLPHYS_NCOEF + nc];
            double *g = o + (size_t)ntab * AERO_TBLP;
            for (int j = 0; j < COLPHYS_PREFR; ++j) g[j] = refr[u * COLPHYS_PREFR + j];
            for (int j = 0; j < COLPHYS_PREFI; ++j) g[COLPHYS_PREFR + j] = refi[u * COLPHYS_PREFI + j];
            double *r = g + COLPHYS_PREFR + COLPHYS_PREFI;       // reciprocal widths, table_interp's |dx| > 1e-20 guard
            for (int j = 0; j + 1 < COLPHYS_PREFR; ++j) {
                const double dx = g[j + 1] - g[j];
                r[j] = std::fabs(dx) > 1.e-20 ? 1.0 / dx : 0.0;
            }
            for (int j = 0; j + 1 < COLPHYS_PREFI; ++j) {
                const double dx = g[COLPHYS_PREFR + j + 1] - g[COLPHYS_PREFR + j];
                r[COLPHYS_PREFR + j] = std::fabs(dx) > 1.e-20 ? 1.0 / dx : 0.0;
            }
            double *lut = o + aero_lut_offset(ntab);
            unsigned char *cnt = reinterpret_cast<unsigned char *>(lut + 2 * AERO_LUTB);
            int32_t *par = reinterpret_cast<int32_t *>(lut + 2 * AERO_LUTB + 2 * AERO_LUTB / 8);
            const bool ok = build_lut(g, COLPHYS_PREFR, lut, cnt, par + 0, par + 1) &&
                            build_lut(g + COLPHYS_PREFR, COLPHYS_PREFI, lut + AERO_LUTB, cnt + AERO_LUTB, par + 2, par + 3);
            par[4] = ok ? 1 : 0;
            lut_all_ok = lut_all_ok && ok;
        }
        return up(h.data(), h.size() * sizeof(double), (const void **)dst);
    };
    {
        const double *sw_t[3] = {tb->extpsw, tb->abspsw, tb->asmpsw}, *lw_t[1] = {tb->absplw};
        const char *nolut = getenv("COLPHYS_AERO_NO_LUT");      // tests: force the linear grid search
        const bool want = !(nolut && atoi(nolut));
        lut_all_ok = true;
        bad |= pack(3, sw_t, tb->refrtabsw, tb->refitabsw, nsw, &A.slices_sw);
        A.lut_ok_sw = lut_all_ok && want;
        lut_all_ok = true;
        bad |= pack(1, lw_t, tb->refrtablw, tb->refitablw, nlw, &A.slices_lw);
        A.lut_ok_lw = lut_all_ok && want;
    }
    {
        std::vector<double> e(nm);
        for (size_t m = 0; m < nm; ++m) {
            const double alnsg = std::log(tb->sigmag[m]);
            e[m] = std::exp(2.0 * alnsg * alnsg);
        }
        bad |= up(e.data(), 8 * nm, (const void **)&A.explnsigma);
        std::vector<double> rd(nm * ns);
        for (size_t i = 0; i < nm * ns; ++i) rd[i] = 1.0 / tb->specdens[i];
        bad |= up(rd.data(), 8 * nm * ns, (const void **)&A.rspecdens);
        std::vector<int32_t> lm(nm, 0);
        for (size_t m = 0; m < nm; ++m)
            for (int l = 0; l < tb->nspec[m]; ++l) {
                const int ty = tb->spectype[m * ns + l];
                bool last = ty >= 0 && ty < 6;
                for (int l2 = l + 1; last && l2 < tb->nspec[m]; ++l2) last = tb->spectype[m * ns + l2] != ty;
                if (last) lm[m] |= 1 << l;
            }
        bad |= up(lm.data(), sizeof(int32_t) * nm, (const void **)&A.lastmask);
        A.rxrange = 1.0 / (tb->xrmax - tb->xrmin);
        A.nspec_total = 0;
        for (size_t m = 0; m < nm; ++m) A.nspec_total += tb->nspec[m];
    }
    if (bad) return fail(COLPHYS_ERR_HIP, "uploading packed aerosol table slices failed");
    g_hf_nspec.assign(tb->nspec, tb->nspec + tb->nmodes);
    g_aero_set = true;
    return COLPHYS_OK;
}

static int aero_dev(bool sw, int ncol, int pver, const colphys_modal_aero_io *io, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!g_aero_set) return fail(COLPHYS_ERR_NOTINIT, "colphys_aero_tables_set has not been called");
    if (!io || !io->pdeldry || !io->specmmr || !io->dgnumwet || !io->qaerwat || !io->tauxar || !io->status)
        return fail(COLPHYS_ERR_ARG, "null pointer in colphys_modal_aero_io");
    if (sw && (!io->wa || !io->ga || !io->fa)) return fail(COLPHYS_ERR_ARG, "SW needs wa, ga, fa");
    ProfScope ps(sw ? COLPHYS_K_AERO_SW : COLPHYS_K_AERO_LW, (hipStream_t)stream);
    rc = sw ? launch_modal_aero_sw(g_A, ncol, pver, *io, (hipStream_t)stream)
            : launch_modal_aero_lw(g_A, ncol, pver, *io, (hipStream_t)stream);
    return rc ? fail(rc, "modal_aero launch failed") : COLPHYS_OK;
}
int colphys_modal_aero_sw_dev(int ncol, int pver, const colphys_modal_aero_io *io, void *stream) { return aero_dev(true, ncol, pver, io, stream); }
int colphys_modal_aero_lw_dev(int ncol, int pver, const colphys_modal_aero_io *io, void *stream) { return aero_dev(false, ncol, pver, io, stream); }

int colphys_aero_kernel_select(int sw_kind, int lw_kind)
{
    if (sw_kind < COLPHYS_AERO_KERNEL_AUTO || sw_kind > COLPHYS_AERO_KERNEL_GROUPED || lw_kind < COLPHYS_AERO_KERNEL_AUTO ||
        lw_kind > COLPHYS_AERO_KERNEL_GROUPED)
        return fail(COLPHYS_ERR_ARG, "unknown aerosol-optics kernel kind");
    std::lock_guard<std::mutex> lk(g_mu);
    aero_kernel_select(sw_kind, lw_kind);
    return COLPHYS_OK;
}

int colphys_aero_grid_probe_dev(int sw, int mode, int band, int which, int use_lut, int64_t n, const double *x, int32_t *ix1,
                                int32_t *ix2, double *t, int32_t *lut_available, void *stream)
{
    int rc;
    if ((rc = check_ready())) return rc;
    if (!g_aero_set) return fail(COLPHYS_ERR_NOTINIT, "colphys_aero_tables_set has not been called");
    const int nb = sw ? g_A.nswbands : g_A.nlwbands;
    if (mode < 0 || mode >= g_A.nmodes || band < 0 || band >= nb || which < 0 || which > 1 || n < 0 || (n > 0 && (!x || !ix1 || !ix2 || !t)))
        return fail(COLPHYS_ERR_ARG, "bad grid probe arguments");
    if (lut_available) *lut_available = sw ? g_A.lut_ok_sw : g_A.lut_ok_lw;
    rc = launch_aero_grid_probe(g_A, sw, mode, band, which, use_lut, n, x, ix1, ix2, t, (hipStream_t)stream);
    return rc ? fail(rc, "grid probe launch failed") : COLPHYS_OK;
}

int colphys_modal_aero_sw_diag_dev(int ncol, int pver, const colphys_modal_aero_io *io, double *lev, double *col, void *stream)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_dims(ncol, pver))) return rc;
    if (!g_aero_set) return fail(COLPHYS_ERR_NOTINIT, "colphys_aero_tables_set has not been called");
    if (!io || !io->pdeldry || !io->pmid || !io->t || !io->specmmr || !io->dgnumwet || !io->qaerwat || !io->troplevchem ||
        !io->tauxar || !lev || !col)
        return fail(COLPHYS_ERR_ARG, "null pointer in colphys_modal_aero_sw_diag_dev (pmid, t, troplevchem and the SW tauxar are required)");
    const int nsw = g_A.nswbands;
    if (g_A.idx_sw_diag < 1 || g_A.idx_sw_diag > nsw || g_A.idx_uv_diag < 1 || g_A.idx_uv_diag > nsw || g_A.idx_nir_diag < 1 ||
        g_A.idx_nir_diag > nsw)
        return fail(COLPHYS_ERR_ARG, "idx_sw_diag / idx_uv_diag / idx_nir_diag of the aerosol tables must be bands 1..nswbands");
    ProfScope ps(COLPHYS_K_AERO_SW_DIAG, (hipStream_t)stream);
    rc = launch_modal_aero_sw_diag(g_A, ncol, pver, *io, lev, col, (hipStream_t)stream);
    return rc ? fail(rc, "modal_aero_sw_diag launch failed (pver above 256 is not supported)") : COLPHYS_OK;
}

// =======================================================================================================
// host-layout (chunk) entry points: Fortran (pcols, nlev, nchunks) <-> device [col][nlev]
//   device column index = chunk*pcols + i; padding columns i >= ncol(chunk) are filled with a copy of the
//   chunk's first column (CAM leaves +inf there) and their results are discarded.
// =======================================================================================================
}  // extern "C"

namespace {

// ---- host worker pool for the layout transposes of the chunk entry points -------------------------------------
// (pcols, nlev, nchunks) <-> [col][nlev] is a strided copy of every field of a rank (~0.3 GB per zm_convr call at ne30):
// one core moves it at ~10 GB/s, less than PCIe, so the chunk range is split over a few persistent threads.
class HostPool {
public:
    static HostPool &get() { static HostPool p; return p; }
    // f(lo, hi) over [0, n) in contiguous pieces; runs inline when the range is small
    void run(int n, const std::function<void(int, int)> &f)
    {
        const int T = (int)workers_.size() + 1;
        if (T == 1 || n < 64 * T) { f(0, n); return; }
        const int per = (n + T - 1) / T;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f; n_ = n; per_ = per; pending_ = T - 1; ++gen_;
        }
        cv_.notify_all();
        f(0, per < n ? per : n);                       // the caller takes the first piece
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
private:
    HostPool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int T = hw ? (int)hw : 1;
        if (const char *e = getenv("COLPHYS_HOST_THREADS")) T = atoi(e);
        T = T < 1 ? 1 : (T > 8 ? 8 : T);               // memory-bound: a few threads saturate the copy
        for (int i = 1; i < T; ++i) workers_.emplace_back([this, i] { loop(i); });
    }
    void loop(int id)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int, int)> *f;
            int lo, hi;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                f = fn_; lo = id * per_; hi = lo + per_ < n_ ? lo + per_ : n_;
            }
            if (f && lo < hi) (*f)(lo, hi);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *fn_ = nullptr;
    int n_ = 0, per_ = 0, pending_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

struct Stager {
    int nchunks, pcols;
    const int32_t *ncol;
    int ncd;                       // device columns = nchunks*pcols
    Carver hc, dc;
    hipStream_t st = nullptr;
    hipEvent_t ev_pending = nullptr;
    bool ok = true;                            // false once any copy / event call of this staging pass failed
    void chk(hipError_t e) { if (e != hipSuccess) ok = false; }
    std::function<void()> scatter_pending;     // scatter of the previous download, run while the next one is in flight
    // Few columns (the per-chunk legacy call: one chunk of pcols columns): the ~40 per-field copies are round trips that
    // dominate the call, so the inputs -- packed back to back in the pinned arena at the SAME offsets as their device
    // buffers -- go up in ONE copy (flush_in) and the outputs come down in ONE copy of the device range out<>() carved,
    // scattered from there.  Many columns: per-field copies, each scatter overlapping the next download, as before.
    bool batch;
    size_t in_lo = 0, in_hi = 0, out_lo = 0, out_hi = 0;
    char *h_out = nullptr;                     // pinned image of the device range [out_lo, out_hi)
    Stager(int nch, int pc, const int32_t *nc, void *hbase, void *dbase)
        : nchunks(nch), pcols(pc), ncol(nc), ncd(nch * pc), hc(hbase), dc(dbase), batch(nch * pc <= 512)
    {
        if (const char *e = getenv("COLPHYS_STAGE_BATCH")) batch = atoi(e) != 0;      // tests: either path at any size
    }
    // uploads the inputs packed so far (batch mode); call before the kernels are enqueued
    void flush_in()
    {
        if (batch && in_hi > in_lo) chk(hipMemcpyAsync(dc.base + in_lo, hc.base + in_lo, in_hi - in_lo, hipMemcpyHostToDevice, st));
        in_lo = in_hi = 0;
    }
    // an input packed at h (pinned) for device buffer d: copy now, or remember the range when host and device offsets agree
    void upload(void *d, const void *h, size_t bytes)
    {
        const size_t ho = (size_t)((const char *)h - hc.base), dof = (size_t)((char *)d - dc.base);
        if (batch && ho == dof && (in_hi == in_lo || ho >= in_hi)) {
            if (in_hi == in_lo) in_lo = ho;
            in_hi = ho + bytes;                // the gap an alignment left between two fields is copied along (pinned, harmless)
            return;
        }
        flush_in();
        chk(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
    }
    ~Stager()
    {
        drain();
        if (ev_pending) hipEventDestroy(ev_pending);
    }
    void drain()
    {
        if (scatter_pending) {
            chk(hipEventSynchronize(ev_pending));
            if (ok) scatter_pending();
            scatter_pending = nullptr;
        }
    }

    // pack a (pcols, nlev, nchunks) Fortran field and start its upload; returns the device pointer
    double *in(const double *src, int nlev) { return in_multi(src, nlev, 1); }
    // nf fields, source field-major (pcols, nlev, nchunks, nf) -> device [nf][ncd][nlev], contiguous
    double *in_multi(const double *src, int nlev, int nf)
    {
        const size_t n = (size_t)ncd * nlev;
        double *h = hc.take<double>(n * nf), *d = dc.take<double>(n * nf);
        HostPool::get().run(nchunks, [&](int c0, int c1) {
            for (int f = 0; f < nf; ++f)
                for (int c = c0; c < c1; ++c) {
                    const double *s = src + (size_t)f * n + (size_t)c * pcols * nlev;
                    for (int i = 0; i < pcols; ++i) {
                        const int isrc = (i < ncol[c]) ? i : 0;
                        double *o = h + (size_t)f * n + ((size_t)c * pcols + i) * nlev;
                        for (int k = 0; k < nlev; ++k) o[k] = s[(size_t)k * pcols + isrc];
                    }
                }
        });
        upload(d, h, n * nf * sizeof(double));
        return d;
    }
    int32_t *in_i(const int32_t *src, int nlev)
    {
        const size_t n = (size_t)ncd * nlev;
        int32_t *h = hc.take<int32_t>(n), *d = dc.take<int32_t>(n);
        HostPool::get().run(nchunks, [&](int c0, int c1) {
            for (int c = c0; c < c1; ++c)
                for (int i = 0; i < pcols; ++i)
                    for (int k = 0; k < nlev; ++k) h[((size_t)c * pcols + i) * nlev + k] = src[(size_t)c * pcols * nlev + (size_t)k * pcols + i];
        });
        upload(d, h, n * sizeof(int32_t));
        return d;
    }
    template <class T> T *out(int nlev)
    {
        flush_in();
        T *d = dc.take<T>((size_t)ncd * nlev);
        const size_t o = (size_t)((char *)d - dc.base);
        if (out_hi == out_lo) out_lo = o;
        out_hi = o + (size_t)ncd * nlev * sizeof(T);
        return d;
    }

    // download a device [col][nlev] field and scatter the first `valid(c)` columns of each chunk; the scatter of one
    // field runs while the download of the next is in flight (the last one is drained by the destructor)
    template <class T> void back(T *dst, const T *dev, int nlev, bool gathered, const int32_t *lengath)
    {
        const size_t n = (size_t)ncd * nlev;
        const size_t dof = (size_t)((const char *)dev - dc.base);
        if (batch && out_hi > out_lo && dof >= out_lo && dof + n * sizeof(T) <= out_hi) {
            if (!h_out) {                              // first output asked for: the whole output range in one synchronous copy
                h_out = hc.take<char>(out_hi - out_lo);
                chk(hipMemcpy(h_out, dc.base + out_lo, out_hi - out_lo, hipMemcpyDeviceToHost));
            }
            if (!ok) return;
            const T *h = (const T *)(h_out + (dof - out_lo));
            for (int c = 0; c < nchunks; ++c) {
                const int nv = gathered ? lengath[c] : ncol[c];
                T *o = dst + (size_t)c * pcols * nlev;
                for (int i = 0; i < nv; ++i) {
                    const T *sp = h + ((size_t)c * pcols + i) * nlev;
                    for (int k = 0; k < nlev; ++k) o[(size_t)k * pcols + i] = sp[k];
                }
            }
            return;
        }
        T *h = hc.take<T>(n);
        chk(hipMemcpyAsync(h, dev, n * sizeof(T), hipMemcpyDeviceToHost, st));
        drain();                                       // previous field: wait for ITS copy, scatter it
        if (!ev_pending) chk(hipEventCreateWithFlags(&ev_pending, hipEventDisableTiming));
        chk(hipEventRecord(ev_pending, st));
        Stager *self = this;
        scatter_pending = [self, dst, h, nlev, gathered, lengath]() {
            HostPool::get().run(self->nchunks, [&](int c0, int c1) {
                for (int c = c0; c < c1; ++c) {
                    const int nv = gathered ? lengath[c] : self->ncol[c];
                    T *o = dst + (size_t)c * self->pcols * nlev;
                    for (int i = 0; i < nv; ++i) {
                        const T *s = h + ((size_t)c * self->pcols + i) * nlev;
                        for (int k = 0; k < nlev; ++k) o[(size_t)k * self->pcols + i] = s[k];
                    }
                }
            });
        };
    }
};

int stage_reserve(size_t host_bytes, size_t dev_bytes)
{
    if (g_pin_arena.reserve(host_bytes) || g_dev_arena.reserve(dev_bytes)) return fail(COLPHYS_ERR_HIP, "staging allocation failed");
    return COLPHYS_OK;
}
int check_chunks(int nchunks, int pcols, const int32_t *ncol)
{
    if (nchunks < 1 || pcols < 1 || !ncol) return fail(COLPHYS_ERR_ARG, "bad chunk description");
    for (int c = 0; c < nchunks; ++c)
        if (ncol[c] < 1 || ncol[c] > pcols) return fail(COLPHYS_ERR_ARG, "ncol(chunk) must be in 1..pcols");
    return COLPHYS_OK;
}

}  // namespace

extern "C" {

int colphys_geopotential_t_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, int dycore_kind, int nq,
                                  int nspecies, const double *piln, const double *pmln, const double *pint,
                                  const double *pmid, const double *pdel, const double *rpdel, const double *t,
                                  const double *q, const double *rair, double gravit, const double *zvir, double *zi,
                                  double *zm)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols, pverp = pver + 1;
    const size_t bytes = 8 * ncd * (4 * pverp + (8 + (size_t)nq) * pver) + 64 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    const bool fv = dycore_kind == COLPHYS_DYCORE_FV;
    double *d_piln = fv ? s.in(piln, pverp) : nullptr, *d_pint = fv ? s.in(pint, pverp) : nullptr;
    double *d_rpdel = fv ? s.in(rpdel, pver) : nullptr;
    double *d_pmid = fv ? nullptr : s.in(pmid, pver), *d_pdel = fv ? nullptr : s.in(pdel, pver);
    double *d_t = s.in(t, pver), *d_rair = s.in(rair, pver), *d_zvir = s.in(zvir, pver);
    // q: (pcols,pver,nq) per chunk in CAM; here the shim passes species-major (pcols,pver,nchunks,nq)
    double *d_q = s.in_multi(q, pver, nq);
    double *d_zi = s.out<double>(pverp), *d_zm = s.out<double>(pver);
    s.flush_in();
    rc = colphys_geopotential_t_dev((int)ncd, pver, dycore_kind, nq, nspecies, d_piln, nullptr, d_pint, d_pmid, d_pdel,
                                    d_rpdel, d_t, d_q, d_rair, gravit, d_zvir, d_zi, d_zm, nullptr);
    if (rc) return rc;
    s.back(zi, d_zi, pverp, false, nullptr);
    s.back(zm, d_zm, pver, false, nullptr);
    (void)pmln;
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

int colphys_geopotential_dse_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, int dycore_kind, int nq, int nspecies,
                                    const double *piln, const double *pmln, const double *pint, const double *pmid,
                                    const double *pdel, const double *rpdel, const double *dse, const double *q, const double *phis,
                                    const double *rair, double gravit, const double *cpair, const double *zvir, double *t, double *zi,
                                    double *zm)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols, pverp = pver + 1;
    const size_t bytes = 8 * ncd * (4 * pverp + (10 + (size_t)nq) * pver + 2) + 64 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    const bool fv = dycore_kind == COLPHYS_DYCORE_FV;
    double *d_piln = fv ? s.in(piln, pverp) : nullptr, *d_pint = fv ? s.in(pint, pverp) : nullptr;
    double *d_rpdel = fv ? s.in(rpdel, pver) : nullptr;
    double *d_pmid = fv ? nullptr : s.in(pmid, pver), *d_pdel = fv ? nullptr : s.in(pdel, pver);
    double *d_dse = s.in(dse, pver), *d_rair = s.in(rair, pver), *d_cpair = s.in(cpair, pver), *d_zvir = s.in(zvir, pver);
    double *d_phis = s.in(phis, 1);
    double *d_q = s.in_multi(q, pver, nq);
    double *d_t = s.out<double>(pver), *d_zi = s.out<double>(pverp), *d_zm = s.out<double>(pver);
    s.flush_in();
    rc = colphys_geopotential_dse_dev((int)ncd, pver, dycore_kind, nq, nspecies, d_piln, nullptr, d_pint, d_pmid, d_pdel, d_rpdel,
                                      d_dse, d_q, d_phis, d_rair, gravit, d_cpair, d_zvir, d_t, d_zi, d_zm, nullptr);
    if (rc) return rc;
    s.back(t, d_t, pver, false, nullptr);
    s.back(zi, d_zi, pverp, false, nullptr);
    s.back(zm, d_zm, pver, false, nullptr);
    (void)pmln;
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

int colphys_zm_convr_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double delt, const colphys_zm_convr_io *io)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols, pverp = pver + 1;
    const size_t bytes = 8 * ncd * (4 * pverp + 22 * (size_t)pver + 16) + 4 * ncd * 8 + 128 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    colphys_zm_convr_io d = *io;
    d.t = s.in(io->t, pver); d.qh = s.in(io->qh, pver); d.pap = s.in(io->pap, pver); d.dpp = s.in(io->dpp, pver);
    d.zm = s.in(io->zm, pver); d.paph = s.in(io->paph, pverp); d.zi = s.in(io->zi, pverp);
    d.geos = s.in(io->geos, 1); d.pblh = s.in(io->pblh, 1); d.tpert = s.in(io->tpert, 1); d.landfrac = s.in(io->landfrac, 1);
    double **mid_d[] = {&d.qtnd, &d.heat, &d.cme, &d.zdu, &d.rprd, &d.dlf, &d.ql, &d.dif, &d.dnlf, &d.dnif, &d.eurt,
                        &d.mu, &d.md, &d.du, &d.eu, &d.ed, &d.dp};
    for (auto pp : mid_d) *pp = s.out<double>(pver);
    d.mcon = s.out<double>(pverp); d.pflx = s.out<double>(pverp);
    double **col_d[] = {&d.prec, &d.jctop, &d.jcbot, &d.cape, &d.rliq, &d.rice, &d.dsubcld};
    for (auto pp : col_d) *pp = s.out<double>(1);
    d.jt = s.out<int32_t>(1); d.maxg = s.out<int32_t>(1); d.ideep = s.out<int32_t>(1); d.status = s.out<int32_t>(1);
    d.lengath = s.dc.take<int32_t>(nchunks);
    s.flush_in();
    rc = colphys_zm_convr_dev((int)ncd, pver, pcols, delt, &d, nullptr);
    if (rc) return rc;
    // padding columns must not appear in the gathered set: they are copies of column 1 of their chunk, so
    // they trigger exactly when it does; drop them on the host from ideep (ascending order is preserved)
    std::vector<int32_t> h_len(nchunks), h_ideep(ncd), h_jt(ncd), h_maxg(ncd);
    std::vector<double> h_dsub(ncd);
    HIPCHK(hipMemcpy(h_len.data(), d.lengath, 4 * (size_t)nchunks, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h_ideep.data(), d.ideep, 4 * ncd, hipMemcpyDeviceToHost));
    for (int c = 0; c < nchunks; ++c) {
        int n = 0;
        for (int j = 0; j < h_len[c]; ++j)
            if (h_ideep[(size_t)c * pcols + j] <= ncol[c]) ++n;     // padding has i > ncol, and sits at the tail
        io->lengath[c] = n;
    }
    const double *mid_s[] = {d.qtnd, d.heat, d.cme, d.zdu, d.rprd, d.dlf, d.ql, d.dif, d.dnlf, d.dnif, d.eurt};
    double *mid_h[] = {io->qtnd, io->heat, io->cme, io->zdu, io->rprd, io->dlf, io->ql, io->dif, io->dnlf, io->dnif, io->eurt};
    for (int f = 0; f < 11; ++f) s.back(mid_h[f], mid_s[f], pver, false, nullptr);
    s.back(io->mcon, d.mcon, pverp, false, nullptr);
    s.back(io->pflx, d.pflx, pverp, false, nullptr);
    const double *col_s[] = {d.prec, d.jctop, d.jcbot, d.cape, d.rliq, d.rice};
    double *col_h[] = {io->prec, io->jctop, io->jcbot, io->cape, io->rliq, io->rice};
    for (int f = 0; f < 6; ++f) s.back(col_h[f], col_s[f], 1, false, nullptr);
    s.back(io->status, (const int32_t *)d.status, 1, false, nullptr);
    const double *g_s[] = {d.mu, d.md, d.du, d.eu, d.ed, d.dp};
    double *g_h[] = {io->mu, io->md, io->du, io->eu, io->ed, io->dp};
    for (int f = 0; f < 6; ++f) s.back(g_h[f], g_s[f], pver, true, io->lengath);
    s.back(io->dsubcld, (const double *)d.dsubcld, 1, true, io->lengath);
    s.back(io->jt, (const int32_t *)d.jt, 1, true, io->lengath);
    s.back(io->maxg, (const int32_t *)d.maxg, 1, true, io->lengath);
    s.back(io->ideep, (const int32_t *)d.ideep, 1, true, io->lengath);
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

int colphys_zm_conv_evap_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double deltat,
                                const colphys_zm_conv_evap_io *io)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols, pverp = pver + 1;
    const size_t bytes = 8 * ncd * (4 * pverp + 14 * (size_t)pver + 8) + 64 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    colphys_zm_conv_evap_io d = *io;
    d.t = s.in(io->t, pver); d.pmid = s.in(io->pmid, pver); d.pdel = s.in(io->pdel, pver); d.q = s.in(io->q, pver);
    d.prdprec = s.in(io->prdprec, pver); d.cldfrc = s.in(io->cldfrc, pver); d.landfrac = s.in(io->landfrac, 1);
    d.prec = s.in(io->prec, 1);
    double **mid_d[] = {&d.tend_s, &d.tend_s_snwprd, &d.tend_s_snwevmlt, &d.tend_q, &d.ntprprd, &d.ntsnprd};
    for (auto pp : mid_d) *pp = s.out<double>(pver);
    d.flxprec = s.out<double>(pverp); d.flxsnow = s.out<double>(pverp); d.snow = s.out<double>(1);
    s.flush_in();
    rc = colphys_zm_conv_evap_dev((int)ncd, pver, deltat, &d, nullptr);
    if (rc) return rc;
    const double *ms[] = {d.tend_s, d.tend_s_snwprd, d.tend_s_snwevmlt, d.tend_q, d.ntprprd, d.ntsnprd};
    double *mh[] = {io->tend_s, io->tend_s_snwprd, io->tend_s_snwevmlt, io->tend_q, io->ntprprd, io->ntsnprd};
    for (int f = 0; f < 6; ++f) s.back(mh[f], ms[f], pver, false, nullptr);
    s.back(io->flxprec, (const double *)d.flxprec, pverp, false, nullptr);
    s.back(io->flxsnow, (const double *)d.flxsnow, pverp, false, nullptr);
    s.back(io->prec, (const double *)d.prec, 1, false, nullptr);
    s.back(io->snow, (const double *)d.snow, 1, false, nullptr);
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

int colphys_momtran_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, double dt, const colphys_momtran_io *io)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols;
    const size_t bytes = 8 * ncd * (20 * (size_t)pver + 8) + 4 * ncd * 4 + 64 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    colphys_momtran_io d = *io;
    // winds: CAM (pcols,pver,2) per chunk; the shim passes component-major (pcols,pver,nchunks,2)
    d.winds = s.in_multi(io->winds, pver, 2);
    d.mu = s.in(io->mu, pver); d.md = s.in(io->md, pver); d.du = s.in(io->du, pver); d.eu = s.in(io->eu, pver);
    d.ed = s.in(io->ed, pver); d.dp = s.in(io->dp, pver); d.dsubcld = s.in(io->dsubcld, 1);
    d.jt = s.in_i(io->jt, 1); d.mx = s.in_i(io->mx, 1); d.ideep = s.in_i(io->ideep, 1);
    int32_t *dl = s.dc.take<int32_t>(nchunks);
    HIPCHK(hipMemcpy(dl, io->lengath, 4 * (size_t)nchunks, hipMemcpyHostToDevice));
    d.lengath = dl;
    double *o3[5];
    for (auto &p : o3) p = s.dc.take<double>(2 * ncd * pver);
    d.dqdt = o3[0]; d.pguall = o3[1]; d.pgdall = o3[2]; d.icwu = o3[3]; d.icwd = o3[4];
    d.seten = s.out<double>(pver);
    s.flush_in();
    rc = colphys_momtran_dev((int)ncd, pver, pcols, dt, &d, nullptr);
    if (rc) return rc;
    double *h3[] = {io->dqdt, io->pguall, io->pgdall, io->icwu, io->icwd};
    for (int f = 0; f < 5; ++f)
        for (int m = 0; m < 2; ++m) s.back(h3[f] + (size_t)m * ncd * pver, (const double *)(o3[f] + (size_t)m * ncd * pver), pver, false, nullptr);
    s.back(io->seten, (const double *)d.seten, pver, false, nullptr);
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

int colphys_convtran_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, int ncnst, double dt,
                            const colphys_convtran_io *io)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    if (!io || ncnst < 1) return fail(COLPHYS_ERR_ARG, "null io / bad ncnst");
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols;
    const size_t bytes = 8 * ncd * ((3 * (size_t)ncnst + 8) * pver + 8) + 4 * ncd * 4 + 64 * 256;
    if ((rc = stage_reserve(2 * bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    colphys_convtran_io d = *io;
    // q / fracis / dqdt: CAM (pcols,pver,ncnst) per chunk; the shim passes tracer-major (pcols,pver,nchunks,ncnst)
    d.q = s.in_multi(io->q, pver, ncnst);
    d.fracis = s.in_multi(io->fracis, pver, ncnst);
    d.mu = s.in(io->mu, pver); d.md = s.in(io->md, pver); d.du = s.in(io->du, pver); d.eu = s.in(io->eu, pver);
    d.ed = s.in(io->ed, pver); d.dp = s.in(io->dp, pver); d.dpdry = s.in(io->dpdry, pver);
    d.jt = s.in_i(io->jt, 1); d.mx = s.in_i(io->mx, 1); d.ideep = s.in_i(io->ideep, 1);
    int32_t *dl = s.dc.take<int32_t>(nchunks);
    HIPCHK(hipMemcpy(dl, io->lengath, 4 * (size_t)nchunks, hipMemcpyHostToDevice));
    d.lengath = dl;
    double *dq = s.dc.take<double>((size_t)ncnst * ncd * pver);
    d.dqdt = dq;
    s.flush_in();
    rc = colphys_convtran_dev((int)ncd, pver, pcols, ncnst, dt, &d, nullptr);
    if (rc) return rc;
    for (int m = 1; m < ncnst; ++m)
        if (io->doconvtran[m])
            s.back(io->dqdt + (size_t)m * ncd * pver, (const double *)(dq + (size_t)m * ncd * pver), pver, false, nullptr);
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}

static int aero_chunks(bool sw, int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io,
                       double *diag_lev = nullptr, double *diag_col = nullptr)
{
    int rc;
    if ((rc = check_ready()) || (rc = check_chunks(nchunks, pcols, ncol))) return rc;
    if (!g_aero_set) return fail(COLPHYS_ERR_NOTINIT, "colphys_aero_tables_set has not been called");
    if (!io) return fail(COLPHYS_ERR_ARG, "null io");
    std::lock_guard<std::mutex> lk(g_stage_mu);
    const size_t ncd = (size_t)nchunks * pcols, pverp = pver + 1;
    const int nm = g_A.nmodes, ns = g_A.nspec_max, nb = sw ? g_A.nswbands : g_A.nlwbands;
    const size_t nout = sw ? 4 * (size_t)nb * pverp : (size_t)nb * pver;
    const bool diag = diag_lev || diag_col;
    if (diag && (!sw || !diag_lev || !diag_col || !io->pmid || !io->t || !io->troplevchem))
        return fail(COLPHYS_ERR_ARG, "SW diagnostics need lev, col, pmid, t and troplevchem");
    const int ncolf = COLPHYS_SWD_NCOL + COLPHYS_SWD_NMODEF * nm;
    const size_t bytes = 8 * ncd * ((3 + (size_t)nm * (ns + 2)) * pver + 2 * nout + 8 + (diag ? 2 * ((size_t)COLPHYS_SWD_NLEV * pver + ncolf) + 3 * (size_t)nm * pver : 0)) + 256 * 256;
    if ((rc = stage_reserve(bytes, bytes))) return rc;
    Stager s(nchunks, pcols, ncol, g_pin_arena.p, g_dev_arena.p);
    colphys_modal_aero_io d = *io;
    d.pdeldry = s.in(io->pdeldry, pver);
    d.pmid = io->pmid ? s.in(io->pmid, pver) : nullptr;
    d.t = io->t ? s.in(io->t, pver) : nullptr;
    // specmmr (pcols,pver,nchunks,nspec_max,nmodes); dgnumwet/qaerwat (pcols,pver,nchunks,nmodes): field-major
    d.specmmr = s.in_multi(io->specmmr, pver, nm * ns);
    d.dgnumwet = s.in_multi(io->dgnumwet, pver, nm);
    d.qaerwat = s.in_multi(io->qaerwat, pver, nm);
    d.troplevchem = diag ? s.in_i(io->troplevchem, 1) : nullptr;
    d.status = s.out<int32_t>(1);
    const int nlevout = sw ? (int)pverp : pver;
    double *o[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int f = 0; f < (sw ? 4 : 1); ++f) o[f] = s.dc.take<double>(ncd * nb * nlevout);
    d.tauxar = o[0]; d.wa = o[1]; d.ga = o[2]; d.fa = o[3];
    d.vis_optics = diag ? s.dc.take<double>(3 * (size_t)nm * ncd * pver) : nullptr;      // visible-band optics handed from SW to its diagnostics
    s.flush_in();
    rc = sw ? colphys_modal_aero_sw_dev((int)ncd, pver, &d, nullptr) : colphys_modal_aero_lw_dev((int)ncd, pver, &d, nullptr);
    if (rc) return rc;
    double *dlev = nullptr, *dcol = nullptr;
    if (diag) {
        dlev = s.dc.take<double>((size_t)COLPHYS_SWD_NLEV * ncd * pver);
        dcol = s.dc.take<double>((size_t)ncolf * ncd);
        if ((rc = colphys_modal_aero_sw_diag_dev((int)ncd, pver, &d, dlev, dcol, nullptr))) return rc;
    }
    // device [col][band][lev] -> Fortran (pcols, lev, nbands) per chunk
    double *hdst[4] = {io->tauxar, io->wa, io->ga, io->fa};
    std::vector<double> h(ncd * nb * nlevout);
    for (int f = 0; f < (sw ? 4 : 1); ++f) {
        HIPCHK(hipMemcpy(h.data(), o[f], h.size() * 8, hipMemcpyDeviceToHost));
        for (int c = 0; c < nchunks; ++c)
            for (int i = 0; i < ncol[c]; ++i)
                for (int b = 0; b < nb; ++b)
                    for (int k = 0; k < nlevout; ++k)
                        hdst[f][(size_t)c * pcols * nlevout * nb + ((size_t)b * nlevout + k) * pcols + i] =
                            h[(((size_t)c * pcols + i) * nb + b) * nlevout + k];
    }
    if (diag) {     // device [field][col][lev] -> (pcols, lev, nchunks, field); [field][col] -> (pcols, nchunks, field)
        for (int f = 0; f < COLPHYS_SWD_NLEV; ++f)
            s.back(diag_lev + (size_t)f * ncd * pver, (const double *)(dlev + (size_t)f * ncd * pver), pver, false, nullptr);
        for (int f = 0; f < ncolf; ++f) s.back(diag_col + (size_t)f * ncd, (const double *)(dcol + (size_t)f * ncd), 1, false, nullptr);
    }
    s.back(io->status, (const int32_t *)d.status, 1, false, nullptr);
    s.drain();
    return s.ok ? COLPHYS_OK : fail(COLPHYS_ERR_HIP, "a staging copy failed");
}
int colphys_modal_aero_sw_diag_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io,
                                      double *lev, double *col)
{
    if (!lev || !col) return fail(COLPHYS_ERR_ARG, "null diagnostics output");
    return aero_chunks(true, nchunks, pcols, pver, ncol, io, lev, col);
}
int colphys_modal_aero_sw_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io) { return aero_chunks(true, nchunks, pcols, pver, ncol, io); }
int colphys_modal_aero_lw_chunks(int nchunks, int pcols, int pver, const int32_t *ncol, const colphys_modal_aero_io *io) { return aero_chunks(false, nchunks, pcols, pver, ncol, io); }

}  // extern "C"

#include "colphys_hostface.inc"
