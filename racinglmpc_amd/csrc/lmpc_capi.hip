// racinglmpc_amd/csrc/lmpc_capi.hip -- host side of liblmpc_hip.so: the C ABI declared in include/lmpc_hip.h.
// Owns the device lap stores, work buffers, stream and HIP-event timers; launches the kernels of
// lmpc_kernels.hip.h.  No CPU compute path exists here: if HIP fails, the call fails.
#include "lmpc_variant.hip.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int set_err(int code, const char *what, const char *detail) { g_err = std::string(what) + ": " + (detail ? detail : ""); return code; }
// Developer knobs read from the environment (round 6, ADVICE r5: a stray variable must not change behaviour silently).  Every knob that is SET is recorded
// here, announced once on stderr and listed by lmpc_active_knobs(); knobs that change RESULTS (LMPC_NO_RETRY) exist in developer flavours only.
static std::mutex g_knob_mu;
static std::string g_knobs;
static const char *dev_knob(const char *name) {
    const char *e = getenv(name);
    if (e) {
        std::lock_guard<std::mutex> lk(g_knob_mu);
        const std::string item = std::string(name) + "=" + e;
        if ((";" + g_knobs + ";").find(";" + item + ";") == std::string::npos) {
            g_knobs += (g_knobs.empty() ? "" : ";") + item;
            fprintf(stderr, "liblmpc_hip: developer knob %s is active\n", item.c_str());
        }
    }
    return e;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(LMPC_E_HIP, #call, hipGetErrorString(e_)); } while (0)
#define ARGCHK(cond) do { if (!(cond)) return set_err(LMPC_E_ARG, "argument check failed", #cond); } while (0)

// ---- device allocations of a context.  Builds with -DLMPC_GUARD (the "asan" developer flavour, racinglmpc_amd.build: host code under AddressSanitizer) put a
// 256-byte guard zone filled with 0xA5 behind every allocation -- and between the work-buffer ranges of the two slabs -- and check it when the buffer is
// freed (lmpc_destroy, store growth, end of a rollout session): a kernel that wrote past a buffer's end is reported on stderr and counted
// (lmpc_debug_guard_failures).  Ordinary builds: plain hipMalloc / hipFree.
#define LMPC_GUARD_BYTES 256
static int g_guard_failures = 0;
#ifdef LMPC_GUARD
#include <map>
#include <mutex>
static std::map<void *, size_t> g_guarded; static std::mutex g_guard_mu;
static int check_guard_zone(const void *zone, const char *what) {
    unsigned char h[LMPC_GUARD_BYTES];
    if (hipMemcpy(h, zone, LMPC_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    for (int i = 0; i < LMPC_GUARD_BYTES; i++) if (h[i] != 0xA5) {
        fprintf(stderr, "liblmpc_hip GUARD: %s: byte %d behind the buffer was overwritten (0x%02x)\n", what, i, h[i]); g_guard_failures++; return 1; }
    return 0;
}
static hipError_t g_malloc_raw(void **p, size_t bytes) {
    const size_t padded = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(p, padded + LMPC_GUARD_BYTES);
    if (e != hipSuccess) return e;
    e = hipMemset((char *)*p + padded, 0xA5, LMPC_GUARD_BYTES);
    std::lock_guard<std::mutex> lk(g_guard_mu); g_guarded[*p] = padded;
    return e;
}
static hipError_t g_free(void *p) {
    if (!p) return hipSuccess;
    size_t padded = 0; bool known = false;
    { std::lock_guard<std::mutex> lk(g_guard_mu); auto it = g_guarded.find(p); if (it != g_guarded.end()) { padded = it->second; known = true; g_guarded.erase(it); } }
    if (known) { (void)hipDeviceSynchronize(); check_guard_zone((char *)p + padded, "device buffer"); }
    return hipFree(p);
}
#else
static hipError_t g_malloc_raw(void **p, size_t bytes) { return hipMalloc(p, bytes); }
static hipError_t g_free(void *p) { return p ? hipFree(p) : hipSuccess; }
#endif
template <class T> static hipError_t g_malloc(T **p, size_t bytes) { return g_malloc_raw((void **)p, bytes); }

struct evpair { hipEvent_t a, b; int kind; };

struct lmpc_ctx {
    lmpc_config cfg;
    lmpc_dev_params dp;
    hipStream_t stream;
    double *mstore, *sstore;                 // device lap stores [lap][col][row]
    unsigned *mquant; double *mqpar; int mq_chunks;   // K1 prefilter image of the model store, see quantise_lap
    std::vector<int> m_order, m_len;         // model: sorted position -> slot ; length per slot
    std::vector<int> s_len;                  // safe set: rows per lap (incl. addPoint extensions)
    std::vector<int> s_laptime;              // LMPC.LapTime (rows at addTrajectory time)
    std::vector<double> s_qlast, s_q0;       // last / first Qfun value per lap
    std::vector<int> s_override;             // explicit selection (lmpc_ss_set_selected)
    // device work buffers (host-pointer entry points), sized for max_batch
    double *w_x0, *w_xLin, *w_uLin, *w_uOld, *w_zt, *w_xPP, *w_A, *w_B, *w_C, *w_ssSel, *w_qSel, *w_succ, *w_succU, *w_ztUsed;
    double *w_xPred, *w_uPred, *w_slack, *w_lam, *w_sT, *w_mu, *w_ztN, *w_ztuN, *w_resid;
    int *w_hasPred, *w_tstep, *w_status, *w_iters, *w_rstatus;
    // the work buffers are ranges of two device slabs (inputs | outputs); small batches move each slab with ONE copy through pinned host
    // mirrors (h_in / h_out) instead of one pageable copy per array -- 24 copies of ~10 us each were 60 % of a batch-1 lmpc_step_batch call
    char *slab_in, *slab_out, *h_in, *h_out; size_t slab_in_bytes, slab_out_bytes;
    char *dm_in, *dm_out;                    // device addresses of the host-mapped mirrors h_in / h_out (small contexts: the step kernels read / write them directly)
    lmpc_variant_api var;                    // launchers of the (N, numSS_points) instantiation of the solve kernels in use
    void *var_dl;                            // dlopen handle when that instantiation lives in its own shared object (lmpc_variant.hip)
    int mw_max_batch, mw2_max_batch, n_cu;   // largest batch served by the four-wave / the two-wave solve kernel
    bool k1_force16;                         // regression kernel: never the 8-rows-per-lane scan (launch_k1)
    int k1_qg_force;                         // (experiments, LMPC_K1_QG at lmpc_create: queries per work-group of the regression kernel; 0 = the rule of k1_grid)
    int fuse_k1;                             // fused step for one-wave batches (LMPC_FUSE=0 turns it off)
    double *ab_pack;                         // global scratch of the one-wave kernel's long-horizon variant ([A_k | B_k] per problem), max_batch x 48 N doubles
    int cd_ok, cd_hasq, cd_mode;             // condensed one-wave kernel (opt-in: libraries built with -DLMPC_WITH_CD only): usable for this configuration / state cost present / LMPC_CD=1 in the environment selects it, at every batch size
    int profiling; bool ev_open; std::vector<evpair> events; lmpc_stats stats;
    // retry pass on demand: launches since the last drain of the stream, the host-mapped ring the kernels flag themselves in, launch counter
    struct pending_solve { lmpc_solve_io io; int B; int epoch; lmpc_dev_params dp; bool shared_abc; };   // dp: the parameter block (selected laps, store pointers) of the launch
    std::vector<pending_solve> pending; int *h_retry, *d_retry; int epoch;
    size_t ab_pack_cap;                      // problems ab_pack holds
    int *w_selStart;                         // lmpc_select_batch: window starts, max_batch x numSS_it
    void *scr_dev; size_t scr_bytes;         // pooled scratch of the small host-buffer entry points (plant step, global position)
    std::vector<char *> gaps;                // guard builds: the 256-byte zones between the work-buffer ranges of the slabs
    unsigned create_flags; int solver_kind;  // lmpc_create_ex flags; which solve kernels serve this context: 0 built-in, 1 variant library, 2 the runtime-(N, S) kernel
    int dbg_capture;                         // lmpc_debug_rollout_capture: rollout sessions also keep the selected safe-set points of the last step (parity probes of the closed loop)
    double *dbg_trace;                       // developer builds (-DLMPC_TRACE): device buffer of the per-iteration side channel, see lmpc_debug_set_trace
    double tr_s[4]; long long tr_n;          // developer trace of lmpc_step_batch's one-QP path (lmpc_debug_step_trace): seconds spent staging / launching / waiting / unstaging
    struct lmpc_rollout_session *ro;
    void *comm; int comm_rank, comm_world;   // RCCL communicator of this rank (lmpc_comm.hip.h); null = single process
    double *ext_rows; size_t ext_rows_bytes;   // staging buffer of lmpc_ss_extend_lap
    void *comm_scr, *comm_scr_h; size_t comm_scr_bytes;   // communicator scratch: device allocation + pinned host mirror (lmpc_comm.hip.h: comm_scratch)
};

#ifdef LMPC_DEV_FAST
// developer build (racinglmpc_amd.build.build_flavour("dev", ["LMPC_DEV_FAST"])): only the N = 12 variants, seconds to compile; LMPC_FORCE_NW=<1|2|4>
// picks the waves per QP at every batch size (1 = the one-wave kernel)
#ifndef LMPC_DEV_N
#define LMPC_DEV_N 12
#endif
static bool builtin_variant(lmpc_variant_api *v, int n, int s) {
    if (n == LMPC_DEV_N && s == 48) return lmpc_variant_fill<LMPC_DEV_N, 48>(v);
    return n == LMPC_DEV_N && s == 0 && lmpc_variant_fill<LMPC_DEV_N, 0>(v);
}
#else
template <int N, int S> static bool try_builtin(lmpc_variant_api *v, int n, int s) { return n == N && s == S && lmpc_variant_fill<N, S>(v); }
static bool builtin_variant(lmpc_variant_api *v, int n, int s) {    // the reference's configurations (main.py:43 N = 14, BASELINE N = 12 / 40) are part of the library
    return try_builtin<8, 0>(v, n, s) || try_builtin<12, 0>(v, n, s) || try_builtin<14, 0>(v, n, s) || try_builtin<20, 0>(v, n, s) || try_builtin<40, 0>(v, n, s) ||
           try_builtin<8, 48>(v, n, s) || try_builtin<12, 48>(v, n, s) || try_builtin<14, 48>(v, n, s) || try_builtin<20, 48>(v, n, s) || try_builtin<40, 48>(v, n, s);
}
#endif
// any other (N, numSS_points): liblmpc_var_N<N>_S<S>.so next to this library
static int load_variant(lmpc_ctx *c, int n, int s) {
    Dl_info info;
    if (!dladdr((const void *)&load_variant, &info) || !info.dli_fname) return set_err(LMPC_E_VARIANT, "variant lookup", "cannot locate liblmpc_hip.so");
    std::string dir(info.dli_fname); const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
    char name[96]; snprintf(name, sizeof(name), "/liblmpc_var_N%d_S%d.so", n, s);
    const std::string path = dir + name;
    void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) return set_err(LMPC_E_VARIANT, "solve-kernel variant not built (racinglmpc_amd.build.build_variant(N, numSS_points) makes it)", path.c_str());
    typedef int (*get_t)(lmpc_variant_api *, int);
    get_t get = (get_t)dlsym(h, "lmpc_variant_get");
    const int rc = get ? get(&c->var, LMPC_VARIANT_ABI) : -1;
    if (rc != 0 || c->var.N != n || c->var.S != s) { dlclose(h); return set_err(rc == -2 ? LMPC_E_ARG : LMPC_E_VARIANT, rc == -2 ? "variant exceeds the LDS of a CU" : "stale or foreign variant library (rebuild it)", path.c_str()); }
    c->var_dl = h;
    return LMPC_OK;
}
static int pick_solver(lmpc_ctx *c) {
    const int n = c->cfg.N, s = c->cfg.numSS_it > 0 ? c->cfg.numSS_points : 0;
    c->solver_kind = 0;
    if (!(c->create_flags & LMPC_CREATE_FORCE_RUNTIME_KERNEL)) {
        if (builtin_variant(&c->var, n, s)) return LMPC_OK;
        c->solver_kind = 1;
        const int rc = load_variant(c, n, s);
        if (rc == LMPC_OK || !(c->create_flags & LMPC_CREATE_RUNTIME_KERNEL)) return rc;
    }
    // no built-in instantiation, no variant library (and nobody to compile one): the runtime-(N, S) kernel (lmpc_solve_rt.hip.h)
    c->solver_kind = 2;
    if (!lmpc_variant_fill_rt(&c->var, n, s)) return set_err(LMPC_E_ARG, "runtime solve kernel", "this (N, numSS_points) exceeds the LDS of a CU");
    return LMPC_OK;
}

extern "C" {

const char *lmpc_last_error(void) { return g_err.c_str(); }
const char *lmpc_active_knobs(void) { std::lock_guard<std::mutex> lk(g_knob_mu); static thread_local std::string copy; copy = g_knobs; return copy.c_str(); }
int lmpc_version(void) { return 100; }
int lmpc_device_memory(int device, unsigned long long *free_bytes, unsigned long long *total_bytes) {
    ARGCHK(free_bytes && total_bytes && device >= 0);
    size_t f = 0, t = 0;
    HIPCHK(hipSetDevice(device)); HIPCHK(hipMemGetInfo(&f, &t));
    *free_bytes = (unsigned long long)f; *total_bytes = (unsigned long long)t;
    return LMPC_OK;
}

int lmpc_config_default(lmpc_config *c) {
    if (!c) return LMPC_E_ARG;
    memset(c, 0, sizeof(*c));
    c->N = 12; c->numSS_it = 4; c->numSS_points = 48; c->trToUse = 4; c->maxNumPoint = 7;
    c->h = 5.0; c->lamb = 0.0; c->dt = 0.1;
    const double sc[5] = {0.1, 1, 1, 1, 1}; memcpy(c->scaling, sc, sizeof(sc));
    c->dR[0] = 5.0; c->dR[1] = 50.0; c->Qslack[0] = 5.0; c->Qslack[1] = 25.0;       // initControllerParameters.py:50-54
    for (int i = 0; i < 6; i++) c->QtermSlack[i * 7] = 500.0;
    c->Fx[5] = 1.0; c->Fx[11] = -1.0; c->bx[0] = c->bx[1] = 0.4;
    const double fu[8] = {1, 0, -1, 0, 0, 1, 0, -1}; memcpy(c->Fu, fu, sizeof(fu));
    c->bu[0] = c->bu[1] = 0.5; c->bu[2] = c->bu[3] = 10.0;
    c->track_rows = 0; c->trackLength = 0.0;
    c->device = 0; c->max_batch = 256; c->max_laps = 64; c->max_lap_len = 2048;
    c->tol_gap = 1e-11; c->tol_res = 1e-9; c->reg_lambda = 1e-6; c->max_iter = 40; c->slacks = 1;
    return LMPC_OK;
}

static void fill_params(lmpc_ctx *c) {
    const lmpc_config &f = c->cfg; lmpc_dev_params &p = c->dp;
    memset(&p, 0, sizeof(p));
    p.N = f.N; p.L = f.numSS_it; p.S = f.numSS_it > 0 ? f.numSS_points : 0; p.ppl = f.numSS_it > 0 ? f.numSS_points / f.numSS_it : 0;
    p.trToUse = f.trToUse; p.maxNumPoint = f.maxNumPoint; p.h = f.h; p.lamb = f.lamb; p.dt = f.dt;
    memcpy(p.scaling, f.scaling, sizeof(p.scaling));
    for (int i = 0; i < 36; i++) { p.Q2[i] = 2 * f.Q[i]; p.Qf2[i] = 2 * f.Qf[i]; }
    for (int i = 0; i < 4; i++) p.R2[i] = 2 * f.R[i];
    p.dR2[0] = 2 * f.dR[0]; p.dR2[1] = 2 * f.dR[1]; p.a_s = 2 * f.Qslack[0]; p.c_s = f.Qslack[1];
    for (int i = 0; i < 6; i++) { p.T2[i] = 2 * f.QtermSlack[i * 7]; p.xRef[i] = f.xRef[i]; }
    memcpy(p.Fx, f.Fx, sizeof(p.Fx)); memcpy(p.bx, f.bx, sizeof(p.bx)); memcpy(p.Fu, f.Fu, sizeof(p.Fu)); memcpy(p.bu, f.bu, sizeof(p.bu));
    memcpy(p.track, f.track, sizeof(double) * 6 * f.track_rows); p.track_rows = f.track_rows; p.TL = f.trackLength;
    p.tol_gap = f.tol_gap; p.tol_res = f.tol_res; p.reg = f.reg_lambda; p.max_iter = f.max_iter; p.slacks = f.slacks ? 1 : 0;
    if (!f.slacks) {
        // MPCParams.slacks = False (PredictiveControllers.py:184-198): hard lane rows.  The solve kernels keep their slack variables with a quadratic
        // weight of 1e12 and no linear term: a lane row can then be exceeded by mu / 2e12 ~ 1e-11 m, a thousand times below the solver's own tolerance,
        // and the stage Hessians stay bounded -- a hard row's barrier weight mu^2 / gap enters them directly and costs the Riccati recursion
        // the dual residual below gap ~ 1e-9 (NumPy model: |x - x*| stalls at 1e-7 .. 1e-5 on the fixture; with the penalty 2e-9 in <= 10 iterations).
        p.a_s = 2.0e12; p.c_s = 0.0;
    }
    p.lap_stride = f.max_lap_len; p.mstore = c->mstore; p.sstore = c->sstore;
    p.mquant = c->mquant; p.mqpar = c->mqpar; p.mq_chunks = c->mq_chunks;
}

}  // extern "C" (helpers)

#define LMPC_RETRY_RING 64
#define LMPC_SLAB_COPY_MAX ((size_t)256 * 1024)     // one-copy path of lmpc_step_batch up to this many output bytes (batch 1 .. ~16)

// everything of lmpc_create that can fail; the caller destroys the context on any error (one failure path, no leaks)
static int create_body(lmpc_ctx *c) {
    const lmpc_config *cfg = &c->cfg;
    HIPCHK(hipSetDevice(cfg->device));
    // an unsupported (N, numSS_points) pair is an ordinary user error: find that out before anything is allocated
    c->k1_force16 = dev_knob("LMPC_K1_RPL16") != nullptr;
    { const char *e = dev_knob("LMPC_K1_QG"); c->k1_qg_force = e ? atoi(e) : 0; }
    { const int rc = pick_solver(c); if (rc) return rc; }
    {   // batches that leave SIMDs idle (B <= number of CUs) run the 4-waves-per-QP kernel
        const char *e = dev_knob("LMPC_MW_MAX_BATCH"); hipDeviceProp_t prop; int cus = 256;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        c->mw_max_batch = e ? atoi(e) : cus; c->n_cu = cus;
    }
    // a variant whose one-wave LDS footprint leaves room for ONE QP per CU only would keep three SIMDs idle at any batch size and runs four waves
    // per QP throughout.  None of the built-in horizons is in that class any more: N = 40 fits two QPs per CU with [A_k | B_k] in LDS (53.5 KB)
    // and four with it in global memory (38.2 KB, lmpc_variant.hip.h: use_abg), so N = 40 follows the ordinary rule below (one wave from batch 257)
    if (!dev_knob("LMPC_MW_MAX_BATCH") && 2 * (c->var.lds_1w_abg ? c->var.lds_1w_abg : c->var.lds_1w) > 160 * 1024) c->mw_max_batch = 1 << 30;
    // Two waves per QP between the four-wave and the one-wave regime: how far up depends on the horizon -- the longer the horizon, the
    // larger the share of the Newton step that is sequential (the helper wave only waits) and the fewer QPs of the multi-wave LDS layout fit a
    // CU.  Measured (solve kernel, ms; two waves | one wave): N=12 B=1024 0.30 | 0.40, B=2048 (two rounds | one round) 0.56 | 0.47;  N=14 B=512 0.33 | 0.42, B=1024 0.48 | 0.44;
    // N=20 B=512 0.43 | 0.56, B=1024 0.79 | 0.63;  N=40 B=512 1.35 | 1.00, B=1024 2.33 | 1.78 (and four waves: 1.18, 2.05).
    c->mw2_max_batch = c->mw_max_batch == c->n_cu ? (cfg->N <= 12 ? 4 : cfg->N <= 24 ? 2 : 0) * c->n_cu : 0;

    // safe sets wider than 58 points (several terminal-block columns per lane): the two-wave kernel is built for one wave per SIMD
    // (two QPs per CU), so its regime ends at two QPs per CU
    if (cfg->numSS_it > 0 && cfg->numSS_points + 6 > WAVE && c->mw2_max_batch > 2 * c->n_cu) c->mw2_max_batch = 2 * c->n_cu;
    // ... and never beyond what is resident at once: a second round of two-wave work-groups loses to the one-wave kernel (N = 12, batch 1024,
    // solve kernel: 0.425 ms in two rounds against 0.349 ms; round 4 found it out the hard way -- 512 bytes more static LDS, three QPs per CU instead of four)
    if (c->var.occ_mw2 > 0 && c->mw2_max_batch > c->var.occ_mw2 * c->n_cu) c->mw2_max_batch = c->var.occ_mw2 * c->n_cu;
    if (c->var.lds_mw == 0) { c->mw_max_batch = 0; c->mw2_max_batch = 0; }        // (a variant without multi-wave kernels)
    if (const char *e = dev_knob("LMPC_MW2_MAX_BATCH")) c->mw2_max_batch = atoi(e);        // (experiments)
    // Fused step (regression inside the one-wave solve kernel): bit-identical results, 42 MB less HBM traffic per step at batch 4096, but
    // MEASURED SLOWER -- 1.37 vs 1.01 + 0.23 ms at batch 4096, 2.48 vs 1.81 + 0.44 ms at batch 8192 with six QPs per CU; 1.92 vs 1.45 + 0.41 ms
    // at batch 8192 with eight: the regression's short dependent chains (DPP minima, 5 x 5 Cholesky, scattered L2 reads) want the four waves
    // per SIMD its own kernel gets, the solve kernel runs two.  Off unless LMPC_FUSE=1.
    { const char *e = dev_knob("LMPC_FUSE"); c->fuse_k1 = e ? atoi(e) : 0; }
    if (c->fuse_k1 && (size_t)(54 * cfg->N + k1_fused_doubles(cfg->N, cfg->trToUse, cfg->maxNumPoint)) * sizeof(double) > (size_t)160 * 1024) c->fuse_k1 = 0;   // (many laps: the regression's work space does not fit beside the solve)
    {   // condensed kernel (lmpc_solve_cd.hip.h): built for 2N <= 32 and one terminal-block column per lane; it takes the state cost as diagonal Q, Qf
        bool diag = true; c->cd_hasq = 0;
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) {
            if (i != j && (cfg->Q[i * 6 + j] != 0.0 || cfg->Qf[i * 6 + j] != 0.0)) diag = false;
            if (cfg->Q[i * 6 + j] != 0.0 || cfg->Qf[i * 6 + j] != 0.0) c->cd_hasq = 1;
            if (i == j && (cfg->Q[i * 6 + j] < 0.0 || cfg->Qf[i * 6 + j] < 0.0)) diag = false;
        }
        c->cd_ok = c->var.lds_cd > 0 && diag;
        const char *e = dev_knob("LMPC_CD"); c->cd_mode = e ? atoi(e) : 0;
    }
    HIPCHK(hipStreamCreate(&c->stream));
    if (c->var.lds_1w_abg > 0 && !dev_knob("LMPC_NO_ABG")) { HIPCHK(g_malloc(&c->ab_pack, sizeof(double) * 48 * (size_t)cfg->N * (size_t)cfg->max_batch)); c->ab_pack_cap = (size_t)cfg->max_batch; }
    HIPCHK(hipHostMalloc(&c->h_retry, sizeof(int) * LMPC_RETRY_RING, hipHostMallocMapped));
    memset(c->h_retry, 0, sizeof(int) * LMPC_RETRY_RING);
    HIPCHK(hipHostGetDevicePointer((void **)&c->d_retry, c->h_retry, 0));
    const size_t store_elems = (size_t)cfg->max_laps * LMPC_COLS * cfg->max_lap_len;
    HIPCHK(g_malloc(&c->mstore, store_elems * sizeof(double)));
    HIPCHK(g_malloc(&c->sstore, store_elems * sizeof(double)));
    HIPCHK(hipMemset(c->mstore, 0, store_elems * sizeof(double)));
    c->mq_chunks = (cfg->max_lap_len + K1_CHUNK - 1) / K1_CHUNK;
    HIPCHK(g_malloc(&c->mquant, (size_t)cfg->max_laps * 3 * cfg->max_lap_len * sizeof(unsigned)));
    HIPCHK(g_malloc(&c->mqpar, (size_t)cfg->max_laps * c->mq_chunks * 6 * sizeof(double)));
    HIPCHK(hipMemset(c->mquant, 0, (size_t)cfg->max_laps * 3 * cfg->max_lap_len * sizeof(unsigned)));
    HIPCHK(hipMemset(c->mqpar, 0, (size_t)cfg->max_laps * c->mq_chunks * 6 * sizeof(double)));
    HIPCHK(hipMemset(c->sstore, 0, store_elems * sizeof(double)));
    const size_t B = cfg->max_batch, N = cfg->N, S = cfg->numSS_it > 0 ? cfg->numSS_points : 0, M = 8 * N + S;
    // two passes over the same list: sizes first, then pointers into the slabs (every range 256-byte aligned)
    for (int pass = 0; pass < 2; pass++) {
        size_t oi = 0, oo = 0;
#ifdef LMPC_GUARD      // (guard builds: a 256-byte zone behind every work-buffer range, see check_slab_gaps)
#define SLAB_GAP(slab, off) do { if (pass) c->gaps.push_back(c->slab + off); off += LMPC_GUARD_BYTES; } while (0)
#else
#define SLAB_GAP(slab, off) do { } while (0)
#endif
#define SLAB(ptr, n, slab, off) do { const size_t bytes_ = (std::max<size_t>((n), 1) * sizeof(*c->ptr) + 255) & ~(size_t)255; \
                                     if (pass) c->ptr = (decltype(c->ptr))(c->slab + off); off += bytes_; SLAB_GAP(slab, off); } while (0)
        SLAB(w_x0, B * 6, slab_in, oi); SLAB(w_xLin, B * (N + 1) * 6, slab_in, oi); SLAB(w_uLin, B * N * 2, slab_in, oi); SLAB(w_uOld, B * 2, slab_in, oi);
        SLAB(w_zt, B * 6, slab_in, oi); SLAB(w_xPP, B * (N + 1) * 6, slab_in, oi); SLAB(w_hasPred, B, slab_in, oi); SLAB(w_tstep, B, slab_in, oi);
        SLAB(w_xPred, B * (N + 1) * 6, slab_out, oo); SLAB(w_uPred, B * N * 2, slab_out, oo); SLAB(w_slack, B * N * 2, slab_out, oo); SLAB(w_lam, B * S, slab_out, oo);
        SLAB(w_sT, B * 6, slab_out, oo); SLAB(w_ztN, B * 6, slab_out, oo); SLAB(w_ztuN, B * 2, slab_out, oo); SLAB(w_ssSel, B * S * 6, slab_out, oo);
        SLAB(w_qSel, B * S, slab_out, oo); SLAB(w_mu, B * M, slab_out, oo); SLAB(w_A, B * N * 36, slab_out, oo); SLAB(w_B, B * N * 12, slab_out, oo);
        SLAB(w_C, B * N * 6, slab_out, oo); SLAB(w_status, B, slab_out, oo); SLAB(w_iters, B, slab_out, oo); SLAB(w_resid, B * 3, slab_out, oo);
        SLAB(w_succ, B * S * 6, slab_out, oo); SLAB(w_succU, B * S * 2, slab_out, oo); SLAB(w_ztUsed, B * 6, slab_out, oo); SLAB(w_rstatus, B * N, slab_out, oo);
        SLAB(w_selStart, B * (size_t)std::max(cfg->numSS_it, 1), slab_out, oo);
#undef SLAB
#undef SLAB_GAP
        if (pass) { for (char *g_ : c->gaps) HIPCHK(hipMemset(g_, 0xA5, LMPC_GUARD_BYTES)); }
        if (!pass) {
            c->slab_in_bytes = oi; c->slab_out_bytes = oo;
            HIPCHK(g_malloc(&c->slab_in, oi)); HIPCHK(g_malloc(&c->slab_out, oo));
            HIPCHK(hipMemset(c->slab_in, 0, oi)); HIPCHK(hipMemset(c->slab_out, 0, oo));
            if (oo <= LMPC_SLAB_COPY_MAX) {
                HIPCHK(hipHostMalloc(&c->h_in, oi, hipHostMallocMapped)); HIPCHK(hipHostMalloc(&c->h_out, oo, hipHostMallocMapped)); memset(c->h_in, 0, oi); memset(c->h_out, 0, oo);
                HIPCHK(hipHostGetDevicePointer((void **)&c->dm_in, c->h_in, 0)); HIPCHK(hipHostGetDevicePointer((void **)&c->dm_out, c->h_out, 0));
            }
        }
    }
    return LMPC_OK;
}

extern "C" {
int lmpc_destroy(lmpc_ctx *c);
int lmpc_create_ex(const lmpc_config *cfg, unsigned flags, lmpc_ctx **out);
int lmpc_create(const lmpc_config *cfg, lmpc_ctx **out) { return lmpc_create_ex(cfg, 0u, out); }
int lmpc_solver_kind(lmpc_ctx *c) { return c ? c->solver_kind : LMPC_E_ARG; }
int lmpc_create_ex(const lmpc_config *cfg, unsigned flags, lmpc_ctx **out) {
    ARGCHK(cfg && out);
    ARGCHK(cfg->N >= 2 && cfg->N <= LMPC_MAX_N);
    ARGCHK(cfg->numSS_it >= 0 && cfg->numSS_it <= LMPC_MAX_USED_LAPS && cfg->trToUse >= 0 && cfg->trToUse <= LMPC_MAX_USED_LAPS);
    ARGCHK(cfg->maxNumPoint >= 1 && cfg->maxNumPoint <= 8);
    if (cfg->numSS_it > 0) {
        ARGCHK(cfg->numSS_points % cfg->numSS_it == 0 && cfg->numSS_points <= LMPC_MAX_SS_POINTS && cfg->numSS_points / cfg->numSS_it + 1 <= WAVE);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) if (i != j) ARGCHK(cfg->QtermSlack[i * 6 + j] == 0.0);   // diagonal terminal-slack weight
        for (int i = 0; i < 6; i++) ARGCHK(cfg->QtermSlack[i * 7] > 0.0);
    }
    ARGCHK(cfg->track_rows >= 0 && cfg->track_rows <= LMPC_MAX_TRACK_ROWS);
    ARGCHK(cfg->slacks || cfg->numSS_it == 0);          // hard lane rows: plain MPC only (the reference's LMPC.unpackSolution mis-slices without slack variables)
    ARGCHK(cfg->max_batch >= 1 && cfg->max_laps >= 1 && cfg->max_lap_len >= 8);
    lmpc_ctx *c = new lmpc_ctx();                      // value-initialised: every pointer starts as nullptr, so lmpc_destroy is safe at any point
    c->cfg = *cfg; c->create_flags = flags; c->profiling = 0; c->ro = nullptr; c->var_dl = nullptr; c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
    memset(&c->stats, 0, sizeof(c->stats));
    int rc = create_body(c);
    if (rc != LMPC_OK) { const std::string keep = g_err; lmpc_destroy(c); g_err = keep; return rc; }
    fill_params(c);
    *out = c;
    return LMPC_OK;
}

static void rollout_free(lmpc_ctx *c);
static void check_slab_gaps(lmpc_ctx *c) {
#ifdef LMPC_GUARD
    (void)hipDeviceSynchronize();
    for (char *g_ : c->gaps) check_guard_zone(g_, "work-buffer range of a slab");
#else
    (void)c;
#endif
}
int lmpc_debug_guard_failures(void) { return g_guard_failures; }      /* guard builds (-DLMPC_GUARD): overruns found so far; 0 in ordinary builds */
int lmpc_destroy(lmpc_ctx *c) {
    if (!c) return LMPC_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rollout_free(c);
    if (c->comm) { (void)ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c->ext_rows) (void)g_free(c->ext_rows);
    if (c->scr_dev) (void)g_free(c->scr_dev);
    if (c->comm_scr) (void)g_free(c->comm_scr);
    if (c->comm_scr_h) (void)hipHostFree(c->comm_scr_h);
    for (auto &e : c->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    void *ptrs[] = {c->mstore, c->sstore, c->mquant, c->mqpar, c->slab_in, c->slab_out, c->ab_pack};      // (the w_* work buffers are ranges of the two slabs)
    check_slab_gaps(c);
    for (void *q : ptrs) if (q) (void)g_free(q);
    if (c->h_retry) (void)hipHostFree(c->h_retry);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->var_dl) dlclose(c->var_dl);
    delete c;
    return LMPC_OK;
}

static int resolve_retries(lmpc_ctx *c);
// Every entry point that changes what a deferred retry pass would read -- the lap stores, caller-visible device buffers (lmpc_dev_upload) -- or
// hands results to someone else (lmpc_comm_allgather_dev, lmpc_ss_get_qfun) first gives the launches still pending their retry pass: a flagged
// problem is then re-solved against the data of ITS launch, never against a changed safe set or overwritten inputs.  (No launch pending -- the
// drop-in flow, where lmpc_step_batch resolves before it returns -- costs nothing: no stream drain is added.)
#define RESOLVE_PENDING() do { const int rc_ = resolve_retries(c); if (rc_) return rc_; } while (0)

// ---------------------------------------------------------------------------------------------- stores
// The reference keeps its laps in Python lists that grow without bound (PredictiveControllers.py:418-445, PredictiveModel.py:35-46; addPoint
// appends a row per closed-loop step, :466-474).  max_laps / max_lap_len of lmpc_config are therefore INITIAL capacities: when a lap or a row
// does not fit, both stores (and the K1 prefilter image) move to allocations of at least twice the size -- strided device-to-device copies,
// [lap][column][row] with the new row stride -- and the device parameter block is rebuilt.  Amortised O(1) per stored row; no kernel is in
// flight while it happens (the stream is drained first).
static int grow_stores(lmpc_ctx *c, int need_laps, int need_len) {
    const int old_laps = c->cfg.max_laps, old_len = c->cfg.max_lap_len;
    int new_laps = old_laps, new_len = old_len;
    while (new_laps < need_laps) new_laps *= 2;
    while (new_len < need_len) new_len *= 2;
    if (new_laps == old_laps && new_len == old_len) return LMPC_OK;
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    const int old_chunks = c->mq_chunks, new_chunks = (new_len + K1_CHUNK - 1) / K1_CHUNK;
    const size_t elems = (size_t)new_laps * LMPC_COLS * new_len;
    double *nm = nullptr, *ns = nullptr, *np_ = nullptr; unsigned *nq = nullptr;
    auto fail = [&](int rc) { if (nm) (void)g_free(nm); if (ns) (void)g_free(ns); if (nq) (void)g_free(nq); if (np_) (void)g_free(np_); return rc; };
#define GROWCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(set_err(LMPC_E_HIP, #call, hipGetErrorString(e_))); } while (0)
    GROWCHK(g_malloc(&nm, elems * sizeof(double))); GROWCHK(g_malloc(&ns, elems * sizeof(double)));
    GROWCHK(g_malloc(&nq, (size_t)new_laps * 3 * new_len * sizeof(unsigned))); GROWCHK(g_malloc(&np_, (size_t)new_laps * new_chunks * 6 * sizeof(double)));
    GROWCHK(hipMemset(nm, 0, elems * sizeof(double))); GROWCHK(hipMemset(ns, 0, elems * sizeof(double)));
    GROWCHK(hipMemset(nq, 0, (size_t)new_laps * 3 * new_len * sizeof(unsigned))); GROWCHK(hipMemset(np_, 0, (size_t)new_laps * new_chunks * 6 * sizeof(double)));
    const size_t nml = c->m_len.size(), nsl = c->s_len.size();
    if (nml) {
        GROWCHK(hipMemcpy2D(nm, (size_t)new_len * 8, c->mstore, (size_t)old_len * 8, (size_t)old_len * 8, nml * LMPC_COLS, hipMemcpyDeviceToDevice));
        GROWCHK(hipMemcpy2D(nq, (size_t)new_len * 4, c->mquant, (size_t)old_len * 4, (size_t)old_len * 4, nml * 3, hipMemcpyDeviceToDevice));
        GROWCHK(hipMemcpy2D(np_, (size_t)new_chunks * 48, c->mqpar, (size_t)old_chunks * 48, (size_t)old_chunks * 48, nml, hipMemcpyDeviceToDevice));
    }
    if (nsl) GROWCHK(hipMemcpy2D(ns, (size_t)new_len * 8, c->sstore, (size_t)old_len * 8, (size_t)old_len * 8, nsl * LMPC_COLS, hipMemcpyDeviceToDevice));
#undef GROWCHK
    (void)g_free(c->mstore); (void)g_free(c->sstore); (void)g_free(c->mquant); (void)g_free(c->mqpar);
    c->mstore = nm; c->sstore = ns; c->mquant = nq; c->mqpar = np_; c->mq_chunks = new_chunks;
    c->cfg.max_laps = new_laps; c->cfg.max_lap_len = new_len;
    const lmpc_dev_params keep = c->dp;
    fill_params(c);
    // (per-launch parts of the parameter block survive: selected slots / lengths are refreshed by every launch anyway)
    memcpy(c->dp.mslot, keep.mslot, sizeof(keep.mslot)); memcpy(c->dp.mlen, keep.mlen, sizeof(keep.mlen));
    memcpy(c->dp.sslot, keep.sslot, sizeof(keep.sslot)); memcpy(c->dp.sslen, keep.sslen, sizeof(keep.sslen)); memcpy(c->dp.sslapid, keep.sslapid, sizeof(keep.sslapid));
    c->dp.cur_it = keep.cur_it;
    return LMPC_OK;
}

static int upload_lap(lmpc_ctx *c, bool model, int slot, const double *x, const double *u, const double *qf, int T) {
    { const int rc = grow_stores(c, slot + 1, T); if (rc) return rc; }
    double *store = model ? c->mstore : c->sstore;
    std::vector<double> col((size_t)T);
    double *base = store + (size_t)slot * LMPC_COLS * c->cfg.max_lap_len;
    for (int cI = 0; cI < LMPC_COLS; cI++) {
        if (cI == 8 && !qf) continue;
        for (int t = 0; t < T; t++) col[t] = cI < 6 ? x[(size_t)t * 6 + cI] : (cI < 8 ? u[(size_t)t * 2 + (cI - 6)] : qf[t]);
        HIPCHK(hipMemcpy(base + (size_t)cI * c->cfg.max_lap_len, col.data(), sizeof(double) * T, hipMemcpyHostToDevice));
    }
    return LMPC_OK;
}

// K1 prefilter image of one model-store lap (PredictiveModel.py:180-197 scans (vx, vy, wz, delta, a) . scaling of rows 0..T-2):
// per 1024-row chunk, every scaled feature is mapped to 16-bit fixed point over the chunk's [min, max] with one common scale
// 65535 / (widest range), packed as (vx | vy << 16), (wz | delta << 16), (a).  The regress kernel ranks rows by integer L1 distances on this image and re-evaluates the survivors
// in FP64, so the image never decides anything by itself; it only has to be within a few units of the exact scaled values.
static int quantise_lap(lmpc_ctx *c, int slot, const double *x, const double *u, int T) {
    const int ls = c->cfg.max_lap_len, nrows = T - 1;
    std::vector<unsigned> q((size_t)3 * ls, 0u);
    std::vector<double> par((size_t)c->mq_chunks * 6, 0.0);
    auto feat = [&](int t, int k) { return (k < 3 ? x[(size_t)t * 6 + k] : u[(size_t)t * 2 + (k - 3)]) * c->cfg.scaling[k]; };
    for (int t0 = 0, ch = 0; t0 < nrows; t0 += K1_CHUNK, ch++) {
        const int t1 = std::min(nrows, t0 + K1_CHUNK);
        double lo[5], rmax = 0.0;
        for (int k = 0; k < 5; k++) {
            double l = INFINITY, h = -INFINITY;
            for (int t = t0; t < t1; t++) { const double v = feat(t, k); if (std::isfinite(v)) { l = std::min(l, v); h = std::max(h, v); } }
            if (!(l <= h)) { l = 0.0; h = 0.0; }
            lo[k] = l; rmax = std::max(rmax, std::max(h - l, 1e-9 * std::max(std::fabs(l), std::fabs(h))));
        }
        const double sc = 65535.0 / std::max(rmax, 1e-300);
        for (int k = 0; k < 5; k++) {
            par[(size_t)ch * 6 + k] = lo[k];
            for (int t = t0; t < t1; t++) {
                const double v = (feat(t, k) - lo[k]) * sc;
                const unsigned qq = std::isfinite(v) ? (unsigned)std::min(std::max(v, 0.0), 65535.0) : 0u;
                q[(size_t)(k >> 1) * ls + t] |= (k & 1) ? qq << 16 : qq;
            }
        }
        par[(size_t)ch * 6 + 5] = sc;
    }
    HIPCHK(hipMemcpy(c->mquant + (size_t)slot * 3 * ls, q.data(), q.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->mqpar + (size_t)slot * c->mq_chunks * 6, par.data(), par.size() * sizeof(double), hipMemcpyHostToDevice));
    return LMPC_OK;
}

int lmpc_model_add_trajectory(lmpc_ctx *c, const double *x, const double *u, int T) {
    ARGCHK(c && x && u && T >= 2);
    HIPCHK(hipSetDevice(c->cfg.device));
    RESOLVE_PENDING();
    const int slot = (int)c->m_len.size();
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = upload_lap(c, true, slot, x, u, nullptr, T); if (rc) return rc;                 // (grows the stores when the lap does not fit)
    rc = quantise_lap(c, slot, x, u, T); if (rc) return rc;
    c->m_len.push_back(T);
    // PredictiveModel.addTrajectory (PredictiveModel.py:35-46): append if empty or T >= last, else insert before first longer lap
    if (c->m_order.empty() || T >= c->m_len[c->m_order.back()]) c->m_order.push_back(slot);
    else { size_t i = 0; while (i < c->m_order.size() && !(T < c->m_len[c->m_order[i]])) i++; c->m_order.insert(c->m_order.begin() + i, slot); }
    return LMPC_OK;
}
int lmpc_model_num_laps(lmpc_ctx *c, int *n) { ARGCHK(c && n); *n = (int)c->m_order.size(); return LMPC_OK; }
int lmpc_model_replace_lap(lmpc_ctx *c, int pos, const double *x, const double *u, int T) {
    ARGCHK(c && x && u && pos >= 0 && pos < (int)c->m_order.size());
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    ARGCHK(T == c->m_len[c->m_order[pos]]);
    int rc = upload_lap(c, true, c->m_order[pos], x, u, nullptr, T); if (rc) return rc;
    return quantise_lap(c, c->m_order[pos], x, u, T);
}

int lmpc_ss_add_trajectory(lmpc_ctx *c, const double *x, const double *u, int T) {
    ARGCHK(c && x && u && T >= 1);
    HIPCHK(hipSetDevice(c->cfg.device));
    RESOLVE_PENDING();
    const int lap = (int)c->s_len.size();
    // LMPC.computeCost (PredictiveControllers.py:447-464)
    std::vector<double> cost((size_t)T, 10000.0);
    for (int i = 0; i < T; i++) {
        const int r = T - 1 - i;
        if (i == 0) cost[r] = 0;
        else if (x[(size_t)r * 6 + 4] < c->cfg.trackLength) cost[r] = cost[r + 1] + 1;
        else cost[r] = 0;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = upload_lap(c, false, lap, x, u, cost.data(), T); if (rc) return rc;
    c->s_len.push_back(T); c->s_laptime.push_back(T); c->s_qlast.push_back(cost[T - 1]); c->s_q0.push_back(cost[0]);
    return LMPC_OK;
}

__global__ void lmpc_store_row_kernel(double *base, int stride, int row, double v0, double v1, double v2, double v3, double v4, double v5, double v6, double v7, double v8) {
    const double v[9] = {v0, v1, v2, v3, v4, v5, v6, v7, v8};
    if (threadIdx.x < LMPC_COLS) base[(size_t)threadIdx.x * stride + row] = v[threadIdx.x];
}

int lmpc_ss_add_point(lmpc_ctx *c, const double *x, const double *u) {
    ARGCHK(c && x && u);
    if (c->s_len.empty()) return set_err(LMPC_E_STATE, "addPoint before any addTrajectory", "");
    HIPCHK(hipSetDevice(c->cfg.device));
    RESOLVE_PENDING();
    const int lap = (int)c->s_len.size() - 1, row = c->s_len[lap];
    if (row >= c->cfg.max_lap_len) { const int rc = grow_stores(c, c->cfg.max_laps, row + 1); if (rc) return rc; }
    const double q = c->s_qlast[lap] - 1.0;                        // :474
    double *base = c->sstore + (size_t)lap * LMPC_COLS * c->cfg.max_lap_len;
    hipLaunchKernelGGL(lmpc_store_row_kernel, dim3(1), dim3(64), 0, c->stream, base, c->cfg.max_lap_len, row,
                       x[0], x[1], x[2], x[3], x[4] + c->cfg.trackLength, x[5], u[0], u[1], q);   // :472-473
    HIPCHK(hipGetLastError());
    c->s_len[lap] = row + 1; c->s_qlast[lap] = q;
    return LMPC_OK;
}
int lmpc_ss_replace_lap(lmpc_ctx *c, int lap, const double *x, const double *u, const double *qfun, int T) {
    ARGCHK(c && x && u && qfun && lap >= 0 && lap < (int)c->s_len.size() && T >= 1);
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    int rc = upload_lap(c, false, lap, x, u, qfun, T); if (rc) return rc;
    c->s_len[lap] = T; c->s_qlast[lap] = qfun[T - 1]; c->s_q0[lap] = qfun[0];
    return LMPC_OK;
}
int lmpc_ss_set_selected(lmpc_ctx *c, const int *laps, int n) {
    ARGCHK(c && n >= 0 && n <= LMPC_MAX_USED_LAPS && (n == 0 || laps));
    if (n == 0) c->s_override.clear(); else c->s_override.assign(laps, laps + n);
    return LMPC_OK;
}
int lmpc_ss_num_laps(lmpc_ctx *c, int *n) { ARGCHK(c && n); *n = (int)c->s_len.size(); return LMPC_OK; }
int lmpc_ss_get_qfun(lmpc_ctx *c, int lap, double *qfun, int *T) {
    ARGCHK(c && T && lap >= 0 && lap < (int)c->s_len.size());
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    *T = c->s_len[lap];
    if (qfun) HIPCHK(hipMemcpy(qfun, c->sstore + ((size_t)lap * LMPC_COLS + 8) * c->cfg.max_lap_len, sizeof(double) * c->s_len[lap], hipMemcpyDeviceToHost));
    return LMPC_OK;
}

int lmpc_store_read_lap(lmpc_ctx *c, int store, int lap, double *x, double *u, double *qfun, int *T) {
    // checkpoint / resume ("optional: dump lap stores as .npz"; the reference imports pickle for it and never uses it, main.py:36): rows of one stored lap back on
    // the host.  store 0: regression store, `lap` = position in its sorted order (PredictiveModel.xStored[lap]); store 1: safe set, `lap` = index in addTrajectory order, with
    // the rows LMPC.addPoint appended and the Q-function.  T is always set; x (T x 6), u (T x 2), qfun (T) may be NULL.
    ARGCHK(c && T && (store == 0 || store == 1));
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    int slot, rows; const double *base;
    if (store == 0) { ARGCHK(lap >= 0 && lap < (int)c->m_order.size()); slot = c->m_order[lap]; rows = c->m_len[slot]; base = c->mstore; }
    else { ARGCHK(lap >= 0 && lap < (int)c->s_len.size()); slot = lap; rows = c->s_len[lap]; base = c->sstore; }
    *T = rows;
    if (!x && !u && !qfun) return LMPC_OK;
    const size_t ls = c->cfg.max_lap_len;
    std::vector<double> col(rows);
    for (int j = 0; j < LMPC_COLS; j++) {
        if ((j < 6 && !x) || (j >= 6 && j < 8 && !u) || (j == 8 && (!qfun || store == 0))) continue;
        HIPCHK(hipMemcpy(col.data(), base + ((size_t)slot * LMPC_COLS + j) * ls, sizeof(double) * rows, hipMemcpyDeviceToHost));
        for (int r = 0; r < rows; r++) { if (j < 6) x[(size_t)r * 6 + j] = col[r]; else if (j < 8) u[(size_t)r * 2 + (j - 6)] = col[r]; else qfun[r] = col[r]; }
    }
    return LMPC_OK;
}

int lmpc_ss_get_laptime(lmpc_ctx *c, int lap, int *T) { ARGCHK(c && T && lap >= 0 && lap < (int)c->s_laptime.size()); *T = c->s_laptime[lap]; return LMPC_OK; }

// refresh the per-launch part of the device parameter block
static int refresh_params(lmpc_ctx *c, bool need_model, bool need_ss) {
    lmpc_dev_params &p = c->dp;
    if (need_model) {
        if ((int)c->m_order.size() < c->cfg.trToUse || c->cfg.trToUse < 1) return set_err(LMPC_E_STATE, "regression needs trToUse stored laps", "");
        for (int i = 0; i < c->cfg.trToUse; i++) { p.mslot[i] = c->m_order[i]; p.mlen[i] = c->m_len[c->m_order[i]]; }   // usedIt = range(trToUse), PredictiveModel.py:31
    }
    if (need_ss) {
        const int L = c->cfg.numSS_it, nl = (int)c->s_len.size();
        if (nl < L) return set_err(LMPC_E_STATE, "safe set holds fewer than numSS_it laps", "");
        std::vector<int> sel;
        if (!c->s_override.empty()) { sel = c->s_override; if ((int)sel.size() != L) return set_err(LMPC_E_ARG, "selection override must list numSS_it laps", ""); }
        else {   // argsort(LapTime)[0:numSS_it] (:395,402), stable
            std::vector<int> idx(nl); for (int i = 0; i < nl; i++) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return c->s_laptime[a] < c->s_laptime[b]; });
            sel.assign(idx.begin(), idx.begin() + L);
        }
        for (int i = 0; i < L; i++) { if (sel[i] < 0 || sel[i] >= nl) return set_err(LMPC_E_ARG, "selected lap out of range", ""); p.sslot[i] = sel[i]; p.sslapid[i] = sel[i]; p.sslen[i] = c->s_len[sel[i]]; }
        p.cur_it = nl;
    }
    return LMPC_OK;
}

// HIP events around a kernel launch (profiling on).  An event record costs ~4 us on the launch stream -- 15 us per step for the two kernels,
// 5 % of a batch-256 step -- so profiling = k times every k-th launch of a kind only: the average duration is sampled over the run.
static void ev_begin(lmpc_ctx *c, int kind) {
    c->ev_open = false;
    if (!c->profiling) return;
    const long long idx = kind == 0 ? c->stats.n_regress : c->stats.n_solve;
    if (idx % c->profiling) return;
    evpair e; hipEventCreate(&e.a); hipEventCreate(&e.b); e.kind = kind; hipEventRecord(e.a, c->stream); c->events.push_back(e); c->ev_open = true;
}
static void ev_end(lmpc_ctx *c) { if (c->ev_open) hipEventRecord(c->events.back().b, c->stream); c->ev_open = false; }

// K1 launch shape: queries per work-group shrink (12 -> 6 -> 4 -> ... -> 1) until the grid covers the chip (one work-group per CU is resident)
static void k1_grid(lmpc_ctx *c, int B, int *qg, int *nblk) {
    const int N = c->cfg.N, cands[] = {12, 6, 4, 3, 2, 1};
    int q = 1, np = N;
    for (int k = 0; k < 6; k++) {
        q = k1_queries_per_block(cands[k], c->cfg.trToUse, c->cfg.maxNumPoint); np = (N + q - 1) / q;
        if (c->k1_qg_force > 0) { if (cands[k] <= c->k1_qg_force) break; else continue; }
        if ((long long)B * np >= (long long)c->n_cu) break;      // (two or more groups per CU measured no faster: round 2 26-35 us against 27 us at batch 256; round 5, LMPC_K1_QG = 6 / 4 / 3 / 2: 20 / 27 / 29 / 39 us against 22)
    }
    *qg = q; *nblk = B * np;
}
// The regression kernel in the build that fits the launch: the occupancy build when the grid exceeds one work-group per CU, and the
// 8-rows-per-lane scan when every lap in use lies inside its first quantisation chunk half (<= 512 rows; k1_scan_lap).
static void launch_k1(lmpc_ctx *c, int nblk, int B, int qg, const double *xLin, int xstride, const double *uLin, double *dA, double *dB, double *dC, int *dst) {
    bool small = true;
    for (int i = 0; i < c->cfg.trToUse; i++) small = small && c->dp.mlen[i] - 1 <= 8 * WAVE;
    if (c->k1_force16) small = false;                                    // (LMPC_K1_RPL16 at lmpc_create: A / B of the two scan builds, tests/test_gpu_configs.py)
    const bool occ = nblk > c->n_cu;
    auto k = occ ? (small ? lmpc_regress_kernel<true, 8> : lmpc_regress_kernel<true, K1_RPL>) : (small ? lmpc_regress_kernel<false, 8> : lmpc_regress_kernel<false, K1_RPL>);
    hipLaunchKernelGGL(k, dim3(nblk), dim3(K1_NT), 0, c->stream, c->dp, B, qg, xLin, xstride, uLin, dA, dB, dC, dst);
}
static int launch_regress(lmpc_ctx *c, int B, const double *d_xLin, int xstride, const double *d_uLin, double *dA, double *dB, double *dC, int *dst) {
    int rc = refresh_params(c, true, false); if (rc) return rc;
    ev_begin(c, 0);
    int qg, nblk; k1_grid(c, B, &qg, &nblk);
    launch_k1(c, nblk, B, qg, d_xLin, xstride, d_uLin, dA, dB, dC, dst);
    ev_end(c);
    HIPCHK(hipGetLastError());
    c->stats.n_regress++;
    return LMPC_OK;
}
extern "C" int lmpc_solver_waves(lmpc_ctx *c, int B);
// Retry pass on demand.  The adaptive equal / separate step rule can fall into a two-cycle (a separate step feeds (alpha_p - alpha_d) H dw into
// the dual residual, the next equal step is blocked): a few problems per million ended at the iteration limit that way in round 1, and most
// of them converge in 10-12 iterations from the start point with equal steps only (lmpc_solve_kernel<N, S, true>).  Rounds 1-2 launched that
// retry kernel behind EVERY solve -- 4.4 us per step in which all work-groups return at once (1.9 % of a batch-256 step, 3 % at batch 1).  Now a
// problem that needs it says so itself: it writes the launch's epoch into a host-mapped ring (flag_retry, system-scope atomic), and the host
// looks at the ring at its next drain of the stream, before any result of the launch can be read: lmpc_step_batch / lmpc_qp_solve_batch
// right after the solve, the device-resident path in lmpc_dev_sync / lmpc_dev_download / lmpc_dev_free.  The retry kernel skips problems
// whose CURRENT status word is clean, so a flagged launch whose buffers a later launch reused is handled correctly.  Closed-loop rollouts
// (lmpc_rollout_run), where the plant consumes uPred without a host round trip, keep the unconditional launch (immediate = true).
static int resolve_retries(lmpc_ctx *c) {
    if (c->pending.empty()) return LMPC_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = LMPC_OK, launched = 0;
#if defined(LMPC_DEV_FAST) || defined(LMPC_TRACE) || defined(LMPC_TIMING)
    static const bool no_retry = dev_knob("LMPC_NO_RETRY") != nullptr;     // (developer probe, developer flavours only: leave the first pass's output as it is, tools/n40_probe.py)
#else
    const bool no_retry = false;
#endif
    for (auto &pe : c->pending) {
        if (no_retry || c->h_retry[pe.epoch % LMPC_RETRY_RING] != pe.epoch) continue;
        pe.io.retry_flag = nullptr;
        // the retry runs against the parameter block of ITS launch (selected laps, lap lengths, cur_it), not the context's current one
        rc = c->var.launch_retry(c->stream, pe.dp, pe.B, pe.io); if (rc) break;
        if (hipGetLastError() != hipSuccess) { rc = set_err(LMPC_E_HIP, "retry launch", ""); break; }
        c->stats.n_retry++; launched++;
    }
    c->pending.clear();
    // nothing is in flight that could still write the ring: start the epochs over (they never grow past the ring size, so the signed
    // counter cannot overflow and `epoch % ring` stays a valid slot)
    if (rc == LMPC_OK && launched && hipStreamSynchronize(c->stream) != hipSuccess) rc = set_err(LMPC_E_HIP, "hipStreamSynchronize", "retry pass");
    c->epoch = 0; memset(c->h_retry, 0, sizeof(int) * LMPC_RETRY_RING);
    return rc;
}
static int launch_solve(lmpc_ctx *c, int B, const lmpc_solve_io &io_in, bool immediate = false) {
    lmpc_solve_io io = io_in;
    // [A_k | B_k] from global memory only when the batch does not fit the CUs with it in LDS (N = 40: more than two QPs per CU); a batch that
    // fits runs 3 % faster from LDS (measured at N = 40, batch 512: 0.997 vs 1.030 ms)
    io.abPack = nullptr;
    if (c->ab_pack && (long long)B > (long long)(160 * 1024 / c->var.lds_1w) * c->n_cu) {
        if ((size_t)B > c->ab_pack_cap) {                   // rollout sessions may run more problems than max_batch: the scratch follows
            int rc = resolve_retries(c); if (rc) return rc;
            HIPCHK(hipStreamSynchronize(c->stream));
            (void)g_free(c->ab_pack); c->ab_pack = nullptr; c->ab_pack_cap = 0;
            HIPCHK(g_malloc(&c->ab_pack, sizeof(double) * 48 * (size_t)c->cfg.N * (size_t)B)); c->ab_pack_cap = (size_t)B;
        }
        io.abPack = c->ab_pack;
    }
    const bool deferred = (io.mode & 2) && !immediate && !io.tbuf;       // (the retry pass of a traced launch would overwrite the trace rows: traced builds still defer, the rows of a retried problem are its retry's)
    if (deferred) {
        if ((int)c->pending.size() >= LMPC_RETRY_RING - 1) { int rc = resolve_retries(c); if (rc) return rc; }
        c->epoch++; io.retry_epoch = c->epoch; io.retry_flag = c->d_retry + (c->epoch % LMPC_RETRY_RING);
    }
    const bool term = c->cfg.numSS_it > 0;
    int rc = refresh_params(c, false, term && (io.mode & 1)); if (rc) return rc;
    const bool timing_buf = io.tbuf != nullptr;              // (the cycle-stamp build pins the kernel route; the trace build below must not)
#ifdef LMPC_TRACE
    if (!io.tbuf && c->dbg_trace && (io.mode & 2)) io.tbuf = (long long *)c->dbg_trace;
#endif
    ev_begin(c, 1);
    // waves per QP: 4 up to one QP per CU, 2 up to mw2_max_batch (four QPs per CU at N <= 12, see create_body), beyond that the one-wave
    // kernel, whose slim LDS layout keeps eight QPs resident per CU at N = 12
#ifdef LMPC_DEV_FAST
    if (const char *f = dev_knob("LMPC_FORCE_NW")) {
        const int nw = atoi(f);
        rc = nw == 4 ? c->var.launch_mw4(c->stream, c->dp, B, io) : nw == 2 ? c->var.launch_mw2(c->stream, c->dp, B, io) : c->var.launch_1w(c->stream, c->dp, B, io);
    } else
#endif
    rc = (c->cd_ok && c->cd_mode == 1 && !(io.mode & 4)) ? c->var.launch_cd(c->stream, c->dp, B, io, c->cd_hasq)
       : (io.mode & 4) ? c->var.launch_1w(c->stream, c->dp, B, io)        // fused step: the one-wave kernel runs the regression itself
       : (B <= c->mw_max_batch && (!timing_buf || dev_knob("LMPC_TIMING_MW"))) ? c->var.launch_mw4(c->stream, c->dp, B, io)
       : (lmpc_solver_waves(c, B) == 2 && !timing_buf) ? c->var.launch_mw2(c->stream, c->dp, B, io) : c->var.launch_1w(c->stream, c->dp, B, io);
    ev_end(c);
    if (rc) return rc;
    HIPCHK(hipGetLastError());
    c->stats.n_solve++; if (io.mode & 2) c->stats.qp_solved += B;
    if (deferred) c->pending.push_back({io, B, io.retry_epoch, c->dp, io.A == c->w_A || io.Bm == c->w_B || io.C == c->w_C});
    else if (io.mode & 2) {
        rc = c->var.launch_retry(c->stream, c->dp, B, io);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
    }
    return LMPC_OK;
}

#define H2D(dst, src, n) HIPCHK(hipMemcpyAsync(dst, src, sizeof(*(dst)) * (size_t)(n), hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, n) do { if (dst) HIPCHK(hipMemcpyAsync(dst, src, sizeof(*(src)) * (size_t)(n), hipMemcpyDeviceToHost, c->stream)); } while (0)

// iteration counts of the batch just solved into the work buffers: copied to the caller (if asked) and added to lmpc_stats.ipm_iters
// (host-buffer entry points only: the device-resident path leaves the counts in the caller's device buffer)
static int fetch_iters(lmpc_ctx *c, int B, int *iters) {
    std::vector<int> tmp; int *dst = iters;
    if (!dst) { tmp.resize((size_t)B); dst = tmp.data(); }
    HIPCHK(hipMemcpyAsync(dst, c->w_iters, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int b = 0; b < B; b++) c->stats.ipm_iters += dst[b];
    return LMPC_OK;
}

int lmpc_regress_batch(lmpc_ctx *c, int B, const double *xLin, int xLinRowStride, const double *uLin, double *A, double *Bm, double *C, int *status) {
    ARGCHK(c && xLin && uLin && A && Bm && C && B >= 1 && B <= c->cfg.max_batch);
    const int N = c->cfg.N;
    ARGCHK(xLinRowStride == N * 6 || xLinRowStride == (N + 1) * 6);
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_xLin, xLin, (size_t)B * xLinRowStride); H2D(c->w_uLin, uLin, (size_t)B * N * 2);
    int rc = launch_regress(c, B, c->w_xLin, xLinRowStride, c->w_uLin, c->w_A, c->w_B, c->w_C, c->w_rstatus); if (rc) return rc;
    D2H(A, c->w_A, (size_t)B * N * 36); D2H(Bm, c->w_B, (size_t)B * N * 12); D2H(C, c->w_C, (size_t)B * N * 6); D2H(status, c->w_rstatus, (size_t)B * N);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

// hasPred without xPredPrev: the selection's Q-function shift (:502-512) would read predictions nobody supplied -- stale rows of the work buffer through the host entry
// points, a null pointer on the device through lmpc_step_batch_dev.  Host arrays are scanned (an all-zero hasPred is what callers pass on a first step); device arrays cannot be.
static bool pred_args_ok(int B, const double *xPredPrev, const int *hasPred) {
    if (!hasPred || xPredPrev) return true;
    for (int i = 0; i < B; i++) if (hasPred[i]) return false;
    return true;
}
int lmpc_regress_points(lmpc_ctx *c, int n, const double *x, const double *u, double *A, double *Bm, double *C, int *status) {
    // PredictiveModel.regressionAndLinearization (PredictiveModel.py:48-197) for n independent linearisation points (x (n x 6), u (n x 2)): the reference's own call
    // shape -- one point per call -- without a horizon around it.  The regression kernel runs with a parameter block whose horizon is 1: one query per work-group.
    ARGCHK(c && x && u && A && Bm && C && n >= 1 && (long long)n <= (long long)c->cfg.max_batch * c->cfg.N);
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_xLin, x, (size_t)n * 6); H2D(c->w_uLin, u, (size_t)n * 2);
    int rc = refresh_params(c, true, false); if (rc) return rc;
    {
        const lmpc_dev_params keep = c->dp;
        c->dp.N = 1;                                                     // (launch_k1 passes c->dp by value)
        ev_begin(c, 0);
        launch_k1(c, n, n, 1, c->w_xLin, 6, c->w_uLin, c->w_A, c->w_B, c->w_C, c->w_rstatus);
        ev_end(c);
        c->dp = keep;
    }
    HIPCHK(hipGetLastError());
    c->stats.n_regress++;
    D2H(A, c->w_A, (size_t)n * 36); D2H(Bm, c->w_B, (size_t)n * 12); D2H(C, c->w_C, (size_t)n * 6); D2H(status, c->w_rstatus, (size_t)n);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_select_batch(lmpc_ctx *c, int B, const double *x0, const double *zt, const double *xPredPrev, const int *hasPred, const int *timeStep,
                      double *ssSel, double *qSel, double *succ, double *succU, double *ztUsed, int *selStart, int *status) {
    ARGCHK(c && x0 && zt && B >= 1 && B <= c->cfg.max_batch && c->cfg.numSS_it > 0);
    ARGCHK(pred_args_ok(B, xPredPrev, hasPred));
    const int N = c->cfg.N, S = c->cfg.numSS_points;
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_x0, x0, (size_t)B * 6); H2D(c->w_zt, zt, (size_t)B * 6);
    if (xPredPrev) H2D(c->w_xPP, xPredPrev, (size_t)B * (N + 1) * 6);
    if (hasPred) H2D(c->w_hasPred, hasPred, B); else HIPCHK(hipMemsetAsync(c->w_hasPred, 0, sizeof(int) * B, c->stream));
    if (timeStep) H2D(c->w_tstep, timeStep, B); else HIPCHK(hipMemsetAsync(c->w_tstep, 0, sizeof(int) * B, c->stream));
    lmpc_solve_io io; memset(&io, 0, sizeof(io));
    io.mode = 1; io.x0 = c->w_x0; io.zt = c->w_zt; io.xPredPrev = c->w_xPP; io.hasPred = c->w_hasPred; io.timeStep = c->w_tstep;
    io.ssSelOut = c->w_ssSel; io.qSelOut = c->w_qSel; io.succOut = c->w_succ; io.succUOut = c->w_succU; io.ztUsed = c->w_ztUsed; io.status = c->w_status; io.iters = c->w_iters;
    io.selStartOut = c->w_selStart;
    int rc = launch_solve(c, B, io); if (rc) return rc;
    D2H(selStart, c->w_selStart, (size_t)B * c->cfg.numSS_it);
    D2H(ssSel, c->w_ssSel, (size_t)B * S * 6); D2H(qSel, c->w_qSel, (size_t)B * S); D2H(succ, c->w_succ, (size_t)B * S * 6); D2H(succU, c->w_succU, (size_t)B * S * 2);
    D2H(ztUsed, c->w_ztUsed, (size_t)B * 6); D2H(status, c->w_status, B);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_qp_solve_batch(lmpc_ctx *c, int B, const double *A, const double *Bm, const double *C, const double *x0, const double *uOld,
                        const double *ssSel, const double *qSel, double *xPred, double *uPred, double *slack, double *lambda, double *sTerm,
                        double *mu, int *status, int *iters, double *resid) {
    ARGCHK(c && A && Bm && C && x0 && uOld && xPred && uPred && B >= 1 && B <= c->cfg.max_batch);
    const int N = c->cfg.N, S = c->cfg.numSS_it > 0 ? c->cfg.numSS_points : 0, M = 8 * N + S;
    if (S > 0) ARGCHK(ssSel && qSel);
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_A, A, (size_t)B * N * 36); H2D(c->w_B, Bm, (size_t)B * N * 12); H2D(c->w_C, C, (size_t)B * N * 6);
    H2D(c->w_x0, x0, (size_t)B * 6); H2D(c->w_uOld, uOld, (size_t)B * 2);
    if (S > 0) { H2D(c->w_ssSel, ssSel, (size_t)B * S * 6); H2D(c->w_qSel, qSel, (size_t)B * S); }
    lmpc_solve_io io; memset(&io, 0, sizeof(io));
    io.mode = 2; io.A = c->w_A; io.Bm = c->w_B; io.C = c->w_C; io.x0 = c->w_x0; io.uOld = c->w_uOld; io.ssSelIn = c->w_ssSel; io.qSelIn = c->w_qSel;
    io.xPred = c->w_xPred; io.uPred = c->w_uPred; io.slack = c->w_slack; io.lambda = c->w_lam; io.sTerm = c->w_sT; io.mu = c->w_mu; io.resid = c->w_resid;
    io.status = c->w_status; io.iters = c->w_iters;
    int rc = launch_solve(c, B, io); if (rc) return rc;
    rc = resolve_retries(c); if (rc) return rc;          // drains the stream; a flagged problem gets its retry pass before anything is copied out
    D2H(xPred, c->w_xPred, (size_t)B * (N + 1) * 6); D2H(uPred, c->w_uPred, (size_t)B * N * 2); D2H(slack, c->w_slack, (size_t)B * N * 2);
    if (S > 0) { D2H(lambda, c->w_lam, (size_t)B * S); D2H(sTerm, c->w_sT, (size_t)B * 6); }
    D2H(mu, c->w_mu, (size_t)B * M); D2H(status, c->w_status, B); D2H(resid, c->w_resid, (size_t)B * 3);
    return fetch_iters(c, B, iters);                     // synchronises the stream
}

int lmpc_step_batch_dev(lmpc_ctx *c, int B, const lmpc_step_dev_args *a) {
    ARGCHK(c && a && B >= 1 && a->x0 && a->xLin && a->uLin && a->uOld && a->xPred && a->uPred && a->status && a->iters);
    const int N = c->cfg.N; const bool term = c->cfg.numSS_it > 0;
    if (term) ARGCHK(a->zt != nullptr);
    ARGCHK(!a->hasPred || a->xPredPrev);                 // (device arrays: a set hasPred[b] would dereference the missing predictions inside the kernel)
    HIPCHK(hipSetDevice(c->cfg.device));
    ARGCHK(B <= c->cfg.max_batch);                      // the work buffers (per-point regression status, A, B, C hand-over) are sized for max_batch
    lmpc_solve_io io; memset(&io, 0, sizeof(io));
    // Regression kernel, then the solve kernel; A, Bm, C are optional outputs (the hand-over then uses the context's work buffers).
    // With LMPC_FUSE=1 batches that run one wave per QP take the fused step instead: every wave runs the regression of its own QP in
    // front of the solve and [A_k | B_k], C_k never leave LDS (see lmpc_create for the measurement that keeps it off by default).
    const bool fused = c->fuse_k1 && lmpc_solver_waves(c, B) == 1;
    double *dA = a->A ? a->A : c->w_A, *dB = a->Bm ? a->Bm : c->w_B, *dC = a->C ? a->C : c->w_C;
    // A, Bm, C not given: the hand-over goes through the context's own buffers, which this launch's regression overwrites -- a launch still
    // pending its retry pass that used them is resolved first (a stream drain).  Callers that queue launches back to back pass their own
    // A / Bm / C (racinglmpc_amd._capi.Context.step_dev_buffers does) and keep every buffer of a launch untouched until lmpc_dev_sync.
    if (dA == c->w_A || dB == c->w_B || dC == c->w_C) for (const auto &pe : c->pending) if (pe.shared_abc) { RESOLVE_PENDING(); break; }
    if (fused) { io.mode = 4; io.xLin = a->xLin; io.uLin = a->uLin; io.Aout = a->A; io.Bout = a->Bm; io.Cout = a->C; if (int rc = refresh_params(c, true, false)) return rc; }
    else { int rc = launch_regress(c, B, a->xLin, (N + 1) * 6, a->uLin, dA, dB, dC, c->w_rstatus); if (rc) return rc; io.rstatus = c->w_rstatus; }
    int rc = 0;
    io.mode |= term ? 3 : 2; io.A = dA; io.Bm = dB; io.C = dC; io.x0 = a->x0; io.uOld = a->uOld;
    io.zt = a->zt; io.xPredPrev = a->xPredPrev; io.hasPred = a->hasPred; io.timeStep = a->timeStep;
    io.xPred = a->xPred; io.uPred = a->uPred; io.slack = a->slack; io.lambda = a->lambda; io.sTerm = a->sTerm; io.mu = a->mu;
    io.ztNext = a->ztNext; io.ztuNext = a->ztuNext; io.ssSelOut = a->ssSel; io.qSelOut = a->qSel; io.resid = a->resid; io.status = a->status; io.iters = a->iters;
    // (io.rstatus / the fused regression: a singular regression or an off-track linearisation point marks status[b]; the reference raises there)
    rc = launch_solve(c, B, io);
    return rc;
}

int lmpc_step_batch(lmpc_ctx *c, int B, const double *x0, const double *xLin, const double *uLin, const double *uOld, const double *zt,
                    const double *xPredPrev, const int *hasPred, const int *timeStep, double *xPred, double *uPred, double *slack, double *lambda,
                    double *sTerm, double *ztNext, double *ztuNext, double *ssSel, double *qSel, double *mu, double *Aout, double *Bout, double *Cout,
                    int *status, int *iters, double *resid) {
    ARGCHK(c && x0 && xLin && uLin && uOld && xPred && uPred && B >= 1 && B <= c->cfg.max_batch);
    ARGCHK(pred_args_ok(B, xPredPrev, hasPred));
    const int N = c->cfg.N; const bool term = c->cfg.numSS_it > 0; const int S = term ? c->cfg.numSS_points : 0;
    if (term) ARGCHK(zt != nullptr);
    HIPCHK(hipSetDevice(c->cfg.device));
    const bool one_copy = c->h_in && B == c->cfg.max_batch;          // small contexts (the drop-in classes: max_batch = 1): one copy each way
    const auto tr0 = std::chrono::steady_clock::now();
    if (one_copy) {
#define STAGE(wptr, src, n) memcpy(c->h_in + ((char *)c->wptr - c->slab_in), src, sizeof(*c->wptr) * (size_t)(n))
        STAGE(w_x0, x0, (size_t)B * 6); STAGE(w_xLin, xLin, (size_t)B * (N + 1) * 6); STAGE(w_uLin, uLin, (size_t)B * N * 2); STAGE(w_uOld, uOld, (size_t)B * 2);
        memset(c->h_in + ((char *)c->w_hasPred - c->slab_in), 0, sizeof(int) * B); memset(c->h_in + ((char *)c->w_tstep - c->slab_in), 0, sizeof(int) * B);
        if (term) {
            STAGE(w_zt, zt, (size_t)B * 6);
            if (xPredPrev) STAGE(w_xPP, xPredPrev, (size_t)B * (N + 1) * 6);
            if (hasPred && xPredPrev) STAGE(w_hasPred, hasPred, B);
            if (timeStep) STAGE(w_tstep, timeStep, B);
        }
#undef STAGE
        if (!c->dm_in) HIPCHK(hipMemcpyAsync(c->slab_in, c->h_in, c->slab_in_bytes, hipMemcpyHostToDevice, c->stream));
    } else {
    H2D(c->w_x0, x0, (size_t)B * 6); H2D(c->w_xLin, xLin, (size_t)B * (N + 1) * 6); H2D(c->w_uLin, uLin, (size_t)B * N * 2); H2D(c->w_uOld, uOld, (size_t)B * 2);
    if (term) {
        H2D(c->w_zt, zt, (size_t)B * 6);
        if (xPredPrev) H2D(c->w_xPP, xPredPrev, (size_t)B * (N + 1) * 6);
        if (hasPred && xPredPrev) H2D(c->w_hasPred, hasPred, B); else HIPCHK(hipMemsetAsync(c->w_hasPred, 0, sizeof(int) * B, c->stream));
        if (timeStep) H2D(c->w_tstep, timeStep, B); else HIPCHK(hipMemsetAsync(c->w_tstep, 0, sizeof(int) * B, c->stream));
    }
    }
    lmpc_step_dev_args a; memset(&a, 0, sizeof(a));
    a.x0 = c->w_x0; a.xLin = c->w_xLin; a.uLin = c->w_uLin; a.uOld = c->w_uOld; a.zt = c->w_zt; a.xPredPrev = c->w_xPP; a.hasPred = c->w_hasPred; a.timeStep = c->w_tstep;
    a.xPred = c->w_xPred; a.uPred = c->w_uPred; a.slack = c->w_slack; a.lambda = c->w_lam; a.sTerm = c->w_sT; a.ztNext = c->w_ztN; a.ztuNext = c->w_ztuN;
    a.ssSel = c->w_ssSel; a.qSel = c->w_qSel; a.A = c->w_A; a.Bm = c->w_B; a.C = c->w_C; a.mu = c->w_mu; a.resid = c->w_resid; a.status = c->w_status; a.iters = c->w_iters;
    const bool zero_copy = one_copy && c->dm_in && c->dm_out;
    if (zero_copy) {
        // Small contexts (the drop-in classes: one QP per call): the kernels read the inputs from and write the outputs to HOST-MAPPED memory --
        // no hipMemcpyAsync in either direction and ONE drain of the stream per call (two copies and two drains were ~25 us of a 0.24 ms LMPC.solve).
        // A few KB cross the host link inside the kernels instead (every array is read once at a kernel's start, loads issued together).
#define MAP_IN(f, w) a.f = (decltype(a.f))(c->dm_in + ((char *)c->w - c->slab_in))
#define MAP_OUT(f, w) a.f = (decltype(a.f))(c->dm_out + ((char *)c->w - c->slab_out))
        MAP_IN(x0, w_x0); MAP_IN(xLin, w_xLin); MAP_IN(uLin, w_uLin); MAP_IN(uOld, w_uOld); MAP_IN(zt, w_zt); MAP_IN(xPredPrev, w_xPP); MAP_IN(hasPred, w_hasPred); MAP_IN(timeStep, w_tstep);
        MAP_OUT(xPred, w_xPred); MAP_OUT(uPred, w_uPred); MAP_OUT(slack, w_slack); MAP_OUT(lambda, w_lam); MAP_OUT(sTerm, w_sT); MAP_OUT(ztNext, w_ztN); MAP_OUT(ztuNext, w_ztuN);
        MAP_OUT(ssSel, w_ssSel); MAP_OUT(qSel, w_qSel); MAP_OUT(mu, w_mu); MAP_OUT(resid, w_resid); MAP_OUT(status, w_status); MAP_OUT(iters, w_iters);
        if (Aout || Bout || Cout) { MAP_OUT(A, w_A); MAP_OUT(Bm, w_B); MAP_OUT(C, w_C); }      // (asked for: the regression writes them to the host, the solve reads them back at its start)
#undef MAP_IN
#undef MAP_OUT
    }
    const auto tr1 = std::chrono::steady_clock::now();
    int rc = lmpc_step_batch_dev(c, B, &a); if (rc) return rc;
    const auto tr2 = std::chrono::steady_clock::now();
    rc = resolve_retries(c); if (rc) return rc;          // drains the stream; a flagged problem gets its retry pass before anything is copied out
    const auto tr3 = std::chrono::steady_clock::now();
    if (one_copy) {
        if (!zero_copy) {
            // everything up to (not including) the selection-only buffers w_succ ..: one copy into the pinned mirror, then plain memcpys
            const size_t nbytes = (size_t)((char *)c->w_succ - c->slab_out);
            HIPCHK(hipMemcpyAsync(c->h_out, c->slab_out, nbytes, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
#define UNSTAGE(dst, wptr, n) do { if (dst) memcpy(dst, c->h_out + ((char *)c->wptr - c->slab_out), sizeof(*c->wptr) * (size_t)(n)); } while (0)
        UNSTAGE(xPred, w_xPred, (size_t)B * (N + 1) * 6); UNSTAGE(uPred, w_uPred, (size_t)B * N * 2); UNSTAGE(slack, w_slack, (size_t)B * N * 2);
        if (term) { UNSTAGE(lambda, w_lam, (size_t)B * S); UNSTAGE(sTerm, w_sT, (size_t)B * 6); UNSTAGE(ssSel, w_ssSel, (size_t)B * S * 6); UNSTAGE(qSel, w_qSel, (size_t)B * S); }
        UNSTAGE(mu, w_mu, (size_t)B * (8 * N + S)); UNSTAGE(ztNext, w_ztN, (size_t)B * 6); UNSTAGE(ztuNext, w_ztuN, (size_t)B * 2);
        UNSTAGE(Aout, w_A, (size_t)B * N * 36); UNSTAGE(Bout, w_B, (size_t)B * N * 12); UNSTAGE(Cout, w_C, (size_t)B * N * 6);
        UNSTAGE(status, w_status, B); UNSTAGE(resid, w_resid, (size_t)B * 3); UNSTAGE(iters, w_iters, B);
#undef UNSTAGE
        const int *it_ = (const int *)(c->h_out + ((char *)c->w_iters - c->slab_out));
        for (int b = 0; b < B; b++) c->stats.ipm_iters += it_[b];
        const auto tr4 = std::chrono::steady_clock::now();
        c->tr_s[0] += std::chrono::duration<double>(tr1 - tr0).count(); c->tr_s[1] += std::chrono::duration<double>(tr2 - tr1).count();
        c->tr_s[2] += std::chrono::duration<double>(tr3 - tr2).count(); c->tr_s[3] += std::chrono::duration<double>(tr4 - tr3).count(); c->tr_n++;
        return LMPC_OK;
    }
    D2H(xPred, c->w_xPred, (size_t)B * (N + 1) * 6); D2H(uPred, c->w_uPred, (size_t)B * N * 2); D2H(slack, c->w_slack, (size_t)B * N * 2);
    if (term) { D2H(lambda, c->w_lam, (size_t)B * S); D2H(sTerm, c->w_sT, (size_t)B * 6); D2H(ssSel, c->w_ssSel, (size_t)B * S * 6); D2H(qSel, c->w_qSel, (size_t)B * S); }
    D2H(mu, c->w_mu, (size_t)B * (8 * N + S));
    D2H(ztNext, c->w_ztN, (size_t)B * 6); D2H(ztuNext, c->w_ztuN, (size_t)B * 2);
    D2H(Aout, c->w_A, (size_t)B * N * 36); D2H(Bout, c->w_B, (size_t)B * N * 12); D2H(Cout, c->w_C, (size_t)B * N * 6);
    D2H(status, c->w_status, B); D2H(resid, c->w_resid, (size_t)B * 3);
    rc = fetch_iters(c, B, iters); if (rc) return rc;    // synchronises the stream
    return LMPC_OK;
}

int lmpc_qp_dims(lmpc_ctx *c, int *nz, int *m_ineq, int *m_eq) {
    ARGCHK(c);
    const int N = c->cfg.N, S = c->cfg.numSS_it > 0 ? c->cfg.numSS_points : 0;
    const int ns = c->cfg.slacks ? 2 * N : 0;
    if (nz) *nz = 6 * (N + 1) + 2 * N + ns + (S > 0 ? S + 6 : 0);
    if (m_ineq) *m_ineq = 6 * N + ns + S;
    if (m_eq) *m_eq = 6 * (N + 1) + (S > 0 ? 7 : 0);
    return LMPC_OK;
}

int lmpc_assemble_batch(lmpc_ctx *c, int B, const double *A, const double *Bm, const double *C, const double *x0, const double *uOld,
                        const double *ssSel, const double *qSel, double *Pdense, double *q, double *Adense, double *l, double *u) {
    ARGCHK(c && A && Bm && C && x0 && uOld && Pdense && q && Adense && l && u && B >= 1 && B <= c->cfg.max_batch);
    int nz, mi, me; lmpc_qp_dims(c, &nz, &mi, &me); const int mm = mi + me;
    const int N = c->cfg.N, S = c->cfg.numSS_it > 0 ? c->cfg.numSS_points : 0;
    if (S > 0) ARGCHK(ssSel && qSel);
    HIPCHK(hipSetDevice(c->cfg.device));
    double *dP, *dq, *dA, *dl, *du;
    HIPCHK(hipMalloc(&dP, sizeof(double) * (size_t)B * nz * nz)); HIPCHK(hipMalloc(&dq, sizeof(double) * (size_t)B * nz));
    HIPCHK(hipMalloc(&dA, sizeof(double) * (size_t)B * mm * nz)); HIPCHK(hipMalloc(&dl, sizeof(double) * (size_t)B * mm)); HIPCHK(hipMalloc(&du, sizeof(double) * (size_t)B * mm));
    H2D(c->w_A, A, (size_t)B * N * 36); H2D(c->w_B, Bm, (size_t)B * N * 12); H2D(c->w_C, C, (size_t)B * N * 6);
    H2D(c->w_x0, x0, (size_t)B * 6); H2D(c->w_uOld, uOld, (size_t)B * 2);
    if (S > 0) { H2D(c->w_ssSel, ssSel, (size_t)B * S * 6); H2D(c->w_qSel, qSel, (size_t)B * S); }
    hipLaunchKernelGGL(lmpc_assemble_kernel, dim3(B), dim3(256), 0, c->stream, c->dp, B, c->w_A, c->w_B, c->w_C, c->w_x0, c->w_uOld, c->w_ssSel, c->w_qSel, dP, dq, dA, dl, du);
    HIPCHK(hipGetLastError());
    D2H(Pdense, dP, (size_t)B * nz * nz); D2H(q, dq, (size_t)B * nz); D2H(Adense, dA, (size_t)B * mm * nz); D2H(l, dl, (size_t)B * mm); D2H(u, du, (size_t)B * mm);
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(dP); hipFree(dq); hipFree(dA); hipFree(dl); hipFree(du);
    return LMPC_OK;
}

// ---------------------------------------------------------------------------------------------- device buffers
int lmpc_dev_alloc(lmpc_ctx *c, long long bytes, void **dptr) { ARGCHK(c && dptr && bytes > 0); HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipMalloc(dptr, (size_t)bytes)); return LMPC_OK; }
int lmpc_dev_free(lmpc_ctx *c, void *dptr) { ARGCHK(c); HIPCHK(hipSetDevice(c->cfg.device)); { int rc = resolve_retries(c); if (rc) return rc; } HIPCHK(hipFree(dptr)); return LMPC_OK; }
int lmpc_dev_upload(lmpc_ctx *c, void *dptr, const void *host, long long bytes) { ARGCHK(c && dptr && host); HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipMemcpyAsync(dptr, host, (size_t)bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return LMPC_OK; }
int lmpc_dev_download(lmpc_ctx *c, void *host, const void *dptr, long long bytes) { ARGCHK(c && dptr && host); HIPCHK(hipSetDevice(c->cfg.device)); { int rc = resolve_retries(c); if (rc) return rc; } HIPCHK(hipMemcpyAsync(host, dptr, (size_t)bytes, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); return LMPC_OK; }
int lmpc_dev_sync(lmpc_ctx *c) { ARGCHK(c); HIPCHK(hipSetDevice(c->cfg.device)); { int rc = resolve_retries(c); if (rc) return rc; } HIPCHK(hipStreamSynchronize(c->stream)); return LMPC_OK; }

// developer trace: host-side seconds of the one-QP path of lmpc_step_batch since the context was made -- out[0..3] = staging the inputs, launching the
// two kernels, waiting for the stream, reading the outputs back; out[4] = calls (tools/dropin_time.py)
int lmpc_debug_step_trace(lmpc_ctx *c, double *out5) { ARGCHK(c && out5); for (int i = 0; i < 4; i++) out5[i] = c->tr_s[i]; out5[4] = (double)c->tr_n; return LMPC_OK; }

// timing build only: one solve of problem 0 of a host batch with cycle stamps (tools/phase_timing.py)
int lmpc_debug_timing(lmpc_ctx *c, const double *A, const double *Bm, const double *C, const double *x0, const double *uOld,
                      const double *ssSel, const double *qSel, long long *tbuf_host, int nt) {
    ARGCHK(c && tbuf_host && nt >= 2);
    const int N = c->cfg.N, S = c->cfg.numSS_it > 0 ? c->cfg.numSS_points : 0;
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_A, A, (size_t)N * 36); H2D(c->w_B, Bm, (size_t)N * 12); H2D(c->w_C, C, (size_t)N * 6); H2D(c->w_x0, x0, 6); H2D(c->w_uOld, uOld, 2);
    if (S > 0) { H2D(c->w_ssSel, ssSel, (size_t)S * 6); H2D(c->w_qSel, qSel, S); }
    long long *dt; HIPCHK(hipMalloc(&dt, sizeof(long long) * nt)); HIPCHK(hipMemsetAsync(dt, 0, sizeof(long long) * nt, c->stream));
    lmpc_solve_io io; memset(&io, 0, sizeof(io));
    io.mode = 2; io.A = c->w_A; io.Bm = c->w_B; io.C = c->w_C; io.x0 = c->w_x0; io.uOld = c->w_uOld; io.ssSelIn = c->w_ssSel; io.qSelIn = c->w_qSel;
    io.xPred = c->w_xPred; io.uPred = c->w_uPred; io.slack = c->w_slack; io.lambda = c->w_lam; io.sTerm = c->w_sT; io.mu = c->w_mu; io.resid = c->w_resid;
    io.status = c->w_status; io.iters = c->w_iters; io.tbuf = dt; io.abPack = c->ab_pack;      // (long horizons: the route a large batch takes, [A_k | B_k] in global memory; LMPC_NO_ABG=1 for the other)
    int rc = launch_solve(c, 1, io); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(tbuf_host, dt, sizeof(long long) * nt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream)); hipFree(dt);
    return LMPC_OK;
}

#ifdef LMPC_TIMING
// (developer build only) phase stamps of work-group 0 of one regression launch: tbuf_host[0..7] = cycle counter at K1STAMP(id)
int lmpc_debug_k1_timing(lmpc_ctx *c, int B, const double *xLin, const double *uLin, long long *tbuf_host) {
    ARGCHK(c && xLin && uLin && tbuf_host && B >= 1 && B <= c->cfg.max_batch);
    const int N = c->cfg.N;
    HIPCHK(hipSetDevice(c->cfg.device));
    H2D(c->w_xLin, xLin, (size_t)B * (N + 1) * 6); H2D(c->w_uLin, uLin, (size_t)B * N * 2);
    long long *dt; HIPCHK(hipMalloc(&dt, sizeof(long long) * 24)); HIPCHK(hipMemsetAsync(dt, 0, sizeof(long long) * 24, c->stream));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_k1_tbuf), &dt, sizeof(dt)));
    for (int rep = 0; rep < 2; rep++) { int rc = launch_regress(c, B, c->w_xLin, (N + 1) * 6, c->w_uLin, c->w_A, c->w_B, c->w_C, c->w_rstatus); if (rc) return rc; }
    HIPCHK(hipMemcpyAsync(tbuf_host, dt, sizeof(long long) * 24, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    long long *nul = nullptr; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_k1_tbuf), &nul, sizeof(nul))); hipFree(dt);
    return LMPC_OK;
}
#endif

// ---------------------------------------------------------------------------------------------- plant / rollouts
}  // extern "C"
// scratch of the small host-buffer entry points below: one device allocation, grown on demand and kept (no hipMalloc / hipFree per call)
static int pooled_scratch(lmpc_ctx *c, size_t bytes, void **out) {
    if (bytes > c->scr_bytes) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->scr_dev) { (void)g_free(c->scr_dev); c->scr_dev = nullptr; c->scr_bytes = 0; }
        const size_t want = std::max<size_t>((bytes + 4095) & ~(size_t)4095, (size_t)64 * 1024);
        HIPCHK(g_malloc(&c->scr_dev, want)); c->scr_bytes = want;
    }
    *out = c->scr_dev;
    return LMPC_OK;
}
extern "C" {
int lmpc_plant_step_batch(lmpc_ctx *c, int B, const double *x, const double *xg, const double *u, const double *noise, double *xn, double *xgn, int *status) {
    ARGCHK(c && x && xg && u && noise && xn && xgn && B >= 1);
    HIPCHK(hipSetDevice(c->cfg.device));
    double *d; { void *q; const int rc = pooled_scratch(c, sizeof(double) * (size_t)B * 29 + sizeof(int) * (size_t)B, &q); if (rc) return rc; d = (double *)q; }
    int *ds = (int *)(d + (size_t)B * 29);
    double *dx = d, *dg = d + (size_t)B * 6, *du = d + (size_t)B * 12, *dn = d + (size_t)B * 14, *dxn = d + (size_t)B * 17, *dgn = d + (size_t)B * 23;
    H2D(dx, x, (size_t)B * 6); H2D(dg, xg, (size_t)B * 6); H2D(du, u, (size_t)B * 2); H2D(dn, noise, (size_t)B * 3);
    hipLaunchKernelGGL(lmpc_plant_kernel, dim3((B + PLANT_CARS - 1) / PLANT_CARS), dim3(PLANT_NT), 0, c->stream, c->dp, B, dx, dg, du, dn, dxn, dgn, ds);
    HIPCHK(hipGetLastError());
    D2H(xn, dxn, (size_t)B * 6); D2H(xgn, dgn, (size_t)B * 6); D2H(status, ds, B);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_global_position_batch(lmpc_ctx *c, int n, const double *s, const double *ey, double *xy, int *status) {
    ARGCHK(c && s && ey && xy && status && n >= 1);
    HIPCHK(hipSetDevice(c->cfg.device));
    int rc = refresh_params(c, false, false); if (rc) return rc;
    double *d; { void *q; const int rc2 = pooled_scratch(c, sizeof(double) * (size_t)n * 4 + sizeof(int) * (size_t)n, &q); if (rc2) return rc2; d = (double *)q; }
    int *ds = (int *)(d + (size_t)n * 4);
    H2D(d, s, (size_t)n); H2D(d + n, ey, (size_t)n);
    hipLaunchKernelGGL(lmpc_global_position_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->dp, n, d, d + n, d + 2 * (size_t)n, ds);
    HIPCHK(hipGetLastError());
    D2H(xy, d + 2 * (size_t)n, (size_t)n * 2); D2H(status, ds, (size_t)n);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

struct lmpc_rollout_session {
    bool active;                                                // between lmpc_rollout_begin and lmpc_rollout_end; the buffers outlive the session (see lmpc_rollout_begin)
    int B, T_max, t;
    hipStream_t pstream; hipEvent_t e_solved, e_plant;          // plant integration runs beside the next regression (lmpc_rollout_plant_kernel)
    std::vector<void *> keep;
    double *d_x, *d_xg, *d_xLin, *d_uLin, *d_uOld, *d_zt, *d_xPP, *d_xPred, *d_uPred, *d_slack, *d_lam, *d_sT, *d_ztN, *d_ztuN, *d_A, *d_B, *d_C, *d_resid;
    double *d_logX, *d_logU, *d_logG, *d_noise, *d_finX, *d_finG;
    int *d_hasPred, *d_tstep, *d_done, *d_nDone, *d_stAcc, *d_status, *d_iters, *d_rst;
    double *d_ssSel, *d_qSel, *d_succ, *d_succU;                // only with lmpc_debug_rollout_capture on (else null: the step does not write them)
};

static void rollout_free(lmpc_ctx *c) {
    if (!c->ro) return;
    for (void *q : c->ro->keep) (void)g_free(q);
    if (c->ro->pstream) { (void)hipStreamDestroy(c->ro->pstream); (void)hipEventDestroy(c->ro->e_solved); (void)hipEventDestroy(c->ro->e_plant); }
    delete c->ro; c->ro = nullptr;
}

int lmpc_rollout_begin(lmpc_ctx *c, int B, int T_max, const double *x0, const double *xg0, const double *xLin0, const double *uLin0, const double *noise) {
    // B closed-loop LMPC laps, state resident on the device.  xLin0 / uLin0: per-rollout first linearisation trajectories
    // (B x (N+1) x 6, B x N x 2) -- LMPC.addTrajectory :431-433.  noise: T_max x B x 3 N(0,1) draws.
    ARGCHK(c && x0 && xg0 && xLin0 && uLin0 && noise && B >= 1 && T_max >= 1 && c->cfg.numSS_it > 0);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t N = c->cfg.N, S = c->cfg.numSS_points, Bz = B;
    // A generation loop begins a session of the same shape every lap: its ~35 device buffers (55 MB of logs at 1024 rollouts x 400 steps), the plant stream and
    // the two events are kept from one session to the next (round 5: allocating and freeing them was ~5 ms of every generation) and released by
    // lmpc_destroy or by a session of another shape.
    if (c->ro && !c->ro->active && c->ro->B == B && c->ro->T_max == T_max && (c->ro->d_ssSel != nullptr) == (c->dbg_capture != 0)) {
        lmpc_rollout_session *r = c->ro; r->t = 0; r->active = true;
        goto init_state;
    }
    rollout_free(c);
    {
    lmpc_rollout_session *r = new lmpc_rollout_session(); c->ro = r; r->B = B; r->T_max = T_max; r->t = 0; r->pstream = nullptr; r->active = true;
    HIPCHK(hipStreamCreate(&r->pstream)); HIPCHK(hipEventCreateWithFlags(&r->e_solved, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&r->e_plant, hipEventDisableTiming));
    bool ok = true;
    auto dalloc = [&](size_t bytes) -> void * { void *q = nullptr; if (g_malloc(&q, std::max<size_t>(bytes, 8)) != hipSuccess) { ok = false; return nullptr; } r->keep.push_back(q); return q; };
#define DA(type, name, n) r->name = (type *)dalloc(sizeof(type) * (size_t)(n));
    DA(double, d_x, Bz * 6) DA(double, d_xg, Bz * 6) DA(double, d_xLin, Bz * (N + 1) * 6) DA(double, d_uLin, Bz * N * 2) DA(double, d_uOld, Bz * 2)
    DA(double, d_zt, Bz * 6) DA(double, d_xPP, Bz * (N + 1) * 6) DA(int, d_hasPred, Bz) DA(int, d_tstep, Bz) DA(int, d_done, Bz) DA(int, d_nDone, 1) DA(int, d_stAcc, Bz)
    DA(double, d_xPred, Bz * (N + 1) * 6) DA(double, d_uPred, Bz * N * 2) DA(double, d_slack, Bz * N * 2) DA(double, d_lam, Bz * S) DA(double, d_sT, Bz * 6)
    DA(double, d_ztN, Bz * 6) DA(double, d_ztuN, Bz * 2) DA(double, d_A, Bz * N * 36) DA(double, d_B, Bz * N * 12) DA(double, d_C, Bz * N * 6)
    DA(double, d_resid, Bz * 3) DA(int, d_status, Bz) DA(int, d_iters, Bz) DA(int, d_rst, Bz * N) DA(double, d_finX, Bz * 6) DA(double, d_finG, Bz * 6)
    r->d_ssSel = nullptr; r->d_qSel = nullptr; r->d_succ = nullptr; r->d_succU = nullptr;
    if (c->dbg_capture) { DA(double, d_ssSel, Bz * S * 6) DA(double, d_qSel, Bz * S) DA(double, d_succ, Bz * S * 6) DA(double, d_succU, Bz * S * 2) }
    DA(double, d_logX, (size_t)T_max * Bz * 6) DA(double, d_logU, (size_t)T_max * Bz * 2) DA(double, d_logG, (size_t)T_max * Bz * 6) DA(double, d_noise, (size_t)T_max * Bz * 3)
#undef DA
    if (!ok) { rollout_free(c); return set_err(LMPC_E_HIP, "hipMalloc", "rollout buffers"); }
    }
init_state:
    lmpc_rollout_session *r = c->ro;
    std::vector<double> ztv((size_t)B * 6, 0.0);
    for (int b = 0; b < B; b++) ztv[(size_t)b * 6 + 4] = 10.0;                                   // LMPC.__init__ :330
    std::vector<int> neg((size_t)B, -1);
    H2D(r->d_x, x0, Bz * 6); H2D(r->d_xg, xg0, Bz * 6); H2D(r->d_xLin, xLin0, Bz * (N + 1) * 6); H2D(r->d_uLin, uLin0, Bz * N * 2); H2D(r->d_zt, ztv.data(), ztv.size());
    H2D(r->d_done, neg.data(), Bz); H2D(r->d_noise, noise, (size_t)T_max * Bz * 3);
    HIPCHK(hipMemsetAsync(r->d_uOld, 0, sizeof(double) * Bz * 2, c->stream)); HIPCHK(hipMemsetAsync(r->d_xPP, 0, sizeof(double) * Bz * (N + 1) * 6, c->stream));
    HIPCHK(hipMemsetAsync(r->d_hasPred, 0, sizeof(int) * Bz, c->stream)); HIPCHK(hipMemsetAsync(r->d_tstep, 0, sizeof(int) * Bz, c->stream));
    HIPCHK(hipMemsetAsync(r->d_nDone, 0, sizeof(int), c->stream)); HIPCHK(hipMemsetAsync(r->d_stAcc, 0, sizeof(int) * Bz, c->stream));
    HIPCHK(hipMemsetAsync(r->d_finX, 0, sizeof(double) * Bz * 6, c->stream)); HIPCHK(hipMemsetAsync(r->d_finG, 0, sizeof(double) * Bz * 6, c->stream));
    // a reused session is indistinguishable from a fresh one: what the previous lap left in the solver's outputs and status words goes too
    HIPCHK(hipMemsetAsync(r->d_xPred, 0, sizeof(double) * Bz * (N + 1) * 6, c->stream)); HIPCHK(hipMemsetAsync(r->d_uPred, 0, sizeof(double) * Bz * N * 2, c->stream));
    HIPCHK(hipMemsetAsync(r->d_lam, 0, sizeof(double) * Bz * S, c->stream)); HIPCHK(hipMemsetAsync(r->d_sT, 0, sizeof(double) * Bz * 6, c->stream));
    HIPCHK(hipMemsetAsync(r->d_ztN, 0, sizeof(double) * Bz * 6, c->stream)); HIPCHK(hipMemsetAsync(r->d_ztuN, 0, sizeof(double) * Bz * 2, c->stream));
    HIPCHK(hipMemsetAsync(r->d_status, 0, sizeof(int) * Bz, c->stream)); HIPCHK(hipMemsetAsync(r->d_iters, 0, sizeof(int) * Bz, c->stream));
    HIPCHK(hipMemsetAsync(r->d_rst, 0, sizeof(int) * Bz * N, c->stream)); HIPCHK(hipMemsetAsync(r->d_resid, 0, sizeof(double) * Bz * 3, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_rollout_run(lmpc_ctx *c, int max_steps, int *steps_total, int *n_done) {
    // advance every rollout by up to max_steps simulated steps (four launches per step on two streams, no host round trip except a
    // finished-lap poll every 8 steps); stops early once every rollout has crossed the finish line
    ARGCHK(c && c->ro && c->ro->active && max_steps >= 1);
    lmpc_rollout_session *r = c->ro;
    const int B = r->B; const size_t N = c->cfg.N;
    HIPCHK(hipSetDevice(c->cfg.device));
    lmpc_rollout_state st; st.x = r->d_x; st.xg = r->d_xg; st.xLin = r->d_xLin; st.uLin = r->d_uLin; st.uOld = r->d_uOld; st.zt = r->d_zt; st.xPP = r->d_xPP;
    st.hasPred = r->d_hasPred; st.timeStep = r->d_tstep; st.doneAt = r->d_done; st.xPred = r->d_xPred; st.uPred = r->d_uPred; st.ztNext = r->d_ztN; st.ztuNext = r->d_ztuN;
    st.status = r->d_status; st.logX = r->d_logX; st.logU = r->d_logU; st.logG = r->d_logG; st.noise = r->d_noise; st.nDone = r->d_nDone; st.statusAcc = r->d_stAcc;
    st.finX = r->d_finX; st.finG = r->d_finG;
    int nd = 0, rc = LMPC_OK;
    const int t_end = std::min(r->T_max, r->t + max_steps);
    while (r->t < t_end) {
        rc = refresh_params(c, true, true); if (rc) return rc;
        ev_begin(c, 0);
        { int qg, nblk; k1_grid(c, B, &qg, &nblk);
          launch_k1(c, nblk, B, qg, (const double *)r->d_xLin, (int)(N + 1) * 6, (const double *)r->d_uLin, r->d_A, r->d_B, r->d_C, r->d_rst); }
        ev_end(c); c->stats.n_regress++;
        lmpc_solve_io io; memset(&io, 0, sizeof(io));
        io.mode = 3; io.A = r->d_A; io.Bm = r->d_B; io.C = r->d_C; io.x0 = r->d_x; io.uOld = r->d_uOld; io.zt = r->d_zt; io.xPredPrev = r->d_xPP; io.hasPred = r->d_hasPred;
        io.timeStep = r->d_tstep; io.xPred = r->d_xPred; io.uPred = r->d_uPred; io.slack = r->d_slack; io.lambda = r->d_lam; io.sTerm = r->d_sT; io.ztNext = r->d_ztN;
        io.ztuNext = r->d_ztuN; io.resid = r->d_resid; io.status = r->d_status; io.iters = r->d_iters; io.rstatus = r->d_rst;
        io.ssSelOut = r->d_ssSel; io.qSelOut = r->d_qSel; io.succOut = r->d_succ; io.succUOut = r->d_succU;
        if (r->t > 0) HIPCHK(hipStreamWaitEvent(c->stream, r->e_plant, 0));          // the solve needs the plant's new state
        rc = launch_solve(c, B, io, true); if (rc) return rc;      // (the plant consumes uPred without a host round trip: unconditional retry pass)
        HIPCHK(hipEventRecord(r->e_solved, c->stream));
        HIPCHK(hipStreamWaitEvent(r->pstream, r->e_solved, 0));
        hipLaunchKernelGGL(lmpc_rollout_plant_kernel, dim3((B + PLANT_CARS - 1) / PLANT_CARS), dim3(PLANT_NT), 0, r->pstream, c->dp, B, r->t, st);
        HIPCHK(hipEventRecord(r->e_plant, r->pstream));
        hipLaunchKernelGGL(lmpc_rollout_shift_kernel, dim3((B * LMPC_SHIFT_TPR((int)N) + 255) / 256), dim3(256), 0, c->stream, c->dp, B, r->t, st);   // then the next step's regression
        HIPCHK(hipGetLastError());
        r->t++;
        if ((r->t & 7) == 0 || r->t == t_end) {
            HIPCHK(hipStreamSynchronize(r->pstream));
            HIPCHK(hipMemcpyAsync(&nd, r->d_nDone, sizeof(int), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream));
            if (nd >= B) break;
        }
    }
    if (steps_total) *steps_total = r->t;
    if (n_done) *n_done = nd;
    return LMPC_OK;
}

int lmpc_rollout_fetch(lmpc_ctx *c, int t0, int t1, double *X, double *U, double *G, int *doneAt, int *status, double *finalX, double *finalG) {
    // logs of steps [t0, t1): X/G (t1-t0) x B x 6, U (t1-t0) x B x 2; doneAt: steps until s > TrackLength (-1: not yet);
    // finalX / finalG: state right after the crossing step (the reference's xF before the TrackLength shift, SysModel.py:50)
    ARGCHK(c && c->ro && c->ro->active && t0 >= 0 && t1 >= t0 && t1 <= c->ro->t);
    lmpc_rollout_session *r = c->ro; const size_t Bz = r->B, n = (size_t)(t1 - t0);
    HIPCHK(hipSetDevice(c->cfg.device));
    if (n) { D2H(X, r->d_logX + (size_t)t0 * Bz * 6, n * Bz * 6); D2H(U, r->d_logU + (size_t)t0 * Bz * 2, n * Bz * 2); D2H(G, r->d_logG + (size_t)t0 * Bz * 6, n * Bz * 6); }
    D2H(doneAt, r->d_done, Bz); D2H(status, r->d_stAcc, Bz); D2H(finalX, r->d_finX, Bz * 6); D2H(finalG, r->d_finG, Bz * 6);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_debug_rollout_peek(lmpc_ctx *c, double *xLin, double *uLin, int *status, int *rstatus) {
    // developer entry point: the rollout session's current linearisation trajectories (B x (N+1) x 6, B x N x 2 -- what the NEXT step's regression will be asked) and the
    // status words of the step just taken (per rollout: B; per horizon point of its regression: B x N).  tools/robustness_sweep.py steps a generation one step at a
    // time with it to capture the inputs of a flagged regression.  Any pointer may be NULL.
    ARGCHK(c && c->ro && c->ro->active);
    lmpc_rollout_session *r = c->ro; const size_t Bz = r->B, N = c->cfg.N;
    HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(r->pstream)); HIPCHK(hipStreamSynchronize(c->stream));
    D2H(xLin, r->d_xLin, Bz * (N + 1) * 6); D2H(uLin, r->d_uLin, Bz * N * 2); D2H(status, r->d_status, Bz); D2H(rstatus, r->d_rst, Bz * N);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_debug_rollout_capture(lmpc_ctx *c, int on) {
    // developer entry point: rollout sessions begun after this call also write the selected safe-set points / Q-values of every step (B x S x 6, B x S) so that
    // lmpc_debug_rollout_qp can hand a closed-loop QP to the oracle.  Off (the default) the step does not write them.
    ARGCHK(c && !(c->ro && c->ro->active));
    c->dbg_capture = on ? 1 : 0;
    return LMPC_OK;
}

int lmpc_debug_rollout_qp(lmpc_ctx *c, int b0, int n, double *A, double *Bm, double *Cm, double *xPred, double *uPred, double *ssSel, double *qSel, double *succ, double *succU,
                          double *lambda, double *ztNext, double *ztuNext, int *iters, int *status) {
    // developer entry point: the QP the LAST simulated step solved for rollouts b0 .. b0 + n - 1 -- the regression's A, B, C (n x N x 36 / 12 / 6), the kernel's
    // answer (n x (N+1) x 6, n x N x 2, n x S; zt / zt_u of feasibleStateInput: n x 6, n x 2), iterations and status, and (capture on) the selected safe-set points, their
    // Q-values and successors (n x S x 6, n x S, n x S x 6, n x S x 2).  Its x0 / uOld are rows t - 1 / t - 2 of the session's logs (lmpc_rollout_fetch).  Meant to be
    // called between lmpc_rollout_run(ctx, 1, ..) calls; any pointer may be NULL.
    ARGCHK(c && c->ro && c->ro->active && c->ro->t >= 1 && b0 >= 0 && n >= 1 && b0 + n <= c->ro->B && ((!ssSel && !qSel && !succ && !succU) || c->ro->d_ssSel));
    lmpc_rollout_session *r = c->ro; const size_t N = c->cfg.N, S = c->cfg.numSS_points, o = b0, nz = n;
    HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(r->pstream)); HIPCHK(hipStreamSynchronize(c->stream));
    D2H(A, r->d_A + o * N * 36, nz * N * 36); D2H(Bm, r->d_B + o * N * 12, nz * N * 12); D2H(Cm, r->d_C + o * N * 6, nz * N * 6);
    D2H(xPred, r->d_xPred + o * (N + 1) * 6, nz * (N + 1) * 6); D2H(uPred, r->d_uPred + o * N * 2, nz * N * 2); D2H(lambda, r->d_lam + o * S, nz * S);
    D2H(ztNext, r->d_ztN + o * 6, nz * 6); D2H(ztuNext, r->d_ztuN + o * 2, nz * 2);
    D2H(ssSel, r->d_ssSel + o * S * 6, nz * S * 6); D2H(qSel, r->d_qSel + o * S, nz * S); D2H(succ, r->d_succ + o * S * 6, nz * S * 6); D2H(succU, r->d_succU + o * S * 2, nz * S * 2);
    D2H(iters, r->d_iters + o, nz); D2H(status, r->d_status + o, nz);
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

int lmpc_rollout_end(lmpc_ctx *c) {
    ARGCHK(c); HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(c->stream));
    if (c->ro) {
        if (c->ro->pstream) HIPCHK(hipStreamSynchronize(c->ro->pstream));
#ifdef LMPC_GUARD
        rollout_free(c);                                   // (guard builds check the zone behind every buffer when it is freed: keep doing that per session)
#else
        c->ro->active = false;
#endif
    }
    return LMPC_OK;
}

int lmpc_rollout_release(lmpc_ctx *c) {
    // lmpc_rollout_end keeps the session's device buffers (~35 allocations, 55 MB at 1024 rollouts x 400 steps), its stream and events for the next lap of the same
    // shape; this gives them back without destroying the context.  Not valid inside a session.
    ARGCHK(c && !(c->ro && c->ro->active));
    HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(c->stream));
    rollout_free(c);
    return LMPC_OK;
}

__global__ void lmpc_store_rows_kernel(double *base, int stride, int row0, int n, const double *rows /* n x 9: x (6, s already shifted) | u (2) | Qfun */) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * LMPC_COLS) { const int i = e / LMPC_COLS, col = e % LMPC_COLS; base[(size_t)col * stride + row0 + i] = rows[e]; }
}

int lmpc_ss_extend_lap(lmpc_ctx *c, int lap, const double *x, const double *u, int n) {
    // batched-mode form of LMPC.addPoint (:466-474) for ANY stored lap: append n points shifted by TrackLength in s,
    // Q-function continuing to count down.  (In the reference the points of lap j extend lap j-1 one by one.)  One upload, one launch.
    ARGCHK(c && lap >= 0 && lap < (int)c->s_len.size() && n >= 0 && (n == 0 || (x && u)));
    if (n == 0) return LMPC_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    RESOLVE_PENDING();
    if (c->s_len[lap] + n > c->cfg.max_lap_len) { const int rc = grow_stores(c, c->cfg.max_laps, c->s_len[lap] + n); if (rc) return rc; }
    std::vector<double> rows((size_t)n * LMPC_COLS);
    double q = c->s_qlast[lap];
    for (int i = 0; i < n; i++) {
        q -= 1.0;
        double *r = &rows[(size_t)i * LMPC_COLS];
        for (int j = 0; j < 6; j++) r[j] = x[(size_t)i * 6 + j];
        r[4] += c->cfg.trackLength; r[6] = u[(size_t)i * 2]; r[7] = u[(size_t)i * 2 + 1]; r[8] = q;
    }
    if (rows.size() * sizeof(double) > c->ext_rows_bytes) {       // staging buffer of the batched addPoint: grown once, kept
        if (c->ext_rows) (void)g_free(c->ext_rows);
        c->ext_rows = nullptr; c->ext_rows_bytes = 0;
        const size_t want = std::max<size_t>(rows.size() * sizeof(double), (size_t)64 * LMPC_COLS * sizeof(double));
        HIPCHK(g_malloc(&c->ext_rows, want)); c->ext_rows_bytes = want;
    }
    double *d_rows = c->ext_rows;
    hipError_t e = hipMemcpyAsync(d_rows, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        double *base = c->sstore + (size_t)lap * LMPC_COLS * c->cfg.max_lap_len;
        hipLaunchKernelGGL(lmpc_store_rows_kernel, dim3((n * LMPC_COLS + 255) / 256), dim3(256), 0, c->stream, base, c->cfg.max_lap_len, c->s_len[lap], n, (const double *)d_rows);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      // (the pageable host rows must outlive the copy)
    if (e != hipSuccess) return set_err(LMPC_E_HIP, "lmpc_ss_extend_lap", hipGetErrorString(e));
    c->s_len[lap] += n; c->s_qlast[lap] = q;
    return LMPC_OK;
}

int lmpc_ss_truncate_lap(lmpc_ctx *c, int lap, int T) {
    ARGCHK(c && lap >= 0 && lap < (int)c->s_len.size() && T >= c->s_laptime[lap] && T <= c->s_len[lap]);
    if (T == c->s_len[lap]) return LMPC_OK;
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    double q = 0.0;                                                  // Qfun of the new last row (addPoint counts on from it, :474)
    HIPCHK(hipMemcpy(&q, c->sstore + ((size_t)lap * LMPC_COLS + 8) * c->cfg.max_lap_len + (T - 1), sizeof(double), hipMemcpyDeviceToHost));
    c->s_len[lap] = T; c->s_qlast[lap] = q;
    return LMPC_OK;
}

int lmpc_lti_regression(int device, const double *x, const double *u, int T, double lamb, double *A, double *B, double *Error, int *status) {
    // Utilities.Regression (fnc/Utilities.py:5-28), called by main.py:74-77 before any controller exists: no context needed
    ARGCHK(x && u && A && B && Error && T >= 3);
    HIPCHK(hipSetDevice(device));
    double *d; HIPCHK(hipMalloc(&d, sizeof(double) * ((size_t)T * 8 + 60) + sizeof(int)));
    double *dx = d, *du = d + (size_t)T * 6, *dout = du + (size_t)T * 2; int *dst = (int *)(dout + 60);
    hipError_t e = hipMemcpy(dx, x, sizeof(double) * (size_t)T * 6, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(du, u, sizeof(double) * (size_t)T * 2, hipMemcpyHostToDevice);
    double hout[60]; int hst = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lmpc_lti_regress_kernel, dim3(1), dim3(LTI_NT), 0, 0, (const double *)dx, (const double *)du, T, lamb, dout, dst);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(hout, dout, sizeof(hout), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&hst, dst, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return set_err(LMPC_E_HIP, "lmpc_lti_regression", hipGetErrorString(e));
    memcpy(A, hout, sizeof(double) * 36); memcpy(B, hout + 36, sizeof(double) * 12); memcpy(Error, hout + 48, sizeof(double) * 12);
    if (status) *status = hst;
    return LMPC_OK;
}

int lmpc_selftest(lmpc_ctx *c) {
    // cross-lane primitives (DPP + v_permlane16/32_swap reductions) against closed-form values
    ARGCHK(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    double *d; HIPCHK(hipMalloc(&d, sizeof(double) * 192));
    hipLaunchKernelGGL(lmpc_selftest_kernel, dim3(1), dim3(WAVE), 0, c->stream, d);
    double hbuf[192]; HIPCHK(hipMemcpyAsync(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); hipFree(d);
    double s = 0, mx = -1e300, mn = 1e300;
    for (int l = 0; l < 64; l++) { const double v = 1.0 + 0.25 * l + ((l * 37) % 11) * 1e-3; s += v; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    for (int l = 0; l < 64; l++) {
        if (fabs(hbuf[l] - s) > 1e-9 * s || hbuf[64 + l] != mx || hbuf[128 + l] != mn) return set_err(LMPC_E_HIP, "wave reduction self test failed", "");
    }
    return LMPC_OK;
}

int lmpc_debug_set_trace(lmpc_ctx *c, double *dev_rows) {
    // developer builds (-DLMPC_TRACE): every later solve launch writes LMPC_TRACE_ROWS x 6 doubles per problem into dev_rows (caller-owned device memory, at least
    // max_batch x 48 x 6 doubles; NULL switches it off): per iteration (gap, r_d, r_e | sigma, alpha_p, alpha_d).  Ordinary builds carry no such code: LMPC_E_ARG.
    ARGCHK(c);
#ifdef LMPC_TRACE
    HIPCHK(hipSetDevice(c->cfg.device)); RESOLVE_PENDING(); HIPCHK(hipStreamSynchronize(c->stream));
    c->dbg_trace = dev_rows;
    return LMPC_OK;
#else
    (void)dev_rows;
    return set_err(LMPC_E_ARG, "lmpc_debug_set_trace", "this library was built without -DLMPC_TRACE (racinglmpc_amd.build.build_flavour)");
#endif
}

int lmpc_debug_exec_audit(lmpc_ctx *c, unsigned long long *out16, int reset) {
    // developer builds (-DLMPC_EXEC_AUDIT): out16[site] = calls of a cross-lane primitive that found an incomplete EXEC mask, out16[8 + site] = calls, since
    // the last reset (sites: lmpc_kernels.hip.h, EXEC_AUDIT); built-in variants only (a liblmpc_var_*.so carries counters of its own).  Ordinary builds: LMPC_E_ARG.
    ARGCHK(c && out16);
#ifdef LMPC_EXEC_AUDIT
    HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_exec_audit), sizeof(unsigned long long) * 2 * LMPC_AUDIT_SITES));
    if (reset) { unsigned long long z[2 * LMPC_AUDIT_SITES] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_exec_audit), z, sizeof(z))); }
    return LMPC_OK;
#else
    (void)reset;
    return set_err(LMPC_E_ARG, "lmpc_debug_exec_audit", "this library was built without -DLMPC_EXEC_AUDIT (racinglmpc_amd.build.build_flavour)");
#endif
}

int lmpc_solver_waves(lmpc_ctx *c, int B) {
    if (!c) return LMPC_E_ARG;
    if (B <= c->mw_max_batch) return 4;
    if (B <= c->mw2_max_batch) return 2;
    return 1;
}

int lmpc_set_profiling(lmpc_ctx *c, int every) { ARGCHK(c); c->profiling = every > 0 ? every : 0; return LMPC_OK; }
static int drain_events(lmpc_ctx *c) {
    HIPCHK(hipSetDevice(c->cfg.device)); HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &e : c->events) {
        float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        if (e.kind == 0) { c->stats.ms_regress += ms; c->stats.n_regress_timed++; } else { c->stats.ms_solve += ms; c->stats.n_solve_timed++; }
        hipEventDestroy(e.a); hipEventDestroy(e.b);
    }
    c->events.clear();
    return LMPC_OK;
}
int lmpc_get_stats(lmpc_ctx *c, lmpc_stats *out) { ARGCHK(c && out); int rc = drain_events(c); if (rc) return rc; *out = c->stats; return LMPC_OK; }
int lmpc_reset_stats(lmpc_ctx *c) { ARGCHK(c); int rc = drain_events(c); if (rc) return rc; memset(&c->stats, 0, sizeof(c->stats)); return LMPC_OK; }

}  // extern "C"

#include "lmpc_comm.hip.h"
