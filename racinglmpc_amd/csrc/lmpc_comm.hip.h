// racinglmpc_amd/csrc/lmpc_comm.hip.h -- multi-GPU exchange step of the LMPC path, RCCL over xGMI, behind the C ABI.
//
// One process per GPU; QPs / rollouts are independent given a read-only safe set, so the data path has no collective.
// The one exchange is per lap: every rank contributes its K fastest valid rollouts as fixed-stride records packed ON THE DEVICE from
// the rollout session's logs (no host staging), one ncclAllGather makes the union resident on every GPU, the (tiny) deterministic
// top-K and the addTrajectory inserts then run identically on every rank.  The unique id travels between the processes outside of
// this library (racinglmpc_amd/parallel.py: a TCP hand-off on the launcher's MASTER_ADDR); nothing here depends on PyTorch.
// Included by lmpc_capi.hip (needs lmpc_ctx).
#pragma once
#include <rccl/rccl.h>

#define NCCLCHK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return set_err(LMPC_E_HIP, #call, ncclGetErrorString(r_)); } while (0)

// record j of a rank: [T_max + 1][14] doubles, rows t < T: x_t (6) | u_t (2) | x_glob_t (6); row T_max: state (6) + global state (6) right
// after the finish line, then the rollout's index in the rank's shard and 0 -- the layout of racinglmpc_amd/parallel.pack_laps.
__global__ void lmpc_pack_laps_kernel(int B, int K, int T_max, const int *__restrict__ sel /*K rollout indices, -1 = empty*/, const int *__restrict__ len /*K*/,
                                      const double *__restrict__ logX, const double *__restrict__ logU, const double *__restrict__ logG,
                                      const double *__restrict__ finX, const double *__restrict__ finG, double *__restrict__ rec) {
    const int j = blockIdx.y, b = sel[j], T = len[j];
    double *out = rec + (size_t)j * (T_max + 1) * 14;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (T_max + 1) * 14; e += gridDim.x * blockDim.x) {
        const int t = e / 14, c = e % 14;
        double v = 0.0;
        if (b >= 0) {
            if (t < T) v = c < 6 ? logX[((size_t)t * B + b) * 6 + c] : (c < 8 ? logU[((size_t)t * B + b) * 2 + (c - 6)] : logG[((size_t)t * B + b) * 6 + (c - 8)]);
            else if (t == T_max) v = c < 6 ? finX[(size_t)b * 6 + c] : (c < 12 ? finG[(size_t)b * 6 + (c - 6)] : (c == 12 ? (double)b : 0.0));
        }
        out[e] = v;
    }
}

// Scratch of the communicator: one device allocation and one pinned host mirror, made by lmpc_comm_init (and on first use in a single process),
// grown only if a call needs more -- no hipMalloc / hipFree inside allreduce / allgather / the per-lap exchange (each costs tens of microseconds
// and a device-wide synchronisation, which would read as "scaling loss" at the end of a 5 ms timed region).
static int comm_scratch(lmpc_ctx *c, size_t bytes) {
    if (bytes <= c->comm_scr_bytes) return LMPC_OK;
    size_t want = std::max<size_t>(bytes, (size_t)64 * 1024);
    want = (want + 4095) & ~(size_t)4095;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->comm_scr) { (void)g_free(c->comm_scr); c->comm_scr = nullptr; }
    if (c->comm_scr_h) { (void)hipHostFree(c->comm_scr_h); c->comm_scr_h = nullptr; }
    c->comm_scr_bytes = 0;
    HIPCHK(g_malloc(&c->comm_scr, want)); HIPCHK(hipHostMalloc(&c->comm_scr_h, want));
    c->comm_scr_bytes = want;
    return LMPC_OK;
}

extern "C" {

int lmpc_comm_unique_id(unsigned char *id /*LMPC_COMM_ID_BYTES*/) {
    ARGCHK(id);
    static_assert(LMPC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u; NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return LMPC_OK;
}

int lmpc_comm_init(lmpc_ctx *c, const unsigned char *id, int rank, int world) {
    ARGCHK(c && id && world >= 1 && rank >= 0 && rank < world);
    if (c->comm) return set_err(LMPC_E_STATE, "lmpc_comm_init", "communicator already initialised");
    HIPCHK(hipSetDevice(c->cfg.device));
    ncclUniqueId u; memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm;
    NCCLCHK(ncclCommInitRank(&comm, world, u, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return comm_scratch(c, (size_t)64 * 1024);
}

int lmpc_comm_destroy(lmpc_ctx *c) {
    ARGCHK(c);
    if (c->comm) { (void)hipSetDevice(c->cfg.device); (void)hipStreamSynchronize(c->stream); (void)ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    c->comm_rank = 0; c->comm_world = 1;
    return LMPC_OK;
}

int lmpc_comm_info(lmpc_ctx *c, int *rank, int *world, int *is_rccl) {
    ARGCHK(c);
    if (rank) *rank = c->comm_rank; if (world) *world = c->comm_world > 0 ? c->comm_world : 1; if (is_rccl) *is_rccl = c->comm ? 1 : 0;
    return LMPC_OK;
}

// device buffers: recv = world x bytes.  Without a communicator (single process) the gather of one rank is a copy.
int lmpc_comm_allgather_dev(lmpc_ctx *c, const void *send, void *recv, long long bytes) {
    ARGCHK(c && send && recv && bytes > 0);
    HIPCHK(hipSetDevice(c->cfg.device));
    RESOLVE_PENDING();                              // (what is shipped to the peers has had its retry pass)
    if (!c->comm) { HIPCHK(hipMemcpyAsync(recv, send, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream)); return LMPC_OK; }
    NCCLCHK(ncclAllGather(send, recv, (size_t)bytes, ncclChar, (ncclComm_t)c->comm, c->stream));
    return LMPC_OK;
}

// host buffers (small control data: lap lengths, timings): staged through the communicator's scratch, same collective
int lmpc_comm_allgather(lmpc_ctx *c, const void *send_host, void *recv_host, long long bytes) {
    ARGCHK(c && send_host && recv_host && bytes > 0);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int world = c->comm ? c->comm_world : 1;
    int rc = comm_scratch(c, (size_t)bytes * (world + 1)); if (rc) return rc;
    char *d = (char *)c->comm_scr, *hm = (char *)c->comm_scr_h;
    memcpy(hm, send_host, (size_t)bytes);
    HIPCHK(hipMemcpyAsync(d, hm, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    rc = lmpc_comm_allgather_dev(c, d, d + bytes, bytes); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(hm + bytes, d + bytes, (size_t)bytes * world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(recv_host, hm + bytes, (size_t)bytes * world);
    return LMPC_OK;
}

int lmpc_comm_allreduce_max(lmpc_ctx *c, double *v, int n) {
    ARGCHK(c && v && n >= 1);
    HIPCHK(hipSetDevice(c->cfg.device));
    if (!c->comm) { HIPCHK(hipStreamSynchronize(c->stream)); return LMPC_OK; }
    int rc = comm_scratch(c, sizeof(double) * (size_t)n); if (rc) return rc;
    double *d = (double *)c->comm_scr, *hm = (double *)c->comm_scr_h;
    memcpy(hm, v, sizeof(double) * n);
    HIPCHK(hipMemcpyAsync(d, hm, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(ncclAllReduce(d, d, (size_t)n, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream));
    HIPCHK(hipMemcpyAsync(hm, d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(v, hm, sizeof(double) * n);
    return LMPC_OK;
}

// Barrier: a one-element all-reduce ENQUEUED on the context's stream behind whatever the rank has launched (no host staging, no allocation),
// then one drain of the stream: when it returns, every rank's stream had reached the same point.
int lmpc_comm_barrier(lmpc_ctx *c) {
    ARGCHK(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    { int rc = resolve_retries(c); if (rc) return rc; }
    if (c->comm) {
        int rc = comm_scratch(c, sizeof(double)); if (rc) return rc;
        NCCLCHK(ncclAllReduce(c->comm_scr, c->comm_scr, 1, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LMPC_OK;
}

// The per-lap exchange on the current rollout session: this rank's K fastest VALID laps (finished, at most T_max steps,
// no status bit other than LMPC_ST_INEXACT; ties towards the lower rollout index) are packed on the device from the session's logs and
// all-gathered.  records: world x K x (T_max + 1) x 14, lens: world x K (steps, -1 = empty slot), both host, identical on every rank.
int lmpc_rollout_exchange(lmpc_ctx *c, int K, int T_max, double *records, long long *lens, int *n_valid_local) {
    ARGCHK(c && c->ro && c->ro->active && records && lens && K >= 1 && T_max >= 1);
    lmpc_rollout_session *r = c->ro;
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(r->pstream));
    const int B = r->B, world = c->comm ? c->comm_world : 1;
    std::vector<int> done((size_t)B), st((size_t)B);
    HIPCHK(hipMemcpyAsync(done.data(), r->d_done, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(st.data(), r->d_stAcc, sizeof(int) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<int> cand;
    for (int b = 0; b < B; b++) if (done[b] >= 1 && done[b] <= T_max && done[b] <= r->t && (st[b] & ~LMPC_ST_INEXACT) == 0) cand.push_back(b);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return done[a] < done[b]; });
    if (n_valid_local) *n_valid_local = (int)cand.size();
    std::vector<int> sel((size_t)2 * K, -1);                      // [0, K): rollout index, [K, 2K): length
    for (int j = 0; j < K && j < (int)cand.size(); j++) { sel[j] = cand[j]; sel[K + j] = done[cand[j]]; }
    const size_t rec_doubles = (size_t)K * (T_max + 1) * 14;
    int *d_sel; double *d_send, *d_recv, *d_len;
    {   // exchange buffers in the communicator's scratch (grown once, kept): [send block | world x recv block | 2K selection ints]
        const size_t blk = sizeof(double) * (rec_doubles + K);
        int rc_ = comm_scratch(c, blk * (size_t)(world + 1) + sizeof(int) * 2 * K + 64); if (rc_) return rc_;
        d_send = (double *)c->comm_scr; d_sel = (int *)((char *)c->comm_scr + blk * (size_t)(world + 1));
    }
    hipError_t e = hipSuccess;
    d_len = d_send + rec_doubles;                                // send block: records | lengths (as doubles), one gather for both
    d_recv = d_send + rec_doubles + K;
    std::vector<double> lenv((size_t)K); for (int j = 0; j < K; j++) lenv[j] = (double)sel[K + j];
    int rc = LMPC_OK;
    e = hipMemcpyAsync(d_sel, sel.data(), sizeof(int) * 2 * K, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_len, lenv.data(), sizeof(double) * K, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lmpc_pack_laps_kernel, dim3(8, K), dim3(256), 0, c->stream, B, K, T_max, (const int *)d_sel, (const int *)(d_sel + K),
                           (const double *)r->d_logX, (const double *)r->d_logU, (const double *)r->d_logG, (const double *)r->d_finX, (const double *)r->d_finG, d_send);
        e = hipGetLastError();
    }
    if (e == hipSuccess) rc = lmpc_comm_allgather_dev(c, d_send, d_recv, (long long)(sizeof(double) * (rec_doubles + K)));
    std::vector<double> host;
    if (e == hipSuccess && rc == LMPC_OK) {
        host.resize((rec_doubles + K) * (size_t)world);
        e = hipMemcpyAsync(host.data(), d_recv, sizeof(double) * host.size(), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (rc) return rc;
    if (e != hipSuccess) return set_err(LMPC_E_HIP, "lmpc_rollout_exchange", hipGetErrorString(e));
    for (int w = 0; w < world; w++) {
        const double *src = host.data() + (size_t)w * (rec_doubles + K);
        memcpy(records + (size_t)w * rec_doubles, src, sizeof(double) * rec_doubles);
        for (int j = 0; j < K; j++) lens[(size_t)w * K + j] = (long long)src[rec_doubles + j];
    }
    return LMPC_OK;
}

}  // extern "C"
