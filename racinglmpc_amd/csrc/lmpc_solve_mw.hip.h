// racinglmpc_amd/csrc/lmpc_solve_mw.hip.h -- multi-wave variant of lmpc_solve_kernel for SMALL batches.
//
// Same algorithm and the same LDS layout as lmpc_solve_kernel<N,S> (lmpc_kernels.hip.h), but one QP is worked on by a
// work-group of NW wavefronts (one per SIMD of a CU).  With B <= #CUs a single wave per QP leaves three SIMDs of every
// CU idle and the solve is bound by the dependent-issue latency of that one instruction stream; here
//   * the sequential recursions (terminal Cholesky, Riccati stage loop, register sweeps) stay on wave 0, ordered inside
//     the wave by s_waitcnt only (no work-group barrier),
//   * every row-/entry-parallel phase (residuals, barrier bookkeeping, right-hand sides, step lengths, costates, update)
//     is spread over all NW*64 threads, different loops landing on different waves so that they issue concurrently,
//   * wave-level reductions are combined across waves through a few LDS slots.
// The host picks this kernel when B is small (lmpc_capi.hip: pick_solver).
#pragma once
#include "lmpc_kernels.hip.h"

// (round 4 tried a compiler fence only -- one wave's LDS operations execute in order -- and nothing got faster: the wait stays)
#define WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Compiler fence behind a group of LDS reads into locals: all of them are issued before the first use (one latency for the group).  With the
// register file full the compiler otherwise interleaves read pairs with their arithmetic, one s_waitcnt and one LDS round trip per pair.
#define LDS_GROUP() asm volatile("" ::: "memory")
// barrier of one pipeline step (phase 2 of the Newton iteration): every wave of the work-group executes the same number of them
#define STEP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#ifdef LMPC_TIMING
// (every wave stamps through its lane 0 into its own quarter of the buffer: 4000 (id, cycle) pairs per wave)
#define TSMW(id) do { if (io.tbuf && b == 0 && lane == 0 && tcnt < 4000) { io.tbuf[2 * (4000 * wave + tcnt)] = (id); io.tbuf[2 * (4000 * wave + tcnt) + 1] = (long long)__builtin_readcyclecounter(); tcnt++; } } while (0)
#else
#define TSMW(id) do { } while (0)
#endif

template <int N, int S, int NW>
// (registers: four waves per QP = one wave per SIMD; two waves per QP = two waves per SIMD only where the LDS footprint lets three or more
// QPs share a CU -- long horizons keep their sweep operands (3 N doubles per lane) in registers and need the full file: at N = 40 the
// 256-register build spilled 221 VGPRs to scratch)
__global__ __launch_bounds__(WAVE *NW, (NW == 4 || solve_lds<N, S>::tot * 8 * 3 > 160 * 1024 || solve_lds<N, S>::CH > 1) ? 1 : 2) void lmpc_solve_kernel_mw(lmpc_dev_params p, int B, lmpc_solve_io io) {
    static_assert(NW == 2 || NW == 4, "wave 0 runs the sequential recursions, waves 1 .. NW-1 everything that can run beside them");
    constexpr bool BCF = true;                              // bound_ctrl cross-lane moves (dpp_mv) in every reduction: the EXEC mask is full there (lmpc_debug_exec_audit)
    extern __shared__ double sm[];
    using LL = solve_lds<N, S>;
    constexpr int M = LL::M;
    constexpr bool term = S > 0;
    constexpr int NT = WAVE * NW;
    constexpr bool ROWS_OFF_W0 = S > 0 && S <= WAVE && NT >= 8 * N + WAVE && SWEEP_BF<N>;     // wave 0 owns the lambda rows and nothing else (see slot_row; the four-wave kernel: the other rows fit threads 64 .. 255)
    constexpr int RPL = ((ROWS_OFF_W0 ? 8 * N + WAVE : M) + NT - 1) / NT;           // inequality rows per thread
    const int b = blockIdx.x;
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: tests on it are scalar branches, not exec-mask regions with a v_cmp each
    const int lg = lane >> 3, lc = lane & 7;
    const bool w0 = wave == 0;
    // the critical wave first wherever an arbiter chooses between waves (LDS, scalar cache): four waves per QP only -- +0.3 % there; with two waves per QP
    // the helper shares its SIMD with another QP's wave 0 and starves: batch 1024 lost 4 %
    if constexpr (NW == 4) { if (w0) __builtin_amdgcn_s_setprio(3); }
    double *AB = sm + LL::oAB, *C = sm + LL::oC, *x = sm + LL::ox, *u = sm + LL::ou, *s = sm + LL::os, *lam = sm + LL::olam;
    double *dx = sm + LL::odx, *du = sm + LL::odu, *ds = sm + LL::ods, *dl = sm + LL::odl, *nu = sm + LL::onu, *dnu = sm + LL::odnu;
    double *m = sm + LL::om, *th = sm + LL::oth, *h = sm + LL::oh, *dm = sm + LL::odm;
    double *rx = sm + LL::orx, *ru = sm + LL::oru, *rs = sm + LL::ors, *rl = sm + LL::orl;
    double *Phi = sm + LL::oPhi, *PiAll = sm + LL::oPiAll, *Mi = sm + LL::oMi, *gam = sm + LL::ogam, *gup = sm + LL::ogup, *pst = sm + LL::opst;
    double *kap = sm + LL::okap, *rDs = sm + LL::orDs, *eta = sm + LL::oeta, *ee = sm + LL::oe;
    double *Ri = sm + LL::oRi, *rsq = sm + LL::orsq, *ct = sm + LL::oct, *Mt = sm + LL::oMt, *Wl = sm + LL::oWl, *McL = sm + LL::oMc;
    double *SS = sm + LL::oSS, *Qsel = sm + LL::oQsel, *y7 = sm + LL::oy7, *z7 = sm + LL::oz7, *w7 = sm + LL::ow7, *PiT = sm + LL::oPiT, *sT = sm + LL::osT;
    double *par = sm + LL::opar;
    const double *Fx = par + PAR_FX, *Fu = par + PAR_FU, *bx = par + PAR_BX, *bu = par + PAR_BU, *Q2 = par + PAR_Q2, *Qf2 = par + PAR_QF2,
                 *R2 = par + PAR_R2, *dR2 = par + PAR_DR2, *T2p = par + PAR_T2, *xRef = par + PAR_XREF;
    __shared__ double red[10 * 4];                           // cross-wave reduction slots
    __shared__ double phi_sh[8 * N];
    __shared__ int st_sh, bad_sh;
    __shared__ double gs0[WAVE];                             // wave 0's per-lane share of sum t mu (it does not take part in the residual reductions)
    // wave 0's lanes that have nothing to store in a stage of the recursion store here (no exec-mask region on the chain).  Same words as gs0, which is
    // written in the step phase and read in phase 1, i.e. dead during the recursion: with 512 bytes of its own the kernel's LDS was 41 496 bytes and
    // the fourth QP of the two-wave kernel no longer fitted a CU (163 840 / 4 = 40 960) -- batch 1024 ran in two rounds, 2.85 -> 2.18 M steps/s.
    double *const dump_sh = gs0;
    __shared__ int sel_start[LMPC_MAX_USED_LAPS];
    __shared__ double ss_rowsum[6];                          // sum over the selected safe-set points of each state (loop invariant)
    double *phi = phi_sh;
    if (tid == 0) { st_sh = 0; bad_sh = 0; }
    // stage the parameter block
    if (tid < 12) par[PAR_FX + tid] = p.Fx[tid];
    if (tid < 8) par[PAR_FU + tid] = p.Fu[tid];
    if (tid < 2) { par[PAR_BX + tid] = p.bx[tid]; par[PAR_DR2 + tid] = p.dR2[tid]; }
    if (tid < 4) { par[PAR_BU + tid] = p.bu[tid]; par[PAR_R2 + tid] = p.R2[tid]; }
    if (tid < 36) { par[PAR_Q2 + tid] = p.Q2[tid]; par[PAR_QF2 + tid] = p.Qf2[tid]; }
    if (tid < 6) { par[PAR_T2 + tid] = p.T2[tid]; par[PAR_XREF + tid] = p.xRef[tid]; }
    if (tid == 0) { par[PAR_AS] = p.a_s; par[PAR_CS] = p.c_s; }
    __syncthreads();
    const double a_s = par[PAR_AS], c_s = par[PAR_CS];

    auto red_put = [&](int slot, double v) { if (lane == 0) red[slot * 4 + wave] = v; };
    auto red_sum = [&](int slot) { double r = red[slot * 4]; for (int w = 1; w < NW; w++) r += red[slot * 4 + w]; return r; };
    auto red_max = [&](int slot) { double r = red[slot * 4]; for (int w = 1; w < NW; w++) r = fmax(r, red[slot * 4 + w]); return r; };
    auto red_min = [&](int slot) { double r = red[slot * 4]; for (int w = 1; w < NW; w++) r = fmin(r, red[slot * 4 + w]); return r; };

    // K2: safe-set selection, one lap per wave (k2_select, lmpc_kernels.hip.h), then the regression status bits of this problem
    if constexpr (term) {
        k2_select<N, S, NW>(p, io, b, lane, wave, SS, Qsel, sel_start, &st_sh); __syncthreads();
        if (tid < 6) { double v = 0.0; for (int c = 0; c < S; c++) v += SS[tid * S + c]; ss_rowsum[tid] = v; }
    }
    if (io.rstatus && tid < N) { const int rs_ = io.rstatus[(size_t)b * N + tid]; if (rs_) atomicOr(&st_sh, rs_); }
    if (!(io.mode & 2)) { if (tid == 0) io.status[b] = st_sh; return; }

    // ------------------------------------------------------------------------------------------------
    // K3: QP solve (same formulation as lmpc_solve_kernel).
    // ------------------------------------------------------------------------------------------------
    for (int i = tid; i < 36 * N; i += NT) { const int k = i / 36, r = (i % 36) / 6, c = i % 6; AB[k * 48 + r * 8 + c] = io.A[(size_t)b * 36 * N + i]; }
    for (int i = tid; i < 12 * N; i += NT) { const int k = i / 12, r = (i % 12) >> 1, c = i & 1; AB[k * 48 + r * 8 + 6 + c] = io.Bm[(size_t)b * 12 * N + i]; }
    for (int i = tid; i < 6 * N; i += NT) { C[i] = io.C[(size_t)b * 6 * N + i]; nu[i] = 0.0; }
    if (tid < 6) x[tid] = io.x0[(size_t)b * 6 + tid];
    for (int i = tid; i < 2 * N; i += NT) u[i] = 0.0;
    const double uOld0 = io.uOld[(size_t)b * 2 + 0], uOld1 = io.uOld[(size_t)b * 2 + 1];
    __syncthreads();
    if (w0) {                                                // (the register form of the one-wave kernel, rollout_start, measured 1.5 % slower here: code generation)
#pragma unroll 1
        for (int k = 0; k < N; k++) {                      // strictly interior start: u = 0, x by roll-out
            if (lane < 6) {
                double v = C[k * 6 + lane];
#pragma unroll
                for (int j = 0; j < 6; j++) v = fma(AB[k * 48 + lane * 8 + j], x[k * 6 + j], v);
                x[(k + 1) * 6 + lane] = v;
            }
            WSYNC();
        }
    }
    __syncthreads();
    const double s_init = c_s > 1.0 ? 1.0 / c_s : 1.0;      // see lmpc_solve_kernel
    for (int i = tid; i < 2 * N; i += NT) {
        const int k = i >> 1, j = i & 1; double f = 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], x[k * 6 + c], f);
        const double viol = f - bx[j];
        s[i] = viol > 0.0 ? viol + 1.0 : s_init;
    }
    double qmax = 0.0;
    if constexpr (term) {
        for (int c = lane; c < S; c += WAVE) { if (w0) lam[c] = 1.0 / (double)S; qmax = fmax(qmax, fabs(Qsel[c])); }
        qmax = wmax(qmax);                                  // every wave computes the same value
    }
    const double mu0 = fmax(1.0, 0.01 * (term ? qmax : 1.0));
    if (tid < 4 && !(bu[tid] > 0.0)) atomicOr(&st_sh, LMPC_ST_NOT_INTERIOR);
    double eta_m = 0.0;
    __syncthreads();

    auto rowF = [&](int r, const double *xx, const double *uu, const double *ss_, const double *ll) -> double {
        if (r < 2 * N) { const int k = r >> 1, j = r & 1; double f = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], xx[k * 6 + c], f);
            return f - ss_[r]; }
        if (r < 6 * N) { const int q = r - 2 * N, k = q >> 2, j = q & 3; return Fu[j * 2] * uu[k * 2] + Fu[j * 2 + 1] * uu[k * 2 + 1]; }
        if (r < 8 * N) return -ss_[r - 6 * N];
        return -ll[r - 8 * N];
    };
    auto rowb = [&](int r) -> double { if (r < 2 * N) return bx[r & 1]; if (r < 6 * N) return bu[(r - 2 * N) & 3]; return 0.0; };

    // Per-thread row state (row = tid + NT j).  The slack t of an inequality row, its reciprocal and the barrier weight theta = mu / t are
    // carried from iteration to iteration by the thread that owns the row (t <- t + alpha dt with dt = -F dw: the rows are linear, and the
    // update keeps the relative accuracy of a slack that has shrunk to 1e-13, which b - F w recomputed from the iterate does not); the
    // terminal slack likewise (s_T <- s_T + alpha ds_T).  One pass over the rows per iteration, together with the step, instead of three.
    // rt is the EFFECTIVE reciprocal 1 / max(t, 1e-11 mu) (barrier weight capped at 1e11, see LMPC_TH_INV): matrix (theta = mu rt) and right-hand
    // sides (predictor h = t mu rt, corrector h = (t mu + dt dmu - sigma gap) rt) use the same one, so the capped row is a regularised row.
    // Row ownership (round 4): slot s = tid + NT j owns inequality row  s < S ? 8 N + s : s - S -- the lambda rows, whose barrier weights are the
    // terminal block's, belong to the FIRST S threads: to wave 0 when S <= 64.  The wave that factorises the terminal block then knows the new
    // weights the moment it has updated its own rows, and starts the next iteration's terminal factor (Gram matrix part) inside the step phase
    // instead of behind the barrier that ends it (TEARLY; four waves per QP, one terminal-block column per lane).
    // (round 6, A / B on the GPU: the same for two waves per QP -- wave 0 owns the lambda rows there too -- changes nothing: 1.692 vs 1.694 M/s at batch 512, 2.98 vs 3.00 M/s at 1024)
    constexpr bool TEARLY = term && NW == 4 && LL::CH == 1 && S <= WAVE;
    // (ROWS_OFF_W0: the other rows start at thread 64.  Wave 0's spare lanes used to take the first 64 - S lane rows, so every pass over the rows
    //  made the critical wave run the lane rows' code as well as the lambda rows' -- divergent paths cost the sum of both -- on a wave that is bound by
    //  instruction issue.)
    auto slot_row = [&](int j) -> int {
        const int s_ = tid + NT * j;
        if constexpr (ROWS_OFF_W0) return s_ < S ? 8 * N + s_ : (s_ >= WAVE && s_ - WAVE < 8 * N ? s_ - WAVE : -1);
        else return s_ >= M ? -1 : (s_ < S ? 8 * N + s_ : s_ - S);
    };
    double t_r[RPL], rt_r[RPL], tp_r[RPL], dt_r[RPL];
    double gsum_c = 0.0;                                   // this thread's share of sum t mu (the complementarity gap)
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const int r = slot_row(j);
        t_r[j] = 1.0; rt_r[j] = 1.0; tp_r[j] = 0.0; dt_r[j] = 0.0;
        if (r >= 0) {
            const double tt = rowb(r) - rowF(r, x, u, s, lam), mm = mu0 / tt; t_r[j] = tt; m[r] = mm;
            const double rt = barrier_rt(tt, mm), thv = mm * rt;
            rt_r[j] = rt; th[r] = thv; h[r] = tt * thv; gsum_c = fma(tt, mm, gsum_c);      // h: the predictor's right-hand side, see the step
        }
    }
    if (w0) gs0[lane] = gsum_c;                            // wave 0's share of sum t mu: folded into the gap by wave 1 (wave 0 skips the residual reductions)
    if constexpr (term) { if (wave == NW - 1) ss_times<S>(SS, lam, x + N * 6, sT, lane); }      // terminal slack s_T = SS lambda - x_N
    // loop-invariant pieces of the stage Hessian W for this lane's (a, c) = (lg, lc) tile entry (used by wave 0)
    const ricc_consts rc = ricc_setup(lane, Q2, Fx, R2, dR2, Fu);
    double ph[N];
    constexpr int CH = LL::CH;                               // terminal-block columns per lane of wave 0 (column = lane + 64 ch; 1 up to 58 safe-set points)
    constexpr bool QX = 8 * LL::CW <= 64 * N;                // second pass of the terminal factor's orthogonalisation: needs the Phi tiles' LDS as its Gram tile
    double mcol[CH][7], tsq_lane[CH];                        // columns of M = [E D^-1/2 | T7^-1/2]; T^-1/2 entry of a slack column (loop invariant)
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        const int col = lane + WAVE * ch;
        tsq_lane[ch] = (term && col >= S && col < S + 6) ? frsqrt(T2p[col - S]) : 0.0;
#pragma unroll
        for (int j = 0; j < 7; j++) mcol[ch][j] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < N; k++) ph[k] = 0.0;
    __syncthreads();

    // thread ranges of the concurrent loops (different loops start on different waves)
    constexpr int O1 = NW > 1 ? WAVE : 0, O2 = NW > 2 ? 2 * WAVE : O1, O3 = NW > 3 ? 3 * WAVE : O2;
#define FOR_OFF(i, n, off) for (int i = (tid >= (off) ? tid - (off) : tid - (off) + NT); i < (n); i += NT)

    // Helper waves stride their loops over the NT3 = NT - 64 threads of waves 1 .. NW-1 (wave 0 is busy with the sequential work)
    constexpr int NT3 = NT - WAVE;
    const int t3 = tid - WAVE;
#define FOR_HELP(i, n, off) for (int i = (t3 >= ((off) % NT3) ? t3 - ((off) % NT3) : t3 - ((off) % NT3) + NT3); i < (n); i += NT3)
    // a loop that belongs to ONE wave (w, or the last wave of a smaller work-group), 64 items per trip.  A wave executes every loop BODY one of its
    // lanes takes part in, so a wave that straddles three loops pays the latency chains of all three: phase 1's helper loops are handed out wave by wave
    // (before: wave 2 ran the tail of the stationarity rows, the u / s rows AND half the lambda rows -- 4.7 k cycles against 3.3 k and 2.9 k)
#define FOR_WAVE(i, n, w) for (int i = (wave == ((w) < NW ? (w) : NW - 1)) ? lane : (n); i < (n); i += WAVE)
    // ... and a loop of up to 128 items in ONE trip: items 0 .. 63 on wave w, the rest on the low lanes of wave w2 (two trips of one wave cost two
    // latency chains; with two waves per QP both names mean wave 1 and the loop takes its trips there)
#define WCL(w) ((w) < NW ? (w) : NW - 1)
#define FOR_WAVE2(i, n, w, w2) for (int i = WCL(w) == WCL(w2) ? (wave == WCL(w) ? lane : (n)) : (wave == (w) ? lane : (wave == (w2) ? WAVE + lane : (n))), \
                                        i##_st = WCL(w) == WCL(w2) ? WAVE : 2 * WAVE; i < (n); i += i##_st)
    int tcnt = 0;
    // Terminal factor of the CURRENT barrier weights of the lambda rows, by wave 0, in two parts.
    //   part A: columns of M = [E D^-1/2 | T^-1/2] (registers + LDS copy)
    //   part B: Gram matrix W = M M' on the matrix cores, Cholesky R'R = W, R^-1, Pi_term
    // Part B sits in phase 1 of the Newton iteration.  Part A does too in the generic kernel; with TEARLY wave 0 runs it at the end of the PREVIOUS
    // iteration's step phase, right after it has updated its own (lambda) rows, while the other waves are still busy with the costates and the
    // primal update (nothing else reads rsq / Mt / Wl / mcol between the corrector solve and the next factorisation).
    auto term_partA = [&]() {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            const int col = lane + WAVE * ch;
            // (per-lane if / else kept here: the branch-free form of the one-wave kernel measured 0.6 % slower at batch 256 and 1.3 % at 1024 in this kernel --
            //  profiles/r5g_ab.txt; what makes an `else` inside the loop safe is the ISA check every build goes through, racinglmpc_amd/isa_check.py)
#pragma unroll
            for (int j = 0; j < 7; j++) mcol[ch][j] = 0.0;
            if (col < S) {
                const double rs_ = frsqrt(th[8 * N + col] + p.reg); rsq[col] = rs_;
#pragma unroll
                for (int j = 0; j < 6; j++) mcol[ch][j] = SS[j * S + col] * rs_;
                mcol[ch][6] = rs_;
            } else {
                rsq[col] = 1.0;
#pragma unroll
                for (int j = 0; j < 6; j++) if (col - S == j) mcol[ch][j] = tsq_lane[ch];
            }
#pragma unroll
            for (int j = 0; j < 7; j++) Mt[col * 8 + j] = mcol[ch][j];
            Mt[col * 8 + 7] = 0.0;
        }
    };
    bool qx_late = false;                                    // second orthogonalisation pass of the terminal factor on (term_reorth): set per iteration, wave-uniform
    auto term_partB = [&](int &numeric_bad) {
        double Rr[7][7], rinv[7];
        WSYNC();
        gram8_mfma<CH>(Mt, Wl, lane);             // Gram matrix W = M M' on the matrix cores
        WSYNC();
#pragma unroll
        for (int i = 0; i < 7; i++)
#pragma unroll
            for (int j = i; j < 7; j++) Rr[i][j] = Wl[i * 8 + j];
        LDS_GROUP();
#pragma unroll
        for (int i = 0; i < 7; i++) {                                                 // Cholesky, row by row
            double d_ = Rr[i][i];
#pragma unroll
            for (int k = 0; k < i; k++) d_ = fma(-Rr[k][i], Rr[k][i], d_);
            if (!(d_ > 0.0)) { numeric_bad = 1; d_ = 1.0; }
            rinv[i] = frsqrt(d_); const double rii = d_ * rinv[i]; Rr[i][i] = rii;
#pragma unroll
            for (int j = i + 1; j < 7; j++) {
                double v = Rr[i][j];
#pragma unroll
                for (int k = 0; k < i; k++) v = fma(-Rr[k][i], Rr[k][j], v);
                Rr[i][j] = v * rinv[i];
            }
        }
        {   // Ri = R^-1 (upper): lane j < 7 back-substitutes column j (R is uniform across the lanes), 28 dependent operations instead of 140.
            // Terms beyond the diagonal multiply exact zeros, so every entry is bit-identical to the column-by-column form.
            double col[7];
#pragma unroll
            for (int i = 6; i >= 0; i--) {
                double v = 0.0;
#pragma unroll
                for (int k = i + 1; k < 7; k++) v = fma(-Rr[i][k], col[k], v);
                col[i] = lane == i ? rinv[i] : (lane > i ? v * rinv[i] : 0.0);
            }
            if (lane < 7) {
#pragma unroll
                for (int i = 0; i < 7; i++) Ri[i * 7 + lane] = col[i];
            }
        }
        WSYNC();
        // late iterations: rows of Q = M' R^-1 explicit in mcol, re-orthogonalised -- where an LDS tile for the second Gram matrix is free (term_reorth, lmpc_kernels.hip.h): the Phi_k
        // tiles are dead between the step phase and the stage recursion -- 64 N doubles against the 8 x 64 CH the tile needs
        term_reorth<CH, false, QX>(mcol, Ri, Phi, Wl, lane, qx_late);
        WSYNC();
        if (lane < 36) {                                 // Pi_term = (Ri Ri')[0:6,0:6]  (Ri is zero below the diagonal: all seven terms, same sum)
            const int i = lane / 6, j = lane % 6; double v = 0.0;
            double ra[7], rb[7];
#pragma unroll
            for (int k = 0; k < 7; k++) { ra[k] = Ri[i * 7 + k]; rb[k] = Ri[j * 7 + k]; }
            LDS_GROUP();
#pragma unroll
            for (int k = 0; k < 7; k++) v = (k >= i && k >= j) ? fma(ra[k], rb[k], v) : v;
            PiT[lane] = v;
        }
                };
    // Corrector solve (right-hand side h in LDS; the factors of this iterate are in place).  Three barriers:
    //   C1  wave 0: terminal costate p_N;  helper waves: gamma_k for every stage, each entry rebuilding the two slack eliminations and the
    //       reduced input gradient of its stage from h (no intermediate arrays, hence no barrier between them and gamma)
    //   C2  wave 0 alone: backward sweep, feed-forward terms phi_k (lane-parallel, two passes of eight stages), forward sweep
    //   C3  slack and terminal steps
    auto kkt_solve = [&](double re_sum) {
        double c_t[CH], xiN = 0.0, mc_g = 0.0, y7v = 0.0, pN = 0.0;   // wave 0's registers: last stage of the forward sweep, (M c~)[lg], y7[lg], p_N[lg]
        if (w0) {
            if constexpr (term) {
#pragma unroll
                for (int ch = 0; ch < CH; ch++) {
                    const int col = lane + WAVE * ch;
                    c_t[ch] = col < S ? (rl[col] + h[8 * N + col]) * rsq[col] : 0.0;
                    ct[col] = c_t[ch];
                }
                WSYNC();
                double acc = 0.0;                               // M c~ : lane (j, part) adds every 8th of the 64 CH columns
                {
                    double mq[8 * CH], cq[8 * CH];
#pragma unroll
                    for (int q = 0; q < 8 * CH; q++) { mq[q] = Mt[lc * 8 + (lg < 7 ? lg : 0) + 64 * q]; cq[q] = ct[lc + 8 * q]; }
                    LDS_GROUP();
#pragma unroll
                    for (int q = 0; q < 8 * CH; q++) acc = fma(mq[q], cq[q], acc);
                    acc = lg < 7 ? acc : 0.0;
                }
                acc = sum_over_c(acc);
                mc_g = acc;                                     // (M c~)[lg] in every lane of group lg
            }
            {   // terminal costate p_N = (rx_N + [Ri (Ri' d0 + y7)]_{0:6}, 0), y7 = Ri' (M c~), in registers (term_costate: lmpc_kernels.hip.h)
                double v = 0.0;
                if constexpr (term) v = term_costate(Ri, mc_g, re_sum, lg, lc, y7v);
                pN = lg < 6 ? rx[N * 6 + lg] + v : 0.0;
                if (lc == 0) pst[N * 8 + lg] = pN;              // (the phi pass reads it from LDS)
            }
            if (lane < 6) dx[lane] = 0.0;
        } else {
            FOR_HELP(i, 8 * N, 0) {                             // gamma_k = [gx' - Fx' eta ; 0] + Phi[6:8,:]' gu'
                const int k = i >> 3, c = i & 7, i0 = 2 * k, i1 = 2 * k + 1;
                const double hl0 = h[i0], hl1 = h[i1];
                const double e0 = -(rs[i0] + hl0 + h[6 * N + i0]), e1 = -(rs[i1] + hl1 + h[6 * N + i1]);     // slack elimination of the stage's two lane rows
                const double eta0 = hl0 + th[i0] * e0 * rDs[i0], eta1 = hl1 + th[i1] * e1 * rDs[i1];
                double g0 = ru[i0], g1 = ru[i1];                // gu' = ru - Fu' h_u
#pragma unroll
                for (int j = 0; j < 4; j++) { const double hu = h[2 * N + 4 * k + j]; g0 -= Fu[j * 2] * hu; g1 -= Fu[j * 2 + 1] * hu; }
                double v = 0.0;
                if (c < 6) { v = rx[k * 6 + c]; v -= Fx[c] * eta0 + Fx[6 + c] * eta1; }
                v = fma(Phi[k * 64 + 48 + c], g0, v);
                v = fma(Phi[k * 64 + 56 + c], g1, v);
                gam[i] = v;
                if (c == 0) { ee[i0] = e0; ee[i1] = e1; gup[i0] = g0; gup[i1] = g1; }
            }
        }
        __syncthreads();                                        // C1
        TSMW(30);
        if (w0) {
            {   // backward sweep p_k = Phi_k' p_{k+1} + gamma_k in registers (see lmpc_solve_kernel)
                double gm[N];
#pragma unroll
                for (int k = 0; k < N; k++) gm[k] = (k & 1) ? gam[k * 8 + lg] : gam[k * 8 + lc];
                double pv = ((N - 1) & 1) ? lane_gather(pN, 32 * lc) : pN;
                if constexpr (SWEEP_BF<N>) {
                    const sweep_dst wd = bwd_sweep_dst<N>(pst, gam, lane, lg, lc); lds_f64 *wE = (lds_f64 *)wd.wE, *wO = (lds_f64 *)wd.wO;   // (branch-free stores, see sweep_dst: gamma is in registers, its LDS is the dump)
                    LDS_GROUP();
#pragma unroll
                    for (int k = N - 1; k >= 0; k--) {
                        double pr = ph[k] * pv;
                        if (k & 1) { pr = sum_over_c<true>(pr); pv = pr + gm[k]; *wO = pv; wO += wd.dO; }
                        else { pr = sum_over_g<true>(pr); pv = pr + gm[k]; *wE = pv; wE += wd.dE; }
                        SWEEP_PIN(wE, wO);
                    }
                } else {
                    LDS_GROUP();
#pragma unroll
                    for (int k = N - 1; k >= 0; k--) {
                        double pr = ph[k] * pv;
                        if (k & 1) { pr = sum_over_c(pr); pv = pr + gm[k]; if (lc == 0) pst[k * 8 + lg] = pv; }
                        else { pr = sum_over_g(pr); pv = pr + gm[k]; if (lg == 0) pst[k * 8 + lc] = pv; }
                    }
                }
            }
            WSYNC();
            TSMW(31);
            // phi_k = [-B k0 ; -k0],  k0_k = M_uu^-1 (gu' + [B; I]' p_{k+1}): group lg takes stage lg + 8 pass, lane lc row lc of [B_k; I]
#pragma unroll
            for (int k0_ = 0; k0_ < N; k0_ += 8) {
                const int k = k0_ + lg; const bool on = k < N; const int kk = on ? k : 0;
                const double bq0 = lc < 6 ? AB[kk * 48 + lc * 8 + 6] : (lc == 6 ? 1.0 : 0.0), bq1 = lc < 6 ? AB[kk * 48 + lc * 8 + 7] : (lc == 7 ? 1.0 : 0.0);
                const double pq = pst[(kk + 1) * 8 + lc], gq = lc >= 6 ? gup[2 * kk + lc - 6] : 0.0;
                const double m00 = Mi[kk * 4], m01 = Mi[kk * 4 + 1], m10 = Mi[kk * 4 + 2], m11 = Mi[kk * 4 + 3];
                LDS_GROUP();
                const double w0_ = sum_over_c<BCF>(fma(bq0, pq, lc == 6 ? gq : 0.0)), w1_ = sum_over_c<BCF>(fma(bq1, pq, lc == 7 ? gq : 0.0));
                const double k00 = m00 * w0_ + m01 * w1_, k01 = m10 * w0_ + m11 * w1_;
                if (on) phi[k * 8 + lc] = -(bq0 * k00 + bq1 * k01);
            }
            WSYNC();
            TSMW(32);
            {   // forward sweep xi_{k+1} = Phi_k xi_k + phi_k in registers
                double fm[N];
#pragma unroll
                for (int k = 0; k < N; k++) fm[k] = (k & 1) ? phi[k * 8 + lc] : phi[k * 8 + lg];
                double xi = 0.0;
                if constexpr (SWEEP_BF<N>) {
                    const sweep_dst wd = fwd_sweep_dst<N>(dx, du, phi, lane, lg, lc); lds_f64 *wE = (lds_f64 *)wd.wE, *wO = (lds_f64 *)wd.wO;   // (phi is in registers)
                    LDS_GROUP();
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double pr = ph[k] * xi;
                        if (k & 1) pr = sum_over_g<true>(pr); else pr = sum_over_c<true>(pr);
                        xi = pr + fm[k];
                        if (k & 1) { *wO = xi; wO += wd.dO; } else { *wE = xi; wE += wd.dE; }
                        SWEEP_PIN(wE, wO);
                    }
                } else {
                    LDS_GROUP();
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        double pr = ph[k] * xi;
                        int idx;
                        if (k & 1) { pr = sum_over_g(pr); idx = lc; } else { pr = sum_over_c(pr); idx = lg; }
                        xi = pr + fm[k];
                        const bool wr = (k & 1) ? (lg == 0) : (lc == 0);
                        if (wr) { if (idx < 6) dx[(k + 1) * 6 + idx] = xi; else du[k * 2 + (idx - 6)] = xi; }
                    }
                }
                xiN = xi;
            }
        }
        __syncthreads();                                        // C2
        TSMW(33);
        FOR_OFF(i, 2 * N, O1) {
            const int k = i >> 1, j = i & 1; double f = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], dx[k * 6 + c], f);
            ds[i] = (th[i] * f + ee[i]) * rDs[i];
        }
        if constexpr (term) {
            if (w0) {
                double wq[7];
                if constexpr ((N - 1) & 1) term_omega(Ri, y7v, xiN, re_sum, lg, lc, wq, qx_late);     // (even N: every lane ends the sweep with xi_N[lc])
                else {   // z7 = Ri' d7 + y7, d7 = (dx_N ; -re_sum);  omega' = Ri z7
                    if (lane < 7) w7[lane] = lane < 6 ? dx[N * 6 + lane] : -re_sum;           // d7 (w7 is free until omega' is written)
                    WSYNC();
                    const double zv = ri_t_times(Ri, w7, lg, lc) + y7v;
                    WSYNC();
                    if (lc == 0 && lg < 7) z7[lg] = zv;
                    WSYNC();
                    if (!qx_late) {                                  // omega' = Ri z7 (late iterations: z7 itself, the lanes' mcol holds Q = M' R^-1 then, see term_omega)
                        const double wv = ri_times(Ri, z7, lg, lc);
                        WSYNC();
                        if (lc == 0 && lg < 7) z7[lg] = wv;
                        WSYNC();
                    }
#pragma unroll
                    for (int j = 0; j < 7; j++) wq[j] = z7[j];
                }
#pragma unroll
                for (int ch = 0; ch < CH; ch++) {
                    const int col = lane + WAVE * ch;
                    double v = -c_t[ch];                        // v = -c~ + Q z7
                    const double rq = rsq[col];
                    LDS_GROUP();
#pragma unroll
                    for (int j = 0; j < 7; j++) v = fma(mcol[ch][j], wq[j], v);
                    if (col < S) dl[col] = v * rq;
                }
            }
        }
        __syncthreads();                                        // C3
    };

    int it = 0, converged = 0, sep = 0;
    double gap = 0.0, rdn = 0.0, ren = 0.0, gap_prev = -1.0, lsum_all = 0.0, step_pp = INFINITY;      // step_pp: the (x, u) step before the last one (the last one is in reduction slot 6)
    const double qscale = fmax(1.0, qmax);
    constexpr int FB_WAVE = NW > 2 ? 2 : 1;                  // wave that runs the feed-forward follower (its own wave when there are three helpers)
#pragma unroll 1
    for (it = 0; it <= p.max_iter; it++) {
        TSMW(10);
        // ---- phase 1: wave 0 factorises the terminal block of this iterate; meanwhile the helper waves evaluate the residuals, the slack
        //      eliminations and everything of the predictor's right-hand side (h = mu) that does not need the factorisation -------------
        int numeric_bad = 0;
        if (w0) {
            if constexpr (term) {
                if (!TEARLY || it == 0) term_partA();
                // (this iterate's gap is being summed by the helper waves right now: the decision uses the previous iterate's, one decade earlier)
                qx_late = QX && it > 0 && gap < 10.0 * LMPC_QX_GAP;
                term_partB(numeric_bad);
            }
            WSYNC();
            TSMW(110);
        } else {
            double rmax = 0.0, remax = 0.0, lsum = 0.0;
            FOR_WAVE2(i, 6 * (N + 1), 1, 3) {                 // stationarity rows of x_k (wave 1, the last few on wave 3)
                const int k = i / 6, c = i % 6; double v = 0.0;
                if (k >= 1) {
                    const double *Qk = k < N ? Q2 : Qf2;
#pragma unroll
                    for (int j = 0; j < 6; j++) v = fma(Qk[c * 6 + j], x[k * 6 + j] - xRef[j], v);
                    v += nu[(k - 1) * 6 + c];
                    if (k < N) {
                        v += Fx[c] * m[2 * k] + Fx[6 + c] * m[2 * k + 1];
#pragma unroll
                        for (int j = 0; j < 6; j++) v -= AB[k * 48 + j * 8 + c] * nu[k * 6 + j];
                    } else if (term) v -= T2p[c] * sT[c];
                    rmax = fmax(rmax, fabs(v));
                }
                rx[i] = v;
                if (i < 6) dx[i] = 0.0;
            }
            FOR_WAVE(i, 2 * N, 2) {                           // rows of u_k and s_k; slack elimination and the predictor's (h = mu) reduced gradients (wave 2)
                const int k = i >> 1, c = i & 1;
                const double up = k > 0 ? u[(k - 1) * 2 + c] : (c == 0 ? uOld0 : uOld1);
                double v = R2[c * 2] * u[k * 2] + R2[c * 2 + 1] * u[k * 2 + 1] + dR2[c] * (u[i] - up);
                if (k < N - 1) v += dR2[c] * (u[i] - u[(k + 1) * 2 + c]);
                double fh = 0.0, fhp = 0.0;
#pragma unroll
                for (int j = 0; j < 4; j++) { fh = fma(Fu[j * 2 + c], m[2 * N + 4 * k + j], fh); fhp = fma(Fu[j * 2 + c], h[2 * N + 4 * k + j], fhp); }
                v += fh;
#pragma unroll
                for (int j = 0; j < 6; j++) v -= AB[k * 48 + j * 8 + 6 + c] * nu[k * 6 + j];
                ru[i] = v; rmax = fmax(rmax, fabs(v));
                gup[i] = v - fhp;                             // gu' = ru - Fu' h_u with the predictor's h (= mu wherever the weight is not capped)
                const double ml = m[i], ms = m[6 * N + i];
                const double vs = a_s * s[i] + c_s - ml - ms;
                rs[i] = vs; rmax = fmax(rmax, fabs(vs));
                const double d_ = frcp(a_s + th[i] + th[6 * N + i]);
                rDs[i] = d_; kap[i] = th[i] * (a_s + th[6 * N + i]) * d_;
                const double hl = h[i], e_ = -(vs + hl + h[6 * N + i]);
                ee[i] = e_; eta[i] = hl + th[i] * e_ * d_;
            }
            if constexpr (term) {
                FOR_WAVE(c, S, 2) {                           // rows of lambda (wave 2)
                    double v = Qsel[c] - m[8 * N + c] + eta_m;
#pragma unroll
                    for (int j = 0; j < 6; j++) v = fma(SS[j * S + c], T2p[j] * sT[j], v);
                    rl[c] = v; rmax = fmax(rmax, fabs(v)); lsum += lam[c];
                }
            }
            FOR_WAVE2(i, 6 * N, 3, 1) {                       // dynamics residual (monitoring only; wave 3, the last few on wave 1)
                const int k = i / 6, c = i % 6;
                double v = x[(k + 1) * 6 + c] - C[i] - AB[k * 48 + c * 8 + 6] * u[k * 2] - AB[k * 48 + c * 8 + 7] * u[k * 2 + 1];
#pragma unroll
                for (int j = 0; j < 6; j++) v -= AB[k * 48 + c * 8 + j] * x[k * 6 + j];
                remax = fmax(remax, fabs(v));
            }
            double gsum = gsum_c;
            if (wave == 1) gsum += gs0[lane];                // (written by wave 0 together with the step, a barrier ago)
            red_put(0, wsum<BCF>(gsum)); red_put(1, wmax<BCF>(rmax)); red_put(2, wsum<BCF>(lsum)); red_put(3, wmax<BCF>(remax));
            TSMW(111);
        }
        __syncthreads();                                     // B1
        {
            double gs = 0.0, rm = 0.0, ls = 0.0, rem = 0.0;
            for (int w = 1; w < NW; w++) { gs += red[w]; rm = fmax(rm, red[4 + w]); ls += red[8 + w]; rem = fmax(rem, red[12 + w]); }
            gap = SWEEP_BF<N> ? gs * (1.0 / (double)M) : gs / (double)M; rdn = rm; ren = rem; lsum_all = ls;   // (an IEEE division is ~30 instructions on every wave; four per Newton step went)
        }
        const double re_sum = term ? lsum_all - 1.0 : 0.0;
        ren = fmax(ren, fabs(re_sum));
        TRACE3(tid == 0, 0, gap, rdn, ren);
        const double step_last = it > 0 ? red_max(6) : INFINITY;         // (written in the step phase, two barriers ago)
        if (gap < p.tol_gap && rdn < p.tol_res * qscale && ren < p.tol_res && step_bound_ok(step_last, step_pp)) { converged = 1; break; }
        if (gap_prev >= 0.0) sep = gap > LMPC_SEP_THRESHOLD * gap_prev;
        gap_prev = gap; step_pp = step_last;
        if (it == p.max_iter) break;
        if (!(gap == gap) || !(rdn == rdn)) { if (tid == 0) atomicOr(&st_sh, LMPC_ST_NUMERIC); break; }
        TSMW(12);
        // ---- phase 2: a pipeline of N + 1 steps, each closed by a work-group barrier.  Step s: wave 0 factorises stage N-1-s (Riccati
        //      recursion on the matrix cores); one step behind, wave 1 extends the predictor's backward sweep  p_k = Phi_k' p_{k+1} + gamma_k
        //      to the stage just factorised and wave FB_WAVE computes that stage's feed-forward term phi_k (it needs the stage's factors and
        //      p_{k+1}, both complete since the previous barrier).  Step 0 of wave 1 is the terminal costate p_N; the last step of wave 0
        //      is the read-back of its sweep operands.  After the last barrier the forward sweep can start at once. ------------------------
        if (w0) {
            if constexpr (NW == 2) {
                // two waves per SIMD (<= 256 registers): the ~40 registers of per-lane recursion constants are rebuilt here every iteration
                // instead of being carried through the whole loop by both waves (the opaque copy of `lane` keeps the compiler from hoisting it;
                // with four waves -- 512 registers per wave -- carrying them measured the same as rebuilding them)
                int l2 = lane; asm volatile("" : "+v"(l2));
                const ricc_consts rc2 = ricc_setup(l2, Q2, Fx, R2, dR2, Fu);
                numeric_bad |= ricc_factor<N, term, true>(rc2, AB, kap, th, Qf2, PiT, Phi, PiAll, Mi, (double *)nullptr, dump_sh + lane);
            } else numeric_bad |= ricc_factor<N, term, true>(rc, AB, kap, th, Qf2, PiT, Phi, PiAll, Mi, (double *)nullptr, dump_sh + lane);
            if (numeric_bad && lane == 0) bad_sh = 1;
#pragma unroll
            for (int k = 0; k < N; k++) ph[k] = (k & 1) ? Phi[k * 64 + lc * 8 + lg] : Phi[k * 64 + lg * 8 + lc];
            TSMW(19);
            STEP_BARRIER();                                  // step N
        } else {
            double pv = 0.0;
            if (wave == 1) {                                 // step 0: terminal costate p_N
                if constexpr (term) {
#pragma unroll
                    for (int ch = 0; ch < CH; ch++) {
                        const int col = lane + WAVE * ch;
                        ct[col] = col < S ? (rl[col] + h[8 * N + col]) * rsq[col] : 0.0;
                    }
                    WSYNC();
                    double acc = 0.0;                               // M c~ : lane (j, part) adds every 8th of the 64 CH columns
                    {
                        double mq[8 * CH], cq[8 * CH];
#pragma unroll
                        for (int q = 0; q < 8 * CH; q++) { mq[q] = Mt[lc * 8 + (lg < 7 ? lg : 0) + 64 * q]; cq[q] = ct[lc + 8 * q]; }
                        LDS_GROUP();
#pragma unroll
                        for (int q = 0; q < 8 * CH; q++) acc = fma(mq[q], cq[q], acc);
                        acc = lg < 7 ? acc : 0.0;
                    }
                    acc = sum_over_c(acc);
                    if (lg < 7 && lc == 0) McL[lg] = acc;
                    WSYNC();
                }
                double v = 0.0;
                if constexpr (term) {
                    const double yv = ri_t_times(Ri, McL, lg, lc);
                    if (lc == 0 && lg < 7) { y7[lg] = yv; z7[lg] = fma(Ri[6 * 7 + lg], -re_sum, yv); }
                    WSYNC();
                    v = ri_times(Ri, z7, lg, lc);
                }
                if (lc == 0) pst[N * 8 + lg] = lg < 6 ? rx[N * 6 + lg] + v : 0.0;
                WSYNC();
                pv = ((N - 1) & 1) ? pst[N * 8 + lc] : pst[N * 8 + lg];
            }
            STEP_BARRIER();                                  // step 0
#pragma unroll
            for (int k = N - 1; k >= 0; k--) {               // step N - k: stage k was factorised during the previous step
                if (NW > 2 && wave == FB_WAVE) {          // (two waves: wave 1 runs the sweep only -- with both followers it, not the recursion, set the pace of the pipeline; phi in bulk below)
                    // phi_k = [-B k0 ; -k0],  k0_k = M_uu^-1 (gu' + [B; I]' p_{k+1}): lane c of every group of 8 holds row c of [B_k; I] and p_{k+1}[c]
                    // (per-lane loads, no chain of broadcast reads), the two 8-term sums are DPP reductions inside the group
                    const double bq0 = lc < 6 ? AB[k * 48 + lc * 8 + 6] : (lc == 6 ? 1.0 : 0.0), bq1 = lc < 6 ? AB[k * 48 + lc * 8 + 7] : (lc == 7 ? 1.0 : 0.0);
                    const double pq = pst[(k + 1) * 8 + lc], gq = lc >= 6 ? gup[2 * k + lc - 6] : 0.0;
                    const double m00 = Mi[k * 4], m01 = Mi[k * 4 + 1], m10 = Mi[k * 4 + 2], m11 = Mi[k * 4 + 3];
                    const double w0_ = sum_over_c<BCF>(fma(bq0, pq, lc == 6 ? gq : 0.0)), w1_ = sum_over_c<BCF>(fma(bq1, pq, lc == 7 ? gq : 0.0));
                    const double k00 = m00 * w0_ + m01 * w1_, k01 = m10 * w0_ + m11 * w1_;
                    if (lane < 8) phi[k * 8 + lc] = -(bq0 * k00 + bq1 * k01);
                }
                if (wave == 1) {
                    // gamma_k = [gx' ; 0] + Phi_k[6:8,:]' gu' in the sweep's register layout, then p_k
                    const int ix = (k & 1) ? lg : lc;
                    double g0 = 0.0;
                    if (ix < 6) { g0 = rx[k * 6 + ix]; g0 -= Fx[ix] * eta[2 * k] + Fx[6 + ix] * eta[2 * k + 1]; }
                    const double phk = (k & 1) ? Phi[k * 64 + lc * 8 + lg] : Phi[k * 64 + lg * 8 + lc];
                    const double gk = fma(Phi[k * 64 + 56 + ix], gup[2 * k + 1], fma(Phi[k * 64 + 48 + ix], gup[2 * k], g0));
                    double pr = phk * pv;
                    if (k & 1) { pr = sum_over_c(pr); pv = pr + gk; if (lc == 0) pst[k * 8 + lg] = pv; }
                    else { pr = sum_over_g(pr); pv = pr + gk; if (lg == 0) pst[k * 8 + lc] = pv; }
                }
                TSMW(200 + k);
                STEP_BARRIER();
            }
        }
        // (the last step's barrier is the hand-over to the forward sweep: LDS writes of a wave that precede its barrier are in place)
        if constexpr (NW == 2) {
            // Two waves per QP: the feed-forward terms phi_k of ALL stages now, eight stages per pass (group lg takes stage lg of the pass, lane lc row lc of [B_k; I]),
            // the passes dealt to the two waves in turn.  As a follower inside the pipeline phi cost wave 1 ~450 cycles per stage on top of its sweep step, and wave 0
            // waited for it at every stage barrier (1 200 instead of 950 cycles per stage, profiles/r5zz_phase_timing_mw2.txt); in bulk it is one pass per wave.
#pragma unroll
            for (int k0_ = 0; k0_ < N; k0_ += 16) {
                const int k = k0_ + 8 * wave + lg; const bool on = k < N; const int kk = on ? k : 0;
                const double bq0 = lc < 6 ? AB[kk * 48 + lc * 8 + 6] : (lc == 6 ? 1.0 : 0.0), bq1 = lc < 6 ? AB[kk * 48 + lc * 8 + 7] : (lc == 7 ? 1.0 : 0.0);
                const double pq = pst[(kk + 1) * 8 + lc], gq = lc >= 6 ? gup[2 * kk + lc - 6] : 0.0;
                const double m00 = Mi[kk * 4], m01 = Mi[kk * 4 + 1], m10 = Mi[kk * 4 + 2], m11 = Mi[kk * 4 + 3];
                LDS_GROUP();
                const double w0_ = sum_over_c<BCF>(fma(bq0, pq, lc == 6 ? gq : 0.0)), w1_ = sum_over_c<BCF>(fma(bq1, pq, lc == 7 ? gq : 0.0));
                const double k00 = m00 * w0_ + m01 * w1_, k01 = m10 * w0_ + m11 * w1_;
                if (on) phi[k * 8 + lc] = -(bq0 * k00 + bq1 * k01);
            }
            STEP_BARRIER();
        }
        if (bad_sh) { if (tid == 0) atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_NUMERIC); break; }   // see lmpc_solve_kernel
        TSMW(13);
        // ---- predictor (affine scaling) direction, rest: forward sweep on wave 0, then the slack and terminal steps ----------------
        double xiN = 0.0;
        if (w0) {
            double fm[N];
#pragma unroll
            for (int k = 0; k < N; k++) fm[k] = (k & 1) ? phi[k * 8 + lc] : phi[k * 8 + lg];
            double xi = 0.0;
            if constexpr (SWEEP_BF<N>) {
                const sweep_dst wd = fwd_sweep_dst<N>(dx, du, phi, lane, lg, lc); lds_f64 *wE = (lds_f64 *)wd.wE, *wO = (lds_f64 *)wd.wO;
                LDS_GROUP();
#pragma unroll
                for (int k = 0; k < N; k++) {
                    double pr = ph[k] * xi;
                    if (k & 1) pr = sum_over_g<true>(pr); else pr = sum_over_c<true>(pr);
                    xi = pr + fm[k];
                    if (k & 1) { *wO = xi; wO += wd.dO; } else { *wE = xi; wE += wd.dE; }
                    SWEEP_PIN(wE, wO);
                }
            } else {
                LDS_GROUP();
#pragma unroll
                for (int k = 0; k < N; k++) {
                    double pr = ph[k] * xi;
                    int idx;
                    if (k & 1) { pr = sum_over_g(pr); idx = lc; } else { pr = sum_over_c(pr); idx = lg; }
                    xi = pr + fm[k];
                    const bool wr = (k & 1) ? (lg == 0) : (lc == 0);
                    if (wr) { if (idx < 6) dx[(k + 1) * 6 + idx] = xi; else du[k * 2 + (idx - 6)] = xi; }
                }
            }
            xiN = xi;
        }
        __syncthreads();
        TSMW(33);
        FOR_OFF(i, 2 * N, O1) {
            const int k = i >> 1, j = i & 1; double f = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], dx[k * 6 + c], f);
            ds[i] = (th[i] * f + ee[i]) * rDs[i];
        }
        if constexpr (term) {
            if (w0) {
                double wq[7];
                if constexpr ((N - 1) & 1) term_omega(Ri, y7[lg < 7 ? lg : 0], xiN, re_sum, lg, lc, wq, qx_late);   // (y7: wave 1's, from step 0 of the pipeline)
                else {   // z7 = Ri' d7 + y7, d7 = (dx_N ; -re_sum);  omega' = Ri z7
                    if (lane < 7) w7[lane] = lane < 6 ? dx[N * 6 + lane] : -re_sum;           // d7 (w7 is free until omega' is written)
                    WSYNC();
                    const double zv = ri_t_times(Ri, w7, lg, lc) + y7[lg < 7 ? lg : 0];
                    WSYNC();
                    if (lc == 0 && lg < 7) z7[lg] = zv;
                    WSYNC();
                    if (!qx_late) {                                  // omega' = Ri z7 (late iterations: z7 itself, the lanes' mcol holds Q = M' R^-1 then, see term_omega)
                        const double wv = ri_times(Ri, z7, lg, lc);
                        WSYNC();
                        if (lc == 0 && lg < 7) z7[lg] = wv;
                        WSYNC();
                    }
#pragma unroll
                    for (int j = 0; j < 7; j++) wq[j] = z7[j];
                }
#pragma unroll
                for (int ch = 0; ch < CH; ch++) {
                    const int col = lane + WAVE * ch;
                    double v = -ct[col];                        // v = -c~ + Q z7
                    const double rq = rsq[col];
                    LDS_GROUP();
#pragma unroll
                    for (int j = 0; j < 7; j++) v = fma(mcol[ch][j], wq[j], v);
                    if (col < S) dl[col] = v * rq;
                }
            }
        }
        __syncthreads();
        TSMW(14);
        // One pass over the rows, one reduction round (round 6; two before): besides the step lengths of the affine direction every thread sums  dt mu,  t dmu  and  dt dmu
        // over its rows -- the complementarity gap after the affine step is  gap + (a_p sum dt mu + a_d sum t dmu + a_p a_d sum dt dmu) / M  for whatever a_p, a_d the
        // reduction then yields, so sigma needs no second pass (and no second barrier) behind the step lengths.  (Slots 0 and 1 are phase 1's, dead since barrier B1.)
        double apmax = 1.0, admax = 1.0, s_dtm = 0.0, s_tdm = 0.0, s_dd = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = slot_row(j);
            if (r >= 0) {
                const double dta = -rowF(r, dx, du, ds, dl), mr = m[r];
                const double dma = -h[r] - th[r] * dta;
                dt_r[j] = dta; tp_r[j] = dta * dma;
                s_dtm = fma(dta, mr, s_dtm); s_tdm = fma(t_r[j], dma, s_tdm); s_dd += tp_r[j];
                if (dta < 0.0) apmax = fmin(apmax, -t_r[j] * frcp(dta));
                if (dma < 0.0) admax = fmin(admax, -mr * frcp(dma));
            }
        }
        red_put(4, wmin<BCF>(apmax)); red_put(5, wmin<BCF>(admax)); red_put(6, wsum<BCF>(s_dtm)); red_put(0, wsum<BCF>(s_tdm)); red_put(1, wsum<BCF>(s_dd));
        __syncthreads();
        apmax = red_min(4); admax = red_min(5);
        if (!sep) { apmax = fmin(apmax, admax); admax = apmax; }
        const double gaff = gap + (apmax * red_sum(6) + admax * red_sum(0) + apmax * admax * red_sum(1)) * (SWEEP_BF<N> ? (1.0 / (double)M) : 1.0 / (double)M);
        double sig = SWEEP_BF<N> ? gaff * frcp(gap) : gaff / gap; sig = centring_sigma(sig);
        const double tgt = fmax(sig * gap, 0.01 * p.tol_gap);
        TSMW(15);
        // ---- corrector ---------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = slot_row(j); if (r >= 0) h[r] = (fma(t_r[j], m[r], LMPC_SO_W * tp_r[j]) - tgt) * rt_r[j]; }
        __syncthreads();
        kkt_solve(re_sum);
        TSMW(16);
        double apx = INFINITY, adx = INFINITY, dsum = 0.0;   // dsum: this thread's lambda rows' share of the step of the multiplier of sum(lambda) = 1
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = slot_row(j);
            if (r >= 0) {
                const double dtt = -rowF(r, dx, du, ds, dl), mr = m[r];
                const double dmm = -h[r] - th[r] * dtt;
                dm[r] = dmm; dt_r[j] = dtt;
                if (dtt < 0.0) apx = fmin(apx, -t_r[j] * frcp(dtt));
                if (dmm < 0.0) adx = fmin(adx, -mr * frcp(dmm));
                if constexpr (term) { if (r >= 8 * N) dsum += dmm - rl[r - 8 * N]; }
            }
        }
        red_put(7, wmin<BCF>(apx)); red_put(8, wmin<BCF>(adx));
        if constexpr (term) red_put(9, wsum<BCF>(dsum));
        if constexpr (term) { if (wave == NW - 1) ss_times<S>(SS, dl, dx + N * 6, w7, lane); }          // d s_T
        __syncthreads();
        const double frac = step_fraction(sig, gap);
        double al = fmin(1.0, frac * red_min(7)), ald = fmin(1.0, frac * red_min(8));
        if (!sep) { al = fmin(al, ald); ald = al; }
        TSMW(17);
        TRACE3(tid == 0, 3, sig, al, ald);
        // ---- step.  The multipliers of the equality rows (costates nu_k = -(Pi_k xi_k + p_k)_x for all stages at once) are formed and
        //      applied in the same pass as the primal step and the inequality rows: nothing here reads what another thread writes. ----------
        FOR_WAVE2(i, 6 * N, 3, 1) {                          // (wave 3 and a few lanes of wave 1: wave 0 has the next terminal factor to start, see below)
            const int k = i / 6 + 1, c = i % 6;
            double g;
            if (k == N) {
                g = rx[N * 6 + c];
#pragma unroll
                for (int j = 0; j < 6; j++) g = fma(Qf2[c * 6 + j], dx[N * 6 + j], g);
                if constexpr (term) g -= T2p[c] * w7[c];
            } else {
                g = pst[k * 8 + c];
#pragma unroll
                for (int j = 0; j < 6; j++) g = fma(PiAll[k * 64 + c * 8 + j], dx[k * 6 + j], g);
                g = fma(PiAll[k * 64 + c * 8 + 6], du[(k - 1) * 2], g);
                g = fma(PiAll[k * 64 + c * 8 + 7], du[(k - 1) * 2 + 1], g);
            }
            nu[i] = fma(ald, -g, nu[i]);
        }
        // multiplier of sum(lambda) = 1: mean over the lambda rows of  -rl_c + dmu_c - SS_c' T ds_T;  the rows' owners summed the first two
        // terms above, the third is sum_j T_j ds_T[j] (sum_c SS[j][c]) with the row sums of SS formed once (every thread: same value)
        double deta = 0.0;
        if constexpr (term) {
            if (wave == WCL(2)) {                              // (eta_m is read by the lambda rows of phase 1 only: the wave that owns them keeps it -- not the critical wave)
                deta = red_sum(9);
#pragma unroll
                for (int j = 0; j < 6; j++) deta -= T2p[j] * w7[j] * ss_rowsum[j];
                if constexpr (SWEEP_BF<N>) deta *= 1.0 / (double)S; else deta /= (double)S;
            }
        }
        {                                                      // (the step, and the length of its (x, u) part for step_bound_ok: slot 6 is free from the predictor's
            double smax = 0.0;                                 //  sigma to the next iteration's, and is read at the top of that iteration, behind this phase's barrier)
            FOR_WAVE2(i, 6 * (N + 1), 1, 2) { smax = fmax(smax, fabs(al * dx[i])); x[i] = fma(al, dx[i], x[i]); }
            FOR_WAVE(i, 2 * N, 2) { smax = fmax(smax, fabs(al * du[i])); u[i] = fma(al, du[i], u[i]); s[i] = fma(al, ds[i], s[i]); }
            red_put(6, wmax<BCF>(smax));
        }
        if constexpr (term) { FOR_WAVE(c, S, 2) lam[c] = fma(al, dl[c], lam[c]); }
        gsum_c = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; j++) {                         // inequality rows: slack, multiplier, barrier weight, gap share
            const int r = slot_row(j);
            if (r >= 0) {
                const double tt = fma(al, dt_r[j], t_r[j]), mm = fma(ald, dm[r], m[r]);
                const double rt = barrier_rt(tt, mm), thv = mm * rt;
                t_r[j] = tt; m[r] = mm; rt_r[j] = rt; th[r] = thv; gsum_c = fma(tt, mm, gsum_c);
                h[r] = tt * thv;                                // next predictor's right-hand side (dm[r], which shares the place, was consumed one line up)
            }
        }
        if (w0) gs0[lane] = gsum_c;
        if constexpr (term) { if (tid >= NT - 6) sT[tid - (NT - 6)] = fma(al, w7[tid - (NT - 6)], sT[tid - (NT - 6)]); }      // (the last wave's spare lanes, not wave 0's)
        if constexpr (TEARLY) {
            // the next iteration's terminal factor, part A: wave 0 owns the lambda rows, so the weights it has just written are all it needs
            // (its own LDS writes are visible to it after the wait); the Gram matrix is in Wl when the barrier below falls
            if (w0) { WSYNC(); TSMW(180); term_partA(); }
        }
        __syncthreads();
        if constexpr (term) eta_m = fma(ald, deta, eta_m);
        TSMW(18);
    }
    if (!converged && tid == 0 && !(st_sh & (LMPC_ST_NUMERIC | LMPC_ST_INEXACT)))
        atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_MAXITER);
    if (!p.slacks && w0) {                                   // hard lane rows exceeded: the hard problem is infeasible (see lmpc_solve_kernel)
        double smax = 0.0;
        for (int i = lane; i < 2 * N; i += WAVE) smax = fmax(smax, s[i]);
        smax = wmax(smax);
        if (smax > 1e-8 && lane == 0) atomicOr(&st_sh, LMPC_ST_INFEASIBLE);
    }
    __syncthreads();

    TSMW(20); TSMW(21);
    // ---- unpackSolution (:364-379) and feasibleStateInput (:382-384) -------------------------------------
    for (int i = tid; i < 6 * (N + 1); i += NT) io.xPred[(size_t)b * 6 * (N + 1) + i] = x[i];
    for (int i = tid; i < 2 * N; i += NT) { io.uPred[(size_t)b * 2 * N + i] = u[i]; if (io.slack) io.slack[(size_t)b * 2 * N + i] = s[i]; }
    if (io.mu) for (int r = tid; r < M; r += NT) io.mu[(size_t)b * M + r] = m[r];
    if constexpr (term) {
        if (io.lambda) for (int c = tid; c < S; c += NT) io.lambda[(size_t)b * S + c] = lam[c];
        if (tid < 6 && io.sTerm) {
            double v = -x[N * 6 + tid];
            for (int c = 0; c < S; c++) v = fma(SS[tid * S + c], lam[c], v);
            io.sTerm[(size_t)b * 6 + tid] = v;
        }
        if (w0 && (io.mode & 1) && (io.ztNext || io.ztuNext)) {
            double acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.0;
            for (int c = lane; c < S; c += WAVE) {
                const int l = c / p.ppl, cc = c % p.ppl;
                const double *base = p.sstore + (size_t)p.sslot[l] * LMPC_COLS * p.lap_stride;
                int r1 = sel_start[l] + cc + 1; r1 = r1 > p.sslen[l] - 1 ? p.sslen[l] - 1 : r1;
                const double lv = lam[c];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = fma(base[j * p.lap_stride + r1], lv, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = wsum(acc[j]);
            if (lane < 6 && io.ztNext) { double v = acc[0];
#pragma unroll
                for (int j = 1; j < 6; j++) if (lane == j) v = acc[j];
                io.ztNext[(size_t)b * 6 + lane] = v; }
            if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = lane == 0 ? acc[6] : acc[7];
        }
    } else {
        if (tid < 6 && io.ztNext) io.ztNext[(size_t)b * 6 + tid] = x[N * 6 + tid];
        if (tid < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + tid] = u[(N - 1) * 2 + tid];
    }
    if (tid == 0) {
        io.status[b] = st_sh; io.iters[b] = it; flag_retry(io, st_sh);
        if (io.resid) { io.resid[(size_t)b * 3] = gap; io.resid[(size_t)b * 3 + 1] = rdn; io.resid[(size_t)b * 3 + 2] = ren; }
    }
#undef FOR_OFF
#undef FOR_HELP
#undef FOR_WAVE
#undef FOR_WAVE2
#undef WCL
}
