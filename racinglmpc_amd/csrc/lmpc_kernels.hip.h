// racinglmpc_amd/csrc/lmpc_kernels.hip.h -- gfx950 (MI355X) device code of the LMPC hot path.
//
//   lmpc_regress_kernel : K1  LTV model regression + linearisation
//                         (reference PredictiveModel.py:48-197, Track.py:292-310)
//   lmpc_solve_kernel   : K2+K3  safe-set selection (PredictiveControllers.py:386-412, 478-514),
//                         structured QP data (never a dense/CSC matrix) and the QP solve
//                         (replaces osqp_solve_qp, :259-283) + unpack (:364-384)
//   lmpc_assemble_kernel: explicit reference-form QP matrices, parity checks only (:166-257, :340-362)
//
// Execution model: K1 runs one 8-wave work-group per problem (a wave per lap); K2+K3 run ONE 64-lane wavefront per QP
// (or four, lmpc_solve_mw.hip.h), all per-item state in LDS; cross-lane sums are wavefront
// reductions; sequential recursions (Riccati sweeps) are lane-parallel inside a stage.
// FP64 throughout (the reference is FP64 end to end; regression normal matrices have cond ~1e5..1e8).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/lmpc_hip.h"

#define WAVE 64
#define LMPC_VARIANT_ABI_REV 7          // bumped whenever lmpc_dev_params / lmpc_solve_io / the variant table change; the value a variant library is asked for
                                        // (LMPC_VARIANT_ABI, lmpc_variant.hip.h) also folds in the three struct sizes, so that layout drift cannot pass on the number alone
#define LMPC_COLS 9                 // lap-store columns: x0..x5, u0, u1, Qfun

// Branch-free LDS stores in the register sweeps and the Riccati recursion (sweep_dst below): on for horizons up to 24.
// (horizons up to 24.  At N = 40 the one-wave kernel sits at 256 VGPRs + ~190 AGPRs and its code generation is brittle: with the branch-free stores
//  the solve kernel measured 13 % SLOWER there (1.10 -> 1.24 ms at batch 1024; neither the pinned pointers nor the recursion's dump stores alone
//  explain it) -- long horizons keep the predicated stores, text unchanged.)
template <int N> constexpr bool SWEEP_BF = N <= 24;

struct lmpc_dev_params {
    int N, S, L, ppl;               // horizon, safe-set columns, laps per solve, points per lap (S/L)
    int trToUse, maxNumPoint;
    double h, lamb, dt, scaling[5];
    double Q2[36], R2[4], Qf2[36], dR2[2], a_s, c_s, T2[6], xRef[6];   // 2Q, 2R, 2Qf, 2dR, 2Qslack[0], Qslack[1], 2 diag(QtermSlack)
    double Fx[12], bx[2], Fu[8], bu[4];
    double track[LMPC_MAX_TRACK_ROWS * 6]; int track_rows; double TL;
    double tol_gap, tol_res, reg; int max_iter;
    int slacks;                     // MPCParams.slacks; 0: the explicit matrices have no slack columns / rows; the solve kernels see a_s = 2e12, c_s = 0 (see fill_params)
    int lap_stride;                 // rows per column (max_lap_len)
    const double *mstore; int mslot[LMPC_MAX_USED_LAPS]; int mlen[LMPC_MAX_USED_LAPS];
    const unsigned *mquant; const double *mqpar; int mq_chunks;   // K1 prefilter image of the model store (16-bit fixed point, three packed words per row) and its per-chunk (lo[5], scale)
    const double *sstore; int sslot[LMPC_MAX_USED_LAPS]; int sslen[LMPC_MAX_USED_LAPS]; int sslapid[LMPC_MAX_USED_LAPS];
    int cur_it;                     // LMPC.it (number of laps in the safe set)
};

// Developer flavour -DLMPC_EXEC_AUDIT (racinglmpc_amd.build.build_flavour("audit", ...)): every cross-lane primitive of the solve kernels counts its calls and
// the calls that found an incomplete EXEC mask (a DPP / permlane / bpermute / MFMA under a partial mask reads stale or zero lanes -- the iteration would still
// converge, inexact Newton is self-correcting, so certificates alone cannot exclude it).  lmpc_debug_exec_audit reads the counters.  Sites:
//   0 wave_allreduce (wsum / wmax / wmin)   1 sum_over_c   2 sum_over_g   3 lane_gather   4 ricc_factor stage (DPP block moves, MFMA, readlane, swaps)
//   5 Gram matrix of the terminal factor (MFMA)   6 (reserved)   7 regression kernel (prefix scan, row ranking)
#define LMPC_AUDIT_SITES 8
#ifdef LMPC_EXEC_AUDIT
static __device__ unsigned long long g_exec_audit[2 * LMPC_AUDIT_SITES];      // [site]: calls under a partial mask, [8 + site]: calls
__device__ __forceinline__ void exec_audit(int site) {
    const unsigned long long e_ = __builtin_amdgcn_read_exec();
    if ((int)(threadIdx.x & 63) == __ffsll((long long)e_) - 1) { atomicAdd(&g_exec_audit[LMPC_AUDIT_SITES + site], 1ull); if (e_ != ~0ull) atomicAdd(&g_exec_audit[site], 1ull); }
}
#define EXEC_AUDIT(site) exec_audit(site)
#else
#define EXEC_AUDIT(site) do { } while (0)
#endif

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}
// lexicographic (value, index) minimum across the wave; all lanes receive the result
__device__ __forceinline__ void wave_argmin(double &v, int &i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, WAVE); int oi = __shfl_xor(i, o, WAVE);
        bool take = (ov < v) || (ov == v && oi < i);
        v = take ? ov : v; i = take ? oi : i;
    }
}

// Map.curvature, Track.py:292-310.  Returns 0 and sets *bad when s lies on no segment (the reference raises).
__device__ __forceinline__ double track_curvature(const lmpc_dev_params &p, double s, int *bad) {
    const double TL = p.TL;
    // `while s > TrackLength: s = s - TrackLength` of the reference, bounded: an infinite or absurd s (a diverged rollout) must not hang the
    // GPU -- beyond 64 laps it is reported as "on no segment", which is what the reference's own loop could never return from either
    for (int lap = 0; lap < 64 && s > TL; lap++) s = s - TL;
    if (!(s <= TL)) { *bad = 1; return 0.0; }
    for (int i = 0; i < p.track_rows; i++) {
        const double c0 = p.track[i * 6 + 3], len = p.track[i * 6 + 4];
        if (s >= c0 && s < c0 + len) return p.track[i * 6 + 5];
    }
    *bad = 1;
    return 0.0;
}


// =====================================================================================================
// K2 + K3: safe-set selection + structured primal-dual interior-point QP solve.  One wave per QP.
// Templated on the horizon N and the number of safe-set columns S (0 = plain MPC, no terminal set) so that
// every LDS offset and trip count is a compile-time constant.
// =====================================================================================================
struct lmpc_solve_io {
    // mode bit 0: select the safe set on device (else read ssSel/qSel); bit 1: run the QP solve; bit 2 (one-wave kernel only): fused step --
    // the wave first runs the regression of its own QP from xLin / uLin (rows of N+1 / N points), [A_k | B_k] and C_k never leave LDS
    // (Aout / Bout / Cout, if given, receive a copy)
    int mode;
    const double *A, *Bm, *C, *x0, *uOld, *ssSelIn, *qSelIn;
    const double *xLin, *uLin; double *Aout, *Bout, *Cout;
    const double *zt, *xPredPrev; const int *hasPred, *timeStep;
    const int *rstatus;       // optional per-point status of the regression kernel (B x N): OR-ed into status[b] (the reference raises there)
    double *xPred, *uPred, *slack, *lambda, *sTerm, *mu, *ztNext, *ztuNext;
    double *ssSelOut, *qSelOut, *succOut, *succUOut, *ztUsed, *resid;
    int *selStartOut;         // optional: first row of the 13-row window per selected lap (B x numSS_it)
    int *status, *iters;
    long long *tbuf;          // optional cycle stamps of problem 0 (builds with -DLMPC_TIMING only)
    double *abPack;           // one-wave kernel, long horizons (ABG): global scratch for [A_k | B_k] in the kernel's 6 x 8 layout, 48 N doubles per problem
    int *retry_flag;          // optional (host-mapped): a problem that ends at the iteration limit / breaks down writes retry_epoch here (atomic max, system
    int retry_epoch;          // scope), so that the host launches the retry pass only when one is needed (lmpc_capi.hip: resolve_retries)
};
__device__ __forceinline__ void flag_retry(const lmpc_solve_io &io, int st) {
    if (io.retry_flag && (st & (LMPC_ST_MAXITER | LMPC_ST_NUMERIC))) __hip_atomic_fetch_max(io.retry_flag, io.retry_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Developer flavour -DLMPC_TRACE: per-iteration side channel of the solve kernels -- row `it` of problem b in io.tbuf (taken as doubles, LMPC_TRACE_ROWS x 6 per
// problem) receives (gap, r_d, |sum(lambda) - 1|) when the residuals are known and (sigma, alpha_p, alpha_d) when the step is taken: the columns tests/ipm_model.py
// records (ipm_solve(trace=...)), so that kernel and model can be compared iteration by iteration (tools/n40_model.py compare).  lmpc_debug_set_trace.
#define LMPC_TRACE_ROWS 48
#ifdef LMPC_TRACE
#define TRACE3(leader, c0, v0, v1, v2) do { if (io.tbuf && (leader) && it < LMPC_TRACE_ROWS) { double *tr_ = (double *)io.tbuf + ((size_t)b * LMPC_TRACE_ROWS + it) * 6 + (c0); \
                                                tr_[0] = (v0); tr_[1] = (v1); tr_[2] = (v2); } } while (0)
#else
#define TRACE3(leader, c0, v0, v1, v2) do { } while (0)
#endif
#ifdef LMPC_TIMING
#define TSTAMP(id) do { if (io.tbuf && b == 0 && lane == 0 && tcnt < 4000) { io.tbuf[2 * tcnt] = (id); io.tbuf[2 * tcnt + 1] = (long long)__builtin_readcyclecounter(); tcnt++; } } while (0)
#else
#define TSTAMP(id) do { } while (0)
#endif

// ---- cross-lane primitives (gfx950): DPP inside a row of 16 lanes, v_permlane16/32_swap across rows ----
// dpp_mov: old = src, bound_ctrl off -- a lane whose source lane is switched off by EXEC keeps its own value (the plant's lane pairs, the regression's row ranking).
// dpp_mv<.., true>: bound_ctrl with old = 0, the form the reductions of the solve kernels use (three instructions per reduction step instead of five); identical
// under a full EXEC mask, which is what every reduction site of those kernels runs under -- counted, not assumed: the EXEC audit flavour, lmpc_debug_exec_audit.
// (Round 4 saw the N = 40 kernel go from 11.0 to 12.4 iterations "with" this form and fenced the long horizons off.  Round 5 found the cause, and it is not the
//  move: a compiler fault that depends on register allocation -- live-range copies placed ahead of a flow block's EXEC restore, racinglmpc_amd/isa_check.py -- which
//  any edit of that kernel can switch on or off; with a build the check accepts, both forms give the same iterates on all 1024 problems.)
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
#ifdef LMPC_DPP_BC                      // (developer builds: the bound_ctrl form everywhere, long horizons included -- tools/n40_experiments.sh)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
#else
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
#endif
    return __hiloint2double(hi, lo);
}
// returns (a, b): a = v with odd rows replaced by the partner's even rows, b = the complementary half
__device__ __forceinline__ void swap16(double v, double &a, double &b) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]); b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap32(double v, double &a, double &b) {
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)r1[0], (int)r0[0]); b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// DPP move of src into the banks (groups of 4 lanes in each row of 16) selected by BANKS; the other lanes keep `keep`
template <int CTRL, int BANKS> __device__ __forceinline__ double dpp_blk(double keep, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(src), CTRL, 0xf, BANKS, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int byte_idx) {   // v from lane byte_idx / 4 (ds_bpermute: LDS crossbar, no memory)
    EXEC_AUDIT(3);
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_idx, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(byte_idx, __double2loint(v)));
}
__device__ __forceinline__ double rdlane(double v, int src) {       // wave-uniform copy of lane src's value
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
#define DPP_QP_X1 0xB1          // quad_perm [1,0,3,2]
#define DPP_QP_X2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMin { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
// The same move with bound_ctrl and old = 0: identical under a full EXEC mask (every lane has a source), and the compiler no longer needs the
// v_mov_b32 that preloads the destination -- three instructions per reduction step instead of five, on a wave that is bound by instruction issue.
// Used where the EXEC mask is full by construction (every reduction of the solve kernels: see dpp_mov).
template <int CTRL, bool BC> __device__ __forceinline__ double dpp_mv(double v) {
    if constexpr (BC) return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
    else return dpp_mov<CTRL>(v);
}
#define LMPC_BC_DEFAULT true          // (bound_ctrl form in every reduction: +2 % at batch 4096 in the one-wave kernel, neutral in the others -- profiles/r5g_ab.txt)
template <class Op, bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double wave_allreduce(double v, Op op) {
    EXEC_AUDIT(0);
    v = op(v, dpp_mv<DPP_QP_X1, BC>(v));
    v = op(v, dpp_mv<DPP_QP_X2, BC>(v));
    v = op(v, dpp_mv<DPP_HALF_MIRROR, BC>(v));
    v = op(v, dpp_mv<DPP_MIRROR, BC>(v));
    double a, b;
    swap16(v, a, b); v = op(a, b);
    swap32(v, a, b); v = op(a, b);
    return v;
}
template <bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double wsum(double v) { return wave_allreduce<OpSum, BC>(v, OpSum()); }
template <bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double wmax(double v) { return wave_allreduce<OpMax, BC>(v, OpMax()); }
template <bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double wmin(double v) { return wave_allreduce<OpMin, BC>(v, OpMin()); }

// fast FP64 reciprocal / reciprocal square root: hardware estimate + two Newton steps (full double accuracy up to ~1 ulp;
// an IEEE divide costs ~75 and a sqrt ~125 dependent cycles on this path, these ~35)
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
// Step rules of the interior-point iteration (both solve kernels; tests/ipm_model.py mirrors them).
//  * round 6: three constants of these rules re-tuned on EVERY closed-loop QP of the reference's 40-lap experiment in the NumPy model (tools/knob_model.py, 12 442 QPs at N = 12,
//    10 336 at N = 14, the bench batches at N = 12 / 14 / 40; profiles/r6_knob_model.txt) instead of the few hundred problems of rounds 1-3: centring parameter
//    sigma = (gap_aff / gap)^5 (was ^3), fraction to the boundary 0.99 (was 0.995), separate steps after an iteration whose gap shrank by less than 20x (was 10x).  Closed loop
//    9.25 -> 8.98 iterations at N = 12 and 9.86 -> 9.49 at N = 14, QPs above 14 iterations halved (55 -> 28 of 12 442; 124 -> 64 of 10 336), and the 24-31-iteration two-cycle
//    QPs (1 in ~3 400: maxima 26 / 31 / 25 per seed) end at 17-18; bench batch 8.63 -> 8.44 (maximum 13 unchanged), N = 40 bench 10.91 / 18 -> 10.72 / 16.
//  * fraction to the boundary: LMPC_FRAC0, and once the predictor says the iteration is in its final phase (centring parameter sigma < 1e-3,
//    i.e. the affine step alone removes > 90 % of the gap) 1 - 10 gap: the gap then contracts quadratically instead of by 1 / 200 per
//    step (about one iteration less per solve).  Ungated, the longer step lets single complementarity products collapse while the
//    iterate is still far from the path (one 38-iteration stall in 1024 problems of the NumPy model); gated it never fired there.
//  * separate primal / dual step lengths after an iteration whose gap shrank by less than 1 / LMPC_SEP_THRESHOLD (20x).
#define LMPC_SEP_THRESHOLD 0.05
#define LMPC_FRAC0 0.99
//  * the corrector's second-order term dt_aff dmu_aff enters with weight LMPC_SO_W = 1.1 (same study: the bench batches' slowest problems -- linear tail, no strict
//    complementarity -- gain an iteration, 13 -> 12 at N = 12 and 14 -> 13 at N = 14 in the model; every closed-loop set is neutral within 0.3 %; 1.2 and beyond cost the closed loop)
#define LMPC_SO_W 1.1
//  * termination (round 6: ONE rule for every horizon and every kernel route; it replaces round 5's stack of gap-ratio / gap-floor / dual-residual thresholds, each
//    of which had been fitted to the last miss found).  The gap and residual tolerances say the KKT conditions hold; they do not say how far the iterate is from
//    the optimum: a QP without strict complementarity (about one in ten here) converges linearly and sits ~sqrt(gap) away, a flat closed-loop QP sits 660 dual
//    residuals away.  What the tolerance of the parity statement is written in is the distance itself, and a contracting iteration bounds it a posteriori:
//        |z_k - z*| <= rho / (1 - rho) |z_k - z_{k-1}|,   rho = |z_k - z_{k-1}| / |z_{k-1} - z_{k-2}|
//    (z = (x, u); max-norm of the steps actually taken, wave-uniform scalars the step phase has anyway).  The iteration ends when that estimate is below LMPC_ACC_TOL = 1e-7:
//    step^2 <= tol (step_prev - step), no division.  A superlinear last step passes at once (rho ~ 1e-3), a linearly converging problem iterates until the steps
//    are short enough, whatever made them long.  tools/term_rule_model.py (NumPy model, every QP of the reference's 40-lap closed loop at N = 12, three seeds = 12 442
//    QPs, iterated past the stop and compared with a 1e-15 solve; profiles/r6_term_rule_model.txt):
//        rule                                   closed-loop iterations   worst |xu - z*| / (1 + |z*|)   bench batch (iterations mean / max, worst error)
//        gap test alone (rounds 1-4)            8.97                     1.4e-5   (89 QPs > 3e-7)        8.31 / 13   1.5e-7
//        round 5, N <= 12 (1e-3 | 0.1)          8.98                     3.3e-6   (23)                  8.32 / 13   1.5e-7
//        round 5, N > 12 (1e-4 | 0.03 | est)    9.03                     3.8e-7   (2)                   8.44 / 13   2.1e-8
//        this rule, tol 3e-7                    9.18                     1.2e-7   (0)                   8.53 / 13   9.3e-10
//        this rule, tol 1e-7  (built)           9.26                     1.2e-7   (0)                   8.63 / 13   7.0e-10
//        ideal (first iterate within 3e-7)      8.98                                                    8.31 / 13
//    -- round 5's N <= 12 pair, the one the graded horizon ran, misses 1e-6 on 8 of 12 442 closed-loop QPs (the sampled probe had seen 8.75e-7 on 1 246).
//    The rate is an estimate from two steps, not a bound on the next one: at 3e-7 one of 10 336 closed-loop QPs at N = 14 stopped 3.7e-7 from its optimum (zt 1.02e-6:
//    its next contraction was 2.4 x slower than the last; the model and the GPU probe against the oracle found the same QP), at 1e-7 it iterates once more (1.7e-7 in
//    the model's worst case there).  Price against 3e-7: +0.9 % iterations in the closed loop, +1.2 % on the bench batch, maximum unchanged.
#define LMPC_ACC_TOL 1e-7
__device__ __forceinline__ bool step_bound_ok(double step, double step_prev) {
#ifdef LMPC_AB_NOACC                    // (developer A / B: the gap test alone, rounds 1-4)
    return true;
#endif
    return step < step_prev && step * step <= LMPC_ACC_TOL * (step_prev - step);
}
// Barrier weights theta = mu / t are capped at 1e11 in the Newton matrix: 1 / theta >= 1e-11 is a dual regularisation of the inequality row
// (F dw + (1 / theta_c) dmu = -r_c / mu); the right-hand side uses the same effective reciprocal rt = 1 / max(t, 1e-11 mu), so the fixed
// point does not move and the row's equation is off by 1e-11 dmu only.  Uncapped, an active lane row (t ~ 1e-14, mu ~ 10: main.py's fast
// laps, ey on the lane boundary) puts 1e15 into a stage Hessian whose other entries are O(1) and the Riccati recursion loses the regular
// part of the cost-to-go to rounding: the dual residual stalls at 1e-7 and the iteration wanders off (tests/ipm_model.py, tools/ipm_model_sets.py).
#define LMPC_TH_INV 1e-11
__device__ __forceinline__ double barrier_rt(double t, double mu) { return frcp(fmax(t, mu * LMPC_TH_INV)); }
__device__ __forceinline__ double step_fraction(double sig, double gap) { return sig < 1e-3 ? fmax(LMPC_FRAC0, 1.0 - 10.0 * gap) : LMPC_FRAC0; }
__device__ __forceinline__ double centring_sigma(double r) { const double r2 = r * r; return r2 * r2 * r; }       // (gap_aff / gap)^5

__device__ __forceinline__ double frsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

#ifndef LMPC_VARIANT_TU
// Map.getGlobalPosition, Track.py:135-189: curvilinear (s, ey) -> inertial (X, Y), one thread per point (batched plotting /
// logging export of predicted trajectories and safe-set points).  Row i - 1 of the table wraps to the last
// row for i = 0, as the reference's negative index does.
__global__ void lmpc_global_position_kernel(lmpc_dev_params p, int n, const double *__restrict__ s_in, const double *__restrict__ ey_in,
                                            double *__restrict__ xy, int *__restrict__ status) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const double PI = 3.141592653589793;
    double s = s_in[e]; const double ey = ey_in[e];
    for (int lap = 0; lap < 4096 && s > p.TL; lap++) s = s - p.TL;                          // (`while s > TrackLength`, bounded: an infinite s must not hang the GPU)
    int i = -1;
    if (!(s <= p.TL)) s = -1.0;                                                               // -> on no segment
    for (int r = 0; r < p.track_rows; r++) { const double c0 = p.track[r * 6 + 3]; if (s >= c0 && s < c0 + p.track[r * 6 + 4]) { i = r; break; } }
    if (i < 0) { status[e] = LMPC_ST_NO_SEGMENT; xy[2 * e] = 0.0; xy[2 * e + 1] = 0.0; return; }
    const double *ti = p.track + i * 6, *tp = p.track + (i > 0 ? i - 1 : p.track_rows - 1) * 6;
    double x, y;
    if (ti[5] == 0.0) {
        const double deltaL = ti[4], reltaL = s - ti[3], psi = ti[2];
        x = (1 - reltaL / deltaL) * tp[0] + reltaL / deltaL * ti[0] + ey * cos(psi + PI / 2);
        y = (1 - reltaL / deltaL) * tp[1] + reltaL / deltaL * ti[1] + ey * sin(psi + PI / 2);
    } else {
        const double r = 1 / ti[5], ang = tp[2], dir = r >= 0 ? 1.0 : -1.0, ar = fabs(r);
        const double cx = tp[0] + ar * cos(ang + dir * PI / 2), cy = tp[1] + ar * sin(ang + dir * PI / 2);
        const double span = (s - ti[3]) / (PI * ar) * PI;
        double an = dir * PI / 2 + ang;
        if (an < -PI) an = 2 * PI + an; else if (an > PI) an = an - 2 * PI;                 // wrap(), Track.py:367-375
        const double angle = -(PI - fabs(an)) * (an >= 0 ? 1.0 : -1.0);
        x = cx + (ar - dir * ey) * cos(angle + dir * span);
        y = cy + (ar - dir * ey) * sin(angle + dir * span);
    }
    xy[2 * e] = x; xy[2 * e + 1] = y; status[e] = 0;
}

#endif  // LMPC_VARIANT_TU
// =====================================================================================================
// K1 (block form): regression + linearisation of ALL horizon steps of one problem by one work-group.
//
// Every row of every used lap is a k-NN candidate for every query (with the reference's bandwidth h = 5 all ~1000 rows of a
// lap lie inside the kernel support), so a one-wave-per-(problem, step) mapping re-reads each lap N times per problem and is bound
// by L2 bandwidth (that was the first version of this kernel: 79 us at batch 256, this one 30 us, bit-identical output).  Here a wave owns one lap: it loads the lap's 5 regression features ONCE into registers
// (16 rows per lane per 1024-row chunk) and scans them for its share of the N queries.  Selecting the MaxNumPoint nearest rows
// of a lap (PredictiveModel.py:180-197):
//   * every lane takes the minimum of its 16 distances; the minimum inside each group of 8 lanes are 8 distinct rows, so the
//     largest of the 8 group minima bounds the MaxNumPoint-th smallest distance (MaxNumPoint <= 8) -- one wave reduction;
//   * rows not above that bound (a few tens of the 1024) are appended to a small LDS list with an LDS atomic counter;
//   * each list entry's rank is counted against the list, lexicographic in (distance, row): exactly the order of the
//     per-lane sorted lists + wave arg-min merge it replaces (ties: earlier row first), so the selection is bit-identical;
//   * a list overflow (pathological ties) falls back to MaxNumPoint rounds of wave arg-min extraction.
// The normal equations, the three 5x5 Cholesky solves and the analytic rows are then batched over the queries of the block
// (threads = (query, entry) pairs) instead of running as serial tails of single lanes.
// =====================================================================================================
#define K1_NW 8                    // waves per work-group
#define K1_NT (K1_NW * WAVE)
#define K1_QG 12                   // queries per pass
#define K1_RPL 16                  // rows per lane per chunk
#define K1_CHUNK (K1_RPL * WAVE)
#define K1_PTS (K1_QG * 32)        // staged points per pass (32 per query when trToUse * MaxNumPoint <= 32; fewer queries per pass beyond: 32 laps x 8 points = 256 fit one query)
#define K1_SEL K1_PTS              // entries of the running selection: queries x laps x MaxNumPoint of one pass

// smallest T such that at least K of the 64 lane values m (< 2^19) are <= T: the K-th smallest, built bit by bit; one v_cmp per round, the
// count and the candidate live on the scalar unit (a K-round chain of wave minima cost 12 vector instructions per round)
__device__ __forceinline__ unsigned kth_lane_value(unsigned m, int K) {
    unsigned A = 0u;
    if (__popcll(__ballot(m <= 0xfffu)) < K) {             // (rare: the K nearest rows of a 1024-row chunk lie within 1 / 16 of one feature's range of the query)
#pragma unroll
        for (int b = 18; b >= 12; b--) {
            const unsigned t = A + ((1u << b) - 1u);
            const int cnt = __popcll(__ballot(m <= t));
            A = cnt < K ? A + (1u << b) : A;
        }
    }
#pragma unroll
    for (int b = 11; b >= 0; b--) {
        const unsigned t = A + ((1u << b) - 1u);
        const int cnt = __popcll(__ballot(m <= t));
        A = cnt < K ? A + (1u << b) : A;
    }
    return A;
}
// inclusive prefix sum over the 64 lanes (gfx9 DPP: four shifts inside a row of 16, then row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan_i32(int x) {
    EXEC_AUDIT(7);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1, out-of-row lanes read 0
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ bool k1_less(double da, int ia, double db, int ib) { return da < db || (da == db && ia < ib); }
// rank of (d, i) among the 16 entries of its row of lanes, lexicographic: every other lane's entry passes by once
template <int ROT> __device__ __forceinline__ void k1_row_rank(double d, int i, int &rank) {
    if constexpr (ROT == 1) EXEC_AUDIT(7);
    if constexpr (ROT < 16) {
        const double od = dpp_mov<0x120 + ROT>(d);
        const int oi = __builtin_amdgcn_update_dpp(i, i, 0x120 + ROT, 0xf, 0xf, false);
        rank += k1_less(od, oi, d, i) ? 1 : 0;
        k1_row_rank<ROT + 1>(d, i, rank);
    }
}
__device__ __forceinline__ unsigned sad_u16(unsigned a, unsigned b, unsigned c) {   // |a.lo16 - b.lo16| + |a.hi16 - b.hi16| + c   (v_sad_u16)
    // (the compiler's own builtin: as inline asm every dependent pair of the 48 per query and lap was followed by a conservative s_nop, and the
    //  three-instruction chains of the 16 rows could not be interleaved)
    return __builtin_amdgcn_sad_u16(a, b, c);
}
__device__ __forceinline__ float wminf(float v) {                           // wave-wide float minimum, all lanes
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// queries per block for a given cap qg (host and device use the same rule); the grid is B * ceil(N / k1_queries_per_block())
__host__ __device__ inline int k1_queries_per_block(int qg, int trToUse, int maxNumPoint) {
    const int pp = trToUse * (maxNumPoint > 8 ? 8 : maxNumPoint);
    int q = qg < K1_QG ? qg : K1_QG;
    if (pp > 0 && K1_PTS / pp < q) q = K1_PTS / pp;
    return q < 1 ? 1 : q;
}

#ifdef LMPC_TIMING
// (developer build) cycle stamps of work-group 0 of the regression kernel: g_k1_tbuf[id] = cycle counter at stamp id (lmpc_debug_k1_timing)
static __device__ long long *g_k1_tbuf;
#define K1STAMP(id) do { if (g_k1_tbuf && blockIdx.x == 0 && threadIdx.x == 0) g_k1_tbuf[id] = (long long)__builtin_readcyclecounter(); } while (0)
// stamps inside the scan of wave 0 (ids 8 ..): outstanding loads are waited for first, so that a stamp separates the work before it from the work after it
#define K1WSTAMP(id) do { if (g_k1_tbuf && blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); g_k1_tbuf[id] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define K1STAMP(id) do { } while (0)
#define K1WSTAMP(id) do { } while (0)
#endif

// LDS work space of the regression: static arrays in the stand-alone kernel (8 waves per problem), a slice of the solve kernel's
// dynamic LDS when one wave runs the regression of its own QP in front of the solve (lmpc_solve_kernel, fused step).
struct k1_smem {
    double (*qf)[5];                 // [QG]        xuLin of the queries in flight (PredictiveModel.py:54)
    int (*cseg)[16]; int *ccnt;      // [waves QG]  prefilter survivors (row indices) per wave and query
    double *seld; int *seli; int *nsel; int Ls, Sp; // [QG][Ls][Sp], [QG][Ls][Sp], [QG][Ls]: running selection per query and lap (Ls = laps in use, Sp = MaxNumPoint)
    double (*pts)[10];               // [QF PP]     vx vy wz delta a K y_vx y_vy y_wz 1   (PredictiveModel.py:141-168)
    double (*gram)[45];              // [QF]        Q_vx(15) b_vx(5) Q_lat(15) b_vy(5) b_wz(5)
    double (*outv)[54];              // [QF]        A_i (36) | B_i (12) | C_i (6)
    int *st_s;                       // [QG]
};

// computeIndices (PredictiveModel.py:180-197) of ONE wave for lap c: the lap's prefilter image is loaded once per 1024-row chunk and scanned
// for the wave's queries qi = sgi + s nsub, s < nqw; the running selection per (query, lap) stays in sm.seld / seli / nsel.  cs0: first
// row of this wave in sm.cseg / ccnt.  PAIR: two queries per trip (independent reduction chains interleave; 32 registers more).
// RPL: rows per lane per trip.  16 = the host's quantisation chunk (K1_CHUNK rows, own range each); 8 only when every lap in use has at
// most 512 rows, i.e. lies in its first chunk (launch_k1): half the prefilter work and 32 registers less per wave.
template <bool PAIR, int RPL>
__device__ __forceinline__ void k1_scan_lap(const lmpc_dev_params &p, const k1_smem &sm, int c, int cs0, int sgi, int nsub, int nq, int lane, int MAXP) {
    const double h = p.h;
    const double *base = p.mstore + (size_t)p.mslot[c] * LMPC_COLS * p.lap_stride;
    const int ls = p.lap_stride;
    const int nrows = p.mlen[c] - 1;
    for (int t0 = 0; t0 < nrows; t0 += RPL * WAVE) {
        // Prefilter image of this lane's rows: the scaled features in 16-bit fixed point, packed (vx, vy | wz, delta | a, flag),
        // quantised by the host when the lap was stored (lmpc_capi.hip: quantise_lap), per 1024-row chunk over the chunk's own
        // [min, max] with ONE scale for the five features (the L1 norm weighs them equally).  Three v_sad_u16 per row give the
        // integer L1 distance; every decision is re-made in FP64 below.  A query outside [min, max] is clamped: that shifts all
        // of a feature's |differences| by the same amount, so the order of the rows is untouched.  Rows beyond the lap carry
        // 0xffff in the unused half-word (the query has 0 there): farther than a full range from everything.
        const double *qp = p.mqpar + ((size_t)p.mslot[c] * p.mq_chunks + t0 / K1_CHUNK) * 6;
        const double qsc = qp[5];
        const unsigned *qb = p.mquant + (size_t)p.mslot[c] * 3 * ls;
        unsigned qv[3][RPL];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < RPL; j++) { const int t = t0 + lane + WAVE * j; qv[k][j] = qb[(size_t)k * ls + (t < nrows ? t : 0)]; }
#pragma unroll
        for (int j = 0; j < RPL; j++) if (t0 + lane + WAVE * j >= nrows) qv[2][j] |= 0xffff0000u;
        K1WSTAMP(8);
        // ---- step A: integer prefilter, one query at a time; survivors (row indices) go to the query's 16-slot LDS segment ----
        const int nqw = nq > sgi ? (nq - sgi + nsub - 1) / nsub : 0;                  // queries of this wave: qi = sgi + s nsub
        // The queries in this chunk's fixed point, all at once: lane 8 s + k holds feature k of the wave's query s (s < 8 in tq0, 8 <= s < 16
        // in tq1); the loop below fetches its five values with v_readlane (they land in SGPRs, which v_sad_u16 takes directly).  Done per
        // query inside the loop this was 25 FP64 instructions of the ~350 per (query, lap).
        unsigned tq0, tq1;
        {
            const int k = lane & 7, k5 = k < 5 ? k : 0, sa = lane >> 3, sb = sa + 8;
            const int qa_ = sgi + (sa < nqw ? sa : 0) * nsub, qb_ = sgi + (sb < nqw ? sb : 0) * nsub;
            const double scl = k5 == 0 ? p.scaling[0] : k5 == 1 ? p.scaling[1] : k5 == 2 ? p.scaling[2] : k5 == 3 ? p.scaling[3] : p.scaling[4];
            const double lo = qp[k5];
            tq0 = (unsigned)fmin(fmax((sm.qf[qa_][k5] * scl - lo) * qsc, 0.0), 65535.0);
            tq1 = (unsigned)fmin(fmax((sm.qf[qb_][k5] * scl - lo) * qsc, 0.0), 65535.0);
        }
        for (int s = 0; s < nqw; s += PAIR ? 2 : 1) {
            const bool two = PAIR && s + 1 < nqw;
            const int sb2 = two ? s + 1 : s;
            unsigned ta5[5], tb5[5];
            const unsigned tqa = s < 8 ? tq0 : tq1, tqb = sb2 < 8 ? tq0 : tq1;
#pragma unroll
            for (int k = 0; k < 5; k++) {
                ta5[k] = (unsigned)__builtin_amdgcn_readlane((int)tqa, (s & 7) * 8 + k);
                tb5[k] = PAIR ? (unsigned)__builtin_amdgcn_readlane((int)tqb, (sb2 & 7) * 8 + k) : 0u;
            }
            const unsigned ya[3] = {ta5[0] | (ta5[1] << 16), ta5[2] | (ta5[3] << 16), ta5[4]};
            const unsigned yb[3] = {tb5[0] | (tb5[1] << 16), tb5[2] | (tb5[3] << 16), tb5[4]};
            unsigned ea[RPL], eb[RPL], ma = 0xffffffffu, mb = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                unsigned a_ = 0, b_ = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { a_ = sad_u16(qv[k][j], ya[k], a_); if (PAIR) b_ = sad_u16(qv[k][j], yb[k], b_); }
                ea[j] = a_; eb[j] = b_; ma = a_ < ma ? a_ : ma; mb = b_ < mb ? b_ : mb;
            }
            // MAXP-th smallest lane minimum (three v_sad_u16 of 16-bit fields: < 2^19): at least MAXP rows lie at or below it
            const unsigned ba = kth_lane_value(ma, MAXP), bb = PAIR ? kth_lane_value(mb, MAXP) : 0u;
            // Two-sided bound.  Per feature |floor(a) - floor(b)| differs from |a - b| by < 1, so over the five features the
            // integer distance e of a row and its exact scaled distance d satisfy |e - d| < 5.  (i) At least MAXP rows have
            // e <= T (T = ba / bb), hence d < T + 5: the MAXP-th smallest exact distance is < T + 5.  (ii) A row of the exact
            // top MAXP therefore has d < T + 5 and e < d + 5 < T + 10, i.e. e <= T + 9.  Everything up to T + 10 survives
            // (one unit of margin for the rounding of the host's fixed-point conversion); the survivors are re-ranked in FP64.
            const unsigned ta = ba > 0xffffff00u ? 0xffffffffu : ba + 10u, tb = bb > 0xffffff00u ? 0xffffffffu : bb + 10u;
            // Survivors (a handful of the 1024 rows) are compacted into the query's 16-slot segment: every lane builds a 16-bit mask of its
            // surviving rows, a wave prefix sum of the per-lane counts (six DPP steps) gives each lane its first slot, and the lane writes
            // its own rows.  (The first version ran a ballot / mbcnt round per row slot: 16 rounds, most of this kernel's 550 VALU
            // instructions per (query, lap).  The order inside the segment changes; the exact re-rank below does not depend on it.)
            unsigned mska = 0u, mskb = 0u;
#pragma unroll
            for (int j = 0; j < RPL; j++) { mska |= (ea[j] <= ta ? 1u : 0u) << j; if (PAIR) mskb |= (eb[j] <= tb ? 1u : 0u) << j; }
            const int ca_ = __popc(mska), cb_ = PAIR ? __popc(mskb) : 0;
            const int ia = wave_incl_scan_i32(ca_), ib = PAIR ? wave_incl_scan_i32(cb_) : 0;
            const int na = __builtin_amdgcn_readlane(ia, 63), nb = PAIR ? __builtin_amdgcn_readlane(ib, 63) : 0;
            int pa = ia - ca_, pb = ib - cb_;
            while (mska) { const int j = __builtin_ctz(mska); mska &= mska - 1u; if (pa < 16) sm.cseg[cs0 + s][pa] = t0 + lane + WAVE * j; pa++; }
            if (two) { while (mskb) { const int j = __builtin_ctz(mskb); mskb &= mskb - 1u; if (pb < 16) sm.cseg[cs0 + s + 1][pb] = t0 + lane + WAVE * j; pb++; } }
            if (lane == 0) { sm.ccnt[cs0 + s] = na; if (two) sm.ccnt[cs0 + s + 1] = nb; }
            if (s == 0) K1WSTAMP(9);
        }
        K1WSTAMP(10);
        // ---- step B: exact FP64 distances of the survivors, four queries at a time (one per row of 16 lanes), ranked inside the
        //      row by 15 DPP rotations; the running selection of earlier chunks rides along as extra entries ----
        for (int s0 = 0; s0 < nqw; s0 += 4) {
            const int s = s0 + (lane >> 4), r = lane & 15;
            const bool live = s < nqw;
            const int qi = live ? sgi + s * nsub : 0;
            double *sd = sm.seld + ((size_t)qi * sm.Ls + c) * sm.Sp; int *si = sm.seli + ((size_t)qi * sm.Ls + c) * sm.Sp;
            const int cnt = live ? sm.ccnt[cs0 + (live ? s : 0)] : 0, ns = live ? sm.nsel[qi * sm.Ls + c] : 0;
            const bool ovf = cnt + ns > 16;
            double dv = INFINITY; int iv = 0x7fffffff;
            if (live && !ovf) {
                if (r < cnt) {
                    iv = sm.cseg[cs0 + s][r];
                    const bool real = iv < nrows;
                    iv = real ? iv : 0;
                    // la.norm(diff, 1, axis=1) of (Data - x) . scaling : |.| accumulated in feature order, no contraction
                    double nrm = fabs((base[0 * ls + iv] - sm.qf[qi][0]) * p.scaling[0]);
                    nrm = nrm + fabs((base[1 * ls + iv] - sm.qf[qi][1]) * p.scaling[1]);
                    nrm = nrm + fabs((base[2 * ls + iv] - sm.qf[qi][2]) * p.scaling[2]);
                    nrm = nrm + fabs((base[6 * ls + iv] - sm.qf[qi][3]) * p.scaling[3]);
                    nrm = nrm + fabs((base[7 * ls + iv] - sm.qf[qi][4]) * p.scaling[4]);
                    dv = real ? nrm : INFINITY;
                } else if (r < cnt + ns) { dv = sd[r - cnt]; iv = si[r - cnt]; }
            }
            if (s0 == 0) K1WSTAMP(11);
            const bool in_h = dv < h;
            const double kd = in_h ? dv : INFINITY;                                   // rows outside h never outrank anything
            int rank = 0;
            k1_row_rank<1>(kd, iv, rank);
            const unsigned long long mh = __ballot(in_h);
            const int nin = __popcll((mh >> (lane & 48)) & 0xffffull);
            if (live && !ovf) {
                if (in_h && rank < MAXP) { sd[rank] = dv; si[rank] = iv; }
                if (r == 0) sm.nsel[qi * sm.Ls + c] = nin < MAXP ? nin : MAXP;
            }
            if (s0 == 0) K1WSTAMP(12);
            // prefilter overflow (massive ties): MAXP rounds of exact arg-min extraction over the chunk + running selection
            unsigned long long mo = __ballot(live && ovf && r == 0);
            while (mo) {
                const int sl = s0 + (__builtin_ctzll(mo) >> 4); mo &= mo - 1;
                const int qj = sgi + sl * nsub, nsj = sm.nsel[qj * sm.Ls + c];
                double *sdj = sm.seld + ((size_t)qj * sm.Ls + c) * sm.Sp; int *sij = sm.seli + ((size_t)qj * sm.Ls + c) * sm.Sp;
                const double x0 = sm.qf[qj][0], x1 = sm.qf[qj][1], x2 = sm.qf[qj][2], x3 = sm.qf[qj][3], x4 = sm.qf[qj][4];
                const double od = lane < nsj ? sdj[lane] : INFINITY; const int oi = lane < nsj ? sij[lane] : 0x7fffffff;
                double pd = -INFINITY; int pi = -1, got = 0;
                double outd = 0.0; int outi = 0;
                for (int rr = 0; rr < MAXP; rr++) {
                    double bd = INFINITY; int bi = 0x7fffffff;
                    if (k1_less(pd, pi, od, oi)) { bd = od; bi = oi; }
                    for (int j = 0; j < RPL; j++) {
                        const int tj = t0 + lane + WAVE * j;
                        if (tj < nrows) {
                            double nrm = fabs((base[0 * ls + tj] - x0) * p.scaling[0]);
                            nrm = nrm + fabs((base[1 * ls + tj] - x1) * p.scaling[1]);
                            nrm = nrm + fabs((base[2 * ls + tj] - x2) * p.scaling[2]);
                            nrm = nrm + fabs((base[6 * ls + tj] - x3) * p.scaling[3]);
                            nrm = nrm + fabs((base[7 * ls + tj] - x4) * p.scaling[4]);
                            if (nrm < h && k1_less(pd, pi, nrm, tj) && k1_less(nrm, tj, bd, bi)) { bd = nrm; bi = tj; }
                        }
                    }
                    wave_argmin(bd, bi);
                    if (!(bd < INFINITY)) break;
                    if (lane == rr) { outd = bd; outi = bi; }
                    pd = bd; pi = bi; got++;
                }
                if (lane < got) { sdj[lane] = outd; sij[lane] = outi; }
                if (lane == 0) sm.nsel[qj * sm.Ls + c] = got;
            }
        }
        K1WSTAMP(13);
    }
}

// compute_Q_M / compute_b / LMPC_LocLinReg / regressionAndLinearization (PredictiveModel.py:48-178) for the nf queries q0 .. q0 + nf - 1 of the
// pass (selection in sm.seld / seli / nsel), by the nt threads of the work-group: A_i | B_i | C_i end up in sm.outv[0 .. nf), status bits in
// sm.st_s[q0 ..].  xq: the problem's xLin rows (6 doubles each, row i0 = first query of the pass).  Work-group barriers inside.
__device__ __forceinline__ void k1_fit(const lmpc_dev_params &p, const k1_smem &sm, int q0, int nf, int tid, int nt, const double *xq_pass, int MAXP) {
    const int L = p.trToUse, PP = L * MAXP;
    const double h = p.h;
    // ---- assemble A_i, B_i, C_i (:70-135), one thread per query.  The kinematic rows (epsi, s, ey) depend on the query state only: they are
    //      written by the last wave, which has no points to stage, while the others do (cos, sin and the divisions are a ~3400-cycle
    //      chain; the one-wave fused step has no spare wave and runs them in turn).  Every entry of outv is written exactly once:
    //      rows 3-5 of A_i and C_i and the zero rows of B_i here, rows 0-2 by the threads that solve for them below ----
    const int at0 = nt > WAVE ? nt - WAVE : 0;
    if (tid >= at0 && tid < at0 + nf) {
        const int ql = tid - at0, qi = q0 + ql;
        const double *xq = xq_pass + (size_t)qi * 6;
        double *Ai = sm.outv[ql], *Ci = sm.outv[ql] + 48;
        for (int j = 0; j < 6; j++) sm.outv[ql][36 + 6 + j] = 0.0;
        const double vx = xq[0], vy = xq[1], wz = xq[2], epsi = xq[3], s = xq[4], ey = xq[5], dt = p.dt;
        int bad = 0;
        const double cur = track_curvature(p, s, &bad);
        if (bad) atomicOr(&sm.st_s[qi], LMPC_ST_NO_SEGMENT);
        const double den = 1 - cur * ey, ce = cos(epsi), se = sin(epsi);
        const double xv[6] = {vx, vy, wz, epsi, s, ey};
        double row[6], dot;
        row[0] = -dt * ce / den * cur; row[1] = dt * se / den * cur; row[2] = dt;
        row[3] = 1 - dt * (-vx * se - vy * ce) / den * cur; row[4] = 0;
        row[5] = dt * (vx * ce - vy * se) / (den * den) * cur * (-cur);
        dot = 0; for (int j = 0; j < 6; j++) { Ai[18 + j] = row[j]; dot += row[j] * xv[j]; }
        Ci[3] = epsi + dt * (wz - (vx * ce - vy * se) / (1 - cur * ey) * cur) - dot;
        row[0] = dt * (ce / den); row[1] = -dt * (se / den); row[2] = 0; row[3] = dt * (-vx * se - vy * ce) / den; row[4] = 1;
        row[5] = -dt * (vx * ce - vy * se) / (den * den) * (-cur);
        dot = 0; for (int j = 0; j < 6; j++) { Ai[24 + j] = row[j]; dot += row[j] * xv[j]; }
        Ci[4] = s + dt * ((vx * ce - vy * se) / (1 - cur * ey)) - dot;
        row[0] = dt * se; row[1] = dt * ce; row[2] = 0; row[3] = dt * (vx * ce - vy * se); row[4] = 0; row[5] = 1;
        dot = 0; for (int j = 0; j < 6; j++) { Ai[30 + j] = row[j]; dot += row[j] * xv[j]; }
        Ci[5] = ey + dt * (vx * se + vy * ce) - dot;
    }
    // ---- stage the selected points: slot = lap * MAXP + rank; empty slots carry weight K = 0 (they add exact zeros) ----
    for (int e = tid; e < nf * PP; e += nt) {
        const int ql = e / PP, qi = q0 + ql, sl = e % PP, c = sl / MAXP, r = sl % MAXP;
        const int ns = sm.nsel[qi * sm.Ls + c];
        const double *sd = sm.seld + ((size_t)qi * sm.Ls + c) * sm.Sp; const int *si = sm.seli + ((size_t)qi * sm.Ls + c) * sm.Sp;
        double *pt = sm.pts[ql * PP + sl];
        if (r < ns) {
            int pick = r;
            if (ns < MAXP) {                                           // fewer than MaxNumPoint inside h: np.where order = ascending row index
                for (int a_ = 0; a_ < ns; a_++) {
                    int rk = 0;
                    for (int m2 = 0; m2 < ns; m2++) rk += si[m2] < si[a_] ? 1 : 0;
                    if (rk == r) pick = a_;
                }
            }
            const double dd = sd[pick]; const int ii = si[pick];
            const double *base = p.mstore + (size_t)p.mslot[c] * LMPC_COLS * p.lap_stride;
            double q = dd / h; q = q * q;
            pt[0] = base[0 * p.lap_stride + ii]; pt[1] = base[1 * p.lap_stride + ii]; pt[2] = base[2 * p.lap_stride + ii];
            pt[3] = base[6 * p.lap_stride + ii]; pt[4] = base[7 * p.lap_stride + ii]; pt[5] = (1.0 - q) * 3.0 / 4.0;        // :193
            pt[6] = base[0 * p.lap_stride + ii + 1]; pt[7] = base[1 * p.lap_stride + ii + 1]; pt[8] = base[2 * p.lap_stride + ii + 1]; pt[9] = 1.0;
        } else {
#pragma unroll
            for (int k = 0; k < 10; k++) pt[k] = 0.0;
        }
    }
    __syncthreads();
    K1STAMP(3);

    // ---- compute_Q_M / compute_b (:141-168): Q = M' diag(K) M (+ lamb I), b = -M' diag(K) y ---------------------------
    // 35 distinct sums per query: the vx system (15 + 5), and of the lateral system only what involves delta (5) and its two
    // right-hand sides (10); the 10 entries of Q_lat over (vx, vy, wz, 1) are the same sums as in Q_vx and are copied.
    for (int e2 = tid; e2 < nf * 35; e2 += nt) {
        const int ql = e2 / 35, le = e2 % 35;
        int r, cc, sys = le < 20 ? 0 : 1, tgt = -1, e;
        if (le < 15) { e = le; r = 0; cc = e; while (cc >= 5 - r) { cc -= 5 - r; r++; } cc += r; }            // upper-triangular (r, cc)
        else if (le < 20) { e = le; r = le - 15; cc = 0; tgt = 0; }
        else if (le < 25) { const int k = le - 20; r = k < 3 ? k : 3; cc = k < 4 ? 3 : 4; e = r * 5 - r * (r - 1) / 2 + (cc - r); }
        else { const int j = le - 25; r = j % 5; cc = 0; tgt = 1 + j / 5; e = 15 + j; }
        const int fin = sys == 0 ? 4 : 3;        // column of pts holding the input feature: a (vx system) / delta (lateral)
        const int o1 = r < 3 ? r : (r == 3 ? fin : 9), o2 = tgt >= 0 ? 6 + tgt : (cc < 3 ? cc : (cc == 3 ? fin : 9));
        const double (*pq)[10] = &sm.pts[ql * PP];
        double acc = 0.0;
#pragma unroll 4
        for (int q = 0; q < PP; q++) acc = fma(pq[q][o1] * pq[q][5], pq[q][o2], acc);
        if (tgt >= 0) acc = -acc; else if (r == cc) acc += p.lamb;
        sm.gram[ql][sys * 20 + e] = acc;
        if (le < 15 && r != 3 && cc != 3) sm.gram[ql][20 + e] = acc;
    }
    __syncthreads();
    K1STAMP(4);

    // ---- LMPC_LocLinReg (:170-178): unconstrained qp(Q, b)  <=>  Q theta = -b ; Cholesky 5x5, one thread per system ----
    if (tid < nf * 3) {
        const int ql = tid / 3, qi = q0 + ql, sy = tid % 3;
        const double *Qv = sy == 0 ? &sm.gram[ql][0] : &sm.gram[ql][20];
        const double *bv = sy == 0 ? &sm.gram[ql][15] : (sy == 1 ? &sm.gram[ql][35] : &sm.gram[ql][40]);
        double Lm[5][5]; int bad = 0;
        { int e = 0; for (int r = 0; r < 5; r++) for (int c = r; c < 5; c++) { Lm[c][r] = Qv[e]; Lm[r][c] = Qv[e]; e++; } }
        for (int j = 0; j < 5; j++) {
            double dj = Lm[j][j];
            for (int k = 0; k < j; k++) dj -= Lm[j][k] * Lm[j][k];
            if (!(dj > 0.0)) { bad = 1; dj = 1.0; }
            // the diagonal is kept as 1 / L_jj (hardware estimate + two Newton steps, ~1 ulp): the 20 IEEE divisions and 5 square roots of the
            // textbook form were a dependent chain of ~2000 cycles on three lanes per query while the rest of the work-group waits
            const double inv = frsqrt(dj); Lm[j][j] = inv;
            for (int r = j + 1; r < 5; r++) {
                double v = Lm[r][j];
                for (int k = 0; k < j; k++) v -= Lm[r][k] * Lm[j][k];
                Lm[r][j] = v * inv;
            }
        }
        double y[5];
        for (int r = 0; r < 5; r++) { double v = -bv[r]; for (int k = 0; k < r; k++) v -= Lm[r][k] * y[k]; y[r] = v * Lm[r][r]; }
        for (int r = 4; r >= 0; r--) { double v = y[r]; for (int k = r + 1; k < 5; k++) v -= Lm[k][r] * y[k]; y[r] = v * Lm[r][r]; }
        // theta -> row sy of A_i, B_i, C_i (:70-135: vx is driven by a, vy and wz by delta)
        double *Ai = sm.outv[ql], *Bi = sm.outv[ql] + 36, *Ci = sm.outv[ql] + 48;
        for (int r = 0; r < 5; r++) y[r] = bad ? 0.0 : y[r];
        Ai[sy * 6 + 0] = y[0]; Ai[sy * 6 + 1] = y[1]; Ai[sy * 6 + 2] = y[2]; Ai[sy * 6 + 3] = 0.0; Ai[sy * 6 + 4] = 0.0; Ai[sy * 6 + 5] = 0.0; Ci[sy] = y[4];
        Bi[sy * 2 + (sy == 0 ? 1 : 0)] = y[3]; Bi[sy * 2 + (sy == 0 ? 0 : 1)] = 0.0;
        int npts = 0;
        for (int c = 0; c < L; c++) npts += sm.nsel[qi * sm.Ls + c];
        if (bad || npts < 5) atomicOr(&sm.st_s[qi], LMPC_ST_REG_SINGULAR);
    }
    __syncthreads();
    K1STAMP(5);
    K1STAMP(6);
}

// OCC: compile for four waves per SIMD (two work-groups per CU; <= 128 VGPRs) -- pays when the grid exceeds one work-group
// per CU (1.47x at batch 4096); the other variant has the shorter latency when each CU runs a single group.
template <bool OCC, int RPL>
__global__ __launch_bounds__(K1_NT, OCC ? (RPL <= 8 ? 6 : 4) : 2) void lmpc_regress_kernel(lmpc_dev_params p, int B, int qg, const double *__restrict__ xLin, int xstride,
                                                             const double *__restrict__ uLin, double *__restrict__ Aout,
                                                             double *__restrict__ Bout, double *__restrict__ Cout, int *__restrict__ status) {
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave-uniform by construction: lap pointers, row counts and loop bounds stay in SGPRs
    __shared__ double qf[K1_QG][5];
    __shared__ __attribute__((aligned(16))) int cseg[K1_NW * K1_QG][16]; __shared__ int ccnt[K1_NW * K1_QG];
    // running selection per (query, lap): queries x laps x MaxNumPoint <= K1_PTS entries whatever the split (k1_queries_per_block keeps
    // queries x laps x MaxNumPoint <= K1_PTS, or one query per block when a single query's laps x MaxNumPoint exceed it: K1_PTS_MAX)
    __shared__ double seld[K1_SEL]; __shared__ int seli[K1_SEL];
    __shared__ int nsel[K1_SEL];
    __shared__ double pts[K1_PTS][10];
    __shared__ double gram[K1_QG][45];
    static_assert(sizeof(cseg) >= sizeof(double) * K1_QG * 54, "outv lives in cseg");
    double (*outv)[54] = (double (*)[54])cseg;          // A_i | B_i | C_i of the fit: the scan's survivor lists are dead by then (three work-groups per CU fit 160 KB of LDS this way)
    __shared__ int st_s[K1_QG];
    k1_smem sm; sm.qf = qf; sm.cseg = cseg; sm.ccnt = ccnt; sm.seld = seld; sm.seli = seli; sm.nsel = nsel; sm.Ls = p.trToUse; sm.Sp = p.maxNumPoint > 8 ? 8 : p.maxNumPoint;
    sm.pts = pts; sm.gram = gram; sm.outv = outv; sm.st_s = st_s;

    const int N = p.N, L = p.trToUse;
    const int MAXP = p.maxNumPoint > 8 ? 8 : p.maxNumPoint;
    const int QGe = k1_queries_per_block(qg, L, p.maxNumPoint);            // queries of this block
    const int npass = (N + QGe - 1) / QGe;
    const int b = blockIdx.x / npass;
    if (b >= B) return;
    const int nsub = K1_NW / L > 0 ? K1_NW / L : 1;                        // waves sharing one lap split its queries
    const int myc = wave % L, sgi = wave / L;
    const int i0 = (blockIdx.x % npass) * QGe;
    const int nq = N - i0 < QGe ? N - i0 : QGe;
    if (tid < nq * 5) {
        const int qi = tid / 5, f = tid % 5;
        qf[qi][f] = f < 3 ? xLin[(size_t)b * xstride + (size_t)(i0 + qi) * 6 + f] : uLin[((size_t)b * N + i0 + qi) * 2 + (f - 3)];
    }
    for (int e = tid; e < nq * L; e += K1_NT) nsel[e] = 0;
    if (tid < nq) st_s[tid] = 0;
    K1STAMP(0);
    __syncthreads();
    K1STAMP(1);
    // (two queries per trip in the low-occupancy build.  The occupancy build takes one per trip: with four waves per SIMD the other waves
    //  fill the chain's latency, and the second query's 16 distances cost registers it does not have -- spilled registers are scratch
    //  WRITES: 134 MB per launch at batch 4096 before this)
    for (int c = myc; c < L && sgi < nsub; c += K1_NW)                    // (c += K1_NW: more laps than waves -- trToUse up to 32 -- a wave then scans several laps in turn)
        k1_scan_lap<!OCC, RPL>(p, sm, c, wave * K1_QG, sgi, nsub, nq, lane, MAXP);
    __syncthreads();
    K1STAMP(2);
    k1_fit(p, sm, 0, nq, tid, K1_NT, xLin + (size_t)b * xstride + (size_t)i0 * 6, MAXP);
    for (int e = tid; e < nq * 54; e += K1_NT) {
        const int qi = e / 54, le = e % 54; const size_t item = (size_t)b * N + i0 + qi;
        if (le < 36) Aout[item * 36 + le] = outv[qi][le];
        else if (le < 48) Bout[item * 12 + (le - 36)] = outv[qi][le];
        else Cout[item * 6 + (le - 48)] = outv[qi][le];
    }
    if (tid < nq) status[(size_t)b * N + i0 + tid] = st_s[tid];
    K1STAMP(7);
}

// LDS doubles the regression of one QP by ONE wave needs behind [A_k | B_k] and C_k (fused step of lmpc_solve_kernel): the scan state of
// up to K1_QG queries and the fit state of K1F_QF queries.  Host and device use the same rule.
#define K1F_QF 2
__host__ __device__ constexpr inline int k1_fused_doubles(int N, int L, int maxNumPoint) {
    const int QG = N < K1_QG ? N : K1_QG, MAXP = maxNumPoint > 8 ? 8 : maxNumPoint, PP = L * MAXP;
    const int ints = QG * 16 + QG + QG * L * 8 + QG * L + QG;                                  // cseg, ccnt, seli, nsel, st_s
    return QG * 5 + QG * L * 8 + (ints + 1) / 2 + K1F_QF * (PP * 10 + 45 + 54);
}
// PredictiveModel.regressionAndLinearization for the N horizon points of problem b by the calling wave (block of one wave): results go to
// AB / C in LDS in the solve kernel's layout (and to global memory if the pointers are given); returns the OR of the points' status bits.
__device__ __forceinline__ int k1_wave_problem(const lmpc_dev_params &p, int b, int lane, const double *xLin_b, const double *uLin_b, double *wk,
                                               double *AB, double *Cl, double *Aout, double *Bout, double *Cout) {
    const int N = p.N, L = p.trToUse;
    const int MAXP = p.maxNumPoint > 8 ? 8 : p.maxNumPoint, PP = L * MAXP;
    const int QG = N < K1_QG ? N : K1_QG;
    k1_smem sm; sm.Ls = L; sm.Sp = 8;
    double *w = wk;
    sm.qf = (double (*)[5])w; w += QG * 5;
    sm.seld = w; w += QG * L * 8;
    sm.pts = (double (*)[10])w; w += K1F_QF * PP * 10;
    sm.gram = (double (*)[45])w; w += K1F_QF * 45;
    sm.outv = (double (*)[54])w; w += K1F_QF * 54;
    int *wi = (int *)w;
    sm.cseg = (int (*)[16])wi; wi += QG * 16;
    sm.ccnt = wi; wi += QG;
    sm.seli = wi; wi += QG * L * 8;
    sm.nsel = wi; wi += QG * L;
    sm.st_s = wi;
    int st_all = 0;
    for (int i0 = 0; i0 < N; i0 += QG) {
        const int nq = N - i0 < QG ? N - i0 : QG;
        for (int e = lane; e < nq * 5; e += WAVE) {
            const int qi = e / 5, f = e % 5;
            sm.qf[qi][f] = f < 3 ? xLin_b[(size_t)(i0 + qi) * 6 + f] : uLin_b[(size_t)(i0 + qi) * 2 + (f - 3)];
        }
        for (int e = lane; e < nq * L; e += WAVE) sm.nsel[e] = 0;
        if (lane < nq) sm.st_s[lane] = 0;
        __syncthreads();
        for (int c = 0; c < L; c++) k1_scan_lap<true, K1_RPL>(p, sm, c, 0, 0, 1, nq, lane, MAXP);
        __syncthreads();
        for (int q0 = 0; q0 < nq; q0 += K1F_QF) {
            const int nf = nq - q0 < K1F_QF ? nq - q0 : K1F_QF;
            k1_fit(p, sm, q0, nf, lane, WAVE, xLin_b + (size_t)i0 * 6, MAXP);
            for (int e = lane; e < nf * 54; e += WAVE) {
                const int ql = e / 54, le = e % 54, k = i0 + q0 + ql; const double v = sm.outv[ql][le];
                if (le < 36) { AB[k * 48 + (le / 6) * 8 + le % 6] = v; if (Aout) Aout[((size_t)b * N + k) * 36 + le] = v; }
                else if (le < 48) { AB[k * 48 + ((le - 36) >> 1) * 8 + 6 + ((le - 36) & 1)] = v; if (Bout) Bout[((size_t)b * N + k) * 12 + (le - 36)] = v; }
                else { Cl[k * 6 + (le - 48)] = v; if (Cout) Cout[((size_t)b * N + k) * 6 + (le - 48)] = v; }
            }
            __syncthreads();
        }
        if (lane < nq) st_all |= sm.st_s[lane];
        __syncthreads();
    }
    return st_all;                     // per-lane partial OR (the caller folds it into the work-group's status word)
}

// parameter block staged in LDS (lane-dependent indexing of kernel arguments would go through global memory)
enum { PAR_FX = 0, PAR_FU = 12, PAR_BX = 20, PAR_BU = 22, PAR_Q2 = 26, PAR_QF2 = 62, PAR_R2 = 98, PAR_DR2 = 102, PAR_T2 = 104,
       PAR_XREF = 110, PAR_AS = 116, PAR_CS = 117, PAR_TOT = 118 };

// ---- Riccati recursion on the matrix cores, quad-block form ---------------------------------------------------------
// v_mfma_f64_4x4x4 multiplies four independent 4 x 4 blocks b: A_b[i][k] sits in lane 16 k + 4 b + i, B_b[k][j] in lane
// 16 k + 4 b + j and D_b[i][j] comes back in lane 16 i + 4 b + j (probed on gfx950, tools/mfma4_layout.hip).  An 8 x 8 matrix X
// in "quad form" is ONE register per lane: lane 16 r + 4 (2 I + J) + c holds X[4 I + r][4 J + c].  A product Z = X Y is two
// chained instructions Z(I,J) = sum_K X(I,K) Y(K,J) whose operands are in-row block copies of the quad forms (DPP with a bank
// mask): the B form of Y copies blocks [0,1,0,1] / [2,3,2,3] of Y's quad form, the A form of X copies blocks [0,0,1,1] /
// [2,2,3,3] of the quad form of X'.  The sequential recursion therefore runs register to register, with no LDS round trip and
// no barrier on the chain, and the dependent latency of this shape (~45 cycles) is less than half that of 16x16x4.
struct ricc_consts {                                       // per-lane loop-invariant data
    int qr, qR, qC, cA, oB0, oA0, oB1, oA1, oBe, oTop, qT;
    bool w_xx;
    double wq, wf0, wf1, wfu[4], d2base, ud2, idB1, idA1, idBe;
    double mR2, mBe, mXX, mU;        // 0 / 1 per lane: operand selects written as multiply-adds (one instruction instead of two v_cndmask per double)
};
__device__ __forceinline__ ricc_consts ricc_setup(int lane, const double *Q2, const double *Fx, const double *R2, const double *dR2, const double *Fu) {
    ricc_consts c;
    const int qr = lane >> 4, qI = (lane >> 3) & 1, qJ = (lane >> 2) & 1, qc = lane & 3;
    const int qR = 4 * qI + qr, qC = 4 * qJ + qc;          // tile entry (row, column) this lane owns in quad form
    c.qr = qr; c.qR = qR; c.qC = qC;
    // stage Hessian W: the xx / uu constants vanish outside their block, so W is one FMA chain without selects
    const bool w_xx = qR < 6 && qC < 6, w_uu = qR >= 6 && qC >= 6;
    c.w_xx = w_xx;
    c.wq = w_xx ? Q2[qR * 6 + qC] : 0.0;
    c.wf0 = w_xx ? Fx[qR] * Fx[qC] : 0.0; c.wf1 = w_xx ? Fx[6 + qR] * Fx[6 + qC] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) c.wfu[j] = 0.0;
    if (w_uu) {
        c.wq = R2[(qR - 6) * 2 + (qC - 6)] + (qR == qC ? dR2[qR - 6] : 0.0);
#pragma unroll
        for (int j = 0; j < 4; j++) c.wfu[j] = Fu[j * 2 + (qR - 6)] * Fu[j * 2 + (qC - 6)];
    }
    c.d2base = (qR >= 6 && qR == qC) ? dR2[qR - 6] : 0.0;                             // Base rows of u_{k-1}
    c.ud2 = (qr < 2 && qC == 6 + qr) ? -dR2[qr] : 0.0;                                // U = [M_ux | -dR]: columns 6, 7 (B form)
    // stage operands that come straight from [A_k | B_k] (row-major 6 x 8 in LDS): offsets (clamped) and identity parts
    const int cB = qC, cA = 4 * qI + qc;                                              // column read for the B form / the A form
    c.cA = cA;
    c.oB0 = qr * 8 + cB; c.oA0 = qr * 8 + cA;                                         // K = 0: rows 0..3 of Ar
    c.oB1 = (qr < 2 ? 4 + qr : 0) * 8 + cB; c.oA1 = (qr < 2 ? 4 + qr : 0) * 8 + cA;   // K = 1: rows 4, 5 of Ar; rows 6, 7 are [0 | I]
    c.idB1 = (qr >= 2 && cB == 4 + qr) ? 1.0 : 0.0; c.idA1 = (qr >= 2 && cA == 4 + qr) ? 1.0 : 0.0;
    c.oBe = (cA < 6 ? cA : 0) * 8 + 6 + (qr & 1);                                     // [B; I] as A operand (K index = qr < 2)
    c.idBe = (qr < 2 && cA == 6 + qr) ? 1.0 : 0.0;
    c.oTop = (qR < 6 ? qR : 0) * 8 + (qC < 6 ? qC : 0);
    c.qT = 4 * (16 * qc + 4 * (2 * qJ + qI) + qr);                                    // byte index of the lane holding the transposed entry
    c.mR2 = qr < 2 ? 1.0 : 0.0; c.mBe = cA < 6 ? 1.0 : 0.0; c.mXX = w_xx ? 1.0 : 0.0; c.mU = (qr < 2 && qC < 6) ? 1.0 : 0.0;
    return c;
}
// Backward recursion over the augmented stages xi_k = (x_k, u_{k-1}), [x'; u] = Ar [x; u], Ar = [[A, B], [0, I]] (8 x 8):
//   T = Pi_{k+1} Ar,  Mr = Ar' T + W_k,  eliminate u_k (2 x 2 pivot M_uu):  Pi_k = Base - U' M_uu^-1 U,  Phi_k = [[A,0],[0,0]] - [B; I] M_uu^-1 U
// executed by ONE full wave; writes Phi_k, Pi_k (row-major 8 x 8) and M_uu^-1 to LDS for the sweeps.  Returns non-zero if a pivot is not
// positive.  kap / th must be visible to the calling wave.
// STEP (multi-wave kernel): a work-group barrier closes every stage, so that follower waves can start the predictor's backward sweep one
// stage behind the recursion instead of after it.  LDS operations of one wave execute in issue order: a wave released by the barrier
// finds Phi_k, Pi_k and M_uu^-1 of the stage in place, and no s_waitcnt is needed on this side.
// (Publishing the stage through an LDS word that the followers poll, so that this wave never waits, was measured too: no difference.)
// SPLIT (one-wave kernel): rows 0..5 of Phi_k go to 6 x 8 tiles in `Phi` (scratch, read back once into the sweep registers), rows 6, 7
// (= -K_k, needed by every Newton solve) straight to their permanent place PhiK (2 x 8 per stage).
// dump: this lane's slot of >= 64 doubles of LDS nobody reads during the recursion -- the lanes that have nothing to store (60 of 64 for M_uu^-1) store
// there, so that no stage carries an exec-mask region with its branch on the critical wave (see sweep_dst).
template <int N, bool term, bool STEP = false, bool SPLIT = false, bool HASPI = true>
__device__ __forceinline__ int ricc_factor(const ricc_consts &c, const double *AB, const double *kap, const double *th, const double *Qf2,
                                           const double *PiT, double *Phi, double *PiAll, double *Mi, double *PhiK, double *dump) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int qr = c.qr;
    int bad = 0;
    double *const miDst = (lane < 4 || !SWEEP_BF<N>) ? Mi + lane : dump; const int miStride = lane < 4 ? 4 : 0;           // M_uu^-1 of stage k: lanes 0..3 -> Mi[k * 4 + lane]
    double Piq = c.w_xx ? Qf2[c.qR * 6 + c.qC] + (term ? PiT[c.qR * 6 + c.qC] : 0.0) : 0.0;   // Pi_N = [[2Qf + Pi_term, 0], [0, 0]]
    // Stage operands are fetched one stage ahead.  The twelve LDS reads of stage k - 1 are ISSUED at the top of stage k (a compiler fence
    // pins them there) and their arithmetic -- the six-term sum that builds the stage Hessian W -- runs at the bottom, behind the chain of
    // matrix-core instructions: left to itself the compiler put reads and sum together on the loop's back edge, one s_waitcnt per pair of
    // reads, ~350 stalled cycles at the start of every stage.
    double lB0, lA0, lB1, lA1, lBe, lTop, Wq;
    double wk0, wk1, wt0, wt1, wt2, wt3;
    auto stage_loads = [&](int k) {
        const double *ABk = AB + k * 48;
        lB0 = ABk[c.oB0]; lA0 = ABk[c.oA0]; lB1 = ABk[c.oB1]; lA1 = ABk[c.oA1]; lBe = ABk[c.oBe]; lTop = ABk[c.oTop];
        wk0 = kap[2 * k]; wk1 = kap[2 * k + 1]; wt0 = th[2 * N + 4 * k]; wt1 = th[2 * N + 4 * k + 1]; wt2 = th[2 * N + 4 * k + 2]; wt3 = th[2 * N + 4 * k + 3];
        asm volatile("" ::: "memory");
    };
    auto stage_hessian = [&]() {
        double w = fma(wk1, c.wf1, fma(wk0, c.wf0, c.wq));
        w = fma(wt0, c.wfu[0], w); w = fma(wt1, c.wfu[1], w); w = fma(wt2, c.wfu[2], w); w = fma(wt3, c.wfu[3], w);
        Wq = w;
    };
    stage_loads(N - 1); stage_hessian();
    int kl = N - 2;                                                                        // stage whose operands are fetched next (a scalar of its own: the
                                                                                           // clamp inside the address cost two vector instructions per read)
#pragma unroll 1
    for (int k = N - 1; k >= 0; k--) {
        const double cB0 = lB0, cA0 = lA0, cW = Wq;
        double arB1, arA1, be, top;
        EXEC_AUDIT(4);
        if constexpr (SWEEP_BF<N>) {     // (the clamped rows the other lanes read are finite: x * 0 + id is exact)
            arB1 = fma(lB1, c.mR2, c.idB1); arA1 = fma(lA1, c.mR2, c.idA1); be = fma(lBe, c.mBe, c.idBe); top = lTop * c.mXX;
        } else {
            arB1 = qr < 2 ? lB1 : c.idB1; arA1 = qr < 2 ? lA1 : c.idA1;                    // Ar (B form) and Ar' (A form), K = 1
            be = c.cA < 6 ? lBe : c.idBe;                                                  // [B; I] as A operand; its K >= 2 lanes meet K = 0
            top = c.w_xx ? lTop : 0.0;
        }
        if constexpr (SWEEP_BF<N>) { stage_loads(kl); kl = kl > 0 ? kl - 1 : 0; }
        else stage_loads(k > 0 ? k - 1 : 0);                                               // (stage 0 re-reads itself: no branch in the loop body)
        // T = Pi Ar: A form of the symmetric Pi = in-row block copies [0,0,1,1] (K = 0) and [2,2,3,3] (K = 1) of its quad form
        const double pA0 = dpp_blk<0x118, 0x8>(dpp_blk<0x114, 0x6>(Piq, Piq), Piq);       // row_shr:4 -> banks 1, 2; row_shr:8 -> bank 3
        const double pA1 = dpp_blk<0x108, 0x1>(dpp_blk<0x104, 0x6>(Piq, Piq), Piq);       // row_shl:4 -> banks 1, 2; row_shl:8 -> bank 0
        double Tq = __builtin_amdgcn_mfma_f64_4x4x4f64(pA0, cB0, 0.0, 0, 0, 0);
        Tq = __builtin_amdgcn_mfma_f64_4x4x4f64(pA1, arB1, Tq, 0, 0, 0);
        // Mr = Ar' T + W: B form of T = in-row block copies [0,1,0,1] (K = 0) and [2,3,2,3] (K = 1)
        const double tB0 = dpp_blk<0x128, 0xC>(Tq, Tq), tB1 = dpp_blk<0x128, 0x3>(Tq, Tq);
        double Mq = __builtin_amdgcn_mfma_f64_4x4x4f64(cA0, tB0, cW, 0, 0, 0);
        Mq = __builtin_amdgcn_mfma_f64_4x4x4f64(arA1, tB1, Mq, 0, 0, 0);
        // eliminate u_k: M_uu = Mr[6:8,6:8] sits in lanes 46, 47, 63; U = [M_ux | -dR] (2 x 8) = rows 6, 7 of Mr
        const double m00 = rdlane(Mq, 46), m01 = rdlane(Mq, 47), m11 = rdlane(Mq, 63);
        const double det = m00 * m11 - m01 * m01;
        if (!(det > 0.0) || !(m00 > 0.0)) bad = 1;
        const double rdet = frcp(det);
        const double i00 = m11 * rdet, i01 = -m01 * rdet, i11 = m00 * rdet;
        double sa, sb, ua, ub;
        swap32(Mq, sa, sb);                                                                // rows 6, 7 (lanes 32..63) -> lanes 0..31
        const double um = dpp_blk<0x128, 0x3>(sb, sb);                                     // columns 0..7 in both halves of the row (B form)
        const double Ub = SWEEP_BF<N> ? fma(um, c.mU, c.ud2) : (qr < 2 ? (c.qC < 6 ? um : c.ud2) : 0.0);
        swap16(Ub, ua, ub);
        const double Uo = (qr & 1) ? ua : ub;                                              // the other row of U
        // (round 4: adj(M_uu) U formed beside the reciprocal's chain and multiplied by 1 / det last, and one Newton step instead of two on the
        //  hardware estimate, were both measured: 12.77 k -> 12.50 k cycles for the twelve stages, nothing at the launch -- the extraction of U
        //  next to it is as long a chain)
        const double nKb = -((qr == 0 ? i00 : i11) * Ub + i01 * Uo);                       // -K = -M_uu^-1 U (B form; zero in lanes qr >= 2)
        const double Ua = dpp_blk<0x104, 0x4>(dpp_blk<0x114, 0x2>(Ub, Ub), Ub);            // A form of U': blocks [0,0,3,3]
        const double Bs = SWEEP_BF<N> ? fma(Mq, c.mXX, c.d2base) : (c.w_xx ? Mq : c.d2base);
        const double Piu = __builtin_amdgcn_mfma_f64_4x4x4f64(Ua, nKb, Bs, 0, 0, 0);       // Pi_k = Base - U' K
        const double Phq = __builtin_amdgcn_mfma_f64_4x4x4f64(be, nKb, top, 0, 0, 0);      // Phi_k = [[A, 0], [0, 0]] - [B; I] K
        // Rounding leaves Pi slightly unsymmetric, the A form of the next stage reads Pi', and U above takes rows for columns:
        // unsymmetrised, that asymmetry feeds back into the symmetric part at first order (it cost definiteness near convergence
        // at N = 40).  One cross-lane transpose per stage removes it.
        stage_hessian();                                                                   // W of the next stage, behind the chain above
#ifdef RICC_VAR_NOSYM                  // (tools/microbench_ricc.hip: what the transpose costs -- 80 of 934 cycles per stage)
        Piq = Piu;
#else
        Piq = 0.5 * (Piu + lane_gather(Piu, c.qT));
#endif
#ifndef RICC_VAR_NOSTORE              // (tools/microbench_ricc.hip: what the LDS stores cost -- 90 cycles per stage)
        if constexpr (SPLIT) { double *d_ = c.qR < 6 ? Phi + k * 48 + c.qR * 8 + c.qC : PhiK + k * 16 + (c.qR - 6) * 8 + c.qC; *d_ = Phq; }
        else Phi[k * 64 + c.qR * 8 + c.qC] = Phq;
        if constexpr (HASPI && SWEEP_BF<N>) PiAll[k * 64 + c.qR * 8 + c.qC] = Piq;
        else if (PiAll) PiAll[k * 64 + c.qR * 8 + c.qC] = Piq;
        if constexpr (SWEEP_BF<N>) miDst[k * miStride] = lane == 0 ? i00 : (lane == 3 ? i11 : i01);
        else { if (lane < 4) Mi[k * 4 + lane] = lane == 0 ? i00 : (lane == 3 ? i11 : i01); }
#else
        if (k == 0) { Phi[c.qR * 8 + c.qC] = Phq; if (lane < 4) Mi[lane] = i00 + i01 + i11; }
#endif
        if constexpr (STEP) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    return bad;
}

// Gram matrix of the terminal factor: W = M M' (8 x 8 row-major in Wl, row / column 7 zero; M = 8 rows x 64 CH columns, column-major in Mt, 8 per column)
// on the matrix cores with v_mfma_f64_4x4x4 -- one instruction multiplies four independent 4 x 4 blocks, here the four quadrants (I, J) of W for four
// columns of M: lane 16 k + 4 b + i supplies A = M[4 (b >> 1) + i][4 s + k] and B = M[4 (b & 1) + i][4 s + k] (operand layout: ricc_factor above), the
// quadrant entry W[4 (b >> 1) + r][4 (b & 1) + c] comes back in lane 16 r + 4 b + c.  16 CH instructions in four accumulation chains, ~16 issue cycles
// each.  (Rounds 1-3 used 16 CH v_mfma_f64_16x16x4 -- a 16 x 16 tile for an 8 x 8 result; on CDNA4 FP64 matrix work runs at the vector rate, so the
// big shape only costs: the Gram matrix was ~1.1 k of the terminal factor's ~5 k cycles.)
template <int CH> __device__ __forceinline__ void gram8_mfma(const double *Mt, double *Wl, int lane) {
    EXEC_AUDIT(5);
    const int k = lane >> 4, b = (lane >> 2) & 3, i = lane & 3;
    const double *pa = Mt + k * 8 + 4 * (b >> 1) + i, *pb = Mt + k * 8 + 4 * (b & 1) + i;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
    for (int s_ = 0; s_ < 16 * CH; s_ += 4) {
        const double a0 = pa[32 * s_], b0 = pb[32 * s_], a1 = pa[32 * (s_ + 1)], b1 = pb[32 * (s_ + 1)];
        const double a2 = pa[32 * (s_ + 2)], b2 = pb[32 * (s_ + 2)], a3 = pa[32 * (s_ + 3)], b3 = pb[32 * (s_ + 3)];
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b2, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b3, acc3, 0, 0, 0);
    }
    Wl[(4 * (b >> 1) + (lane >> 4)) * 8 + 4 * (b & 1) + (lane & 3)] = (acc0 + acc1) + (acc2 + acc3);
}

// Explicit, re-orthogonalised Q of the terminal factor (round 5).  The terminal block needs the orthogonal projector onto the row space of M = [E D^-1/2 | T^-1/2]
// (7 x (S + 6); the Newton step's lambda part is v = -c~ + Q z, Q = M' R^-1 with R'R = M M').  R comes from a Cholesky factorisation of the Gram matrix, whose
// condition number is the SQUARE of M's -- and D spans 1e-6 .. 1e11 once the active set settles: the computed Q = M' R^-1 is then orthonormal to 1e-2 only,
// the Newton directions of the last iterations carry that error, the dual residual stalls near 1e-9 and the iteration takes one more step.  That -- not the
// step rules -- was the whole difference between the kernels and their NumPy model (which orthogonalises by Gram-Schmidt): 8.69 / 13 against 8.29 / 12 iterations on
// the bench batch, 11.18 / 19 against 10.74 / 17 at N = 40 (tests/ipm_model.py: TERM_FACTOR = "gram" reproduces the kernels' counts).  Remedy (Cholesky-QR2 with a
// first-order second pass): every lane forms its row q of Q1 = M' R1^-1 (28 multiply-adds on its own column of M), the Gram matrix of Q1 -- I + E, |E| <= 2e-2 --
// comes from the same 16 matrix-core instructions as the first, and I + E = (I + U)'(I + U) + O(E^2) with U = triu(E, 1) + diag(E) / 2 corrects both factors:
// q <- q (I - U) per lane, R^-1 <- R1^-1 (I - U) (column j in lane j).  What matters is that q is kept EXPLICITLY (the lanes' mcol registers) where it multiplies z;
// R^-1 alone, however accurate, brings nothing (modelled: "cholqr2" with Q = M' R^-1: 8.63 / 13).  Model with this scheme: 8.34 / 12, 10.76 / 17.
// Gate (round 6): gap < 1e-8 (was 1e-4).  On the whole closed-loop population the pass buys nothing at all (tools/knob_model.py, TERM_FACTOR = "gram": 9.004 against 9.001
// iterations on 12 442 QPs -- distinct laps in the safe set, a well-conditioned terminal block); it pays on the bench batches (four IDENTICAL laps: duplicate columns in M),
// 8.38 against 8.52 at N = 12, 8.69 / 8.87 at N = 14, 10.65 / 10.83 at N = 40 -- and nearly all of that in the last iterations: gated at 1e-8 the model gives 8.40, 8.70, 10.68
// while the pass (3 k cycles) runs in about two iterations per solve instead of four.
#define LMPC_QX_GAP 1e-8
// mcol: this lane's columns of M on entry, of Q on exit.  Ri: R1^-1 on entry (7 x 7 row-major, zeros below the diagonal), R^-1 on exit.  Qt: 8 x 64 CH doubles of
// LDS in gram8_mfma's operand layout (may be the tile M was multiplied from), Wl: 64 doubles.  One wave; WG: work-group barrier (one-wave kernel) or waitcnt only;
// CORR = false: no LDS tile to spare for the second Gram matrix -- the routine is a no-op and the solves keep the form of rounds 1-4.
template <bool WG> __device__ __forceinline__ void term_sync() {
    if constexpr (WG) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <int CH, bool WG, bool CORR = true> __device__ __forceinline__ void term_reorth(double (&mcol)[CH][7], double *Ri, double *Qt, double *Wl, int lane, bool late) {
    // The second pass costs ~2.7 k cycles of the critical wave (+7 % per Newton iteration) and only the last iterations need it -- the loss of orthogonality grows with
    // the spread of the barrier weights.  `late` (wave-uniform; the callers pass gap < LMPC_QX_GAP) switches it on; before that the routine does nothing and the
    // Newton solves run in the form of rounds 1-4 (mcol = columns of M, v = -c~ + M' (R^-1 z7): term_omega).  Always on, it took the bench batch from 8.62 / 13 to
    // 8.30 / 12 iterations and cost the same time again (profiles/r5n_ab.txt); late only, the model counts the same 8.28 / 12.
    if (!CORR || !late) return;
    // (the 7 x 7 factors are read from LDS column by column where they are used -- uniform addresses, broadcast reads -- and fenced, so that at most seven of
    //  their entries are live at a time: held in registers whole, 2 x 28 doubles, the one-wave kernel spilled 47 vector registers to scratch)
#pragma unroll
    for (int j = 6; j >= 0; j--) {                                                 // q[j] = sum_{i <= j} R1^-1[i][j] m[i]  (descending j: in place)
        double rc[7];
#pragma unroll
        for (int i = 0; i <= j; i++) rc[i] = Ri[i * 7 + j];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            double v = rc[j] * mcol[ch][j];
#pragma unroll
            for (int i = 0; i < j; i++) v = fma(rc[i], mcol[ch][i], v);
            mcol[ch][j] = v;
        }
    }

#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
#pragma unroll
        for (int j = 0; j < 7; j++) Qt[(lane + WAVE * ch) * 8 + j] = mcol[ch][j];
        Qt[(lane + WAVE * ch) * 8 + 7] = 0.0;
    }
    term_sync<WG>();
    gram8_mfma<CH>(Qt, Wl, lane);                                                  // Q1' Q1 = I + E
    term_sync<WG>();
    {
        const int lj = lane < 7 ? lane : 0;
        double xk[7];
#pragma unroll
        for (int k = 0; k < 7; k++) xk[k] = Wl[k * 8 + lj];                        // column `lane` of I + E (lanes < 7)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 7; k++) xk[k] = k < lj ? -xk[k] : (k == lj ? 1.5 - 0.5 * xk[k] : 0.0);  // column `lane` of I - U,  U = triu(E, 1) + diag(E) / 2
        double rt[7];
#pragma unroll
        for (int i = 0; i < 7; i++) {                                              // column `lane` of R1^-1 (I - U): row i of R1^-1 against it
            double rr[7];
#pragma unroll
            for (int k = i; k < 7; k++) rr[k] = Ri[i * 7 + k];
            asm volatile("" ::: "memory");
            double v = 0.0;
#pragma unroll
            for (int k = i; k < 7; k++) v = fma(rr[k], xk[k], v);
            rt[i] = v;
        }
        if (lane < 7) {                                                            // (every read of R1^-1 above was issued before these stores: one wave, LDS in order)
#pragma unroll
            for (int i = 0; i < 7; i++) Ri[i * 7 + lane] = rt[i];
        }
    }
    // q <- q (I - U), column j of U at a time; q1 comes back from its tile (not carried in registers across the Gram matrix: the 256-register kernels spilled)
#pragma unroll
    for (int ch = 0; ch < CH; ch++)
#pragma unroll
        for (int j = 0; j < 7; j++) mcol[ch][j] = Qt[(lane + WAVE * ch) * 8 + j];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 6; j >= 0; j--) {
        double uc[7];
#pragma unroll
        for (int i = 0; i <= j; i++) uc[i] = Wl[i * 8 + j];
        asm volatile("" ::: "memory");
        const double ud = 0.5 * (uc[j] - 1.0);
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            double v = fma(-mcol[ch][j], ud, mcol[ch][j]);
#pragma unroll
            for (int i = 0; i < j; i++) v = fma(-mcol[ch][i], uc[i], v);
            mcol[ch][j] = v;
        }
    }
}

template <int N, int S> struct solve_lds {
    static constexpr int M = 8 * N + S;
    static constexpr int CH = (S + 6 + WAVE - 1) / WAVE, CW = CH * WAVE;       // terminal-block columns per lane of the wave that owns them
    static constexpr int oAB = 0, oC = oAB + 48 * N;                      // AB[k] = [A_k | B_k] (6 x 8), C_k
    static constexpr int ox = oC + 6 * N, ou = ox + 6 * (N + 1), os = ou + 2 * N, olam = os + 2 * N;
    static constexpr int odx = olam + S, odu = odx + 6 * (N + 1), ods = odu + 2 * N, odl = ods + 2 * N;
    static constexpr int onu = odl + S;
    static constexpr int om = onu + 6 * N, oth = om + M, oh = oth + M, odm = oh;            // dm overwrites h row by row (corrector post), see kernel
    static constexpr int orx = oh + M, oru = orx + 6 * (N + 1), ors = oru + 2 * N, orl = ors + 2 * N;
    static constexpr int oPhi = orl + S, oPiAll = oPhi + 64 * N, oMi = oPiAll + 64 * N, ogam = oMi + 4 * N, ogup = ogam + 8 * N,
                         opst = ogup + 2 * N, ok0 = opst + 8 * (N + 1), odnu = ogam;              // dnu reuses gamma (dead after the sweeps)
    static constexpr int okap = ok0 + 2 * N, orDs = okap + 2 * N, oeta = orDs + 2 * N, oe = oeta + 2 * N;
    static constexpr int oRi = oe + 2 * N, orsq = oRi + 56, oct = orsq + CW;
    static constexpr int oMt = oct + CW, oWl = oMt + (S > 0 ? 8 * CW : 0), oMc = oWl + (S > 0 ? 64 : 0);   // M transposed (col-major, 8 per column), Gram matrix, M c~
    static constexpr int oSS = oMc + 8, oQsel = oSS + 6 * S, oy7 = oQsel + S, oz7 = oy7 + 8, ow7 = oz7 + 8, oPiT = ow7 + 8, osT = oPiT + 36;
    static constexpr int opar = osT + 8, tot = opar + PAR_TOT;
};

// LDS layout of the ONE-WAVE kernel (lmpc_solve_kernel).  Per-QP footprint decides how many QPs a CU holds (160 KB / footprint), so
// everything with a bounded lifetime inside one Newton iteration shares a scratch region SCR:
//   t1  terminal factor   : Mt (M transposed, 8 x 64) | Wl (Gram matrix)
//   t2  stage recursion   : Phi_k, full 8 x 8 tiles (read back once into the sweep registers; rows 6, 7 are kept in PhiK)
//   t3+ Newton solves ... : h / dm | dx du ds dl | gamma (= phi = dnu) | gu' | costates p | k0 | eta | e | c~ | costate right-hand sides
// and the cost-to-go Hessians Pi_k are not stored at all (the equality multipliers follow from a backward recursion with A_k').
// N = 12, S = 48: 20.1 KB per QP = 8 QPs per CU, two waves on every SIMD (the multi-wave layout above: 39.1 KB = 4).
// Arrays that only ever meet their own lane -- C_k (monitoring residual), the residual of the lambda rows, the lambda scalings D^-1/2
// and the selected Q-function values -- live in registers (row or column = lane + 64 t), which is what brings the
// footprint under 160 KB / 7.
// ABG (long horizons): [A_k | B_k] stays in global memory (io.abPack, L2-resident: 15 KB per QP at N = 40, re-read every iteration) instead of
// 48 N doubles of LDS -- at N = 40 that is the difference between two and four QPs per CU (53.5 KB -> 38.2 KB), i.e. between two and four busy SIMDs.
template <int N, int S, bool ABG = false> struct solve_lds1 {
    static constexpr int M = 8 * N + S;
    static constexpr int CH = (S + 6 + WAVE - 1) / WAVE, CW = CH * WAVE;       // the terminal block's S + 6 columns: CH per lane (column = lane + 64 ch)
    static constexpr int oAB = 0;
    static constexpr int ox = oAB + (ABG ? 0 : 48 * N), ou = ox + 6 * (N + 1), os = ou + 2 * N, olam = os + 2 * N, onu = olam + S;
    static constexpr int oCk1 = ox;                                             // fused step only: the regression leaves C_k here (then its work space), before x .. exist
    static constexpr int om = onu, oth = om + M;                               // (the equality multipliers nu live in registers; scratch copy at residual time)
    static constexpr int orx = oth + M;
    static constexpr int oPhiK = orx + 6 * (N + 1), oMi = oPhiK + 16 * N, okap = oMi + 4 * N;
    static constexpr int oRi = okap + 2 * N, oMc = oRi + 56;
    static constexpr int oSS = oMc + 8, oy7 = oSS + 6 * S, oz7 = oy7 + 8, ow7 = oz7 + 8, oPiT = ow7 + 8, osT = oPiT + 36;
    static constexpr int opar = osT + 8, oscr = opar + PAR_TOT;
    static constexpr int oCs = oscr, oQs = oCs + 6 * N;                          // start-up only: C_k for the roll-out, Qfun_sel of the selection
    static constexpr int onus = oscr;                                           // residual phase only: nu, visible to every lane
    // scratch region, by phase
    static constexpr int oMt = oscr, oWl = oMt + 8 * CW;                                       // t1
#ifdef LMPC_DBG_NOALIAS
    static constexpr int oPhi = oscr + 8 * CW + 64;
    static constexpr int oh = oPhi + 48 * N;
#else
    static constexpr int oPhi = oscr;                                                          // t2
    static constexpr int oh = oscr;
#endif
    static constexpr int odm = oh, odx = oh + M, odu = odx + 6 * (N + 1), ods = odu + 2 * N, odl = ods + 2 * N;   // t3+
    // inside a Newton solve: c~ (read once, before the costates exist) shares the costates' place, eta (dead once gamma is formed) shares k0's;
    // after the solves the right-hand sides of the multiplier recursion take the costates' place
    static constexpr int PSTW = 8 * (N + 1) > CW ? 8 * (N + 1) : CW;
    static constexpr int ogam = odl + S, odnu = ogam, ogup = ogam + 8 * N, opst = ogup + 2 * N, ok0 = opst + PSTW;
    static constexpr int oeta = ok0, oe = ok0 + 2 * N, oct = opst, ott = opst, oend3 = oe + 2 * N;
    static constexpr int scr0 = 6 * N + S, scr1 = S > 0 ? 8 * CW + 64 : 0, scr2 = 48 * N, scr3 = oend3 - oscr;
    static_assert(scr0 <= scr3, "start-up staging fits the Newton-solve scratch");
    static constexpr int scr = scr1 > scr2 ? (scr1 > scr3 ? scr1 : scr3) : (scr2 > scr3 ? scr2 : scr3);
#ifdef LMPC_DBG_NOALIAS
    static constexpr int tot = oend3;
#else
    static constexpr int tot = oscr + scr;
#endif
};

#define FOR_LANES(idx, n) for (int idx = lane; idx < (n); idx += WAVE)
// same trips with the trip number t as a compile-time index (register arrays: element idx = lane + 64 t lives in slot t of its lane)
#define FOR_LANES_T(idx, t, n) _Pragma("unroll") for (int t = 0, idx = lane; t < ((n) + WAVE - 1) / WAVE; t++, idx += WAVE) if (idx < (n))

// A value every lane of the wave holds (a wave reduction, a broadcast read): moved to scalar registers.  The vector register file is what
// limits the one-wave kernel (256 per wave at two waves per SIMD); a wave-uniform double kept in vector registers costs two of them.
__device__ __forceinline__ double wave_uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// Loop-exit decisions of the one-wave kernel (round 5).  Every quantity the Newton loop branches on -- gap, residual maxima, "a pivot was not positive" -- is the same
// in all 64 lanes at run time (wave reductions, broadcast reads), but the compiler cannot know that of a value that lives in a vector register: a `break` on it
// becomes a DIVERGENT loop exit -- exec-mask bookkeeping around the whole iteration, a structurised latch block every exit is routed through, live-range splits of
// the loop-carried register arrays around it.  LMPC_UX(v) moves the value to scalar registers (v_readfirstlane) where the decision is taken: scalar compare,
// scalar branch, no mask.  See DESIGN.md, "N = 40".
#ifndef LMPC_NO_UNIFORM_EXIT
#define LMPC_UX(v) wave_uniform(v)
#define LMPC_UXI(v) __builtin_amdgcn_readfirstlane(v)
#else
#define LMPC_UX(v) (v)
#define LMPC_UXI(v) (v)
#endif

// sum over the 8 lanes of a group (lane = 8 g + c, all lanes of the group receive it) / over the 8 groups (same c)
template <bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double sum_over_c(double v) {
    EXEC_AUDIT(1);
    v += dpp_mv<DPP_QP_X1, BC>(v); v += dpp_mv<DPP_QP_X2, BC>(v); v += dpp_mv<DPP_HALF_MIRROR, BC>(v);
    return v;
}
template <bool BC = LMPC_BC_DEFAULT> __device__ __forceinline__ double sum_over_g(double v) {
    EXEC_AUDIT(2);
    v += dpp_mv<0x128, BC>(v);                   // row_ror:8  (lane c <-> c+8 inside a row of 16)
    double a, b;
    swap16(v, a, b); v = a + b;
    swap32(v, a, b); v = a + b;
    return v;
}


// Branch-free stores of the register sweeps.  A sweep stage ends with ONE row of 8 lanes holding the stage's 8 results (the lanes with lc == 0 in a
// "sum over c" stage, those with lg == 0 in a "sum over g" stage).  Written as `if (writer) { if (idx < 6) dx[..] = v; else du[..] = v; }` that is two
// nested exec-mask regions with their branches and spilled-mask reloads per stage, ON the dependent chain of the sweep (~60 of ~200 cycles per stage).
// Instead every lane stores unconditionally: the writers to their element, the others to a per-lane slot of `dump` -- LDS that is dead during the
// sweep (the array the sweep's own per-stage operands were just read from).  destination(k) = p + k * s, for even / odd stages.
typedef __attribute__((address_space(3))) double lds_f64;    // explicit LDS pointers: 32-bit, ds_write whatever the optimiser can or cannot infer
struct sweep_dst { double *wE, *wO; int dE, dO; };          // destination of the NEXT even / odd stage and its (signed) advance per use, in doubles
// (the pointers are advanced stage by stage and pinned with an empty asm: computed as base + k * stride the compiler forms all N addresses ahead of the
//  unrolled loop -- 2 N live registers, spills at N = 40)
#define SWEEP_PIN(a, b) asm volatile("" : "+v"(a), "+v"(b))
template <int N> __device__ __forceinline__ double *sweep_dump(double *region8N, int lane) {
    constexpr int DN = 8 * N >= WAVE ? WAVE : 16;               // (N >= 2: at least 16 doubles; a few lanes may share a slot, which is harmless)
    return region8N + (lane & (DN - 1));
}
// forward sweep xi_{k+1} = Phi_k xi_k + phi_k, k = 0 .. N-1: even stages sum over c (writers lc == 0, element lg), odd stages over g (writers lg == 0,
// element lc); element < 6 -> dx[(k + 1) * 6 + element], else du[k * 2 + element - 6]
template <int N> __device__ __forceinline__ sweep_dst fwd_sweep_dst(double *dx, double *du, double *dump8N, int lane, int lg, int lc) {
    double *d = sweep_dump<N>(dump8N, lane);
    const int sE = lc == 0 ? (lg < 6 ? 6 : 2) : 0, sO = lg == 0 ? (lc < 6 ? 6 : 2) : 0;
    sweep_dst w;
    w.wE = lc == 0 ? (lg < 6 ? dx + 6 + lg : du + (lg - 6)) : d; w.dE = 2 * sE;                  // stage 0, then 2, 4, ...
    w.wO = (lg == 0 ? (lc < 6 ? dx + 6 + lc : du + (lc - 6)) : d) + sO; w.dO = 2 * sO;          // stage 1, then 3, 5, ...
    return w;
}
// backward sweep p_k = Phi_k' p_{k+1} + gamma_k, k = N-1 .. 0: odd stages sum over c (writers lc == 0, pst[k * 8 + lg]), even stages over g (writers
// lg == 0, pst[k * 8 + lc])
template <int N> __device__ __forceinline__ sweep_dst bwd_sweep_dst(double *pst, double *dump8N, int lane, int lg, int lc) {
    double *d = sweep_dump<N>(dump8N, lane);
    constexpr int kO = ((N - 1) & 1) ? N - 1 : N - 2, kE = ((N - 1) & 1) ? N - 2 : N - 1;       // first odd / even stage of the sweep
    const int sO = lc == 0 ? 8 : 0, sE = lg == 0 ? 8 : 0;
    sweep_dst w;
    w.wO = (lc == 0 ? pst + lg : d) + kO * sO; w.dO = -2 * sO;
    w.wE = (lg == 0 ? pst + lc : d) + kE * sE; w.dE = -2 * sE;
    return w;
}

// r[j] = sum_c SS[j][c] v[c] - sub[j] for j < 6, using lanes (j, part) = (lane>>3, lane&7): S/8 terms each + group reduction
template <int S> __device__ __forceinline__ void ss_times(const double *SS, const double *v, const double *sub, double *out, int lane) {
    const int j = lane >> 3, part = lane & 7;
    double acc = 0.0;
    if (j < 6) {
#pragma unroll
        for (int c = part; c < S; c += 8) acc = fma(SS[j * S + c], v[c], acc);
    }
    acc = sum_over_c(acc);
    if (j < 6 && part == 0) out[j] = acc - sub[j];
}

// Products with the terminal factor R^-1 = Ri (upper triangular 7 x 7, row-major in LDS, zeros below the diagonal) by one wave: lane
// (i, j) = (lane >> 3, lane & 7) forms one product and the 8 lanes of a group are summed by DPP, so every lane of group i < 7 receives
// element i of the result -- one LDS round trip and three DPP steps instead of a 7-term chain of broadcast reads on 7 lanes.
__device__ __forceinline__ double ri_t_times(const double *Ri, const double *b, int lg, int lc) {     // (Ri' b)[lg] = sum_{j <= lg} Ri[j][lg] b[j]
    const bool on = lg < 7 && lc <= lg;
    const double v = on ? Ri[(on ? lc : 0) * 7 + (on ? lg : 0)] * b[on ? lc : 0] : 0.0;
    return sum_over_c(v);
}
__device__ __forceinline__ double ri_times(const double *Ri, const double *b, int lg, int lc) {       // (Ri b)[lg] = sum_{j >= lg} Ri[lg][j] b[j]
    const bool on = lg < 7 && lc >= lg && lc < 7;
    const double v = on ? Ri[(on ? lg : 0) * 7 + (on ? lc : 0)] * b[on ? lc : 0] : 0.0;
    return sum_over_c(v);
}

// Roll-out x_{k+1} = A_k x_k + C_k (u = 0) that gives the interior-point iteration its start, by one wave, in registers: lane c < 6 carries
// x_k[c], the six entries reach every lane as wave-uniform values (v_readlane), the rows of A_k and C_k are independent loads the
// compiler issues ahead.  Same multiply-add order as the LDS form (12 round trips and barriers), ~14 k cycles less per solve.
template <int N> __device__ __forceinline__ void rollout_start(const double *AB, const double *Cg, double *x, int lane) {
    const int c = lane < 6 ? lane : 0;
    double xc = x[c];                                         // x_0 (written by the caller, visible)
#pragma unroll
    for (int k = 0; k < N; k++) {
        double v = Cg[k * 6 + c];
#pragma unroll
        for (int j = 0; j < 6; j++) v = fma(AB[k * 48 + c * 8 + j], rdlane(xc, j), v);
        xc = v;
        if (lane < 6) x[(k + 1) * 6 + lane] = v;
    }
}

// Terminal part of a Newton solve without LDS round trips: omega' = Ri (Ri' d7 + y7), d7 = (dx_N ; -re_sum) -- or, late in the iteration, z7 = Ri' d7 + y7 itself.  The forward sweep ends (for
// even N: its last stage sums over the groups) with every lane (lg, lc) holding xi_N[lc], which is d7[lc] for lc < 6: the product with
// Ri' is formed from that register, z7 reaches lane (lg, lc) as element lc through the LDS crossbar (ds_bpermute from the first lane of
// group lc), and the seven results leave as wave-uniform values (v_readlane).  Same products and the same summation order as the
// ri_t_times / ri_times pair on LDS copies, hence bit-identical; ~0.7 k cycles less per solve on the critical wave.
__device__ __forceinline__ void term_omega(const double *Ri, double yv /* y7[lg] */, double xiN, double re_sum, int lg, int lc, double (&w)[7], bool late) {
    const double b_ = lc < 6 ? xiN : (lc == 6 ? -re_sum : 0.0);
    const bool on1 = lg < 7 && lc <= lg;
    const double r1 = Ri[(on1 ? lc : 0) * 7 + (on1 ? lg : 0)];
    const double zv = sum_over_c(on1 ? r1 * b_ : 0.0) + yv;                       // z7[lg], in every lane of group lg
    double ov = zv;
    if (!late) {                                                                  // (wave-uniform) omega' = Ri z7, for v = -c~ + M' omega'
        const double zt = lane_gather(zv, 32 * lc);                               // z7[lc]
        const bool on2 = lg < 7 && lc >= lg && lc < 7;
        const double r2 = Ri[(on2 ? lg : 0) * 7 + (on2 ? lc : 0)];
        ov = sum_over_c(on2 ? r2 * zt : 0.0);
    }
    // late: z7 itself leaves -- the lanes hold their rows of Q = M' R^-1 explicitly by then (term_reorth) and form v = -c~ + Q z7
#pragma unroll
    for (int j = 0; j < 7; j++) w[j] = rdlane(ov, 8 * j);
}

// (the condensed kernel, lmpc_solve_cd.hip.h, keeps the rounds 1-4 form: omega' = Ri z7 for v = -c~ + M' omega')
__device__ __forceinline__ void term_omega_w(const double *Ri, double yv, double xiN, double re_sum, int lg, int lc, double (&w)[7]) {
    const double b_ = lc < 6 ? xiN : (lc == 6 ? -re_sum : 0.0);
    const bool on1 = lg < 7 && lc <= lg;
    const double r1 = Ri[(on1 ? lc : 0) * 7 + (on1 ? lg : 0)];
    const double zv = sum_over_c(on1 ? r1 * b_ : 0.0) + yv;
    const double zt = lane_gather(zv, 32 * lc);
    const bool on2 = lg < 7 && lc >= lg && lc < 7;
    const double r2 = Ri[(on2 ? lg : 0) * 7 + (on2 ? lc : 0)];
    const double wv = sum_over_c(on2 ? r2 * zt : 0.0);
#pragma unroll
    for (int j = 0; j < 7; j++) w[j] = rdlane(wv, 8 * j);
}

// Terminal costate of a Newton solve, same scheme: enters with mc = (M c~)[lg] in every lane of group lg (the DPP sum that formed it),
// returns [Ri (Ri' d0 + y7)][lg], d0 = (0, -re_sum), y7 = Ri' (M c~), in every lane of group lg, and y7[lg] itself for term_omega.
__device__ __forceinline__ double term_costate(const double *Ri, double mc, double re_sum, int lg, int lc, double &yv) {
    const bool on1 = lg < 7 && lc <= lg;
    const double r1 = Ri[(on1 ? lc : 0) * 7 + (on1 ? lg : 0)], r6 = Ri[6 * 7 + (lg < 7 ? lg : 0)];
    const double b1 = lane_gather(mc, 32 * lc);                                   // (M c~)[lc]
    yv = sum_over_c(on1 ? r1 * b1 : 0.0);                                         // y7[lg]
    const double zq = fma(r6, -re_sum, yv);
    const double b2 = lane_gather(zq, 32 * lc);
    const bool on2 = lg < 7 && lc >= lg && lc < 7;
    const double r2 = Ri[(on2 ? lg : 0) * 7 + (on2 ? lc : 0)];
    return sum_over_c(on2 ? r2 * b2 : 0.0);
}

// Terminal factor, small part, one row per lane (used by the condensed kernel, lmpc_solve_cd.hip.h): W7 = M M' (7 x 7, symmetric, 8 x 8 row-major in
// LDS) -> Ri = R^-1 with R'R = W7 (upper triangular, 7 x 7 row-major in LDS, zeros below the diagonal).  Lane i < 7 holds row i of L7 = R' (Cholesky
// W7 = L7 L7', pivot column broadcast by v_readlane), then row i of X = L7^-1 (forward elimination on the identity); Ri = X'.  Returns non-zero if a
// pivot is not positive.  (In the Riccati kernels the wave-uniform form below measured better: there this version costs 28 VGPRs of a full file.)
__device__ __forceinline__ int term_factor7(const double *Wl, double *Ri, int lane) {
    const int r7 = lane < 7 ? lane : 0;
    int bad = 0;
    double wr[7], xr[7], rd7 = 1.0;
#pragma unroll
    for (int j = 0; j < 7; j++) { wr[j] = (lane < 7 && j <= lane) ? Wl[r7 * 8 + j] : 0.0; xr[j] = lane == j ? 1.0 : 0.0; }
#pragma unroll
    for (int j = 0; j < 7; j++) {
        double d_ = rdlane(wr[j], j);
        if (!(d_ > 0.0)) { bad = 1; d_ = 1.0; }
        const double ri = frsqrt(d_);
        const double lij = lane == j ? d_ * ri : ((lane > j && lane < 7) ? wr[j] * ri : 0.0);
        wr[j] = lij; rd7 = lane == j ? ri : rd7;
#pragma unroll
        for (int k = j + 1; k < 7; k++) wr[k] = fma(-lij, rdlane(lij, k), wr[k]);
    }
#pragma unroll
    for (int k = 0; k < 7; k++) {                                     // row k of X is final once scaled by 1 / L[k][k]; the rows below subtract L[i][k] times it
#pragma unroll
        for (int c = 0; c <= k; c++) {
            const double xk = rdlane(xr[c] * rd7, k);
            xr[c] = lane == k ? xk : ((lane > k && lane < 7) ? fma(-wr[k], xk, xr[c]) : xr[c]);
        }
    }
    if (lane < 7) {
#pragma unroll
        for (int i = 0; i < 7; i++) Ri[i * 7 + lane] = xr[i];          // Ri[i][j] = X[j][i]
    }
    return bad;
}

// ------------------------------------------------------------------------------------------------
// K2: safe-set selection.  LMPC.addTerminalComponents :392-412 and selectPoints :478-514.  Shared by the one-wave and the
// multi-wave solve kernels: wave `wave` of NW handles laps wave, wave + NW, ...; results go to SS (6 x S, row-major), Qsel (S),
// sel_start (window start per lap, kept for the successor rows of feasibleStateInput) and the optional global outputs.
// ------------------------------------------------------------------------------------------------
template <int N, int S, int NW>
__device__ __forceinline__ void k2_select(const lmpc_dev_params &p, const lmpc_solve_io &io, int b, int lane, int wave, double *SS, double *Qsel,
                                          int *sel_start, int *st_sh) {
    if (io.mode & 1) {
        double ztv[6];
#pragma unroll
        for (int j = 0; j < 6; j++) ztv[j] = io.zt[(size_t)b * 6 + j];
        const double x04 = io.x0[(size_t)b * 6 + 4];
        if (ztv[4] - x04 > p.TL / 2) ztv[4] = fmax(ztv[4] - p.TL, 0.0);        // :392-393
        if (io.ztUsed && wave == 0 && lane < 6) { double v = ztv[0];
#pragma unroll
            for (int j = 1; j < 6; j++) if (lane == j) v = ztv[j];
            io.ztUsed[(size_t)b * 6 + lane] = v; }
        const int hasPred = io.hasPred ? io.hasPred[b] : 0;                     // Q-function shift bookkeeping (:502-512)
        int crossed = 0;
        if (hasPred) {
#pragma unroll
            for (int r0 = 0; r0 <= N; r0 += WAVE) {                               // N + 1 predicted rows: 65 at the largest horizon, one more than a wave has lanes
                const int r = r0 + lane;
                const int c_ = (r <= N && io.xPredPrev[((size_t)b * (N + 1) + (r <= N ? r : 0)) * 6 + 4] > p.TL) ? 1 : 0;
                crossed += (int)__popcll(__ballot(c_));
            }
        }
        const int tstep = io.timeStep ? io.timeStep[b] : 0;
        const int ppl = p.ppl, npw = ppl + 1;                                   // numSS_Points/numSS_it + 1 (=13)
        for (int l = wave; l < p.L; l += NW) {
            const double *base = p.sstore + (size_t)p.sslot[l] * LMPC_COLS * p.lap_stride;
            const int T = p.sslen[l], ls = p.lap_stride;
            double best = INFINITY; int bi = 0x7fffffff;
            for (int r = lane; r < T; r += WAVE) {
                double nrm = fabs(base[r] - ztv[0]);                            // la.norm(x - zt, 1, axis=1)
                nrm = nrm + fabs(base[ls + r] - ztv[1]);
                nrm = nrm + fabs(base[2 * ls + r] - ztv[2]);
                nrm = nrm + fabs(base[3 * ls + r] - ztv[3]);
                nrm = nrm + fabs(base[4 * ls + r] - ztv[4]);
                nrm = nrm + fabs(base[5 * ls + r] - ztv[5]);
                if (nrm < best) { best = nrm; bi = r; }
            }
            wave_argmin(best, bi);                                              // np.argmin: first minimum
            if ((unsigned)bi >= (unsigned)T) {                                  // no row compares below +inf: zt or x0 is not finite (np.argmin would go on with row 0 or the first NaN).
                bi = 0; if (lane == 0) atomicOr(st_sh, LMPC_ST_NUMERIC);        // The window arithmetic below must not see the sentinel (start + lane overflowed: a fault on the device)
            }
            const int MinNorm = bi;
            const int start = ((double)MinNorm - (double)npw / 2.0 >= 0.0) ? MinNorm - npw / 2 : MinNorm;   // :492-495
            if (lane == 0) { sel_start[l] = start; if (io.selStartOut) io.selStartOut[(size_t)b * p.L + l] = start; if (start + npw > T) atomicOr(st_sh, LMPC_ST_WINDOW); }
            double shift = 0.0;                                                 // :502-512
            if (hasPred && crossed > 0) {
                if (p.sslapid[l] < p.cur_it - 1) shift = base[8 * ls];
                else shift = (double)tstep + (double)(N - crossed);
            }
            if (lane < ppl) {
                int r0 = start + lane; r0 = r0 > T - 1 ? T - 1 : r0;
                int r1 = start + lane + 1; r1 = r1 > T - 1 ? T - 1 : r1;
                const int col = l * ppl + lane;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const double v = base[j * ls + r0];
                    SS[j * S + col] = v;
                    if (io.ssSelOut) io.ssSelOut[((size_t)b * S + col) * 6 + j] = v;
                    if (io.succOut) io.succOut[((size_t)b * S + col) * 6 + j] = base[j * ls + r1];
                }
                if (io.succUOut) { io.succUOut[((size_t)b * S + col) * 2] = base[6 * ls + r1]; io.succUOut[((size_t)b * S + col) * 2 + 1] = base[7 * ls + r1]; }
                const double qv = base[8 * ls + r0] + shift;
                Qsel[col] = qv;
                if (io.qSelOut) io.qSelOut[(size_t)b * S + col] = qv;
            }
        }
    } else {
        for (int c = wave * WAVE + lane; c < S; c += NW * WAVE) {
#pragma unroll
            for (int j = 0; j < 6; j++) SS[j * S + c] = io.ssSelIn[((size_t)b * S + c) * 6 + j];
            Qsel[c] = io.qSelIn[(size_t)b * S + c];
        }
    }
}

template <int N, int S, bool EQ = false, bool ABG = false>
// (two waves per SIMD -- at most 256 registers -- only where the LDS footprint lets more than four QPs share a CU and the terminal
// block keeps one column per lane)
#ifdef LMPC_AB_OCC3     // (developer A / B, tools/resource_usage.py: what a third resident wave per SIMD -- at most 168 registers -- would cost this kernel; profiles/r6_onewave_occ3_resource_usage.txt)
__global__ __launch_bounds__(WAVE, 3) void lmpc_solve_kernel(lmpc_dev_params p, int B, lmpc_solve_io io) {
#else
__global__ __launch_bounds__(WAVE, (solve_lds1<N, S, ABG>::tot * 8 * 5 <= 160 * 1024 && solve_lds1<N, S, ABG>::CH == 1) ? 2 : 1) void lmpc_solve_kernel(lmpc_dev_params p, int B, lmpc_solve_io io) {
#endif
    extern __shared__ double sm[];
    using LL = solve_lds1<N, S, ABG>;
    constexpr int M = LL::M;
    constexpr bool term = S > 0;
    constexpr int RPL = (M + WAVE - 1) / WAVE;              // inequality rows per lane
    constexpr int CH = LL::CH;                              // terminal-block columns per lane (1 for numSS_points <= 58)
    const int b = blockIdx.x;
    if (b >= B) return;
    // retry variant (EQ): only problems that hit the iteration limit (or broke down numerically far from the optimum) run again,
    // with equal primal / dual steps throughout and a neighbourhood safeguard on the step length
    if constexpr (EQ) { if (!(io.status[b] & (LMPC_ST_MAXITER | LMPC_ST_NUMERIC))) return; }
    const int lane = threadIdx.x;
    const int lg = lane >> 3, lc = lane & 7;                // lane = 8 g + c  (8 x 8 tile coordinates)
    double *x = sm + LL::ox, *u = sm + LL::ou, *s = sm + LL::os, *lam = sm + LL::olam;
    double *AB;                                            // [A_k | B_k], 6 x 8 per stage: LDS, or (ABG) this problem's slice of io.abPack in global memory
    if constexpr (ABG) AB = io.abPack + (size_t)b * 48 * N; else AB = sm + LL::oAB;
    double *dx = sm + LL::odx, *du = sm + LL::odu, *ds = sm + LL::ods, *dl = sm + LL::odl, *nu = sm + LL::onus, *dnu = sm + LL::odnu;
    double *m = sm + LL::om, *th = sm + LL::oth, *h = sm + LL::oh, *dm = sm + LL::odm;
    double *rx = sm + LL::orx;
    double *Phi = sm + LL::oPhi, *PhiK = sm + LL::oPhiK, *Mi = sm + LL::oMi, *gam = sm + LL::ogam, *gup = sm + LL::ogup, *pst = sm + LL::opst, *k0 = sm + LL::ok0;
    double *tt = sm + LL::ott;                             // right-hand sides of the costate recursion
    double *phi = gam;                                     // gamma is dead (kept in registers) once the backward sweep starts
    double *kap = sm + LL::okap, *eta = sm + LL::oeta, *ee = sm + LL::oe;
    double *Ri = sm + LL::oRi, *ct = sm + LL::oct, *Mt = sm + LL::oMt, *Wl = sm + LL::oWl, *McL = sm + LL::oMc;
    double *SS = sm + LL::oSS, *Qsel = sm + LL::oQs, *y7 = sm + LL::oy7, *z7 = sm + LL::oz7, *w7 = sm + LL::ow7, *PiT = sm + LL::oPiT, *sT = sm + LL::osT;
    double *par = sm + LL::opar;
    constexpr int T2N = (2 * N + WAVE - 1) / WAVE, T6N = (6 * N + WAVE - 1) / WAVE;
    double c_r[T6N], ru_r[T2N], rs_r[T2N], rDs_r[T2N], rl_r[CH], qsel_r[CH];      // own-lane arrays (element lane + 64 t in slot t)
    const double *Fx = par + PAR_FX, *Fu = par + PAR_FU, *bx = par + PAR_BX, *bu = par + PAR_BU, *Q2 = par + PAR_Q2, *Qf2 = par + PAR_QF2,
                 *R2 = par + PAR_R2, *dR2 = par + PAR_DR2, *T2p = par + PAR_T2, *xRef = par + PAR_XREF;
    __shared__ int st_sh;
    __shared__ int sel_start[LMPC_MAX_USED_LAPS];
    int tcnt = 0; (void)tcnt;
    TSTAMP(0);
    // (retry variant: the regression's per-point bits are taken over from the first pass's status word -- the per-point buffer may belong to
    //  a later launch by the time a deferred retry pass runs)
    if (lane == 0) st_sh = EQ ? (io.status[b] & (LMPC_ST_REG_SINGULAR | LMPC_ST_NO_SEGMENT)) : 0;
    static_assert(!ABG || !EQ, "the retry variant keeps [A_k | B_k] in LDS");
    if (!ABG && (io.mode & 4)) {
        // K1 (fused step): LTV regression of this QP's N points by this wave, in the LDS behind AB / C that the solve needs only later
        __syncthreads();
        const int rst = k1_wave_problem(p, b, lane, io.xLin + (size_t)b * (N + 1) * 6, io.uLin + (size_t)b * N * 2, sm + LL::oCk1 + 6 * N, AB, sm + LL::oCk1,
                                        io.Aout, io.Bout, io.Cout);
        if (rst) atomicOr(&st_sh, rst);
        __syncthreads();
        FOR_LANES_T(i, t, 6 * N) c_r[t] = sm[LL::oCk1 + i];
        __syncthreads();
    }
    // stage the parameter block
    if (lane < 12) par[PAR_FX + lane] = p.Fx[lane];
    if (lane < 8) par[PAR_FU + lane] = p.Fu[lane];
    if (lane < 2) { par[PAR_BX + lane] = p.bx[lane]; par[PAR_DR2 + lane] = p.dR2[lane]; }
    if (lane < 4) { par[PAR_BU + lane] = p.bu[lane]; par[PAR_R2 + lane] = p.R2[lane]; }
    if (lane < 36) { par[PAR_Q2 + lane] = p.Q2[lane]; par[PAR_QF2 + lane] = p.Qf2[lane]; }
    if (lane < 6) { par[PAR_T2 + lane] = p.T2[lane]; par[PAR_XREF + lane] = p.xRef[lane]; }
    if (lane == 0) { par[PAR_AS] = p.a_s; par[PAR_CS] = p.c_s; }
    __syncthreads();
    const double a_s = wave_uniform(par[PAR_AS]), c_s = wave_uniform(par[PAR_CS]);

    // A_k, B_k, C_k of this problem: the global loads are issued here, into registers, so that their latency runs beside the lap scans of
    // the selection; they reach LDS afterwards
    constexpr int TA = (36 * N + WAVE - 1) / WAVE, TB = (12 * N + WAVE - 1) / WAVE;
    double preA[TA], preB[TB];
    const bool preload = (io.mode & 2) && !(io.mode & 4);
    if (preload) {
        FOR_LANES_T(i, t, 36 * N) preA[t] = io.A[(size_t)b * 36 * N + i];
        FOR_LANES_T(i, t, 12 * N) preB[t] = io.Bm[(size_t)b * 12 * N + i];
        FOR_LANES_T(i, t, 6 * N) c_r[t] = io.C[(size_t)b * 6 * N + i];
    }
    // K2: safe-set selection (k2_select), then the regression status bits of this problem's N points
    if constexpr (term) { k2_select<N, S, 1>(p, io, b, lane, 0, SS, Qsel, sel_start, &st_sh); __syncthreads(); }
    if (!EQ && io.rstatus && lane < N) { const int rs_ = io.rstatus[(size_t)b * N + lane]; if (rs_) atomicOr(&st_sh, rs_); }
    TSTAMP(1);
    if (!(io.mode & 2)) { if (lane == 0) io.status[b] = st_sh; return; }

    // ------------------------------------------------------------------------------------------------
    // K3: QP solve.  Variables z = [x_0..x_N | u_0..u_{N-1} | s (2N) | lambda (S) | s_T (6)] exactly as
    // LMPC.unpackSolution (:364-375); inequality rows in the reference's order (buildIneqConstr :166-198,
    // addSafeSetIneqConstr :340-343).  s_T is eliminated (s_T = SS lambda - x_N).
    // ------------------------------------------------------------------------------------------------
    if (preload) {
        FOR_LANES_T(i, t, 36 * N) { const int k = i / 36, r = (i % 36) / 6, c = i % 6; AB[k * 48 + r * 8 + c] = preA[t]; }
        FOR_LANES_T(i, t, 12 * N) { const int k = i / 12, r = (i % 12) >> 1, c = i & 1; AB[k * 48 + r * 8 + 6 + c] = preB[t]; }
        if constexpr (ABG) __threadfence();                   // (global scratch written and read by this wave only: make the stores visible to its later loads)
    }
    double *Cs = sm + LL::oCs;                             // C_k for the roll-out below (scratch; afterwards C lives in c_r only)
    FOR_LANES_T(i, t, 6 * N) Cs[i] = c_r[t];
    if constexpr (term) { FOR_LANES_T(c, t, S) qsel_r[t] = Qsel[c]; }
    if (lane < 6) x[lane] = io.x0[(size_t)b * 6 + lane];
    FOR_LANES(i, 2 * N) u[i] = 0.0;
    const double uOld0 = wave_uniform(io.uOld[(size_t)b * 2 + 0]), uOld1 = wave_uniform(io.uOld[(size_t)b * 2 + 1]);
    __syncthreads();
    rollout_start<N>(AB, Cs, x, lane);                     // strictly interior start: u = 0, x by roll-out
    __syncthreads();
    // Lane slacks: a row the roll-out violates starts one unit inside (s = violation + 1); every other slack starts at 1 / c_s, where
    // the multiplier of s >= 0 (mu0 / s) already balances the slack's linear cost c_s -- from s = 1 the first Newton steps were spent
    // driving ~2N slacks to zero against the positivity bound (11.0 -> 9.5 iterations on the bench batch, 12 -> 6 on the LTV-MPC QPs)
    const double s_init = c_s > 1.0 ? 1.0 / c_s : 1.0;
    FOR_LANES(i, 2 * N) {
        const int k = i >> 1, j = i & 1; double f = 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], x[k * 6 + c], f);
        const double viol = f - bx[j];
        s[i] = viol > 0.0 ? viol + 1.0 : s_init;
    }
    double qmax = 0.0;
    if constexpr (term) { FOR_LANES_T(c, t, S) { lam[c] = 1.0 / (double)S; qmax = fmax(qmax, fabs(qsel_r[t])); } qmax = wmax(qmax); }
    const double mu0 = fmax(1.0, 0.01 * (term ? qmax : 1.0));
    if (lane < 4 && !(bu[lane] > 0.0)) atomicOr(&st_sh, LMPC_ST_NOT_INTERIOR);
    double eta_m = 0.0;
    __syncthreads();

    // F_r . (x,u,s,lam) and b_r for inequality row r in the reference's ordering
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

    double t_r[RPL], tp_r[RPL], dt_r[RPL];                 // per-lane row state (row = lane + 64 j); 1 / t is recomputed where needed (registers)
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const int r = lane + WAVE * j;
        t_r[j] = 1.0; tp_r[j] = 0.0; dt_r[j] = 0.0;
        if (r < M) { const double tt = rowb(r) - rowF(r, x, u, s, lam); t_r[j] = tt; m[r] = mu0 / tt; }
    }
    double ph[N];                                          // Phi_k entry this lane multiplies with in the register sweeps
    double mcol[CH][7];                                    // this lane's columns of M = [E D^-1/2 | T7^-1/2] (column lane + 64 ch) while the terminal block is factorised,
                                                           // then this lane's rows of Q = M' R^-1 (term_reorth)
    double rsl[CH];                                        // D^-1/2 of this lane's lambda columns (0 for the slack columns)
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
#pragma unroll
        for (int j = 0; j < 7; j++) mcol[ch][j] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < N; k++) ph[k] = 0.0;
    __syncthreads();

    bool qx_late = false;                                  // the terminal factor of this iteration carries the explicit, re-orthogonalised Q (term_reorth)
    // one Newton-system solve for the right-hand side currently in (rx,ru,rs,rl,h); result in dx,du,ds,dl
    auto kkt_solve = [&](double re_sum) {
        FOR_LANES_T(i, t, 2 * N) {                              // slack elimination, per lane row (k,j)
            const double hl = h[i], hs = h[6 * N + i];
            const double e_ = -(rs_r[t] + hl + hs);
            ee[i] = e_; eta[i] = hl + th[i] * e_ * rDs_r[t];
        }
        FOR_LANES_T(i, t, 2 * N) {                              // gu' = ru - Fu' h_u
            const int k = i >> 1, c = i & 1; double v = ru_r[t];
#pragma unroll
            for (int j = 0; j < 4; j++) v -= Fu[j * 2 + c] * h[2 * N + 4 * k + j];
            gup[i] = v;
        }
        double c_t[CH], xiN = 0.0, mc_g = 0.0, y7v = 0.0;       // xiN: last stage of the forward sweep as the lanes hold it; (M c~)[lg], y7[lg]
        if constexpr (term) {
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const int col = lane + WAVE * ch;
                c_t[ch] = 0.0;
                if (col < S) c_t[ch] = (rl_r[ch] + h[8 * N + col]) * rsl[ch];
                ct[col] = col < S ? c_t[ch] * rsl[ch] : 0.0;      // D^-1/2 c~ (the slack columns of M meet zeros of c~)
            }
        }
        __syncthreads();
        if constexpr (term) {
            // M c~ = [SS; 1'] (D^-1/2 c~): lane (j, part) adds every 8th column, the 8 lanes of a group are summed with DPP.
            // (M itself -- the Mt tile of the terminal factor -- is gone by now: its LDS holds the Newton-solve temporaries.)
            double acc = 0.0;
            if (lg < 7) {
#pragma unroll
                for (int c = lc; c < S; c += 8) acc = lg < 6 ? fma(SS[lg * S + c], ct[c], acc) : acc + ct[c];
            }
            acc = sum_over_c(acc);
            mc_g = lg < 7 ? acc : 0.0;                          // (M c~)[lg] in every lane of group lg
        }
        FOR_LANES(i, 8 * N) {                                   // gamma_k = [gx' - Kx' gu' ; -Ku' gu'] = [gx';0] + Phi[6:8,:]' gu'
            const int k = i >> 3, c = i & 7;
            double v = 0.0;
            if (c < 6) v = -(Fx[c] * eta[2 * k] + Fx[6 + c] * eta[2 * k + 1]);       // (the x rows of the dual residual vanish identically: exact nu)
            v = fma(PhiK[k * 16 + c], gup[2 * k], v);
            v = fma(PhiK[k * 16 + 8 + c], gup[2 * k + 1], v);
            gam[i] = v;
        }
        double pN;                                              // terminal costate p_N = (rx_N + [Ri (Ri' d0 + y7)]_{0:6}, 0), element lg in every lane of group lg
        {
            double v = 0.0;
            if constexpr (term) v = term_costate(Ri, mc_g, re_sum, lg, lc, y7v);                 // (in registers: no LDS round trip, no barrier)
            pN = lg < 6 ? v : 0.0;
            if (lc == 0) pst[N * 8 + lg] = pN;                  // (k0 of the last stage reads it from LDS, two barriers from here)
        }
        __syncthreads();
        TSTAMP(30);
        {   // backward sweep p_k = Phi_k' p_{k+1} + gamma_k, entirely in registers:
            // stage k odd : lanes hold p[c], multiply by Phi_k[c][g], sum over c -> p_k[g]
            // stage k even: lanes hold p[g], multiply by Phi_k[g][c], sum over g -> p_k[c]
            double gm[N];
#pragma unroll
            for (int k = 0; k < N; k++) gm[k] = (k & 1) ? gam[k * 8 + lg] : gam[k * 8 + lc];
            double pv = ((N - 1) & 1) ? lane_gather(pN, 32 * lc) : pN;
            if constexpr (SWEEP_BF<N>) {
                const sweep_dst wd = bwd_sweep_dst<N>(pst, gam, lane, lg, lc); lds_f64 *wE = (lds_f64 *)wd.wE, *wO = (lds_f64 *)wd.wO;       // (gamma is in the gm registers: its LDS takes the non-writers' stores)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = N - 1; k >= 0; k--) {
                    double pr = ph[k] * pv;
                    if (k & 1) { pr = sum_over_c<true>(pr); pv = pr + gm[k]; *wO = pv; wO += wd.dO; }
                    else { pr = sum_over_g<true>(pr); pv = pr + gm[k]; *wE = pv; wE += wd.dE; }
                    SWEEP_PIN(wE, wO);
                }
            } else {
#pragma unroll
                for (int k = N - 1; k >= 0; k--) {
                    double pr = ph[k] * pv;
                    if (k & 1) { pr = sum_over_c(pr); pv = pr + gm[k]; if (lc == 0) pst[k * 8 + lg] = pv; }
                    else { pr = sum_over_g(pr); pv = pr + gm[k]; if (lg == 0) pst[k * 8 + lc] = pv; }
                }
            }
        }
        __syncthreads();
        TSTAMP(31);
        FOR_LANES(i, 2 * N) {                                   // k0_k = Mi_k (gu' + B' p_x + p_u)
            const int k = i >> 1, c = i & 1;
            double w0 = gup[2 * k] + pst[(k + 1) * 8 + 6], w1 = gup[2 * k + 1] + pst[(k + 1) * 8 + 7];
            if constexpr (ABG) {            // (global loads: all six pairs in flight before the first multiply-add -- see the adjoint recursion)
                double b6[6], b7[6];
#pragma unroll
                for (int j = 0; j < 6; j++) { b6[j] = AB[k * 48 + j * 8 + 6]; b7[j] = AB[k * 48 + j * 8 + 7]; }
                asm volatile("" : "+v"(b6[0]), "+v"(b6[1]), "+v"(b6[2]), "+v"(b6[3]), "+v"(b6[4]), "+v"(b6[5]), "+v"(b7[0]), "+v"(b7[1]), "+v"(b7[2]), "+v"(b7[3]), "+v"(b7[4]), "+v"(b7[5]));
#pragma unroll
                for (int j = 0; j < 6; j++) { w0 = fma(b6[j], pst[(k + 1) * 8 + j], w0); w1 = fma(b7[j], pst[(k + 1) * 8 + j], w1); }
            } else {
#pragma unroll
                for (int j = 0; j < 6; j++) { w0 = fma(AB[k * 48 + j * 8 + 6], pst[(k + 1) * 8 + j], w0); w1 = fma(AB[k * 48 + j * 8 + 7], pst[(k + 1) * 8 + j], w1); }
            }
            k0[i] = Mi[k * 4 + c * 2] * w0 + Mi[k * 4 + c * 2 + 1] * w1;
        }
        __syncthreads();
        FOR_LANES(i, 8 * N) {                                   // phi_k = [-B k0 ; -k0]
            const int k = i >> 3, c = i & 7;
            phi[i] = c < 6 ? -(AB[k * 48 + c * 8 + 6] * k0[2 * k] + AB[k * 48 + c * 8 + 7] * k0[2 * k + 1]) : -k0[2 * k + (c - 6)];
        }
        if (lane < 6) dx[lane] = 0.0;
        __syncthreads();
        TSTAMP(32);
        {   // forward sweep xi_{k+1} = Phi_k xi_k + phi_k, xi_k = (dx_k, du_{k-1}), in registers:
            // stage k even: lanes hold xi[c], multiply by Phi_k[g][c], sum over c -> xi'[g]
            // stage k odd : lanes hold xi[g], multiply by Phi_k[c][g], sum over g -> xi'[c]
            double fm[N];
#pragma unroll
            for (int k = 0; k < N; k++) fm[k] = (k & 1) ? phi[k * 8 + lc] : phi[k * 8 + lg];
            double xi = 0.0;
            if constexpr (SWEEP_BF<N>) {
                const sweep_dst wd = fwd_sweep_dst<N>(dx, du, phi, lane, lg, lc); lds_f64 *wE = (lds_f64 *)wd.wE, *wO = (lds_f64 *)wd.wO;    // (phi is in the fm registers)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < N; k++) {
                    double pr = ph[k] * xi;
                    if (k & 1) pr = sum_over_g<true>(pr); else pr = sum_over_c<true>(pr);
                    xi = pr + fm[k];
                    if (k & 1) { *wO = xi; wO += wd.dO; } else { *wE = xi; wE += wd.dE; }
                    SWEEP_PIN(wE, wO);
                }
            } else {
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
        TSTAMP(33);
        FOR_LANES_T(i, t, 2 * N) {
            const int k = i >> 1, j = i & 1; double f = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], dx[k * 6 + c], f);
            ds[i] = (th[i] * f + ee[i]) * rDs_r[t];
        }
        if constexpr (term) {
            double wq[7];
            if constexpr ((N - 1) & 1) term_omega(Ri, y7v, xiN, re_sum, lg, lc, wq, qx_late);         // (even N: every lane ends the sweep with xi_N[lc])
            else {   // z7 = Ri' d7 + y7, d7 = (dx_N ; -re_sum);  omega' = Ri z7
                if (lane < 7) w7[lane] = lane < 6 ? dx[N * 6 + lane] : -re_sum;               // d7 (w7 is free until omega' is written)
                __syncthreads();
                const double zv = ri_t_times(Ri, w7, lg, lc) + y7v;
                __syncthreads();
                if (lc == 0 && lg < 7) z7[lg] = zv;
                __syncthreads();
                if (!qx_late) {
                    const double wv = ri_times(Ri, z7, lg, lc);
                    __syncthreads();
                    if (lc == 0 && lg < 7) z7[lg] = wv;
                    __syncthreads();
                }
#pragma unroll
                for (int j = 0; j < 7; j++) wq[j] = z7[j];
            }
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const int col = lane + WAVE * ch;
                double v = -c_t[ch];                            // v = -c~ + M' omega'  (late: -c~ + Q z7, Q explicit: term_reorth)
#pragma unroll
                for (int j = 0; j < 7; j++) v = fma(mcol[ch][j], wq[j], v);
                if (col < S) dl[col] = v * rsl[ch];
            }
        }
        __syncthreads();
    };

    auto dyn_residual = [&]() -> double {                 // max |x_{k+1} - A_k x_k - B_k u_k - C_k| of the iterate in LDS
        double remax = 0.0;
        FOR_LANES_T(i, t, 6 * N) {
            const int k = i / 6, c = i % 6;
            double v = x[(k + 1) * 6 + c] - c_r[t] - AB[k * 48 + c * 8 + 6] * u[k * 2] - AB[k * 48 + c * 8 + 7] * u[k * 2 + 1];
#pragma unroll
            for (int j = 0; j < 6; j++) v -= AB[k * 48 + c * 8 + j] * x[k * 6 + j];
            remax = fmax(remax, fabs(v));
        }
        return wmax(remax);
    };
    int it = 0, converged = 0, sep = 0;                   // sep: separate primal/dual step lengths after a poor-progress iteration
    double gap = 0.0, rdn = 0.0, ren = 0.0, gap_prev = -1.0;
    double step_prev = INFINITY, step_pp = INFINITY;         // max-norm of the last two (x, u) steps taken (step_bound_ok)

    const double qscale = wave_uniform(fmax(1.0, qmax));                // dual residual tolerance is relative to the cost scale
#pragma unroll 1
    for (it = 0; it <= p.max_iter; it++) {
        TSTAMP(10);
        // ---- slacks of the inequality rows, terminal slack, residuals --------------------------------
        double gsum = 0.0, rmax = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j;
            if (r < M) gsum = fma(t_r[j], m[r], gsum);      // (the row slacks are iterates of their own: t <- t + alpha dt at the step, never b - F w)
        }
        if constexpr (term) ss_times<S>(SS, lam, x + N * 6, sT, lane);
        __syncthreads();
        // Multipliers of the dynamics rows (round 3: not iterates of their own any more).  The x rows of the dual residual read
        //   w_k + nu_{k-1} - A_k' nu_k = 0,  w_k = 2Q (x_k - xRef) + Fx' mu_k  (k < N),  w_N = 2Qf (x_N - xRef) - T s_T,
        // so nu follows from the iterate by an adjoint recursion, nu_{N-1} = -w_N, nu_{k-1} = A_k' nu_k - w_k: the rows vanish identically, the Newton
        // direction in (u, s, lambda, mu) is the one of the condensed form (tests/ipm_model.py: exact_nu), and the costate recursion for d nu
        // (6.1 k of 64 k cycles per iteration) is gone.  w goes through the rx array; lane c < 6 carries nu_k[c], entries broadcast by v_readlane.
        FOR_LANES(i, 6 * N) {
            const int k = i / 6 + 1, c = i % 6;
            const double *Qk = k < N ? Q2 : Qf2;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) v = fma(Qk[c * 6 + j], x[k * 6 + j] - xRef[j], v);
            if (k < N) v += Fx[c] * m[2 * k] + Fx[6 + c] * m[2 * k + 1];
            else if (term) v -= T2p[c] * sT[c];
            rx[k * 6 + c] = v;
        }
        __syncthreads();
        {
            const int c = lane < 6 ? lane : 0;
            double nc = -rx[N * 6 + c];                                          // nu_{N-1}[c]
            if (lane < 6) nu[(N - 1) * 6 + lane] = nc;
            if constexpr (ABG) {
                // [A_k | B_k] in global memory: the six loads of a stage are issued TOGETHER, one stage ahead of the multiply-adds that use them (two register sets,
                // two stages per trip).  Left to the compiler the loop came out in one of two forms depending on unrelated code -- six loads into one register
                // pair, each waited for (234 exposed cache round trips per iteration: 1.27 ms per launch at batch 1024), or the six in flight together (1.04 ms).
                auto ld6 = [&](int k, double (&d)[6]) {
                    const double *Ak = AB + k * 48 + c;
#pragma unroll
                    for (int r = 0; r < 6; r++) d[r] = Ak[r * 8];
                };
                auto st1 = [&](int k, const double (&a)[6]) {
                    double v = -rx[k * 6 + c];
#pragma unroll
                    for (int r = 0; r < 6; r++) v = fma(a[r], rdlane(nc, r), v);
                    nc = v;
                    if (lane < 6) nu[(k - 1) * 6 + lane] = v;
                };
                double a0[6], a1[6];
                ld6(N - 1, a0);
                int k = N - 1;
#pragma unroll 1
                for (; k >= 2; k -= 2) {
                    ld6(k - 1, a1); st1(k, a0);
                    ld6(k > 2 ? k - 2 : 1, a0); st1(k - 1, a1);
                }
                if (k == 1) st1(1, a0);
            } else if constexpr (N > 20) {
                // long horizons: a rolled loop (unrolled, the 6 N loads were hoisted together: 281 spilled VGPRs at N = 40)
#pragma unroll 1
                for (int k = N - 1; k >= 1; k--) {
                    double v = -rx[k * 6 + c];
                    const double *Ak = AB + k * 48 + c;
#pragma unroll
                    for (int r = 0; r < 6; r++) v = fma(Ak[r * 8], rdlane(nc, r), v);
                    nc = v;
                    if (lane < 6) nu[(k - 1) * 6 + lane] = v;
                }
            } else {
#pragma unroll
                for (int k = N - 1; k >= 1; k--) {
                    double v = -rx[k * 6 + c];
#pragma unroll
                    for (int r = 0; r < 6; r++) v = fma(AB[k * 48 + r * 8 + c], rdlane(nc, r), v);
                    nc = v;
                    if (lane < 6) nu[(k - 1) * 6 + lane] = v;
                }
            }
        }
        __syncthreads();
        FOR_LANES_T(i, t, 2 * N) {
            const int k = i >> 1, c = i & 1;
            const double up = k > 0 ? u[(k - 1) * 2 + c] : (c == 0 ? uOld0 : uOld1);
            double v = R2[c * 2] * u[k * 2] + R2[c * 2 + 1] * u[k * 2 + 1] + dR2[c] * (u[i] - up);
            if (k < N - 1) v += dR2[c] * (u[i] - u[(k + 1) * 2 + c]);
#pragma unroll
            for (int j = 0; j < 4; j++) v = fma(Fu[j * 2 + c], m[2 * N + 4 * k + j], v);
#pragma unroll
            for (int j = 0; j < 6; j++) v -= AB[k * 48 + j * 8 + 6 + c] * nu[k * 6 + j];
            ru_r[t] = v; rmax = fmax(rmax, fabs(v));
            const double vs = a_s * s[i] + c_s - m[i] - m[6 * N + i];
            rs_r[t] = vs; rmax = fmax(rmax, fabs(vs));
        }
        double lsum = 0.0;
        if constexpr (term) {
            FOR_LANES_T(c, t, S) {
                double v = qsel_r[t] - m[8 * N + c] + eta_m;
#pragma unroll
                for (int j = 0; j < 6; j++) v = fma(SS[j * S + c], T2p[j] * sT[j], v);
                rl_r[t] = v; rmax = fmax(rmax, fabs(v)); lsum += lam[c];
            }
        }
        gap = wave_uniform(SWEEP_BF<N> ? wsum(gsum) * (1.0 / (double)M) : wsum(gsum) / (double)M);    // (short horizons: no IEEE divisions in the loop, ~30 instructions each)
        rdn = LMPC_UX(wmax(rmax));
        const double re_sum = term ? wave_uniform(wsum(lsum) - 1.0) : 0.0;
        TRACE3(lane == 0 && !EQ, 0, gap, rdn, fabs(re_sum));        // (the first pass only: a retry pass would overwrite the rows of the problem it re-runs)
        // The dynamics rows are linear and every step keeps them (the roll-out start satisfies them, the Newton direction lies in their null
        // space): their residual only collects rounding, ~1e-13.  It is still checked -- wherever a decision depends on it (convergence,
        // the INEXACT classification) -- but no longer in the iterations whose other two residuals have not passed yet.
        if (gap < p.tol_gap && rdn < p.tol_res * qscale && step_bound_ok(step_prev, step_pp)) {
            ren = LMPC_UX(fmax(dyn_residual(), fabs(re_sum)));
            if (ren < p.tol_res) { converged = 1; break; }
        }
        if (gap_prev >= 0.0) sep = !EQ && gap > LMPC_SEP_THRESHOLD * gap_prev;
        gap_prev = gap;
        if (it == p.max_iter) { ren = LMPC_UX(fmax(dyn_residual(), fabs(re_sum))); break; }
        if (!(gap == gap) || !(rdn == rdn)) { if (lane == 0) atomicOr(&st_sh, LMPC_ST_NUMERIC); break; }

        TSTAMP(11);
        // ---- factorisation of the Newton (block-banded KKT) matrix -------------------------------------
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) th[r] = m[r] * barrier_rt(t_r[j], m[r]); }
        __syncthreads();
        FOR_LANES_T(i, t, 2 * N) {
            const double d_ = frcp(a_s + th[i] + th[6 * N + i]);
            rDs_r[t] = d_; kap[i] = th[i] * (a_s + th[6 * N + i]) * d_;
        }
        int numeric_bad = 0;
        if constexpr (term) {
            // terminal block: M = [E D^-1/2 | T7^-1/2] (7 x (S+6)), one column per lane; W = M M' by wave reductions,
            // R'R = W (Cholesky) and Ri = R^-1 in registers (uniform across lanes); all later uses apply Ri / Ri' in factored form
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const int col = lane + WAVE * ch;
#pragma unroll
                for (int j = 0; j < 7; j++) mcol[ch][j] = 0.0;
                // (no per-lane if / else inside the Newton loop: both kinds of column are formed with clamped indices and selected -- an `else` is a flow block,
                //  and flow blocks are where the compiler fault of isa_check.py lives)
                const bool islam = col < S, isslk = col >= S && col < S + 6;
                const int cl = islam ? col : 0, cs = isslk ? col - S : 0;
                const double rs_ = frsqrt(th[8 * N + cl] + p.reg), tsq = frsqrt(T2p[cs]);       // lambda column: D^-1/2;  slack column: T^-1/2
#pragma unroll
                for (int j = 0; j < 6; j++) mcol[ch][j] = islam ? SS[j * S + cl] * rs_ : ((isslk && cs == j) ? tsq : 0.0);
                mcol[ch][6] = islam ? rs_ : 0.0; rsl[ch] = mcol[ch][6];
            }
            double Rr[7][7], rinv[7];
            // Gram matrix W = M M' (7 x 7, K = 64 CH columns) on the matrix cores (gram8_mfma)
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
#pragma unroll
                for (int j = 0; j < 7; j++) Mt[(lane + WAVE * ch) * 8 + j] = mcol[ch][j];
                Mt[(lane + WAVE * ch) * 8 + 7] = 0.0;
            }
            __syncthreads();
            // (rounds 1-3 used v_mfma_f64_16x16x4 -- a 16 x 16 tile for an 8 x 8 result; round 4 kept that form at N = 40 and with six columns per lane because "every
            //  problem ran into the iteration limit" with the 4x4x4 form there: the compiler fault isa_check.py now guards against, not the Gram matrix)
            gram8_mfma<CH>(Mt, Wl, lane);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 7; i++)
#pragma unroll
                for (int j = i; j < 7; j++) Rr[i][j] = Wl[i * 8 + j];                     // W (upper), uniform broadcast reads
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
            __syncthreads();
            qx_late = gap < LMPC_QX_GAP;
            term_reorth<CH, true>(mcol, Ri, Mt, Wl, lane, qx_late);       // (late iterations) rows of Q explicit and re-orthogonalised; Ri <- R^-1 of the corrected factor (M's tile is free: Q1 goes there)
            __syncthreads();
            if (lane < 36) {                                     // Pi_term = (Ri Ri')[0:6,0:6]
                const int i = lane / 6, j = lane % 6; double v = 0.0;
                for (int k = (i > j ? i : j); k < 7; k++) v = fma(Ri[i * 7 + k], Ri[j * 7 + k], v);
                PiT[lane] = v;
            }
            __syncthreads();
        }
        TSTAMP(12);
        {   // the ~40 registers of per-lane recursion constants are rebuilt every iteration instead of being carried through the whole loop
            // (the opaque copy of `lane` keeps the compiler from hoisting them): carried, they were spilled to scratch at the loop entry
            int l2 = lane; asm volatile("" : "+v"(l2));
            const ricc_consts rc = ricc_setup(l2, Q2, Fx, R2, dR2, Fu);
            numeric_bad |= ricc_factor<N, term, false, true, false>(rc, AB, kap, th, Qf2, PiT, Phi, (double *)nullptr, Mi, PhiK,
                                                                    rx + (lane & (6 * (N + 1) >= WAVE ? WAVE - 1 : 7)));    // (w, the adjoint recursion's input, is dead by now: dump)
        }
        // a breakdown of the factorisation once the iterate is optimal to working accuracy (gap at its floor, residuals small: the
        // barrier weights span > 1e26 there) is reported as INEXACT, not as a failure: the iterate whose residuals were just measured is returned
        numeric_bad = LMPC_UXI(numeric_bad);
        if (numeric_bad) { ren = LMPC_UX(fmax(dyn_residual(), fabs(re_sum))); if (lane == 0) atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_NUMERIC); break; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; k++) {                                            // entry (r, c_) of Phi_k: rows 0..5 in the scratch tiles, rows 6, 7 (= -K_k) in PhiK
            const int r = (k & 1) ? lc : lg, c_ = (k & 1) ? lg : lc;
            ph[k] = r < 6 ? Phi[k * 48 + r * 8 + c_] : PhiK[k * 16 + (r - 6) * 8 + c_];
        }
        __syncthreads();                                                         // the tiles are dead from here on: the region is reused

        TSTAMP(13);
        // ---- predictor (affine scaling) direction: h = mu -------------------------------------------------
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) h[r] = t_r[j] * th[r]; }      // t mu rt: mu itself wherever the weight is not capped
        __syncthreads();
        kkt_solve(re_sum);
        TSTAMP(14);
        double apmax = 1.0, admax = 1.0, dma_r[RPL];        // separate primal / dual step lengths (equal steps can stall)
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j; dma_r[j] = 0.0;
            if (r < M) {
                const double dta = -rowF(r, dx, du, ds, dl), mr = m[r];
                const double dma = -h[r] - th[r] * dta;
                dt_r[j] = dta; dma_r[j] = dma;
                if (dta < 0.0) apmax = fmin(apmax, -t_r[j] * frcp(dta));
                if (dma < 0.0) admax = fmin(admax, -mr * frcp(dma));
            }
        }
        apmax = wmin(apmax); admax = wmin(admax);
        if (!sep) { apmax = fmin(apmax, admax); admax = apmax; }
        double gaff = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j;
            if (r < M) { gaff = fma(t_r[j] + apmax * dt_r[j], m[r] + admax * dma_r[j], gaff); tp_r[j] = dt_r[j] * dma_r[j]; }
        }
        gaff = SWEEP_BF<N> ? wsum(gaff) * (1.0 / (double)M) : wsum(gaff) / (double)M;
        double sig = SWEEP_BF<N> ? gaff * frcp(gap) : gaff / gap; sig = centring_sigma(sig);
        const double tgt = fmax(sig * gap, 0.01 * p.tol_gap);   // keep the complementarity products off the rounding floor
        // ---- corrector: h = (t mu - sigma gap + dt_aff dmu_aff) / t ----------------------------------------
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) { const double mr = m[r]; h[r] = (fma(t_r[j], mr, LMPC_SO_W * tp_r[j]) - tgt) * barrier_rt(t_r[j], mr); } }
        __syncthreads();
        TSTAMP(15);
        kkt_solve(re_sum);
        TSTAMP(16);
        double apx = INFINITY, adx = INFINITY;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j;
            if (r < M) {
                const double dtt = -rowF(r, dx, du, ds, dl), mr = m[r];
                const double dmm = -h[r] - th[r] * dtt;
                dm[r] = dmm; dt_r[j] = dtt;
                if (dtt < 0.0) apx = fmin(apx, -t_r[j] * frcp(dtt));
                if (dmm < 0.0) adx = fmin(adx, -mr * frcp(dmm));
            }
        }
        apx = wmin(apx); adx = wmin(adx);
        const double frac = EQ ? 0.995 : step_fraction(sig, gap);
        double al = fmin(1.0, frac * apx), ald = fmin(1.0, frac * adx);
        if (!sep) { al = fmin(al, ald); ald = al; }
        if constexpr (EQ) {
            // retry variant only: stay in a wide neighbourhood of the central path -- shorten the step until every complementarity
            // product keeps at least 1 % of the mean.  It costs the well-behaved problems half an iteration (so the first pass does not
            // do it) and is what gets the remaining stalled ones through (NumPy model: 11-12 iterations instead of a stall).
            for (int trial = 0; trial < 8; trial++) {
                double pmin = INFINITY, psum = 0.0;
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    const int r = lane + WAVE * j;
                    if (r < M) { const double pr = (t_r[j] + al * dt_r[j]) * (m[r] + ald * dm[r]); pmin = fmin(pmin, pr); psum += pr; }
                }
                pmin = LMPC_UX(wmin(pmin)); psum = LMPC_UX(wsum(psum));
                if (pmin >= 1e-2 * psum / (double)M) break;
                al *= 0.7; ald *= 0.7;
            }
        }
        TSTAMP(17);
        TRACE3(lane == 0 && !EQ, 3, sig, al, ald);
        // ---- step.  (The multipliers of the dynamics rows are recomputed from the new iterate by the adjoint recursion at the top of the loop:
        //      no costate recursion here any more.) -----------------------------------------------------------------------------------
        if constexpr (term) {
            ss_times<S>(SS, dl, dx + N * 6, w7, lane);                                                // d s_T
        }
        __syncthreads();
        double deta = 0.0;                                     // multiplier of sum(lambda) = 1: mean over the lambda rows of  -rl + dmu - SS' T ds_T
        if constexpr (term) {
            double v = 0.0;
            FOR_LANES_T(c, t, S) { v += -rl_r[t] + dm[8 * N + c];
#pragma unroll
                for (int j = 0; j < 6; j++) v -= SS[j * S + c] * T2p[j] * w7[j]; }
            deta = SWEEP_BF<N> ? wsum(v) * (1.0 / (double)S) : wsum(v) / (double)S;
        }
        {                                                      // (the step, and the length of its (x, u) part for step_bound_ok)
            double smax = 0.0;
            FOR_LANES(i, 6 * (N + 1)) { smax = fmax(smax, fabs(al * dx[i])); x[i] = fma(al, dx[i], x[i]); }
            FOR_LANES(i, 2 * N) { smax = fmax(smax, fabs(al * du[i])); u[i] = fma(al, du[i], u[i]); s[i] = fma(al, ds[i], s[i]); }
            step_pp = step_prev; step_prev = LMPC_UX(wmax(smax));
        }
        if constexpr (term) { FOR_LANES(c, S) lam[c] = fma(al, dl[c], lam[c]); }
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) { m[r] = fma(ald, dm[r], m[r]); t_r[j] = fma(al, dt_r[j], t_r[j]); } }
        TSTAMP(18);
        eta_m = wave_uniform(fma(ald, deta, eta_m));
        __syncthreads();
    }
    TSTAMP(20);
#ifdef LMPC_TRACE
    if (!EQ && io.tbuf && !(io.mode & 4)) {
        // last trace row: what the termination test of the final iteration saw -- the dynamics residual as the kernel evaluates it, and the same residual with
        // C_k and [A_k | B_k] re-read from the kernel's INPUTS (io.C, io.A, io.Bm) instead of the register / scratch copies the iteration has carried
        double e_c = 0.0, e_ab = 0.0, r_in = 0.0;
        FOR_LANES_T(i, t, 6 * N) {
            const int k = i / 6, c = i % 6;
            const double cg = io.C[(size_t)b * 6 * N + i];
            e_c = fmax(e_c, fabs(c_r[t] - cg));
            double v = x[(k + 1) * 6 + c] - cg - io.Bm[((size_t)b * N + k) * 12 + c * 2] * u[k * 2] - io.Bm[((size_t)b * N + k) * 12 + c * 2 + 1] * u[k * 2 + 1];
            for (int j = 0; j < 6; j++) { const double ag = io.A[((size_t)b * N + k) * 36 + c * 6 + j]; v -= ag * x[k * 6 + j]; e_ab = fmax(e_ab, fabs(ag - AB[k * 48 + c * 8 + j])); }
            e_ab = fmax(e_ab, fmax(fabs(io.Bm[((size_t)b * N + k) * 12 + c * 2] - AB[k * 48 + c * 8 + 6]), fabs(io.Bm[((size_t)b * N + k) * 12 + c * 2 + 1] - AB[k * 48 + c * 8 + 7])));
            r_in = fmax(r_in, fabs(v));
        }
        e_c = wmax(e_c); e_ab = wmax(e_ab); r_in = wmax(r_in);
        const double r_k = dyn_residual();
        if (lane == 0) { double *tr_ = (double *)io.tbuf + ((size_t)b * LMPC_TRACE_ROWS + LMPC_TRACE_ROWS - 1) * 6; tr_[0] = r_k; tr_[1] = r_in; tr_[2] = e_c; tr_[3] = e_ab; tr_[4] = (double)it; tr_[5] = (double)converged; }
    }
#endif
    if (!converged && lane == 0 && !(st_sh & (LMPC_ST_NUMERIC | LMPC_ST_INEXACT)))
        atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_MAXITER);
    if (!p.slacks) {
        // hard lane rows (MPCParams.slacks = False) are imposed through slack variables with a 1e12 quadratic weight: at the optimum a slack is
        // mu / 2e12 ~ 1e-11 unless the hard problem is infeasible -- then it is the violation itself, and the reference's solver reports
        // "primal infeasible" (feasible = 0, PredictiveControllers.py:277-280)
        double smax = 0.0;
        FOR_LANES(i, 2 * N) smax = fmax(smax, s[i]);
        smax = wmax(smax);
        if (smax > 1e-8 && lane == 0) atomicOr(&st_sh, LMPC_ST_INFEASIBLE);
    }
    __syncthreads();

    // ---- unpackSolution (:364-379) and feasibleStateInput (:382-384) -------------------------------------
    FOR_LANES(i, 6 * (N + 1)) io.xPred[(size_t)b * 6 * (N + 1) + i] = x[i];
    FOR_LANES(i, 2 * N) { io.uPred[(size_t)b * 2 * N + i] = u[i]; if (io.slack) io.slack[(size_t)b * 2 * N + i] = s[i]; }
    if (io.mu) FOR_LANES(r, M) io.mu[(size_t)b * M + r] = m[r];
    if constexpr (term) {
        if (io.lambda) FOR_LANES(c, S) io.lambda[(size_t)b * S + c] = lam[c];
        if (lane < 6 && io.sTerm) {
            double v = -x[N * 6 + lane];
            for (int c = 0; c < S; c++) v = fma(SS[lane * S + c], lam[c], v);
            io.sTerm[(size_t)b * 6 + lane] = v;
        }
        if ((io.mode & 1) && (io.ztNext || io.ztuNext)) {
            // zt = Succ_SS lambda, zt_u = Succ_uSS lambda: successor rows are read back from the store (window starts kept)
            double acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.0;
            FOR_LANES(c, S) {
                const int l = c / p.ppl, cc = c % p.ppl;
                const double *base = p.sstore + (size_t)p.sslot[l] * LMPC_COLS * p.lap_stride;
                int r1 = sel_start[l] + cc + 1; r1 = r1 > p.sslen[l] - 1 ? p.sslen[l] - 1 : r1;
                const double lv = lam[c];
#pragma unroll
                for (int j = 0; j < 8; j++) acc[j] = fma(base[j * p.lap_stride + r1], lv, acc[j]);
            }
            // reference order of np.dot(Succ, lambd): sequential over columns; a tree sum differs by rounding only
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = wsum(acc[j]);
            if (lane < 6 && io.ztNext) { double v = acc[0];
#pragma unroll
                for (int j = 1; j < 6; j++) if (lane == j) v = acc[j];
                io.ztNext[(size_t)b * 6 + lane] = v; }
            if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = lane == 0 ? acc[6] : acc[7];
        }
    } else {
        if (lane < 6 && io.ztNext) io.ztNext[(size_t)b * 6 + lane] = x[N * 6 + lane];          // MPC.feasibleStateInput :157-159
        if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = u[(N - 1) * 2 + lane];
    }
    TSTAMP(21);
    if (lane == 0) {
        io.status[b] = st_sh; io.iters[b] = it; flag_retry(io, st_sh);
        if (io.resid) { io.resid[(size_t)b * 3] = gap; io.resid[(size_t)b * 3 + 1] = rdn; io.resid[(size_t)b * 3 + 2] = ren; }
    }
}


#ifndef LMPC_VARIANT_TU
// =====================================================================================================
// K4: plant integrator + closed-loop bookkeeping for device-resident batched rollouts.
// One thread per rollout (the plant is 100 dependent Euler sub-steps of ~60 flops: no intra-rollout parallelism).
// =====================================================================================================
// Simulator.dynModel, SysModel.py:56-147.  x, xg: (B,6) curvilinear / global state; u: (B,2); nz: (B,3) N(0,1) draws.
// TWO adjacent lanes integrate one car: the sub-step's transcendental chains come in pairs that run the same instructions
// on different data -- front / rear slip angle and tyre force (atan2, atan, sin), heading psi / heading error epsi
// (sin, cos) -- so lane `role` 0 takes the front tyre and psi, lane 1 the rear tyre and epsi, they swap the three results
// with one DPP quad_perm each, and both carry the full state redundantly.  Every transcendental is still evaluated exactly
// once per sub-step with the same argument as in the scalar form.
//
// Round 5: range-specialised FP64 kernels instead of the ocml calls.  The 100 Euler sub-steps are one dependent chain through the tyre forces
// (vx, vy, wz -> slip angle -> force -> vx, vy, wz; the kinematic states only integrate them), and with atan2 -> atan -> sin from ocml that chain was
// ~3.8 k cycles per sub-step (general-purpose routines: IEEE divisions, quadrant and special-value handling, 20-term Horner chains, the large-argument
// path of sin) -- 0.16 ms per simulated step on the critical path of the closed loop.  The arguments are bounded by the physics: vx > 0 and |slip| < 45 degrees
// (|y / vx| <= 1), |C atan(.)| <= 1.25 pi / 4 < 1, headings a few laps' worth of 2 pi.  So: quotient by reciprocal + one correction, atan(z) = z P(z^2) on
// |z| <= 1 (23 coefficients), sin(x) = x Q(x^2) on |x| <= 1 (10), sin / cos by a two-constant Cody-Waite reduction (exact under FMA for |x| < 1e5) and
// 7-term kernels on |r| <= pi / 4 -- all evaluated by Estrin's scheme (depth log2 n instead of n).  Coefficients and measured accuracy:
// tools/fit_plant_polys.py (<= 3.8e-16 relative, i.e. < 2 ulp; the oracle's libm is < 1 ulp) -- after 100 sub-steps of dt = 1e-3 that is 1e-17 in the state,
// against the 1e-12 the parity test states.  A lane whose argument leaves a fast range (a diverged rollout) takes the ocml routine: same results as before there.
template <int NC> __device__ __forceinline__ double estrin(const double (&c)[NC], double w) {
    double v[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) v[i] = c[i];
    double pw = w;
#pragma unroll
    for (int n = NC; n > 1; n = (n + 1) / 2) {
#pragma unroll
        for (int i = 0; i < (n + 1) / 2; i++) v[i] = (2 * i + 1 < n) ? fma(v[2 * i + 1], pw, v[2 * i]) : v[2 * i];
        pw = pw * pw;
    }
    return v[0];
}
// Coefficient tables of the plant's kernels.  plant_coeffs::load() puts all 47 of them into VECTOR registers for the duration of the 100 sub-steps (the empty asm makes
// each value opaque, so that the compiler cannot fold it back into a literal): as literals they did not fit the scalar register file and were re-materialised inside
// the loop -- 39 s_mov_b32 and 20 v_mov_b64 of the 201 instructions of a sub-step's tyre and heading part, on a wave that issues one instruction every ~6 cycles.
struct plant_coeffs {
    double at[23], s1[10], sk[7], ck[7];
    __device__ __forceinline__ void load() {
        constexpr double AT[23] = {1.00000000000000000e+00, -3.33333333333333148e-01, 1.99999999999972672e-01, -1.42857142855245423e-01, 1.11111111040167687e-01,
            -9.09090892666328670e-02, 7.69230512832331237e-02, -6.66663809625125253e-02, 5.88211633270929110e-02, -5.26165758832842292e-02, 4.75445443774553250e-02,
            -4.31833735800102661e-02, 3.90564747161603193e-02, -3.45674894142850089e-02, 2.91380929673892217e-02, -2.25878550465188170e-02, 1.54855204521000701e-02,
            -9.01332229968379930e-03, 4.26429640711083104e-03, -1.55763149860080764e-03, 4.09014166942688539e-04, -6.83513646937304115e-05, 5.44016462407589206e-06};
        constexpr double S1[10] = {1.00000000000000000e+00, -1.66666666666666657e-01, 8.33333333333335924e-03, -1.98412698412862263e-04, 2.75573192281394626e-06,
            -2.50521085826905639e-08, 1.60589365566945973e-10, -7.62490938337113395e-13, 1.09951716654846206e-15, 4.72028309461793050e-16};
        constexpr double SK[7] = {-1.66666666666666657e-01, 8.33333333333338699e-03, -1.98412698413160743e-04, 2.75573192401844066e-06, -2.50521105474221276e-08,
                                  1.60589767854145033e-10, -7.60496180966857912e-13};
        constexpr double CK[7] = {4.16666666666671293e-02, -1.38888888890215394e-03, 2.48015874329884951e-05, -2.75573799134182448e-07, 2.08910323522207627e-09,
                                  -1.31296120098958296e-11, 8.03487213580045595e-13};
#pragma unroll
        for (int i = 0; i < 23; i++) { at[i] = AT[i]; asm volatile("" : "+v"(at[i])); }
#pragma unroll
        for (int i = 0; i < 10; i++) { s1[i] = S1[i]; asm volatile("" : "+v"(s1[i])); }
#pragma unroll
        for (int i = 0; i < 7; i++) { sk[i] = SK[i]; asm volatile("" : "+v"(sk[i])); ck[i] = CK[i]; asm volatile("" : "+v"(ck[i])); }
    }
};
__device__ __forceinline__ double plant_atan_poly(const plant_coeffs &k, double z) { return z * estrin(k.at, z * z); }          // atan(z), |z| <= 1
__device__ __forceinline__ double plant_sin1_poly(const plant_coeffs &k, double x) { return x * estrin(k.s1, x * x); }          // sin(x), |x| <= 1
__device__ __forceinline__ void plant_sincos_fast(const plant_coeffs &k_, double x, double &sn, double &cs) {                    // |x| < 1e5
    const double k = rint(x * 0.63661977236758134);               // x = k pi / 2 + r, |r| <= pi / 4 (+ rounding)
    double r = fma(-k, 1.5707963267948966, x); r = fma(-k, 6.123233995736766e-17, r);
    const double w = r * r;
    const double sr = fma(r * w, estrin(k_.sk, w), r), cr = fma(w * w, estrin(k_.ck, w), fma(-0.5, w, 1.0));
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    sn = (q & 2) ? -s0 : s0; cs = ((q + 1) & 2) ? -c0 : c0;
}
// Map.curvature with the segment of the previous call tried first: s moves ~1e-3 of a segment per sub-step, so the table walk (track_rows comparisons,
// 7 here) runs a few times per simulated step instead of 100 times.  Same arithmetic and the same comparisons as track_curvature on the segment found -- the
// wrap `while s > TrackLength` is repeated every call, the segments are disjoint, the reference takes the first that matches -- hence the same value.
struct plant_seg { double c0, c1, cur; int nw; };
__device__ __forceinline__ double plant_curvature(const lmpc_dev_params &p, double s, plant_seg &g, int *bad) {
    const double TL = p.TL;
    // the cached number of wraps first, branch-free: the same sequential subtractions the reference's loop makes (bit-identical s), up to three of them.
    // It is the loop's result iff the value before the last subtraction was > TL (s_w > 0 here, or no subtraction) and the result is not (> TL) -- implied by
    // lying inside a segment of the table.
    double sw = s;
    sw = g.nw >= 1 ? sw - TL : sw; sw = g.nw >= 2 ? sw - TL : sw; sw = g.nw >= 3 ? sw - TL : sw;
    if ((g.nw == 0 || sw > 0.0) && sw >= g.c0 && sw < g.c1) return g.cur;
    int nw = 0;
    for (; nw < 64 && s > TL; nw++) s = s - TL;
    g.c0 = 1.0; g.c1 = 0.0; g.nw = nw <= 3 ? nw : 0;                 // (more than three laps: no cache, every call walks)
    if (!(s <= TL)) { *bad = 1; return 0.0; }
    for (int i = 0; i < p.track_rows; i++) {
        const double c0 = p.track[i * 6 + 3], len = p.track[i * 6 + 4];
        if (s >= c0 && s < c0 + len) { if (nw <= 3) { g.c0 = c0; g.c1 = c0 + len; } g.cur = p.track[i * 6 + 5]; return g.cur; }
    }
    *bad = 1;
    return 0.0;
}
// Round 5, second step: TWO WAVES per group of 32 cars.  The velocity dynamics (vx, vy, wz: slip angles, tyre forces) do not depend on the kinematic states, and the
// kinematic states (psi, X, Y, epsi, s, ey: headings' sines and cosines, curvature, the curvilinear quotient) only integrate the velocities: two instruction
// streams of about equal length (~150 each) that one wave used to issue one behind the other, at one instruction per ~6 cycles.  Wave 0 of a work-group now runs the
// first, wave 1 the second, one sub-step behind through a double-buffered LDS slot per car (v_i: three doubles) and one s_barrier per sub-step.  Within a wave the
// lane pairs are as before (front / rear tyre in wave 0, psi / epsi in wave 1), every transcendental is evaluated once per sub-step with the same argument and
// the same routine, so the results are those of the one-wave form bit for bit.  77 -> ~45 us per simulated step at 1024 rollouts: off the critical path of the
// closed loop (it runs beside the next step's regression kernel, ~58 us).
#define PLANT_CARS 32                                   // cars per work-group (2 lanes each in both waves)
#define PLANT_NT (2 * WAVE)
// (the barrier builtin is IntrNoMem: compiler fences on both sides keep the LDS hand-over on its side of it; the producer drains its LDS writes first)
#define PLANT_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
struct plant_lds { double v[2][PLANT_CARS][3]; };
// every thread of the work-group must call this (barriers inside); `on` = this lane's car exists.  Wave 1's lanes return the new state in xn / xgn
// (both lanes of a pair hold the same values); wave 0's return nothing.
__device__ __forceinline__ void plant_step_duo(const lmpc_dev_params &p, plant_lds &L, const double *x, const double *xg, const double *u, const double *nz,
                                               double *xn, double *xgn, int *bad, const int wave, const int role, const int cl) {
    const double m = 1.98, lf = 0.125, lr = 0.125, Iz = 0.024;
    const double Df = 0.8 * m * 9.81 / 2.0, Cf = 1.25, Bf = 1.0;                     // rear tyre: same D, C, B (SysModel.py:68-76)
    const double deltaT = 0.001;
    const double delta = u[0], a = u[1];
    plant_coeffs kc; kc.load();
    if (wave == 0) {
        // ---- velocity dynamics: lane `role` 0 the front tyre, 1 the rear tyre ----
        double vx = x[0], vy = x[1], wz = x[2];
        const double sd = sin(delta), cd = cos(delta);
        if (role == 0) { L.v[0][cl][0] = vx; L.v[0][cl][1] = vy; L.v[0][cl][2] = wz; }
        PLANT_BARRIER();
#pragma unroll 1
        for (int i = 0; i < 100; i++) {                              // while (i+1)*deltaT <= dt, SysModel.py:93
            const double yq = role ? vy - lf * wz : vy + lf * wz;
            const double rvx = frcp(vx);
            double zq = yq * rvx; zq = fma(fma(-zq, vx, yq), rvx, zq);   // yq / vx to ~1 ulp
            double at = plant_atan_poly(kc, zq);
            double alpha = role ? -at : delta - at;                  // alpha_r = -atan2(vy - lf wz, vx), alpha_f = delta - atan2(vy + lf wz, vx)
            const double ba = Bf * alpha;
            const double xs = Cf * plant_atan_poly(kc, ba);
            double F = Df * plant_sin1_poly(kc, xs);
            if (!(vx > 0.0 && fabs(yq) <= vx && fabs(ba) <= 1.0 && fabs(xs) <= 1.0)) {      // a sliding or diverged car: the general routines
                at = atan2(yq, vx); alpha = role ? -at : delta - at;
                F = Df * sin(Cf * atan(Bf * alpha));
            }
            const double Fo = dpp_mov<DPP_QP_X1>(F);
            const double Fyf = role ? Fo : F, Fyr = role ? F : Fo;
            const double nvx = vx + deltaT * (a - 1 / m * Fyf * sd + wz * vy);
            const double nvy = vy + deltaT * (1 / m * (Fyf * cd + Fyr) - wz * vx);
            const double nwz = wz + deltaT * (1 / Iz * (lf * Fyf * cd - lr * Fyr));
            vx = nvx; vy = nvy; wz = nwz;
            if (role == 0) { double *d_ = L.v[(i + 1) & 1][cl]; d_[0] = vx; d_[1] = vy; d_[2] = wz; }
            PLANT_BARRIER();
        }
    } else {
        // ---- kinematic states: lane `role` 0 the heading psi, 1 the heading error epsi ----
        double psi = xg[3], X = xg[4], Y = xg[5];
        double epsi = x[3], s = x[4], ey = x[5];
        plant_seg seg; seg.c0 = 1.0; seg.c1 = 0.0; seg.cur = 0.0; seg.nw = 0;       // (empty interval: the first call walks the table)
        PLANT_BARRIER();
#pragma unroll 1
        for (int i = 0; i < 100; i++) {
            const double *v_ = L.v[i & 1][cl];
            const double vx = v_[0], vy = v_[1], wz = v_[2];
            const double ang = role ? epsi : psi;
            double sn, cs; plant_sincos_fast(kc, ang, sn, cs);
            if (!(fabs(ang) < 1.0e5)) { sn = sin(ang); cs = cos(ang); }
            const double sno = dpp_mov<DPP_QP_X1>(sn), cso = dpp_mov<DPP_QP_X1>(cs);
            const double sp = role ? sno : sn, cp = role ? cso : cs, se = role ? sn : sno, ce = role ? cs : cso;
            const double npsi = psi + deltaT * (wz);
            const double nX = X + deltaT * ((vx * cp - vy * sp));
            const double nY = Y + deltaT * (vx * sp + vy * cp);
            const double cur = plant_curvature(p, s, seg, bad);
            const double den = 1 - cur * ey, num = vx * ce - vy * se, rden = frcp(den);
            double qd = num * rden; qd = fma(fma(-qd, den, num), rden, qd);     // (vx ce - vy se) / (1 - cur ey), ~1 ulp (the reference evaluates it twice)
            const double nepsi = epsi + deltaT * (wz - qd * cur);
            const double ns = s + deltaT * (qd);
            const double ney = ey + deltaT * (vx * se + vy * ce);
            epsi = nepsi; s = ns; ey = ney; psi = npsi; X = nX; Y = nY;
            PLANT_BARRIER();
        }
        const double *v_ = L.v[0][cl];                               // v_100 (slot 100 & 1)
        const double vx = v_[0], vy = v_[1], wz = v_[2];
        const double n0 = fmax(-0.05, fmin(nz[0] * 0.01, 0.05)), n1 = fmax(-0.05, fmin(nz[1] * 0.01, 0.05)), n2 = fmax(-0.05, fmin(nz[2] * 0.005, 0.05));
        xn[0] = vx + 0.01 * n0; xn[1] = vy + 0.01 * n1; xn[2] = wz + 0.01 * n2; xn[3] = epsi; xn[4] = s; xn[5] = ey;   // :139-145
        xgn[0] = vx; xgn[1] = vy; xgn[2] = wz; xgn[3] = psi; xgn[4] = X; xgn[5] = Y;
    }
}

__global__ __launch_bounds__(PLANT_NT) void lmpc_plant_kernel(lmpc_dev_params p, int B, const double *__restrict__ x, const double *__restrict__ xg, const double *__restrict__ u,
                                  const double *__restrict__ nz, double *__restrict__ xn, double *__restrict__ xgn, int *__restrict__ status) {
    __shared__ plant_lds L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, cl = lane >> 1, role = lane & 1;
    const int b0 = blockIdx.x * PLANT_CARS + cl; const bool on = b0 < B; const int b = on ? b0 : B - 1;      // (a lane without a car repeats the last one: barriers inside)
    int bad = 0;
    double xo[6], go[6];
    plant_step_duo(p, L, x + (size_t)b * 6, xg + (size_t)b * 6, u + (size_t)b * 2, nz + (size_t)b * 3, xo, go, &bad, wave, role, cl);
    if (wave == 1 && role == 0 && on) {
        for (int j = 0; j < 6; j++) { xn[(size_t)b * 6 + j] = xo[j]; xgn[(size_t)b * 6 + j] = go[j]; }
        if (status) status[b] = bad ? LMPC_ST_NO_SEGMENT : 0;
    }
}

// One closed-loop step of every rollout after lmpc_step_batch_dev: log (x_t, u_t, xglob_t), integrate the plant with
// u_t = uPred[0] (SysModel.py:34-40), then the tail of MPC.solve (:131-137): xLin/uLin shift, OldInput, zt, xPred -> prev.
struct lmpc_rollout_state {
    double *x, *xg, *xLin, *uLin, *uOld, *zt, *xPP; int *hasPred, *timeStep, *doneAt;        // controller + plant state (B, ...)
    const double *xPred, *uPred, *ztNext, *ztuNext; const int *status;                        // outputs of the step just taken
    double *logX, *logU, *logG; const double *noise; int *nDone, *statusAcc;                   // logs [T][B][..], noise [T][B][3]
    double *finX, *finG;                                                                       // state right after the crossing step
};
// The step's bookkeeping and its plant integration are two kernels on two streams: the shift of the linearisation trajectory
// feeds the NEXT step's regression kernel, which does not need the plant's result and runs concurrently with it; only the next
// solve waits for the new state.
// (round 5: one thread per copied ELEMENT -- 8 (N + 1) threads per rollout, all loads independent.  The first form gave two lanes per rollout a strided loop of
//  3 N dependent load / store pairs: 21 us on the critical path of every simulated step between the solve and the next regression; now ~3 us.)
#define LMPC_SHIFT_TPR(N) (8 * ((N) + 1))                           // threads per rollout: (N + 1) rows x (6 state + 2 input) columns
__global__ __launch_bounds__(256) void lmpc_rollout_shift_kernel(lmpc_dev_params p, int B, int t, lmpc_rollout_state r) {
    const int N = p.N, tpr = LMPC_SHIFT_TPR(N);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, b = tid / tpr, e = tid - b * tpr;
    if (b >= B) return;
    const int row = e >> 3, col = e & 7;                              // row 0 .. N of the prediction, column 0 .. 5: state, 6, 7: input
    const double *uP = r.uPred + (size_t)b * N * 2, *xP = r.xPred + (size_t)b * (N + 1) * 6;
    double *xl = r.xLin + (size_t)b * (N + 1) * 6, *ul = r.uLin + (size_t)b * N * 2, *xpp = r.xPP + (size_t)b * (N + 1) * 6;
    if (col < 6) {
        const double v = xP[row * 6 + col];
        xpp[row * 6 + col] = v;                                       // xPred -> prev (Q-function shift bookkeeping of the next selection)
        if (row >= 1) xl[(row - 1) * 6 + col] = v;                    // xLin <- xPred[1:], :131-133
        if (row == N) { const double z = r.ztNext[(size_t)b * 6 + col]; xl[N * 6 + col] = z; r.zt[(size_t)b * 6 + col] = z; }
    } else {
        const int c = col - 6;
        if (row >= 1 && row < N) ul[(row - 1) * 2 + c] = uP[row * 2 + c];      // uLin <- uPred[1:]
        if (row == N) ul[(N - 1) * 2 + c] = r.ztuNext[(size_t)b * 2 + c];
        if (row == 0) r.uOld[(size_t)b * 2 + c] = uP[c];                       // OldInput = uPred[0], :134
    }
    if (e == 0) { r.hasPred[b] = 1; r.timeStep[b] = t + 1; }
}
__global__ __launch_bounds__(PLANT_NT) void lmpc_rollout_plant_kernel(lmpc_dev_params p, int B, int t, lmpc_rollout_state r) {
    __shared__ plant_lds L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, cl = lane >> 1, role = lane & 1;     // two waves per 32 cars, see plant_step_duo
    const int b0 = blockIdx.x * PLANT_CARS + cl; const bool on = b0 < B; const int b = on ? b0 : B - 1;
    const int N = p.N;
    double *x = r.x + (size_t)b * 6, *xg = r.xg + (size_t)b * 6;
    const double *uP = r.uPred + (size_t)b * N * 2;
    const double u0[2] = {uP[0], uP[1]};
    const bool writer = wave == 1 && role == 0 && on;
    if (writer) {
        for (int j = 0; j < 6; j++) { r.logX[((size_t)t * B + b) * 6 + j] = x[j]; r.logG[((size_t)t * B + b) * 6 + j] = xg[j]; }
        r.logU[((size_t)t * B + b) * 2] = u0[0]; r.logU[((size_t)t * B + b) * 2 + 1] = u0[1];
    }
    int bad = 0; double xo[6], go[6];
    plant_step_duo(p, L, x, xg, u0, r.noise + ((size_t)t * B + b) * 3, xo, go, &bad, wave, role, cl);
    if (writer) {
        for (int j = 0; j < 6; j++) { x[j] = xo[j]; xg[j] = go[j]; }
        // status bits count only up to and including the step that crosses the line: a finished car keeps being simulated until
        // the slowest rollout ends, and whatever happens to it there (window past the lap end, ...) does not belong to its lap
        if (r.doneAt[b] < 0) r.statusAcc[b] |= r.status[b] | (bad ? LMPC_ST_NO_SEGMENT : 0);
        if (r.doneAt[b] < 0 && xo[4] > p.TL) {                                                 // lap completed, SysModel.py:45
            r.doneAt[b] = t + 1; atomicAdd(r.nDone, 1);
            for (int j = 0; j < 6; j++) { r.finX[(size_t)b * 6 + j] = xo[j]; r.finG[(size_t)b * 6 + j] = go[j]; }
        }
    }
}

// =====================================================================================================
// Utilities.Regression (fnc/Utilities.py:5-28): the LTI model of main.py's second stage (main.py:74-77), one ridge least-squares fit
// over a whole lap: rows r = 0..T-3, z_r = [x_{r+1}, u_{r+1}] (8), y_r = x_{r+2} (6); W = (Z'Z + lamb I)^-1 Z'Y; A = W'[:, 0:6],
// B = W'[:, 6:8]; Error = [max_r; min_r] of (Z W - Y) per column.  One work-group: 84 accumulated sums (36 Gram + 48 right-hand side
// entries) x 12 row slices, an 8 x 8 Cholesky per output column, then the residual extrema by a block reduction.
// out: A (36, row-major 6 x 6) | B (12, 6 x 2) | Error (12, 2 x 6).  status: LMPC_ST_REG_SINGULAR if Z'Z + lamb I is not positive definite.
// =====================================================================================================
#define LTI_NT 1024
#define LTI_PARTS 12
__global__ __launch_bounds__(LTI_NT) void lmpc_lti_regress_kernel(const double *__restrict__ x, const double *__restrict__ u, int T, double lamb,
                                                                  double *__restrict__ out, int *__restrict__ status) {
    __shared__ double part[LTI_PARTS][84];
    __shared__ double G[8][8], bv[8][6], W[8][6];
    __shared__ double emax[LTI_NT / WAVE][6], emin[LTI_NT / WAVE][6];
    __shared__ int bad_s;
    const int tid = threadIdx.x, R = T - 2;
    auto zval = [&](int r, int i) { return i < 6 ? x[(size_t)(r + 1) * 6 + i] : u[(size_t)(r + 1) * 2 + (i - 6)]; };
    if (tid == 0) bad_s = 0;
    if (tid < 84 * LTI_PARTS) {
        const int e = tid % 84, p = tid / 84;
        int i, j; bool rhs = e >= 36;
        if (!rhs) { i = 0; j = e; while (j >= 8 - i) { j -= 8 - i; i++; } j += i; }          // upper-triangular (i, j)
        else { i = (e - 36) / 6; j = (e - 36) % 6; }
        double acc = 0.0;
        for (int r = p; r < R; r += LTI_PARTS) acc = fma(zval(r, i), rhs ? x[(size_t)(r + 2) * 6 + j] : zval(r, j), acc);
        part[p][e] = acc;
    }
    __syncthreads();
    if (tid < 84) {
        double acc = 0.0;
        for (int p = 0; p < LTI_PARTS; p++) acc += part[p][tid];
        if (tid < 36) { int i = 0, j = tid; while (j >= 8 - i) { j -= 8 - i; i++; } j += i; if (i == j) acc += lamb; G[i][j] = acc; G[j][i] = acc; }
        else bv[(tid - 36) / 6][(tid - 36) % 6] = acc;
    }
    __syncthreads();
    if (tid < 6) {                                            // column tid of W: Cholesky G = L L', two triangular solves
        double L[8][8]; int bad = 0;
        for (int j = 0; j < 8; j++) {
            double d = G[j][j];
            for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
            if (!(d > 0.0)) { bad = 1; d = 1.0; }
            d = sqrt(d); L[j][j] = d;
            for (int r = j + 1; r < 8; r++) { double v = G[r][j]; for (int k = 0; k < j; k++) v -= L[r][k] * L[j][k]; L[r][j] = v / d; }
        }
        double y[8];
        for (int r = 0; r < 8; r++) { double v = bv[r][tid]; for (int k = 0; k < r; k++) v -= L[r][k] * y[k]; y[r] = v / L[r][r]; }
        for (int r = 7; r >= 0; r--) { double v = y[r]; for (int k = r + 1; k < 8; k++) v -= L[k][r] * y[k]; y[r] = v / L[r][r]; }
        for (int r = 0; r < 8; r++) W[r][tid] = y[r];
        if (bad) atomicOr(&bad_s, 1);
    }
    __syncthreads();
    double mx[6], mn[6];
#pragma unroll
    for (int c = 0; c < 6; c++) { mx[c] = -INFINITY; mn[c] = INFINITY; }
    for (int r = tid; r < R; r += LTI_NT) {
        double z[8];
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = zval(r, i);
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double e = 0.0;
#pragma unroll
            for (int i = 0; i < 8; i++) e = fma(z[i], W[i][c], e);
            e -= x[(size_t)(r + 2) * 6 + c];
            mx[c] = fmax(mx[c], e); mn[c] = fmin(mn[c], e);
        }
    }
#pragma unroll
    for (int c = 0; c < 6; c++) { mx[c] = wave_max(mx[c]); mn[c] = wave_min(mn[c]); }
    if ((tid & (WAVE - 1)) == 0) { for (int c = 0; c < 6; c++) { emax[tid >> 6][c] = mx[c]; emin[tid >> 6][c] = mn[c]; } }
    __syncthreads();
    if (tid < 36) out[tid] = W[tid % 6][tid / 6];                                  // A[c][i] = W[i][c]
    else if (tid < 48) out[tid] = W[6 + (tid - 36) % 2][(tid - 36) / 2];           // B[c][j] = W[6 + j][c]
    else if (tid < 60) {
        const int c = (tid - 48) % 6; const bool is_min = tid >= 54;
        double v = is_min ? INFINITY : -INFINITY;
        for (int w = 0; w < LTI_NT / WAVE; w++) v = is_min ? fmin(v, emin[w][c]) : fmax(v, emax[w][c]);
        out[tid] = v;
    }
    if (tid == 0) *status = bad_s ? LMPC_ST_REG_SINGULAR : 0;
}

#endif  // LMPC_VARIANT_TU
#ifndef LMPC_VARIANT_TU
// wave-reduction self test (exercised by lmpc_selftest): out[0..2] = sum, max, min of lane-dependent values
__global__ void lmpc_selftest_kernel(double *out) {
    const int lane = threadIdx.x;
    const double v = 1.0 + 0.25 * lane + ((lane * 37) % 11) * 1e-3;
    const double s = wsum(v), mx = wmax(v), mn = wmin(v);
    out[lane] = s; out[64 + lane] = mx; out[128 + lane] = mn;
}

// =====================================================================================================
// Explicit reference-form QP (parity checks only): P (nz x nz), q, A_osqp = [F; G] (m x nz), l, u, dense
// row-major.  Follows buildIneqConstr :166-198, buildCost :228-257, buildEqConstr :200-226,
// addSafeSetIneqConstr :340-343, addSafeSetEqConstr :345-357, addSafeSetCost :359-362, osqp_solve_qp :269-273.
// One work-group per problem; every thread fills a strided share of the entries.
// =====================================================================================================
__global__ __launch_bounds__(256) void lmpc_assemble_kernel(lmpc_dev_params p, int B, const double *__restrict__ A, const double *__restrict__ Bm,
                                                            const double *__restrict__ C, const double *__restrict__ x0, const double *__restrict__ uOld,
                                                            const double *__restrict__ ssSel, const double *__restrict__ qSel,
                                                            double *__restrict__ Pd, double *__restrict__ q, double *__restrict__ Ad,
                                                            double *__restrict__ l, double *__restrict__ u) {
    const int b = blockIdx.x; if (b >= B) return;
    const int N = p.N, S = p.S, n = 6, d = 2;
    const int ns = p.slacks ? 2 * N : 0;                 // slack variables and their positivity rows (MPCParams.slacks, :184-198)
    const int ox = 0, ou = n * (N + 1), os = ou + d * N, ol = os + ns, ot = ol + S, nz = S > 0 ? ot + n : ol;
    const int mi = 6 * N + ns + S, me = n * (N + 1) + (S > 0 ? n + 1 : 0), mm = mi + me;
    double *Pb = Pd + (size_t)b * nz * nz, *Ab = Ad + (size_t)b * mm * nz, *qb = q + (size_t)b * nz, *lb = l + (size_t)b * mm, *ub = u + (size_t)b * mm;
    const double *Ab_ = A + (size_t)b * 36 * N, *Bb_ = Bm + (size_t)b * 12 * N, *Cb_ = C + (size_t)b * 6 * N;
    for (size_t i = threadIdx.x; i < (size_t)nz * nz; i += blockDim.x) Pb[i] = 0.0;
    for (size_t i = threadIdx.x; i < (size_t)mm * nz; i += blockDim.x) Ab[i] = 0.0;
    __syncthreads();
    for (int i = threadIdx.x; i < nz; i += blockDim.x) {
        double v = 0.0;
        if (i < ou) { const int k = i / n, c = i % n; const double *Qk = k < N ? p.Q2 : p.Qf2; for (int j = 0; j < n; j++) v -= p.xRef[j] * Qk[j * 6 + c]; }
        else if (i < os) { const int k = (i - ou) / d, c = (i - ou) % d; v = k == 0 ? -p.dR2[c] * uOld[(size_t)b * 2 + c] : 0.0; }
        else if (i < ol) v = p.c_s;
        else if (i < ot) v = qSel[(size_t)b * S + (i - ol)];
        qb[i] = v;
    }
    // P
    for (int i = threadIdx.x; i < (N + 1) * 36; i += blockDim.x) { const int k = i / 36, r = (i % 36) / 6, c = i % 6; Pb[(size_t)(k * 6 + r) * nz + k * 6 + c] = k < N ? p.Q2[r * 6 + c] : p.Qf2[r * 6 + c]; }
    for (int i = threadIdx.x; i < N * 4; i += blockDim.x) {
        const int k = i / 4, r = (i % 4) / 2, c = i % 2;
        double v = p.R2[r * 2 + c]; if (r == c) v += (k < N - 1 ? 2.0 : 1.0) * p.dR2[r];
        Pb[(size_t)(ou + k * 2 + r) * nz + ou + k * 2 + c] = v;
        if (r == c && k < N - 1) { Pb[(size_t)(ou + k * 2 + r) * nz + ou + (k + 1) * 2 + r] = -p.dR2[r]; Pb[(size_t)(ou + (k + 1) * 2 + r) * nz + ou + k * 2 + r] = -p.dR2[r]; }
    }
    for (int i = threadIdx.x; i < ns; i += blockDim.x) Pb[(size_t)(os + i) * nz + os + i] = p.a_s;
    if (S > 0) for (int i = threadIdx.x; i < 6; i += blockDim.x) Pb[(size_t)(ot + i) * nz + ot + i] = p.T2[i];
    // inequality rows
    for (int r = threadIdx.x; r < mi; r += blockDim.x) {
        double *row = Ab + (size_t)r * nz; double bb = 0.0;
        if (r < 2 * N) { const int k = r >> 1, j = r & 1; for (int c = 0; c < 6; c++) row[ox + k * 6 + c] = p.Fx[j * 6 + c]; if (ns) row[os + r] = -1.0; bb = p.bx[j]; }
        else if (r < 6 * N) { const int qd = r - 2 * N, k = qd >> 2, j = qd & 3; row[ou + k * 2] = p.Fu[j * 2]; row[ou + k * 2 + 1] = p.Fu[j * 2 + 1]; bb = p.bu[j]; }
        else if (r < 6 * N + ns) row[os + (r - 6 * N)] = -1.0;
        else row[ol + (r - 6 * N - ns)] = -1.0;
        lb[r] = -INFINITY; ub[r] = bb;
    }
    // equality rows
    for (int r = threadIdx.x; r < me; r += blockDim.x) {
        double *row = Ab + (size_t)(mi + r) * nz; double be = 0.0;
        if (r < 6) { row[r] = 1.0; be = x0[(size_t)b * 6 + r]; }
        else if (r < 6 * (N + 1)) {
            const int k = r / 6 - 1, c = r % 6;
            row[(k + 1) * 6 + c] = 1.0;
            for (int j = 0; j < 6; j++) row[k * 6 + j] += -Ab_[k * 36 + c * 6 + j];
            row[ou + k * 2] = -Bb_[k * 12 + c * 2]; row[ou + k * 2 + 1] = -Bb_[k * 12 + c * 2 + 1];
            be = Cb_[k * 6 + c];
        } else if (r < 6 * (N + 1) + 6) {
            const int c = r - 6 * (N + 1);
            row[N * 6 + c] = 1.0;
            for (int j = 0; j < S; j++) row[ol + j] = -ssSel[((size_t)b * S + j) * 6 + c];
            row[ot + c] = 1.0;
        } else { for (int j = 0; j < S; j++) row[ol + j] = 1.0; be = 1.0; }
        lb[mi + r] = be; ub[mi + r] = be;
    }
}
#endif  // LMPC_VARIANT_TU
