// racinglmpc_amd/csrc/lmpc_solve_cd.hip.h -- K2 + K3 in CONDENSED form (round 3): one wave per QP, short horizons (N <= 16).
//
// Same QP, same interior-point rules as lmpc_solve_kernel (Mehrotra predictor-corrector, capped barrier weights, analytic elimination of the
// lane slacks, 7 x 7 terminal factor), but the states are eliminated: x = X(u) is the roll-out of the inputs through x_{k+1} = A_k x_k + B_k u_k
// + C_k, so the dynamics rows disappear, the only equality row left is sum(lambda) = 1, and the Newton matrix is DENSE in du (2N x 2N: 24 x 24
// at N = 12):
//     Hc = H0 (input cost, input-rate coupling, state cost) + blockdiag(Fu' Theta_u Fu) + V' K V + Y' Y
//        V_r = d(Fx_j x_k)/du  (row r = 2k + j of the lane constraints, constant per QP),   K = diag(kappa_r) (lane rows after slack elimination),
//        Y   = P' G,  G = d x_N / du,  P P' = Schur complement of the terminal block on x_N (P = first six rows of R^-1 of the terminal factor).
// What replaces what (cycles per Newton iteration on one wave, N = 12, round-2 timing build):
//   * Riccati recursion over 12 stages, six dependent 4x4x4 MFMAs per stage (13.9 k)  ->  Hc by 24 v_mfma_f64_16x16x4 (three 16 x 16 tiles of W' D W,
//     K = 2N + 7 rows of W = [V; Y]), then a 24 x 24 Cholesky with one matrix row per lane and the pivot column broadcast by v_readlane
//   * backward / feed-forward / forward sweeps of every Newton solve (5.9 k per right-hand side) -> two triangular substitutions in registers and
//     dense products with V and G
//   * costate recursion for the multipliers of the dynamics rows (6.1 k)             ->  gone (no such rows); the adjoint recursion
//     p_k = A_k' p_{k+1} + w_k that gives the state part of the input rows' residual and the roll-out of the new inputs remain, one each per iteration
// tests/ipm_model.py::ipm_solve_cd is the NumPy model this kernel follows statement by statement.
//
// Limits: 2N <= 32 (two 16-wide tile columns; longer horizons stay with the Riccati kernels), S + 6 <= 64, diagonal Q and Qf (the reference's tunings).
#pragma once
#include "lmpc_kernels.hip.h"

template <int N, int S> struct solve_ldsc {
    static constexpr int M = 8 * N + S, NV = 2 * N, NVP = NV + 1;
    static constexpr int KW = 2 * N + 7, KP = (KW + 3) & ~3;                   // rows of W = [V (2N) ; Y (7)], padded to whole K-chunks of the MFMA
    static constexpr int CW = WAVE;                                             // terminal-block columns (one per lane)
    static constexpr int ox = 0, ou = ox + 6 * (N + 1), os = ou + NV, olam = os + NV;
    static constexpr int om = olam + S, oth = om + M;
    static constexpr int oW = oth + M, odW = oW + KP * NVP, oG = odW + KP;
    static constexpr int oeta = oG + 6 * NVP, oee = oeta + NV;
    static constexpr int oh = oee + NV, odm = oh;
    static constexpr int odu = oh + M, ods = odu + NV, odl = ods + NV, ofl = odl + S;
    static constexpr int ogam = ofl + NV, ofadd = ogam + 8 * (N + 1), obtp = ofadd + 8 * N;
    static constexpr int oSS = obtp + NV, oRi = oSS + 6 * S, oMc = oRi + 56, opT = oMc + 8, osT = opT + 8, ow7 = osT + 8, oz7 = ow7 + 8;
    static constexpr int opar = oz7 + 8, oscr = opar + PAR_TOT;
    // scratch, by phase: start-up [A_k | B_k] (48 N), C_k (6 N), Qfun_sel (S)  |  terminal factor: Mt (8 x 64), Gram matrix (64)  |  Hc / L (NV x NVP)
    static constexpr int oAB = oscr, oCs = oAB + 48 * N, oQs = oCs + 6 * N;
    static constexpr int oMt = oscr, oWl = oMt + 8 * CW;
    static constexpr int oHs = oscr;
    static constexpr int scr0 = 54 * N + S, scr1 = S > 0 ? 8 * CW + 64 : 0, scr2 = NV * NVP;
    static constexpr int scr = scr0 > scr1 ? (scr0 > scr2 ? scr0 : scr2) : (scr1 > scr2 ? scr1 : scr2);
    static constexpr int oHq = oscr + scr;                                      // constant state-cost part of Hc (only when Q or Qf is non-zero)
    static constexpr int tot = oHq, tot_q = oHq + NV * NVP;
};

template <int N, int S, bool EQ = false>
__global__ __launch_bounds__(WAVE, 2) void lmpc_solve_kernel_cd(lmpc_dev_params p, int B, lmpc_solve_io io, int hasQ) {
    extern __shared__ double sm[];
    using LL = solve_ldsc<N, S>;
    static_assert(2 * N <= 32 && S + 6 <= WAVE, "condensed kernel: short horizons, one terminal-block column per lane");
    constexpr int M = LL::M, NV = LL::NV, NVP = LL::NVP, KP = LL::KP;
    constexpr bool term = S > 0;
    constexpr int RPL = (M + WAVE - 1) / WAVE;
    const int b = blockIdx.x;
    if (b >= B) return;
    if constexpr (EQ) { if (!(io.status[b] & (LMPC_ST_MAXITER | LMPC_ST_NUMERIC))) return; }
    const int lane = threadIdx.x;
    const int lg = lane >> 3, lc = lane & 7;
    double *x = sm + LL::ox, *u = sm + LL::ou, *s = sm + LL::os, *lam = sm + LL::olam, *m = sm + LL::om, *th = sm + LL::oth;
    double *W = sm + LL::oW, *dW = sm + LL::odW, *G = sm + LL::oG, *eta = sm + LL::oeta, *ee = sm + LL::oee;
    double *h = sm + LL::oh, *dm = sm + LL::odm, *du = sm + LL::odu, *ds = sm + LL::ods, *dl = sm + LL::odl, *fl = sm + LL::ofl;
    double *gam = sm + LL::ogam, *fadd = sm + LL::ofadd, *btp = sm + LL::obtp;
    double *SS = sm + LL::oSS, *Ri = sm + LL::oRi, *pT = sm + LL::opT, *sT = sm + LL::osT, *w7 = sm + LL::ow7, *z7 = sm + LL::oz7;
    double *par = sm + LL::opar, *AB = sm + LL::oAB, *Cs = sm + LL::oCs, *Qsel = sm + LL::oQs, *Mt = sm + LL::oMt, *Wl = sm + LL::oWl, *Hs = sm + LL::oHs;
    double *Hq = sm + LL::oHq;
    const double *Fx = par + PAR_FX, *Fu = par + PAR_FU, *bx = par + PAR_BX, *bu = par + PAR_BU, *Q2 = par + PAR_Q2, *Qf2 = par + PAR_QF2,
                 *R2 = par + PAR_R2, *dR2 = par + PAR_DR2, *T2p = par + PAR_T2, *xRef = par + PAR_XREF;
    __shared__ int st_sh;
    __shared__ int sel_start[LMPC_MAX_USED_LAPS];
    int tcnt = 0; (void)tcnt;
    TSTAMP(0);
    if (lane == 0) st_sh = 0;
    if (lane < 12) par[PAR_FX + lane] = p.Fx[lane];
    if (lane < 8) par[PAR_FU + lane] = p.Fu[lane];
    if (lane < 2) { par[PAR_BX + lane] = p.bx[lane]; par[PAR_DR2 + lane] = p.dR2[lane]; }
    if (lane < 4) { par[PAR_BU + lane] = p.bu[lane]; par[PAR_R2 + lane] = p.R2[lane]; }
    if (lane < 36) { par[PAR_Q2 + lane] = p.Q2[lane]; par[PAR_QF2 + lane] = p.Qf2[lane]; }
    if (lane < 6) { par[PAR_T2 + lane] = p.T2[lane]; par[PAR_XREF + lane] = p.xRef[lane]; }
    if (lane == 0) { par[PAR_AS] = p.a_s; par[PAR_CS] = p.c_s; }
    __syncthreads();
    const double a_s = wave_uniform(par[PAR_AS]), c_s = wave_uniform(par[PAR_CS]);

    // A_k, B_k, C_k: global loads issued before the selection's lap scans (their latency runs beside them)
    constexpr int TA = (36 * N + WAVE - 1) / WAVE, TB = (12 * N + WAVE - 1) / WAVE, T6N = (6 * N + WAVE - 1) / WAVE;
    double preA[TA], preB[TB], preC[T6N];
    if (io.mode & 2) {
        FOR_LANES_T(i, t, 36 * N) preA[t] = io.A[(size_t)b * 36 * N + i];
        FOR_LANES_T(i, t, 12 * N) preB[t] = io.Bm[(size_t)b * 12 * N + i];
        FOR_LANES_T(i, t, 6 * N) preC[t] = io.C[(size_t)b * 6 * N + i];
    }
    if constexpr (term) { k2_select<N, S, 1>(p, io, b, lane, 0, SS, Qsel, sel_start, &st_sh); __syncthreads(); }
    if (io.rstatus && lane < N) { const int rs_ = io.rstatus[(size_t)b * N + lane]; if (rs_) atomicOr(&st_sh, rs_); }
    TSTAMP(1);
    if (!(io.mode & 2)) { if (lane == 0) io.status[b] = st_sh; return; }

    FOR_LANES_T(i, t, 36 * N) { const int k = i / 36, r = (i % 36) / 6, c = i % 6; AB[k * 48 + r * 8 + c] = preA[t]; }
    FOR_LANES_T(i, t, 12 * N) { const int k = i / 12, r = (i % 12) >> 1, c = i & 1; AB[k * 48 + r * 8 + 6 + c] = preB[t]; }
    FOR_LANES_T(i, t, 6 * N) { const int k = i / 6, c = i % 6; fadd[k * 8 + c] = preC[t]; }               // additive terms of the roll-out: (C_k ; u_{k+1})
    double qsel_r = 0.0;
    if constexpr (term) { if (lane < S) qsel_r = Qsel[lane]; }
    if (lane < 6) x[lane] = io.x0[(size_t)b * 6 + lane];
    if (lane < NV) u[lane] = 0.0;
    FOR_LANES(i, 8 * N) { if ((i & 7) >= 6) fadd[i] = 0.0; }
    FOR_LANES(i, 8 * (N + 1)) gam[i] = 0.0;
    FOR_LANES(i, KP * NVP) W[i] = 0.0;
    FOR_LANES(i, KP) dW[i] = (i >= 2 * N && i < 2 * N + 7) ? 1.0 : 0.0;
    const double uOld0 = wave_uniform(io.uOld[(size_t)b * 2 + 0]), uOld1 = wave_uniform(io.uOld[(size_t)b * 2 + 1]);
    __syncthreads();

    // sweep operands: entry of Ar_k = [[A_k, B_k], [0, 0]] (8 x 8) this lane multiplies with in the register sweeps (roll-out forward, adjoint backward):
    // stage k even: lane (g, c) holds Ar_k[g][c], stage k odd: Ar_k[c][g]  (same alternating layout as the sweeps of lmpc_solve_kernel)
    double ph[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int r = (k & 1) ? lc : lg, c_ = (k & 1) ? lg : lc;
        const double v = AB[k * 48 + (r < 6 ? r : 0) * 8 + c_];
        ph[k] = r < 6 ? v : 0.0;
    }
    // x = X(u): forward sweep xi_{k+1} = Ar_k xi_k + (C_k ; u_{k+1}), xi_k = (x_k ; u_k), in registers; x_1 .. x_N go to LDS
    auto rollout = [&]() {
        double fm[N];
#pragma unroll
        for (int k = 0; k < N; k++) fm[k] = (k & 1) ? fadd[k * 8 + lc] : fadd[k * 8 + lg];
        double xi = lc < 6 ? x[lc] : u[lc - 6];                                      // xi_0[lc]
#pragma unroll
        for (int k = 0; k < N; k++) {
            double pr = ph[k] * xi;
            int idx;
            if (k & 1) { pr = sum_over_g(pr); idx = lc; } else { pr = sum_over_c(pr); idx = lg; }
            xi = pr + fm[k];
            const bool wr = (k & 1) ? (lg == 0) : (lc == 0);
            if (wr && idx < 6) x[(k + 1) * 6 + idx] = xi;
        }
    };
    rollout();                                                                       // strictly interior start: u = 0
    // Per-QP constants of the condensed form.  Lane i < 2N carries column i = (stage j, input c) of the sensitivities: w = d x_k / d u_i, propagated
    // stage by stage (w = B_j[:, c] at k = j + 1, then w <- A_k w); V_r[i] = Fx_jj . w for the lane rows r = 2k + jj, G[:, i] = w at k = N.
    // With a state cost the constant part of Hc, sum_k Su_k' Q2 Su_k + G' Qf2 G (diagonal Q, Qf), is accumulated on the matrix cores along the way.
    {
        typedef double v4d __attribute__((ext_vector_type(4)));
        v4d q00 = {0.0, 0.0, 0.0, 0.0}, q10 = q00, q11 = q00;
        const int kk = lane >> 4, ii = lane & 15;
        double w[6];
#pragma unroll
        for (int r = 0; r < 6; r++) w[r] = 0.0;
        const int js = lane >> 1, cs_ = lane & 1;
#pragma unroll 1
        for (int k = 0; k < N; k++) {                                                // after this step w = d x_{k+1} / d u_i
            double nw[6];
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double v = 0.0;
#pragma unroll
                for (int c = 0; c < 6; c++) v = fma(AB[k * 48 + r * 8 + c], w[c], v);
                nw[r] = (lane < NV && k == js) ? AB[k * 48 + r * 8 + 6 + cs_] : ((lane < NV && k > js) ? v : 0.0);
            }
#pragma unroll
            for (int r = 0; r < 6; r++) w[r] = nw[r];
            if (lane < NV) {
                if (k + 1 < N) {
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) {
                        double v = 0.0;
#pragma unroll
                        for (int c = 0; c < 6; c++) v = fma(Fx[jj * 6 + c], w[c], v);
                        W[(2 * (k + 1) + jj) * NVP + lane] = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 6; r++) G[r * NVP + lane] = w[r];
                }
            }
            if (hasQ) {                                                              // Z = sqrt(q) . Su_{k+1} (6 x 2N) through the scratch behind C_k, then Z' Z on the matrix cores
                double *Z = sm + LL::oQs;                                            // (Qfun_sel is in registers by now: 8 x 32 doubles fit S + ... only if the scratch allows; checked by the host)
                const double *Qd = (k + 1 < N) ? Q2 : Qf2;
                __syncthreads();
                if (lane < 32) {
#pragma unroll
                    for (int r = 0; r < 8; r++) Z[r * 32 + lane] = (r < 6 && lane < NV) ? sqrt(fmax(Qd[r * 7], 0.0)) * w[r] : 0.0;
                }
                __syncthreads();
#pragma unroll
                for (int k0 = 0; k0 < 8; k0 += 4) {
                    const double z0 = Z[(k0 + kk) * 32 + ii], z1 = Z[(k0 + kk) * 32 + 16 + ii];
                    q00 = __builtin_amdgcn_mfma_f64_16x16x4f64(z0, z0, q00, 0, 0, 0);
                    q10 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z0, q10, 0, 0, 0);
                    q11 = __builtin_amdgcn_mfma_f64_16x16x4f64(z1, z1, q11, 0, 0, 0);
                }
            }
        }
        if (hasQ) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * r + kk, col = ii;
                if (row < NV && col < NV) Hq[row * NVP + col] = q00[r];
                if (16 + row < NV && col < NV) Hq[(16 + row) * NVP + col] = q10[r];
                if (16 + row < NV && 16 + col < NV) Hq[(16 + row) * NVP + 16 + col] = q11[r];
            }
        }
    }
    __syncthreads();                                                                 // (AB is dead from here on: the scratch region is reused)

    const double s_init = c_s > 1.0 ? 1.0 / c_s : 1.0;
    if (lane < NV) {
        const int k = lane >> 1, j = lane & 1; double f = 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], x[k * 6 + c], f);
        const double viol = f - bx[j];
        s[lane] = viol > 0.0 ? viol + 1.0 : s_init;
    }
    double qmax = 0.0;
    if constexpr (term) { if (lane < S) { lam[lane] = 1.0 / (double)S; qmax = fabs(qsel_r); } qmax = wmax(qmax); }
    const double mu0 = fmax(1.0, 0.01 * (term ? qmax : 1.0));
    if (lane < 4 && !(bu[lane] > 0.0)) atomicOr(&st_sh, LMPC_ST_NOT_INTERIOR);
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
    // step of inequality row r along the direction in (fl, du, ds, dl): F_r dw
    auto rowFd = [&](int r) -> double {
        if (r < 2 * N) return fl[r] - ds[r];
        if (r < 6 * N) { const int q = r - 2 * N, k = q >> 2, j = q & 3; return Fu[j * 2] * du[k * 2] + Fu[j * 2 + 1] * du[k * 2 + 1]; }
        if (r < 8 * N) return -ds[r - 6 * N];
        return -dl[r - 8 * N];
    };
    double t_r[RPL], tp_r[RPL], dt_r[RPL];
#pragma unroll
    for (int j = 0; j < RPL; j++) {
        const int r = lane + WAVE * j;
        t_r[j] = 1.0; tp_r[j] = 0.0; dt_r[j] = 0.0;
        if (r < M) { const double tt = rowb(r) - rowF(r, x, u, s, lam); t_r[j] = tt; m[r] = mu0 / tt; }
    }
    double mcol[7];
#pragma unroll
    for (int j = 0; j < 7; j++) mcol[j] = 0.0;
    double rdiag = 1.0;                                   // 1 / L[lane][lane] of the Cholesky factor of Hc (L itself stays in LDS between factorisation and solves)
    const int lrow = lane < NV ? lane : 0;
    double ru_r = 0.0, rs_r = 0.0, rDs_r = 0.0, rl_r = 0.0;    // residuals of this lane's u / s / lambda row, 1 / D_s of its lane row
    __syncthreads();

    // du = -Hc^-1 g: forward and backward substitution with L (row per lane) and L' (column per lane), the solved entry broadcast by v_readlane
    // (L is read from LDS: row `lane` for the forward pass, column `lane` for the backward pass -- the loads do not depend on the chain and are
    // issued ahead of it; kept in registers, the two copies cost 96 VGPRs and spilled.)
    auto chol_solve = [&](double g) -> double {
        double lr[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) lr[j] = Hs[lrow * NVP + j];
        double y = 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const double yj = rdlane(g * rdiag, j);
            y = lane == j ? yj : y;
            g = fma(-lr[j], yj, g);
        }
#pragma unroll
        for (int k = 0; k < NV; k++) lr[k] = Hs[k * NVP + lrow];
        double z = 0.0;
#pragma unroll
        for (int k = NV - 1; k >= 0; k--) {
            const double zk = rdlane(y * rdiag, k);
            z = lane == k ? zk : z;
            y = fma(-lr[k], zk, y);
        }
        return -z;
    };

    // one Newton-system solve for the right-hand side in h (rows) : result in du, fl, ds, dl (LDS); returns d x_N [lc] in every lane
    auto kkt_solve = [&](double re_sum) -> double {
        if (lane < NV) {                                        // slack elimination of this lane's lane row
            const double hl = h[lane], hs = h[6 * N + lane];
            const double e_ = -(rs_r + hl + hs);
            ee[lane] = e_; eta[lane] = hl + th[lane] * e_ * rDs_r;
        }
        double c_t = 0.0, mc_g = 0.0, y7v = 0.0;
        if constexpr (term) {
            if (lane < S) { c_t = (rl_r + h[8 * N + lane]) * mcol[6]; dl[lane] = c_t * mcol[6]; }      // D^-1/2 c~, staged in dl
        }
        __syncthreads();
        TSTAMP(30);
        double pv = 0.0;
        if constexpr (term) {
            double acc = 0.0;
            if (lg < 7) {
#pragma unroll
                for (int c = lc; c < S; c += 8) acc = lg < 6 ? fma(SS[lg * S + c], dl[c], acc) : acc + dl[c];
            }
            acc = sum_over_c(acc);
            mc_g = lg < 7 ? acc : 0.0;
            pv = term_costate(Ri, mc_g, re_sum, lg, lc, y7v);    // [Ri (Ri' d0 + y7)][lg] in every lane of group lg
            if (lc == 0 && lg < 6) pT[lg] = pv;
        }
        __syncthreads();
        TSTAMP(31);
        double g = 0.0;
        if (lane < NV) {                                        // g = ru - Fu' h_u - V' eta + G' p_T
            const int k = lane >> 1, c = lane & 1;
            g = ru_r;
#pragma unroll
            for (int j = 0; j < 4; j++) g -= Fu[j * 2 + c] * h[2 * N + 4 * k + j];
#pragma unroll
            for (int r = 2; r < 2 * N; r++) g = fma(-W[r * NVP + lane], eta[r], g);
            if constexpr (term) {
#pragma unroll
                for (int j = 0; j < 6; j++) g = fma(G[j * NVP + lane], pT[j], g);
            }
        }
        TSTAMP(32);
        const double dui = chol_solve(g);
        if (lane < NV) du[lane] = dui;
        __syncthreads();
        TSTAMP(33);
        double xiN = 0.0;
        if (lane < NV) {                                        // F_x dx of this lane's lane row, and its slack step
            double f = 0.0;
            if (lane >= 2) {
#pragma unroll
                for (int v = 0; v < NV; v++) f = fma(W[lane * NVP + v], du[v], f);
            }
            fl[lane] = f;
            ds[lane] = (th[lane] * f + ee[lane]) * rDs_r;
        }
        if (lc < 6) {
#pragma unroll
            for (int v = 0; v < NV; v++) xiN = fma(G[lc * NVP + v], du[v], xiN);
        }
        TSTAMP(34);
        if constexpr (term) {
            double wq[7];
            term_omega_w(Ri, y7v, xiN, re_sum, lg, lc, wq);
            double v = -c_t;
#pragma unroll
            for (int j = 0; j < 7; j++) v = fma(mcol[j], wq[j], v);
            __syncthreads();                                    // (dl held D^-1/2 c~ until here)
            if (lane < S) dl[lane] = v * mcol[6];
        }
        __syncthreads();
        return xiN;
    };

    int it = 0, converged = 0, sep = 0;
    double gap = 0.0, rdn = 0.0, ren = 0.0, gap_prev = -1.0, step_prev = INFINITY, step_pp = INFINITY;
    const double qscale = wave_uniform(fmax(1.0, qmax));
#pragma unroll 1
    for (it = 0; it <= p.max_iter; it++) {
        TSTAMP(10);
        // ---- residuals.  The state part of the input rows comes from the adjoint recursion p_k = Ar_k' p_{k+1} + (w_k ; 0),
        //      w_k = 2Q (x_k - xRef) + Fx' mu_lane,k, w_N = 2Qf (x_N - xRef) - T s_T: entries 6, 7 of stage k's product are B_k' p_{k+1} ----
        double gsum = 0.0, rmax = 0.0;
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) gsum = fma(t_r[j], m[r], gsum); }
        if constexpr (term) ss_times<S>(SS, lam, x + N * 6, sT, lane);
        __syncthreads();
        FOR_LANES(i, 6 * N) {
            const int k = i / 6 + 1, c = i % 6;                  // w_k, k = 1 .. N
            const double *Qk = k < N ? Q2 : Qf2;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 6; j++) v = fma(Qk[c * 6 + j], x[k * 6 + j] - xRef[j], v);
            if (k < N) v += Fx[c] * m[2 * k] + Fx[6 + c] * m[2 * k + 1];
            else if (term) v -= T2p[c] * sT[c];
            gam[k * 8 + c] = v;
        }
        __syncthreads();
        {
            double gm[N];
#pragma unroll
            for (int k = 1; k < N; k++) gm[k] = (k & 1) ? gam[k * 8 + lg] : gam[k * 8 + lc];
            double pv = ((N - 1) & 1) ? gam[N * 8 + lc] : gam[N * 8 + lg];           // p_N = w_N in the layout stage N-1 multiplies with
#pragma unroll
            for (int k = N - 1; k >= 0; k--) {
                double pr = ph[k] * pv;
                if (k & 1) { pr = sum_over_c(pr); if (lc == 0 && lg >= 6) btp[2 * k + lg - 6] = pr; pv = pr + (k > 0 ? gm[k] : 0.0); }
                else { pr = sum_over_g(pr); if (lg == 0 && lc >= 6) btp[2 * k + lc - 6] = pr; pv = pr + (k > 0 ? gm[k] : 0.0); }
            }
        }
        __syncthreads();
        if (lane < NV) {
            const int k = lane >> 1, c = lane & 1;
            const double up = k > 0 ? u[lane - 2] : (c == 0 ? uOld0 : uOld1);
            double v = R2[c * 2] * u[k * 2] + R2[c * 2 + 1] * u[k * 2 + 1] + dR2[c] * (u[lane] - up);
            if (k < N - 1) v += dR2[c] * (u[lane] - u[lane + 2]);
#pragma unroll
            for (int j = 0; j < 4; j++) v = fma(Fu[j * 2 + c], m[2 * N + 4 * k + j], v);
            v += btp[lane];
            ru_r = v; rmax = fmax(rmax, fabs(v));
            const double vs = a_s * s[lane] + c_s - m[lane] - m[6 * N + lane];
            rs_r = vs; rmax = fmax(rmax, fabs(vs));
        }
        double lsum = 0.0;
        if constexpr (term) {
            if (lane < S) {
                double v = qsel_r - m[8 * N + lane] + eta_m;
#pragma unroll
                for (int j = 0; j < 6; j++) v = fma(SS[j * S + lane], T2p[j] * sT[j], v);
                rl_r = v; rmax = fmax(rmax, fabs(v)); lsum = lam[lane];
            }
        }
        gap = wave_uniform(wsum(gsum) / (double)M);
        rdn = wmax(rmax);
        const double re_sum = term ? wave_uniform(wsum(lsum) - 1.0) : 0.0;
        ren = fabs(re_sum);
        if (gap < p.tol_gap && rdn < p.tol_res * qscale && ren < p.tol_res && step_bound_ok(step_prev, step_pp)) { converged = 1; break; }
        if (gap_prev >= 0.0) sep = !EQ && gap > LMPC_SEP_THRESHOLD * gap_prev;
        gap_prev = gap;
        if (it == p.max_iter) break;
        if (!(gap == gap) || !(rdn == rdn)) { if (lane == 0) atomicOr(&st_sh, LMPC_ST_NUMERIC); break; }

        TSTAMP(11);
        // ---- factorisation ---------------------------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) th[r] = m[r] * barrier_rt(t_r[j], m[r]); }
        __syncthreads();
        if (lane < NV) {
            const double d_ = frcp(a_s + th[lane] + th[6 * N + lane]);
            rDs_r = d_; dW[lane] = th[lane] * (a_s + th[6 * N + lane]) * d_;          // kappa of this lane row = its weight in W' D W
        }
        int numeric_bad = 0;
        if constexpr (term) {
#pragma unroll
            for (int j = 0; j < 7; j++) mcol[j] = 0.0;
            if (lane < S) {
                const double rs_ = frsqrt(th[8 * N + lane] + p.reg);
#pragma unroll
                for (int j = 0; j < 6; j++) mcol[j] = SS[j * S + lane] * rs_;
                mcol[6] = rs_;
            } else {
                const double tsq = lane < S + 6 ? frsqrt(T2p[lane < S + 6 ? lane - S : 0]) : 0.0;
#pragma unroll
                for (int j = 0; j < 6; j++) if (lane - S == j) mcol[j] = tsq;
            }
#pragma unroll
            for (int j = 0; j < 7; j++) Mt[lane * 8 + j] = mcol[j];
            Mt[lane * 8 + 7] = 0.0;
            __syncthreads();
            {
                typedef double v4d __attribute__((ext_vector_type(4)));
                v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
                const int kk = lane >> 4, ii = lane & 15;
                const bool live = ii < 8;
#pragma unroll
                for (int s_ = 0; s_ < 16; s_ += 4) {
                    double a0 = Mt[(4 * s_ + kk) * 8 + (ii & 7)], a1 = Mt[(4 * (s_ + 1) + kk) * 8 + (ii & 7)], a2 = Mt[(4 * (s_ + 2) + kk) * 8 + (ii & 7)], a3 = Mt[(4 * (s_ + 3) + kk) * 8 + (ii & 7)];
                    a0 = live ? a0 : 0.0; a1 = live ? a1 : 0.0; a2 = live ? a2 : 0.0; a3 = live ? a3 : 0.0;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc3, 0, 0, 0);
                }
                if (ii < 8) { Wl[kk * 8 + ii] = (acc0[0] + acc1[0]) + (acc2[0] + acc3[0]); Wl[(4 + kk) * 8 + ii] = (acc0[1] + acc1[1]) + (acc2[1] + acc3[1]); }
            }
            __syncthreads();
            TSTAMP(50);
            numeric_bad |= term_factor7(Wl, Ri, lane);                                   // R'R = W7 (Cholesky), Ri = R^-1: one row per lane
            __syncthreads();
            TSTAMP(51);
            // Y = P' G, P = Ri[0:6, 0:7]:  rows 2N .. 2N + 6 of W
            FOR_LANES(i, 7 * NV) {
                const int k = i / NV, v = i % NV; double a = 0.0;
#pragma unroll
                for (int r = 0; r < 6; r++) a = fma(Ri[r * 7 + k], G[r * NVP + v], a);          // (Ri is upper triangular: entries below the diagonal are stored zeros)
                W[(2 * N + k) * NVP + v] = a;
            }
        }
        __syncthreads();
        TSTAMP(12);
        {   // Hc = W' D W on the matrix cores: lower tiles (0,0), (1,0), (1,1) of 16 x 16; A operand = d_k W[k][i], B operand = W[k][j]
            typedef double v4d __attribute__((ext_vector_type(4)));
            v4d a00 = {0.0, 0.0, 0.0, 0.0}, a10 = a00, a11 = a00;
            const int kk = lane >> 4, ii = lane & 15;
#pragma unroll
            for (int k0 = 0; k0 < KP; k0 += 4) {
                const int r = k0 + kk;
                const double d_ = dW[r];
                const double w0 = ii < NV ? W[r * NVP + (ii < NV ? ii : 0)] : 0.0;
                const double w1 = 16 + ii < NV ? W[r * NVP + (16 + ii < NV ? 16 + ii : 0)] : 0.0;
                a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0 * d_, w0, a00, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1 * d_, w0, a10, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1 * d_, w1, a11, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * r + kk, col = ii;
                if (row < NV && col < NV) Hs[row * NVP + col] = a00[r];
                if (16 + row < NV && col < NV) Hs[(16 + row) * NVP + col] = a10[r];
                if (16 + row < NV && 16 + col < NV) Hs[(16 + row) * NVP + 16 + col] = a11[r];
            }
        }
        __syncthreads();
        if (lane < NV) {   // the parts of Hc that never pass through W: input cost, input-rate coupling, input box (and the constant state-cost block)
            const int i = lane, k = i >> 1, c = i & 1;
            double hu00 = 0.0, hu01 = 0.0, hu11 = 0.0;
#pragma unroll
            for (int j = 0; j < 4; j++) { const double t_ = th[2 * N + 4 * k + j]; hu00 = fma(Fu[j * 2] * Fu[j * 2], t_, hu00); hu01 = fma(Fu[j * 2] * Fu[j * 2 + 1], t_, hu01); hu11 = fma(Fu[j * 2 + 1] * Fu[j * 2 + 1], t_, hu11); }
            Hs[i * NVP + i] += R2[c * 3] + dR2[c] * (k < N - 1 ? 2.0 : 1.0) + (c == 0 ? hu00 : hu11);
            if (c == 1) Hs[i * NVP + i - 1] += R2[1] + hu01;        // (row 2k + 1, column 2k)
            if (i >= 2) Hs[i * NVP + i - 2] -= dR2[c];
        }
        __syncthreads();
        TSTAMP(13);
        {   // Cholesky Hc = L L', one row per lane, pivot column broadcast by v_readlane (registers only); L goes back to LDS for the solves
            double hrow[NV];
#pragma unroll
            for (int j = 0; j < NV; j++) {
                double v = Hs[lrow * NVP + j];
                if (hasQ) v += Hq[lrow * NVP + j];
                hrow[j] = (lane < NV && j <= lane) ? v : 0.0;
            }
            __syncthreads();
            TSTAMP(40);
            rdiag = 1.0;
#pragma unroll
            for (int j = 0; j < NV; j++) {
                double d_ = rdlane(hrow[j], j);
                if (!(d_ > 0.0)) { numeric_bad = 1; d_ = 1.0; }
                const double ri = frsqrt(d_);
                const double lij = lane == j ? d_ * ri : (lane > j ? hrow[j] * ri : 0.0);
                hrow[j] = lij;
                rdiag = lane == j ? ri : rdiag;
#pragma unroll
                for (int k = j + 1; k < NV; k++) hrow[k] = fma(-lij, rdlane(lij, k), hrow[k]);
            }
            TSTAMP(41);
            if (lane < NV) {
#pragma unroll
                for (int j = 0; j < NV; j++) Hs[lane * NVP + j] = hrow[j];
            }
        }
        if (numeric_bad) { if (lane == 0) atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_NUMERIC); break; }
        __syncthreads();
        TSTAMP(14);
        // ---- predictor: h = t mu rt (= mu wherever the barrier weight is not capped) ------------------------------------------------
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) h[r] = t_r[j] * th[r]; }
        __syncthreads();
        kkt_solve(re_sum);
        TSTAMP(15);
        double apmax = 1.0, admax = 1.0, dma_r[RPL];
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j; dma_r[j] = 0.0;
            if (r < M) {
                const double dta = -rowFd(r), mr = m[r];
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
        gaff = wsum(gaff) / (double)M;
        double sig = gaff / gap; sig = centring_sigma(sig);
        const double tgt = fmax(sig * gap, 0.01 * p.tol_gap);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) { const double mr = m[r]; h[r] = (fma(t_r[j], mr, LMPC_SO_W * tp_r[j]) - tgt) * barrier_rt(t_r[j], mr); } }
        __syncthreads();
        TSTAMP(16);
        const double xiN = kkt_solve(re_sum);
        TSTAMP(17);
        double apx = INFINITY, adx = INFINITY;
#pragma unroll
        for (int j = 0; j < RPL; j++) {
            const int r = lane + WAVE * j;
            if (r < M) {
                const double dtt = -rowFd(r), mr = m[r];
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
            for (int trial = 0; trial < 8; trial++) {
                double pmin = INFINITY, psum = 0.0;
#pragma unroll
                for (int j = 0; j < RPL; j++) {
                    const int r = lane + WAVE * j;
                    if (r < M) { const double pr = (t_r[j] + al * dt_r[j]) * (m[r] + ald * dm[r]); pmin = fmin(pmin, pr); psum += pr; }
                }
                pmin = wmin(pmin); psum = wsum(psum);
                if (pmin >= 1e-2 * psum / (double)M) break;
                al *= 0.7; ald *= 0.7;
            }
        }
        // multiplier of sum(lambda) = 1: mean over the lambda rows of  -rl + dmu - SS' T ds_T,  ds_T = SS dlambda - dx_N
        double deta = 0.0;
        if constexpr (term) {
            if (lg == 0 && lc < 6) z7[lc] = xiN;                                       // d x_N
            __syncthreads();
            ss_times<S>(SS, dl, z7, w7, lane);
            __syncthreads();
            double v = 0.0;
            if (lane < S) { v = -rl_r + dm[8 * N + lane];
#pragma unroll
                for (int j = 0; j < 6; j++) v -= SS[j * S + lane] * T2p[j] * w7[j]; }
            deta = wsum(v) / (double)S;
        }
        __syncthreads();
        TSTAMP(18);
        // ---- step, then the roll-out of the new inputs ---------------------------------------------------------------------------------
        double xo0 = 0.0, xo1 = 0.0;                           // the states before the step (step_bound_ok measures the (x, u) step actually taken; 6 (N + 1) <= 102 entries)
        if (lane < 6 * (N + 1)) xo0 = x[lane];
        if (lane + WAVE < 6 * (N + 1)) xo1 = x[lane + WAVE];
        double smax = lane < NV ? fabs(al * du[lane]) : 0.0;
        if (lane < NV) { u[lane] = fma(al, du[lane], u[lane]); s[lane] = fma(al, ds[lane], s[lane]); }
        if constexpr (term) { if (lane < S) lam[lane] = fma(al, dl[lane], lam[lane]); }
#pragma unroll
        for (int j = 0; j < RPL; j++) { const int r = lane + WAVE * j; if (r < M) { m[r] = fma(ald, dm[r], m[r]); t_r[j] = fma(al, dt_r[j], t_r[j]); } }
        eta_m = wave_uniform(fma(ald, deta, eta_m));
        __syncthreads();
        if (lane >= 2 && lane < NV) fadd[(lane >> 1) * 8 - 8 + 6 + (lane & 1)] = u[lane];   // u_{k+1} rides in the additive term of stage k
        __syncthreads();
        rollout();
        __syncthreads();
        if (lane < 6 * (N + 1)) smax = fmax(smax, fabs(x[lane] - xo0));
        if (lane + WAVE < 6 * (N + 1)) smax = fmax(smax, fabs(x[lane + WAVE] - xo1));
        step_pp = step_prev; step_prev = wave_uniform(wmax(smax));
    }
    TSTAMP(20);
    if (!converged && lane == 0 && !(st_sh & (LMPC_ST_NUMERIC | LMPC_ST_INEXACT)))
        atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_MAXITER);
    if (!p.slacks) {
        double smax = lane < NV ? s[lane] : 0.0;
        smax = wmax(smax);
        if (smax > 1e-8 && lane == 0) atomicOr(&st_sh, LMPC_ST_INFEASIBLE);
    }
    __syncthreads();

    // ---- unpackSolution (:364-379) and feasibleStateInput (:382-384) -------------------------------------
    FOR_LANES(i, 6 * (N + 1)) io.xPred[(size_t)b * 6 * (N + 1) + i] = x[i];
    if (lane < NV) { io.uPred[(size_t)b * NV + lane] = u[lane]; if (io.slack) io.slack[(size_t)b * NV + lane] = s[lane]; }
    if (io.mu) FOR_LANES(r, M) io.mu[(size_t)b * M + r] = m[r];
    if constexpr (term) {
        if (io.lambda && lane < S) io.lambda[(size_t)b * S + lane] = lam[lane];
        if (lane < 6 && io.sTerm) {
            double v = -x[N * 6 + lane];
            for (int c = 0; c < S; c++) v = fma(SS[lane * S + c], lam[c], v);
            io.sTerm[(size_t)b * 6 + lane] = v;
        }
        if ((io.mode & 1) && (io.ztNext || io.ztuNext)) {
            double acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0.0;
            if (lane < S) {
                const int l = lane / p.ppl, cc = lane % p.ppl;
                const double *base = p.sstore + (size_t)p.sslot[l] * LMPC_COLS * p.lap_stride;
                int r1 = sel_start[l] + cc + 1; r1 = r1 > p.sslen[l] - 1 ? p.sslen[l] - 1 : r1;
                const double lv = lam[lane];
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
        if (lane < 6 && io.ztNext) io.ztNext[(size_t)b * 6 + lane] = x[N * 6 + lane];
        if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = u[(N - 1) * 2 + lane];
    }
    TSTAMP(21);
    if (lane == 0) {
        io.status[b] = st_sh; io.iters[b] = it; flag_retry(io, st_sh);
        if (io.resid) { io.resid[(size_t)b * 3] = gap; io.resid[(size_t)b * 3 + 1] = rdn; io.resid[(size_t)b * 3 + 2] = ren; }
    }
}
