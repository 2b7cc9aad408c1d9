// racinglmpc_amd/csrc/lmpc_solve_rt.hip.h -- the RUNTIME-(N, S) solve kernel: any horizon, any safe-set size, no compiler on the box.
//
// The fast solve kernels are templates on the horizon N and the number of safe-set columns S; a pair outside the built-in set needs its own
// shared object (hipcc, ~20 s).  The reference takes any N as a plain parameter (PredictiveControllers.py:63-107, main.py:43), so a box without
// hipcc must still serve any MPCParams.N: this kernel does, with N and S read from the parameter block, every LDS offset computed at launch
// (rt_lds) and plain FP64 multiply-adds instead of matrix-core tiles.  It is the FALLBACK of lmpc_create (flag LMPC_CREATE_RUNTIME_KERNEL,
// include/lmpc_hip.h): built-in and variant kernels stay the fast path; this one is several times slower (one wave per QP, a work-group
// barrier after every small dense step) and exists so that LMPC_E_VARIANT never has to reach a user.
//
// Same QP, same interior-point method and the same rules as lmpc_solve_kernel (one-wave form: multipliers of the dynamics rows from the adjoint
// recursion), written after tests/ipm_model.py (ipm_solve, kkt_factor, kkt_solve) line by line:
//   K2  selection (k2_select_rt: LMPC.addTerminalComponents :386-416, selectPoints :478-514)
//   K3  Mehrotra predictor-corrector on the block-banded KKT system: barrier weights capped (LMPC_TH_INV), lane slacks eliminated, terminal block
//       (lambda, s_T) through a 7-column modified Gram-Schmidt QR (two passes), Riccati recursion on the augmented state (x_k, u_{k-1}),
//       step rules and termination (step_bound_ok) of lmpc_kernels.hip.h
//   epilogue  unpackSolution :364-379, feasibleStateInput :382-384
#pragma once
#include "lmpc_kernels.hip.h"

struct rt_lds {               // offsets in doubles; the same function on host (launch size) and device
    int A, B, C, x, u, s, lam, nu, t, m, th, rt, h, dt, dm, tp, Ds, kap, e, eta, ru, rs, rl, red, Pi, Kx, Ku, Mi, tmp, V, R1, R2, Ri, W7i, sqD, ct, y7, v7, dx, du, ds, dl, k0,
        SS, Qsel, sT, dsT, par, tot;
};
__host__ __device__ inline rt_lds rt_layout(int N, int S) {
    rt_lds L; int o = 0; const int M = 8 * N + S;
    auto take = [&](int n) { const int r = o; o += (n + 1) & ~1; return r; };
    L.A = take(36 * N); L.B = take(12 * N); L.C = take(6 * N); L.x = take(6 * (N + 1)); L.u = take(2 * N); L.s = take(2 * N); L.lam = take(S); L.nu = take(6 * N);
    L.t = take(M); L.m = take(M); L.th = take(M); L.rt = take(M); L.h = take(M); L.dt = take(M); L.dm = take(M); L.tp = take(M);
    L.Ds = take(2 * N); L.kap = take(2 * N); L.e = take(2 * N); L.eta = take(2 * N); L.ru = take(2 * N); L.rs = take(2 * N); L.rl = take(S); L.red = take(6 * N);
    L.Pi = take(64 * N); L.Kx = take(12 * N); L.Ku = take(4 * N); L.Mi = take(4 * N); L.tmp = take(128);
    L.V = take(7 * (S + 6)); L.R1 = take(49); L.R2 = take(49); L.Ri = take(49); L.W7i = take(49); L.sqD = take(S); L.ct = take(S + 6); L.y7 = take(8); L.v7 = take(8);
    L.dx = take(6 * (N + 1)); L.du = take(2 * N); L.ds = take(2 * N); L.dl = take(S); L.k0 = take(2 * N);
    L.SS = take(6 * S); L.Qsel = take(S); L.sT = take(8); L.dsT = take(8); L.par = take(PAR_TOT);
    L.tot = o;
    return L;
}

// K2 with runtime N, S (one wave): the selection of k2_select, statement for statement
__device__ inline void k2_select_rt(const lmpc_dev_params &p, const lmpc_solve_io &io, int b, int lane, int N, int S, double *SS, double *Qsel, int *sel_start, int *st_sh) {
    if (io.mode & 1) {
        double ztv[6];
        for (int j = 0; j < 6; j++) ztv[j] = io.zt[(size_t)b * 6 + j];
        const double x04 = io.x0[(size_t)b * 6 + 4];
        if (ztv[4] - x04 > p.TL / 2) ztv[4] = fmax(ztv[4] - p.TL, 0.0);        // :392-393
        if (io.ztUsed && lane < 6) io.ztUsed[(size_t)b * 6 + lane] = ztv[lane < 6 ? lane : 0];
        const int hasPred = io.hasPred ? io.hasPred[b] : 0;                     // Q-function shift bookkeeping (:502-512)
        int crossed = 0;
        if (hasPred) {
            for (int r0 = 0; r0 <= N; r0 += WAVE) {
                const int r = r0 + lane;
                const int c_ = (r <= N && io.xPredPrev[((size_t)b * (N + 1) + (r <= N ? r : 0)) * 6 + 4] > p.TL) ? 1 : 0;
                crossed += (int)__popcll(__ballot(c_));
            }
        }
        const int tstep = io.timeStep ? io.timeStep[b] : 0;
        const int ppl = p.ppl, npw = ppl + 1;
        for (int l = 0; l < p.L; l++) {
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
            for (int cc = lane; cc < ppl; cc += WAVE) {
                int r0 = start + cc; r0 = r0 > T - 1 ? T - 1 : r0;
                int r1 = start + cc + 1; r1 = r1 > T - 1 ? T - 1 : r1;
                const int col = l * ppl + cc;
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
        for (int c = lane; c < S; c += WAVE) {
            for (int j = 0; j < 6; j++) SS[j * S + c] = io.ssSelIn[((size_t)b * S + c) * 6 + j];
            Qsel[c] = io.qSelIn[(size_t)b * S + c];
        }
    }
}

#define RT_SYNC() __syncthreads()
#define RT_FOR(i, n) for (int i = lane; i < (n); i += WAVE)

// EQ: the retry pass (equal primal / dual steps, neighbourhood safeguard), as in lmpc_solve_kernel<N, S, true>
template <bool EQ>
__global__ __launch_bounds__(WAVE) void lmpc_solve_kernel_rt(lmpc_dev_params p, int B, lmpc_solve_io io) {
    extern __shared__ double sm[];
    const int b = blockIdx.x;
    if (b >= B) return;
    if constexpr (EQ) { if (!(io.status[b] & (LMPC_ST_MAXITER | LMPC_ST_NUMERIC))) return; }
    const int lane = threadIdx.x;
    const int N = p.N, S = p.S, M = 8 * N + S;
    const bool term = S > 0;
    const rt_lds L = rt_layout(N, S);
    double *A = sm + L.A, *Bm = sm + L.B, *C = sm + L.C, *x = sm + L.x, *u = sm + L.u, *s = sm + L.s, *lam = sm + L.lam, *nu = sm + L.nu;
    double *t = sm + L.t, *m = sm + L.m, *th = sm + L.th, *rt = sm + L.rt, *h = sm + L.h, *dt = sm + L.dt, *dm = sm + L.dm, *tp = sm + L.tp;
    double *Ds = sm + L.Ds, *kap = sm + L.kap, *ee = sm + L.e, *eta = sm + L.eta, *ru = sm + L.ru, *rs = sm + L.rs, *rl = sm + L.rl, *red = sm + L.red;
    double *Pi = sm + L.Pi, *Kx = sm + L.Kx, *Ku = sm + L.Ku, *Mi = sm + L.Mi, *tmp = sm + L.tmp;
    double *V = sm + L.V, *R1 = sm + L.R1, *R2m = sm + L.R2, *Ri = sm + L.Ri, *W7i = sm + L.W7i, *sqD = sm + L.sqD, *ct = sm + L.ct, *y7 = sm + L.y7, *v7 = sm + L.v7;
    double *dx = sm + L.dx, *du = sm + L.du, *ds = sm + L.ds, *dl = sm + L.dl, *k0 = sm + L.k0;
    double *SS = sm + L.SS, *Qsel = sm + L.Qsel, *sT = sm + L.sT, *dsT = sm + L.dsT, *par = sm + L.par;
    const double *Fx = par + PAR_FX, *Fu = par + PAR_FU, *bx = par + PAR_BX, *bu = par + PAR_BU, *Q2 = par + PAR_Q2, *Qf2 = par + PAR_QF2,
                 *R2 = par + PAR_R2, *dR2 = par + PAR_DR2, *T2p = par + PAR_T2, *xRef = par + PAR_XREF;
    __shared__ int st_sh;
    __shared__ int sel_start[LMPC_MAX_USED_LAPS];
    if (lane == 0) st_sh = EQ ? (io.status[b] & (LMPC_ST_REG_SINGULAR | LMPC_ST_NO_SEGMENT)) : 0;
    if (lane < 12) par[PAR_FX + lane] = p.Fx[lane];
    if (lane < 8) par[PAR_FU + lane] = p.Fu[lane];
    if (lane < 2) { par[PAR_BX + lane] = p.bx[lane]; par[PAR_DR2 + lane] = p.dR2[lane]; }
    if (lane < 4) { par[PAR_BU + lane] = p.bu[lane]; par[PAR_R2 + lane] = p.R2[lane]; }
    if (lane < 36) { par[PAR_Q2 + lane] = p.Q2[lane]; par[PAR_QF2 + lane] = p.Qf2[lane]; }
    if (lane < 6) { par[PAR_T2 + lane] = p.T2[lane]; par[PAR_XREF + lane] = p.xRef[lane]; }
    if (lane == 0) { par[PAR_AS] = p.a_s; par[PAR_CS] = p.c_s; }
    RT_SYNC();
    const double a_s = par[PAR_AS], c_s = par[PAR_CS];
    if (term) { k2_select_rt(p, io, b, lane, N, S, SS, Qsel, sel_start, &st_sh); RT_SYNC(); }
    if (!EQ && io.rstatus) { RT_FOR(i, N) { const int rs_ = io.rstatus[(size_t)b * N + i]; if (rs_) atomicOr(&st_sh, rs_); } }
    if (!(io.mode & 2)) { RT_SYNC(); if (lane == 0) io.status[b] = st_sh; return; }

    // ---- K3 --------------------------------------------------------------------------------------------------------------------------
    RT_FOR(i, 36 * N) A[i] = io.A[(size_t)b * 36 * N + i];
    RT_FOR(i, 12 * N) Bm[i] = io.Bm[(size_t)b * 12 * N + i];
    RT_FOR(i, 6 * N) C[i] = io.C[(size_t)b * 6 * N + i];
    if (lane < 6) x[lane] = io.x0[(size_t)b * 6 + lane];
    RT_FOR(i, 2 * N) u[i] = 0.0;
    const double uOld0 = io.uOld[(size_t)b * 2 + 0], uOld1 = io.uOld[(size_t)b * 2 + 1];
    RT_SYNC();
    for (int k = 0; k < N; k++) {                          // strictly interior start: u = 0, x by roll-out
        if (lane < 6) {
            double v = C[k * 6 + lane];
            for (int j = 0; j < 6; j++) v = fma(A[k * 36 + lane * 6 + j], x[k * 6 + j], v);
            x[(k + 1) * 6 + lane] = v;
        }
        RT_SYNC();
    }
    const double s_init = c_s > 1.0 ? 1.0 / c_s : 1.0;      // (see lmpc_solve_kernel)
    RT_FOR(i, 2 * N) {
        const int k = i >> 1, j = i & 1; double f = 0.0;
        for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], x[k * 6 + c], f);
        const double viol = f - bx[j];
        s[i] = viol > 0.0 ? viol + 1.0 : s_init;
    }
    double qmax = 0.0;
    if (term) { RT_FOR(c, S) { lam[c] = 1.0 / (double)S; qmax = fmax(qmax, fabs(Qsel[c])); } }
    qmax = wmax(qmax);
    const double mu0 = fmax(1.0, 0.01 * (term ? qmax : 1.0));
    if (lane < 4 && !(bu[lane] > 0.0)) atomicOr(&st_sh, LMPC_ST_NOT_INTERIOR);
    double eta_m = 0.0;
    RT_SYNC();
    // inequality rows in the reference's order: lane rows (2N) | input rows (4N) | slack positivity (2N) | lambda >= 0 (S)
    auto rowF = [&](int r, const double *xx, const double *uu, const double *ss_, const double *ll) -> double {
        if (r < 2 * N) { const int k = r >> 1, j = r & 1; double f = 0.0;
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], xx[k * 6 + c], f);
            return f - ss_[r]; }
        if (r < 6 * N) { const int q = r - 2 * N, k = q >> 2, j = q & 3; return Fu[j * 2] * uu[k * 2] + Fu[j * 2 + 1] * uu[k * 2 + 1]; }
        if (r < 8 * N) return -ss_[r - 6 * N];
        return -ll[r - 8 * N];
    };
    auto rowb = [&](int r) -> double { if (r < 2 * N) return bx[r & 1]; if (r < 6 * N) return bu[(r - 2 * N) & 3]; return 0.0; };
    RT_FOR(r, M) { const double tt = rowb(r) - rowF(r, x, u, s, lam); t[r] = tt; m[r] = mu0 / tt; tp[r] = 0.0; }
    RT_SYNC();

    // one Newton-system solve for the right-hand side in h (tests/ipm_model.py: kkt_solve with gx = 0, gu = ru, gs = rs, gl = rl + h_l)
    auto kkt_solve = [&](double re_sum) {
        RT_FOR(i, 2 * N) {
            const double e_ = -(rs[i] + h[i] + h[6 * N + i]);
            ee[i] = e_; eta[i] = h[i] + th[i] * e_ / Ds[i];
        }
        double *pv = tmp;                                   // costate (8), then per-stage scratch: z (6) at tmp + 8, zu (2) at + 14, mx (6) at + 16, mu_ (2) at + 22
        if (lane < 8) pv[lane] = 0.0;
        if (term) {
            RT_FOR(c, S + 6) ct[c] = c < S ? (rl[c] + h[8 * N + c]) / sqD[c] : 0.0;
            RT_SYNC();
            for (int j = 0; j < 7; j++) {                   // y7 = Qm' c~
                double a = 0.0;
                RT_FOR(r, S + 6) a = fma(V[j * (S + 6) + r], ct[r], a);
                a = wsum(a);
                if (lane == 0) y7[j] = a;
            }
            RT_SYNC();
            if (lane < 7) v7[lane] = fma(Ri[6 * 7 + lane], -re_sum, y7[lane]);       // Ri' d0 + y7, d0 = (0, .., 0, -re_sum)
            RT_SYNC();
            if (lane < 6) { double a = 0.0; for (int j = 0; j < 7; j++) a = fma(Ri[lane * 7 + j], v7[j], a); pv[lane] = a; }
        }
        RT_SYNC();
        for (int k = N - 1; k >= 0; k--) {
            const double *Pk = Pi + k * 64, *Ak = A + k * 36, *Bk = Bm + k * 12;
            if (lane < 8) {                                 // z = Pi[:, :6] c + pv,  c = -re_dyn[k]
                double a = pv[lane];
                for (int j = 0; j < 6; j++) a = fma(Pk[lane * 8 + j], -red[k * 6 + j], a);
                tmp[8 + lane] = a;
            }
            RT_SYNC();
            if (lane < 6) {                                 // mx = -Fx' eta_k + A' z
                double a = -(Fx[lane] * eta[2 * k] + Fx[6 + lane] * eta[2 * k + 1]);
                for (int j = 0; j < 6; j++) a = fma(Ak[j * 6 + lane], tmp[8 + j], a);
                tmp[16 + lane] = a;
            } else if (lane < 8) {                          // mu_ = ru_k - Fu' h_u[k] + B' z + zu
                const int c = lane - 6;
                double a = ru[2 * k + c];
                for (int j = 0; j < 4; j++) a -= Fu[j * 2 + c] * h[2 * N + 4 * k + j];
                for (int j = 0; j < 6; j++) a = fma(Bk[j * 2 + c], tmp[8 + j], a);
                tmp[16 + lane] = a + tmp[14 + c];
            }
            RT_SYNC();
            if (lane < 6) pv[lane] = tmp[16 + lane] - (Kx[k * 12 + lane] * tmp[22] + Kx[k * 12 + 6 + lane] * tmp[23]);      // mx - Kx' mu_
            else if (lane < 8) { const int c = lane - 6; pv[lane] = -(Ku[k * 4 + c] * tmp[22] + Ku[k * 4 + 2 + c] * tmp[23]);       // -Ku' mu_
                                 k0[2 * k + c] = Mi[k * 4 + c * 2] * tmp[22] + Mi[k * 4 + c * 2 + 1] * tmp[23]; }
            RT_SYNC();
        }
        if (lane < 6) dx[lane] = 0.0;
        RT_SYNC();
        for (int k = 0; k < N; k++) {
            if (lane < 2) {                                 // du_k = -Kx dx_k - Ku du_{k-1} - k0_k
                double a = -k0[2 * k + lane];
                for (int j = 0; j < 6; j++) a -= Kx[k * 12 + lane * 6 + j] * dx[k * 6 + j];
                if (k > 0) a -= Ku[k * 4 + lane * 2] * du[2 * k - 2] + Ku[k * 4 + lane * 2 + 1] * du[2 * k - 1];
                du[2 * k + lane] = a;
            }
            RT_SYNC();
            if (lane < 6) {
                double a = -red[k * 6 + lane];
                for (int j = 0; j < 6; j++) a = fma(A[k * 36 + lane * 6 + j], dx[k * 6 + j], a);
                a = fma(Bm[k * 12 + lane * 2], du[2 * k], a); a = fma(Bm[k * 12 + lane * 2 + 1], du[2 * k + 1], a);
                dx[(k + 1) * 6 + lane] = a;
            }
            RT_SYNC();
        }
        RT_FOR(i, 2 * N) {
            const int k = i >> 1, j = i & 1; double f = 0.0;
            for (int c = 0; c < 6; c++) f = fma(Fx[j * 6 + c], dx[k * 6 + c], f);
            ds[i] = (th[i] * f + ee[i]) / Ds[i];
        }
        if (term) {
            if (lane < 7) { double a = 0.0;                                           // z7 = Ri' d7 + y7, d7 = (dx_N ; -re_sum)
                for (int j = 0; j < 7; j++) a = fma(Ri[j * 7 + lane], j < 6 ? dx[N * 6 + j] : -re_sum, a);
                v7[lane] = a + y7[lane]; }
            RT_SYNC();
            RT_FOR(c, S) {                                  // dl = (-c~ + Qm z7) / sqrt(D)
                double a = -ct[c];
                for (int j = 0; j < 7; j++) a = fma(V[j * (S + 6) + c], v7[j], a);
                dl[c] = a / sqD[c];
            }
        }
        RT_SYNC();
    };

    int it = 0, converged = 0, sep = 0;
    double gap = 0.0, rdn = 0.0, ren = 0.0, gap_prev = -1.0, step_prev = INFINITY, step_pp = INFINITY;
    const double qscale = fmax(1.0, qmax);
#pragma unroll 1
    for (it = 0; it <= p.max_iter; it++) {
        // ---- residuals; the multipliers of the dynamics rows come from the adjoint recursion, so the x rows of the dual residual vanish ----
        double gsum = 0.0;
        RT_FOR(r, M) gsum = fma(t[r], m[r], gsum);
        gap = wsum(gsum) / (double)M;
        if (term) {
            if (lane < 6) { double a = -x[N * 6 + lane]; for (int c = 0; c < S; c++) a = fma(SS[lane * S + c], lam[c], a); sT[lane] = a; }
            RT_SYNC();
        }
        if (lane < 6) {
            double a = 0.0;
            for (int j = 0; j < 6; j++) a = fma(Qf2[lane * 6 + j], x[N * 6 + j] - xRef[j], a);
            if (term) a -= T2p[lane] * sT[lane];
            nu[(N - 1) * 6 + lane] = -a;
        }
        RT_SYNC();
        for (int k = N - 1; k >= 1; k--) {
            if (lane < 6) {
                double a = 0.0;
                for (int j = 0; j < 6; j++) a = fma(A[k * 36 + j * 6 + lane], nu[k * 6 + j], a);
                double w = Fx[lane] * m[2 * k] + Fx[6 + lane] * m[2 * k + 1];
                for (int j = 0; j < 6; j++) w = fma(Q2[lane * 6 + j], x[k * 6 + j] - xRef[j], w);
                nu[(k - 1) * 6 + lane] = a - w;
            }
            RT_SYNC();
        }
        double rmax = 0.0, remax = 0.0, lsum = 0.0;
        RT_FOR(i, 2 * N) {
            const int k = i >> 1, c = i & 1;
            const double up = k > 0 ? u[(k - 1) * 2 + c] : (c == 0 ? uOld0 : uOld1);
            double v = R2[c * 2] * u[k * 2] + R2[c * 2 + 1] * u[k * 2 + 1] + dR2[c] * (u[i] - up);
            if (k < N - 1) v += dR2[c] * (u[i] - u[(k + 1) * 2 + c]);
            for (int j = 0; j < 4; j++) v = fma(Fu[j * 2 + c], m[2 * N + 4 * k + j], v);
            for (int j = 0; j < 6; j++) v -= Bm[k * 12 + j * 2 + c] * nu[k * 6 + j];
            ru[i] = v; rmax = fmax(rmax, fabs(v));
            const double vs = a_s * s[i] + c_s - m[i] - m[6 * N + i];
            rs[i] = vs; rmax = fmax(rmax, fabs(vs));
        }
        if (term) {
            RT_FOR(c, S) {
                double v = Qsel[c] - m[8 * N + c] + eta_m;
                for (int j = 0; j < 6; j++) v = fma(SS[j * S + c], T2p[j] * sT[j], v);
                rl[c] = v; rmax = fmax(rmax, fabs(v)); lsum += lam[c];
            }
        }
        RT_FOR(i, 6 * N) {
            const int k = i / 6, c = i % 6;
            double v = x[(k + 1) * 6 + c] - C[i] - Bm[k * 12 + c * 2] * u[k * 2] - Bm[k * 12 + c * 2 + 1] * u[k * 2 + 1];
            for (int j = 0; j < 6; j++) v -= A[k * 36 + c * 6 + j] * x[k * 6 + j];
            red[i] = v; remax = fmax(remax, fabs(v));
        }
        rdn = wmax(rmax);
        const double re_sum = term ? wsum(lsum) - 1.0 : 0.0;
        ren = fmax(wmax(remax), fabs(re_sum));
        if (gap < p.tol_gap && rdn < p.tol_res * qscale && ren < p.tol_res && step_bound_ok(step_prev, step_pp)) { converged = 1; break; }
        if (gap_prev >= 0.0) sep = !EQ && gap > LMPC_SEP_THRESHOLD * gap_prev;
        gap_prev = gap;
        if (it == p.max_iter) break;
        if (!(gap == gap) || !(rdn == rdn)) { if (lane == 0) atomicOr(&st_sh, LMPC_ST_NUMERIC); break; }

        // ---- barrier weights (capped: LMPC_TH_INV), slack elimination, terminal factor ----------------------------------------------------
        RT_FOR(r, M) { const double rr = 1.0 / fmax(t[r], m[r] * LMPC_TH_INV); rt[r] = rr; th[r] = m[r] * rr; h[r] = t[r] * m[r] * rr; }   // h: the predictor's right-hand side
        RT_SYNC();
        RT_FOR(i, 2 * N) { const double d_ = a_s + th[i] + th[6 * N + i]; Ds[i] = d_; kap[i] = th[i] * (a_s + th[6 * N + i]) / d_; }
        int bad = 0;
        if (term) {
            // M' = [ (E / sqrt(D))' ; diag(T^-1/2) (6 rows, 7th column zero) ] ((S + 6) x 7, column-major in V), E = [SS; 1'], D = theta_lambda + reg
            const int SR = S + 6;
            RT_FOR(c, S) sqD[c] = sqrt(th[8 * N + c] + p.reg);
            RT_SYNC();
            RT_FOR(i, 7 * SR) {
                const int j = i / SR, r = i % SR;
                V[i] = r < S ? (j < 6 ? SS[j * S + r] : 1.0) / sqD[r] : ((j < 6 && r - S == j) ? 1.0 / sqrt(T2p[j]) : 0.0);
            }
            RT_SYNC();
            for (int pass = 0; pass < 2; pass++) {          // modified Gram-Schmidt, twice (tests/ipm_model.py: mgs_QR)
                double *R = pass == 0 ? R1 : R2m;
                if (lane < 49) R[lane] = 0.0;
                RT_SYNC();
                for (int i = 0; i < 7; i++) {
                    double a = 0.0;
                    RT_FOR(r, SR) a = fma(V[i * SR + r], V[i * SR + r], a);
                    a = wsum(a);
                    if (!(a > 0.0)) { bad = 1; a = 1.0; }
                    const double rii = sqrt(a), ri = 1.0 / rii;
                    RT_FOR(r, SR) V[i * SR + r] *= ri;
                    if (lane == 0) R[i * 7 + i] = rii;
                    RT_SYNC();
                    for (int j = i + 1; j < 7; j++) {
                        double d_ = 0.0;
                        RT_FOR(r, SR) d_ = fma(V[i * SR + r], V[j * SR + r], d_);
                        d_ = wsum(d_);
                        RT_FOR(r, SR) V[j * SR + r] = fma(-d_, V[i * SR + r], V[j * SR + r]);
                        if (lane == 0) R[i * 7 + j] = d_;
                    }
                    RT_SYNC();
                }
            }
            if (lane < 49) { const int i = lane / 7, j = lane % 7; double a = 0.0; for (int k = 0; k < 7; k++) a = fma(R2m[i * 7 + k], R1[k * 7 + j], a); tmp[64 + lane] = a; }   // R = R2 R1
            RT_SYNC();
            if (lane < 7) {                                 // Ri = R^-1 (upper), column `lane` by back substitution
                const double *R = tmp + 64; const int j = lane;
                for (int i = 6; i >= 0; i--) {
                    double a = i == j ? 1.0 : 0.0;
                    for (int k = i + 1; k <= j; k++) a -= R[i * 7 + k] * Ri[k * 7 + j];
                    Ri[i * 7 + j] = i <= j ? a / R[i * 7 + i] : 0.0;
                }
            }
            RT_SYNC();
            if (lane < 49) { const int i = lane / 7, j = lane % 7; double a = 0.0; for (int k = 0; k < 7; k++) a = fma(Ri[i * 7 + k], Ri[j * 7 + k], a); W7i[lane] = a; }
            RT_SYNC();
        }
        // ---- Riccati recursion on the augmented state (x_k, u_{k-1}) (tests/ipm_model.py: kkt_factor) -------------------------------------
        {
            double *P0 = Pi + (N - 1) * 64;                 // Pi_N = [[2 Qf + W7^-1[0:6,0:6], 0], [0, 0]]
            { const int r = lane >> 3, c = lane & 7; P0[lane] = (r < 6 && c < 6) ? Qf2[r * 6 + c] + (term ? W7i[r * 7 + c] : 0.0) : 0.0; }
            RT_SYNC();
            for (int k = N - 1; k >= 0; k--) {
                const double *Pk = Pi + k * 64, *Ak = A + k * 36, *Bk = Bm + k * 12;
                double *PA = tmp, *T2 = tmp + 36, *Mxx = tmp + 48, *Mxu = tmp + 84, *Muu = tmp + 96;       // (tmp: 128 doubles; R of the terminal factor is dead by now)
                if (lane < 36) { const int r = lane / 6, c = lane % 6; double a = 0.0; for (int j = 0; j < 6; j++) a = fma(Pk[r * 8 + j], Ak[j * 6 + c], a); PA[lane] = a; }
                else if (lane < 48) { const int i = lane - 36, r = i >> 1, c = i & 1; double a = Pk[r * 8 + 6 + c]; for (int j = 0; j < 6; j++) a = fma(Pk[r * 8 + j], Bk[j * 2 + c], a); T2[i] = a; }
                RT_SYNC();
                if (lane < 36) {                            // Mxx = Hx + A' (Pxx A),  Hx = 2Q + Fx' diag(kappa_k) Fx
                    const int r = lane / 6, c = lane % 6;
                    double a = Q2[r * 6 + c] + kap[2 * k] * Fx[r] * Fx[c] + kap[2 * k + 1] * Fx[6 + r] * Fx[6 + c];
                    for (int j = 0; j < 6; j++) a = fma(Ak[j * 6 + r], PA[j * 6 + c], a);
                    Mxx[lane] = a;
                } else if (lane < 48) {                     // Mxu = A' T2
                    const int i = lane - 36, r = i >> 1, c = i & 1; double a = 0.0;
                    for (int j = 0; j < 6; j++) a = fma(Ak[j * 6 + r], T2[j * 2 + c], a);
                    Mxu[i] = a;
                } else if (lane < 52) {                     // Muu = Hu + diag(2 dR) + B' T2 + Pxu' B + Puu,  Hu = 2R + Fu' diag(theta_u[k]) Fu
                    const int i = lane - 48, r = i >> 1, c = i & 1;
                    double a = R2[r * 2 + c] + (r == c ? dR2[r] : 0.0) + Pk[(6 + r) * 8 + 6 + c];
                    for (int j = 0; j < 4; j++) a = fma(th[2 * N + 4 * k + j] * Fu[j * 2 + r], Fu[j * 2 + c], a);
                    for (int j = 0; j < 6; j++) a = fma(Bk[j * 2 + r], T2[j * 2 + c], fma(Pk[j * 8 + 6 + r], Bk[j * 2 + c], a));
                    Muu[i] = a;
                }
                RT_SYNC();
                {
                    const double m00 = Muu[0], m01 = 0.5 * (Muu[1] + Muu[2]), m11 = Muu[3], det = m00 * m11 - m01 * m01;
                    if (!(det > 0.0) || !(m00 > 0.0)) bad = 1;
                    const double rdet = 1.0 / det, i00 = m11 * rdet, i01 = -m01 * rdet, i11 = m00 * rdet;
                    if (lane < 4) Mi[k * 4 + lane] = lane == 0 ? i00 : (lane == 3 ? i11 : i01);
                    if (lane < 12) { const int r = lane / 6, c = lane % 6; Kx[k * 12 + lane] = (r == 0 ? i00 : i01) * Mxu[c * 2] + (r == 0 ? i01 : i11) * Mxu[c * 2 + 1]; }     // Kx = Muu^-1 Mxu'
                    else if (lane < 16) { const int i = lane - 12, r = i >> 1, c = i & 1; Ku[k * 4 + i] = -(r == 0 ? (c == 0 ? i00 : i01) : (c == 0 ? i01 : i11)) * dR2[c]; }   // Ku = Muu^-1 (-diag(2 dR))
                }
                RT_SYNC();
                if (k > 0) {                                // Pi_k (8 x 8), symmetrised
                    double *Pn = Pi + (k - 1) * 64;
                    const int r = lane >> 3, c = lane & 7;
                    auto entry = [&](int r_, int c_) -> double {
                        if (r_ < 6 && c_ < 6) return Mxx[r_ * 6 + c_] - (Mxu[r_ * 2] * Kx[k * 12 + c_] + Mxu[r_ * 2 + 1] * Kx[k * 12 + 6 + c_]);
                        if (r_ < 6) return -(Mxu[r_ * 2] * Ku[k * 4 + (c_ - 6)] + Mxu[r_ * 2 + 1] * Ku[k * 4 + 2 + (c_ - 6)]);
                        if (c_ < 6) return -(Mxu[c_ * 2] * Ku[k * 4 + (r_ - 6)] + Mxu[c_ * 2 + 1] * Ku[k * 4 + 2 + (r_ - 6)]);
                        return (r_ == c_ ? dR2[r_ - 6] : 0.0) + dR2[r_ - 6] * Ku[k * 4 + (r_ - 6) * 2 + (c_ - 6)];
                    };
                    Pn[lane] = 0.5 * (entry(r, c) + entry(c, r));
                }
                RT_SYNC();
            }
        }
        if (bad) { if (lane == 0) atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_NUMERIC); break; }

        // ---- predictor (affine scaling) ----------------------------------------------------------------------------------------------------
        kkt_solve(re_sum);
        double apmax = 1.0, admax = 1.0;
        RT_FOR(r, M) {
            const double dta = -rowF(r, dx, du, ds, dl), dma = -h[r] - th[r] * dta;
            dt[r] = dta; dm[r] = dma;
            if (dta < 0.0) apmax = fmin(apmax, -t[r] / dta);
            if (dma < 0.0) admax = fmin(admax, -m[r] / dma);
        }
        apmax = wmin(apmax); admax = wmin(admax);
        if (!sep) { apmax = fmin(apmax, admax); admax = apmax; }
        double gaff = 0.0;
        RT_FOR(r, M) { gaff = fma(t[r] + apmax * dt[r], m[r] + admax * dm[r], gaff); tp[r] = dt[r] * dm[r]; }
        gaff = wsum(gaff) / (double)M;
        double sig = gaff / gap; sig = centring_sigma(sig);
        const double tgt = fmax(sig * gap, 0.01 * p.tol_gap);
        // ---- corrector ----------------------------------------------------------------------------------------------------------------------
        RT_SYNC();
        RT_FOR(r, M) h[r] = (fma(t[r], m[r], LMPC_SO_W * tp[r]) - tgt) * rt[r];
        RT_SYNC();
        kkt_solve(re_sum);
        double apx = INFINITY, adx = INFINITY;
        RT_FOR(r, M) {
            const double dtt = -rowF(r, dx, du, ds, dl), dmm = -h[r] - th[r] * dtt;
            dt[r] = dtt; dm[r] = dmm;
            if (dtt < 0.0) apx = fmin(apx, -t[r] / dtt);
            if (dmm < 0.0) adx = fmin(adx, -m[r] / dmm);
        }
        apx = wmin(apx); adx = wmin(adx);
        const double frac = EQ ? 0.995 : step_fraction(sig, gap);
        double al = fmin(1.0, frac * apx), ald = fmin(1.0, frac * adx);
        if (!sep) { al = fmin(al, ald); ald = al; }
        RT_SYNC();
        if constexpr (EQ) {                                 // (retry pass: stay in a wide neighbourhood of the central path, see lmpc_solve_kernel)
            for (int trial = 0; trial < 8; trial++) {
                double pmin = INFINITY, psum = 0.0;
                RT_FOR(r, M) { const double pr = (t[r] + al * dt[r]) * (m[r] + ald * dm[r]); pmin = fmin(pmin, pr); psum += pr; }
                pmin = wmin(pmin); psum = wsum(psum);
                if (pmin >= 1e-2 * psum / (double)M) break;
                al *= 0.7; ald *= 0.7;
            }
        }
        // ---- step ---------------------------------------------------------------------------------------------------------------------------
        double deta = 0.0;
        if (term) {
            if (lane < 6) { double a = -dx[N * 6 + lane]; for (int c = 0; c < S; c++) a = fma(SS[lane * S + c], dl[c], a); dsT[lane] = a; }
            RT_SYNC();
            double v = 0.0;
            RT_FOR(c, S) { v += -rl[c] + dm[8 * N + c]; for (int j = 0; j < 6; j++) v -= SS[j * S + c] * T2p[j] * dsT[j]; }
            deta = wsum(v) / (double)S;
        }
        double smax = 0.0;
        RT_FOR(i, 6 * (N + 1)) { smax = fmax(smax, fabs(al * dx[i])); x[i] = fma(al, dx[i], x[i]); }
        RT_FOR(i, 2 * N) { smax = fmax(smax, fabs(al * du[i])); u[i] = fma(al, du[i], u[i]); s[i] = fma(al, ds[i], s[i]); }
        step_pp = step_prev; step_prev = wmax(smax);
        if (term) { RT_FOR(c, S) lam[c] = fma(al, dl[c], lam[c]); }
        RT_FOR(r, M) { t[r] = fma(al, dt[r], t[r]); m[r] = fma(ald, dm[r], m[r]); }
        eta_m = fma(ald, deta, eta_m);
        RT_SYNC();
    }
    if (!converged && lane == 0 && !(st_sh & (LMPC_ST_NUMERIC | LMPC_ST_INEXACT)))
        atomicOr(&st_sh, (gap < 1e-9 && rdn < 1e-5 * qscale && ren < 1e-7) ? LMPC_ST_INEXACT : LMPC_ST_MAXITER);
    if (!p.slacks) {                                        // hard lane rows exceeded: the hard problem is infeasible (see lmpc_solve_kernel)
        double smax = 0.0;
        RT_FOR(i, 2 * N) smax = fmax(smax, s[i]);
        smax = wmax(smax);
        if (smax > 1e-8 && lane == 0) atomicOr(&st_sh, LMPC_ST_INFEASIBLE);
    }
    RT_SYNC();
    // ---- unpackSolution (:364-379) and feasibleStateInput (:382-384) -----------------------------------------------------------------------
    RT_FOR(i, 6 * (N + 1)) io.xPred[(size_t)b * 6 * (N + 1) + i] = x[i];
    RT_FOR(i, 2 * N) { io.uPred[(size_t)b * 2 * N + i] = u[i]; if (io.slack) io.slack[(size_t)b * 2 * N + i] = s[i]; }
    if (io.mu) { RT_FOR(r, M) io.mu[(size_t)b * M + r] = m[r]; }
    if (term) {
        if (io.lambda) { RT_FOR(c, S) io.lambda[(size_t)b * S + c] = lam[c]; }
        if (lane < 6 && io.sTerm) {
            double v = -x[N * 6 + lane];
            for (int c = 0; c < S; c++) v = fma(SS[lane * S + c], lam[c], v);
            io.sTerm[(size_t)b * 6 + lane] = v;
        }
        if ((io.mode & 1) && (io.ztNext || io.ztuNext)) {   // zt = Succ_SS lambda, zt_u = Succ_uSS lambda
            double acc[8];
            for (int j = 0; j < 8; j++) acc[j] = 0.0;
            RT_FOR(c, S) {
                const int l = c / p.ppl, cc = c % p.ppl;
                const double *base = p.sstore + (size_t)p.sslot[l] * LMPC_COLS * p.lap_stride;
                int r1 = sel_start[l] + cc + 1; r1 = r1 > p.sslen[l] - 1 ? p.sslen[l] - 1 : r1;
                const double lv = lam[c];
                for (int j = 0; j < 8; j++) acc[j] = fma(base[j * p.lap_stride + r1], lv, acc[j]);
            }
            for (int j = 0; j < 8; j++) acc[j] = wsum(acc[j]);
            if (lane < 6 && io.ztNext) { double v = acc[0]; for (int j = 1; j < 6; j++) if (lane == j) v = acc[j]; io.ztNext[(size_t)b * 6 + lane] = v; }
            if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = lane == 0 ? acc[6] : acc[7];
        }
    } else {
        if (lane < 6 && io.ztNext) io.ztNext[(size_t)b * 6 + lane] = x[N * 6 + lane];          // MPC.feasibleStateInput :157-159
        if (lane < 2 && io.ztuNext) io.ztuNext[(size_t)b * 2 + lane] = u[(N - 1) * 2 + lane];
    }
    if (lane == 0) {
        io.status[b] = st_sh; io.iters[b] = it; flag_retry(io, st_sh);
        if (io.resid) { io.resid[(size_t)b * 3] = gap; io.resid[(size_t)b * 3 + 1] = rdn; io.resid[(size_t)b * 3 + 2] = ren; }
    }
}
#undef RT_SYNC
#undef RT_FOR

// launchers of the runtime kernel (the table of lmpc_variant.hip.h is filled by lmpc_variant_fill_rt there): one wave per QP on every route
static int lmpc_rt_launch(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
    hipLaunchKernelGGL((lmpc_solve_kernel_rt<false>), dim3(B), dim3(WAVE), (size_t)rt_layout(p.N, p.S).tot * sizeof(double), st, p, B, io); return 0; }
static int lmpc_rt_launch_retry(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io) {
    hipLaunchKernelGGL((lmpc_solve_kernel_rt<true>), dim3(B), dim3(WAVE), (size_t)rt_layout(p.N, p.S).tot * sizeof(double), st, p, B, io); return 0; }
static int lmpc_rt_launch_cd(hipStream_t st, const lmpc_dev_params &p, int B, const lmpc_solve_io &io, int) { return lmpc_rt_launch(st, p, B, io); }
