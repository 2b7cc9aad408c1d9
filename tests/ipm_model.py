"""tests/ipm_model.py -- NumPy model of the algorithm the HIP solve kernel runs (test aid, not product).

Structure-exploiting primal-dual interior point (Mehrotra predictor-corrector) for the LMPC /
LTV-MPC QP of the reference (PredictiveControllers.py:166-257, 340-362):
  * inequality rows handled by the barrier (slack t = b - F w is recomputed, so r_i = 0),
  * equalities (dynamics, terminal convex-hull, sum(lambda)=1) kept as constraints, true residuals
    r_d, r_e are re-evaluated every iteration (self-correcting, inexact Newton tolerated),
  * Newton system = block-banded KKT, solved by a Riccati recursion on the augmented state
    xi_k = (x_k, u_{k-1}) (input-rate cost), lane slacks eliminated analytically,
    terminal block (lambda, s_T) eliminated through a 7x7 square-root (MGS-QR) factor.
The kernels in racinglmpc_amd/csrc follow this file step by step:
  ipm_solve(exact_nu=True)   lmpc_solve_kernel (one wave per QP): multipliers of the dynamics rows from the adjoint recursion, no costate recursion (round 3)
  ipm_solve(exact_nu=False)  lmpc_solve_kernel_mw (two / four waves per QP): the same multipliers as a damped iterate
  ipm_solve_cd               lmpc_solve_kernel_cd (condensed form, experimental)
Round 3: barrier weights capped at th_max in the Newton matrix with consistent right-hand sides, row slacks carried as iterates (carry_t).
"""
import numpy as np


class StructQP:
    """Structured data of one QP: what K1/K2 hand to K3."""

    def __init__(self, par, A, B, C, x0, uOld, SS=None, Qsel=None):
        self.par, self.A, self.B, self.C, self.x0, self.uOld = par, A, B, C, np.asarray(x0, float), np.asarray(uOld, float).reshape(-1)
        self.N = par.N
        self.term = SS is not None
        self.SS, self.Qsel = SS, Qsel
        self.S = SS.shape[1] if self.term else 0


def mgs_QR(Mt, passes=2):
    """Modified Gram-Schmidt, run `passes` times (re-orthogonalisation): Mt = Q R."""
    V = Mt.copy(); n = V.shape[1]; Rtot = np.eye(n)
    for _ in range(passes):
        R = np.zeros((n, n))
        for i in range(n):
            R[i, i] = np.sqrt(V[:, i] @ V[:, i])
            V[:, i] = V[:, i] / R[i, i]
            for j in range(i + 1, n):
                R[i, j] = V[:, i] @ V[:, j]
                V[:, j] -= R[i, j] * V[:, i]
        Rtot = R @ Rtot
    return V, Rtot


def tri_inv_upper(R):
    n = R.shape[0]; X = np.zeros((n, n))
    for j in range(n):
        X[j, j] = 1.0 / R[j, j]
        for i in range(j - 1, -1, -1):
            X[i, j] = -(R[i, i + 1:j + 1] @ X[i + 1:j + 1, j]) / R[i, i]
    return X


TGT_FLOOR = 0.01
MW_LATE_FACTOR = 10            # (what lmpc_solve_kernel_mw does; None: this iterate's own gap, as the one-wave kernel)
TERM_LATE_GAP = 1e-8          # (LMPC_QX_GAP of the kernels; rounds 5: 1e-4)
SEP_STICKY = 0.5
SIG_EXP, FRAC0, FRAC_SIG, SEP_THR = 5, 0.99, 1e-3, 0.05      # (tools/knob_model.py: experiments on the step rules; the values the kernels use -- rounds 1-5: 3, 0.995, 1e-3, 0.1)
SEP_RULE = "kernel"       # "kernel": separate primal / dual steps after an iteration whose gap shrank by less than 10x; "noinc", "off": experiments
TERM_LATE = False          # set by ipm_solve per iteration: the iterate is in its final phase (gap below 1e-4)
TERM_FACTOR = "cholqr2_fo_late"        # (round 5: what the kernels do)   how kkt_factor factorises the terminal block: "mgs" (two passes of modified Gram-Schmidt on M'), "gram" (Cholesky of M M', the kernels' way), "gram_scaled"


class Factor:
    pass


def kkt_factor(qp, th_lane, th_u, th_s, th_l, reg_l=0.0):
    """Riccati backward matrix pass. th_*: barrier weights mu/t per inequality row class."""
    par = qp.par; N = qp.N; Fx, Fu = par.Fx, par.Fu; dR2 = 2 * par.dR; a = 2 * par.Qslack[0]
    Q2, Qf2, R2 = 2 * par.Q, 2 * par.Qf, 2 * par.R
    f = Factor()
    f.Ds = a + th_lane + th_s
    f.kap = th_lane * (a + th_s) / f.Ds
    Pi = np.zeros((8, 8)); Pi[:6, :6] = Qf2
    if qp.term:
        T = np.diag(2 * par.QterminalSlack)
        f.D = th_l + reg_l
        E = np.vstack([qp.SS, np.ones(qp.S)])
        Mt = np.vstack([(E / np.sqrt(f.D)).T, np.diag(np.concatenate([1 / np.sqrt(T), [0]]))[:6]])
        if TERM_FACTOR == "mgs":
            f.Qm, R = mgs_QR(Mt)
        else:
            # what the kernels do: Gram matrix W = M M' (on the matrix cores), Cholesky R'R = W; "gram_scaled": rows of M equilibrated first (unit diagonal of W)
            Mm = Mt.T                                             # 7 x (S + 6)
            dsc = 1.0 / np.sqrt(np.sum(Mm * Mm, axis=1)) if TERM_FACTOR == "gram_scaled" else np.ones(7)
            W = (Mm * dsc[:, None]) @ (Mm * dsc[:, None]).T
            R = np.linalg.cholesky(W).T / dsc[None, :]            # R'R = M M'
            if TERM_FACTOR.startswith("cholqr2") and (not TERM_FACTOR.endswith("_late") or TERM_LATE):
                # second pass (Cholesky-QR2): M1 = R^-T M has a Gram matrix close to the identity; its Cholesky factor corrects R
                M1 = tri_inv_upper(R).T @ Mm
                W2 = M1 @ M1.T
                if "fo" in TERM_FACTOR:                           # first-order factor of a Gram matrix next to the identity: I + E = (I + U)'(I + U) + O(E^2), U = triu(E, 1) + diag(E) / 2
                    E_ = W2 - np.eye(7); Rb = np.eye(7) + np.triu(E_, 1) + np.diag(np.diag(E_)) / 2
                    globals()["FO_EMAX"] = max(globals().get("FO_EMAX", 0.0), float(np.abs(E_).max()))
                else:
                    Rb = np.linalg.cholesky(W2).T
                R = Rb @ R
                f.Qm = M1.T @ tri_inv_upper(Rb)                   # Q kept explicitly, column by column of M: Q = (M' R1^-1) R2^-1, never M' (R2 R1)^-1
                if "yimpl" in TERM_FACTOR:                        # (experiment: y7 = R^-T (M c~), the explicit Q only where it multiplies z7)
                    f.Qm_y = Mt @ tri_inv_upper(R)
            else:
                f.Qm = Mt @ tri_inv_upper(R)
        f.Ri = tri_inv_upper(R); f.W7i = f.Ri @ f.Ri.T; f.E = E; f.T = T; f.sqD = np.sqrt(f.D)
        Pi[:6, :6] += f.W7i[:6, :6]
    f.Kx = np.zeros((N, 2, 6)); f.Ku = np.zeros((N, 2, 2)); f.Mi = np.zeros((N, 2, 2)); f.Hx = np.zeros((N, 6, 6))
    f.PiNext = np.zeros((N, 8, 8))
    for k in range(N - 1, -1, -1):
        A, B = qp.A[k], qp.B[k]
        f.PiNext[k] = Pi
        Hx = Q2 + (Fx.T * f.kap[k]) @ Fx; f.Hx[k] = Hx
        Hu = R2 + (Fu.T * th_u[k]) @ Fu
        Pxx, Pxu, Puu = Pi[:6, :6], Pi[:6, 6:], Pi[6:, 6:]
        T2 = Pxx @ B + Pxu
        Mxx = Hx + A.T @ (Pxx @ A); Mxu = A.T @ T2
        Muu = Hu + np.diag(dR2) + B.T @ T2 + Pxu.T @ B + Puu
        Mi = np.linalg.inv(Muu); f.Mi[k] = Mi
        f.Kx[k] = Mi @ Mxu.T; f.Ku[k] = Mi @ (-np.diag(dR2))
        Pn = np.zeros((8, 8))
        Pn[:6, :6] = Mxx - Mxu @ f.Kx[k]; Pn[:6, 6:] = -Mxu @ f.Ku[k]; Pn[6:, :6] = Pn[:6, 6:].T
        Pn[6:, 6:] = np.diag(dR2) + np.diag(dR2) @ f.Ku[k]
        Pi = Pn
    return f


def kkt_solve(qp, f, th_lane, th_s, gx, gu, gs, h_lane, h_u, h_s, gl, re_dyn, re_sum):
    """Solve the Newton system for one right-hand side.
       (P + F'ThF) dw + G'dnu = -(g) + F'h ,  G dw = -r_e   (dx_0 = 0, du_{-1} = 0)
    gx (N+1,6), gu (N,2), gs (N,2): gradient part; h_*: barrier rhs per row; gl (S): gradient + (-h) for lambda
    (already combined: gl = g_lambda + h_lambda since F_lambda = -I); re_dyn (N,6): residual of row k+1.
    Terminal slack s_T is eliminated (s_T = SS lam - x_N), so its gradient enters through the caller."""
    par = qp.par; N = qp.N; Fx, Fu = par.Fx, par.Fu; dR2 = 2 * par.dR
    e = -(gs + h_lane + h_s)
    eta = h_lane + th_lane * e / f.Ds
    pv = np.zeros(8); pv[:6] = gx[N]
    if qp.term:
        ct = np.concatenate([gl / f.sqD, np.zeros(6)])
        y7 = (f.Qm_y if hasattr(f, "Qm_y") else f.Qm).T @ ct
        d0 = np.concatenate([np.zeros(6), [-re_sum]])
        pv[:6] += (f.Ri @ (f.Ri.T @ d0 + y7))[:6]
    k0 = np.zeros((N, 2))
    for k in range(N - 1, -1, -1):
        A, B = qp.A[k], qp.B[k]; Pi = f.PiNext[k]
        c = -re_dyn[k]                                   # dx_{k+1} = A dx + B du + c
        z = Pi[:6, :6] @ c + pv[:6]
        zu = Pi[6:, :6] @ c + pv[6:]
        mx = gx[k] - Fx.T @ eta[k] + A.T @ z
        mu_ = gu[k] - Fu.T @ h_u[k] + B.T @ z + zu
        k0[k] = f.Mi[k] @ mu_
        pn = np.zeros(8); pn[:6] = mx - f.Kx[k].T @ mu_; pn[6:] = -f.Ku[k].T @ mu_
        pv = pn
    dx = np.zeros((N + 1, 6)); du = np.zeros((N, 2)); up = np.zeros(2)
    for k in range(N):
        du[k] = -f.Kx[k] @ dx[k] - f.Ku[k] @ up - k0[k]
        dx[k + 1] = qp.A[k] @ dx[k] + qp.B[k] @ du[k] - re_dyn[k]; up = du[k]
    fl = dx[:N] @ Fx.T
    ds = (th_lane * fl + e) / f.Ds
    dl = None
    if qp.term:
        d7 = np.concatenate([dx[N], [-re_sum]])
        v = -(ct - f.Qm @ y7) + f.Qm @ (f.Ri.T @ d7)
        dl = v[:qp.S] / f.sqD
    return dx, du, ds, dl


def ipm_solve(qp, ncorr=None, tol_gap=1e-11, tol_res=1e-9, maxit=40, reg_l=1e-6, verbose=False, th_max=1e11, carry_t=True, exact_nu=True, polish=None, pex=None, so_w=1.1, start=None, trace=None, degen_tol=None, acc_rule="kernel", snaps=None):
    """Returns dict(x,u,s,lam,sT, mu (ineq duals in reference row order), iters, gap, rd, re).
    exact_nu (round 3): the multipliers of the dynamics rows are not iterates of their own; every iteration takes them from the adjoint recursion
    nu_{k-1} = A_k' nu_k - w_k (w_k: gradient of the state rows' other terms), so the x rows of the dual residual vanish identically and the
    costate recursion for d nu is gone -- the same Newton direction in (u, s, lambda, mu) as the condensed form below."""
    par = qp.par; N = qp.N; S = qp.S; Fx, Fu = par.Fx, par.Fu; bx, bu = par.bx, par.bu
    dR2 = 2 * par.dR; a = 2 * par.Qslack[0]; c1 = par.Qslack[1]
    Q2, Qf2, R2 = 2 * par.Q, 2 * par.Qf, 2 * par.R; xRef = par.xRef
    A, B, C = qp.A, qp.B, qp.C
    T = np.diag(2 * par.QterminalSlack) if qp.term else None
    # ---- strictly interior start
    x = np.zeros((N + 1, 6)); u = np.zeros((N, 2)); x[0] = qp.x0
    start = start or {}
    if isinstance(start.get("u"), str) and start.get("u") == "uold":         # (experiment: the previous input held over the horizon, pulled inside the box)
        ub = np.array([bu[0], bu[2]]) * start.get("shrink", 0.9)
        u[:] = np.clip(qp.uOld, -ub, ub)
    elif start.get("u") is not None:                                         # (round 5, warm start: the shifted input sequence of the previous closed-loop step, pulled inside the box)
        ub = np.array([bu[0], bu[2]]) * start.get("shrink", 0.98)
        u[:] = np.clip(np.asarray(start["u"], float), -ub, ub)
    for k in range(N):
        x[k + 1] = A[k] @ x[k] + B[k] @ u[k] + C[k]
    viol = x[:N] @ Fx.T - bx
    s = np.where(viol > 0, viol + 1.0, 1.0 / c1 if c1 > 1.0 else 1.0)      # violated rows one unit inside, the others at 1 / c_s
    lam = np.ones(S) / S if qp.term else np.zeros(0)
    nu = np.zeros((N, 6)); eta_m = 0.0
    def slacks():
        fl = x[:N] @ Fx.T
        return bx - (fl - s), bu - u @ Fu.T, s.copy(), lam.copy()
    t_lane, t_u, t_s, t_l = slacks()
    mu0 = max(1.0, 0.01 * (np.max(np.abs(qp.Qsel)) if qp.term else 1.0)) * start.get("mu_scale", 1.0)
    if start.get("slack") == "tight":                                        # (warm start: lane slacks at their violation plus a margin instead of one unit inside)
        s = np.where(viol > 0, viol + start.get("s_margin", 0.05), start.get("s_margin", 0.05))
    if start.get("lam") is not None:
        lam = np.maximum(np.asarray(start["lam"], float), start.get("lam_floor", 1e-3)); lam = lam / lam.sum()
    m_lane, m_u, m_s, m_l = mu0 / t_lane, mu0 / t_u, mu0 / t_s, (mu0 / t_l if qp.term else np.zeros(0))
    if start.get("mu") is not None:
        # (warm start, dual part: the previous step's multipliers of the lane / input / slack rows, shifted by one stage, floored so that every complementarity
        #  product is at least mu0 -- and capped at cap * mu0 / t, so that a row that was active and is not any more does not start far off the central path)
        pm_lane, pm_u, pm_s = start["mu"]
        cap = start.get("mu_cap", 100.0)
        m_lane = np.clip(pm_lane, mu0 / t_lane, cap * mu0 / t_lane); m_u = np.clip(pm_u, mu0 / t_u, cap * mu0 / t_u); m_s = np.clip(pm_s, mu0 / t_s, cap * mu0 / t_s)
    mtot = 8 * N + S
    info = {}
    sep = False; gap_prev = None          # separate primal/dual steps only after an iteration with poor progress
    qscale = max(1.0, float(np.max(np.abs(qp.Qsel)))) if qp.term else 1.0
    pol_backup = None; pol_rej = 0; nfact = 0; act_pred = None; pol_again = False; pol_chain = 0

    def residuals(x, u, s, lam, nu, eta_m, m_lane, m_u, m_s, m_l):
        """True residuals of the KKT conditions at (x, u, s, lam | nu, eta_m, mu); with exact_nu the multipliers of the dynamics rows are
        recomputed in place from the adjoint recursion (the x rows then vanish identically)."""
        sT = qp.SS @ lam - x[N] if qp.term else None
        rx = np.zeros((N + 1, 6))
        if exact_nu:
            nu[N - 1] = -(Qf2 @ (x[N] - xRef) - (T * sT if qp.term else 0))
            for k in range(N - 1, 0, -1):
                nu[k - 1] = A[k].T @ nu[k] - (Q2 @ (x[k] - xRef) + Fx.T @ m_lane[k])
        else:
            for k in range(1, N):
                rx[k] = Q2 @ (x[k] - xRef) + Fx.T @ m_lane[k] + nu[k - 1] - A[k].T @ nu[k]
            rx[N] = Qf2 @ (x[N] - xRef) + nu[N - 1] - (T * sT if qp.term else 0)
        ru = np.zeros((N, 2))
        for k in range(N):
            upv = u[k - 1] if k > 0 else qp.uOld
            ru[k] = R2 @ u[k] + dR2 * (u[k] - upv) + Fu.T @ m_u[k] - B[k].T @ nu[k]
            if k < N - 1:
                ru[k] += dR2 * (u[k] - u[k + 1])
        rs = a * s + c1 - m_lane - m_s
        rl = (qp.Qsel - m_l + qp.SS.T @ (T * sT) + eta_m) if qp.term else np.zeros(0)
        re_dyn = np.array([x[k + 1] - A[k] @ x[k] - B[k] @ u[k] - C[k] for k in range(N)])
        re_sum = (lam.sum() - 1.0) if qp.term else 0.0
        rd = max(np.abs(rx[1:]).max(), np.abs(ru).max(), np.abs(rs).max(), np.abs(rl).max() if qp.term else 0)
        re = max(np.abs(re_dyn).max(), abs(re_sum))
        return rx, ru, rs, rl, re_dyn, re_sum, rd, re

    def costate_steps(dx, dl, dm_lane, dm_l, rx, rl):
        """Steps of the equality multipliers that go with a Newton direction (dx, dl, dmu): dnu by the backward recursion over the x rows, deta as the
        mean over the lambda rows."""
        dnu = np.zeros((N, 6)); dsT = None
        if qp.term:
            dsT = qp.SS @ dl - dx[N]; g = rx[N] + Qf2 @ dx[N] - T * dsT
        else:
            g = rx[N] + Qf2 @ dx[N]
        dnu[N - 1] = -g
        for k in range(N - 1, 0, -1):
            dnu[k - 1] = -(rx[k] + Q2 @ dx[k] + Fx.T @ dm_lane[k]) + A[k].T @ dnu[k]
        deta = np.mean(-rl + dm_l - qp.SS.T @ (T * dsT)) if qp.term else 0.0
        return dnu, deta
    for it in range(maxit):
        if not carry_t or it == 0:
            t_lane, t_u, t_s, t_l = slacks()       # (carry_t: the row slacks are iterates of their own, t += alpha dt, never a difference of O(1) numbers)
        gap = (t_lane.ravel() @ m_lane.ravel() + t_u.ravel() @ m_u.ravel() + t_s.ravel() @ m_s.ravel() + t_l @ m_l) / mtot
        gp_before, sep_before = gap_prev, sep
        if gap_prev is not None:
            sep = gap > SEP_THR * gap_prev
            if SEP_RULE == "noinc":                # (round 5) ... and not after an iteration that INCREASED the gap: separate steps that do, alternate with equal steps
                sep = sep and gap < gap_prev       # that repair it -- a two-cycle of up to 20 iterations on 1 in 3 400 closed-loop QPs (tools/capture_slow_qps.py)
            elif SEP_RULE == "off":
                sep = False
            elif SEP_RULE == "sticky":             # (round 6) separate steps are switched off for the rest of the solve by the first separate-step iteration that fails to
                if sep_before and gap > SEP_STICKY * gp_before:      # contract the gap by SEP_STICKY: what breaks the two-cycle (separate step shrinks the gap 3x, the next one
                    info["sep_dead"] = True                          # grows it 3x) before it starts
                sep = sep and not info.get("sep_dead", False)
            elif SEP_RULE == "guard":              # (round 6) ... and from then on equal steps INSIDE a wide neighbourhood of the central path (the retry kernels' safeguard: the step
                if sep_before and gap > SEP_STICKY * gp_before:      # is shortened until every complementarity product keeps 1 % of the mean)
                    info["sep_dead"] = True
                sep = sep and not info.get("sep_dead", False)
        gap_prev = gap
        # ---- residuals
        rx, ru, rs, rl, re_dyn, re_sum, rd, re = residuals(x, u, s, lam, nu, eta_m, m_lane, m_u, m_s, m_l)
        if verbose:
            print("it %2d gap %.2e rd %.2e re %.2e" % (it, gap, rd, re))
        info.update(iters=it, gap=gap, rd=rd, re=re)
        if trace is not None:
            trace.append([gap, rd, re, np.nan, np.nan, np.nan])          # (gap, r_d, r_e | sigma, alpha_p, alpha_d of the step taken from here)
        if snaps is not None:                                             # (tools/term_rule_model.py: the iterate every termination rule is evaluated on, and the scalars the kernels have at this point)
            snaps.append(dict(x=x.copy(), u=u.copy(), lam=lam.copy(), gap=gap, rd=rd, re=re, step_prev=info.get("step_prev", 0.0), lstep_prev=info.get("lstep_prev", 0.0), rd_prev=info.get("rd_prev", 0.0), gap_prev=gp_before))
        if pol_backup is not None:
            # the previous iteration was an active-set (polish) step: accept it only if the true residuals meet the ordinary tolerances and
            # the signs hold (slacks of the rows taken as inactive, multipliers of the rows taken as active); otherwise back to the iterate it started from
            tsn = (t_lane, t_u, t_s, t_l); msn = (m_lane, m_u, m_s, m_l)
            tmin = min([t.min() for t in tsn if t.size]); mmin = min([m.min() for m in msn if m.size])
            ok = np.isfinite(gap) and abs(gap) < tol_gap and rd < tol_res * qscale and re < tol_res and tmin > -polish.get("tol_t", 1e-9) and mmin > -polish.get("tol_m", 1e-9) * qscale
            info["polish"] = "ok" if ok else "rejected"
            if verbose:
                print("   polish %s: gap %.2e rd %.2e re %.2e tmin %.2e mmin %.2e" % (info["polish"], gap, rd, re, tmin, mmin))
            if ok:
                info["pol_ok"] = 1
                break
            if polish.get("pdas", 0) > pol_chain and np.isfinite(gap) and np.isfinite(rd):
                # primal-dual active-set step: keep the point, re-classify the rows by mu - t > 0 (a violated inactive row has mu = 0, t < 0: becomes
                # active; an active row with a negative multiplier has t = 0, mu < 0: becomes inactive) and solve again
                pol_chain += 1; pol_again = True
            else:
                pol_again = False
        if pol_backup is not None and not pol_again:
            (x, u, s, lam, nu, eta_m, t_lane, t_u, t_s, t_l, m_lane, m_u, m_s, m_l, gap_prev, sep) = pol_backup
            pol_backup = None; pol_rej += 1
            continue
        # (round 5) degen_tol: rows that are neither clearly active nor clearly inactive -- slack AND multiplier both above degen_tol (multiplier in units of the
        # cost scale) -- mark a QP without strict complementarity, where the distance to the optimum goes like sqrt(gap), not like gap: such a problem iterates on
        # (round 5) acc_rule = dict(ratio, step, floor): the gap test alone lets a QP without strict complementarity stop ~sqrt(gap) away from its optimum (measured:
        # 2.4e-6 in x_N at gap 2e-12).  Such problems are the ones whose gap contracts LINEARLY at the end; they iterate on until the gap is below `floor` (the
        # kernels: 0.1 tol_gap) -- or, in the variant that was modelled and not built, until the last step taken was short.  A superlinear last step (gap ratio
        # below `ratio`) ends the iteration as before.
        acc_ok = True
        if acc_rule == "kernel":
            # (round 6) step_bound_ok of the kernels, lmpc_kernels.hip.h: the a-posteriori bound of a contracting iteration, |z - z*| <= rho / (1 - rho) |last step| with the rate
            # measured on the last two (x, u) steps, below LMPC_ACC_TOL = 1e-7 -- one rule for every horizon (tools/term_rule_model.py)
            acc_rule = lambda q: q["step_prev"] < q["step_pp"] and q["step_prev"] ** 2 <= 1e-7 * (q["step_pp"] - q["step_prev"])
        elif acc_rule == "r5":
            acc_rule = dict(ratio=(1e-4 if N > 12 else 1e-3), step=0.0, floor=(0.03 if N > 12 else 0.1) * tol_gap, est=(1e-6 if N > 12 else None))   # round 5's LMPC_ACC_RATIO<N> / LMPC_ACC_FLOOR<N> / LMPC_ACC_EST
        if callable(acc_rule):                                            # (tools/term_rule_model.py: any rule on the scalars of this point)
            acc_ok = bool(acc_rule(dict(gap=gap, gap_prev=gp_before, rd=rd, re=re, rd_prev=info.get("rd_prev", 0.0), step_prev=info.get("step_prev", np.inf), step_pp=info.get("step_pp", np.inf), lstep_prev=info.get("lstep_prev", 0.0),
                                        base_ok=gap < tol_gap and rd < tol_res * qscale and re < tol_res, it=it)))
        elif acc_rule is not None and gp_before is not None:
            acc_ok = gap <= acc_rule["ratio"] * gp_before or info.get("step_prev", 0.0) <= acc_rule["step"] or gap < acc_rule["floor"]
            if acc_rule.get("est") is not None and info.get("rd_prev", 0.0) > 0.0:
                # (round 5) a-posteriori estimate of the distance to the optimum from wave-uniform scalars: the last Newton step was ~ H^-1 r_prev, so |H^-1| ~ step_prev / rd_prev
                # and the error left is ~ |H^-1| rd.  A flat QP (1 of ~1 000 sampled closed-loop QPs: error = 660 x dual residual) fails this and iterates once more.
                acc_ok = acc_ok and info.get("step_prev", 0.0) * rd <= acc_rule["est"] * info["rd_prev"]
        if acc_ok and gap < tol_gap and rd < tol_res * qscale and re < tol_res and (degen_tol is None or max(
                [np.minimum(t, m / qscale).max() for t, m in zip((t_lane, t_u, t_s, t_l), (m_lane, m_u, m_s, m_l)) if t.size]) < degen_tol):
            break
        ts = (t_lane, t_u, t_s, t_l); ms = (m_lane, m_u, m_s, m_l)
        slow_ok = polish is None or "slow" not in polish or (gp_before is not None and gap > polish["slow"] * gp_before)      # only behind an iteration that contracted the gap by less than 1 / slow
        if polish is not None and (pol_again or (slow_ok and pol_rej <= polish.get("retries", 0) and gap < polish["gap"] * (polish.get("retry_factor", 1e-2) ** pol_rej) and rd < polish.get("rd", np.inf) * qscale)):
            # active-set finisher (the reference's own polish=True, PredictiveControllers.py:275): rows with t < mu are taken as active
            # (weight th_pol, right-hand side t th_pol: the step drives their slack to zero), the others as inactive (weight 0, right-hand side mu:
            # the step drives their multiplier to zero); ONE factorisation + solve, full step
            thp = polish.get("th", 1e11)
            act = [t < m for t, m in zip(ts, ms)]
            if polish.get("lookahead") and act_pred is not None:
                act = act_pred
            ths_p = [np.where(a_, thp, 0.0) for a_ in act]
            hp = [np.where(a_, t * thp, m) for a_, t, m in zip(act, ts, ms)]
            f = kkt_factor(qp, ths_p[0], ths_p[1], ths_p[2], ths_p[3], polish.get("reg", reg_l))
            dx, du, ds, dl = kkt_solve(qp, f, ths_p[0], ths_p[2], rx, ru, rs, hp[0], hp[1], hp[2], rl + hp[3], re_dyn, re_sum)
            fl = dx[:N] @ Fx.T
            dt = (-(fl - ds), -(du @ Fu.T), ds, (dl if qp.term else np.zeros(0)))
            dm = [-h_ - th * d for h_, th, d in zip(hp, ths_p, dt)]
            if not pol_again:
                pol_backup = (x.copy(), u.copy(), s.copy(), lam.copy(), nu.copy(), eta_m, t_lane, t_u, t_s, t_l, m_lane, m_u, m_s, m_l, gp_before, sep_before); pol_chain = 0
            pol_again = False
            dnu = np.zeros((N, 6))
            if qp.term:
                dsT = qp.SS @ dl - dx[N]; g = rx[N] + Qf2 @ dx[N] - T * dsT
            else:
                g = rx[N] + Qf2 @ dx[N]
            dnu[N - 1] = -g
            for k in range(N - 1, 0, -1):
                dnu[k - 1] = -(rx[k] + Q2 @ dx[k] + Fx.T @ dm[0][k]) + A[k].T @ dnu[k]
            deta = np.mean(-rl + dm[3] - qp.SS.T @ (T * dsT)) if qp.term else 0.0
            x = x + dx; u = u + du; s = s + ds
            if qp.term:
                lam = lam + dl
            t_lane, t_u, t_s, t_l = [t + d for t, d in zip(ts, dt)]
            m_lane, m_u, m_s, m_l = [m + d for m, d in zip(ms, dm)]
            if not exact_nu:
                nu = nu + dnu
            eta_m += deta
            info["pol_it"] = it; nfact += 1
            if polish.get("debug"):
                names = ("lane", "u", "s", "lam")
                for nm, t0, m0, tn, mn, a_ in zip(names, ts, ms, (t_lane, t_u, t_s, t_l), (m_lane, m_u, m_s, m_l), act):
                    for idx in np.argwhere((tn.ravel() < -1e-9) | (mn.ravel() < -1e-9)).ravel():
                        print("   row %s[%d] act=%d  t %.3e mu %.3e -> t %.3e mu %.3e" % (nm, idx, a_.ravel()[idx], t0.ravel()[idx], m0.ravel()[idx], tn.ravel()[idx], mn.ravel()[idx]))
            continue
        # Barrier weights theta = mu / t are capped at th_max in the Newton matrix: 1 / theta >= 1 / th_max is a dual regularisation of the row
        # (F dw + (1 / theta_c) dmu = -r_c / mu), so the right-hand side uses the same effective reciprocal rt = 1 / max(t, mu / th_max) as the
        # matrix and the fixed point does not move.  Uncapped, an active lane row (t ~ 1e-14, mu ~ 10) puts 1e15 into a stage Hessian whose
        # other entries are O(1): the Riccati recursion then loses the regular part of the cost-to-go to rounding (main.py's fast laps).
        cap = (lambda t, m: np.maximum(t, m / th_max)) if th_max is not None else (lambda t, m: t)
        rts = [1.0 / cap(t, m) if t.size else t for t, m in zip(ts, ms)]
        th_lane, th_u, th_s, th_l = ths = [m * rt for m, rt in zip(ms, rts)]
        # (the multi-wave kernels decide while the helper waves are still summing this iterate's gap: they look at the PREVIOUS iterate's, MW_LATE_FACTOR times the gate)
        globals()["TERM_LATE"] = gap < TERM_LATE_GAP if (exact_nu or MW_LATE_FACTOR is None) else (gp_before is not None and gp_before < MW_LATE_FACTOR * TERM_LATE_GAP)
        f = kkt_factor(qp, th_lane, th_u, th_s, th_l, reg_l); nfact += 1
        # terminal slack eliminated: its Hessian T enters W7 through T^-1 (kept in factor)
        def solve(h_lane, h_u, h_s, h_l):
            return kkt_solve(qp, f, th_lane, th_s, rx, ru, rs, h_lane, h_u, h_s, rl + h_l, re_dyn, re_sum)
        # predictor: h = mu
        hp = [t * m * rt for t, m, rt in zip(ts, ms, rts)]                # (= mu wherever the weight is not capped)
        dxa, dua, dsa, dla = solve(*hp)
        def ineq_steps(dx, du, ds, dl):
            fl = dx[:N] @ Fx.T
            return -(fl - ds), -(du @ Fu.T), ds, (dl if qp.term else np.zeros(0))
        dt = ineq_steps(dxa, dua, dsa, dla)
        dma = [-h_ - th * d for h_, th, d in zip(hp, ths, dt)]
        if pex is not None and gap < pex.get("gap", 1e-3):
            # predictor-extrapolated termination: the full affine-scaling step lands on the tangent's estimate of the optimum (the stationarity rows
            # are linear, so its dual residual is the solve's own error; its complementarity products are dt dmu, second order).  If that point
            # meets every tolerance -- true residuals re-evaluated, slacks and multipliers not below -tol -- it is returned.
            xc, uc, sc, lc = x + dxa, u + dua, s + dsa, (lam + dla if qp.term else lam)
            tcs = [t + d for t, d in zip(ts, dt)]; mcs = [m + d for m, d in zip(ms, dma)]
            dnu_a, deta_a = costate_steps(dxa, dla, dma[0], dma[3], rx, rl)
            nuc = nu + dnu_a; etac = eta_m + deta_a
            rdc, rec = residuals(xc, uc, sc, lc, nuc, etac, *mcs)[6:]
            gapc = sum(np.abs(t * m).sum() for t, m in zip(tcs, mcs)) / mtot
            tmin = min(t.min() for t in tcs if t.size); mmin = min(m.min() for m in mcs if m.size)
            if verbose:
                print("   pex: gap %.2e rd %.2e re %.2e tmin %.2e mmin %.2e" % (gapc, rdc, rec, tmin, mmin))
            if gapc < tol_gap and rdc < tol_res * qscale and rec < tol_res and tmin > -pex.get("tol_t", 1e-9) and mmin > -pex.get("tol_m", 1e-9) * qscale:
                x, u, s, lam, nu, eta_m = xc, uc, sc, lc, nuc, etac
                m_lane, m_u, m_s, m_l = mcs
                info.update(gap=gapc, rd=rdc, re=rec, pex=1)
                break
        def maxstep(vs, dvs):
            al = np.inf
            for v, dv in zip(vs, dvs):
                v = v.ravel(); dv = dv.ravel(); neg = dv < 0
                if neg.any():
                    al = min(al, np.min(-v[neg] / dv[neg]))
            return al
        aap = min(1.0, maxstep(ts, dt)); aad = min(1.0, maxstep(ms, dma))
        if not sep:
            aap = aad = min(aap, aad)
        gap_aff = sum(((t + aap * d).ravel() @ (m + aad * dm).ravel()) for t, d, m, dm in zip(ts, dt, ms, dma)) / mtot
        sig = (gap_aff / gap) ** SIG_EXP
        tgt = max(sig * gap, TGT_FLOOR * tol_gap)          # keep the complementarity products off the rounding floor
        rc = [t * m - tgt + so_w * d * dm for t, m, d, dm in zip(ts, ms, dt, dma)]      # (so_w = 0: no second-order term -- affine + centring only, what a two-right-hand-side single sweep could deliver)
        hs = [r * rt for r, rt in zip(rc, rts)]
        dx, du, ds, dl = solve(*hs)
        dt = ineq_steps(dx, du, ds, dl)
        dm = [-h_ - th * d for h_, th, d in zip(hs, ths, dt)]
        if ncorr is not None and gap < ncorr.get("gap", 1e-4):
            # (round 5 experiment, negative -- profiles/r5_ncorr_model.txt) iterated corrector: the second-order term of the right-hand side taken from the combined
            # direction just computed instead of the affine one, same factorisation, one more pair of sweeps per repetition
            for _ in range(ncorr.get("n", 1)):
                rc = [t * m - tgt + d * dm_ for t, m, d, dm_ in zip(ts, ms, dt, dm)]
                hs = [r * rt for r, rt in zip(rc, rts)]
                dx, du, ds, dl = solve(*hs)
                dt = ineq_steps(dx, du, ds, dl)
                dm = [-h_ - th * d for h_, th, d in zip(hs, ths, dt)]
                info["ncorr"] = info.get("ncorr", 0) + 1
        frac = max(FRAC0, 1.0 - 10.0 * gap) if sig < FRAC_SIG else FRAC0      # longer steps in the final phase only (step_fraction in the kernel)
        al = min(1.0, frac * maxstep(ts, dt)); ald = min(1.0, frac * maxstep(ms, dm))
        if not sep:
            al = ald = min(al, ald)
        if SEP_RULE == "guard" and info.get("sep_dead", False):
            for _ in range(8):
                pr = np.concatenate([((t + al * d) * (m + ald * dm_)).ravel() for t, d, m, dm_ in zip(ts, dt, ms, dm)])
                if pr.min() >= 1e-2 * pr.sum() / mtot:
                    break
                al *= 0.7; ald *= 0.7
        # costates (delta): dnu_N from terminal, then backwards
        dnu = np.zeros((N, 6))
        # dnu_N (row x_N): stationarity of the Newton system in x_N
        if qp.term:
            dsT = qp.SS @ dl - dx[N]
            g = rx[N] + Qf2 @ dx[N] - T * dsT
        else:
            g = rx[N] + Qf2 @ dx[N]
        dnu[N - 1] = -g
        for k in range(N - 1, 0, -1):
            # row x_k: Hx dx_k + Fx'(dmu_lane) ... use eliminated form: (Q2) dx + Fx' dm_lane + dnu_{k-1} - A_k' dnu_k = -rx_k
            dnu[k - 1] = -(rx[k] + Q2 @ dx[k] + Fx.T @ dm[0][k]) + A[k].T @ dnu[k]
        deta = 0.0
        if qp.term:
            # row lambda_i: -dm_l + SS'(T dsT) + deta = -rl  -> average over rows for robustness
            deta = np.mean(-rl + dm[3] - qp.SS.T @ (T * dsT))
        act_pred = [(t + d) < (m + d2) for t, d, m, d2 in zip(ts, dt, ms, dm)]      # full-step (Newton target) classification of the rows
        if trace is not None:
            trace[-1][3:] = [sig, al, ald]
        info["step_pp"] = info.get("step_prev", np.inf)
        info["step_prev"] = al * max(np.abs(dx).max(), np.abs(du).max())
        info["rd_prev"] = rd
        info["lstep_prev"] = al * np.abs(dl).max() if qp.term else 0.0
        x += al * dx; u += al * du; s += al * ds
        if carry_t:
            t_lane, t_u, t_s, t_l = [t + al * d for t, d in zip(ts, dt)]
        if qp.term:
            lam += al * dl
        m_lane, m_u, m_s, m_l = [m + ald * d for m, d in zip(ms, dm)]
        if not exact_nu:
            nu += ald * dnu
        eta_m += ald * deta
    out = dict(x=x, u=u, s=s, lam=lam, sT=(qp.SS @ lam - x[N]) if qp.term else None,
               mu=np.concatenate([m_lane.ravel(), m_u.ravel(), m_s.ravel(), m_l]), nu=nu, eta=eta_m)
    out.update(info); out["nfact"] = nfact
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# Condensed form (round 3): the states are eliminated, x = X(u) by roll-out, so the only equality row left is sum(lambda) = 1 and the
# Newton matrix is DENSE in du (2N x 2N: 24 x 24 at N = 12) -- Cholesky instead of the Riccati recursion, dense mat-vecs instead of the
# backward / forward sweeps.  Same interior-point rules as ipm_solve (start point, capped barrier weights, Mehrotra predictor-corrector,
# step-length rules, termination).  lmpc_solve_kernel_cd follows this function.
# ------------------------------------------------------------------------------------------------------------------------------------
def condense(qp):
    """Per-QP constants: Su_k = d x_k / d u (6 x 2N, k = 0..N), V_k = Fx Su_k (2 x 2N), G = Su_N, and the constant part of the reduced Hessian
    Hc0 = blockdiag(R2) + input-rate coupling + sum_k Su_k' Q2 Su_k + G' Qf2 G."""
    par = qp.par; N = qp.N; nv = 2 * N
    Su = np.zeros((N + 1, 6, nv))
    for k in range(N):
        Su[k + 1] = qp.A[k] @ Su[k]
        Su[k + 1][:, 2 * k:2 * k + 2] += qp.B[k]
    V = np.einsum("jc,kcv->kjv", par.Fx, Su[:N])                     # row (k, j) of the lane constraints as a function of du
    G = Su[N]
    R2, dR2, Q2, Qf2 = 2 * par.R, 2 * par.dR, 2 * par.Q, 2 * par.Qf
    H0 = np.zeros((nv, nv))
    for k in range(N):
        H0[2 * k:2 * k + 2, 2 * k:2 * k + 2] += R2 + np.diag(dR2) * (2 if k < N - 1 else 1)
        if k < N - 1:
            H0[2 * k:2 * k + 2, 2 * k + 2:2 * k + 4] -= np.diag(dR2); H0[2 * k + 2:2 * k + 4, 2 * k:2 * k + 2] -= np.diag(dR2)
    for k in range(1, N):
        H0 += Su[k].T @ Q2 @ Su[k]
    H0 += G.T @ Qf2 @ G
    return V, G, H0


def ipm_solve_cd(qp, tol_gap=1e-11, tol_res=1e-9, maxit=40, reg_l=1e-6, verbose=False, th_max=1e11):
    par = qp.par; N = qp.N; S = qp.S; Fx, Fu = par.Fx, par.Fu; bx, bu = par.bx, par.bu; nv = 2 * N
    dR2 = 2 * par.dR; a = 2 * par.Qslack[0]; c1 = par.Qslack[1]
    Q2, Qf2, R2 = 2 * par.Q, 2 * par.Qf, 2 * par.R; xRef = par.xRef
    A, B, C = qp.A, qp.B, qp.C
    T = np.diag(2 * par.QterminalSlack) if qp.term else None
    V, G, H0 = condense(qp)

    def rollout(u):
        x = np.zeros((N + 1, 6)); x[0] = qp.x0
        for k in range(N):
            x[k + 1] = A[k] @ x[k] + B[k] @ u[k] + C[k]
        return x
    u = np.zeros((N, 2)); x = rollout(u)
    viol = x[:N] @ Fx.T - bx
    s = np.where(viol > 0, viol + 1.0, 1.0 / c1 if c1 > 1.0 else 1.0)
    lam = np.ones(S) / S if qp.term else np.zeros(0)
    eta_m = 0.0
    t_lane, t_u, t_s, t_l = bx - (x[:N] @ Fx.T - s), bu - u @ Fu.T, s.copy(), lam.copy()
    mu0 = max(1.0, 0.01 * (np.max(np.abs(qp.Qsel)) if qp.term else 1.0))
    m_lane, m_u, m_s, m_l = mu0 / t_lane, mu0 / t_u, mu0 / t_s, (mu0 / t_l if qp.term else np.zeros(0))
    if start.get("mu") is not None:
        # (warm start, dual part: the previous step's multipliers of the lane / input / slack rows, shifted by one stage, floored so that every complementarity
        #  product is at least mu0 -- and capped at cap * mu0 / t, so that a row that was active and is not any more does not start far off the central path)
        pm_lane, pm_u, pm_s = start["mu"]
        cap = start.get("mu_cap", 100.0)
        m_lane = np.clip(pm_lane, mu0 / t_lane, cap * mu0 / t_lane); m_u = np.clip(pm_u, mu0 / t_u, cap * mu0 / t_u); m_s = np.clip(pm_s, mu0 / t_s, cap * mu0 / t_s)
    mtot = 8 * N + S
    sep = False; gap_prev = None
    qscale = max(1.0, float(np.max(np.abs(qp.Qsel)))) if qp.term else 1.0
    info = {}
    E7 = np.vstack([qp.SS, np.ones(S)]) if qp.term else None
    for it in range(maxit):
        ts = (t_lane, t_u, t_s, t_l); ms = (m_lane, m_u, m_s, m_l)
        gap = sum(t.ravel() @ m.ravel() for t, m in zip(ts, ms)) / mtot
        if gap_prev is not None:
            sep = gap > SEP_THR * gap_prev
        gap_prev = gap
        # ---- residuals: adjoint recursion p_k = w_k + A_k' p_{k+1} gives the state part of the u rows
        sT = qp.SS @ lam - x[N] if qp.term else None
        p = Qf2 @ (x[N] - xRef) - (T * sT if qp.term else 0)
        ru = np.zeros((N, 2))
        for k in range(N - 1, -1, -1):
            upv = u[k - 1] if k > 0 else qp.uOld
            g = R2 @ u[k] + dR2 * (u[k] - upv) + Fu.T @ m_u[k]
            if k < N - 1:
                g = g + dR2 * (u[k] - u[k + 1])
            ru[k] = g + B[k].T @ p
            if k > 0:
                p = Q2 @ (x[k] - xRef) + Fx.T @ m_lane[k] + A[k].T @ p
        rs = a * s + c1 - m_lane - m_s
        rl = (qp.Qsel - m_l + qp.SS.T @ (T * sT) + eta_m) if qp.term else np.zeros(0)
        re_sum = (lam.sum() - 1.0) if qp.term else 0.0
        rd = max(np.abs(ru).max(), np.abs(rs).max(), np.abs(rl).max() if qp.term else 0)
        re = abs(re_sum)
        if verbose:
            print("it %2d gap %.2e rd %.2e re %.2e" % (it, gap, rd, re))
        info = dict(iters=it, gap=gap, rd=rd, re=re)
        if gap < tol_gap and rd < tol_res * qscale and re < tol_res:
            break
        cap = (lambda t, m: np.maximum(t, m / th_max))
        rts = [1.0 / cap(t, m) if t.size else t for t, m in zip(ts, ms)]
        th_lane, th_u, th_s, th_l = ths = [m * rt for m, rt in zip(ms, rts)]
        # ---- factorisation: terminal 7 x 7 block as before, then the dense reduced Hessian
        Ds = a + th_lane + th_s; kap = th_lane * (a + th_s) / Ds
        Hc = H0.copy()
        for k in range(N):
            Hc[2 * k:2 * k + 2, 2 * k:2 * k + 2] += (Fu.T * th_u[k]) @ Fu
            Hc += (V[k].T * kap[k]) @ V[k]
        if qp.term:
            D = th_l + reg_l; sqD = np.sqrt(D)
            Mt = np.vstack([(E7 / sqD).T, np.diag(np.concatenate([1 / np.sqrt(T), [0]]))[:6]])
            Qm, R = mgs_QR(Mt); Ri = tri_inv_upper(R)
            PiT = (Ri @ Ri.T)[:6, :6]
            Hc += G.T @ PiT @ G
        L = np.linalg.cholesky(Hc)

        def solve(h_lane, h_u, h_s, h_l):
            e = -(rs + h_lane + h_s)
            eta = h_lane + th_lane * e / Ds
            g = ru.ravel() - (h_u @ Fu).ravel() - np.einsum("kjv,kj->v", V, eta)
            if qp.term:
                ct = np.concatenate([(rl + h_l) / sqD, np.zeros(6)])
                y7 = Qm.T @ ct
                d0 = np.concatenate([np.zeros(6), [-re_sum]])
                g = g + G.T @ (Ri @ (Ri.T @ d0 + y7))[:6]
            du = -np.linalg.solve(L.T, np.linalg.solve(L, g))
            fl = np.einsum("kjv,v->kj", V, du)
            ds = (th_lane * fl + e) / Ds
            dl = None; dxN = G @ du
            if qp.term:
                d7 = np.concatenate([dxN, [-re_sum]])
                v = -(ct - Qm @ y7) + Qm @ (Ri.T @ d7)
                dl = v[:S] / sqD
            return du.reshape(N, 2), fl, ds, dl, dxN

        hp = [t * m * rt for t, m, rt in zip(ts, ms, rts)]
        dua, fla, dsa, dla, _ = solve(*hp)
        steps = lambda du, fl, ds, dl: (-(fl - ds), -(du @ Fu.T), ds, (dl if qp.term else np.zeros(0)))
        dt = steps(dua, fla, dsa, dla)
        dma = [-h_ - th * d for h_, th, d in zip(hp, ths, dt)]

        def maxstep(vs, dvs):
            al = np.inf
            for v, dv in zip(vs, dvs):
                v = v.ravel(); dv = dv.ravel(); neg = dv < 0
                if neg.any():
                    al = min(al, np.min(-v[neg] / dv[neg]))
            return al
        aap = min(1.0, maxstep(ts, dt)); aad = min(1.0, maxstep(ms, dma))
        if not sep:
            aap = aad = min(aap, aad)
        gap_aff = sum(((t + aap * d).ravel() @ (m + aad * dm).ravel()) for t, d, m, dm in zip(ts, dt, ms, dma)) / mtot
        sig = (gap_aff / gap) ** SIG_EXP
        tgt = max(sig * gap, 0.01 * tol_gap)
        rc = [t * m - tgt + 1.1 * d * dm for t, m, d, dm in zip(ts, ms, dt, dma)]          # (LMPC_SO_W)
        hs = [r * rt for r, rt in zip(rc, rts)]
        du, fl, ds, dl, dxN = solve(*hs)
        dt = steps(du, fl, ds, dl)
        dm = [-h_ - th * d for h_, th, d in zip(hs, ths, dt)]
        frac = max(FRAC0, 1.0 - 10.0 * gap) if sig < FRAC_SIG else FRAC0
        al = min(1.0, frac * maxstep(ts, dt)); ald = min(1.0, frac * maxstep(ms, dm))
        if not sep:
            al = ald = min(al, ald)
        deta = 0.0
        if qp.term:
            dsT = qp.SS @ dl - dxN
            deta = np.mean(-rl + dm[3] - qp.SS.T @ (T * dsT))
        u = u + al * du; s = s + al * ds
        if qp.term:
            lam = lam + al * dl
        x = rollout(u)
        t_lane, t_u, t_s, t_l = [t + al * d for t, d in zip(ts, dt)]
        m_lane, m_u, m_s, m_l = [m + ald * d for m, d in zip(ms, dm)]
        eta_m += ald * deta
    out = dict(x=x, u=u, s=s, lam=lam, sT=(qp.SS @ lam - x[N]) if qp.term else None,
               mu=np.concatenate([m_lane.ravel(), m_u.ravel(), m_s.ravel(), m_l]), eta=eta_m)
    out.update(info)
    return out
