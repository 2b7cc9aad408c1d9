"""Vectorised, solver-independent KKT certificate for a whole batch of LMPC / MPC QPs (test infrastructure).

The QP is the reference's (PredictiveControllers.py:166-257, 340-362):
    z = [x_0..x_N | u_0..u_{N-1} | s (2N) | lambda (S) | s_T (6)],   min 1/2 z'Pz + q'z,   F z <= b,   G z = E x0 + L.
Given a primal point and the multipliers `mu` of the inequality rows (reference row order), the equality multipliers follow
from the stationarity rows of x_N .. x_1 and s_T by a backward recursion over the banded structure (no linear solve, nothing
from the solver under test); the REMAINING stationarity rows (u, s, lambda) are the certificate, together with primal
feasibility, dual feasibility and complementarity.  All arrays carry a leading batch axis.

`par` needs: Q, R, Qf, dR, Qslack, Fx, bx, Fu, bu, xRef and (S > 0) QterminalSlack -- e.g. oracle.QPParams, used here only as
a parameter record.
"""
import numpy as np


def certificate(par, A, Bm, C, x0, uOld, xPred, uPred, slack, mu, ssSel=None, qSel=None, lambd=None, sTerm=None):
    """Returns a dict of per-problem arrays (B,): stat (stationarity), prim_eq, prim_ineq, dual (negative part of mu),
    comp (max |mu_i t_i|), scale (cost scale the tolerances are relative to) and `worst` = max of the five."""
    x = np.asarray(xPred, float); u = np.asarray(uPred, float); s = np.asarray(slack, float); mu = np.asarray(mu, float)
    Bn, N1, _ = x.shape
    N = N1 - 1
    S = 0 if ssSel is None else np.asarray(ssSel).shape[1]
    Q2 = 2.0 * np.asarray(par.Q, float); Qf2 = 2.0 * np.asarray(par.Qf, float); R2 = 2.0 * np.asarray(par.R, float)
    dR2 = 2.0 * np.asarray(par.dR, float).reshape(2)
    a_s = 2.0 * float(np.asarray(par.Qslack).reshape(-1)[0]); c_s = float(np.asarray(par.Qslack).reshape(-1)[1])
    Fx = np.asarray(par.Fx, float).reshape(2, 6); bx = np.asarray(par.bx, float).reshape(2)
    Fu = np.asarray(par.Fu, float).reshape(4, 2); bu = np.asarray(par.bu, float).reshape(4)
    xRef = np.asarray(par.xRef, float).reshape(6)
    A = np.asarray(A, float).reshape(Bn, N, 6, 6); Bm = np.asarray(Bm, float).reshape(Bn, N, 6, 2); C = np.asarray(C, float).reshape(Bn, N, 6)
    mu_x = mu[:, 0:2 * N].reshape(Bn, N, 2); mu_u = mu[:, 2 * N:6 * N].reshape(Bn, N, 4); mu_s = mu[:, 6 * N:8 * N]
    mu_l = mu[:, 8 * N:8 * N + S]

    # ---- primal feasibility -------------------------------------------------------------------------------------------
    dyn = np.einsum("bkij,bkj->bki", A, x[:, :-1]) + np.einsum("bkij,bkj->bki", Bm, u) + C - x[:, 1:]
    prim_eq = np.maximum(np.abs(dyn).reshape(Bn, -1).max(1), np.abs(x[:, 0] - np.asarray(x0, float)).max(1))
    t_x = bx[None, None, :] - (np.einsum("jc,bkc->bkj", Fx, x[:, :N]) - s.reshape(Bn, N, 2))         # rows [0, 2N)
    t_u = bu[None, None, :] - np.einsum("jc,bkc->bkj", Fu, u)                                          # rows [2N, 6N)
    t_s = s.copy()                                                                                     # rows [6N, 8N): -s <= 0
    tt = [t_x.reshape(Bn, -1), t_u.reshape(Bn, -1), t_s]
    if S:
        lam = np.asarray(lambd, float); sT = np.asarray(sTerm, float); SS = np.asarray(ssSel, float)     # SS: (B, S, 6)
        T2 = 2.0 * np.diag(np.asarray(par.QterminalSlack, float))
        term = x[:, N] - np.einsum("bcj,bc->bj", SS, lam) + sT
        prim_eq = np.maximum(prim_eq, np.maximum(np.abs(term).max(1), np.abs(lam.sum(1) - 1.0)))
        tt.append(lam)
    t = np.concatenate(tt, axis=1)
    prim_ineq = np.maximum(-t, 0.0).max(1)
    dual = np.maximum(-mu, 0.0).max(1)
    comp = np.abs(mu * t).max(1)

    # ---- equality multipliers from the x_k / s_T stationarity rows (backward over the band) ---------------------------------
    nu = np.zeros((Bn, N + 1, 6))                       # nu[:, k] belongs to the row that defines x_k (k >= 1)
    nuT = np.zeros((Bn, 6))
    if S:
        nuT = -T2[None, :] * sT
    nu[:, N] = -np.einsum("ij,bj->bi", Qf2, x[:, N] - xRef) - nuT
    for k in range(N - 1, 0, -1):
        nu[:, k] = (np.einsum("bji,bj->bi", A[:, k], nu[:, k + 1]) - np.einsum("ij,bj->bi", Q2, x[:, k] - xRef)
                    - np.einsum("jc,bj->bc", Fx, mu_x[:, k]))
    # ---- certificate rows: u, s, lambda -------------------------------------------------------------------------------------
    uprev = np.concatenate([np.asarray(uOld, float)[:, None, :], u[:, :-1]], axis=1)
    gu = np.einsum("ij,bkj->bki", R2, u) + dR2 * (u - uprev)
    gu[:, :-1] += dR2 * (u[:, :-1] - u[:, 1:])
    ru = gu + np.einsum("jc,bkj->bkc", Fu, mu_u) - np.einsum("bkji,bkj->bki", Bm, nu[:, 1:])
    rs = a_s * s + c_s - mu_x.reshape(Bn, -1) - mu_s
    stat = np.maximum(np.abs(ru).reshape(Bn, -1).max(1), np.abs(rs).max(1))
    scale = np.ones(Bn)
    if S:
        qS = np.asarray(qSel, float)
        rl0 = qS - mu_l - np.einsum("bcj,bj->bc", SS, nuT)
        eta = -rl0.mean(1)                               # multiplier of sum(lambda) = 1: the least-squares value
        stat = np.maximum(stat, np.abs(rl0 + eta[:, None]).max(1))
        scale = np.maximum(1.0, np.abs(qS).max(1))
    out = dict(stat=stat, prim_eq=prim_eq, prim_ineq=prim_ineq, dual=dual, comp=comp, scale=scale)
    out["worst"] = np.max(np.stack([stat / scale, prim_eq, prim_ineq, dual / scale, comp / scale]), axis=0)
    return out


def assert_certified(cert, tol, what=""):
    w = cert["worst"]
    bad = np.nonzero(~(w <= tol))[0]
    assert bad.size == 0, "%s: %d of %d problems exceed the KKT tolerance %.1e (worst %.3e at %d: %s)" % (
        what, bad.size, w.size, tol, np.nanmax(w), int(np.nanargmax(w)), {k: float(v[int(np.nanargmax(w))]) for k, v in cert.items()})
    return float(w.max())
