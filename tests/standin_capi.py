"""tests/standin_capi.py -- a CPU stand-in for racinglmpc_amd._capi, backed by the oracle (test infrastructure, never product).

Purpose: run the reference's UNCHANGED main.py (and plot.py) against the drop-in Python layer in a container that has the reference but no GPU
(tests/test_reference_main_seam.py, tools/run_reference_main.py).  Installed as sys.modules["racinglmpc_amd._capi"] BEFORE the drop-in modules are
imported, it offers the names and call signatures those modules use (config_from, Context.step_batch / regress_batch / qp_solve_batch / select_batch /
assemble_batch / ss_* / model_add_trajectory, lti_regression, the ST_* bits) and records every call; the arithmetic behind them is the oracle's
restatement (oracle/lmpc_oracle.py), each QP solved to its certified optimum as the GPU path does.  What this proves is the SEAM: every module
name, class, method and attribute main.py, SysModel.Simulator and plot.py touch exists in the drop-in layer with the reference's shapes.  The
kernels are checked against the same oracle elsewhere (tests/test_gpu_*.py)."""
import types

import numpy as np

from oracle import lmpc_oracle as orc

ST_MAXITER, ST_REG_SINGULAR, ST_NO_SEGMENT, ST_WINDOW, ST_NUMERIC, ST_NOT_INTERIOR, ST_INEXACT, ST_INFEASIBLE = 1, 2, 4, 8, 16, 32, 64, 128
CALLS = []                         # (name, detail) in call order


class LmpcError(RuntimeError):
    pass


def _rec(name, detail=None):
    CALLS.append((name, detail))


def lti_regression(x, u, lamb, device=0):
    _rec("lti_regression", np.asarray(x).shape)
    A, B, E = orc.lti_regression(np.asarray(x, float), np.asarray(u, float), lamb)
    return A, B, E, 0


def config_from(N, Q, R, Qf, dR, Qslack, Fx, bx, Fu, bu, xRef, QterminalSlack=None, numSS_Points=0, numSS_it=0, trToUse=0,
                track=None, trackLength=0.0, max_batch=256, max_laps=64, max_lap_len=2048, device=0, slacks=True, **solver):
    _rec("config_from", dict(N=N, numSS_it=numSS_it, numSS_Points=numSS_Points, trToUse=trToUse))
    cfg = types.SimpleNamespace(N=int(N), numSS_it=int(numSS_it), numSS_points=int(numSS_Points) if numSS_it else 0, trToUse=int(trToUse),
                                par=orc.QPParams(int(N), Q, R, Qf, dR, Qslack, Fx, bx, Fu, bu, xRef, QterminalSlack, int(numSS_Points), int(numSS_it), slacks),
                                track=None if track is None else np.asarray(track, float), trackLength=float(trackLength), max_batch=max_batch,
                                maxNumPoint=7, h=5.0, lamb=0.0, dt=0.1, scaling=[0.1, 1.0, 1.0, 1.0, 1.0], slacks=1 if slacks else 0)
    return cfg


class Context:
    def __init__(self, cfg):
        _rec("Context", dict(N=cfg.N, numSS_it=cfg.numSS_it))
        self.cfg, self.N, self.par = cfg, cfg.N, cfg.par
        self.S = cfg.numSS_points if cfg.numSS_it > 0 else 0
        self.M = 8 * self.N + self.S
        self.xStored, self.uStored, self.lapTimeM = [], [], []
        self.SS, self.uSS, self.Qfun, self.LapTime = [], [], [], []
        self.sel = None

    def close(self):
        pass

    # ---- stores
    def model_add_trajectory(self, x, u):
        _rec("model_add_trajectory", np.asarray(x).shape)
        orc.model_sorted_insert(self.xStored, self.uStored, self.lapTimeM, np.array(x, float), np.array(u, float))

    def ss_add_trajectory(self, x, u):
        _rec("ss_add_trajectory", np.asarray(x).shape)
        x = np.array(x, float)
        self.SS.append(x); self.uSS.append(np.array(u, float)); self.LapTime.append(x.shape[0])
        self.Qfun.append(orc.compute_cost(x, self.cfg.trackLength))

    def ss_get_qfun(self, lap):
        return self.Qfun[lap].copy()

    def ss_add_point(self, x, u):
        _rec("ss_add_point")
        TL = self.cfg.trackLength
        self.SS[-1] = np.append(self.SS[-1], np.array([np.asarray(x, float) + np.array([0, 0, 0, 0, TL, 0])]), axis=0)
        self.uSS[-1] = np.append(self.uSS[-1], np.array([np.asarray(u, float)]), axis=0)
        self.Qfun[-1] = np.append(self.Qfun[-1], self.Qfun[-1][-1] - 1)

    def ss_replace_lap(self, lap, x, u, qfun):
        _rec("ss_replace_lap", lap)
        self.SS[lap] = np.array(x, float); self.uSS[lap] = np.array(u, float); self.Qfun[lap] = np.array(qfun, float)

    def ss_set_selected(self, laps):
        self.sel = [int(l) for l in laps]

    # ---- compute
    def _model(self):
        return self.xStored, self.uStored, list(range(self.cfg.trToUse)), self.cfg.track

    def regress_batch(self, xLin, uLin):
        _rec("regress_batch")
        xs, us, used, pt = self._model()
        A, B, C = orc.compute_ltv_dynamics(xs, us, used, pt, np.asarray(xLin, float)[0], np.asarray(uLin, float)[0], self.N)
        return A[None], B[None], C[None], np.zeros((1, self.N), np.int32)

    def regress_points(self, x, u):
        _rec("regress_points")
        xs, us, used, pt = self._model()
        x = np.asarray(x, float).reshape(-1, 6); u = np.asarray(u, float).reshape(-1, 2)
        out = [orc.regression_and_linearization(xs, us, used, pt, xi, ui) for xi, ui in zip(x, u)]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out]), np.array([o[2] for o in out]), np.zeros(x.shape[0], np.int32)

    def _select(self, x0, zt, xPredPrev, hasPred, timeStep):
        TL = self.cfg.trackLength
        z = np.array(zt, float)
        if z[4] - x0[4] > TL / 2:
            z[4] = np.max([z[4] - TL, 0])
        order = self.sel if self.sel is not None else list(np.argsort(np.array(self.LapTime))[0:self.cfg.numSS_it])
        xp = np.asarray(xPredPrev, float) if hasPred else None
        SSsel, Qsel, Succ, SuccU = orc.terminal_components(self.SS, self.uSS, self.Qfun, self.LapTime, z, self.cfg.numSS_points, self.cfg.numSS_it,
                                                           xp, len(self.SS), int(timeStep), self.N, TL, sortedLapTime=np.array(order))
        npw = self.cfg.numSS_points // self.cfg.numSS_it + 1
        starts = []
        for l in order:
            nrm = np.abs(self.SS[l] - z[None]).sum(axis=1); mn = int(np.argmin(nrm))
            starts.append(mn - npw // 2 if mn - npw / 2 >= 0 else mn)
        return SSsel, Qsel, Succ, SuccU, z, starts

    def select_batch(self, x0, zt, xPredPrev=None, hasPred=None, timeStep=None):
        _rec("select_batch")
        hp = 0 if hasPred is None else int(np.asarray(hasPred)[0]); ts = 0 if timeStep is None else int(np.asarray(timeStep)[0])
        SSsel, Qsel, Succ, SuccU, z, starts = self._select(np.asarray(x0, float)[0], np.asarray(zt, float)[0], None if xPredPrev is None else xPredPrev[0], hp, ts)
        return dict(ssSel=SSsel.T[None], qSel=Qsel[None], succ=Succ.T[None], succU=SuccU.T[None], ztUsed=z[None], selStart=np.array([starts], np.int32),
                    status=np.zeros(1, np.int32))

    def _solve(self, A, B, C, x0, uOld, SSsel=None, Qsel=None):
        N, p = self.N, self.par
        if SSsel is None:
            P, q, Ad, l, u = orc.assemble_mpc_qp(p, A, B, C, x0, uOld)
        else:
            P, q, Ad, l, u = orc.assemble_lmpc_qp(p, A, B, C, x0, uOld, SSsel, Qsel)
        res, cert = orc.osqp_solve_exact(P, q, Ad, l, u)
        sol = res.x
        i0 = 6 * (N + 1) + 2 * N
        ns = 2 * N if p.slacks else 0
        out = dict(xPred=sol[:6 * (N + 1)].reshape(1, N + 1, 6), uPred=sol[6 * (N + 1):i0].reshape(1, N, 2),
                   slack=(sol[i0:i0 + ns] if ns else np.zeros(2 * N))[None], lambd=np.zeros((1, self.S)), sTerm=np.zeros((1, 6)),
                   mu=np.zeros((1, self.M)), status=np.array([0 if cert < 1e-6 else ST_MAXITER], np.int32), iters=np.array([res.iter], np.int32), resid=np.zeros((1, 3)))
        if SSsel is not None:
            out["lambd"] = sol[i0 + ns:i0 + ns + self.S][None]; out["sTerm"] = sol[i0 + ns + self.S:][None]
        return out

    def qp_solve_batch(self, A, Bm, Cc, x0, uOld, ssSel=None, qSel=None):
        _rec("qp_solve_batch")
        return self._solve(np.asarray(A, float)[0], np.asarray(Bm, float)[0], np.asarray(Cc, float)[0], np.asarray(x0, float)[0], np.asarray(uOld, float)[0])

    def step_batch(self, x0, xLin, uLin, uOld, zt=None, xPredPrev=None, hasPred=None, timeStep=None):
        _rec("step_batch", dict(lmpc=zt is not None))
        x0 = np.asarray(x0, float)[0]; uOld = np.asarray(uOld, float)[0]
        xs, us, used, pt = self._model()
        A, B, C = orc.compute_ltv_dynamics(xs, us, used, pt, np.asarray(xLin, float)[0], np.asarray(uLin, float)[0], self.N)
        if self.S == 0:
            out = self._solve(A, B, C, x0, uOld)
        else:
            SSsel, Qsel, Succ, SuccU, z, starts = self._select(x0, np.asarray(zt, float)[0], xPredPrev[0], int(np.asarray(hasPred)[0]), int(np.asarray(timeStep)[0]))
            out = self._solve(A, B, C, x0, uOld, SSsel, Qsel)
            lam = out["lambd"][0]
            out.update(ssSel=SSsel.T[None], qSel=Qsel[None], ztNext=(Succ @ lam)[None], ztuNext=(SuccU @ lam)[None])
        out.update(A=A[None], B=B[None], C=C[None])
        return out

    def assemble_batch(self, A, Bm, Cc, x0, uOld, ssSel=None, qSel=None):
        _rec("assemble_batch")
        A, Bm, Cc, x0, uOld = [np.asarray(a, float)[0] for a in (A, Bm, Cc, x0, uOld)]
        if ssSel is None:
            r = orc.assemble_mpc_qp(self.par, A, Bm, Cc, x0, uOld)
        else:
            r = orc.assemble_lmpc_qp(self.par, A, Bm, Cc, x0, uOld, np.asarray(ssSel, float)[0].T, np.asarray(qSel, float)[0])
        dense = lambda m: np.asarray(m.todense()) if hasattr(m, "todense") else np.asarray(m)
        return tuple(dense(a)[None] for a in r)
