"""tests/closed_loop.py -- the reference's LMPC experiment (main.py:97-121) as a reusable closed loop (test infrastructure, not product).

main.py runs `Laps - numSS_it = 40` LMPC laps at N = 14 (initControllerParameters.py:46): every lap is one Simulator.sim call
(SysModel.py:22-54: solve / uPred[0, :] / addPoint / dynModel until s > TrackLength), followed by LMPC.addTrajectory and
PredictiveModel.addTrajectory (main.py:113-119).  This module drives that loop for

  * the drop-in classes (GPU path, racinglmpc_amd.PredictiveControllers.LMPC),
  * the oracle's restatement of the reference flow (OracleLMPC: restated OSQP at eps = 1e-3 + polish, or the certified optimum),
  * the oracle flow with the NumPy model of the kernel's interior-point iteration as its QP solver (tests/ipm_model.py) --
    the CPU stand-in used to study the solver in the regime the reference converges to without a GPU,

with the oracle's restatement of Simulator.dynModel as the plant (bit-exact against the reference, test_oracle_golden.py) and a
seeded noise stream (three N(0,1) draws per step, in the reference's order; by default the legacy RandomState stream the reference's own
Simulator consumes after np.random.seed(seed), see noise_source).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class TrackMap:
    """What the controllers read of Track.Map: PointAndTangent, TrackLength, halfWidth."""
    def __init__(self, g):
        self.PointAndTangent = np.array(g["track"]); self.TrackLength = float(g["trackLength"]); self.halfWidth = 0.4


def lmpc_params(N, cls):
    """initLMPCParams (initControllerParameters.py:28-59), values only; cls = MPCParams of the drop-in."""
    Fx = np.array([[0., 0., 0., 0., 0., 1.], [0., 0., 0., 0., 0., -1.]])
    Fu = np.kron(np.eye(2), np.array([1, -1])).T
    bu = np.array([[0.5], [0.5], [10.0], [10.0]])
    par = cls(n=6, d=2, N=N, Q=0 * np.eye(6), R=0 * np.eye(2), dR=5 * np.array([1.0, 10.0]), Fx=Fx, bx=(np.array([[0.4], [0.4]]),),
              Fu=Fu, bu=bu, slacks=True, Qslack=1 * np.array([5, 25]))
    return 4, 48, 500 * np.diag([1, 1, 1, 1, 1, 1]), par            # numSS_it, numSS_Points, QterminalSlack, lmpcParameters


class DropinFlow:
    """main.py:100-110 on the drop-in classes (GPU)."""
    name = "dropin"

    def __init__(self, g, N):
        from racinglmpc_amd.PredictiveControllers import LMPC, MPCParams
        from racinglmpc_amd.PredictiveModel import PredictiveModel
        self.map = TrackMap(g)
        numSS_it, numSS_Points, QterminalSlack, par = lmpc_params(N, MPCParams)
        self.pm = PredictiveModel(6, 2, self.map, 4)
        for _ in range(4):
            self.pm.addTrajectory(g["xPID"], g["uPID"])
        par.timeVarying = True
        self.ctrl = LMPC(numSS_Points, numSS_it, QterminalSlack, par, self.pm)
        for _ in range(4):
            self.ctrl.addTrajectory(g["xPID"], g["uPID"], g["xPID_glob"])

    def solve(self, x):
        self.ctrl.solve(x)
        o = self.ctrl._out
        return self.ctrl.uPred[0, :].copy(), int(o["status"][0]), int(o["iters"][0])

    def add_point(self, x, u):
        self.ctrl.addPoint(x, u)

    def end_lap(self, x, u, xg):
        self.ctrl.addTrajectory(x, u, xg); self.pm.addTrajectory(x, u)
        return float(self.ctrl.Qfun[self.ctrl.it - 1][0])


class OracleFlow:
    """The same flow on the oracle's restatement of the reference classes.  solver: "osqp" (reference flow: restated OSQP, eps = 1e-3,
    polish), "exact" (certified optimum), "ipm" (NumPy model of the kernel's interior-point iteration, tests/ipm_model.py)."""

    def __init__(self, g, N, solver="osqp", ipm_kw=None):
        from oracle import lmpc_oracle as orc
        self.orc = orc; self.name = "oracle-" + solver; self.solver = solver; self.ipm_kw = ipm_kw or {}
        self.pt = np.array(g["track"])
        p = orc.QPParams.lmpc_default(N)
        self.p = p
        self.model = orc.OracleModel(self.pt, 4)
        for _ in range(4):
            self.model.addTrajectory(np.array(g["xPID"]), np.array(g["uPID"]))
        self.ctrl = orc.OracleLMPC(p, self.model, exact=(solver == "exact"))
        for _ in range(4):
            self.ctrl.addTrajectory(np.array(g["xPID"]), np.array(g["uPID"]))
        self.dump = None                    # list collecting StructQP inputs when set
        if solver == "ipm":
            self._patch_ipm()

    def _patch_ipm(self):
        """Replace the QP solve of OracleLMPC.solve by the NumPy interior-point model (same unpack)."""
        from tests import ipm_model
        orc, flow = self.orc, self

        class _Res:
            pass

        def fake_osqp(P, q, A, l, u, polish=True, **kw):
            c = flow.ctrl
            qp = ipm_model.StructQP(flow.p, np.array(c.A), np.array(c.B), np.array(c.C), flow._x0, np.reshape(c.OldInput, -1),
                                    c.SS_PointSelectedTot, c.Qfun_SelectedTot)
            r = ipm_model.ipm_solve(qp, **flow.ipm_kw)
            res = _Res()
            res.x = np.concatenate([r["x"].ravel(), r["u"].ravel(), r["s"].ravel(), r["lam"], r["sT"]])
            res.status = 1; res.iter = r["iters"]; res.info = r
            return res
        self._fake = fake_osqp

    def solve(self, x):
        self._x0 = np.array(x, float)
        if self.solver == "ipm":
            real = self.orc.osqp_solve
            self.orc.osqp_solve = self._fake
            try:
                self.ctrl.solve(self._x0)
            finally:
                self.orc.osqp_solve = real
            st = int(self.ctrl.res.info.get("status", 0))
        else:
            self.ctrl.solve(self._x0)
            st = 0 if self.ctrl.feasible else 1
        if self.dump is not None:
            c = self.ctrl
            self.dump.append(dict(A=np.array(c.A), B=np.array(c.B), C=np.array(c.C), x0=self._x0.copy(), uOld=np.array(self._uOld_before),
                                  SS=c.SS_PointSelectedTot.copy(), Qsel=c.Qfun_SelectedTot.copy(), sol=c.res.x.copy()))
        return self.ctrl.uPred[0, :].copy(), st, int(self.ctrl.res.iter)

    def add_point(self, x, u):
        self.ctrl.addPoint(x, u)

    def end_lap(self, x, u, xg):
        self.ctrl.addTrajectory(x, u); self.model.addTrajectory(x, u)
        return float(self.ctrl.Qfun[self.ctrl.it - 1][0])


def noise_source(seed, kind="legacy"):
    """N(0,1) draws of the plant noise, one per call.  "legacy": np.random.RandomState(seed).randn -- the stream the reference's own
    Simulator.dynModel (SysModel.py:139-141: np.random.randn()) consumes after np.random.seed(seed), so that the executed reference
    (tests/golden/make_flow_golden.py), the oracle flow and the GPU closed loop see IDENTICAL draws; "pcg": np.random.default_rng(seed)."""
    if kind == "legacy":
        return np.random.RandomState(seed).randn
    return np.random.default_rng(seed).standard_normal


def run_laps(flow, g, laps, seed=5, max_steps=400, on_lap=None, dump_from=None, noise="legacy"):
    """main.py:113-119.  Returns a list of per-lap dicts: steps, lap time (Qfun[0]), status histogram, iterations (mean, max), max vx, max |ey|."""
    from oracle import lmpc_oracle as orc
    pt = np.array(g["track"]); TL = float(g["trackLength"])
    draw = noise_source(seed, noise)
    x0 = np.array([0.5, 0, 0, 0, 0, 0.0]); xS = [x0, x0]
    out = []
    for lap in range(laps):
        if dump_from is not None and lap >= dump_from and hasattr(flow, "dump") and flow.dump is None:
            flow.dump = []
        x_cl, g_cl, u_cl, sts, its = [np.array(xS[0], float)], [np.array(xS[1], float)], [], [], []
        t0 = time.time(); err = None
        for i in range(max_steps):
            if hasattr(flow, "ctrl") and hasattr(flow.ctrl, "OldInput"):
                flow._uOld_before = np.reshape(np.array(flow.ctrl.OldInput, float), -1).copy()
            try:
                u, st, it = flow.solve(x_cl[-1])
            except Exception as e:                                   # the reference raises here too (singular regression, window past a lap's end)
                err = "%s: %s" % (type(e).__name__, e); break
            u_cl.append(u); sts.append(st); its.append(it)
            flow.add_point(x_cl[-1], u_cl[-1])
            xt, gt = orc.dyn_model(pt, x_cl[-1], g_cl[-1], u_cl[-1], draw)
            x_cl.append(xt); g_cl.append(gt)
            if x_cl[-1][4] > TL:
                break
        if err is not None or x_cl[-1][4] <= TL:
            out.append(dict(lap=lap, steps=len(u_cl), error=err or "lap not finished in %d steps" % max_steps,
                            status=_hist(sts)))
            if on_lap:
                on_lap(out[-1])
            break
        xS = [np.array(x_cl[-1]) - np.array([0, 0, 0, 0, TL, 0]), np.array(g_cl[-1])]
        x_cl.pop(); g_cl.pop()
        X, U, G = np.array(x_cl), np.array(u_cl), np.array(g_cl)
        q0 = flow.end_lap(X, U, G)
        rec = dict(lap=lap, steps=int(X.shape[0]), lap_time=q0, status=_hist(sts), iters_mean=float(np.mean(its)), iters_max=int(np.max(its)),
                   vx_max=float(X[:, 0].max()), ey_max=float(np.abs(X[:, 5]).max()), seconds=round(time.time() - t0, 2))
        out.append(rec)
        if on_lap:
            on_lap(rec)
    return out


def _hist(sts):
    v, c = np.unique(np.array(sts, dtype=np.int64), return_counts=True) if len(sts) else ([], [])
    return {int(a): int(b) for a, b in zip(v, c)}
