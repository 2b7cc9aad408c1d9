"""Drop-in for the reference's fnc/controller/PredictiveControllers.py: MPCParams, MPC, LMPC.

Same constructors, methods (solve / addTrajectory / addPoint / computeCost ...) and result attributes
(uPred, xPred, zt, zt_u, xLin, uLin, OldInput, feasible, lambd, slack, slackTerminal, SS, uSS, Qfun, SS_glob,
LapTime, it, timeStep, xStoredPredTraj, uStoredPredTraj, SSStoredPredTraj ...), so that the reference's main.py,
SysModel.Simulator and plot.py run unchanged.  One solve(x0) = one lmpc_step_batch call with B = 1: regression for
the N linearisation points, safe-set selection, QP solve and unpack all run on the GPU (liblmpc_hip.so).
There is no CPU fallback.

Differences to the reference that a caller can observe
  * the QP is solved to its certified optimum (KKT residuals <= 1e-9), whereas the reference returns OSQP's
    eps = 1e-3 iterate whenever OSQP's polish step fails;
  * the dense matrices H, q, F, b, G, E, L are not kept as attributes (they are never materialised); use
    qp_matrices() to get the reference-form H_FTOCP, q_FTOCP, [F;G], l, u of the current step for inspection;
  * slacks=False: MPC only (hard lane rows, PredictiveControllers.py:184-198); LMPC(slacks=False) fails inside the reference itself (unpackSolution
    mis-slices the solution without slack variables), so the drop-in refuses it.
"""
import datetime
from dataclasses import dataclass, field

import numpy as np

from . import _capi


@dataclass
class PythonMsg:
    def __setattr__(self, key, value):
        if not hasattr(self, key):
            raise TypeError('Cannot add new field "%s" to frozen class %s' % (key, self))
        object.__setattr__(self, key, value)


@dataclass
class MPCParams(PythonMsg):
    n: int = field(default=None)
    d: int = field(default=None)
    N: int = field(default=None)
    A: np.array = field(default=None)
    B: np.array = field(default=None)
    Q: np.array = field(default=None)
    R: np.array = field(default=None)
    Qf: np.array = field(default=None)
    dR: np.array = field(default=None)
    Qslack: float = field(default=None)
    Fx: np.array = field(default=None)
    bx: np.array = field(default=None)
    Fu: np.array = field(default=None)
    bu: np.array = field(default=None)
    xRef: np.array = field(default=None)
    slacks: bool = field(default=True)
    timeVarying: bool = field(default=False)

    def __post_init__(self):
        if self.Qf is None: self.Qf = np.zeros((self.n, self.n))
        if self.dR is None: self.dR = np.zeros(self.d)
        if self.xRef is None: self.xRef = np.zeros(self.n)


def _zero_dt():
    t = datetime.datetime.now()
    return t - t


class MPC():
    """Model predictive controller (LTI when timeVarying is False, LTV otherwise); no terminal set."""

    _numSS_it = 0
    _numSS_Points = 0
    _QterminalSlack = None

    def __init__(self, mpcParameters, predictiveModel=[]):
        p = mpcParameters
        self.N, self.Qslack, self.Q, self.Qf, self.R, self.dR = p.N, p.Qslack, p.Q, p.Qf, p.R, p.dR
        self.n, self.d, self.A, self.B = p.n, p.d, p.A, p.B
        self.Fx, self.Fu, self.bx, self.bu, self.xRef = p.Fx, p.Fu, p.bx, p.bu, p.xRef
        self.slacks, self.timeVarying = p.slacks, p.timeVarying
        self.predictiveModel = predictiveModel
        if self.n != 6 or self.d != 2:
            raise _capi.LmpcError("the GPU path is built for the racing model: n = 6 states, d = 2 inputs")
        if not self.slacks and self._numSS_it > 0:
            # PredictiveControllers.py:364-375: LMPC.unpackSolution computes the offsets of lambda and of the terminal slack as if the 2N slack
            # variables were there; without them lambd is read 2N entries late (and too short), and feasibleStateInput (:382-384) then fails on
            # the shapes.  There is no reference behaviour to reproduce.
            raise NotImplementedError("LMPC with slacks=False: the reference's own LMPC.unpackSolution / feasibleStateInput fail on this combination")
        self._ctx = None
        self._make_context()
        if self.timeVarying == True:
            self.xLin = self.predictiveModel.xStored[-1][0:self.N + 1, :]
            self.uLin = self.predictiveModel.uStored[-1][0:self.N, :]
            self.computeLTVdynamics()
        self.OldInput = np.zeros((1, 2))
        self.xPred = []
        self.solverTime = _zero_dt()
        self.linearizationTime = _zero_dt()
        self.timeStep = 0
        self.feasible = 1
        self.zt = None
        self.zt_u = None

    # -- context --------------------------------------------------------------------------------------------
    def _make_context(self):
        pm = self.predictiveModel
        has_model = self.timeVarying == True
        track = pm.map.PointAndTangent if has_model else None
        TL = pm.map.TrackLength if has_model else 0.0
        cfg = _capi.config_from(self.N, self.Q, self.R, self.Qf, self.dR, self.Qslack, self.Fx, self.bx, self.Fu, self.bu, self.xRef,
                                QterminalSlack=self._QterminalSlack, numSS_Points=self._numSS_Points, numSS_it=self._numSS_it,
                                trToUse=len(pm.usedIt) if has_model else 0, track=track, trackLength=TL, max_batch=1, slacks=bool(self.slacks))
        if has_model:
            cfg.maxNumPoint = int(pm.MaxNumPoint); cfg.h = float(pm.h); cfg.lamb = float(pm.lamb); cfg.dt = float(pm.dt)
            for i in range(5):
                cfg.scaling[i] = float(np.asarray(pm.scaling)[i, i])
        self._ctx = _capi.Context(cfg)
        if has_model:
            pm._attach(self._ctx)

    # -- reference API ------------------------------------------------------------------------------------------
    def solve(self, x0):
        x0 = np.asarray(x0, dtype=float)
        N = self.N
        t0 = datetime.datetime.now()
        self._last_x0, self._last_uOld = x0.copy(), np.reshape(np.asarray(self.OldInput, float), (2,)).copy()
        if self.timeVarying == True:
            out = self._ctx.step_batch(x0[None], np.asarray(self.xLin, float)[None, 0:N + 1], np.asarray(self.uLin, float)[None],
                                       np.reshape(np.asarray(self.OldInput, float), (1, 2)))
            self.A, self.B, self.C = list(out["A"][0]), list(out["B"][0]), list(out["C"][0])
        else:
            A = np.tile(np.asarray(self.A, float)[None, None], (1, N, 1, 1)); B = np.tile(np.asarray(self.B, float)[None, None], (1, N, 1, 1))
            out = self._ctx.qp_solve_batch(A, B, np.zeros((1, N, 6)), x0[None], np.reshape(np.asarray(self.OldInput, float), (1, 2)))
        self._raise_on_status(out["status"][0], x0)
        self.feasible = 1 if (out["status"][0] & ~_capi.ST_INEXACT) == 0 else 0
        self._out = out
        self.unpackSolution()
        self.solverTime = datetime.datetime.now() - t0
        self.feasibleStateInput()
        if self.timeVarying == True:
            self.xLin = np.vstack((self.xPred[1:, :], self.zt))
            self.uLin = np.vstack((self.uPred[1:, :], self.zt_u))
        self.OldInput = self.uPred[0, :]
        self.timeStep += 1

    def computeLTVdynamics(self):
        A, B, C, st = self._ctx.regress_batch(np.asarray(self.xLin, float)[None, 0:self.N], np.asarray(self.uLin, float)[None])
        self._raise_on_status(int(np.bitwise_or.reduce(st.ravel())), None)
        self.A, self.B, self.C = list(A[0]), list(B[0]), list(C[0])

    def addTerminalComponents(self, x0):
        pass                                         # nothing to add for the plain MPC (reference :147-155 copies matrices)

    def feasibleStateInput(self):
        self.zt = self.xPred[-1, :]
        self.zt_u = self.uPred[-1, :]

    def unpackSolution(self):
        out = self._out
        self.xPred = out["xPred"][0].copy()
        self.uPred = out["uPred"][0].copy()
        self.slack = out["slack"][0].copy() if self.slacks else np.zeros(0)            # slacks=False: z = [x, u] only (:218-221)
        self.Solution = np.concatenate([self.xPred.ravel(), self.uPred.ravel(), self.slack])

    def _raise_on_status(self, st, x0):
        # conditions on which the reference raises; everything else only clears `feasible`
        if st & _capi.ST_REG_SINGULAR:
            raise ArithmeticError("local regression is singular (fewer than 5 independent neighbours within h)")
        if st & _capi.ST_NO_SEGMENT:
            raise ValueError("curvature(): a linearisation point lies on no track segment")
        if st & _capi.ST_WINDOW:
            raise IndexError("safe-set window runs past the end of a stored lap")

    def qp_matrices(self):
        """Reference-form QP of the last solve: (H_FTOCP, q_FTOCP, [F_FTOCP; G_FTOCP], l, u), dense."""
        N = self.N
        ss = None if self._numSS_it == 0 else self.SS_PointSelectedTot.T[None]
        qs = None if self._numSS_it == 0 else self.Qfun_SelectedTot[None]
        if self.timeVarying == True:
            A, B, C = np.array(self.A)[None], np.array(self.B)[None], np.array(self.C)[None]
        else:       # LTI: one (A, B) for every stage, no affine term
            A = np.tile(np.asarray(self.A, float)[None, None], (1, N, 1, 1)); B = np.tile(np.asarray(self.B, float)[None, None], (1, N, 1, 1))
            C = np.zeros((1, N, 6))
        P, q, Ad, l, u = self._ctx.assemble_batch(A, B, C, self._last_x0[None], self._last_uOld[None], ss, qs)
        return P[0], q[0], Ad[0], l[0], u[0]


class LMPC(MPC):
    """Learning MPC: safe set + Q-function terminal components on top of the LTV MPC."""

    def __init__(self, numSS_Points, numSS_it, QterminalSlack, mpcPrameters, predictiveModel, dt=0.1):
        self._numSS_Points, self._numSS_it, self._QterminalSlack = int(numSS_Points), int(numSS_it), QterminalSlack
        super().__init__(mpcPrameters, predictiveModel)
        self.numSS_Points = numSS_Points
        self.numSS_it = numSS_it
        self.QterminalSlack = QterminalSlack
        self.OldInput = np.zeros((1, 2))
        self.xPred = []
        self.LapTime, self.SS, self.uSS, self.Qfun, self.SS_glob = [], [], [], [], []
        self.xStoredPredTraj, self.xStoredPredTraj_it = [], []
        self.uStoredPredTraj, self.uStoredPredTraj_it = [], []
        self.SSStoredPredTraj, self.SSStoredPredTraj_it = [], []
        self.zt = np.array([0.0, 0.0, 0.0, 0.0, 10.0, 0.0])
        self.it = 0

    def solve(self, x0):
        x0 = np.asarray(x0, dtype=float)
        N = self.N
        TL = self.predictiveModel.map.TrackLength
        t0 = datetime.datetime.now()
        xLin_dev = np.array(self.xLin, dtype=float)[0:N + 1]          # regression sees xLin as it is NOW (reference order :117 then :121)
        zt_dev = np.array(self.zt, dtype=float)
        # addTerminalComponents :392-394, including the in-place write through a view of a stored lap (quirk E-2)
        if (self.zt[4] - x0[4] > TL / 2):
            self.zt[4] = np.max([self.zt[4] - TL, 0])
            self.xLin[4, -1] = self.xLin[4, -1] - TL
            self._resync_aliased_laps()
        sortedLapTime = np.argsort(np.array(self.LapTime))
        self._ctx.ss_set_selected(sortedLapTime[0:self.numSS_it])
        has_pred = 0 if isinstance(self.xPred, list) else 1
        xpp = np.zeros((N + 1, 6)) if not has_pred else self.xPred
        self._last_x0, self._last_uOld = x0.copy(), np.reshape(np.asarray(self.OldInput, float), (2,)).copy()
        out = self._ctx.step_batch(x0[None], xLin_dev[None], np.asarray(self.uLin, float)[None], self._last_uOld[None],
                                   zt=zt_dev[None], xPredPrev=xpp[None], hasPred=np.array([has_pred]), timeStep=np.array([self.timeStep]))
        self.linearizationTime = datetime.datetime.now() - t0
        self._raise_on_status(out["status"][0], x0)
        self.feasible = 1 if (out["status"][0] & ~_capi.ST_INEXACT) == 0 else 0
        self._out = out
        self.A, self.B, self.C = list(out["A"][0]), list(out["B"][0]), list(out["C"][0])
        self.SS_PointSelectedTot = out["ssSel"][0].T.copy()
        self.Qfun_SelectedTot = out["qSel"][0].copy()
        self.unpackSolution()
        self.solverTime = datetime.datetime.now() - t0
        self.feasibleStateInput()
        self.xLin = np.vstack((self.xPred[1:, :], self.zt))
        self.uLin = np.vstack((self.uPred[1:, :], self.zt_u))
        self.OldInput = self.uPred[0, :]
        self.timeStep += 1

    def unpackSolution(self):
        out = self._out
        self.xPred = out["xPred"][0].copy()
        self.uPred = out["uPred"][0].copy()
        self.slack = out["slack"][0].copy()
        self.lambd = out["lambd"][0].copy()
        self.slackTerminal = out["sTerm"][0].copy()
        self.Solution = np.concatenate([self.xPred.ravel(), self.uPred.ravel(), self.slack, self.lambd, self.slackTerminal])
        self.xStoredPredTraj_it.append(self.xPred)
        self.uStoredPredTraj_it.append(self.uPred)
        self.SSStoredPredTraj_it.append(self.SS_PointSelectedTot.T)

    def feasibleStateInput(self):
        self.zt = self._out["ztNext"][0].copy()
        self.zt_u = self._out["ztuNext"][0].copy()

    def addTerminalComponents(self, x0):
        """Reference :386-416 as a call of its own (solve() does the same inside the fused step): wrap of zt[4] (:392-394, with its write
        through xLin), the numSS_it fastest laps, one window of numSS_Points / numSS_it + 1 rows per lap around the row nearest to zt.  Sets
        SS_PointSelectedTot (6, numSS_Points), Succ_SS_PointSelectedTot, Succ_uSS_PointSelectedTot (2, numSS_Points), Qfun_SelectedTot.
        The terminal equality rows and cost the reference then appends (:345-362) are what qp_matrices() returns after a solve."""
        x0 = np.asarray(x0, dtype=float)
        TL = self.predictiveModel.map.TrackLength
        if (self.zt[4] - x0[4] > TL / 2):
            self.zt[4] = np.max([self.zt[4] - TL, 0])
            self.xLin[4, -1] = self.xLin[4, -1] - TL
            self._resync_aliased_laps()
        self._ctx.ss_set_selected(np.argsort(np.array(self.LapTime))[0:self.numSS_it])
        has_pred = 0 if isinstance(self.xPred, list) else 1
        xpp = np.zeros((self.N + 1, 6)) if not has_pred else self.xPred
        o = self._ctx.select_batch(x0[None], np.array(self.zt, dtype=float)[None], xpp[None], np.array([has_pred]), np.array([self.timeStep]))
        self._raise_on_status(o["status"][0], x0)
        self.SS_PointSelectedTot = o["ssSel"][0].T.copy()
        self.Succ_SS_PointSelectedTot = o["succ"][0].T.copy()
        self.Succ_uSS_PointSelectedTot = o["succU"][0].T.copy()
        self.Qfun_SelectedTot = o["qSel"][0].copy()

    def addTrajectory(self, x, u, x_glob):
        x = np.asarray(x, dtype=float); u = np.asarray(u, dtype=float)
        self.LapTime.append(x.shape[0])
        self.SS.append(x)
        self.SS_glob.append(x_glob)
        self.uSS.append(u)
        self._ctx.ss_add_trajectory(x, u)                          # computeCost (:447-464) runs in the library
        self.Qfun.append(self._ctx.ss_get_qfun(len(self.SS) - 1))
        if self.it == 0:
            self.xLin = self.SS[self.it][1:self.N + 2, :]
            self.uLin = self.uSS[self.it][1:self.N + 1, :]
        self.xStoredPredTraj.append(self.xStoredPredTraj_it); self.xStoredPredTraj_it = []
        self.uStoredPredTraj.append(self.uStoredPredTraj_it); self.uStoredPredTraj_it = []
        self.SSStoredPredTraj.append(self.SSStoredPredTraj_it); self.SSStoredPredTraj_it = []
        self.it = self.it + 1
        self.timeStep = 0

    def computeCost(self, x, u):
        """Cost-to-go of a stored lap: steps until s >= TrackLength, 0 from there on (reference :447-464).  addTrajectory takes it from
        the device store; this host form exists for callers of the reference's method."""
        s = np.asarray(x, float)[:, 4]
        T = s.shape[0]
        stop = s >= self.predictiveModel.map.TrackLength
        stop[T - 1] = True                                                   # the last row always costs 0
        nxt = np.where(stop, np.arange(T), T)                                # index of the next row (>= r) with cost 0
        nxt = np.minimum.accumulate(nxt[::-1])[::-1]
        return (nxt - np.arange(T)).astype(float)

    def addPoint(self, x, u):
        TL = self.predictiveModel.map.TrackLength
        x = np.asarray(x, dtype=float); u = np.asarray(u, dtype=float)
        self.SS[self.it - 1] = np.append(self.SS[self.it - 1], np.array([x + np.array([0, 0, 0, 0, TL, 0])]), axis=0)
        self.uSS[self.it - 1] = np.append(self.uSS[self.it - 1], np.array([u]), axis=0)
        self.Qfun[self.it - 1] = np.append(self.Qfun[self.it - 1], self.Qfun[self.it - 1][-1] - 1)
        self._ctx.ss_add_point(x, u)

    def selectPoints(self, it, zt, numPoints):
        """Reference signature and return value (:478-514): (SS_Points (6, numPoints), SSu_Points (2, numPoints), Sel_Qfun (numPoints,)) of
        stored lap `it` around its point nearest to zt.  The search (argmin of the 1-norm over all rows) runs on the GPU; the window rows
        are then read from the host mirrors of the lap, the Q-function shift from the library's own selected costs."""
        numPoints = int(numPoints)
        ppl = self.numSS_Points // self.numSS_it
        if numPoints != ppl + 1:
            raise ValueError("the device selection uses windows of numSS_Points / numSS_it + 1 = %d points" % (ppl + 1))
        order = list(np.argsort(np.array(self.LapTime))[0:self.numSS_it])
        if it not in order:
            order[-1] = it
        self._ctx.ss_set_selected(order)
        has_pred = 0 if isinstance(self.xPred, list) else 1
        xpp = np.zeros((self.N + 1, 6)) if not has_pred else self.xPred
        z = np.asarray(zt, float)[None]
        o = self._ctx.select_batch(z, z, xpp[None], np.array([has_pred]), np.array([self.timeStep]))    # x0 = zt: no wrap of zt[4] here
        j = order.index(it)
        start = int(o["selStart"][0, j])
        if start + numPoints > self.SS[it].shape[0]:
            raise IndexError("safe-set window runs past the end of stored lap %d" % it)
        rows = np.arange(start, start + numPoints)
        shift = o["qSel"][0, j * ppl] - self.Qfun[it][start]                  # :502-512, as applied by the library
        return self.SS[it][rows, :].T, self.uSS[it][rows, :].T, self.Qfun[it][rows] + shift

    def _resync_aliased_laps(self):
        """Quirk E-2: xLin may be a VIEW of a stored lap (first solve after addTrajectory, reference :432) -- the
        reference then edits the stored lap itself.  Mirror that edit into the device stores."""
        for l, lap in enumerate(self.SS):
            if np.may_share_memory(self.xLin, lap):
                self._ctx.ss_replace_lap(l, lap, self.uSS[l], self.Qfun[l])
