"""-m gpu: the four stages of the reference's main.py (main.py:60-121) driven through the drop-in modules, located the way main.py
locates them -- by bare module name on sys.path (racinglmpc_amd/dropin is the path seam) -- with N = 14 as in
main.py:43:  PID lap -> Regression -> MPC(mpcParam) (LTI) -> MPC(ltvmpcParam, model) (LTV) -> LMPC seeded 4x -> LMPC laps through
a Simulator.sim-shaped loop (solve / uPred[0, :] / addPoint, SysModel.py:22-54).  The plant is the oracle's restatement of
Simulator.dynModel (bit-exact against the reference, test_oracle_golden.py); the PID lap is the one the executed reference recorded.
Checked: every stage stays feasible, LMPC lap time decreases, and every attribute plot.py reads (plot.py:51-56, 107-110, 146-169) has
the reference's shape."""
import os
import sys

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


class _Map:
    """What the controllers read of Track.Map: PointAndTangent, TrackLength, halfWidth."""
    def __init__(self, g):
        self.PointAndTangent = np.array(g["track"]); self.TrackLength = float(g["trackLength"]); self.halfWidth = 0.4


def _sim(ctrl, pt, TL, xS, rng, max_steps, lmpc=False, multi_lap=True):
    """Simulator.sim (SysModel.py:22-54): returns x_cl, u_cl, x_cl_glob, xF."""
    from oracle import lmpc_oracle as orc
    x_cl, g_cl, u_cl = [np.array(xS[0], float)], [np.array(xS[1], float)], []
    for i in range(max_steps):
        ctrl.solve(x_cl[-1])
        u_cl.append(ctrl.uPred[0, :].copy())
        if lmpc:
            ctrl.addPoint(x_cl[-1], u_cl[-1])
        xt, gt = orc.dyn_model(pt, x_cl[-1], g_cl[-1], u_cl[-1], rng.standard_normal)
        x_cl.append(xt); g_cl.append(gt)
        if not multi_lap and x_cl[-1][4] > TL:
            break
    xF = [np.array(x_cl[-1]) - np.array([0, 0, 0, 0, TL, 0]), np.array(g_cl[-1])]
    x_cl.pop(); g_cl.pop()
    return np.array(x_cl), np.array(u_cl), np.array(g_cl), xF


def test_main_py_flow_through_dropin_modules(built):
    seam = os.path.join(common.ROOT, "racinglmpc_amd", "dropin")
    sys.path.insert(0, seam)
    try:
        for m in ("PredictiveControllers", "PredictiveModel"):
            sys.modules.pop(m, None)
        from PredictiveControllers import MPC, LMPC, MPCParams          # bare names, as main.py:28-31 / initControllerParameters.py:2
        from PredictiveModel import PredictiveModel
        from racinglmpc_amd.Utilities import Regression
    finally:
        sys.path.remove(seam)
    g = common.load_lmpc_golden()
    map_ = _Map(g); pt, TL = map_.PointAndTangent, map_.TrackLength
    rng = np.random.default_rng(5)
    N, n, d, vt = 14, 6, 2, 0.8                                          # main.py:43-50
    x0 = np.array([0.5, 0, 0, 0, 0, 0.0]); xS = [x0, x0]
    xPID, uPID, xPID_glob = g["xPID"], g["uPID"], g["xPID_glob"]

    # ---- initMPCParams / initLMPCParams (initControllerParameters.py:4-59), values only
    Fx = np.array([[0., 0., 0., 0., 0., 1.], [0., 0., 0., 0., 0., -1.]])
    Fu = np.kron(np.eye(2), np.array([1, -1])).T
    bu = np.array([[0.5], [0.5], [10.0], [10.0]])
    Q = np.diag([1.0, 1.0, 1, 1, 0.0, 100.0]); R = np.diag([1.0, 10.0]); xRef = np.array([vt, 0, 0, 0, 0, 0])
    mk = lambda: MPCParams(n=n, d=d, N=N, Q=Q, R=R, Fx=Fx, bx=(np.array([[2.], [2.]]),), Fu=Fu, bu=bu, xRef=xRef, slacks=True, Qslack=1 * np.array([0, 50]))
    mpcParam, ltvmpcParam = mk(), mk()
    numSS_it, numSS_Points = 4, 48
    lmpcParameters = MPCParams(n=n, d=d, N=N, Q=0 * np.eye(6), R=0 * np.eye(2), dR=5 * np.array([1.0, 10.0]), Fx=Fx, bx=(np.array([[0.4], [0.4]]),),
                               Fu=Fu, bu=bu, slacks=True, Qslack=1 * np.array([5, 25]))
    QterminalSlack = 500 * np.diag([1, 1, 1, 1, 1, 1])

    # ---- stage 2: LTI MPC on the regressed model (main.py:72-80)
    A, B, Error = Regression(xPID, uPID, 0.0000001)
    mpcParam.A = A; mpcParam.B = B
    mpc = MPC(mpcParam)
    xM, uM, gM, _ = _sim(mpc, pt, TL, xS, rng, 80)
    assert mpc.feasible == 1 and xM.shape == (80, 6) and mpc.xPred.shape == (N + 1, 6) and mpc.uPred.shape == (N, 2)
    assert abs(xM[-1, 0] - vt) < 0.15 and np.abs(xM[:, 5]).max() < 0.4   # follows the centre line at the target speed
    P, q, Ad, l, u = mpc.qp_matrices()
    assert P.shape == (10 * N + 6, 10 * N + 6) and Ad.shape[1] == 10 * N + 6

    # ---- stage 3: LTV MPC with the local-regression model (main.py:82-95)
    pm1 = PredictiveModel(n, d, map_, 1)
    pm1.addTrajectory(xPID, uPID)
    ltvmpcParam.timeVarying = True
    tv = MPC(ltvmpcParam, pm1)
    xT, uT, gT, _ = _sim(tv, pt, TL, xS, rng, 80)
    # (the reference's own LTV-MPC holds vx ~ 0.51 over these 80 steps with this tuning -- checked against the executed reference classes)
    assert tv.feasible == 1 and 0.45 < xT[-1, 0] < 0.9 and np.abs(xT[:, 5]).max() < 0.4
    assert len(tv.A) == N and tv.A[0].shape == (6, 6) and tv.B[0].shape == (6, 2) and tv.C[0].shape == (6,)
    assert tv.qp_matrices()[0].shape == P.shape

    # ---- stage 4: LMPC (main.py:97-121)
    pm = PredictiveModel(n, d, map_, 4)
    for i in range(4):
        pm.addTrajectory(xPID, uPID)
    lmpcParameters.timeVarying = True
    lmpc = LMPC(numSS_Points, numSS_it, QterminalSlack, lmpcParameters, pm)
    for i in range(4):
        lmpc.addTrajectory(xPID, uPID, xPID_glob)
    assert lmpc.it == 4 and lmpc.Qfun[0].shape == (1000,) and lmpc.xLin.shape == (N + 1, 6)
    assert np.array_equal(lmpc.Qfun[0], lmpc.computeCost(xPID, uPID))   # device computeCost == host form
    lap_times = []
    for it in range(numSS_it, numSS_it + 2):
        xL, uL, gL, xS = _sim(lmpc, pt, TL, xS, rng, 400, lmpc=True, multi_lap=False)
        assert lmpc.feasible == 1
        if it == numSS_it:                                               # the reference-form QP of the last step is available
            Pl, ql, Al, ll, ul = lmpc.qp_matrices()
            nz = 10 * N + 6 + numSS_Points + 6
            assert Pl.shape == (nz, nz) and Al.shape == (8 * N + numSS_Points + 6 * (N + 1) + 7, nz)
        lmpc.addTrajectory(xL, uL, gL)
        pm.addTrajectory(xL, uL)
        lap_times.append(lmpc.Qfun[it][0])
        print("Completed lap: ", it, " in ", np.round(lmpc.Qfun[it][0] * 0.1, 2), " seconds")
    assert lap_times[0] < 300 and lap_times[1] < lap_times[0]           # main.py's printout: lap time decreases

    # ---- what plot.py reads (plotClosedLoopLMPC :51-56, :107-110; animation_xy :146-169)
    assert lmpc.it == 6 and lmpc.N == N and lmpc.numSS_Points == numSS_Points and len(lmpc.LapTime) == 6
    for i in range(lmpc.it):
        T = lmpc.LapTime[i]
        assert lmpc.SS[i].shape[1] == 6 and lmpc.SS[i].shape[0] >= T and lmpc.uSS[i].shape == (lmpc.SS[i].shape[0], 2)
        assert lmpc.SS_glob[i].shape == (T, 6) and lmpc.Qfun[i].shape[0] == lmpc.SS[i].shape[0]
    for it in (4, 5):
        assert len(lmpc.xStoredPredTraj[it]) == lmpc.LapTime[it] == len(lmpc.SSStoredPredTraj[it]) == len(lmpc.uStoredPredTraj[it])
        assert lmpc.xStoredPredTraj[it][3].shape == (N + 1, 6) and lmpc.SSStoredPredTraj[it][3].shape == (numSS_Points, 6)
        assert lmpc.uStoredPredTraj[it][3].shape == (N, 2)
    # lap 4 was extended by addPoint while lap 5 was driven (SysModel.py:37-38): it reaches beyond the finish line
    assert lmpc.SS[4].shape[0] == lmpc.LapTime[4] + lmpc.LapTime[5] and lmpc.SS[4][-1, 4] > TL
    # addTerminalComponents as a call of its own (:386-416): the windows of the four fastest laps around zt, successors one row later
    lmpc.addTerminalComponents(xS[0])
    assert lmpc.SS_PointSelectedTot.shape == (6, numSS_Points) and lmpc.Succ_SS_PointSelectedTot.shape == (6, numSS_Points)
    assert lmpc.Succ_uSS_PointSelectedTot.shape == (2, numSS_Points) and lmpc.Qfun_SelectedTot.shape == (numSS_Points,)
    order = np.argsort(np.array(lmpc.LapTime))[0:numSS_it]
    ppl = numSS_Points // numSS_it
    for j, l in enumerate(order):
        d = np.abs(lmpc.SS[l] - lmpc.zt).sum(axis=1); mn = int(np.argmin(d))
        start = mn - (ppl + 1) // 2 if mn - (ppl + 1) / 2 >= 0 else mn
        assert np.array_equal(lmpc.SS_PointSelectedTot[:, j * ppl:(j + 1) * ppl], lmpc.SS[l][start:start + ppl].T)
        assert np.array_equal(lmpc.Succ_SS_PointSelectedTot[:, j * ppl:(j + 1) * ppl], lmpc.SS[l][start + 1:start + ppl + 1].T)
        assert np.array_equal(lmpc.Succ_uSS_PointSelectedTot[:, j * ppl:(j + 1) * ppl], lmpc.uSS[l][start + 1:start + ppl + 1].T)
    # selectPoints return contract (:478-514)
    ss, ssu, qf = lmpc.selectPoints(5, lmpc.SS[5][40], numSS_Points / numSS_it + 1)
    assert ss.shape == (6, 13) and ssu.shape == (2, 13) and qf.shape == (13,)
    assert np.array_equal(ss[:, 6], lmpc.SS[5][40]) and np.array_equal(ssu[:, 6], lmpc.uSS[5][40])      # window centred on the nearest row
    assert np.all(np.diff(qf) == -1)
