"""tests/golden/make_wide_golden.py -- fixtures for LMPC configurations other than lmpc_n12.npz's, produced by EXECUTING the reference's LMPC
class exactly as tests/golden/make_golden.py does (same stand-ins for cvxopt.qp / osqp.OSQP, same NumPy>=2 fix):

  lmpc_wide_n12.npz   N = 12, numSS_it = 6, numSS_Points = 72: a safe set WIDER than one wavefront (72 + 6 terminal columns > 64 lanes)
  lmpc_n14.npz        N = 14, numSS_it = 4, numSS_Points = 48: the horizon main.py itself uses (main.py:43)
  lmpc_n40.npz        N = 40, numSS_it = 4, numSS_Points = 48: BASELINE.json configs[4]'s horizon (8 steps)
  mpc_n14.npz         N = 14: main.py's stage 2 (LTI MPC on Utilities.Regression's A, B) and stage 3 (LTV MPC), ten closed-loop steps each
  lmpc_30laps_n12.npz N = 12, 30 PID laps of different speeds in both stores (BASELINE.json configs[2]): sorted insert, the four fastest
  lmpc_30laps_stress_n12.npz  the same 30 laps with numSS_it = trToUse = 30, numSS_Points = 360 (SURVEY 8(d)'s stress variant); `... make_wide_golden.py stress` makes only this one

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_wide_golden.py          (needs /root/reference)

main.py:100-110 with numSS_it = 6 (initControllerParameters.py:43-44 sets numSS_Points = 12 numSS_it): six copies of the PID lap in the
safe set, four in the regression store, then a few closed-loop steps (plant = the reference's Simulator.dynModel, driven with the certified
optimum's first input).  No addPoint between the steps, so the stores stay what they were after the first solve (whose addTerminalComponents
edits one entry of lap 0 in place through the xLin view, reference quirk E-2: every lap is handed over as its own copy here, so only SS[0]
is touched; the fixture keeps SS0 and asserts that the other laps still equal the PID lap).  Per step: the controller's inputs, the
reference's own A, B, C, selection (SS, Qfun, successors), assembled QP, and the certified optimum of that QP.
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import lmpc_oracle as orc  # noqa: E402


def make(PC, ICP, PM, SM, TR, UT, N, numSS_it, fname, steps=12):
    n, d = 6, 2
    numSS_Points = 12 * numSS_it                           # initControllerParameters.py:43-44
    x0 = np.array([0.5, 0, 0, 0, 0, 0]); xS = [x0, x0]
    np.random.seed(0)
    map_ = TR.Map(0.4)
    simulator = SM.Simulator(map_)
    xPID, uPID, xPID_glob, _ = simulator.sim(xS, UT.PID(0.8))
    _it, _pts, Laps, TimeLMPC, QterminalSlack, lmpcParameters = ICP.initLMPCParams(map_, N)
    lmpcParameters.timeVarying = True
    pm = PM.PredictiveModel(n, d, map_, 4)
    for i in range(4):
        pm.addTrajectory(xPID.copy(), uPID.copy())
    lmpc = PC.LMPC(numSS_Points, numSS_it, QterminalSlack, lmpcParameters, pm)
    for i in range(numSS_it):
        lmpc.addTrajectory(xPID.copy(), uPID.copy(), xPID_glob.copy())
    recs = []
    np.random.seed(3)
    # start 40 steps into the lap (the car is moving; the first windows are centred, not clipped at row 0)
    t0 = 40
    xc, xg = xPID[t0].copy(), xPID_glob[t0].copy()
    lmpc.xLin = xPID[t0 + 1:t0 + N + 2].copy(); lmpc.uLin = uPID[t0 + 1:t0 + N + 1].copy()
    lmpc.zt = xPID[t0 + N + 1].copy(); lmpc.OldInput = uPID[t0 - 1].copy(); lmpc.timeStep = t0
    for t in range(steps):
        xpp = None if isinstance(lmpc.xPred, list) else lmpc.xPred.copy()
        rec = dict(t=lmpc.timeStep, x0=xc.copy(), xLin=np.array(lmpc.xLin).copy(), uLin=np.array(lmpc.uLin).copy(),
                   OldInput=np.array(lmpc.OldInput, float).reshape(-1).copy(), zt=np.array(lmpc.zt).copy(),
                   hasPred=0 if xpp is None else 1, xPredPrev=np.zeros((N + 1, n)) if xpp is None else xpp)
        mg.CAPTURE.clear()
        lmpc.solve(xc)
        P, q, A, l, u, sol, y, status, it_, sp = mg.CAPTURE[-1]
        rec.update(A=np.array(lmpc.A), B=np.array(lmpc.B), C=np.array(lmpc.C),
                   SSsel=lmpc.SS_PointSelectedTot.copy(), Qsel=lmpc.Qfun_SelectedTot.copy(),
                   Succ=lmpc.Succ_SS_PointSelectedTot.copy(), SuccU=lmpc.Succ_uSS_PointSelectedTot.copy(),
                   q=q, l=l, u=u, sol=sol, status=status)
        rec["Pp"], rec["Pi"], rec["Px"] = mg.csc_parts(P)
        rec["Ap"], rec["Ai"], rec["Ax"] = mg.csc_parts(A)
        ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
        rec["sol_opt"], rec["y_opt"], rec["cert_opt"] = ex.x, ex.y, cert
        recs.append(rec)
        # continue from the certified optimum (the restated OSQP's eps = 1e-3 answer is what the reference flow would carry on with;
        # the optimum keeps the recorded inputs of the next step independent of that solver)
        xo = ex.x[:n * (N + 1)].reshape(N + 1, n); uo = ex.x[n * (N + 1):n * (N + 1) + d * N].reshape(N, d)
        lam = ex.x[n * (N + 1) + d * N + 2 * N:n * (N + 1) + d * N + 2 * N + numSS_Points]
        lmpc.xPred, lmpc.uPred = xo.copy(), uo.copy()
        lmpc.zt = np.dot(lmpc.Succ_SS_PointSelectedTot, lam); lmpc.zt_u = np.dot(lmpc.Succ_uSS_PointSelectedTot, lam)
        lmpc.xLin = np.vstack((xo[1:, :], lmpc.zt)); lmpc.uLin = np.vstack((uo[1:, :], lmpc.zt_u)); lmpc.OldInput = uo[0, :].copy()
        xc, xg = simulator.dynModel(xc, xg, uo[0, :].copy())
    out = {k: mg.stack([r[k] for r in recs]) for k in recs[0].keys()}
    out.update(xPID=xPID.copy(), uPID=uPID.copy(), track=map_.PointAndTangent.copy(), trackLength=map_.TrackLength,
               numSS_it=numSS_it, numSS_Points=numSS_Points, nSS=len(lmpc.SS))
    out["N"] = N
    out["SS0"] = lmpc.SS[0].copy(); out["Qfun"] = lmpc.Qfun[0].copy()                   # (lap 0: edited in place by the first solve, quirk E-2)
    assert int(np.sum(lmpc.SS[0] != xPID)) <= 1                                         # (here the first solve does not trigger the edit: zt starts next to x0)
    for l_ in range(len(lmpc.SS)):
        assert np.array_equal(lmpc.uSS[l_], uPID) and np.array_equal(lmpc.Qfun[l_], lmpc.Qfun[0]) and (l_ == 0 or np.array_equal(lmpc.SS[l_], xPID))
    np.savez_compressed(os.path.join(HERE, fname), **out)
    nact = [int(np.sum(r["sol_opt"][n * (N + 1) + d * N + 2 * N:n * (N + 1) + d * N + 2 * N + numSS_Points] > 1e-6)) for r in recs]
    print(fname + ": %d steps, nz = %d, m = %d, certificates <= %.1e, lambdas above 1e-6 per step: %s" % (
        len(recs), out["q"].shape[1], out["l"].shape[1], out["cert_opt"].max(), nact))


def make_30laps(PC, ICP, PM, SM, TR, UT, fname="lmpc_30laps_n12.npz", steps=10, numSS_it=4, trToUse=4, store_laps=True):
    """BASELINE.json configs[2] / SURVEY 8(d) "safe set from 30 laps": 30 single-lap PID trajectories (lap i at target speed 0.6 + 0.02 i,
    np.random.seed(i)) handed to PredictiveModel.addTrajectory (sorted insert, :35-46; the regression uses the first trToUse = 4 of the sorted
    store) and LMPC.addTrajectory (the selection uses argsort(LapTime)[:4], :395-402), then closed-loop steps started on lap 29."""
    N, n, d = 12, 6, 2
    numSS_Points = 12 * numSS_it                           # initControllerParameters.py:43-44
    map_ = TR.Map(0.4)
    x0 = np.array([0.5, 0, 0, 0, 0, 0])
    _it, _pts, Laps, TimeLMPC, QterminalSlack, lmpcParameters = ICP.initLMPCParams(map_, N)
    lmpcParameters.timeVarying = True
    pm = PM.PredictiveModel(n, d, map_, trToUse)
    laps = []
    for i in range(30):
        np.random.seed(i)
        sim1 = SM.Simulator(map_, multiLap=False)
        xl, ul, gl, _ = sim1.sim([x0, x0], UT.PID(0.6 + 0.02 * i))
        laps.append((xl.copy(), ul.copy(), gl.copy()))
        pm.addTrajectory(xl.copy(), ul.copy())
    lmpc = PC.LMPC(numSS_Points, numSS_it, QterminalSlack, lmpcParameters, pm)      # (its constructor reads the model's last stored lap, :89)
    for xl, ul, gl in laps:
        lmpc.addTrajectory(xl.copy(), ul.copy(), gl.copy())
    xq, uq, gq = laps[29]
    recs = []
    simulator = SM.Simulator(map_)
    np.random.seed(3)
    t0 = 30
    xc, xg = xq[t0].copy(), gq[t0].copy()
    lmpc.xLin = xq[t0 + 1:t0 + N + 2].copy(); lmpc.uLin = uq[t0 + 1:t0 + N + 1].copy()
    lmpc.zt = xq[t0 + N + 1].copy(); lmpc.OldInput = uq[t0 - 1].copy(); lmpc.timeStep = t0
    for t in range(steps):
        xpp = None if isinstance(lmpc.xPred, list) else lmpc.xPred.copy()
        rec = dict(t=lmpc.timeStep, x0=xc.copy(), xLin=np.array(lmpc.xLin).copy(), uLin=np.array(lmpc.uLin).copy(),
                   OldInput=np.array(lmpc.OldInput, float).reshape(-1).copy(), zt=np.array(lmpc.zt).copy(),
                   hasPred=0 if xpp is None else 1, xPredPrev=np.zeros((N + 1, n)) if xpp is None else xpp)
        mg.CAPTURE.clear()
        lmpc.solve(xc)
        P, q, A, l, u, sol, y, status, it_, sp = mg.CAPTURE[-1]
        rec.update(A=np.array(lmpc.A), B=np.array(lmpc.B), C=np.array(lmpc.C),
                   SSsel=lmpc.SS_PointSelectedTot.copy(), Qsel=lmpc.Qfun_SelectedTot.copy(),
                   Succ=lmpc.Succ_SS_PointSelectedTot.copy(), SuccU=lmpc.Succ_uSS_PointSelectedTot.copy(),
                   q=q, l=l, u=u, sol=sol, status=status)
        rec["Pp"], rec["Pi"], rec["Px"] = mg.csc_parts(P)
        rec["Ap"], rec["Ai"], rec["Ax"] = mg.csc_parts(A)
        ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
        rec["sol_opt"], rec["y_opt"], rec["cert_opt"] = ex.x, ex.y, cert
        recs.append(rec)
        xo = ex.x[:n * (N + 1)].reshape(N + 1, n); uo = ex.x[n * (N + 1):n * (N + 1) + d * N].reshape(N, d)
        lam = ex.x[n * (N + 1) + d * N + 2 * N:n * (N + 1) + d * N + 2 * N + numSS_Points]
        lmpc.xPred, lmpc.uPred = xo.copy(), uo.copy()
        lmpc.zt = np.dot(lmpc.Succ_SS_PointSelectedTot, lam); lmpc.zt_u = np.dot(lmpc.Succ_uSS_PointSelectedTot, lam)
        lmpc.xLin = np.vstack((xo[1:, :], lmpc.zt)); lmpc.uLin = np.vstack((uo[1:, :], lmpc.zt_u)); lmpc.OldInput = uo[0, :].copy()
        xc, xg = simulator.dynModel(xc, xg, uo[0, :].copy())
    out = {k: mg.stack([r[k] for r in recs]) for k in recs[0].keys()}
    out.update(track=map_.PointAndTangent.copy(), trackLength=map_.TrackLength, numSS_it=numSS_it, numSS_Points=numSS_Points, N=N, nLaps=30, trToUse=trToUse)
    base = None if store_laps else np.load(os.path.join(HERE, "lmpc_30laps_n12.npz"))      # (the stress fixture shares the 30 laps of lmpc_30laps_n12.npz: same seeds, asserted equal)
    for i, (xl, ul, gl) in enumerate(laps):
        assert np.array_equal(lmpc.SS[i], xl) and np.array_equal(lmpc.uSS[i], ul)          # (no in-place edit happened: xLin was replaced before the first solve)
        if store_laps:
            out["lapx%d" % i] = xl; out["lapu%d" % i] = ul; out["Qfun%d" % i] = lmpc.Qfun[i].copy()
        else:
            assert np.array_equal(base["lapx%d" % i], xl) and np.array_equal(base["lapu%d" % i], ul) and np.array_equal(base["Qfun%d" % i], lmpc.Qfun[i])
    out["modelOrder"] = np.array([[j for j, (xl, _, _) in enumerate(laps) if xl is not None and xs.shape == xl.shape and np.array_equal(xs, xl)][0] for xs in pm.xStored])
    out["LapTime"] = np.array(lmpc.LapTime)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname + ": %d steps, lap lengths %d..%d, model store order (first 6): %s, fastest four: %s, certificates <= %.1e" % (
        len(recs), min(lmpc.LapTime), max(lmpc.LapTime), list(out["modelOrder"][:6]), list(np.argsort(lmpc.LapTime)[:4]), out["cert_opt"].max()))


def make_30laps_stress(*ref):
    """SURVEY 8(d)'s scan-heavy stress variant on the executed reference: numSS_it = trToUse = 30 (every stored lap takes part in the regression,
    PredictiveModel.py:31, 52-58, and in the safe set, PredictiveControllers.py:395-412), numSS_Points = 360 (nz = 492 variables, 366 terminal-block
    columns), six closed-loop steps from lap 29.  The 30 laps are those of lmpc_30laps_n12.npz (same seeds; asserted equal, not stored twice)."""
    make_30laps(*ref, fname="lmpc_30laps_stress_n12.npz", steps=6, numSS_it=30, trToUse=30, store_laps=False)


def make_mpc_n14(PC, ICP, PM, SM, TR, UT, fname="mpc_n14.npz", steps=10):
    """main.py:71-94 at its own horizon N = 14: stage 2, the LTI MPC on (A, B) from Utilities.Regression (lamb = 1e-7), and stage 3, the LTV
    MPC on the local regressions around the shifted prediction; ten closed-loop steps each from x0 = [0.5, 0, 0, 0, 0, 0]."""
    N, n, d, vt = 14, 6, 2, 0.8
    x0 = np.array([0.5, 0, 0, 0, 0, 0]); xS = [x0, x0]
    np.random.seed(0)
    map_ = TR.Map(0.4)
    simulator = SM.Simulator(map_)
    xPID, uPID, xPID_glob, _ = simulator.sim(xS, UT.PID(vt))
    mpcParam, ltvmpcParam = ICP.initMPCParams(n, d, N, vt)
    A_lti, B_lti, _err = UT.Regression(xPID, uPID, 0.0000001)
    mpcParam.A = A_lti; mpcParam.B = B_lti
    out = dict(xPID=xPID.copy(), uPID=uPID.copy(), track=map_.PointAndTangent.copy(), trackLength=map_.TrackLength, N=N, A_lti=A_lti.copy(), B_lti=B_lti.copy())
    for tag, ctrl in (("lti", PC.MPC(mpcParam)), ("ltv", None)):
        if ctrl is None:
            pm1 = PM.PredictiveModel(n, d, map_, 1)
            pm1.addTrajectory(xPID.copy(), uPID.copy())
            ltvmpcParam.timeVarying = True
            ctrl = PC.MPC(ltvmpcParam, pm1)
        recs = []
        np.random.seed(1)
        xc, xg = x0.copy(), x0.copy()
        for t in range(steps):
            rec = dict(x0=xc.copy(), OldInput=np.array(ctrl.OldInput, float).reshape(-1).copy())
            if tag == "ltv":
                rec.update(xLin=np.array(ctrl.xLin).copy(), uLin=np.array(ctrl.uLin).copy())
            mg.CAPTURE.clear()
            ctrl.solve(xc)
            P, q, A, l, u, sol, y, status, it_, sp = mg.CAPTURE[-1]
            if tag == "ltv":
                rec.update(A=np.array(ctrl.A), B=np.array(ctrl.B), C=np.array(ctrl.C))
            rec.update(q=q, l=l, u=u, status=status)
            rec["Pp"], rec["Pi"], rec["Px"] = mg.csc_parts(P)
            rec["Ap"], rec["Ai"], rec["Ax"] = mg.csc_parts(A)
            ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
            rec["sol_opt"], rec["cert_opt"] = ex.x, cert
            recs.append(rec)
            uo = ex.x[n * (N + 1):n * (N + 1) + d * N].reshape(N, d); xo = ex.x[:n * (N + 1)].reshape(N + 1, n)
            # carry on from the certified optimum (MPC.solve's tail, :131-137, with the optimum in place of the eps = 1e-3 answer)
            ctrl.xPred, ctrl.uPred = xo.copy(), uo.copy()
            ctrl.zt, ctrl.zt_u = xo[-1, :].copy(), uo[-1, :].copy()
            if tag == "ltv":
                ctrl.xLin = np.vstack((xo[1:, :], ctrl.zt)); ctrl.uLin = np.vstack((uo[1:, :], ctrl.zt_u))
            ctrl.OldInput = uo[0, :].copy()
            xc, xg = simulator.dynModel(xc, xg, uo[0, :].copy())
        for k in recs[0].keys():
            out[tag + "_" + k] = mg.stack([r[k] for r in recs])
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname + ": %d + %d steps, nz = %d, certificates <= %.1e / %.1e" % (steps, steps, out["lti_q"].shape[1], out["lti_cert_opt"].max(), out["ltv_cert_opt"].max()))


def main():
    mg.install_standins()
    ref = mg.load_reference()
    if sys.argv[1:] == ["stress"]:                       # (only the stress fixture: the others are not touched)
        make_30laps_stress(*ref)
        return
    make_mpc_n14(*ref)
    make_30laps(*ref)
    make(*ref, 12, 6, "lmpc_wide_n12.npz")
    make(*ref, 14, 4, "lmpc_n14.npz")
    make(*ref, 40, 4, "lmpc_n40.npz", steps=8)
    make_30laps_stress(*ref)
    for root, dirs, files in os.walk(mg.REF):
        assert "__pycache__" not in dirs, "reference tree was written to"


if __name__ == "__main__":
    main()
