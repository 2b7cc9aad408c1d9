"""tests/golden/make_noslack_golden.py -- fixture for MPCParams(slacks=False) (PredictiveControllers.py:184-198, 218-221, 248-254), produced by
EXECUTING the reference's MPC class exactly as tests/golden/make_golden.py does (same stand-ins for cvxopt.qp / osqp.OSQP, same NumPy>=2 fix).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_noslack_golden.py          (needs /root/reference)

The LTV-MPC of main.py:86-94 with hard lane constraints: a few closed-loop steps with the lane half-width tightened to 1 cm so that
the hard rows are active in the recorded optima (with initMPCParams' own bx = 2 m no state constraint is ever active and slacks on / off
give the same answer).  Output: tests/golden/ltvmpc_noslack_n12.npz (same keys as ltvmpc_n12.npz plus bx).
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from oracle import lmpc_oracle as orc  # noqa: E402


def main():
    mg.install_standins()
    PC, ICP, PM, SM, TR, UT = mg.load_reference()
    N, n, d = 12, 6, 2
    x0 = np.array([0.5, 0, 0, 0, 0, 0]); xS = [x0, x0]
    np.random.seed(0)
    map_ = TR.Map(0.4)
    vt = 0.8
    simulator = SM.Simulator(map_)
    xPID, uPID, xPID_glob, _ = simulator.sim(xS, UT.PID(vt))
    xPID_orig = xPID.copy()
    BX = 0.01
    _, p = ICP.initMPCParams(n, d, N, vt)
    p.timeVarying = True
    p.slacks = False
    p.bx = (np.array([[BX], [BX]]),)                       # the reference's own (trailing-comma) shape, initControllerParameters.py:9-10
    pm1 = PM.PredictiveModel(n, d, map_, 1)
    pm1.addTrajectory(xPID, uPID)
    mpc = PC.MPC(p, pm1)
    recs = []
    np.random.seed(1)
    xc, xg = x0.copy(), x0.copy()
    nact = []
    for t in range(40):
        rec = dict(x0=xc.copy(), xLin=np.array(mpc.xLin).copy(), uLin=np.array(mpc.uLin).copy(),
                   OldInput=np.array(mpc.OldInput, float).reshape(-1).copy())
        mg.CAPTURE.clear()
        mpc.solve(xc)
        P, q, A, l, u, sol, y, status, it_, sp = mg.CAPTURE[-1]
        rec.update(A=np.array(mpc.A), B=np.array(mpc.B), C=np.array(mpc.C), q=q, l=l, u=u, sol=sol, y=y,
                   status=status, xPred=mpc.xPred.copy(), uPred=mpc.uPred.copy())
        rec["Pp"], rec["Pi"], rec["Px"] = mg.csc_parts(P)
        rec["Ap"], rec["Ai"], rec["Ax"] = mg.csc_parts(A)
        ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
        rec["sol_opt"], rec["y_opt"], rec["cert_opt"] = ex.x, ex.y, cert
        Ax = A @ ex.x
        nact.append(int(np.sum(np.abs(Ax[:2 * N] - u[:2 * N]) < 1e-7)))
        recs.append(rec)
        # closed loop with the certified optimum's first input (the restated OSQP's eps = 1e-3 answer may leave the 1 cm lane)
        xc, xg = simulator.dynModel(xc, xg, ex.x[n * (N + 1):n * (N + 1) + d].copy())
        mpc.uPred = ex.x[n * (N + 1):n * (N + 1) + d * N].reshape(N, d)
    keep = [i for i in range(len(recs)) if i < 4 or nact[i] > 0][:14]
    recs = [recs[i] for i in keep]
    out = {k: mg.stack([r[k] for r in recs]) for k in recs[0].keys()}
    out.update(xPID=xPID_orig, uPID=uPID.copy(), track=map_.PointAndTangent.copy(), trackLength=map_.TrackLength, bx=BX)
    np.savez_compressed(os.path.join(HERE, "ltvmpc_noslack_n12.npz"), **out)
    print("no-slack fixture: %d steps kept (of 40), active hard lane rows per kept step: %s, nz = %d, certificates <= %.1e" % (
        len(recs), [nact[i] for i in keep], out["q"].shape[1], out["cert_opt"].max()))
    for root, dirs, files in os.walk(mg.REF):
        assert "__pycache__" not in dirs, "reference tree was written to"


if __name__ == "__main__":
    main()
