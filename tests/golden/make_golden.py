"""tests/golden/make_golden.py -- generates the committed golden fixtures by EXECUTING the reference.

Run (only possible where /root/reference exists, i.e. the build container):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is reference code and what is not:
  * Track.Map, SysModel.Simulator, Utilities.PID, PredictiveModel.PredictiveModel,
    PredictiveControllers.{MPCParams,MPC,LMPC}, initControllerParameters.* are imported from
    /root/reference/src and run UNMODIFIED, except for the one-line NumPy>=2 fix of
    PredictiveControllers.py:502 (`self.xPred == []` raises on NumPy 2; intended meaning
    "no prediction yet") which is applied to the module source in memory before exec.
  * `sol_opt` / `y_opt` / `cert_opt`: the certified optimum of the recorded QP (oracle.osqp_solve_exact) and its
    solver-independent KKT certificate -- the value GPU results are compared with.
  * `cvxopt` and `osqp` are third-party packages that are not installed here.  In-memory stand-ins
    are registered in sys.modules: cvxopt.solvers.qp(Q, b) -> numpy.linalg.solve(Q, -b) (what an
    unconstrained qp is), osqp.OSQP -> oracle.lmpc_oracle.osqp_solve (the restated OSQP algorithm).
    Hence fields produced by the reference's OWN arithmetic (A, B, C, safe-set selection, P, q,
    A_osqp, l, u, state machine) are reference-pinned; the QP solution `sol` is oracle-produced.

Outputs (tests/golden/):
  lmpc_n12.npz  -- PID seed lap + 2 closed-loop LMPC laps (N=12), per-step records (sampled)
  ltvmpc_n12.npz -- a few LTV-MPC (MPC class, timeVarying=True) steps for the no-terminal-set variant
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src"

from oracle import lmpc_oracle as orc  # noqa: E402

CAPTURE = []   # the osqp stand-in appends (P, q, A, l, u, x, y, status, iter) here


def install_standins():
    cv = types.ModuleType("cvxopt")
    solvers = types.ModuleType("cvxopt.solvers")
    solvers.options = {}

    def matrix(a, *args, **kw):
        return np.array(a, dtype=float)

    def spmatrix(*a, **k):
        raise NotImplementedError

    def qp(Q, b, *a, **k):
        x = np.linalg.solve(np.asarray(Q), -np.asarray(b))
        return {"x": x.reshape(-1, 1)}

    solvers.qp = qp
    cv.matrix, cv.spmatrix, cv.solvers = matrix, spmatrix, solvers
    sys.modules["cvxopt"] = cv
    sys.modules["cvxopt.solvers"] = solvers

    oq = types.ModuleType("osqp")

    class _Info:
        pass

    class _Res:
        pass

    class OSQP:
        def setup(self, P=None, q=None, A=None, l=None, u=None, verbose=False, polish=False, **kw):
            self.args = (P, q, A, l, u, polish)

        def warm_start(self, x=None, y=None):
            raise NotImplementedError

        def solve(self):
            P, q, A, l, u, polish = self.args
            r = orc.osqp_solve(P, q, A, l, u, polish=polish)
            CAPTURE.append((P.copy(), np.array(q, float), A.copy(), np.array(l, float), np.array(u, float),
                            r.x.copy(), r.y.copy(), r.status, r.iter, r.status_polish))
            res = _Res(); res.x = r.x; res.y = r.y; res.info = _Info(); res.info.status_val = r.status
            return res

    oq.OSQP = OSQP
    sys.modules["osqp"] = oq


def load_reference():
    for sub in ("fnc/simulator", "fnc/controller", "fnc", ""):
        sys.path.append(os.path.join(REF, sub))
    src = open(os.path.join(REF, "fnc/controller/PredictiveControllers.py")).read()
    old = "if self.xPred == []:"
    assert src.count(old) == 1
    src = src.replace(old, "if isinstance(self.xPred, list):   # NumPy>=2 fix of :502")
    mod = types.ModuleType("PredictiveControllers")
    mod.__file__ = os.path.join(REF, "fnc/controller/PredictiveControllers.py")
    sys.modules["PredictiveControllers"] = mod
    exec(compile(src, mod.__file__, "exec"), mod.__dict__)
    import initControllerParameters, PredictiveModel, SysModel, Track, Utilities  # noqa: E401
    return mod, initControllerParameters, PredictiveModel, SysModel, Track, Utilities


def csc_parts(M):
    from scipy import sparse
    M = sparse.csc_matrix(M); M.sort_indices()
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(float)


def stack(arrs):
    """np.array of a list of arrays; ragged 1-D int/float lists (CSC parts) are zero-padded to the longest."""
    arrs = [np.asarray(a) for a in arrs]
    if len({a.shape for a in arrs}) == 1:
        return np.array(arrs)
    L = max(a.shape[0] for a in arrs)
    out = np.zeros((len(arrs), L), dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        out[i, :a.shape[0]] = a
    return out


def main():
    install_standins()
    PC, ICP, PM, SM, TR, UT = load_reference()
    N, n, d = 12, 6, 2
    x0 = np.array([0.5, 0, 0, 0, 0, 0]); xS = [x0, x0]
    np.random.seed(0)
    map_ = TR.Map(0.4)
    vt = 0.8
    mpcParam, ltvmpcParam = ICP.initMPCParams(n, d, N, vt)
    numSS_it, numSS_Points, Laps, TimeLMPC, QterminalSlack, lmpcParameters = ICP.initLMPCParams(map_, N)
    simulator = SM.Simulator(map_)
    LMPCsim = SM.Simulator(map_, multiLap=False, flagLMPC=True)
    pid = UT.PID(vt)
    xPID, uPID, xPID_glob, _ = simulator.sim(xS, pid)
    xPID_orig = xPID.copy()          # the reference later corrupts one entry in place (quirk E-2)
    out = dict(track=map_.PointAndTangent.copy(), trackLength=map_.TrackLength, xPID=xPID_orig, uPID=uPID.copy(),
               xPID_glob=xPID_glob.copy())

    # ------------------------------------------------------------------ LTV-MPC (main.py:86-94), few steps
    pm1 = PM.PredictiveModel(n, d, map_, 1)
    pm1.addTrajectory(xPID, uPID)
    ltvmpcParam.timeVarying = True
    mpc = PC.MPC(ltvmpcParam, pm1)
    recs = []
    np.random.seed(1)
    xc, xg = x0.copy(), x0.copy()
    for t in range(12):
        rec = dict(x0=xc.copy(), xLin=np.array(mpc.xLin).copy(), uLin=np.array(mpc.uLin).copy(),
                   OldInput=np.array(mpc.OldInput, float).reshape(-1).copy())
        CAPTURE.clear()
        mpc.solve(xc)
        P, q, A, l, u, sol, y, status, it_, sp = CAPTURE[-1]
        rec.update(A=np.array(mpc.A), B=np.array(mpc.B), C=np.array(mpc.C), q=q, l=l, u=u, sol=sol, y=y,
                   status=status, xPred=mpc.xPred.copy(), uPred=mpc.uPred.copy())
        rec["Pp"], rec["Pi"], rec["Px"] = csc_parts(P)
        rec["Ap"], rec["Ai"], rec["Ax"] = csc_parts(A)
        ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
        rec["sol_opt"], rec["y_opt"], rec["cert_opt"] = ex.x, ex.y, cert
        recs.append(rec)
        xc, xg = simulator.dynModel(xc, xg, mpc.uPred[0, :].copy())
    keys = recs[0].keys()
    ltv = {k: stack([r[k] for r in recs]) for k in keys}
    ltv.update(xPID=xPID_orig, uPID=uPID.copy(), track=out["track"], trackLength=out["trackLength"])
    np.savez_compressed(os.path.join(HERE, "ltvmpc_n12.npz"), **ltv)
    print("ltvmpc fixture: %d steps, status all solved: %s" % (len(recs), np.all(ltv["status"] == 1)))

    # ------------------------------------------------------------------ LMPC (main.py:100-121), 2 laps
    pm = PM.PredictiveModel(n, d, map_, 4)
    for i in range(4):
        pm.addTrajectory(xPID, uPID)
    lmpcParameters.timeVarying = True
    lmpc = PC.LMPC(numSS_Points, numSS_it, QterminalSlack, lmpcParameters, pm)
    for i in range(4):
        lmpc.addTrajectory(xPID, uPID, xPID_glob)

    records = []
    orig_solve = lmpc.solve
    step_counter = [0]

    def rec_solve(x0_):
        t = step_counter[0]
        lap = lmpc.it
        xpp = None if isinstance(lmpc.xPred, list) else lmpc.xPred.copy()
        rec = dict(lap=lap, t=lmpc.timeStep, x0=np.array(x0_, float).copy(), xLin=np.array(lmpc.xLin).copy(),
                   uLin=np.array(lmpc.uLin).copy(), OldInput=np.array(lmpc.OldInput, float).reshape(-1).copy(),
                   zt=lmpc.zt.copy(), hasPred=0 if xpp is None else 1,
                   xPredPrev=np.zeros((N + 1, n)) if xpp is None else xpp,
                   ssLen=np.array([s.shape[0] for s in lmpc.SS] + [0] * (8 - len(lmpc.SS))),
                   modelLen=np.array(list(pm.lapTime) + [0] * (8 - len(pm.lapTime))),
                   LapTime=np.array(list(lmpc.LapTime) + [0] * (8 - len(lmpc.LapTime))))
        CAPTURE.clear()
        orig_solve(x0_)
        P, q, A, l, u, sol, y, status, it_, sp = CAPTURE[-1]
        rec.update(A=np.array(lmpc.A), B=np.array(lmpc.B), C=np.array(lmpc.C), ztWrapped=np.array(rec["zt"]),
                   SSsel=lmpc.SS_PointSelectedTot.copy(), Qsel=lmpc.Qfun_SelectedTot.copy(),
                   Succ=lmpc.Succ_SS_PointSelectedTot.copy(), SuccU=lmpc.Succ_uSS_PointSelectedTot.copy(),
                   q=q, l=l, u=u, sol=sol, y=y, status=status, iters=it_, polish=sp,
                   xPred=lmpc.xPred.copy(), uPred=lmpc.uPred.copy(), ztNext=lmpc.zt.copy(), ztuNext=lmpc.zt_u.copy())
        rec["Pp"], rec["Pi"], rec["Px"] = csc_parts(P)
        rec["Ap"], rec["Ai"], rec["Ax"] = csc_parts(A)
        rec["keep"] = 1 if (lmpc.timeStep - 1 < 3 or (lmpc.timeStep - 1) % 8 == 0) else 0
        if rec["keep"]:
            ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
            rec["sol_opt"], rec["y_opt"], rec["cert_opt"] = ex.x, ex.y, cert
        records.append(rec)
        step_counter[0] = t + 1

    lmpc.solve = rec_solve
    laps_x, laps_u = [], []
    np.random.seed(2)
    nlaps = 2
    for it in range(numSS_it, numSS_it + nlaps):
        xL, uL, xLg, xS = LMPCsim.sim(xS, lmpc)
        lmpc.addTrajectory(xL, uL, xLg)
        pm.addTrajectory(xL, uL)
        laps_x.append(xL.copy()); laps_u.append(uL.copy())
        print("lap", it, "steps", xL.shape[0], "Qfun0", lmpc.Qfun[it][0])

    # final (append-only) stores; a record at step t sees the first ssLen[l] rows of SS[l]
    for l_, (S, U, Qf) in enumerate(zip(lmpc.SS, lmpc.uSS, lmpc.Qfun)):
        out["SS%d" % l_] = S.copy(); out["uSS%d" % l_] = U.copy(); out["Qfun%d" % l_] = Qf.copy()
    out["nSS"] = len(lmpc.SS)
    for l_, (X, U) in enumerate(zip(pm.xStored, pm.uStored)):
        out["xStored%d" % l_] = X.copy(); out["uStored%d" % l_] = U.copy()
    out["nModel"] = len(pm.xStored)
    for i, (X, U) in enumerate(zip(laps_x, laps_u)):
        out["lapx%d" % i] = X; out["lapu%d" % i] = U
    # full closed-loop inputs of every step (small) for state-machine replay
    out["all_x0"] = np.array([r["x0"] for r in records]); out["all_lap"] = np.array([r["lap"] for r in records])
    out["all_u0"] = np.array([r["uPred"][0] for r in records])
    out["all_status"] = np.array([r["status"] for r in records]); out["all_iters"] = np.array([r["iters"] for r in records])
    out["all_polish"] = np.array([r["polish"] for r in records])
    out["all_xPred"] = np.array([r["xPred"] for r in records]); out["all_uPred"] = np.array([r["uPred"] for r in records])
    # controller inputs of EVERY step (the rec_* records are a sample): lets a test re-run all closed-loop steps through the
    # HIP path and compare with the answer the reference flow produced at its own solver settings
    for k in ("t", "xLin", "uLin", "OldInput", "zt", "hasPred", "xPredPrev", "ztNext", "ztuNext", "Qsel", "sol", "y"):
        out["all_" + k] = np.array([r[k] for r in records])
    kept = [r for r in records if r["keep"]]
    for k in kept[0].keys():
        if k == "keep":
            continue
        out["rec_" + k] = stack([r[k] for r in kept])
    np.savez_compressed(os.path.join(HERE, "lmpc_n12.npz"), **out)
    print("lmpc fixture: %d steps total, %d kept; solved=%d polish_ok=%d; iters min/med/max = %d/%d/%d" % (
        len(records), len(kept), int(np.sum(out["all_status"] == 1)), int(np.sum(out["all_polish"] == 1)),
        out["all_iters"].min(), int(np.median(out["all_iters"])), out["all_iters"].max()))
    for p_ in (REF,):
        for root, dirs, files in os.walk(p_):
            assert "__pycache__" not in dirs, "reference tree was written to"


if __name__ == "__main__":
    main()
