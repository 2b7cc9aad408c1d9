"""tools/time_reference.py -- times the REFERENCE'S OWN classes on this container's host CPU.

Only possible where /root/reference exists (the build container); the GPU box has no reference tree, so bench.py carries the result
of this script (profiles/cpu_reference.json) next to its own same-box `cpu_baseline` (oracle port), clearly marked as another box.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py > profiles/cpu_reference.json

What runs: PredictiveControllers.MPC / LMPC, PredictiveModel, SysModel.Simulator, Track.Map, initControllerParameters imported from
/root/reference/src, unmodified except the one-line NumPy >= 2 fix (see tests/golden/make_golden.py).  cvxopt and osqp are not
installable here: cvxopt.solvers.qp -> numpy.linalg.solve, osqp.OSQP -> the restated OSQP (oracle/osqp_restated.c) at the
reference's settings, so the solver share of these times is the restatement's, everything else is the reference's Python.
  cfg 1   : main.py:86-94, LTV-MPC path following, N = 12, 1000 closed-loop steps (Simulator.sim), single core
  cfg 256 : the bench batch (bench.synth_batch, seed 1234): LMPC.solve on each of the 256 inputs, sequentially, single core
"""
import json
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
try:
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)
except Exception:
    pass
import make_golden as mg  # noqa: E402
import bench  # noqa: E402


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return platform.processor()


def main():
    mg.install_standins()
    PC, ICP, PM, SM, TR, UT = mg.load_reference()
    N, n, d = 12, 6, 2
    g = bench.load_seed()
    map_ = TR.Map(0.4)
    out = dict(box="build container (not the GPU box)", cpu=cpu_model(), host_cores=os.cpu_count(), cores_used=1,
               solver="restated OSQP (eps 1e-3, polish) + numpy.linalg.solve stand-ins inside the reference's own classes")
    # ---- cfg 1: LTV-MPC lap (main.py:82-95)
    np.random.seed(0)
    mpcParam, ltv = ICP.initMPCParams(n, d, N, 0.8)
    pm1 = PM.PredictiveModel(n, d, map_, 1); pm1.addTrajectory(g["xPID"], g["uPID"])
    ltv.timeVarying = True
    mpc = PC.MPC(ltv, pm1)
    sim = SM.Simulator(map_)
    x0 = np.array([0.5, 0, 0, 0, 0, 0])
    t0 = time.perf_counter()
    x_cl, u_cl, g_cl, _ = sim.sim([x0, x0], mpc)                                   # 1000 steps (100 s / 0.1 s, multiLap)
    dt = time.perf_counter() - t0
    out["cfg1_ltv_mpc_lap"] = dict(steps=int(u_cl.shape[0]), seconds=dt, solves_per_s=u_cl.shape[0] / dt,
                                   note="includes the reference's plant integration (Simulator.dynModel, ~3.7 ms/step)")
    # ---- cfg 256: LMPC.solve on the bench inputs
    numSS_it, numSS_Points, Laps, TimeLMPC, Qt, lp = ICP.initLMPCParams(map_, N)
    pm = PM.PredictiveModel(n, d, map_, 4)
    for _ in range(4):
        pm.addTrajectory(g["xPID"], g["uPID"])
    lp.timeVarying = True
    lmpc = PC.LMPC(numSS_Points, numSS_it, Qt, lp, pm)
    for _ in range(4):
        lmpc.addTrajectory(g["xPID"].copy(), g["uPID"].copy(), g["xPID"].copy())
    inp = bench.synth_batch(g, 256, N, seed=1234)
    t0 = time.perf_counter()
    for b in range(256):
        lmpc.xLin = inp["xLin"][b].copy(); lmpc.uLin = inp["uLin"][b].copy(); lmpc.OldInput = inp["uOld"][b].copy()
        lmpc.zt = inp["zt"][b].copy(); lmpc.timeStep = int(inp["timeStep"][b]); lmpc.xPred = []
        lmpc.solve(inp["x0"][b])
    dt = time.perf_counter() - t0
    out["cfg256_lmpc_solves"] = dict(solves=256, seconds=dt, solves_per_s=256 / dt)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
