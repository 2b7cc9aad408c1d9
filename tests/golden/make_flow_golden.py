"""tests/golden/make_flow_golden.py -- the closed-loop yardstick of tests/test_gpu_closed_loop.py: lap-length sequences of main.py's LMPC
experiment (main.py:97-121: 40 laps at N = 14) produced by EXECUTING the reference's own classes -- LMPC, PredictiveModel,
Simulator.sim / dynModel, initLMPCParams -- exactly as tests/golden/make_golden.py does (in-memory stand-ins only for cvxopt.qp and osqp.OSQP:
numpy.linalg.solve and the restated OSQP at the reference's settings eps = 1e-3 + polish; the one-line NumPy >= 2 fix of :502), after
np.random.seed(seed) (the plant noise is np.random.randn(), SysModel.py:139-141), for three seeds; and the first three laps for eight seeds.

Beside it the ORACLE's restatement of the same flow (OracleLMPC + oracle plant) on the same RandomState stream (tests/closed_loop.noise_source):
tests/test_oracle_golden.py asserts that it reproduces the executed reference lap for lap over the first laps (both flows run the same
algorithm; their regressions agree to 1e-10, not to the bit, so the chaotic loop separates them by single steps later on) and stays inside
its scatter afterwards.  ~10 min.   Needs /root/reference.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_flow_golden.py"""
import json
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
from tests import closed_loop, common  # noqa: E402


def executed_reference_laps(ref, g, seed, laps, N=14):
    """main.py:100-119 on the executed reference classes.  Returns [(steps, Qfun[it][0])]."""
    PC, ICP, PM, SM, TR, UT = ref
    n, d = 6, 2
    x0 = np.array([0.5, 0, 0, 0, 0, 0]); xS = [x0, x0]
    map_ = TR.Map(0.4)
    np.random.seed(0)
    xPID, uPID, xPID_glob, _ = SM.Simulator(map_).sim(xS, UT.PID(0.8))
    assert np.array_equal(xPID, g["xPID"]) and np.array_equal(uPID, g["uPID"])          # the seed lap of every fixture
    numSS_it, numSS_Points, Laps, TimeLMPC, QterminalSlack, lmpcParameters = ICP.initLMPCParams(map_, N)
    LMPCsim = SM.Simulator(map_, multiLap=False, flagLMPC=True)
    pm = PM.PredictiveModel(n, d, map_, 4)
    for _ in range(4):
        pm.addTrajectory(xPID, uPID)
    lmpcParameters.timeVarying = True
    lmpc = PC.LMPC(numSS_Points, numSS_it, QterminalSlack, lmpcParameters, pm)
    for _ in range(4):
        lmpc.addTrajectory(xPID, uPID, xPID_glob)
    np.random.seed(seed)
    out = []
    for it in range(numSS_it, numSS_it + laps):
        xL, uL, xLg, xS = LMPCsim.sim(xS, lmpc)
        lmpc.addTrajectory(xL, uL, xLg)
        pm.addTrajectory(xL, uL)
        out.append((int(xL.shape[0]), float(lmpc.Qfun[it][0])))
    return out


def main():
    mg.install_standins()
    ref = mg.load_reference()
    g = common.load_lmpc_golden()
    out = dict(horizon=14, flow="executed reference", solver="restated OSQP (eps 1e-3, polish) stand-in inside the reference's own LMPC / PredictiveModel / Simulator classes",
               noise="np.random.seed(seed); np.random.randn() per draw (SysModel.py:139-141)", laps40={}, qfun40={}, laps3={}, oracle_laps40={}, oracle_laps3={})
    import contextlib, io
    for seed in (5, 6, 7):
        with contextlib.redirect_stdout(io.StringIO()):                 # ("Lap completed" per lap)
            r = executed_reference_laps(ref, g, seed, 40)
        out["laps40"][str(seed)] = [a for a, _ in r]; out["qfun40"][str(seed)] = [b for _, b in r]
        o = closed_loop.run_laps(closed_loop.OracleFlow(g, 14, solver="osqp"), g, 40, seed=seed, noise="legacy")
        out["oracle_laps40"][str(seed)] = [x["steps"] for x in o]
        same = int(np.argmax(np.array(out["laps40"][str(seed)]) != np.array(out["oracle_laps40"][str(seed)]))) if out["laps40"][str(seed)] != out["oracle_laps40"][str(seed)] else 40
        print("seed %d executed reference %s\n        oracle flow        %s\n        identical over the first %d laps" % (
            seed, out["laps40"][str(seed)], out["oracle_laps40"][str(seed)], same), flush=True)
    for seed in range(8):
        with contextlib.redirect_stdout(io.StringIO()):
            r = executed_reference_laps(ref, g, seed, 3)
        out["laps3"][str(seed)] = [a for a, _ in r]
        o = closed_loop.run_laps(closed_loop.OracleFlow(g, 14, solver="osqp"), g, 3, seed=seed, noise="legacy")
        out["oracle_laps3"][str(seed)] = [x["steps"] for x in o]
    print("three laps, eight seeds: executed", out["laps3"], "oracle", out["oracle_laps3"])
    with open(os.path.join(HERE, "reference_flow_laps_n14.json"), "w") as f:
        json.dump(out, f, indent=1)
    for root, dirs, files in os.walk(mg.REF):
        assert "__pycache__" not in dirs, "reference tree was written to"


if __name__ == "__main__":
    main()
