"""Developer tool: wall time of one drop-in LMPC.solve call (N = 14, closed-loop inputs from the PID lap), with a cProfile of the host side."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import common
from tests.test_gpu_dropin_main import _Map
from racinglmpc_amd.PredictiveControllers import LMPC, MPCParams
from racinglmpc_amd.PredictiveModel import PredictiveModel
N = 14
g = common.load_lmpc_golden()
map_ = _Map(g)
n, d = 6, 2
Fx = np.array([[0., 0., 0., 0., 0., 1.], [0., 0., 0., 0., 0., -1.]]); Fu = np.kron(np.eye(2), np.array([1, -1])).T
bu = np.array([[0.5], [0.5], [10.0], [10.0]])
lp = MPCParams(n=n, d=d, N=N, Q=np.diag([0.0] * 6), R=0 * np.diag([1.0, 1.0]), dR=1 * np.array([5.0, 10 * 5.0]), Fx=Fx, bx=np.array([[0.4], [0.4]]), Fu=Fu, bu=bu,
               xRef=np.zeros(6), slacks=True, Qslack=1 * np.array([5, 25]), timeVarying=True)
xPID, uPID, xPID_glob = g["xPID"], g["uPID"], g["xPID_glob"]
xq = np.array(xPID)      # query states: a copy -- quirk E-2 (PredictiveControllers.py:394) edits row 5 of the STORED lap in place through xLin, as the reference does
pm = PredictiveModel(n, d, map_, 4)
for i in range(4): pm.addTrajectory(xPID, uPID)
lmpc = LMPC(48, 4, 500 * np.diag([1.0] * 6), lp, pm)
for i in range(4): lmpc.addTrajectory(xPID, uPID, xPID_glob)
x = xPID[0].copy()
ts = []
for t in range(300):
    try:
        t0 = time.perf_counter(); lmpc.solve(xq[t]); ts.append(time.perf_counter() - t0)
    except Exception as e:
        print("t", t, "EXC", e); print("xLin s", np.array(lmpc.xLin)[:, 4]); print("x0", xq[t]); print("zt", lmpc.zt); raise
    lmpc.addPoint(xq[t], lmpc.uPred[0])
ts = np.array(ts[20:]) * 1e3
print("drop-in LMPC.solve wall time per call: median %.3f ms, p90 %.3f ms (N = %d)" % (np.median(ts), np.percentile(ts, 90), N))
import ctypes as C
tr = (C.c_double * 5)()
lmpc._ctx.lib.lmpc_debug_step_trace(lmpc._ctx._h, tr)
n = max(tr[4], 1.0)
print("inside lmpc_step_batch, per call: stage inputs %.1f us, launch K1 + K3 %.1f us, wait for the stream %.1f us, read outputs %.1f us  (%d calls)" % (
    tr[0] / n * 1e6, tr[1] / n * 1e6, tr[2] / n * 1e6, tr[3] / n * 1e6, int(tr[4])))
st = lmpc._ctx.stats()
lmpc._ctx.set_profiling(1)
for t in range(300, 340): lmpc.solve(xq[t])
st = lmpc._ctx.stats()
print("kernel time per call (HIP events): regression %.1f us, solve %.1f us" % (st.ms_regress / max(st.n_regress_timed, 1) * 1e3, st.ms_solve / max(st.n_solve_timed, 1) * 1e3))
lmpc._ctx.set_profiling(0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for t in range(340, 440): lmpc.solve(xq[t])
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
