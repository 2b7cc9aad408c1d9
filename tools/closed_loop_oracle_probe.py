"""Developer tool (GPU box): the reference's 40-lap experiment (N = 14) on the drop-in classes; every `stride`-th closed-loop QP -- real LMPC laps in the safe set,
lane slacks active in the fast laps -- is solved again by the oracle from the kernel's own A, B, C and selection (both restated on the explicit QP of the
reference, oracle.assemble_lmpc_qp) to its certified optimum, and (x, u) compared.     python tools/closed_loop_oracle_probe.py [seed] [stride] [laps] [N]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import closed_loop, common
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 20
laps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
NH = int(sys.argv[4]) if len(sys.argv) > 4 else 14
g = common.load_lmpc_golden()
flow = closed_loop.DropinFlow(g, NH)
rec = []; cnt = [0]; state = dict(lap=0)
inner = flow.solve
def solve(x):
    u, st, it = inner(x)
    if cnt[0] % stride == 0:
        o = flow.ctrl._out
        rec.append(dict(A=o["A"][0].copy(), B=o["B"][0].copy(), C=o["C"][0].copy(), x0=np.array(x, float), uOld=flow._uOld_before.copy(), SS=np.ascontiguousarray(o["ssSel"][0].T),
                        Qsel=o["qSel"][0].copy(), xu=np.concatenate([o["xPred"][0].ravel(), o["uPred"][0].ravel()]), it=it, lap=state["lap"], st=st))
    cnt[0] += 1
    return u, st, it
flow.solve = solve
def on_lap(r):
    state["lap"] += 1
out = closed_loop.run_laps(flow, g, laps, seed=seed, on_lap=on_lap)
print("N = %d, seed %d: %d laps, %d QPs, %d sampled; last lap %d steps" % (NH, seed, len(out), cnt[0], len(rec), out[-1]["steps"]))

def work(r):
    from oracle import lmpc_oracle as orc
    par = orc.QPParams.lmpc_default(NH)
    P, q, Ao, l, u = orc.assemble_lmpc_qp(par, r["A"], r["B"], r["C"], r["x0"], r["uOld"], r["SS"], r["Qsel"])
    ex, cert = orc.osqp_solve_exact(P, q, Ao, l, u, want=1e-8)
    r2 = orc.dense_ipm_solve(P, q, Ao, l, u)                  # (a second certified optimum by the other method: on a flat QP -- error = 660 x residual on one of these -- one
    n = r["xu"].shape[0]                                      #  of the oracle's two answers can itself be 1e-6 off; the kernel is compared with the nearer one, as in the tests)
    return min(float((np.abs(r["xu"] - o[:n]) / (1 + np.abs(o[:n]))).max()) for o in (ex.x, r2.x)), float(cert)
import multiprocessing as mp
with mp.get_context("fork").Pool(min(64, (os.cpu_count() or 2) - 2)) as pool:
    res = pool.map(work, rec)
err = np.array([a for a, _ in res]); cert = np.array([c for _, c in res]); lap = np.array([r["lap"] for r in rec])
w = int(np.argmax(err))
print("worst |xu - z*| / (1 + |z*|) %.2e (lap %d, %d iterations), oracle certificates <= %.1e; by lap thirds: %s; n > 5e-7: %d" % (
    err.max(), rec[w]["lap"], rec[w]["it"], cert.max(), [float("%.2e" % err[(lap >= a) & (lap < b)].max()) for a, b in ((0, 13), (13, 27), (27, 99))], int((err > 5e-7).sum())))
bad = np.argsort(-err)[:6]
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "closed_loop_probe_worst_N%d_seed%d.npz" % (NH, seed)), err=err[bad], **{k: np.array([rec[i][k] for i in bad]) for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel", "xu", "it", "lap")})

