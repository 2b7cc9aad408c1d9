"""Developer tool (GPU box): the reference's 40-lap experiment (N = 14) on the drop-in classes, recording every closed-loop QP that needed at least MIN_IT
interior-point iterations -- inputs as tests/ipm_model.StructQP takes them -- plus the iteration histogram of the run.
    python tools/capture_slow_qps.py [seed] [min_it] [laps]      -> gpurun_out/slow_qps_seed<seed>.npz"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import closed_loop, common
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
min_it = int(sys.argv[2]) if len(sys.argv) > 2 else 17
laps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
g = common.load_lmpc_golden()
flow = closed_loop.DropinFlow(g, 14)
rec = {k: [] for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel", "iters", "lap")}
all_it = []; state = dict(lap=0)
inner = flow.solve
def solve(x):
    u, st, it = inner(x)
    all_it.append(it)
    if it >= min_it:
        o = flow.ctrl._out
        rec["A"].append(o["A"][0].copy()); rec["B"].append(o["B"][0].copy()); rec["C"].append(o["C"][0].copy()); rec["x0"].append(np.array(x, float))
        rec["uOld"].append(flow._uOld_before.copy()); rec["SS"].append(np.ascontiguousarray(o["ssSel"][0].T)); rec["Qsel"].append(o["qSel"][0].copy())
        rec["iters"].append(it); rec["lap"].append(state["lap"])
    return u, st, it
flow.solve = solve
def on_lap(r):
    state["lap"] += 1
out = closed_loop.run_laps(flow, g, laps, seed=seed, on_lap=on_lap)
all_it = np.array(all_it)
print("seed %d: %d QPs, iterations mean %.2f max %d, hist(from 5) %s; %d QPs with >= %d iterations; status %s" % (
    seed, len(all_it), all_it.mean(), all_it.max(), np.bincount(all_it)[5:].tolist(), len(rec["iters"]), min_it, [r.get("status") for r in out if r.get("status") != {0: r["steps"]}]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "slow_qps_seed%d.npz" % seed), **{k: np.array(v) for k, v in rec.items()}, all_iters=all_it)
