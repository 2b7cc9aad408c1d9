"""Developer tool: closed-loop robustness sweep of the solve kernels.  R rollouts (wide spread of start states) x G generations of
device-resident LMPC laps = R x ~200 x G closed-loop QPs; prints the histogram of the accumulated status bits per rollout and the
lap-time progression.  R > 1024 exercises the one-wave kernel, R <= 256 the four-wave kernel.
    python tools/robustness_sweep.py [R] [G] [N] [stop]        (stop: end early once the best lap has at most this many steps)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from racinglmpc_amd import rollout

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 12          # horizon
STOP = int(sys.argv[4]) if len(sys.argv) > 4 else 0
g = bench.load_seed()
ctx = bench.make_ctx(g, NH, R, 0)
ro = rollout.BatchedRollouts(ctx, g["track"], seed=7)
rng = np.random.default_rng(3)
x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (R, 1))
x0[:, 0] += rng.uniform(-0.1, 0.3, R); x0[:, 5] = rng.uniform(-0.25, 0.25, R); x0[:, 3] = rng.uniform(-0.1, 0.1, R); x0[:, 1] = rng.normal(size=R) * 0.02
gen = rollout.LmpcGeneration(ro, R, K=4, T_max=400, ext=40)
tot = 0; flagged = {}
for it in range(G):
    best = gen.run(x0, g["xPID"][1:NH + 2], g["uPID"][1:NH + 1])
    st = gen.last_status
    steps = ctx._ro_t
    tot += R * steps
    vals, cnt = np.unique(st, return_counts=True)
    print("generation %d: %d steps, best laps %s, status histogram over %d rollouts: %s, unfinished %d" % (
        it, steps, [b[4] for b in best], R, dict(zip(vals.tolist(), cnt.tolist())), int(np.sum(gen.last_done < 0))), flush=True)
    for v, c in zip(vals.tolist(), cnt.tolist()):
        if v:
            flagged[v] = flagged.get(v, 0) + c
    if STOP and min(b[4] for b in best) <= STOP:
        break
print("rollouts with a status bit, summed over the generations: %s" % (flagged or "none"))
print("closed-loop QPs solved: %d (N = %d, %d waves per QP)" % (tot, NH, ctx.solver_waves(R)))
