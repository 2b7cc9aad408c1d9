"""Developer tool (GPU box): capture the inputs of closed-loop regressions that end LMPC_ST_REG_SINGULAR.

The robustness sweep `tools/robustness_sweep.py 768 10 12` shows five rollouts with that bit in generation 2.  This tool runs the same sweep, but steps the
generations one simulated step at a time (same kernels, same order, same noise: the rollouts are the same) and looks at the per-point status of every step's
regression; for a flagged (rollout, step) it keeps the step's queries xLin / uLin -- read BEFORE the step through lmpc_debug_rollout_peek -- and the regression
store as it was (the laps handed to model_add_trajectory so far).  Output: gpurun_out/reg_singular_capture.npz; the committed fixture
tests/golden/reg_singular_capture.npz keeps the captures and the four laps the regression used (the sorted store's first trToUse).

    python tools/capture_reg_singular.py [R] [G] [N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from racinglmpc_amd import _capi, rollout

R = int(sys.argv[1]) if len(sys.argv) > 1 else 768
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 12
g = bench.load_seed()
ctx = bench.make_ctx(g, NH, R, 0)
model_laps = [(np.array(g["xPID"]), np.array(g["uPID"]))] * 4          # what bench.make_ctx stored
_add = ctx.model_add_trajectory
def add(x, u):
    model_laps.append((np.array(x), np.array(u))); _add(x, u)
ctx.model_add_trajectory = add
captures = []
_run = ctx.rollout_run
state = {"gen": 0}
def stepping_run(max_steps):
    t = nd = 0
    for _ in range(max_steps):
        xl, ul, _, _ = ctx.debug_rollout_peek(R)
        t_before = ctx._ro_t
        t, nd = _run(1)
        if t == t_before:
            break
        _, _, st, rs = ctx.debug_rollout_peek(R)
        for b in np.nonzero((rs & _capi.ST_REG_SINGULAR).any(axis=1))[0]:
            captures.append(dict(gen=state["gen"], t=t_before, b=int(b), xLin=xl[b].copy(), uLin=ul[b].copy(), rst=rs[b].copy(), n_model=len(model_laps)))
        if nd >= R:
            break
    return t, nd
ctx.rollout_run = stepping_run
ro = rollout.BatchedRollouts(ctx, g["track"], seed=7)
rng = np.random.default_rng(3)
x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (R, 1))
x0[:, 0] += rng.uniform(-0.1, 0.3, R); x0[:, 5] = rng.uniform(-0.25, 0.25, R); x0[:, 3] = rng.uniform(-0.1, 0.1, R); x0[:, 1] = rng.normal(size=R) * 0.02
gen = rollout.LmpcGeneration(ro, R, K=4, T_max=400, ext=40)
for it in range(G):
    state["gen"] = it
    best = gen.run(x0, g["xPID"][1:NH + 2], g["uPID"][1:NH + 1])
    done = np.asarray(gen.last_done)
    # a finished car keeps being simulated until the slowest rollout ends, and what happens to it there does not belong to its lap (lmpc_rollout_plant_kernel):
    # keep the flags raised up to and including the crossing step
    captures[:] = [c for c in captures if c["gen"] != it or done[c["b"]] < 0 or c["t"] < done[c["b"]]]
    vals, cnt = np.unique(gen.last_status, return_counts=True)
    print("generation %d: %d steps, best laps %s, status histogram %s, captured so far %d" % (it, ctx._ro_t, [b[4] for b in best], dict(zip(vals.tolist(), cnt.tolist())), len(captures)), flush=True)
if captures:
    # first flagged step of each rollout only (later steps of a flagged rollout follow from it)
    first = {}
    for c in captures:
        first.setdefault((c["gen"], c["b"]), c)
    caps = list(first.values())
    nm = max(c["n_model"] for c in caps)
    out = dict(N=NH, track=np.array(g["track"]), trackLength=g["trackLength"], n_caps=len(caps), n_model=np.array([c["n_model"] for c in caps]),
               gen=np.array([c["gen"] for c in caps]), t=np.array([c["t"] for c in caps]), b=np.array([c["b"] for c in caps]),
               xLin=np.stack([c["xLin"] for c in caps]), uLin=np.stack([c["uLin"] for c in caps]), rst=np.stack([c["rst"] for c in caps]))
    for i in range(nm):
        out["mx%d" % i], out["mu%d" % i] = model_laps[i]
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "reg_singular_capture.npz"), **out)
    print("captured %d flagged regressions (first per rollout); %d model laps" % (len(caps), nm))
else:
    print("no LMPC_ST_REG_SINGULAR in %d generations" % G)
