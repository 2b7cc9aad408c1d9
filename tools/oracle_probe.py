"""Developer tool (GPU box): K3 of a synthetic batch at horizon N against the oracle's certified optimum on every `stride`-th problem (bench.synth_batch inputs,
four PID laps in both stores): worst scaled errors of (x, u) and of zt where lambda* is determinate, iterations.     python tools/oracle_probe.py N B stride"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import lmpc_oracle as orc
from racinglmpc_amd import _capi
from tests import oracle_pool, common
N, B, stride = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = common.load_lmpc_golden(); pt = np.array(g["track"]); TL = float(g["trackLength"])
par = orc.QPParams.lmpc_default(N); pid = (np.array(g["xPID"]), np.array(g["uPID"]))
inp = bench.synth_batch(g, B, N, seed=4321)
res = oracle_pool.oracle_batch(par, pt, TL, [pid] * 4, N, inp, range(0, B, stride), solve_idx=range(0, B, stride))
cfg, _ = common.lmpc_config(g, N, max_batch=B); ctx = _capi.Context(cfg)
for _ in range(4):
    ctx.model_add_trajectory(*pid); ctx.ss_add_trajectory(*pid)
out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
nxu = 6 * (N + 1) + 2 * N; rows = []
for r in res:
    b = r["b"]; S = r["Qsel"].shape[0]; sl = slice(nxu + 2 * N, nxu + 2 * N + S)
    det = common.zt_err(r["Succ"] @ r["opt"][sl], r["SuccU"] @ r["opt"][sl], r["Succ"], r["SuccU"], r["opt2"][sl])
    e = min(common.zt_err(out["ztNext"][b], out["ztuNext"][b], r["Succ"], r["SuccU"], o[sl]) for o in (r["opt"], r["opt2"]))
    w = np.concatenate([out["xPred"][b].ravel(), out["uPred"][b].ravel()])
    exu = min((np.abs(w - o[:nxu]) / (1 + np.abs(o[:nxu]))).max() for o in (r["opt"], r["opt2"]))
    rows.append((exu, e if det < 1e-7 else 0.0, b, det, int(out["iters"][b]), max(r["cert"], r["cert2"])))
rows = np.array(rows)
print("N = %d, batch %d (%d waves per QP), %d problems: status != 0: %d; worst |xu - z*| / (1 + |z*|) %.2e (problem %d), worst |zt - Succ lambda*| / (1 + |zt|) where determinate %.2e (problem %d, %d determinate); "
      "iterations mean %.2f max %d; oracle certificates <= %.1e" % (N, B, ctx.solver_waves(B), len(rows), int(np.sum(out["status"] != 0)), rows[:, 0].max(), int(rows[np.argmax(rows[:, 0]), 2]),
                                                               rows[:, 1].max(), int(rows[np.argmax(rows[:, 1]), 2]), int(np.sum(rows[:, 3] < 1e-7)), out["iters"].mean(), out["iters"].max(), rows[:, 5].max()))
