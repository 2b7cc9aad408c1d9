"""Developer tool (GPU box): BASELINE configs[4] (N = 40, batch 1024) -- every 4th problem against the oracle's certified optimum, the eight with the largest
|zt - Succ lambda*| / (1 + |zt|): how far the two oracle methods are apart on them (lambda* determinate?), error in (x, u), in lambda, iterations, final gap.
    python tools/n40_zt_probe.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lmpc_oracle as orc
from racinglmpc_amd import _capi
from tests import oracle_pool, common
from tools.n40_model import inputs
g = common.load_lmpc_golden()
pt = np.array(g["track"]); TL = float(g["trackLength"]); N, B = 40, 1024
par = orc.QPParams.lmpc_default(N); pid = (np.array(g["xPID"]), np.array(g["uPID"]))
inp = inputs(g, N, B)
res = oracle_pool.oracle_batch(par, pt, TL, [pid] * 4, N, inp, range(0, B, 4), solve_idx=range(0, B, 4))
cfg, _ = common.lmpc_config(g, N, max_batch=B); ctx = _capi.Context(cfg)
for _ in range(4): ctx.model_add_trajectory(*pid); ctx.ss_add_trajectory(*pid)
out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
nxu = 6 * (N + 1) + 2 * N; rows=[]
for r in res:
    b=r["b"]; S=r["Qsel"].shape[0]; sl=slice(nxu+2*N, nxu+2*N+S)
    det = common.zt_err(r["Succ"] @ r["opt"][sl], r["SuccU"] @ r["opt"][sl], r["Succ"], r["SuccU"], r["opt2"][sl])
    e = min(common.zt_err(out["ztNext"][b], out["ztuNext"][b], r["Succ"], r["SuccU"], o[sl]) for o in (r["opt"], r["opt2"]))
    w = np.concatenate([out["xPred"][b].ravel(), out["uPred"][b].ravel()])
    exu = min((np.abs(w - o[:nxu]) / (1 + np.abs(o[:nxu]))).max() for o in (r["opt"], r["opt2"]))
    dl = min(np.abs(out["lambd"][b]-o[sl]).max() for o in (r["opt"], r["opt2"]))
    rows.append((e,b,det,exu,dl,int(out["iters"][b]),float(out["resid"][b][0]) if "resid" in out else -1))
rows.sort(reverse=True)
for e,b,det,exu,dl,it,gp in rows[:8]: print("b=%d zt err %.2e  oracle methods differ %.2e  xu err %.2e  |dlambda| %.2e  iters %d  gap %.2e"%(b,e,det,exu,dl,it,gp))
