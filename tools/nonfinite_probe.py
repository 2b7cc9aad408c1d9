"""Developer tool (GPU): which non-finite input does what to a step -- one poisoned field per run (argv[1]: x0nan / x0inf / uoldnan / xlinnan / ztinf / ulinnan)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import common
g = common.load_lmpc_golden()
ctx, par = common.make_lmpc_ctx(g, 4, max_batch=16)
inp = common.synthetic_inputs(g, 12, 16)
bad = {k: np.array(v, copy=True) for k, v in inp.items()}
what = sys.argv[1]
if what == "x0nan": bad["x0"][3, 0] = np.nan
if what == "x0inf": bad["x0"][3, 5] = np.inf
if what == "uoldnan": bad["uOld"][3, 1] = np.nan
if what == "xlinnan": bad["xLin"][3, 4, 1] = np.nan
if what == "ulinnan": bad["uLin"][3, 4, 0] = np.nan
if what == "ztinf": bad["zt"][3, 2] = -np.inf
out = ctx.step_batch(**bad)
print(what, "status", [hex(int(s)) for s in out["status"]], "iters", out["iters"].tolist(), flush=True)
ctx.close()
