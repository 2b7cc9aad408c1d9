"""Developer tool (GPU box): iteration histogram and launch time of the N = 40 bench batch through the library LMPC_LIB names (one process per library)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tests import common
from tests.test_gpu_certificates import _ctx_pid
from tools.n40_model import inputs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = common.load_lmpc_golden(); ctx, par = _ctx_pid(g, 40, B); inp = inputs(g, 40, B)
for _ in range(3):
    t0 = time.perf_counter(); out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"]); dt = time.perf_counter() - t0
it = np.asarray(out["iters"])
print("%s B=%d: host call %.3f ms, iterations mean %.3f max %d sum %d hist(from 5) %s, status %s, n_retry %d" % (os.path.basename(os.environ.get("LMPC_LIB", "product")), B, dt * 1e3, it.mean(), it.max(), it.sum(),
      np.bincount(it)[5:].tolist(), dict(zip(*[a.tolist() for a in np.unique(out["status"], return_counts=True)])), int(ctx.stats().n_retry)))
