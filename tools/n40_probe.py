"""Developer tool (GPU box): the N = 40 batch through two builds of the library with the iteration limit set to k = 0, 1, 2 and the retry pass switched off
(LMPC_NO_RETRY=1) -- the iterate and the residual triple after exactly k Newton steps, without touching the kernels' text (a code change re-rolls the
register allocation that decides whether a build shows the fault).    python tools/n40_probe.py <good.so> <bad.so>"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(tag, B=1024, N=40):
    from tests import common
    from racinglmpc_amd import _capi
    from tools.n40_model import inputs
    g = common.load_lmpc_golden()
    inp = inputs(g, N, B)
    out = {}
    for k in (0, 1, 2, 3):
        cfg, _ = common.lmpc_config(g, N, max_batch=B)
        cfg.max_iter = k
        ctx = _capi.Context(cfg)
        for _ in range(4):
            ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
        o = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
        for key in ("xPred", "uPred", "slack", "lambd", "sTerm", "mu", "resid", "status", "iters"):
            out["%s_%d" % (key, k)] = np.asarray(o[key])
        ctx.close()
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r5_n40_probe_%s.npz" % tag), **out)


if __name__ == "__main__":
    if sys.argv[1] == "_one":
        one(sys.argv[2])
    else:
        for tag, lib in zip(("good", "bad"), sys.argv[1:3]):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "_one", tag], env=dict(os.environ, LMPC_LIB=os.path.abspath(lib), LMPC_NO_RETRY="1"), capture_output=True, text=True)
            print(tag, r.returncode, r.stderr[-300:])
        a = np.load(os.path.join(ROOT, "gpurun_out", "r5_n40_probe_good.npz")); b = np.load(os.path.join(ROOT, "gpurun_out", "r5_n40_probe_bad.npz"))
        for k in (0, 1, 2, 3):
            print("after %d step(s):" % k, {key: float(np.nanmax(np.abs(a["%s_%d" % (key, k)] - b["%s_%d" % (key, k)]))) for key in ("xPred", "uPred", "slack", "lambd", "sTerm", "mu")},
                  "resid good", a["resid_%d" % k][0], "bad", b["resid_%d" % k][0], "status", np.unique(a["status_%d" % k]), np.unique(b["status_%d" % k]))
