"""-m gpu: the retry pass (equal-step variant of the solve kernel for problems that end at the iteration limit) is launched on demand: a flagged
problem writes the launch's epoch into a host-mapped ring, the host looks at it at its next drain of the stream (lmpc_capi.hip: resolve_retries)."""
import numpy as np
import pytest

import bench
from tests import common

pytestmark = pytest.mark.gpu


def test_retry_pass_runs_only_when_a_problem_asks_for_it(built):
    from racinglmpc_amd import _capi
    g = common.load_lmpc_golden()
    B = 64
    inp = bench.synth_batch(g, B, 12, seed=1234)
    # healthy batch: no retry pass, through the host-buffer path and through the device-resident path
    cfg, _ = common.lmpc_config(g, 12, max_batch=B)
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    ref = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    assert np.all(ref["status"] == 0)
    a, keep = ctx.step_dev_buffers(inp)
    for _ in range(70):                                    # more launches than the ring has slots: the ring wraps through a drain
        ctx.step_batch_dev(B, a)
    out = ctx.step_dev_fetch(a, B)
    assert np.array_equal(out["xPred"], ref["xPred"]) and np.all(out["status"] == 0)
    assert ctx.stats().n_retry == 0 and ctx.stats().n_solve == 71
    for p in keep:
        ctx.dev_free(p)
    ctx.close()
    # iteration limit too low for most problems: they flag themselves, the retry pass runs once per launch (and, with the same limit, ends the same way)
    cfg, _ = common.lmpc_config(g, 12, max_batch=B, max_iter=7)
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    hit = (out["status"] & _capi.ST_MAXITER) != 0
    done = out["status"] == 0
    assert hit.any() and done.any()
    assert np.abs(out["xPred"][done] - ref["xPred"][done]).max() < 1e-6         # the problems that converged within the limit are untouched
    assert ctx.stats().n_retry == 1
    a, keep = ctx.step_dev_buffers(inp)
    for _ in range(3):
        ctx.step_batch_dev(B, a)
    ctx.sync()
    assert ctx.stats().n_retry == 1 + 3
    for p in keep:
        ctx.dev_free(p)
    ctx.close()
