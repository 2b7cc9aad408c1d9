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


def test_deferred_retry_runs_against_its_own_launch(built):
    """A launch whose retry pass is still pending must be re-solved against ITS data: the safe set as it was (lmpc_ss_add_trajectory drains and
    resolves first), its own parameter block, its own A / B / C hand-over (the context's shared hand-over buffers are not overwritten by a later
    launch before the pending one is resolved)."""
    from racinglmpc_amd import _capi
    g = common.load_lmpc_golden()
    B = 48
    inp1 = bench.synth_batch(g, B, 12, seed=1)
    inp2 = bench.synth_batch(g, B, 12, seed=2); inp2["x0"] = inp2["x0"][::-1].copy(); inp2["xLin"] = inp2["xLin"][::-1].copy(); inp2["uLin"] = inp2["uLin"][::-1].copy()
    inp2["uOld"] = inp2["uOld"][::-1].copy(); inp2["zt"] = inp2["zt"][::-1].copy(); inp2["timeStep"] = inp2["timeStep"][::-1].copy()

    def fresh():
        cfg, _ = common.lmpc_config(g, 12, max_batch=B, max_iter=7)          # most problems end at the limit: every launch asks for its retry pass
        c = _capi.Context(cfg)
        for _ in range(4):
            c.model_add_trajectory(g["xPID"], g["uPID"]); c.ss_add_trajectory(g["xPID"], g["uPID"])
        return c
    keys = ("xPred", "uPred", "lambd", "ztNext", "status", "iters")
    ctx = fresh()
    ref1 = ctx.step_batch(inp1["x0"], inp1["xLin"], inp1["uLin"], inp1["uOld"], zt=inp1["zt"], timeStep=inp1["timeStep"])
    ref2 = ctx.step_batch(inp2["x0"], inp2["xLin"], inp2["uLin"], inp2["uOld"], zt=inp2["zt"], timeStep=inp2["timeStep"])
    assert ((ref1["status"] & _capi.ST_MAXITER) != 0).any() and ctx.stats().n_retry == 2
    ctx.close()
    # (a) a store mutation behind a pending launch: resolved first, against the four-lap safe set
    ctx = fresh()
    a1, keep1 = ctx.step_dev_buffers(inp1, diagnostics=False)
    ctx.step_batch_dev(B, a1)
    assert ctx.stats().n_retry == 0
    lap5 = g["xPID"][:-30] * np.array([1.3, 1, 1, 1, 1, 1.0])
    ctx.ss_add_trajectory(lap5, g["uPID"][:-30])                        # a faster fifth lap: would change the selection of a late retry
    assert ctx.stats().n_retry == 1
    out1 = ctx.step_dev_fetch(a1, B)
    for k in keys:
        assert np.array_equal(out1[k], ref1[k]), k
    for p in keep1:
        ctx.dev_free(p)
    ctx.close()
    # (b) two launches through the context's shared hand-over buffers (A, Bm, C NULL) with different inputs
    ctx = fresh()
    a1, keep1 = ctx.step_dev_buffers(inp1, diagnostics=False); a2, keep2 = ctx.step_dev_buffers(inp2, diagnostics=False)
    for a in (a1, a2):
        a.A = None; a.Bm = None; a.C = None
    ctx.step_batch_dev(B, a1); ctx.step_batch_dev(B, a2)
    assert ctx.stats().n_retry == 1                                    # launch 1 got its pass before launch 2's regression overwrote the hand-over
    out1 = ctx.step_dev_fetch(a1, B); out2 = ctx.step_dev_fetch(a2, B)
    assert ctx.stats().n_retry == 2
    for k in keys:
        assert np.array_equal(out1[k], ref1[k]) and np.array_equal(out2[k], ref2[k]), k
    for p in keep1 + keep2:
        ctx.dev_free(p)
    ctx.close()
