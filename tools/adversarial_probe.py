"""Developer tool (GPU): hostile inputs at the C ABI, one case per process (a device fault kills the process: the driver below reports the exit status).

    python tools/adversarial_probe.py            # runs every case in a subprocess with a time limit, prints one line per case
    python tools/adversarial_probe.py <case>     # one case in this process
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _ctx(B=16, laps=True, **kw):
    from tests import common
    g = common.load_lmpc_golden()
    if laps:
        ctx, par = common.make_lmpc_ctx(g, 4, max_batch=B, **kw)
    else:
        from racinglmpc_amd import _capi
        cfg, par = common.lmpc_config(g, 12, max_batch=B, **kw)
        ctx = _capi.Context(cfg)
    return g, ctx, common.synthetic_inputs(g, 12, B)


def _show(tag, out, rows=(3,)):
    st = np.asarray(out["status"]); print(tag, "status of the hostile rows", [hex(int(st[r])) for r in rows], "| others non-zero:", int(np.sum(np.delete(st, list(rows)) != 0)),
                                           "| iters", np.asarray(out["iters"])[list(rows)].tolist(), flush=True)


def case_haspred_nan():
    g, ctx, inp = _ctx()
    inp["hasPred"] = np.ones(16, np.int32); xp = np.array(inp["xLin"], copy=True); xp[3] = np.nan; xp[5, :, 4] = np.inf; inp["xPredPrev"] = xp
    _show("hasPred with NaN / inf predictions:", ctx.step_batch(**inp), (3, 5))


def case_timestep():
    g, ctx, inp = _ctx()
    ts = np.array(inp["timeStep"], copy=True); ts[3] = -5; ts[5] = 2**31 - 1; inp["timeStep"] = ts
    inp["hasPred"] = np.ones(16, np.int32); xp = np.array(inp["xLin"], copy=True); xp[:, -1, 4] += 100.0; inp["xPredPrev"] = xp
    _show("timeStep -5 / INT_MAX with a crossed prediction:", ctx.step_batch(**inp), (3, 5))


def case_huge():
    g, ctx, inp = _ctx()
    inp["x0"][3] = 1e300; inp["zt"][5] = -1e300; inp["uOld"][7] = 1e308; inp["xLin"][9, 2] = 1e200; inp["uLin"][11, 3] = -1e300
    _show("finite but absurd values:", ctx.step_batch(**inp), (3, 5, 7, 9, 11))


def case_qp_nan():
    g, ctx, inp = _ctx()
    out = ctx.step_batch(**inp)
    A, B, C = np.array(out["A"]), np.array(out["B"]), np.array(out["C"]); ss, qs = np.array(out["ssSel"]), np.array(out["qSel"])
    A[3, 4, 2, 2] = np.nan; B[5, 0, 0, 0] = np.inf; C[7, 11, 5] = np.nan; ss[9, 10, 3] = np.nan; qs[11, 0] = np.inf; qs[13] = -np.inf
    _show("qp_solve_batch with NaN / inf in A, B, C, SS, Q:", ctx.qp_solve_batch(A, B, C, inp["x0"], inp["uOld"], ssSel=ss, qSel=qs), (3, 5, 7, 9, 11, 13))


def case_no_laps():
    from racinglmpc_amd import _capi
    g, ctx, inp = _ctx(laps=False)
    for what, call in (("step_batch", lambda: ctx.step_batch(**inp)), ("regress_batch", lambda: ctx.regress_batch(inp["xLin"], inp["uLin"])),
                       ("select_batch", lambda: ctx.select_batch(inp["x0"], inp["zt"]))):
        try:
            out = call(); st = out["status"] if isinstance(out, dict) else out[-1]
            print("no laps stored,", what, "returned; status", sorted(set(hex(int(s)) for s in np.asarray(st).ravel())), flush=True)
        except _capi.LmpcError as e:
            print("no laps stored,", what, "raised LmpcError:", str(e)[:100], flush=True)


def case_short_laps():
    from racinglmpc_amd import _capi
    g, ctx, inp = _ctx(laps=False)
    x, u = np.array(g["xPID"]), np.array(g["uPID"])
    for n in (1, 5, 13):
        try:
            ctx.model_add_trajectory(x[:n], u[:n]); ctx.ss_add_trajectory(x[:n], u[:n])
            print("lap of %d rows stored" % n, flush=True)
        except (_capi.LmpcError, AssertionError) as e:
            print("lap of %d rows refused: %s" % (n, str(e)[:80]), flush=True)
    for _ in range(4):
        ctx.model_add_trajectory(x[:20], u[:20]); ctx.ss_add_trajectory(x[:20], u[:20])
    try:
        _show("20-row laps only:", ctx.step_batch(**inp), (0, 1, 2))
    except _capi.LmpcError as e:
        print("20-row laps only: LmpcError", str(e)[:100], flush=True)


def case_nan_laps():
    g, ctx, inp = _ctx()
    x, u = np.array(g["xPID"], copy=True), np.array(g["uPID"], copy=True)
    x[100:200] = np.nan; x[300, 4] = np.inf; u[400] = np.nan
    ctx.model_add_trajectory(x, u); ctx.ss_add_trajectory(x, u)
    out = ctx.step_batch(**inp)
    st = np.asarray(out["status"]); print("a stored lap with NaN / inf rows: status values", sorted(set(hex(int(s)) for s in st)), "iters max", int(np.max(out["iters"])), flush=True)


def case_rollout_nan():
    from racinglmpc_amd import rollout
    g, ctx, inp = _ctx(B=16)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=3)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (16, 1)); x0[:, 5] = np.linspace(-0.05, 0.05, 16)
    x0[3, 0] = np.nan; x0[5, 5] = np.inf; x0[7, 4] = -np.inf; x0[9] = 1e300
    best = rollout.lap_and_exchange(ro, x0, g["SS0"][1:14], g["uSS0"][1:13], K=2, T_max=400)
    print("rollouts with NaN / inf / 1e300 start states: best laps", [b[4] for b in best], "from cars", [int(b[5]) if len(b) > 5 and np.isscalar(b[5]) else "?" for b in best], flush=True)
    ro.close()


def case_haspred_without_predictions():
    from racinglmpc_amd import _capi
    g, ctx, inp = _ctx()
    inp["hasPred"] = np.ones(16, np.int32)
    for what, call in (("step_batch", lambda: ctx.step_batch(**inp)), ("select_batch", lambda: ctx.select_batch(inp["x0"], inp["zt"], hasPred=inp["hasPred"]))):
        try:
            call(); print("hasPred = 1 without xPredPrev,", what, "RETURNED (stale predictions were read)", flush=True)
        except _capi.LmpcError as e:
            print("hasPred = 1 without xPredPrev,", what, "raised LmpcError:", str(e)[:90], flush=True)
    try:
        a, keep = ctx.step_dev_buffers(inp, diagnostics=False); a.xPredPrev = None; ctx.step_batch_dev(16, a); ctx.sync()      # (the C caller's mistake: a NULL the kernel would dereference)
        print("hasPred on the device without xPredPrev: RETURNED", flush=True)
    except _capi.LmpcError as e:
        print("hasPred on the device without xPredPrev raised LmpcError:", str(e)[:90], flush=True)
    inp["hasPred"] = np.zeros(16, np.int32)
    _show("hasPred all zero without xPredPrev (a first step):", ctx.step_batch(**inp), (0,))


def case_minimal_dev_args():
    g, ctx, inp = _ctx()
    ref = ctx.step_batch(**inp)
    a, keep = ctx.step_dev_buffers(inp, diagnostics=False)
    for f in ("slack", "lambda_", "sTerm", "ztNext", "ztuNext", "ssSel", "A", "Bm", "C", "timeStep", "hasPred", "xPredPrev"):
        setattr(a, f, None)                                  # every optional pointer of lmpc_step_dev_args NULL: the kernels skip those stores
    ctx.step_batch_dev(16, a); ctx.sync()
    xp = np.zeros((16, 13, 6)); st = np.zeros(16, np.int32); ctx.dev_download(a.xPred, xp); ctx.dev_download(a.status, st)
    inp2 = dict(inp); inp2["timeStep"] = np.zeros(16, np.int32)
    ref0 = ctx.step_batch(**inp2)
    print("device step with only the required pointers: status", sorted(set(hex(int(s_)) for s_ in st)), "| xPred equals the host entry point's (timeStep = 0):", bool(np.array_equal(xp, ref0["xPred"])), flush=True)


def case_hostile_config():
    from racinglmpc_amd import _capi
    from tests import common
    g = common.load_lmpc_golden()
    inp = common.synthetic_inputs(g, 12, 16)
    model, ss = common.stores_at_lap(g, 4)

    def run(tag, edit):
        cfg, par = common.lmpc_config(g, 12, max_batch=16)
        edit(cfg)
        try:
            ctx = _capi.Context(cfg)
        except _capi.LmpcError as e:
            print("config %-34s lmpc_create refused: %s" % (tag, str(e)[:80]), flush=True); return
        for x, u in model: ctx.model_add_trajectory(x, u)
        for x, u, qf in ss: ctx.ss_add_trajectory(x, u)
        out = ctx.step_batch(**inp)
        print("config %-34s status values %s, iterations max %d" % (tag, sorted(set(hex(int(s_)) for s_ in out["status"])), int(np.max(out["iters"]))), flush=True)
        ctx.close()

    def setv(name, idx, v):
        def f(cfg): getattr(cfg, name)[idx] = v
        return f
    run("Q[0][0] = NaN", setv("Q", 0, float("nan")))
    run("R = 0 and dR = 0", lambda cfg: [cfg.R.__setitem__(i, 0.0) for i in range(4)] + [cfg.dR.__setitem__(i, 0.0) for i in range(2)])
    run("R[0][0] = -1 (indefinite)", setv("R", 0, -1.0))
    run("bx = -1 (lane of negative width)", lambda cfg: [cfg.bx.__setitem__(i, -1.0) for i in range(2)])
    run("bu = 0 (no input authority)", lambda cfg: [cfg.bu.__setitem__(i, 0.0) for i in range(4)])
    run("Qslack = 0", lambda cfg: [cfg.Qslack.__setitem__(i, 0.0) for i in range(2)])
    run("hard lane rows, lane of width 0.02", lambda cfg: (setattr(cfg, "slacks", 0), [cfg.bx.__setitem__(i, 0.01) for i in range(2)]))
    run("QtermSlack = inf", lambda cfg: [cfg.QtermSlack.__setitem__(7 * i, float("inf")) for i in range(6)])
    run("trackLength = 0", lambda cfg: setattr(cfg, "trackLength", 0.0))
    run("xRef = 1e200", lambda cfg: [cfg.xRef.__setitem__(i, 1e200) for i in range(6)])


def case_rollouts_nobody_finishes():
    from racinglmpc_amd import rollout, _capi
    g, ctx, inp = _ctx(B=16)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=3)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (16, 1)); x0[:, 5] = np.linspace(-0.05, 0.05, 16)
    try:
        best = rollout.lap_and_exchange(ro, x0, g["SS0"][1:14], g["uSS0"][1:13], K=2, T_max=40)
        print("40 simulated steps, nobody finishes: exchange returned", len(best), "laps of", [b[4] for b in best], "steps", flush=True)
    except (_capi.LmpcError, RuntimeError, ValueError) as e:
        print("40 simulated steps, nobody finishes:", type(e).__name__, str(e)[:120], flush=True)
    x0[:] = np.nan
    try:
        best = rollout.lap_and_exchange(ro, x0, g["SS0"][1:14], g["uSS0"][1:13], K=2, T_max=120)
        print("every start state NaN: exchange returned", len(best), "laps", flush=True)
    except (_capi.LmpcError, RuntimeError, ValueError) as e:
        print("every start state NaN:", type(e).__name__, str(e)[:120], flush=True)
    ro.close()


CASES = {k[5:]: v for k, v in globals().items() if k.startswith("case_")}

if __name__ == "__main__":
    if len(sys.argv) > 1:
        CASES[sys.argv[1]](); sys.exit(0)
    for name in CASES:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=120)
            lines = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and not l.startswith("  File") and "Warning" not in l and "warnings.warn" not in l]
            print("== %s: exit status %d" % (name, r.returncode)); [print("   " + l[:300]) for l in (lines[:8] if r.returncode else lines[-14:])]
        except subprocess.TimeoutExpired:
            print("== %s: TIMEOUT (120 s)" % name)
        sys.stdout.flush()
