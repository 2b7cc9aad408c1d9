"""Test infrastructure (GPU box): closed-loop QPs of the reference's LMPC experiment solved again by the oracle (SURVEY 8(c)-3 on the QPs the closed loop really meets).

Two routes into the solve kernels:
  * probe():          main.py's loop on the drop-in classes, one QP per step (batch 1: the four-wave kernel) -- every `stride`-th QP;
  * rollout_probe():  batched device-resident rollouts (racinglmpc_amd.rollout.LmpcGeneration; 257..1024 rollouts per GPU: the two-wave kernel, BASELINE configs[3]'s
                      per-GPU load), stepped one simulated step at a time, a few rollouts sampled per step through Context.debug_rollout_qp.
Each sampled QP is restated on the reference's explicit form (oracle.assemble_lmpc_qp) from the kernel's own A, B, C and selection (both pinned bit-exact / 1e-9 against
the reference elsewhere) and solved to its certified optimum by the oracle's dense interior-point solver; the kernel's (x, u) is compared with it.
(round 6: moved here from tools/closed_loop_oracle_probe.py -- GPU tests import test infrastructure from tests/ only.)
"""
import os

import numpy as np

from tests import common


def _work(args):
    r, NH, fast = args
    from oracle import lmpc_oracle as orc
    par = orc.QPParams.lmpc_default(NH)
    P, q, Ao, l, u = orc.assemble_lmpc_qp(par, r["A"], r["B"], r["C"], r["x0"], r["uOld"], r["SS"], r["Qsel"])
    n = r["xu"].shape[0]
    r2 = orc.dense_ipm_solve(P, q, Ao, l, u)                  # the oracle's dense interior-point solver on the explicit QP, certified by the solver-independent KKT check
    c2 = max(orc.kkt_certificate(P, q, Ao, l, u, r2.x, r2.y).values())
    opts = [r2.x]; cert = c2
    if not fast:                                              # ... and the restated ADMM + polish (up to 20 s on the near-degenerate QPs of the first laps): on a flat QP --
        ex, cert1 = orc.osqp_solve_exact(P, q, Ao, l, u, want=1e-8)      # error = 660 x residual on one of these -- either answer can itself be 1e-6 off; the kernel is compared
        opts.append(ex.x); cert = max(cert, cert1)                     # with the nearer one, as in the tests
    exu = min(float((np.abs(r["xu"] - o[:n]) / (1 + np.abs(o[:n]))).max()) for o in opts)
    ezt = 0.0; lam_star = None
    if r.get("zt") is not None:
        S = r["Qsel"].shape[0]; sl = slice(n + 2 * NH, n + 2 * NH + S)
        ezt = min(common.zt_err(r["zt"], r["ztu"], r["Succ"], r["SuccU"], o[sl]) for o in opts)
        if ezt > 2e-7:
            lam_star = r2.x[sl].copy()                       # (the parent looks at this QP again: _solve_all)
    return exu, float(cert), ezt, lam_star


def _solve_all(rec, NH, fast):
    import multiprocessing as mp
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(1)                     # (inherited by the forked children: one BLAS thread per process -- 64 processes x 256 BLAS threads each took minutes)
    except Exception:                                 # noqa: BLE001
        lim = None
    try:
        with mp.get_context("fork").Pool(max(1, min(64, (os.cpu_count() or 2) - 2, len(rec)))) as pool:          # (children never touch HIP: NumPy only)
            res = pool.map_async(_work, [(r, NH, fast) for r in rec], chunksize=1).get(timeout=420)      # (a worker that dies leaves map() waiting for ever: fail instead)
    finally:
        if lim is not None and hasattr(lim, "restore_original_limits"):
            lim.restore_original_limits()
    # zt = Succ lambda, zt_u = SuccU lambda (feasibleStateInput, :382-384) against the oracle's lambda*.  x, u, s_T are unique, lambda need not be: a QP
    # whose zt is more than 2e-7 off gets the interval of VALID values of every zt entry over all optimal lambdas (common.zt_face_err: 16 small LPs); the kernel's zt is
    # measured against that interval, and the QP is counted as one with more than one optimal lambda when an interval is wider than 1e-7.  (In THIS process: the LP
    # solver is multi-threaded, and a forked pool worker that inherits its state can deadlock -- it did, once, on the GPU box.)
    ezt = np.array([a[2] for a in res]); indet = np.zeros(len(res), int)
    for i, a in enumerate(res):
        if a[3] is not None:
            r = rec[i]
            ezt[i], width = common.zt_face_err(r["zt"], r["ztu"], r["Succ"], r["SuccU"], r["SS"], r["Qsel"], a[3])
            indet[i] = int(width > 1e-7)
    return np.array([a[0] for a in res]), np.array([a[1] for a in res]), ezt, indet


def probe(seed=5, stride=10, laps=40, NH=14, fast=False):
    """Drop-in route.  Returns (records, err, cert, out, n): the sampled QPs (inputs, the kernel's (x, u), iterations, lap), their scaled distance to the nearer oracle
    optimum, the oracle's certificates, the per-lap records of closed_loop.run_laps, the number of QPs the run solved."""
    from tests import closed_loop
    g = common.load_lmpc_golden()
    flow = closed_loop.DropinFlow(g, NH)
    rec = []; cnt = [0]; state = dict(lap=0)
    inner = flow.solve

    def solve(x):
        c = flow.ctrl
        take = cnt[0] % stride == 0
        if take:                                               # the selection's inputs as the step is about to see them (for the successor rows, below)
            hp = 0 if isinstance(c.xPred, list) else 1
            before = (np.array(c.zt, float), np.zeros((NH + 1, 6)) if not hp else np.array(c.xPred, float), hp, int(c.timeStep))
        u, st, it = inner(x)
        if take:
            o = flow.ctrl._out
            # successor rows of the selected points (Succ_SS / Succ_uSS, :404-414) from the selection entry point on the same inputs: bit-exact against the reference elsewhere
            o2 = c._ctx.select_batch(np.array(x, float)[None], before[0][None], before[1][None], np.array([before[2]]), np.array([before[3]]))
            assert np.array_equal(o2["ssSel"], o["ssSel"]) and np.array_equal(o2["qSel"], o["qSel"])
            rec.append(dict(zt=o["ztNext"][0].copy(), ztu=o["ztuNext"][0].copy(), Succ=np.ascontiguousarray(o2["succ"][0].T), SuccU=np.ascontiguousarray(o2["succU"][0].T),
                            A=o["A"][0].copy(), B=o["B"][0].copy(), C=o["C"][0].copy(), x0=np.array(x, float), uOld=flow._uOld_before.copy(), SS=np.ascontiguousarray(o["ssSel"][0].T),
                            Qsel=o["qSel"][0].copy(), xu=np.concatenate([o["xPred"][0].ravel(), o["uPred"][0].ravel()]), it=it, lap=state["lap"], st=st))
        cnt[0] += 1
        return u, st, it
    flow.solve = solve

    def on_lap(r):
        state["lap"] += 1
    out = closed_loop.run_laps(flow, g, laps, seed=seed, on_lap=on_lap)
    err, cert, ezt, indet = _solve_all(rec, NH, fast)
    for r, z, i in zip(rec, ezt, indet):
        r["ezt"] = float(z); r["indet"] = int(i)
    return rec, err, cert, out, cnt[0]


def rollout_probe(seed=5, NH=12, rollouts=1024, generations=3, per_step=4, T_max=400, K=4):
    """Batched-rollout route.  `generations` LMPC generations of `rollouts` device-resident laps each (first from the PID laps, later ones continuing the K fastest laps, with real
    LMPC laps in the safe set); in EVERY generation each simulated step hands `per_step` rollouts' QPs to the sample (rollouts spread over the batch, a different phase per step).
    Returns (records, err, cert, waves per QP, iterations (max over all QPs of all steps), status bits seen)."""
    from racinglmpc_amd import _capi, rollout
    g = common.load_lmpc_golden()
    cfg, _ = common.lmpc_config(g, NH, max_batch=rollouts)
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    ctx.debug_rollout_capture(True)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=seed)
    gen = rollout.LmpcGeneration(ro, rollouts, K=K, T_max=T_max, ext=40)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (rollouts, 1)); x0[:, 5] = np.linspace(-0.1, 0.1, rollouts); x0[:, 0] += np.linspace(0.0, 0.1, rollouts)
    rec = []; state = dict(gen=0, it_max=0, bits=0)

    def hook(t):
        # the step just taken is step t - 1: its QP had x0 = logX[t - 1], uOld = logU[t - 2] (0 at the first step)
        X, U, _, done, st, _, _ = ctx.rollout_fetch(max(t - 2, 0), t)
        for j in range(per_step):
            b = (j * (rollouts // per_step) + 37 * t + 11 * state["gen"]) % rollouts
            if done[b] >= 0 and done[b] < t - 1:
                continue                                       # (this car has crossed the line: its lane idles)
            q = ctx.debug_rollout_qp(b, 1)
            state["it_max"] = max(state["it_max"], int(q["iters"][0])); state["bits"] |= int(q["status"][0])
            rec.append(dict(zt=q["ztNext"][0], ztu=q["ztuNext"][0], Succ=np.ascontiguousarray(q["succ"][0].T), SuccU=np.ascontiguousarray(q["succU"][0].T),
                            A=q["A"][0], B=q["B"][0], C=q["C"][0], x0=X[-1, b].copy(), uOld=(U[-2, b].copy() if t >= 2 else np.zeros(2)), SS=np.ascontiguousarray(q["ssSel"][0].T),
                            Qsel=q["qSel"][0].copy(), xu=np.concatenate([q["xPred"][0].ravel(), q["uPred"][0].ravel()]), it=int(q["iters"][0]), lap=state["gen"], st=int(q["status"][0]), b=b, t=t - 1))
    gen.step_hook = hook
    waves = ctx.solver_waves(rollouts)
    for gi in range(generations):
        state["gen"] = gi
        gen.run(x0, g["xPID"][1:NH + 2], g["uPID"][1:NH + 1])
    gen.close(); ctx.close()
    err, cert, ezt, indet = _solve_all(rec, NH, True)
    for r, z, i in zip(rec, ezt, indet):
        r["ezt"] = float(z); r["indet"] = int(i)
    return rec, err, cert, waves, state["it_max"], state["bits"]
