"""Developer tool (CPU): shifted warm start of the interior-point iteration in closed loop, on the NumPy model of the kernels (tests/ipm_model.py) -- VERDICT r4 item 5.
Runs the oracle's LMPC state machine with the model as its QP solver for a few laps (the reference's experiment, N = 12 / 14), once cold and once per warm-start
variant, with the same plant noise; prints iterations per solve (mean / max / histogram) and the worst distance of the closed-loop inputs from the cold run's.

    python tools/warm_start_model.py [laps] [N]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(g, N, laps, variant):
    from tests import closed_loop, ipm_model
    flow = closed_loop.OracleFlow(g, N, solver="ipm")
    its = []; us = []
    state = {"u": None, "lam": None}
    base_fake = flow._fake

    def fake(P, q, A, l, u, polish=True, **kw):
        c = flow.ctrl
        qp = ipm_model.StructQP(flow.p, np.array(c.A), np.array(c.B), np.array(c.C), flow._x0, np.reshape(c.OldInput, -1), c.SS_PointSelectedTot, c.Qfun_SelectedTot)
        start = None
        if variant is not None and state["u"] is not None:
            up = state["u"]
            start = dict(variant, u=np.vstack([up[1:], up[-1:]]))
            if variant.get("use_lam"):
                start["lam"] = state["lam"]
            if variant.get("use_mu"):
                m = state["mu"]; Nn = flow.p.N
                ml, mu_, ms = m[:2 * Nn].reshape(Nn, 2), m[2 * Nn:6 * Nn].reshape(Nn, 4), m[6 * Nn:8 * Nn].reshape(Nn, 2)
                sh = lambda a: np.vstack([a[1:], a[-1:]])
                start["mu"] = (sh(ml), sh(mu_), sh(ms))
        with np.errstate(all="ignore"):
            r = ipm_model.ipm_solve(qp, start=start)
        ok = np.isfinite(r["gap"]) and r["gap"] < 1e-11
        if not ok and start is not None:                  # a warm start that fails falls back to the cold start (counted with both solves' iterations)
            with np.errstate(all="ignore"):
                r2 = ipm_model.ipm_solve(qp)
            r2["iters"] += r["iters"]; r = r2; state["fallbacks"] = state.get("fallbacks", 0) + 1
        state["u"] = r["u"].copy(); state["lam"] = r["lam"].copy(); state["mu"] = r["mu"].copy()
        its.append(r["iters"]); us.append(r["u"][0].copy())

        class _R:
            pass
        res = _R(); res.x = np.concatenate([r["x"].ravel(), r["u"].ravel(), r["s"].ravel(), r["lam"], r["sT"]]); res.status = 1; res.iter = r["iters"]; res.info = r
        return res
    flow._fake = fake
    out = closed_loop.run_laps(flow, g, laps, seed=5)
    return np.array(its), np.array(us), [o.get("steps") for o in out], state.get("fallbacks", 0)


if __name__ == "__main__":
    from tests import common
    laps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    g = common.load_lmpc_golden()
    cold_its, cold_u, steps, _ = run(g, N, laps, None)
    print("cold                                     : steps %s  iterations mean %.2f max %d  hist %s" % (steps, cold_its.mean(), cold_its.max(), np.bincount(cold_its)[5:].tolist()), flush=True)
    for name, v in (("u + mu shifted, mu0 x 0.1, cap 100", dict(mu_scale=0.1, use_mu=True)), ("u + mu shifted, mu0 x 0.01, cap 100", dict(mu_scale=0.01, use_mu=True)),
                    ("u + mu shifted, mu0 x 0.01, cap 1e4", dict(mu_scale=0.01, use_mu=True, mu_cap=1e4)), ("u + mu shifted, tight, mu0 x 0.01, cap 1e3", dict(mu_scale=0.01, use_mu=True, mu_cap=1e3, slack="tight")),
                    ("u + mu + lambda shifted, mu0 x 0.01", dict(mu_scale=0.01, use_mu=True, use_lam=True, mu_cap=1e3)),
                    ("u shifted, mu0 x 1", dict()), ("u shifted, mu0 x 0.1", dict(mu_scale=0.1)), ("u shifted, mu0 x 0.01", dict(mu_scale=0.01)),
                    ("u shifted, tight slacks, mu0 x 0.1", dict(mu_scale=0.1, slack="tight")), ("u shifted, tight slacks, mu0 x 0.01", dict(mu_scale=0.01, slack="tight")),
                    ("u shifted, tight slacks, mu0 x 0.001", dict(mu_scale=0.001, slack="tight"))):
        its, u, st, fb = run(g, N, laps, v)
        n = min(len(u), len(cold_u))
        print("%-41s: steps %s  iterations mean %.2f max %d  hist %s  fallbacks %d  |u0 - cold| over the first lap %.1e" % (
            name, st, its.mean(), its.max(), np.bincount(its)[3:].tolist(), fb, np.abs(u[:steps[0]] - cold_u[:steps[0]]).max()), flush=True)
