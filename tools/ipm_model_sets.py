"""Developer tool (CPU): iteration histograms of the NumPy model of the solve kernel (tests/ipm_model.py) on fixed problem sets, for
tuning the interior-point rules before any GPU time is spent.

    python tools/ipm_model_sets.py build            # writes build_tmp/ipm_sets.npz  (oracle regression + selection; ~1 min)
    python tools/ipm_model_sets.py run "dict(th_max=1e12, carry_t=True)" [set ...]

Sets:  bench   -- the 256 problems of bench.synth_batch at N = 12 (BASELINE configs[1])
       noisy   -- the same rows with 5x the state noise
       fast    -- every 8th closed-loop QP of laps 12..39 of the reference flow at N = 14 (main.py's experiment, 66..85-step laps,
                  vx up to 3.5 m/s, lane slacks active) plus every QP of that run the round-2 rules failed on
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build_tmp", "ipm_sets.npz")


def build():
    import bench
    from oracle import lmpc_oracle as orc
    from tests import closed_loop, common, ipm_model
    g = common.load_lmpc_golden()
    pt = np.array(g["track"]); TL = float(g["trackLength"])
    sets = {}
    xs, us = [np.array(g["xPID"])] * 4, [np.array(g["uPID"])] * 4
    qf = [orc.compute_cost(xs[0], TL)] * 4
    for name, scale in (("bench", 1.0), ("noisy", 5.0)):
        inp = bench.synth_batch(g, 256, 12)
        if scale != 1.0:
            base = bench.synth_batch(g, 256, 12)
            rows = (37 * np.arange(256)) % 900
            inp["x0"] = g["xPID"][rows] + scale * (base["x0"] - g["xPID"][rows])
        recs = []
        for b in range(256):
            A, B, C = orc.compute_ltv_dynamics(xs, us, [0, 1, 2, 3], pt, inp["xLin"][b], inp["uLin"][b], 12)
            SS, Qs, _, _ = orc.terminal_components(xs, us, qf, [1000] * 4, inp["zt"][b], 48, 4, None, 4, int(inp["timeStep"][b]), 12, TL)
            recs.append(dict(A=A, B=B, C=C, x0=inp["x0"][b], uOld=inp["uOld"][b], SS=SS, Qsel=Qs))
        sets[name] = recs
        print(name, len(recs), flush=True)
    flow = closed_loop.OracleFlow(g, 14, solver="osqp")
    closed_loop.run_laps(flow, g, 40, seed=5, dump_from=12, noise="pcg")     # (the stream build_tmp/ipm_sets.npz was made with)
    d = flow.dump
    p14 = orc.QPParams.lmpc_default(14)
    keep = []
    for i, e in enumerate(d):
        hard = False
        if i % 8:
            qp = ipm_model.StructQP(p14, e["A"], e["B"], e["C"], e["x0"], e["uOld"], e["SS"], e["Qsel"])
            with np.errstate(all="ignore"):
                r = ipm_model.ipm_solve(qp, th_max=None, carry_t=False)
            hard = r["iters"] >= 20 or not np.isfinite(r["gap"])
        if i % 8 == 0 or hard:
            keep.append({k: e[k] for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel")})
    sets["fast"] = keep
    print("fast", len(keep), flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **{"%s_%s" % (n, k): np.array([e[k] for e in recs]) for n, recs in sets.items() for k in recs[0]})


def run(kw, names):
    from oracle import lmpc_oracle as orc
    from tests import ipm_model
    d = np.load(OUT)
    for name in names:
        N = d[name + "_A"].shape[1]
        p = orc.QPParams.lmpc_default(N)
        its, fails = [], []
        for i in range(d[name + "_x0"].shape[0]):
            qp = ipm_model.StructQP(p, *[d["%s_%s" % (name, k)][i] for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel")])
            with np.errstate(all="ignore"):
                r = ipm_model.ipm_solve(qp, **kw)
            its.append(r["iters"])
            ok = np.isfinite(r["gap"]) and r["gap"] < 1e-11 and r["rd"] < 1e-9 * max(1.0, np.abs(qp.Qsel).max()) and r["re"] < 1e-9
            if not ok:
                fails.append((i, r["iters"], "%.1e %.1e %.1e" % (r["gap"], r["rd"], r["re"])))
        its = np.array(its)
        print("%-6s n=%4d  mean %.2f  max %2d  hist %s  not converged: %d %s" % (name, len(its), its.mean(), its.max(), np.bincount(its)[5:].tolist(), len(fails), fails[:6]), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(eval(sys.argv[2]), sys.argv[3:] or ["bench", "noisy", "fast"])
