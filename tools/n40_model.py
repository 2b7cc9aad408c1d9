"""Developer tool (CPU): BASELINE configs[4] (N = 40, batch 1024) through the NumPy model of the solve kernel (tests/ipm_model.py).

    python tools/n40_model.py build                 # build_tmp/n40_set.npz: A, B, C (oracle regression) + selection of the 1024 bench problems (~1 min)
    python tools/n40_model.py run [n]               # iteration histogram + per-iteration trace (gap, r_d, r_e, sigma, alpha_p, alpha_d) of the first n problems
                                                    #   -> profiles/r5_n40_model.json
    python tools/n40_model.py compare <gpu.npz>     # against the kernel's own trace (tools/n40_trace.py on the GPU box): first diverging quantity per problem

The inputs are the ones tests/test_gpu_certificates.py::test_other_horizons_certificate[40-1024] and bench.py's config_N40 use.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "build_tmp", "n40_set.npz")


def inputs(g, N, B):
    from tests import common
    return common.synthetic_inputs(g, N, B)


def build(N=40, B=1024):
    from oracle import lmpc_oracle as orc
    from tests import common
    g = common.load_lmpc_golden()
    pt = np.array(g["track"]); TL = float(g["trackLength"])
    xs, us = [np.array(g["xPID"])] * 4, [np.array(g["uPID"])] * 4
    qf = [orc.compute_cost(xs[0], TL)] * 4
    inp = inputs(g, N, B)
    rec = {k: [] for k in ("A", "B", "C", "SS", "Qsel")}
    for b in range(B):
        A, Bm, C = orc.compute_ltv_dynamics(xs, us, [0, 1, 2, 3], pt, inp["xLin"][b], inp["uLin"][b], N)
        SS, Qs, _, _ = orc.terminal_components(xs, us, qf, [1000] * 4, inp["zt"][b], 48, 4, None, 4, int(inp["timeStep"][b]), N, TL)
        for k, v in zip(("A", "B", "C", "SS", "Qsel"), (A, Bm, C, SS, Qs)):
            rec[k].append(np.asarray(v))
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, x0=inp["x0"], uOld=inp["uOld"], **{k: np.array(v) for k, v in rec.items()})
    print("wrote", OUT)


def run(n=1024, N=40, exact_nu=True, out="profiles/r5_n40_model.json"):
    from oracle import lmpc_oracle as orc
    from tests import ipm_model
    d = np.load(OUT)
    p = orc.QPParams.lmpc_default(N)
    its, traces = [], []
    for i in range(min(n, d["x0"].shape[0])):
        qp = ipm_model.StructQP(p, d["A"][i], d["B"][i], d["C"][i], d["x0"][i], d["uOld"][i], d["SS"][i], d["Qsel"][i])
        tr = []
        with np.errstate(all="ignore"):
            r = ipm_model.ipm_solve(qp, exact_nu=exact_nu, trace=tr)
        its.append(int(r["iters"])); traces.append(tr)
    its = np.array(its)
    print("N=%d n=%d exact_nu=%s: mean %.3f max %d hist(from 5) %s" % (N, len(its), exact_nu, its.mean(), its.max(), np.bincount(its)[5:].tolist()))
    with open(os.path.join(ROOT, out), "w") as f:
        json.dump(dict(N=N, n=len(its), exact_nu=exact_nu, mean=float(its.mean()), max=int(its.max()), hist=np.bincount(its).tolist(), iters=its.tolist(),
                       trace_columns=["gap", "r_d", "r_e", "sigma", "alpha_p", "alpha_d"], traces=traces[:64]), f)
    return its, traces


def compare(gpu_npz, N=40):
    """gpu_npz: iters (B), trace (B, maxit + 1, 6) written by tools/n40_trace.py from the kernel's LMPC_TRACE side channel."""
    g = np.load(gpu_npz)
    m = json.load(open(os.path.join(ROOT, "profiles/r5_n40_model.json")))
    gi, mi = np.asarray(g["iters"]), np.asarray(m["iters"])
    n = min(len(gi), len(mi))
    print("GPU  : mean %.3f max %d hist(from 5) %s" % (gi[:n].mean(), gi[:n].max(), np.bincount(gi[:n])[5:].tolist()))
    print("model: mean %.3f max %d hist(from 5) %s" % (mi[:n].mean(), mi[:n].max(), np.bincount(mi[:n])[5:].tolist()))
    print("identical iteration counts: %d of %d; GPU - model: %s" % ((gi[:n] == mi[:n]).sum(), n, dict(zip(*np.unique(gi[:n] - mi[:n], return_counts=True)))))
    names = m["trace_columns"]
    first = {}
    for b in range(min(n, len(m["traces"]))):
        tm = np.array(m["traces"][b], float); tg = np.asarray(g["trace"][b])[:len(tm)]
        for it in range(min(len(tm), len(tg))):
            rel = np.abs(tg[it] - tm[it]) / (np.abs(tm[it]) + 1e-300)
            badc = [names[c] for c in range(6) if np.isfinite(tm[it][c]) and rel[c] > 1e-3 and abs(tg[it][c] - tm[it][c]) > 1e-14]
            if badc:
                first[b] = (it, badc, tg[it].tolist(), tm[it].tolist()); break
    print("problems (of the first %d) whose trace leaves the model's by > 1e-3 relative: %d" % (min(n, len(m["traces"])), len(first)))
    for b, v in list(first.items())[:8]:
        print("  problem %d: iteration %d, %s\n     gpu   %s\n     model %s" % (b, v[0], v[1], v[2], v[3]))


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1024)
    else:
        compare(sys.argv[2])
