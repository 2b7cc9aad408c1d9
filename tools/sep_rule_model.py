"""Developer tool (CPU): the separate / equal step rule of the interior-point iteration on EVERY closed-loop QP of the model's sets (build_tmp/term_sets_N<N>.npz, tools/term_rule_model.py build):
iteration histograms per rule.   python tools/sep_rule_model.py N rule[:param] ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
KEYS = ("A", "B", "C", "x0", "uOld", "SS", "Qsel")

def _one(args):
    N, rec, rule, par = args
    from oracle import lmpc_oracle as orc
    from tests import ipm_model
    ipm_model.SEP_RULE = rule
    if par is not None:
        ipm_model.SEP_STICKY = par
    qp = ipm_model.StructQP(orc.QPParams.lmpc_default(N), *rec)
    with np.errstate(all="ignore"):
        r = ipm_model.ipm_solve(qp, exact_nu=False)
    ok = np.isfinite(r["gap"]) and r["gap"] < 1e-11 and r["rd"] < 1e-9 * max(1.0, np.abs(rec[6]).max()) and r["re"] < 1e-9
    return r["iters"], ok

if __name__ == "__main__":
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)
    N = int(sys.argv[1])
    d = {k: v for k, v in np.load(os.path.join(ROOT, "build_tmp", "term_sets_N%d.npz" % N)).items()}
    names = sorted({k.rsplit("_", 1)[0] for k in d if k.endswith("_x0")})
    with mp.get_context("fork").Pool(8) as pool:
        for spec in sys.argv[2:]:
            rule, _, par = spec.partition(":")
            for name in names:
                n = d[name + "_x0"].shape[0]
                res = pool.map(_one, [(N, tuple(d["%s_%s" % (name, k)][i] for k in KEYS), rule, float(par) if par else None) for i in range(n)], chunksize=16)
                its = np.array([r[0] for r in res]); bad = sum(not r[1] for r in res)
                print("%-14s %-6s n=%5d iterations %.3f / %2d  n>16: %d  n>20: %d  not converged: %d  hist(from 6) %s" % (spec, name, n, its.mean(), its.max(), (its > 16).sum(), (its > 20).sum(), bad, np.bincount(its)[6:].tolist()), flush=True)
