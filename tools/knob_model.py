"""Developer tool (CPU): scalar knobs of the interior-point rules (tests/ipm_model.py: SIG_EXP, FRAC0, FRAC_SIG, SEP_THR, mu_scale) on every closed-loop QP of the model's
sets (build_tmp/term_sets_N<N>.npz).   python tools/knob_model.py N "dict(SIG_EXP=4)" "dict(mu_scale=0.3)" ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
KEYS = ("A", "B", "C", "x0", "uOld", "SS", "Qsel")

def _one(args):
    N, rec, kn = args
    from oracle import lmpc_oracle as orc
    from tests import ipm_model
    start = {}
    ipm_model.SIG_EXP, ipm_model.FRAC0, ipm_model.FRAC_SIG, ipm_model.SEP_THR = (3, 0.995, 1e-3, 0.1) if os.environ.get("BASE_R5") else (5, 0.99, 1e-3, 0.05)      # (pool workers live across specs: start from the values of rounds 1-5 every time; "dict()" = those)
    kwargs = {}
    for k, v in kn.items():
        if k == "mu_scale":
            start["mu_scale"] = v
        elif k in ("so_w", "ncorr", "reg_l", "th_max"):      # keyword arguments of ipm_solve
            kwargs[k] = v
        else:
            setattr(ipm_model, k, v)
    qp = ipm_model.StructQP(orc.QPParams.lmpc_default(N), *rec)
    with np.errstate(all="ignore"):
        r = ipm_model.ipm_solve(qp, exact_nu=bool(os.environ.get("EXACT_NU")), start=start or None, **kwargs)
    ok = np.isfinite(r["gap"]) and r["gap"] < 1e-11 and r["rd"] < 1e-9 * max(1.0, np.abs(rec[6]).max()) and r["re"] < 1e-9
    return r["iters"], ok

if __name__ == "__main__":
    import multiprocessing as mp
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)
    N = int(sys.argv[1])
    d = {k: v for k, v in np.load(os.path.join(ROOT, "build_tmp", "term_sets_N%d.npz" % N)).items()}
    names = [n for n in sorted({k.rsplit("_", 1)[0] for k in d if k.endswith("_x0")}) if os.environ.get("ALL_SETS") or n in ("bench", "cl5")]
    with mp.get_context("fork").Pool(8) as pool:
        for spec in sys.argv[2:]:
            kn = eval(spec)
            for name in names:
                n = d[name + "_x0"].shape[0]
                res = pool.map(_one, [(N, tuple(d["%s_%s" % (name, k)][i] for k in KEYS), kn) for i in range(n)], chunksize=16)
                its = np.array([r[0] for r in res]); bad = sum(not r[1] for r in res)
                print("%-34s %-6s n=%5d iterations %.3f / %2d  n>14: %d  not converged: %d  hist(from 6) %s" % (spec, name, n, its.mean(), its.max(), (its > 14).sum(), bad, np.bincount(its)[6:18].tolist()), flush=True)
