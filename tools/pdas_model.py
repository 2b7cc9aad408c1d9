"""Developer tool (CPU, feasibility study): a warm-started primal-dual active-set iteration on CONSECUTIVE closed-loop QPs (model sets of tools/term_rule_model.py).
QP i is solved by the interior-point model; QP i + 1 starts from its solution shifted by one stage, classifies the rows by t < mu and takes active-set Newton steps (the
model's `polish` machinery: weight 1e11 on the rows taken as active, 0 on the others, full steps, rows re-classified after every step) until the point meets the ordinary
tolerances with the right signs -- up to K steps, else the interior-point iteration takes over from its cold start.     python tools/pdas_model.py [N] [set] [count] [K]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import lmpc_oracle as orc
from tests import ipm_model
KEYS = ("A", "B", "C", "x0", "uOld", "SS", "Qsel")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
name = sys.argv[2] if len(sys.argv) > 2 else "cl5"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
K = int(sys.argv[4]) if len(sys.argv) > 4 else 6
d = {k: v for k, v in np.load(os.path.join(ROOT, "build_tmp", "term_sets_N%d.npz" % N)).items()}
lap = d[name + "_lap"]; p = orc.QPParams.lmpc_default(N)
qp = lambda i: ipm_model.StructQP(p, *[d["%s_%s" % (name, k)][i] for k in KEYS])
prev = None; ok = []; steps = []; cold = []; err = []
start_i = int(os.environ.get("START", "0"))
for i in range(start_i, start_i + count):
    q = qp(i)
    with np.errstate(all="ignore"):
        r = ipm_model.ipm_solve(q, exact_nu=False)
    cold.append(r["iters"])
    if prev is not None and lap[i] == lap[i - 1]:
        u0 = np.vstack([prev["u"][1:], prev["u"][-1:]])
        mu = prev["mu"]; ml = mu[:2 * N].reshape(N, 2); mu_ = mu[2 * N:6 * N].reshape(N, 4); ms = mu[6 * N:8 * N].reshape(N, 2)
        sh = lambda a: np.vstack([a[1:], a[-1:]])
        st = dict(u=u0, mu=(sh(ml), sh(mu_), sh(ms)), lam=prev["lam"], slack="tight", s_margin=1e-3, shrink=1.0, mu_scale=1e-6, mu_cap=1e12, lam_floor=1e-9)
        with np.errstate(all="ignore"):
            w = ipm_model.ipm_solve(q, exact_nu=False, start=st, polish=dict(gap=np.inf, pdas=K, retries=0, tol_t=1e-8, tol_m=1e-8), maxit=K + 3)
        good = bool(w.get("pol_ok")) and np.isfinite(w["gap"])
        ok.append(good); steps.append(w.get("nfact", 0))
        if good:
            err.append(max(np.abs(w["x"] - r["x"]).max(), np.abs(w["u"] - r["u"]).max()))
    prev = r
ok = np.array(ok); steps = np.array(steps)
print("N = %d, %s[%d:%d]: %d warm-started QPs; active-set iteration accepted on %d (%.1f %%), factorisations when accepted: mean %.2f, histogram %s; cold interior point: %.2f iterations; worst |xu - cold| when accepted %.1e"
      % (N, name, start_i, start_i + count, len(ok), ok.sum(), 100.0 * ok.mean(), steps[ok].mean() if ok.any() else 0, np.bincount(steps[ok]).tolist() if ok.any() else [], np.mean(cold), max(err) if err else 0))
