"""Developer tool (CPU): who is right on the QPs tools/capture_probe_misses.py kept -- the kernel, or the oracle's dense interior-point answer?
For every kept QP: the oracle's dense IPM (what the probe compared with), the oracle's restated ADMM + polish (osqp_solve_exact), and the NumPy model of the kernel at
tolerances 1e-15 / 1e-11; objective values and KKT certificates of all of them and of the kernel's point.      python tools/analyse_probe_misses.py gpurun_out/probe_misses_*.npz"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lmpc_oracle as orc
from tests import ipm_model

for path in sys.argv[1:]:
    d = np.load(path); n = d["err"].shape[0]
    NH = d["A"].shape[1]; par = orc.QPParams.lmpc_default(NH); nxu = d["xu"].shape[1]
    print("==", os.path.basename(path), n, "QPs")
    for i in np.argsort(-d["err"])[:int(os.environ.get("TOP", "12"))]:
        P, q, Ao, l, u = orc.assemble_lmpc_qp(par, d["A"][i], d["B"][i], d["C"][i], d["x0"][i], d["uOld"][i], d["SS"][i], d["Qsel"][i])
        Pd = np.asarray(P.todense()) if hasattr(P, "todense") else np.asarray(P)
        obj = lambda z: float(0.5 * z @ Pd @ z + q @ z)
        r2 = orc.dense_ipm_solve(P, q, Ao, l, u)
        qp = ipm_model.StructQP(par, d["A"][i], d["B"][i], d["C"][i], d["x0"][i], d["uOld"][i], d["SS"][i], d["Qsel"][i])
        with np.errstate(all="ignore"):
            t = ipm_model.ipm_solve(qp, tol_gap=1e-15, tol_res=1e-11, acc_rule=None, exact_nu=False)
            k = ipm_model.ipm_solve(qp, exact_nu=False)
        zt = np.concatenate([t["x"].ravel(), t["u"].ravel()]); zk = np.concatenate([k["x"].ravel(), k["u"].ravel()])
        sc = lambda a, b: float((np.abs(a - b) / (1 + np.abs(b))).max())
        full = lambda r: np.concatenate([r["x"].ravel(), r["u"].ravel(), r["s"].ravel(), r["lam"], r["sT"]])
        msg = "lap %2d it %2d: kernel-vs-denseIPM %.2e | kernel-vs-model(1e-15) %.2e | denseIPM-vs-model(1e-15) %.2e | model(kernel rule, %d it)-vs-model(1e-15) %.2e | obj: denseIPM - tight %.2e" % (
            int(d["lap"][i]), int(d["it"][i]), d["err"][i], sc(d["xu"][i], zt), sc(r2.x[:nxu], zt), k["iters"], sc(zk, zt), obj(r2.x) - obj(full(t)))
        if os.environ.get("ADMM"):
            ex, cert = orc.osqp_solve_exact(P, q, Ao, l, u, want=1e-8)
            msg += " | ADMM-vs-tight %.2e (cert %.1e)" % (sc(ex.x[:nxu], zt), cert)
        print(msg, flush=True)
