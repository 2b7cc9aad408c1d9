"""Developer tool (CPU): ONE termination rule for the interior-point iteration, chosen on traces.

    python tools/term_rule_model.py build [N] [seeds...]   # closed-loop QPs of the reference's 40-lap experiment at horizon N (oracle flow, NumPy model of the kernel as the
                                                           # solver, every QP dumped) + the bench batch -> build_tmp/term_sets_N<N>.npz
    python tools/term_rule_model.py trace [N]              # every QP iterated PAST the kernels' stop (three more iterations), every iterate compared with a solve at
                                                           # tolerances 1e-15 / 1e-11 -> build_tmp/term_traces_N<N>.npz
    python tools/term_rule_model.py rules [N]              # candidate rules evaluated on the traces: iterations (mean / max / histogram) and the worst scaled error at the stop

A rule sees only what the kernels have at the convergence test: gap, gap_prev, r_d, r_d,prev, the length of the last (x, u) step.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = os.path.join(ROOT, "build_tmp")
KEYS = ("A", "B", "C", "x0", "uOld", "SS", "Qsel")


def _laps(args):
    N, seed, exact_nu = args
    from tests import closed_loop, common
    g = common.load_lmpc_golden()
    flow = closed_loop.OracleFlow(g, N, solver="ipm", ipm_kw=dict(exact_nu=exact_nu))
    flow.dump = []
    out = closed_loop.run_laps(flow, g, 40, seed=seed)
    lap = np.concatenate([[r["lap"]] * r["steps"] for r in out if "steps" in r])
    return seed, flow.dump[:len(lap)], lap, [r.get("steps") for r in out]


def build(N, seeds):
    import multiprocessing as mp
    import bench
    from oracle import lmpc_oracle as orc
    from tests import common
    g = common.load_lmpc_golden(); pt = np.array(g["track"]); TL = float(g["trackLength"])
    sets = {}
    with mp.get_context("fork").Pool(len(seeds)) as pool:
        for seed, dump, lap, steps in pool.map(_laps, [(N, s, False) for s in seeds]):
            print("seed %d: %d QPs, laps %s" % (seed, len(dump), steps), flush=True)
            sets["cl%d" % seed] = dump; sets["cl%d_lap" % seed] = lap
    xs, us = [np.array(g["xPID"])] * 4, [np.array(g["uPID"])] * 4
    qf = [orc.compute_cost(xs[0], TL)] * 4
    inp = bench.synth_batch(g, 256, N)
    recs = []
    for b in range(256):
        A, B, C = orc.compute_ltv_dynamics(xs, us, [0, 1, 2, 3], pt, inp["xLin"][b], inp["uLin"][b], N)
        SS, Qs, _, _ = orc.terminal_components(xs, us, qf, [1000] * 4, inp["zt"][b], 48, 4, None, 4, int(inp["timeStep"][b]), N, TL)
        recs.append(dict(A=A, B=B, C=C, x0=inp["x0"][b], uOld=inp["uOld"][b], SS=SS, Qsel=Qs))
    sets["bench"] = recs
    out = {}
    for name, recs in sets.items():
        if name.endswith("_lap"):
            out[name] = recs
        else:
            for k in KEYS:
                out["%s_%s" % (name, k)] = np.array([e[k] for e in recs])
    np.savez_compressed(os.path.join(TMP, "term_sets_N%d.npz" % N), **out)


def _trace(args):
    N, rec, exact_nu = args
    from oracle import lmpc_oracle as orc
    from tests import ipm_model
    p = orc.QPParams.lmpc_default(N)
    qp = ipm_model.StructQP(p, *rec)
    with np.errstate(all="ignore"):
        t = ipm_model.ipm_solve(qp, tol_gap=1e-15, tol_res=1e-11, acc_rule=None, exact_nu=exact_nu)
        zs = np.concatenate([t["x"].ravel(), t["u"].ravel()])
        state = dict(stop=None)

        def rule(s):                                       # the kernels' rule decides where "past the stop" begins; the trace runs three iterations beyond it
            if state["stop"] is None and s["base_ok"] and s["gap_prev"] is not None and (s["gap"] <= 1e-3 * s["gap_prev"] or s["gap"] < 1e-12):
                state["stop"] = s["it"]
            return state["stop"] is not None and s["it"] >= state["stop"] + 3
        snaps = []
        ipm_model.ipm_solve(qp, acc_rule=rule, snaps=snaps, exact_nu=exact_nu)
    rows = []
    for s in snaps:
        z = np.concatenate([s["x"].ravel(), s["u"].ravel()])
        rows.append([s["gap"], s["rd"], s["re"], s["step_prev"], s["lstep_prev"], float((np.abs(z - zs) / (1 + np.abs(zs))).max()), float(np.abs(s["lam"] - t["lam"]).max())])
    qs = max(1.0, float(np.abs(rec[6]).max()))
    return np.array(rows), qs, t["gap"], t["rd"]


def trace(N, exact_nu=False):
    import multiprocessing as mp
    d = {k: v for k, v in np.load(os.path.join(TMP, "term_sets_N%d.npz" % N)).items()}       # (decompressed once)
    names = sorted({k.rsplit("_", 1)[0] for k in d if k.endswith("_x0")})
    out = {}
    from threadpoolctl import threadpool_limits
    threadpool_limits(1)                                   # one BLAS thread per worker (inherited by the forked children)
    with mp.get_context("fork").Pool(8) as pool:
        for name in names:
            n = d[name + "_x0"].shape[0]
            res = pool.map(_trace, [(N, tuple(d["%s_%s" % (name, k)][i] for k in KEYS), exact_nu) for i in range(n)], chunksize=8)
            L = max(r[0].shape[0] for r in res)
            T = np.full((n, L, 7), np.nan)
            for i, r in enumerate(res):
                T[i, :r[0].shape[0]] = r[0]
            out[name] = T; out[name + "_qs"] = np.array([r[1] for r in res]); out[name + "_tight"] = np.array([[r[2], r[3]] for r in res])
            print(name, n, "traced; longest", L, flush=True)
    np.savez_compressed(os.path.join(TMP, "term_traces_N%d%s.npz" % (N, "_exactnu" if exact_nu else "")), **out)


def evaluate(T, qs, rule, tol_gap=1e-11, tol_res=1e-9):
    """First iterate of every trace that meets the base tests and `rule(gap, gap_prev, rd, rd_prev, step_prev, step_pp)`; returns (iterations, error there)."""
    its = np.zeros(T.shape[0], int); err = np.zeros(T.shape[0]); lerr = np.zeros(T.shape[0])
    for i in range(T.shape[0]):
        tr = T[i]; n = int(np.sum(np.isfinite(tr[:, 0])))
        stop = n - 1
        for k in range(1, n):
            gap, rd, re, sp = tr[k, 0], tr[k, 1], tr[k, 2], tr[k, 3]
            if gap < tol_gap and rd < tol_res * qs[i] and re < tol_res and rule(gap, tr[k - 1, 0], rd, tr[k - 1, 1], sp, tr[k - 1, 3] if k > 1 else np.inf):
                stop = k; break
        its[i] = stop; err[i] = tr[stop, 5]; lerr[i] = tr[stop, 6]
    return its, err, lerr


RULES = {
    "gap only (rounds 1-4)": lambda g, gp, rd, rdp, sp, spp: True,
    "r5 N<=12: ratio 1e-3 | floor 0.1": lambda g, gp, rd, rdp, sp, spp: g <= 1e-3 * gp or g < 1e-12,
    "r5 N>12: ratio 1e-4 | floor 0.03, est 1e-6": lambda g, gp, rd, rdp, sp, spp: (g <= 1e-4 * gp or g < 3e-13) and sp * rd <= 1e-6 * rdp,
    "r5 N<=12 + est 1e-6": lambda g, gp, rd, rdp, sp, spp: (g <= 1e-3 * gp or g < 1e-12) and sp * rd <= 1e-6 * rdp,
    "est 1e-6 alone": lambda g, gp, rd, rdp, sp, spp: sp * rd <= 1e-6 * rdp,
    "est 1e-7 alone": lambda g, gp, rd, rdp, sp, spp: sp * rd <= 1e-7 * rdp,
}


def _contr(tol, floor=0.0):
    # a-posteriori bound of a contracting iteration: |z - z*| <= rho / (1 - rho) |last step|, rho = the larger of the two contraction rates the kernel sees --
    # sqrt(gap ratio) (the error of a QP without strict complementarity goes like sqrt(gap)) and the dual residual's ratio (a flat QP: error = |H^-1| r_d)
    def rule(g, gp, rd, rdp, sp, spp):
        rho = max(np.sqrt(g / gp), rd / rdp if rdp > 0 else 0.0, floor)
        return rho < 1.0 and sp * rho <= tol * (1.0 - rho)
    return rule


for tol in (1e-6, 3e-7, 1e-7, 3e-8):
    RULES["contraction bound %.0e" % tol] = _contr(tol)


def _aitken(tol, with_gap=False, with_rd=False):
    # the same bound with the contraction rate measured on the (x, u) steps themselves: rho = |step_k| / |step_k-1|
    def rule(g, gp, rd, rdp, sp, spp):
        rho = sp / spp if spp > 0 else 1.0
        if with_gap:
            rho = max(rho, np.sqrt(g / gp))
        if with_rd:
            rho = max(rho, rd / rdp if rdp > 0 else 0.0)
        return rho < 1.0 and sp * rho <= tol * (1.0 - rho)
    return rule


for tol in (1e-6, 3e-7, 1e-7):
    RULES["step-ratio bound %.0e" % tol] = _aitken(tol)
    RULES["step-ratio|sqrt-gap bound %.0e" % tol] = _aitken(tol, True)
    RULES["step-ratio|rd bound %.0e" % tol] = _aitken(tol, False, True)


def ideal(T, qs, tol, tol_gap=1e-11, tol_res=1e-9):
    its = np.zeros(T.shape[0], int)
    for i in range(T.shape[0]):
        tr = T[i]; n = int(np.sum(np.isfinite(tr[:, 0]))); stop = n - 1
        for k in range(1, n):
            if tr[k, 0] < tol_gap and tr[k, 1] < tol_res * qs[i] and tr[k, 2] < tol_res and tr[k, 5] <= tol:
                stop = k; break
        its[i] = stop
    return its


def rules(N, exact_nu=False):
    d = np.load(os.path.join(TMP, "term_traces_N%d%s.npz" % (N, "_exactnu" if exact_nu else "")))
    names = [k for k in d.files if not k.endswith("_qs") and not k.endswith("_tight")]
    only = [a for a in sys.argv[3:] if a != "exactnu"]
    for tol in (3e-7, 1e-7):
        for name in names:
            its = ideal(d[name], d[name + "_qs"], tol)
            print("%-46s %-6s n=%5d  iterations %.3f / %2d  hist %s" % ("IDEAL (stop at the first iterate within %.0e)" % tol, name, len(its), its.mean(), its.max(), np.bincount(its)[6:].tolist()))
    for rname, rule in RULES.items():
        if only and not any(o in rname for o in only):
            continue
        for name in names:
            its, err, lerr = evaluate(d[name], d[name + "_qs"], rule)
            print("%-46s %-6s n=%5d  iterations %.3f / %2d  hist %s  worst |xu-z*| %.2e  n>3e-7 %d  n>1e-7 %d  worst |lam-lam*| %.2e" % (
                rname, name, len(its), its.mean(), its.max(), np.bincount(its)[6:].tolist(), err.max(), int((err > 3e-7).sum()), int((err > 1e-7).sum()), lerr.max()), flush=True)


if __name__ == "__main__":
    os.makedirs(TMP, exist_ok=True)
    cmd = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    if cmd == "build":
        build(N, [int(a) for a in sys.argv[3:]] or [5, 6, 7])
    elif cmd == "trace":
        trace(N, exact_nu="exactnu" in sys.argv)
    else:
        rules(N, exact_nu="exactnu" in sys.argv)
