"""Developer tool (GPU box): the reference's 40-lap experiment on the drop-in classes; every `stride`-th closed-loop QP -- real LMPC laps in the safe set,
lane slacks active in the fast laps -- is solved again by the oracle from the kernel's own A, B, C and selection (both restated on the explicit QP of the
reference, oracle.assemble_lmpc_qp) to its certified optimum, and (x, u) compared.     python tools/closed_loop_oracle_probe.py [seed] [stride] [laps] [N]
(tests/test_gpu_closed_loop.py runs probe() on the first laps.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    return min(float((np.abs(r["xu"] - o[:n]) / (1 + np.abs(o[:n]))).max()) for o in opts), float(cert)


def probe(seed=5, stride=10, laps=40, NH=14, fast=False):
    """Returns (records, err, cert, out): the sampled QPs (inputs, the kernel's (x, u), iterations, lap), their scaled distance to the nearer oracle optimum, the
    oracle's certificates, the per-lap records of closed_loop.run_laps."""
    from tests import closed_loop, common
    g = common.load_lmpc_golden()
    flow = closed_loop.DropinFlow(g, NH)
    rec = []; cnt = [0]; state = dict(lap=0)
    inner = flow.solve

    def solve(x):
        u, st, it = inner(x)
        if cnt[0] % stride == 0:
            o = flow.ctrl._out
            rec.append(dict(A=o["A"][0].copy(), B=o["B"][0].copy(), C=o["C"][0].copy(), x0=np.array(x, float), uOld=flow._uOld_before.copy(), SS=np.ascontiguousarray(o["ssSel"][0].T),
                            Qsel=o["qSel"][0].copy(), xu=np.concatenate([o["xPred"][0].ravel(), o["uPred"][0].ravel()]), it=it, lap=state["lap"], st=st))
        cnt[0] += 1
        return u, st, it
    flow.solve = solve

    def on_lap(r):
        state["lap"] += 1
    out = closed_loop.run_laps(flow, g, laps, seed=seed, on_lap=on_lap)
    import multiprocessing as mp
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(1)                     # (inherited by the forked children: one BLAS thread per process -- 64 processes x 256 BLAS threads each took minutes)
    except Exception:                                 # noqa: BLE001
        lim = None
    try:
        with mp.get_context("fork").Pool(max(1, min(64, (os.cpu_count() or 2) - 2, len(rec)))) as pool:          # (children never touch HIP: NumPy only)
            res = pool.map(_work, [(r, NH, fast) for r in rec], chunksize=1)
    finally:
        if lim is not None and hasattr(lim, "restore_original_limits"):
            lim.restore_original_limits()
    return rec, np.array([a for a, _ in res]), np.array([c for _, c in res]), out, cnt[0]


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    laps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    NH = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    rec, err, cert, out, n = probe(seed, stride, laps, NH)
    lap = np.array([r["lap"] for r in rec]); w = int(np.argmax(err))
    print("N = %d, seed %d: %d laps, %d QPs, %d sampled; last lap %d steps" % (NH, seed, len(out), n, len(rec), out[-1]["steps"]))
    print("worst |xu - z*| / (1 + |z*|) %.2e (lap %d, %d iterations), oracle certificates <= %.1e; by lap thirds: %s; n > 5e-7: %d" % (
        err.max(), rec[w]["lap"], rec[w]["it"], cert.max(), [float("%.2e" % err[(lap >= a) & (lap < b)].max()) for a, b in ((0, 13), (13, 27), (27, 99))], int((err > 5e-7).sum())))
    bad = np.argsort(-err)[:6]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "closed_loop_probe_worst_N%d_seed%d.npz" % (NH, seed)), err=err[bad],
                        **{k: np.array([rec[i][k] for i in bad]) for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel", "xu", "it", "lap")})
