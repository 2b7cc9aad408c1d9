"""Test infrastructure: the oracle (oracle/lmpc_oracle.py) over many problems of a batch, one single-threaded process per host core.

The per-problem oracle work of the all-problem parity tests -- regression of N points (~1 ms each), selection, and, where asked, the QP solved
to its certified optimum (osqp_solve_exact: 0.1 s at N = 12, 0.3-1 s at N = 40) -- is minutes of one core for a 1024-problem batch; the GPU
box has a few hundred hardware threads.  Children never touch HIP (fork start method: they inherit the parent's memory, run NumPy, return arrays).
"""
import os

import numpy as np

_JOB = {}


def _work(b):
    from oracle import lmpc_oracle as orc
    j = _JOB
    inp, N, TL, pt = j["inp"], j["N"], j["TL"], j["pt"]
    xS = [l[0] for l in j["laps"]]; uS = [l[1] for l in j["laps"]]
    A, B, C = orc.compute_ltv_dynamics(xS, uS, list(range(len(xS))), pt, inp["xLin"][b], inp["uLin"][b], N)
    Qf = [orc.compute_cost(x, TL) for x in xS]
    z = inp["zt"][b].copy()
    if z[4] - inp["x0"][b][4] > TL / 2:
        z[4] = np.max([z[4] - TL, 0])
    L = len(xS)
    SSsel, Qsel, Succ, SuccU = orc.terminal_components(xS, uS, Qf, [x.shape[0] for x in xS], z, 12 * L, L, None, L, int(inp["timeStep"][b]), N, TL)
    res = dict(b=b, A=A, B=B, C=C, SSsel=SSsel, Qsel=Qsel, Succ=Succ, SuccU=SuccU)
    if b in j["solve"]:
        P, q, Ao, l, u = orc.assemble_lmpc_qp(j["par"], A, B, C, inp["x0"][b], inp["uOld"][b], SSsel, Qsel)
        ex, cert = orc.osqp_solve_exact(P, q, Ao, l, u, want=1e-8)
        res.update(opt=ex.x, cert=cert, obj=float(0.5 * ex.x @ P @ ex.x + q @ ex.x), Pq=(np.asarray(P.todense()) if hasattr(P, "todense") else np.asarray(P), np.asarray(q)))
        # a second certified optimum by the other method (dense interior point on the explicit matrices): where the two disagree on lambda the QP has more
        # than one optimal lambda (x, u are unique, lambda is not: SURVEY 8(c)-3) and zt = Succ lambda is not a function of the QP alone
        r2 = orc.dense_ipm_solve(P, q, Ao, l, u)
        res.update(opt2=r2.x, cert2=max(orc.kkt_certificate(P, q, Ao, l, u, r2.x, r2.y).values()))
    return res


def _objf(P, q):
    return lambda z: float(0.5 * z @ P @ z + q @ z)


def oracle_batch(par, pt, TL, laps, N, inp, idx, solve_idx=(), procs=None):
    """Oracle results for the problems `idx` of `inp` (stores: `laps`, used both as regression store and as safe set, in the library's order):
    list of dicts b, A, B, C, SSsel (6, S), Qsel, Succ, SuccU and -- for b in solve_idx -- opt (the certified optimum z*), cert, obj."""
    import multiprocessing as mp
    try:
        from threadpoolctl import threadpool_limits
        lim = threadpool_limits(1)                     # (inherited by the forked children: one BLAS thread per process)
    except Exception:                                 # noqa: BLE001
        lim = None
    _JOB.clear()
    _JOB.update(par=par, pt=np.asarray(pt), TL=float(TL), laps=[(np.asarray(x), np.asarray(u)) for x, u in laps], N=int(N), inp=inp, solve=set(int(b) for b in solve_idx))
    idx = [int(b) for b in idx]
    n = procs or max(1, min(64, (os.cpu_count() or 2) - 2, len(idx)))
    try:
        if n == 1:
            out = [_work(b) for b in idx]
        else:
            with mp.get_context("fork").Pool(n) as pool:
                out = pool.map_async(_work, idx, chunksize=max(1, len(idx) // (4 * n))).get(timeout=420)     # (a worker that dies leaves map() waiting for ever: fail instead)
        for r in out:
            if "Pq" in r:
                r["objf"] = _objf(*r.pop("Pq"))
        return out
    finally:
        _JOB.clear()
        if lim is not None:
            lim.restore_original_limits() if hasattr(lim, "restore_original_limits") else None
