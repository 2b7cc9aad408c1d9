"""Shared helpers for the parity tests (and __graft_entry__.smoke)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Stated parity tolerances (see DESIGN.md "Parity").
TOL_ABC = 1e-9        # |A,B,C(gpu) - A,B,C(reference)| <= TOL_ABC * (1 + |ref|): regression normal matrices have cond 1e5..1e8
TOL_XU = 1e-6         # |xPred,uPred(gpu) - certified optimum| (absolute); reference OSQP runs at eps_abs = eps_rel = 1e-3
TOL_ZT = 1e-6         # |zt, zt_u - Succ lambda*, SuccU lambda*| <= TOL_ZT (1 + |ref|), lambda* of the certified optimum (feasibleStateInput, PredictiveControllers.py:382-384).
                      # Relative like xPred / uPred in SURVEY 8(c)-3: zt is linear in lambda with coefficients up to 60 (the s column of Succ_SS over three unwrapped
                      # laps), and two CERTIFIED optima of one QP differ by ~1e-8 in x and ~3e-8 in lambda (measured: restated ADMM + polish against dense_ipm_solve)
TOL_KKT = 1e-7        # solver-independent certificate of the GPU solution


def zt_err(zt, ztu, Succ, SuccU, lam):
    """max over the entries of |zt - Succ lam| / (1 + |Succ lam|) and the same for zt_u: the scaled error TOL_ZT bounds."""
    rz, ru = Succ @ lam, SuccU @ lam
    return float(max((np.abs(zt - rz) / (1 + np.abs(rz))).max(), (np.abs(ztu - ru) / (1 + np.abs(ru))).max()))


def load_lmpc_golden():
    return np.load(os.path.join(GOLDEN, "lmpc_n12.npz"))


def load_variant_golden(name):
    """Reference-executed LMPC steps of another configuration (tests/golden/make_wide_golden.py): "lmpc_wide_n12" (numSS_it = 6, numSS_Points = 72)
    or "lmpc_n14" (main.py's horizon).  The safe set is numSS_it copies of the PID lap, lap 0 as the reference left it (SS0)."""
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    L = int(g["nSS"])
    g["SS"] = [g["SS0"]] + [g["xPID"]] * (L - 1); g["uSS"] = [g["uPID"]] * L; g["Qf"] = [g["Qfun"]] * L
    return g


def load_30laps_golden(name="lmpc_30laps_n12"):
    """Reference-executed fixtures on 30 laps of different lengths (tests/golden/make_wide_golden.py): "lmpc_30laps_n12" (numSS_it = trToUse = 4,
    BASELINE configs[2]) or "lmpc_30laps_stress_n12" (numSS_it = trToUse = 30, numSS_Points = 360: SURVEY 8(d)'s stress variant, which shares the
    laps of the first -- asserted equal when it was generated)."""
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "lapx0" not in g:
        base = np.load(os.path.join(GOLDEN, "lmpc_30laps_n12.npz"))
        for i in range(int(g["nLaps"])):
            for k in ("lapx%d", "lapu%d", "Qfun%d"):
                g[k % i] = base[k % i]
    g.setdefault("trToUse", np.int64(4))
    return g


def load_ltv_golden():
    return np.load(os.path.join(GOLDEN, "ltvmpc_n12.npz"))


def lmpc_config(g, N=12, max_batch=64, numSS_it=4, numSS_Points=None, trToUse=4, **kw):
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd import _capi
    par = orc.QPParams.lmpc_default(N)
    cfg = _capi.config_from(N, par.Q, par.R, par.Qf, par.dR, par.Qslack, par.Fx, par.bx, par.Fu, par.bu, par.xRef,
                            QterminalSlack=par.QterminalSlack, numSS_Points=12 * numSS_it if numSS_Points is None else numSS_Points, numSS_it=numSS_it, trToUse=trToUse,
                            track=g["track"], trackLength=float(g["trackLength"]), max_batch=max_batch, **kw)
    return cfg, par


def mpc_config(g, N=12, max_batch=64, bx=None, **kw):
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd import _capi
    par = orc.QPParams.mpc_default(N, 0.8)
    if bx is not None:
        par.bx = np.array([float(bx), float(bx)])
    par.slacks = bool(kw.get("slacks", True))
    cfg = _capi.config_from(N, par.Q, par.R, par.Qf, par.dR, par.Qslack, par.Fx, par.bx, par.Fu, par.bu, par.xRef,
                            numSS_it=0, trToUse=1, track=g["track"], trackLength=float(g["trackLength"]), max_batch=max_batch, **kw)
    return cfg, par


def dense_from_csc(g, r, prefix="rec_"):
    from scipy import sparse
    Pp, Pi, Px = g[prefix + "Pp"][r], g[prefix + "Pi"][r], g[prefix + "Px"][r]
    Ap, Ai, Ax = g[prefix + "Ap"][r], g[prefix + "Ai"][r], g[prefix + "Ax"][r]
    n = len(Pp) - 1
    m = g[prefix + "l"][r].shape[0]
    P = sparse.csc_matrix((Px[:Pp[-1]], Pi[:Pp[-1]], Pp), shape=(n, n)).toarray()
    A = sparse.csc_matrix((Ax[:Ap[-1]], Ai[:Ap[-1]], Ap), shape=(m, n)).toarray()
    return P, g[prefix + "q"][r], A, g[prefix + "l"][r], g[prefix + "u"][r]


def stores_at_lap(g, lap):
    """Lap stores as they were when LMPC lap `lap` (4 or 5 in the fixture) started.

    Returns (model laps in the reference's addTrajectory call order, safe-set laps [(x, u, qfun-or-None)]).
    The stored PID lap carries reference quirk E-2 (PredictiveControllers.py:394 runs on a VIEW of the stored lap at
    the first LMPC solve: ey of row 5 is decremented by TrackLength in place, before the safe-set selection of that
    same solve); the fixture's SS0 holds that array.  During lap `lap`, addPoint extends safe-set lap `lap - 1`;
    the caller replays those calls."""
    xPID, uPID = g["xPID"], g["uPID"]
    model = [(xPID, uPID)] * 4
    ss = [(g["SS0"], g["uSS0"], None)] * 3
    if lap == 4:
        ss = ss + [(g["SS3"][:1000], g["uSS3"][:1000], None)]
    else:
        model = model + [(g["lapx0"], g["lapu0"])]
        ss = ss + [(g["SS3"], g["uSS3"], g["Qfun3"]), (g["lapx0"], g["lapu0"], None)]
    return model, ss


def make_lmpc_ctx(g, lap, max_batch=64, runtime_kernel=False, **kw):
    from racinglmpc_amd import _capi
    cfg, par = lmpc_config(g, 12, max_batch=max_batch, **kw)
    ctx = _capi.Context(cfg, runtime_kernel=runtime_kernel)
    model, ss = stores_at_lap(g, lap)
    for x, u in model:
        ctx.model_add_trajectory(x, u)
    for i, (x, u, qf) in enumerate(ss):
        if qf is None:
            ctx.ss_add_trajectory(x, u)
        else:   # lap already extended by addPoint: add at its addTrajectory-time length, then install the extended rows
            ctx.ss_add_trajectory(x[:1000], u[:1000])
            ctx.ss_replace_lap(i, x, u, qf)
    return ctx, par


def certificate(P, q, A, l, u, w, mu, mi):
    """KKT certificate of a primal solution w with inequality duals mu (equality duals by least squares)."""
    from oracle import lmpc_oracle as orc
    F, G = A[:mi], A[mi:]
    nu = np.linalg.lstsq(G.T, -(P @ w + q + F.T @ mu), rcond=None)[0]
    return orc.kkt_certificate(P, q, A, l, u, w, np.concatenate([mu, nu]))


def replay_lap(g, lap, ctx, on_record, max_records=None):
    """Walk through the recorded closed-loop steps of LMPC lap `lap`, calling on_record(r) at every recorded step
    (store state identical to the reference's at that step) and lmpc_ss_add_point after each step (SysModel.py:38)."""
    rec_lap, rec_t = g["rec_lap"], g["rec_t"]
    all_lap, all_x0, all_u0 = g["all_lap"], g["all_x0"], g["all_u0"]
    steps = np.where(all_lap == lap)[0]
    recs = {int(rec_t[r]): r for r in range(len(rec_lap)) if rec_lap[r] == lap}
    n = 0
    for t, gi in enumerate(steps):
        if t in recs and (max_records is None or n < max_records):
            on_record(recs[t]); n += 1
        ctx.ss_add_point(all_x0[gi], all_u0[gi])
    return n


def run_golden_step_check(max_records=None, dev_every=0, runtime_kernel=False):
    """Replay the recorded reference laps through the HIP path (lmpc_step_batch, B = 1 per recorded step, addPoint
    in between) and compare with the certified optimum of the reference-assembled QP.  Used by smoke() and tests."""
    g = load_lmpc_golden()
    errs, stats, iters, zerr, ndev, zterr = [], [], [], [], [], []
    left = max_records
    for lap in (4, 5):
        if left is not None and left <= 0:
            break
        ctx, par = make_lmpc_ctx(g, lap, max_batch=4, runtime_kernel=runtime_kernel)

        def on_record(r):
            out = ctx.step_batch(g["rec_x0"][r][None], g["rec_xLin"][r][None], g["rec_uLin"][r][None], g["rec_OldInput"][r][None],
                                 zt=g["rec_zt"][r][None], xPredPrev=g["rec_xPredPrev"][r][None],
                                 hasPred=np.array([g["rec_hasPred"][r]]), timeStep=np.array([g["rec_t"][r]]))
            opt = g["rec_sol_opt"][r]
            ex = np.abs(out["xPred"][0].ravel() - opt[:78]).max()
            eu = np.abs(out["uPred"][0].ravel() - opt[78:102]).max()
            errs.append(max(ex, eu)); stats.append(int(out["status"][0])); iters.append(int(out["iters"][0]))
            zerr.append(np.abs(out["ssSel"][0] - g["rec_SSsel"][r].T).max())
            # feasibleStateInput (:382-384): zt = Succ_SS lambda, zt_u = Succ_uSS lambda with the reference's own successor rows and lambda* of the certified optimum
            lam = opt[126:174]
            zterr.append(zt_err(out["ztNext"][0], out["ztuNext"][0], g["rec_Succ"][r], g["rec_SuccU"][r], lam))
            if dev_every and len(errs) % dev_every == 0:
                # the timed entry point as bench.py calls it: lmpc_step_batch_dev with the optional outputs (mu, residual triple, Q-function of the
                # selection) NULL -- every output it does produce must be bit-identical to the host-buffer entry point's
                inp = dict(x0=g["rec_x0"][r][None], xLin=g["rec_xLin"][r][None], uLin=g["rec_uLin"][r][None], uOld=g["rec_OldInput"][r][None],
                           zt=g["rec_zt"][r][None], xPredPrev=g["rec_xPredPrev"][r][None], hasPred=np.array([g["rec_hasPred"][r]]), timeStep=np.array([g["rec_t"][r]]))
                a, keep = ctx.step_dev_buffers(inp, diagnostics=False)
                assert not a.mu and not a.resid and not a.qSel
                ctx.step_batch_dev(1, a)
                dev = ctx.step_dev_fetch(a, 1)
                for k in ("xPred", "uPred", "slack", "lambd", "sTerm", "ztNext", "ztuNext", "ssSel", "A", "B", "C", "status", "iters"):
                    assert np.array_equal(dev[k], out[k]), (r, k)
                for q_ in keep:
                    ctx.dev_free(q_)
                ndev.append(r)

        n = replay_lap(g, lap, ctx, on_record, left)
        if left is not None:
            left -= n
        ctx.close()
    return dict(max_err_xu=float(np.max(errs)), max_err_sssel=float(np.max(zerr)), max_err_zt=float(np.max(zterr)), n=len(errs), status=np.array(stats),
                iters_mean=float(np.mean(iters)), iters_max=int(np.max(iters)), n_dev=len(ndev))


def synthetic_inputs(g, N, B):
    """The synthetic batch of BASELINE configs[4] and of the other-horizon tests (x0 around rows of the PID lap, linearisation = the rows that follow): what
    tests/test_gpu_certificates.py::test_other_horizons_certificate, bench.py's config_N40 and tools/n40_model.py use.  (round 6: lived in tools/n40_model.py.)"""
    xP, uP = np.array(g["xPID"]), np.array(g["uPID"])
    tb = (37 * np.arange(B)) % 900
    rng = np.random.default_rng(1234)
    return dict(x0=xP[tb] + rng.normal(size=(B, 6)) * np.array([.02, .01, .02, .01, 0.0, .02]),
                xLin=np.stack([xP[t + 1:t + N + 2] for t in tb]), uLin=np.stack([uP[t + 1:t + N + 1] for t in tb]),
                uOld=uP[tb].copy(), zt=xP[tb + N + 1].copy(), timeStep=(tb % 300).astype(np.int32))


def zt_face_err(zt, ztu, Succ, SuccU, SS, Qsel, lam_star):
    """zt = Succ lambda, zt_u = SuccU lambda (feasibleStateInput, :382-384) when the QP has MORE THAN ONE optimal lambda (x, u, s_T are unique, lambda need not be: SURVEY
    8(c)-3).  The optimal lambdas are the polytope { lambda >= 0 : [SS; 1'; Qsel'] lambda = [SS; 1'; Qsel'] lambda* } (the rest of the objective is fixed by the unique part
    of the optimum), so every entry of zt has an interval of valid values: 16 small LPs (scipy / HiGHS, tolerances 1e-10) give it.  Returns (err, width): the scaled
    distance of the given zt / zt_u from those intervals, and the widest interval (0 = lambda* is determinate as far as zt is concerned)."""
    from scipy.optimize import linprog
    E = np.vstack([SS, np.ones((1, SS.shape[1])), Qsel[None, :]])
    b = E @ lam_star
    rows = np.vstack([Succ, SuccU]); val = np.concatenate([zt, ztu])
    err = 0.0; width = 0.0
    opt = {"primal_feasibility_tolerance": 1e-10, "dual_feasibility_tolerance": 1e-10}
    for r, v in zip(rows, val):
        lo = linprog(r, A_eq=E, b_eq=b, bounds=(0, None), method="highs", options=opt)
        hi = linprog(-r, A_eq=E, b_eq=b, bounds=(0, None), method="highs", options=opt)
        ref = float(r @ lam_star)
        a = lo.fun if lo.status == 0 else ref; c = -hi.fun if hi.status == 0 else ref
        a, c = min(a, ref), max(c, ref)
        width = max(width, c - a)
        err = max(err, max(a - v, v - c, 0.0) / (1 + abs(v)))
    return float(err), float(width)
