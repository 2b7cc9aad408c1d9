"""-m gpu: parity of the HIP path (through the C ABI) against the reference-executed golden fixtures and the
certified optimum.  Needs a real MI355X; nothing here reads /root/reference."""
import os

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g(built):
    return common.load_lmpc_golden()


def test_regression_matches_reference(g):
    """K1 vs the reference's own regressionAndLinearization output (rec_A/B/C were produced by the reference classes)."""
    worst = 0.0
    for lap in (4, 5):
        ctx, par = common.make_lmpc_ctx(g, lap, max_batch=64)
        idx = np.where(g["rec_lap"] == lap)[0]
        A, B, C, st = ctx.regress_batch(g["rec_xLin"][idx], g["rec_uLin"][idx])
        assert np.all(st == 0), st
        for got, ref in ((A, g["rec_A"][idx]), (B, g["rec_B"][idx]), (C, g["rec_C"][idx])):
            err = np.abs(got - ref) / (1.0 + np.abs(ref))
            worst = max(worst, err.max())
        # structural zeros of A_i, B_i must be exact
        assert np.all(A[:, :, 0:3, 3:6] == 0) and np.all(B[:, :, 0, 0] == 0) and np.all(B[:, :, 3:6, :] == 0)
        ctx.close()
    print("regression worst rel err", worst)
    assert worst < common.TOL_ABC


def test_regress_points_is_the_batch_regression_point_by_point(g):
    """lmpc_regress_points (the reference's own call shape: PredictiveModel.regressionAndLinearization(x, u) for ONE point, no horizon around it) returns, for every
    horizon point of the recorded steps, bit for bit what the batched regression returns for it; the drop-in PredictiveModel goes through it."""
    from racinglmpc_amd.PredictiveModel import PredictiveModel
    from tests import closed_loop
    ctx, par = common.make_lmpc_ctx(g, 5, max_batch=64)
    idx = np.where(g["rec_lap"] == 5)[0][:16]
    A, B, C, st = ctx.regress_batch(g["rec_xLin"][idx], g["rec_uLin"][idx])
    xq = g["rec_xLin"][idx][:, :12].reshape(-1, 6); uq = g["rec_uLin"][idx].reshape(-1, 2)
    Ap, Bp, Cp, stp = ctx.regress_points(xq, uq)
    assert np.array_equal(Ap, A.reshape(-1, 6, 6)) and np.array_equal(Bp, B.reshape(-1, 6, 2)) and np.array_equal(Cp, C.reshape(-1, 6)) and np.all(stp == 0)
    ctx.close()
    pm = PredictiveModel(6, 2, closed_loop.TrackMap(g), 4)
    for _ in range(4):
        pm.addTrajectory(g["xPID"], g["uPID"])
    Ai, Bi, Ci = pm.regressionAndLinearization(g["xPID"][100], g["uPID"][100])
    from oracle import lmpc_oracle as orc
    Ao, Bo, Co = orc.regression_and_linearization([g["xPID"]] * 4, [g["uPID"]] * 4, [0, 1, 2, 3], np.array(g["track"]), g["xPID"][100], g["uPID"][100])
    for got, ref in ((Ai, Ao), (Bi, Bo), (Ci, Co)):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < common.TOL_ABC


def test_selection_bit_exact(g):
    """K2 vs the reference's selectPoints/addTerminalComponents output, store state replayed step by step."""
    checked = 0
    for lap in (4, 5):
        ctx, par = common.make_lmpc_ctx(g, lap, max_batch=4)

        def on_record(r):
            out = ctx.select_batch(g["rec_x0"][r][None], g["rec_zt"][r][None], g["rec_xPredPrev"][r][None],
                                   np.array([g["rec_hasPred"][r]]), np.array([g["rec_t"][r]]))
            assert out["status"][0] == 0
            assert np.array_equal(out["ssSel"][0], g["rec_SSsel"][r].T)
            assert np.array_equal(out["qSel"][0], g["rec_Qsel"][r])
            assert np.array_equal(out["succ"][0], g["rec_Succ"][r].T)
            assert np.array_equal(out["succU"][0], g["rec_SuccU"][r].T)

        checked += common.replay_lap(g, lap, ctx, on_record)
        ctx.close()
    assert checked == len(g["rec_lap"])


def test_assembly_matches_reference_matrices(g):
    """Explicit OSQP-form matrices built on the GPU equal the reference's H_FTOCP/q_FTOCP/[F;G]/l/u exactly."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=16)
    idx = np.arange(0, len(g["rec_lap"]), 5)
    P, q, A, l, u = ctx.assemble_batch(g["rec_A"][idx], g["rec_B"][idx], g["rec_C"][idx], g["rec_x0"][idx], g["rec_OldInput"][idx],
                                       np.transpose(g["rec_SSsel"][idx], (0, 2, 1)), g["rec_Qsel"][idx])
    for i, r in enumerate(idx):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r)
        assert np.array_equal(P[i], Pr) and np.array_equal(A[i], Ar)
        assert np.array_equal(q[i], qr) and np.array_equal(l[i], lr) and np.array_equal(u[i], ur)
    ctx.close()


def test_qp_solve_reaches_certified_optimum(g):
    """K3 alone on the reference's own (A,B,C,SS_sel,Qfun_sel): distance to the certified optimum + KKT certificate."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=64)
    R = len(g["rec_lap"])
    out = ctx.qp_solve_batch(g["rec_A"], g["rec_B"], g["rec_C"], g["rec_x0"], g["rec_OldInput"],
                             np.transpose(g["rec_SSsel"], (0, 2, 1)), g["rec_Qsel"])
    assert np.all(out["status"] == 0), out["status"]
    worst, worst_cert = 0.0, 0.0
    for r in range(R):
        opt = g["rec_sol_opt"][r]
        w = np.concatenate([out["xPred"][r].ravel(), out["uPred"][r].ravel(), out["slack"][r], out["lambd"][r], out["sTerm"][r]])
        worst = max(worst, np.abs(w[:102] - opt[:102]).max())
        P, q, A, l, u = common.dense_from_csc(g, r)
        c = common.certificate(P, q, A, l, u, w, out["mu"][r], 144)
        worst_cert = max(worst_cert, max(c.values()))
        # objective value agrees
        obj = 0.5 * w @ P @ w + q @ w; obj_ref = 0.5 * opt @ P @ opt + q @ opt
        assert abs(obj - obj_ref) <= 1e-8 * (1 + abs(obj_ref))
    print("qp: worst |xu - opt| %.2e, worst certificate %.2e, iters mean %.1f max %d" % (worst, worst_cert, out["iters"].mean(), out["iters"].max()))
    assert worst < common.TOL_XU and worst_cert < common.TOL_KKT
    ctx.close()


def test_full_step_matches_reference_path(g):
    """a3 -> a19 fused (lmpc_step_batch) replaying both recorded laps: selection identical, xPred/uPred at the optimum."""
    res = common.run_golden_step_check()
    print(res)
    assert res["n"] == len(g["rec_lap"])
    assert np.all(res["status"] == 0)
    assert res["max_err_sssel"] == 0.0
    assert res["max_err_xu"] < common.TOL_XU
    assert res["max_err_zt"] < common.TOL_ZT         # zt / zt_u of feasibleStateInput from the reference's successor rows and lambda* of the certified optimum, all 60 records


def test_full_step_device_entry_without_diagnostics(g):
    """The same replay with every second recorded step also taken through lmpc_step_batch_dev the way bench.py's timed loop calls it
    (diagnostics=False: mu, resid, qSel NULL): bit-identical outputs."""
    res = common.run_golden_step_check(dev_every=2)
    assert res["n_dev"] >= 25 and np.all(res["status"] == 0) and res["max_err_xu"] < common.TOL_XU


@pytest.mark.parametrize("name", ["lmpc_wide_n12", "lmpc_n14", "lmpc_n40"])
def test_other_configurations_match_reference(built, name):
    """Steps recorded from the EXECUTED reference in two more configurations (tests/golden/make_wide_golden.py): numSS_it = 6, numSS_Points = 72
    -- more terminal-block columns (78) than a wavefront has lanes: the one-wave kernel with several columns per lane at every batch size --
    and main.py's own horizon N = 14.  Selection bit-exact (SS, Q-function with its shift, successor rows), A, B, C within the regression
    tolerance, explicit QP matrices bit-exact, xPred / uPred at the certified optimum of the reference's own QP, zt / zt_u of
    feasibleStateInput."""
    from racinglmpc_amd import _capi
    g = common.load_variant_golden(name)
    N, S, L = int(g["N"]), int(g["numSS_Points"]), int(g["numSS_it"])
    gl = common.load_lmpc_golden()
    cfg, par = common.lmpc_config(gl, N, max_batch=16, numSS_it=L, numSS_Points=S)
    ctx = _capi.Context(cfg)
    assert ctx.S == S and ctx.solver_waves(1) == 4
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"])
    for l in range(L):
        ctx.ss_add_trajectory(g["xPID"], g["uPID"])
        ctx.ss_replace_lap(l, g["SS"][l], g["uSS"][l], g["Qf"][l])
    R = g["x0"].shape[0]
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"], zt=g["zt"], xPredPrev=g["xPredPrev"], hasPred=g["hasPred"].astype(np.int32),
                         timeStep=g["t"].astype(np.int32))
    assert np.all(out["status"] == 0), out["status"]
    assert np.array_equal(out["ssSel"], np.transpose(g["SSsel"], (0, 2, 1))) and np.array_equal(out["qSel"], g["Qsel"])
    for got, ref in ((out["A"], g["A"]), (out["B"], g["B"]), (out["C"], g["C"])):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < common.TOL_ABC
    sel = ctx.select_batch(g["x0"], g["zt"], g["xPredPrev"], g["hasPred"].astype(np.int32), g["t"].astype(np.int32))
    assert np.array_equal(sel["succ"], np.transpose(g["Succ"], (0, 2, 1))) and np.array_equal(sel["succU"], np.transpose(g["SuccU"], (0, 2, 1)))
    P, q, A, l, u = ctx.assemble_batch(g["A"], g["B"], g["C"], g["x0"], g["OldInput"], np.transpose(g["SSsel"], (0, 2, 1)), g["Qsel"])
    worst = worst_zt = 0.0
    nxu = 6 * (N + 1) + 2 * N
    for r in range(R):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
        assert np.array_equal(P[r], Pr) and np.array_equal(A[r], Ar) and np.array_equal(q[r], qr) and np.array_equal(l[r], lr) and np.array_equal(u[r], ur)
        w = np.concatenate([out["xPred"][r].ravel(), out["uPred"][r].ravel()])
        worst = max(worst, np.abs(w - g["sol_opt"][r][:nxu]).max())
        lam = g["sol_opt"][r][nxu + 2 * N:nxu + 2 * N + S]
        worst_zt = max(worst_zt, common.zt_err(out["ztNext"][r], out["ztuNext"][r], g["Succ"][r], g["SuccU"][r], lam))
    # the QP solve alone on the reference's own A, B, C and selection
    out2 = ctx.qp_solve_batch(g["A"], g["B"], g["C"], g["x0"], g["OldInput"], np.transpose(g["SSsel"], (0, 2, 1)), g["Qsel"])
    assert np.all(out2["status"] == 0)
    for r in range(R):
        w = np.concatenate([out2["xPred"][r].ravel(), out2["uPred"][r].ravel(), out2["slack"][r], out2["lambd"][r], out2["sTerm"][r]])
        worst = max(worst, np.abs(w[:nxu] - g["sol_opt"][r][:nxu]).max())
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
        c = common.certificate(Pr, qr, Ar, lr, ur, w, out2["mu"][r], 8 * N + S)
        assert max(c.values()) < common.TOL_KKT
    print("%s (N = %d, %d points from %d laps): worst |xu - certified optimum| %.2e, |zt, zt_u - Succ lambda*| %.2e, IPM iterations max %d" % (name, N, S, L, worst, worst_zt, out["iters"].max()))
    assert worst < common.TOL_XU and worst_zt < common.TOL_ZT
    ctx.close()


@pytest.mark.parametrize("name", ["lmpc_30laps_n12", "lmpc_30laps_stress_n12"])
def test_30_lap_stores_match_reference(built, name):
    """(lmpc_30laps_stress_n12: SURVEY 8(d)'s stress variant, numSS_it = trToUse = 30 and numSS_Points = 360 on the same laps -- a wave of the
    regression kernel scans four laps in turn, the terminal block carries six columns per lane.)
    BASELINE configs[2] pinned by the executed reference (lmpc_30laps_n12.npz): 30 laps of different lengths handed to both stores in the
    reference's order -- the library's sorted insert (regression over the first four) and its choice of the four fastest laps for the safe set
    must reproduce the reference's A, B, C (tolerance), selection (bit-exact), QP matrices (bit-exact) and the certified optimum."""
    from racinglmpc_amd import _capi
    g = common.load_30laps_golden(name)
    gl = common.load_lmpc_golden()
    N = int(g["N"]); nl = int(g["nLaps"]); S = int(g["numSS_Points"])
    cfg, par = common.lmpc_config(gl, N, max_batch=16, max_laps=40, max_lap_len=1024, numSS_it=int(g["numSS_it"]), trToUse=int(g["trToUse"]))
    ctx = _capi.Context(cfg)
    for i in range(nl):
        ctx.model_add_trajectory(g["lapx%d" % i], g["lapu%d" % i]); ctx.ss_add_trajectory(g["lapx%d" % i], g["lapu%d" % i])
        assert np.array_equal(ctx.ss_get_qfun(i), g["Qfun%d" % i])                   # computeCost of every lap
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"], zt=g["zt"], xPredPrev=g["xPredPrev"], hasPred=g["hasPred"].astype(np.int32),
                         timeStep=g["t"].astype(np.int32))
    assert np.all(out["status"] == 0), out["status"]
    assert np.array_equal(out["ssSel"], np.transpose(g["SSsel"], (0, 2, 1))) and np.array_equal(out["qSel"], g["Qsel"])
    for got, ref in ((out["A"], g["A"]), (out["B"], g["B"]), (out["C"], g["C"])):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < common.TOL_ABC
    P, q, A, l, u = ctx.assemble_batch(out["A"], out["B"], out["C"], g["x0"], g["OldInput"], out["ssSel"], out["qSel"])
    worst = worst_zt = 0.0
    nxu = 6 * (N + 1) + 2 * N
    for r in range(g["x0"].shape[0]):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
        assert np.array_equal(P[r], Pr) and np.array_equal(q[r], qr) and np.array_equal(l[r][:8 * N + S], lr[:8 * N + S]) and np.array_equal(u[r][:8 * N + S], ur[:8 * N + S])
        assert np.allclose(A[r], Ar, rtol=0, atol=1e-9) and np.allclose(l[r], lr, rtol=0, atol=1e-9)     # (G and E x0 + L carry this path's own A, B, C)
        w = np.concatenate([out["xPred"][r].ravel(), out["uPred"][r].ravel()])
        worst = max(worst, np.abs(w - g["sol_opt"][r][:nxu]).max())
        lam = g["sol_opt"][r][nxu + 2 * N:nxu + 2 * N + S]
        worst_zt = max(worst_zt, common.zt_err(out["ztNext"][r], out["ztuNext"][r], g["Succ"][r], g["SuccU"][r], lam))
    print("30 laps in both stores: worst |xu - certified optimum| %.2e, |zt, zt_u - Succ lambda*| %.2e, IPM iterations max %d" % (worst, worst_zt, out["iters"].max()))
    assert worst < common.TOL_XU and worst_zt < common.TOL_ZT
    ctx.close()


def test_ltv_mpc_variant(built):
    """No terminal set (MPC class, timeVarying=True, main.py:86-94): regression + QP vs the reference-executed fixture."""
    from racinglmpc_amd import _capi
    g = common.load_ltv_golden()
    cfg, par = common.mpc_config(g, 12, max_batch=16)
    ctx = _capi.Context(cfg)
    ctx.model_add_trajectory(g["xPID"], g["uPID"])
    A, B, C, st = ctx.regress_batch(g["xLin"], g["uLin"])
    assert np.all(st == 0)
    for got, ref in ((A, g["A"]), (B, g["B"]), (C, g["C"])):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < common.TOL_ABC
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"])
    assert np.all(out["status"] == 0), out["status"]
    w = np.concatenate([out["xPred"].reshape(12, -1), out["uPred"].reshape(12, -1), out["slack"]], axis=1)
    err = np.abs(w[:, :102] - g["sol_opt"][:, :102]).max()
    print("ltv-mpc |xu - opt|", err, "iters", out["iters"])
    assert err < common.TOL_XU
    # zt / zt_u of the plain MPC are the last predicted state / input (MPC.feasibleStateInput)
    assert np.array_equal(out["ztNext"], out["xPred"][:, -1, :]) and np.array_equal(out["ztuNext"], out["uPred"][:, -1, :])
    ctx.close()


def test_mpc_n14_matches_reference(built):
    """main.py's stages 2 and 3 at N = 14 against the executed reference (mpc_n14.npz): Utilities.Regression's (A, B) from the HIP kernel, the LTI
    MPC's optimum on them, the LTV MPC's regressions and optimum; explicit matrices of both bit-exact."""
    from racinglmpc_amd import _capi, Utilities
    g = dict(np.load(common.GOLDEN + "/mpc_n14.npz"))
    N = int(g["N"]); nxu = 6 * (N + 1) + 2 * N
    A2, B2, _E = Utilities.Regression(g["xPID"], g["uPID"], 0.0000001)
    assert max((np.abs(A2 - g["A_lti"]) / (1 + np.abs(g["A_lti"]))).max(), (np.abs(B2 - g["B_lti"]) / (1 + np.abs(g["B_lti"]))).max()) < 1e-7
    cfg, par = common.mpc_config(g, N, max_batch=16)
    ctx = _capi.Context(cfg)
    ctx.model_add_trajectory(g["xPID"], g["uPID"])
    R = g["lti_x0"].shape[0]
    At = np.tile(g["A_lti"][None, None], (R, N, 1, 1)); Bt = np.tile(g["B_lti"][None, None], (R, N, 1, 1)); Ct = np.zeros((R, N, 6))
    out = ctx.qp_solve_batch(At, Bt, Ct, g["lti_x0"], g["lti_OldInput"])
    assert np.all(out["status"] == 0)
    w = np.concatenate([out["xPred"].reshape(R, -1), out["uPred"].reshape(R, -1)], axis=1)
    e_lti = np.abs(w - g["lti_sol_opt"][:, :nxu]).max()
    P, q, A, l, u = ctx.assemble_batch(At, Bt, Ct, g["lti_x0"], g["lti_OldInput"])
    for r in range(R):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="lti_")
        assert np.array_equal(P[r], Pr) and np.array_equal(A[r], Ar) and np.array_equal(q[r], qr) and np.array_equal(l[r], lr) and np.array_equal(u[r], ur)
    out = ctx.step_batch(g["ltv_x0"], g["ltv_xLin"], g["ltv_uLin"], g["ltv_OldInput"])
    assert np.all(out["status"] == 0)
    for got, ref in ((out["A"], g["ltv_A"]), (out["B"], g["ltv_B"]), (out["C"], g["ltv_C"])):
        assert (np.abs(got - ref) / (1 + np.abs(ref))).max() < common.TOL_ABC
    w = np.concatenate([out["xPred"].reshape(R, -1), out["uPred"].reshape(R, -1)], axis=1)
    e_ltv = np.abs(w - g["ltv_sol_opt"][:, :nxu]).max()
    P, q, A, l, u = ctx.assemble_batch(g["ltv_A"], g["ltv_B"], g["ltv_C"], g["ltv_x0"], g["ltv_OldInput"])
    for r in range(R):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="ltv_")
        assert np.array_equal(P[r], Pr) and np.array_equal(A[r], Ar) and np.array_equal(q[r], qr) and np.array_equal(l[r], lr) and np.array_equal(u[r], ur)
    print("N = 14: LTI MPC |xu - opt| %.2e, LTV MPC |xu - opt| %.2e" % (e_lti, e_ltv))
    assert e_lti < common.TOL_XU and e_ltv < common.TOL_XU
    ctx.close()


def test_mpc_hard_lane_constraints(built):
    """MPCParams(slacks=False) (PredictiveControllers.py:184-198, 218-221, 248-254): no slack variables, hard lane rows.  Fixture recorded from the
    executed reference class (tests/golden/make_noslack_golden.py; lane half-width 1 cm so that up to five hard rows are active):
    explicit matrices bit-exact, solution at the certified optimum of the reference's own QP, multipliers certify it, every kernel variant.
    (The solve kernels impose the hard rows through slack variables with a quadratic weight of 1e12, see fill_params in lmpc_capi.hip.)"""
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd import _capi
    g = np.load(os.path.join(common.GOLDEN, "ltvmpc_noslack_n12.npz"))
    N, n = 12, len(g["x0"])
    cfg, par = common.mpc_config(g, N, max_batch=1200, bx=float(g["bx"]), slacks=False)
    ctx = _capi.Context(cfg)
    assert ctx.qp_dims() == (6 * (N + 1) + 2 * N, 6 * N, 6 * (N + 1))
    ctx.model_add_trajectory(g["xPID"], g["uPID"])
    # a10-a12 without slacks: the explicit matrices are the reference's, bit for bit
    P, q, Ad, l, u = ctx.assemble_batch(g["A"], g["B"], g["C"], g["x0"], g["OldInput"])
    for r in range(n):
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
        assert np.array_equal(P[r], Pr) and np.array_equal(q[r], qr) and np.array_equal(Ad[r], Ar) and np.array_equal(l[r], lr) and np.array_equal(u[r], ur)
    worst = wcert = 0.0
    for reps in (1, 40, 80):                                     # 14 / 560 / 1120 problems: four, two and one wave(s) per QP
        rep = lambda a: np.tile(a, (reps,) + (1,) * (a.ndim - 1))
        out = ctx.step_batch(rep(g["x0"]), rep(g["xLin"]), rep(g["uLin"]), rep(g["OldInput"]))
        assert np.all(out["status"] == 0), (reps, np.unique(out["status"], return_counts=True))
        assert np.abs(out["slack"]).max() < 1e-9                                       # excess over a hard row: mu / 2e12
        for b in range(n * reps):
            r = b % n
            w = np.concatenate([out["xPred"][b].ravel(), out["uPred"][b].ravel()])
            worst = max(worst, np.abs(w - g["sol_opt"][r]).max())
            if b < n:
                Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
                wcert = max(wcert, max(common.certificate(Pr, qr, Ar, lr, ur, w, out["mu"][b][:6 * N], 6 * N).values()))
        assert len({ctx.solver_waves(n * k) for k in (1, 40, 80)}) == 3
    nact = int(np.sum(np.abs((Ad[:, :2 * N] @ g["sol_opt"][..., None])[..., 0] - u[:, :2 * N]) < 1e-7))
    print("slacks=False: worst |xu - opt| %.2e, certificate %.2e, active hard lane rows in the fixture %d" % (worst, wcert, nact))
    assert worst < common.TOL_XU and wcert < common.TOL_KKT and nact >= 20
    ctx.close()


def test_status_flags(g):
    """Error semantics: the reference raises (IndexError / int(np.where) / cvxopt) -- the batch API flags per problem."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=4)
    xLin = g["rec_xLin"][:2].copy(); uLin = g["rec_uLin"][:2].copy()
    xLin[1, 3, 4] = -1.0                       # s < 0: on no track segment
    A, B, C, st = ctx.regress_batch(xLin, uLin)
    assert st[0].max() == 0 and st[1, 3] & _flag("ST_NO_SEGMENT") and st[1, 2] == 0
    zt = g["rec_zt"][:1].copy(); zt[0, 4] = g["SS0"][-1, 4]; zt[0, :4] = g["SS0"][-1, :4]; zt[0, 5] = g["SS0"][-1, 5]
    out = ctx.select_batch(g["rec_x0"][:1] * 0 + zt, zt)          # nearest point = last row of the lap: window overruns
    assert out["status"][0] & _flag("ST_WINDOW")
    ctx.close()


def test_more_status_flags(g):
    """Singular local regression (reference: cvxopt raises), iteration limit, non-interior start, unsupported shapes."""
    from racinglmpc_amd import _capi
    # bandwidth so small that no stored row lies within h -> fewer than 5 neighbours -> singular normal matrix
    cfg, par = common.lmpc_config(g, 12, max_batch=4)
    cfg.h = 1e-6
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"])
    A, B, C, st = ctx.regress_batch(g["rec_xLin"][:1], g["rec_uLin"][:1])
    assert np.all(st & _flag("ST_REG_SINGULAR")) and np.all(np.isfinite(A))
    ctx.close()
    # iteration limit
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=4, max_iter=3)
    out = ctx.qp_solve_batch(g["rec_A"][:2], g["rec_B"][:2], g["rec_C"][:2], g["rec_x0"][:2], g["rec_OldInput"][:2],
                             np.transpose(g["rec_SSsel"][:2], (0, 2, 1)), g["rec_Qsel"][:2])
    assert np.all(out["status"] == _flag("ST_MAXITER")) and np.all(out["iters"] == 3) and np.all(np.isfinite(out["xPred"]))
    ctx.close()
    # u = 0 not strictly inside the input box
    cfg, par = common.lmpc_config(g, 12, max_batch=4)
    cfg.bu[0] = 0.0
    ctx = _capi.Context(cfg)
    out = ctx.qp_solve_batch(g["rec_A"][:1], g["rec_B"][:1], g["rec_C"][:1], g["rec_x0"][:1], g["rec_OldInput"][:1],
                             np.transpose(g["rec_SSsel"][:1], (0, 2, 1)), g["rec_Qsel"][:1])
    assert out["status"][0] & _flag("ST_NOT_INTERIOR")
    ctx.close()
    # unsupported configuration fails loudly at creation, with a message
    cfg, par = common.lmpc_config(g, 12, max_batch=4, numSS_it=4, numSS_Points=388)      # beyond LMPC_MAX_SS_POINTS (and more than 63 points per lap)
    with pytest.raises(_capi.LmpcError, match="argument check failed"):
        _capi.Context(cfg)
    # calls before any lap is stored
    cfg, par = common.lmpc_config(g, 12, max_batch=4)
    ctx = _capi.Context(cfg)
    with pytest.raises(_capi.LmpcError):
        ctx.regress_batch(g["rec_xLin"][:1], g["rec_uLin"][:1])
    with pytest.raises(_capi.LmpcError):
        ctx.ss_add_point(np.zeros(6), np.zeros(2))
    ctx.close()


@pytest.mark.timeout(120)
def test_diverged_states_do_not_hang_the_device(g):
    """The reference wraps s with `while s > TrackLength: s = s - TrackLength` (Track.py:292-310): with s = +inf (a diverged rollout) that
    loop never returns.  On the device it is bounded -- an infinite, NaN or absurd s ends up as LMPC_ST_NO_SEGMENT in every kernel that
    looks the curvature up (regression, plant, global position) and the call returns."""
    from racinglmpc_amd import _capi
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=8)
    bad_s = np.array([np.inf, 1e300, np.nan, -np.inf, 19.0 * 1e6])
    xy, st = ctx.global_position_batch(np.concatenate([bad_s, [3.0, 19.2296 * 3 + 1.0]]), np.zeros(7))
    assert np.all(st[:5] == _capi.ST_NO_SEGMENT) and np.all(st[5:] == 0)
    xl = np.tile(g["rec_xLin"][:1], (5, 1, 1)); ul = np.tile(g["rec_uLin"][:1], (5, 1, 1))
    xl[np.arange(5), 3, 4] = bad_s                                            # one linearisation point per problem off the rails
    A, B, C, rst = ctx.regress_batch(xl, ul)
    assert np.all(rst[:, 3] & _capi.ST_NO_SEGMENT) and np.all(np.delete(rst, 3, axis=1) == 0)
    x = np.tile(g["xPID"][50], (5, 1)); x[:, 4] = bad_s
    xn, xgn, pst = ctx.plant_step_batch(x, x.copy(), np.tile(g["uPID"][50], (5, 1)), np.zeros((5, 3)))
    assert np.all(pst != 0)
    ctx.close()


def _flag(name):
    from racinglmpc_amd import _capi
    return getattr(_capi, name)


def test_device_selftest(g):
    """DPP / v_permlane16_swap / v_permlane32_swap based wave reductions return exact sums / extrema."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=1)
    ctx.selftest()
    ctx.close()


class _Map:
    """What the drop-in classes read from the reference's Map object."""
    def __init__(self, g):
        self.PointAndTangent = np.array(g["track"]); self.TrackLength = float(g["trackLength"]); self.halfWidth = 0.4


def test_dropin_lmpc_closed_loop_matches_restated_reference(g):
    """The drop-in LMPC class (reference constructor/solve/addPoint surface) against the oracle's restatement of the
    reference state machine solved to the certified optimum, in lock-step over a closed loop with the restated plant.
    Covers quirk E-2 (in-place edit of the stored lap at the first solve), zt wrap, xLin/uLin shift, OldInput, timeStep."""
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd.PredictiveControllers import LMPC, MPCParams
    from racinglmpc_amd.PredictiveModel import PredictiveModel
    N = 12
    mp = _Map(g)
    par = orc.QPParams.lmpc_default(N)
    xA, uA = g["xPID"].copy(), g["uPID"].copy()
    xB, uB = g["xPID"].copy(), g["uPID"].copy()
    pm = PredictiveModel(6, 2, mp, 4)
    for _ in range(4):
        pm.addTrajectory(xA, uA)
    params = MPCParams(n=6, d=2, N=N, Q=par.Q, R=par.R, dR=par.dR, Fx=par.Fx, bx=(np.array([[0.4], [0.4]]),), Fu=par.Fu,
                       bu=np.array([[0.5], [0.5], [10.0], [10.0]]), slacks=True, Qslack=par.Qslack, timeVarying=True)
    lmpc = LMPC(48, 4, par.QterminalSlack, params, pm)
    om = orc.OracleModel(np.array(g["track"]), 4)
    for _ in range(4):
        om.addTrajectory(xB, uB)
    ol = orc.OracleLMPC(par, om, exact=True)
    for _ in range(4):
        lmpc.addTrajectory(xA, uA, g["xPID_glob"]); ol.addTrajectory(xB, uB)
    rs = np.random.RandomState(5)
    x = np.array([0.5, 0, 0, 0, 0, 0.0]); xg = x.copy()
    worst_u, worst_x, worst_zt = 0.0, 0.0, 0.0
    for t in range(20):
        lmpc.solve(x); ol.solve(x)
        assert lmpc.feasible == 1
        worst_u = max(worst_u, np.abs(lmpc.uPred - ol.uPred).max()); worst_x = max(worst_x, np.abs(lmpc.xPred - ol.xPred).max())
        worst_zt = max(worst_zt, np.abs(lmpc.zt - ol.zt).max(), np.abs(lmpc.zt_u - ol.zt_u).max())
        assert lmpc.uPred.shape == (N, 2) and lmpc.xPred.shape == (N + 1, 6) and lmpc.SS_PointSelectedTot.shape == (6, 48)
        if t == 0:
            assert xA[5, 5] == xB[5, 5] and xA[5, 5] < -19.0            # quirk E-2 reproduced on both sides
            assert np.array_equal(lmpc.SS_PointSelectedTot, ol.SS_PointSelectedTot)
        u = lmpc.uPred[0, :].copy()
        lmpc.addPoint(x, u); ol.addPoint(x, u)
        x, xg = orc.dyn_model(om.pt, x, xg, u, rs.randn)
    print("drop-in closed loop: worst |du| %.2e |dx| %.2e |dzt| %.2e" % (worst_u, worst_x, worst_zt))
    assert worst_u < 1e-5 and worst_x < 1e-5 and worst_zt < 1e-5
    assert lmpc.timeStep == 20 and len(lmpc.xStoredPredTraj_it) == 20 and lmpc.SS[3].shape[0] == 1020
    assert np.array_equal(lmpc.Qfun[3], ol.Qfun[3])


def test_dropin_mpc_variants(built):
    """MPC class: LTI (A, B given) and LTV (timeVarying) variants run and agree with the fixture's certified optimum."""
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd.PredictiveControllers import MPC, MPCParams
    from racinglmpc_amd.PredictiveModel import PredictiveModel
    gl = common.load_ltv_golden()
    N = 12
    mp = _Map(gl)
    par = orc.QPParams.mpc_default(N, 0.8)
    pm = PredictiveModel(6, 2, mp, 1)
    pm.addTrajectory(gl["xPID"].copy(), gl["uPID"].copy())
    params = MPCParams(n=6, d=2, N=N, Q=par.Q, R=par.R, Fx=par.Fx, bx=(np.array([[2.], [2.]]),), Fu=par.Fu, bu=np.array([[0.5], [0.5], [10.0], [10.0]]),
                       xRef=par.xRef, slacks=True, Qslack=par.Qslack, timeVarying=True)
    mpc = MPC(params, pm)
    assert np.allclose(np.array(mpc.A), gl["A"][0], atol=1e-9)          # MPC.__init__ runs computeLTVdynamics (:88-91)
    mpc.solve(gl["x0"][0])
    w = np.concatenate([mpc.xPred.ravel(), mpc.uPred.ravel()])
    assert np.abs(w - gl["sol_opt"][0][:102]).max() < common.TOL_XU
    assert np.array_equal(mpc.zt, mpc.xPred[-1]) and np.array_equal(mpc.OldInput, mpc.uPred[0])
    # LTI: A, B from the fixture's first stage, no model
    p2 = MPCParams(n=6, d=2, N=N, A=gl["A"][0][0], B=gl["B"][0][0], Q=par.Q, R=par.R, Fx=par.Fx, bx=(np.array([[2.], [2.]]),), Fu=par.Fu,
                   bu=np.array([[0.5], [0.5], [10.0], [10.0]]), xRef=par.xRef, slacks=True, Qslack=par.Qslack)
    # slacks=False (hard lane rows, here 1 cm wide): first closed-loop step of the reference-executed fixture
    gn = np.load(os.path.join(common.GOLDEN, "ltvmpc_noslack_n12.npz"))
    bxn = float(gn["bx"])
    p3 = MPCParams(n=6, d=2, N=N, Q=par.Q, R=par.R, Fx=par.Fx, bx=(np.array([[bxn], [bxn]]),), Fu=par.Fu, bu=np.array([[0.5], [0.5], [10.0], [10.0]]),
                   xRef=par.xRef, slacks=False, Qslack=par.Qslack, timeVarying=True)
    pm3 = PredictiveModel(6, 2, mp, 1)
    pm3.addTrajectory(gn["xPID"].copy(), gn["uPID"].copy())
    hard = MPC(p3, pm3)
    hard.solve(gn["x0"][0])
    assert hard.feasible == 1 and hard.Solution.shape == (6 * (N + 1) + 2 * N,)                 # z = [x, u]: no slack entries (:218-221)
    assert np.abs(hard.Solution - gn["sol_opt"][0]).max() < common.TOL_XU
    Ph, qh, Ah, lh, uh = hard.qp_matrices()
    Pr, qr, Ar, lr, ur = common.dense_from_csc(gn, 0, prefix="")
    assert Ph.shape == Pr.shape and Ah.shape == Ar.shape and np.array_equal(Ph, Pr) and np.array_equal(qh, qr)
    assert np.array_equal(Ah[:6 * N], Ar[:6 * N]) and np.array_equal(uh[:6 * N], ur[:6 * N])     # F, b: exact; G carries the regression's A_i, B_i (1e-9 parity)
    assert np.allclose(Ah, Ar, rtol=0, atol=1e-8) and np.allclose(uh, ur, rtol=0, atol=1e-8)
    # ... and an INFEASIBLE hard problem: x0 outside the 1 cm lane (its k = 0 row is fixed by x_0 = x0).  The reference's OSQP reports primal
    # infeasible and MPC.solve sets feasible = 0 (:277-280); the penalised problem the kernels solve has an optimum, so the library says so itself
    x_out = gn["x0"][0].copy(); x_out[5] = 5 * bxn
    hard2 = MPC(p3, pm3)
    hard2.solve(x_out)
    assert hard2.feasible == 0 and (int(hard2._out["status"][0]) & _flag("ST_INFEASIBLE"))
    assert hard2.xPred.shape == (N + 1, 6) and np.all(np.isfinite(hard2.xPred))                  # the penalised optimum is still returned
    lti = MPC(p2)
    lti.solve(gl["x0"][0])
    P, q, A, l, u = orc.assemble_mpc_qp(par, gl["A"][0][0], gl["B"][0][0], None, gl["x0"][0], np.zeros(2))
    ex, cert = orc.osqp_solve_exact(P, q, A, l, u)
    assert np.abs(np.concatenate([lti.xPred.ravel(), lti.uPred.ravel()]) - ex.x[:102]).max() < common.TOL_XU


def test_batched_rollouts_full_lap_and_exchange(g):
    """B closed-loop LMPC cars against the 4x PID safe set complete a lap; the per-lap exchange (world = 1 here) feeds the
    K fastest laps back into both stores and the next lap is no slower than the PID seed lap."""
    from racinglmpc_amd import rollout
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=16)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=3)
    B = 16
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.05, 0.05, B)
    xLin0 = g["SS0"][1:14]; uLin0 = g["uSS0"][1:13]
    best = rollout.lap_and_exchange(ro, x0, xLin0, uLin0, K=4, T_max=400)
    assert len(best) == 4
    Ts = [b[4] for b in best]
    print("rollout lap lengths (best 4 of %d):" % B, Ts)
    assert all(100 < T < 400 for T in Ts)
    for x, u, xg, src, T, extra in best:
        assert x[-1, 4] <= float(g["trackLength"]) + 1.0 and np.abs(x[:, 5]).max() < 0.6        # stayed on (soft-constrained) track
        assert np.abs(u[:, 0]).max() <= 0.5 + 1e-9 and np.abs(u[:, 1]).max() <= 10 + 1e-9       # hard input bounds respected
    ctx.close()


def test_plant_kernel_matches_reference_plant(g):
    """K4 (lmpc_plant_kernel) vs the oracle's restatement of Simulator.dynModel (which is bit-exact against the
    reference, tests/test_oracle_golden.py): only the transcendental functions differ (device libm vs NumPy)."""
    from oracle import lmpc_oracle as orc
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=4)
    rng = np.random.default_rng(11)
    idx = rng.integers(0, 990, size=64)
    x = g["xPID"][idx].copy(); xg = g["xPID_glob"][idx].copy(); u = g["uPID"][idx].copy()
    x[:, 4] = np.mod(x[:, 4], float(g["trackLength"])) + (idx % 3) * float(g["trackLength"]) * (idx % 2)     # some beyond one lap
    nz = rng.standard_normal((64, 3)) * 3.0                                                               # exercises the clipping
    xn, xgn, st = ctx.plant_step_batch(x, xg, u, nz)
    assert np.all(st == 0)
    pt = np.array(g["track"])
    worst = 0.0
    for b in range(64):
        it = iter(nz[b])
        xo, go = orc.dyn_model(pt, x[b], xg[b], u[b], lambda: next(it))
        worst = max(worst, np.abs(xn[b] - xo).max(), np.abs(xgn[b] - go).max())
    print("plant kernel worst abs err", worst)
    assert worst < 1e-12
    ctx.close()


def test_device_resident_rollouts(g):
    """lmpc_rollout_lap (regression + solve + plant + bookkeeping on the GPU for a whole lap) reproduces the host-driven
    loop (same controller calls, NumPy plant) when fed the same noise."""
    from racinglmpc_amd import rollout
    B = 8
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.04, 0.04, B)
    xLin0 = g["SS0"][1:14]; uLin0 = g["uSS0"][1:13]
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=B)
    ro_d = rollout.BatchedRollouts(ctx, g["track"], seed=9)
    laps_d = ro_d.run_lap_device(x0, xLin0, uLin0, max_steps=320)
    assert np.all(ro_d.last_status == 0)
    from tests import host_rollout
    noise = np.random.default_rng(9).standard_normal((320, B, 3))       # the same draws run_lap_device consumed

    class _Replay:
        def __init__(self, z): self.z, self.t = z, 0
        def standard_normal(self, shape):
            out = self.z[self.t]; self.t += 1; return out
    import unittest.mock as um
    with um.patch.object(np.random, "default_rng", lambda seed=None: _Replay(noise)):
        laps_h = host_rollout.run_lap_host(ctx, g["track"], x0, xLin0, uLin0, max_steps=320)
    assert len(laps_d) == len(laps_h) == B
    for (xd, ud, gd, fin, done, st), (xh, uh, gh) in zip(laps_d, laps_h):
        assert xd.shape == xh.shape and 150 < xd.shape[0] < 320
        # (the two loops differ in the plant's cos / sin only -- device libm against NumPy, 1e-16 -- and every QP is solved to a 1e-9 certificate, so
        #  the closed loop amplifies last-bit input differences: typically 1e-9 over a lap, 2e-6 on one of these eight since the regression's
        #  5 x 5 solves use the reciprocal square root)
        assert np.abs(xd - xh).max() < 2e-5 and np.abs(ud - uh).max() < 2e-5
    print("device rollouts: lap lengths", [l[0].shape[0] for l in laps_d])
    ctx.close()


def test_rollout_exchange_records_and_validity(g):
    """lmpc_rollout_exchange (device-packed records + ncclAllGather when a communicator exists, a copy otherwise): the bytes equal
    parallel.pack_laps on the same laps fetched to the host; rollouts that did not finish or carry a status bit are never offered."""
    from racinglmpc_amd import parallel, rollout, _capi
    B, K, T_max = 12, 4, 330
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.04, 0.04, B)
    xl = np.tile(g["SS0"][1:14][None], (B, 1, 1)); ul = np.tile(g["uSS0"][1:13][None], (B, 1, 1))
    xl[5, 2, 4] = -3.0                                                  # rollout 5: off-track linearisation point at step 0 -> flagged
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=B)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=4)
    ro.begin(x0, xl, ul, None, T_max)
    ctx.rollout_run(T_max)
    recs, lens, n_valid = ctx.rollout_exchange(K, T_max)
    X, U, G, done, st, fx, fg = ctx.rollout_fetch(0, ctx._ro_t)
    ctx.rollout_end()
    assert st[5] & _capi.ST_NO_SEGMENT and n_valid == B - 1
    laps = [(X[:done[b], b], U[:done[b], b], G[:done[b], b], np.concatenate([fx[b], fg[b]])) for b in range(B)]
    valid = [b for b in range(B) if done[b] >= 0 and (st[b] & ~_capi.ST_INEXACT) == 0]
    rec_h, len_h = parallel.pack_laps([laps[b] for b in valid], K, T_max)
    rec_h[:, T_max, 12] = [valid[int(i)] for i in rec_h[:, T_max, 12]]     # pack_laps numbers the laps it was given; the device the rollouts
    assert recs.shape == (1, K, T_max + 1, 14) and np.array_equal(lens[0], len_h)
    assert np.array_equal(recs[0], rec_h)
    assert 5 not in recs[0, :, T_max, 12]
    ctx.close()


def test_rccl_communicator_single_rank(g):
    """The RCCL branch with one rank on the one GPU of the test box: ncclGetUniqueId / ncclCommInitRank / ncclAllGather /
    ncclAllReduce through the C ABI give what the communicator-free path gives (the N-rank run is the driver's; its record
    layout and top-K logic are covered by the world-size-2 CPU test)."""
    from racinglmpc_amd import parallel, rollout
    B, K, T_max = 8, 3, 330
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=B)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.04, 0.04, B)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=4)
    ro.begin(x0, g["SS0"][1:14], g["uSS0"][1:13], None, T_max); ctx.rollout_run(T_max)
    ref_rec, ref_len, nv = ctx.rollout_exchange(K, T_max)
    comm = parallel.RcclComm(ctx, 0, 1)
    assert ctx.comm_info() == (0, 1, True)
    a = np.arange(24.0).reshape(2, 3, 4)
    assert np.array_equal(comm.allgather(a), a[None]) and comm.allgather(np.arange(5, dtype=np.int64)).dtype == np.int64
    assert comm.allreduce_max([1.5, -2.0]).tolist() == [1.5, -2.0]
    comm.barrier()
    rec, ln, nv2 = ctx.rollout_exchange(K, T_max)
    assert nv2 == nv and np.array_equal(rec, ref_rec) and np.array_equal(ln, ref_len)
    ctx.rollout_end()
    comm.close()
    assert ctx.comm_info() == (0, 1, False)
    ctx.close()


def test_lmpc_generations_improve_lap_time(g):
    """Iterated batched LMPC (device-resident laps, fastest laps fed back, stored laps extended past the finish line):
    lap times do not get worse from generation to generation and every rollout finishes its lap."""
    from racinglmpc_amd import rollout
    B = 32
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=B)
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=21)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (B, 1)); x0[:, 5] = np.linspace(-0.05, 0.05, B)
    gen = rollout.LmpcGeneration(ro, B, K=4, T_max=400, ext=40)
    times = []
    for it in range(3):
        best = gen.run(x0, g["SS0"][1:14], g["uSS0"][1:13])
        times.append([b[4] for b in best])
    print("generation lap times (steps):", times)
    assert max(times[0]) < 300 and times[1][0] <= times[0][0] and times[2][0] <= times[1][0] + 2
    assert times[2][0] < times[0][0]
    ctx.close()


def test_global_position_batch(g):
    """lmpc_global_position_batch vs the reference's Map.getGlobalPosition (fixture from the executed reference) and the oracle."""
    import os
    from oracle import lmpc_oracle as orc
    t = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "track_xy.npz"))
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=4)
    xy, st = ctx.global_position_batch(t["s"], t["ey"])
    assert np.all(st == 0)
    assert np.abs(xy - t["xy"]).max() < 1e-12
    xy2, st2 = ctx.global_position_batch(np.array([2 * float(t["trackLength"]), 1.0]), np.array([0.0, 0.1]))
    assert st2[0] == 4 and st2[1] == 0                                     # LMPC_ST_NO_SEGMENT where the reference raises
    assert np.allclose(xy2[1], orc.get_global_position(t["track"], 1.0, 0.1), atol=1e-13)
    ctx.close()


def test_lti_regression_kernel_matches_reference(built):
    """Utilities.Regression (main.py:74-77) as a HIP kernel vs the output of the executed reference function.  The normal matrix
    has cond 1e5..5e7 and the reference forms an explicit inverse, so agreement is bounded by cond * eps: the stated tolerance is
    1e-7 * (1 + |ref|) on A and B, 1e-9 on the residual extrema."""
    from racinglmpc_amd import Utilities
    from tests.test_oracle_golden import _regression_cases
    worst = 0.0
    for name, x, u, lamb, A, B, E, cond in _regression_cases():
        A2, B2, E2 = Utilities.Regression(x, u, lamb)
        assert A2.shape == (6, 6) and B2.shape == (6, 2) and E2.shape == (2, 6)
        err = max((np.abs(A2 - A) / (1 + np.abs(A))).max(), (np.abs(B2 - B) / (1 + np.abs(B))).max())
        worst = max(worst, err)
        assert err < 1e-7, (name, err, cond)
        assert np.abs(E2 - E).max() < 1e-9, (name, np.abs(E2 - E).max())
    print("LTI regression worst rel err", worst)
    with pytest.raises(np.linalg.LinAlgError):
        Utilities.Regression(np.zeros((30, 6)), np.zeros((30, 2)), 0.0)



@pytest.mark.timeout(120)
@pytest.mark.parametrize("runtime_kernel", [False, True])
def test_non_finite_inputs_stay_inside_their_problem(g, runtime_kernel):
    """NaN / inf in one problem's x0, linearisation points or previous input: the launch returns (the iteration is bounded), THAT problem carries a status bit,
    and every other problem of the batch gets, bit for bit, what it gets in a clean batch (the reference would raise inside cvxopt / OSQP for the whole call).
    Round 6: zt = -inf left the selection's arg-min at its sentinel and the window arithmetic overflowed -- a memory access fault on the device; fixed in k2_select."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=16, runtime_kernel=runtime_kernel)
    inp = common.synthetic_inputs(g, 12, 16)
    clean = ctx.step_batch(**inp)
    assert np.all(clean["status"] == 0)
    bad = {k: np.array(v, copy=True) for k, v in inp.items()}
    bad["x0"][3, 0] = np.nan; bad["x0"][7, 5] = np.inf; bad["uOld"][11, 1] = np.nan; bad["xLin"][13, 4, 1] = np.nan; bad["zt"][14, 2] = -np.inf
    out = ctx.step_batch(**bad)
    hit = [3, 7, 11, 13, 14]; rest = [b for b in range(16) if b not in hit]
    print("status of the poisoned problems:", [hex(int(out["status"][b])) for b in hit], "regression status:", [hex(int(np.bitwise_or.reduce(out["rstatus"][b]))) for b in hit] if "rstatus" in out else "")
    assert all(out["status"][b] != 0 for b in hit)
    for k in ("xPred", "uPred", "lambd", "ztNext", "status", "iters"):
        assert np.array_equal(np.asarray(out[k])[rest], np.asarray(clean[k])[rest]), k
    ctx.close()
