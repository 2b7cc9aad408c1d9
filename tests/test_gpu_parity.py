"""-m gpu: parity of the HIP path (through the C ABI) against the reference-executed golden fixtures and the
certified optimum.  Needs a real MI355X; nothing here reads /root/reference."""
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


def _flag(name):
    from racinglmpc_amd import _capi
    return getattr(_capi, name)


def test_device_selftest(g):
    """DPP / v_permlane16_swap / v_permlane32_swap based wave reductions return exact sums / extrema."""
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=1)
    ctx.selftest()
    ctx.close()
