"""-m gpu: the runtime-(N, S) solve kernel (csrc/lmpc_solve_rt.hip.h) -- any horizon without a compiler on the box.

MPCParams.N is a plain parameter in the reference (PredictiveControllers.py:63-107, main.py:43).  The fast solve kernels are templates on (N, numSS_points): a pair
outside the built-in / pre-built set needs hipcc on the box.  Where there is none, lmpc_create_ex(LMPC_CREATE_RUNTIME_KERNEL) serves the pair with a kernel that reads
N and S at run time -- same QP, same interior-point rules, plain FP64 multiply-adds, several times slower.  Here:
  * the recorded reference laps through that kernel (forced): the parity statement of the fast path, unchanged;
  * two pairs nobody built (N = 13; N = 17 with 60 safe-set points) with HIPCC pointing at /bin/false: Context() succeeds, says which kernel it got, and meets the
    oracle's certified optimum at the stated 1e-6;
  * the bench batch through both kernels: same answers.
"""
import os

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


def test_runtime_kernel_on_the_recorded_reference_laps(built):
    res = common.run_golden_step_check(runtime_kernel=True)
    print({k: v for k, v in res.items() if k != "status"})
    assert np.all(res["status"] == 0), res["status"]
    assert res["max_err_sssel"] == 0.0
    assert res["max_err_xu"] < common.TOL_XU and res["max_err_zt"] < common.TOL_ZT
    assert res["n"] == 60


@pytest.mark.parametrize("N,numSS_it", [(13, 4), (17, 5)])
def test_any_horizon_without_a_compiler(built, monkeypatch, N, numSS_it):
    import bench
    from oracle import lmpc_oracle as orc
    from racinglmpc_amd import _capi, build
    from tests import oracle_pool
    S = 12 * numSS_it
    assert (N, S) not in build.BUILTIN and (N, S) not in build.EXTRA_VARIANTS and not os.path.exists(build.variant_path(N, S)), "pick a pair nobody has built"
    monkeypatch.setenv("HIPCC", "/bin/false")
    g = common.load_lmpc_golden(); pt = np.array(g["track"]); TL = float(g["trackLength"])
    B = 48
    cfg, _ = common.lmpc_config(g, N, max_batch=B, numSS_it=numSS_it, trToUse=4)
    with pytest.warns(UserWarning, match="runtime-\\(N, S\\) kernel"):
        ctx = _capi.Context(cfg)
    assert ctx.solver_kind == 2 and not os.path.exists(build.variant_path(N, S))
    pid = (np.array(g["xPID"]), np.array(g["uPID"]))
    for _ in range(numSS_it):
        ctx.ss_add_trajectory(*pid)
    for _ in range(4):
        ctx.model_add_trajectory(*pid)
    inp = bench.synth_batch(g, B, N, seed=77)
    out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    assert np.all(out["status"] == 0), out["status"]
    par = orc.QPParams.lmpc_default(N); par.numSS_Points = S
    idx = list(range(0, B, 2))
    res = oracle_pool.oracle_batch(par, pt, TL, [pid] * numSS_it, N, inp, idx, solve_idx=idx)
    nxu = 6 * (N + 1) + 2 * N; worst = 0.0
    for r in res:
        b = r["b"]
        assert np.array_equal(out["ssSel"][b], r["SSsel"].T) and np.array_equal(out["qSel"][b], r["Qsel"])
        w = np.concatenate([out["xPred"][b].ravel(), out["uPred"][b].ravel()])
        worst = max(worst, min(float((np.abs(w - o[:nxu]) / (1 + np.abs(o[:nxu]))).max()) for o in (r["opt"], r["opt2"])))
        assert max(r["cert"], r["cert2"]) < 1e-8
    print("N = %d, %d safe-set points, runtime kernel: %d problems against the oracle's optimum: worst |xu - z*| / (1 + |z*|) %.2e; iterations mean %.2f max %d" % (
        N, S, len(res), worst, out["iters"].mean(), out["iters"].max()))
    assert worst < common.TOL_XU
    ctx.close()


def test_runtime_kernel_agrees_with_the_fast_kernels(built):
    """The bench batch (N = 12, 256 problems) and plain LTV-MPC steps (no terminal set) through both kernels."""
    import bench
    from racinglmpc_amd import _capi
    g = common.load_lmpc_golden()
    pid = (np.array(g["xPID"]), np.array(g["uPID"]))
    B = 256
    inp = bench.synth_batch(g, B, 12)
    outs = []
    for rt in (False, True):
        cfg, _ = common.lmpc_config(g, 12, max_batch=B)
        ctx = _capi.Context(cfg, runtime_kernel=rt)
        assert ctx.solver_kind == (2 if rt else 0)
        for _ in range(4):
            ctx.model_add_trajectory(*pid); ctx.ss_add_trajectory(*pid)
        outs.append(ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"]))
        ctx.close()
    a, r = outs
    assert np.all(a["status"] == 0) and np.all(r["status"] == 0)
    assert np.array_equal(a["ssSel"], r["ssSel"]) and np.array_equal(a["qSel"], r["qSel"])
    dxu = max(np.abs(a["xPred"] - r["xPred"]).max(), np.abs(a["uPred"] - r["uPred"]).max())
    print("bench batch: runtime kernel vs four-wave kernel: |xu| differ by %.2e; iterations %.2f / %d vs %.2f / %d" % (dxu, r["iters"].mean(), r["iters"].max(), a["iters"].mean(), a["iters"].max()))
    assert dxu < 2e-7
    gl = common.load_ltv_golden()
    res = []
    for rt in (False, True):
        cfg, _ = common.mpc_config(gl, 12, max_batch=16)
        ctx = _capi.Context(cfg, runtime_kernel=rt)
        ctx.model_add_trajectory(gl["xPID"], gl["uPID"])
        res.append(ctx.step_batch(gl["x0"], gl["xLin"], gl["uLin"], gl["OldInput"]))
        ctx.close()
    w = np.concatenate([res[1]["xPred"].reshape(12, -1), res[1]["uPred"].reshape(12, -1)], axis=1)
    assert np.all(res[1]["status"] == 0) and np.abs(w - gl["sol_opt"][:, :102]).max() < common.TOL_XU
    assert np.array_equal(res[1]["ztNext"], res[1]["xPred"][:, -1, :])


@pytest.mark.parametrize("name", ["lmpc_wide_n12", "lmpc_n14", "lmpc_n40"])
def test_runtime_kernel_on_the_other_reference_fixtures(built, name):
    """The runtime-(N, S) kernel (forced) on the steps recorded from the executed reference in the other configurations: 72 safe-set points from 6 laps (more
    terminal-block columns than a wavefront has lanes), main.py's N = 14, BASELINE's N = 40 -- selection bit-exact, (x, u) and zt at the stated tolerances, KKT certificate."""
    from racinglmpc_amd import _capi
    g = common.load_variant_golden(name)
    N, S, L = int(g["N"]), int(g["numSS_Points"]), int(g["numSS_it"])
    gl = common.load_lmpc_golden()
    cfg, par = common.lmpc_config(gl, N, max_batch=16, numSS_it=L, numSS_Points=S)
    ctx = _capi.Context(cfg, runtime_kernel=True)
    assert ctx.solver_kind == 2
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"])
    for l in range(L):
        ctx.ss_add_trajectory(g["xPID"], g["uPID"])
        ctx.ss_replace_lap(l, g["SS"][l], g["uSS"][l], g["Qf"][l])
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"], zt=g["zt"], xPredPrev=g["xPredPrev"], hasPred=g["hasPred"].astype(np.int32), timeStep=g["t"].astype(np.int32))
    assert np.all(out["status"] == 0), out["status"]
    assert np.array_equal(out["ssSel"], np.transpose(g["SSsel"], (0, 2, 1))) and np.array_equal(out["qSel"], g["Qsel"])
    nxu = 6 * (N + 1) + 2 * N; worst = worst_zt = 0.0
    for r in range(g["x0"].shape[0]):
        w = np.concatenate([out["xPred"][r].ravel(), out["uPred"][r].ravel(), out["slack"][r], out["lambd"][r], out["sTerm"][r]])
        worst = max(worst, np.abs(w[:nxu] - g["sol_opt"][r][:nxu]).max())
        lam = g["sol_opt"][r][nxu + 2 * N:nxu + 2 * N + S]
        worst_zt = max(worst_zt, common.zt_err(out["ztNext"][r], out["ztuNext"][r], g["Succ"][r], g["SuccU"][r], lam))
        Pr, qr, Ar, lr, ur = common.dense_from_csc(g, r, prefix="")
        assert max(common.certificate(Pr, qr, Ar, lr, ur, w, out["mu"][r], 8 * N + S).values()) < common.TOL_KKT
    print("%s through the runtime kernel: worst |xu - certified optimum| %.2e, zt %.2e, iterations max %d" % (name, worst, worst_zt, out["iters"].max()))
    assert worst < common.TOL_XU and worst_zt < common.TOL_ZT
    ctx.close()


def test_runtime_kernel_on_the_360_point_stress_configuration(built):
    """SURVEY 8(d)'s stress variant (numSS_it = trToUse = 30, 360 safe-set points: 12 points per lap, 366 terminal-block rows) through the runtime kernel: 97 KB of LDS per QP."""
    from racinglmpc_amd import _capi
    g = common.load_30laps_golden("lmpc_30laps_stress_n12")
    gl = common.load_lmpc_golden()
    N = int(g["N"]); nl = int(g["nLaps"]); S = int(g["numSS_Points"])
    cfg, par = common.lmpc_config(gl, N, max_batch=16, max_laps=40, max_lap_len=1024, numSS_it=int(g["numSS_it"]), trToUse=int(g["trToUse"]))
    ctx = _capi.Context(cfg, runtime_kernel=True)
    for i in range(nl):
        ctx.model_add_trajectory(g["lapx%d" % i], g["lapu%d" % i]); ctx.ss_add_trajectory(g["lapx%d" % i], g["lapu%d" % i])
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"], zt=g["zt"], xPredPrev=g["xPredPrev"], hasPred=g["hasPred"].astype(np.int32), timeStep=g["t"].astype(np.int32))
    assert np.all(out["status"] == 0), out["status"]
    assert np.array_equal(out["ssSel"], np.transpose(g["SSsel"], (0, 2, 1))) and np.array_equal(out["qSel"], g["Qsel"])
    nxu = 6 * (N + 1) + 2 * N
    worst = max(np.abs(np.concatenate([out["xPred"][r].ravel(), out["uPred"][r].ravel()]) - g["sol_opt"][r][:nxu]).max() for r in range(g["x0"].shape[0]))
    print("30 laps / %d points through the runtime kernel: worst |xu - certified optimum| %.2e, iterations max %d" % (S, worst, out["iters"].max()))
    assert worst < common.TOL_XU
    ctx.close()
