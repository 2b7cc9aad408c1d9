"""-m gpu: the condensed solve kernel (lmpc_solve_kernel_cd, experimental, LMPC_CD=1): same QPs, same answers as the Riccati kernels.
Dense 2N x 2N Newton matrix on the matrix cores + register Cholesky instead of the stage recursion; tests/ipm_model.py::ipm_solve_cd is its model."""
import os

import numpy as np
import pytest

import bench
from tests import common, kkt_batch

pytestmark = pytest.mark.gpu


CD_LIB = os.path.join(common.ROOT, "racinglmpc_amd", "liblmpc_hip_cd.so")


@pytest.fixture(autouse=True)
def cd_library():
    """The condensed kernel is not part of the default library any more (a measured alternative that is not faster: 3.5 MB and two minutes of
    hipcc for a negative result): these tests run against the opt-in flavour, racinglmpc_amd.build.build_flavour("cd", ["LMPC_WITH_CD"]), where it exists."""
    from racinglmpc_amd import _capi
    if not os.path.exists(CD_LIB):
        pytest.skip("liblmpc_hip_cd.so not built (python -c 'from racinglmpc_amd import build; build.build_flavour(\"cd\", [\"LMPC_WITH_CD\"])')")
    keep = (_capi.LIB_PATH, _capi._lib)
    _capi.LIB_PATH, _capi._lib = CD_LIB, None
    yield
    _capi.LIB_PATH, _capi._lib = keep


@pytest.fixture
def condensed(monkeypatch):
    monkeypatch.setenv("LMPC_CD", "1")          # read by lmpc_create
    yield
    monkeypatch.delenv("LMPC_CD", raising=False)


def _lmpc_ctx(g, N, B):
    from racinglmpc_amd import _capi
    cfg, par = common.lmpc_config(g, N, max_batch=B)
    ctx = _capi.Context(cfg)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    return ctx, par


@pytest.mark.parametrize("N", [12, 14, 8])
def test_condensed_kernel_matches_riccati_kernels(built, monkeypatch, N):
    g = common.load_lmpc_golden()
    B = 300
    inp = bench.synth_batch(g, B, N, seed=1234)
    ctx, par = _lmpc_ctx(g, N, B)
    ref = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    ctx.close()
    monkeypatch.setenv("LMPC_CD", "1")
    ctx, par = _lmpc_ctx(g, N, B)
    out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    ctx.close()
    assert np.all(ref["status"] == 0) and np.all(out["status"] == 0)
    d = max(np.abs(out["xPred"] - ref["xPred"]).max(), np.abs(out["uPred"] - ref["uPred"]).max())
    c = kkt_batch.certificate(par, out["A"], out["B"], out["C"], inp["x0"], inp["uOld"], out["xPred"], out["uPred"], out["slack"], out["mu"],
                              ssSel=out["ssSel"], qSel=out["qSel"], lambd=out["lambd"], sTerm=out["sTerm"])
    print("N=%d: condensed vs Riccati |dxu| %.2e, certificate %.2e, iterations %.2f / %.2f (max %d / %d)" % (
        N, d, c["worst"].max(), out["iters"].mean(), ref["iters"].mean(), out["iters"].max(), ref["iters"].max()))
    assert d < common.TOL_XU and c["worst"].max() < common.TOL_KKT
    # (zt = Succ lambda: two iterates that both meet the termination test can differ by ~1e-6 here -- the four seed laps are identical, so
    #  lambda is not unique; each kernel's zt against the optimum of the reference's QP is what test_gpu_parity.py checks)
    assert np.array_equal(out["ssSel"], ref["ssSel"]) and np.abs(out["ztNext"] - ref["ztNext"]).max() < 10 * common.TOL_ZT


def test_condensed_kernel_ltv_mpc_with_state_cost(built, condensed):
    """No terminal set, Q = diag(1, 1, 1, 1, 0, 100): the constant state-cost block of the reduced Hessian (sum_k Su_k' 2Q Su_k, accumulated on the
    matrix cores while the sensitivities are propagated) against the reference-executed LTV-MPC fixture."""
    from racinglmpc_amd import _capi
    g = common.load_ltv_golden()
    cfg, par = common.mpc_config(g, 12, max_batch=16)
    ctx = _capi.Context(cfg)
    ctx.model_add_trajectory(g["xPID"], g["uPID"])
    out = ctx.step_batch(g["x0"], g["xLin"], g["uLin"], g["OldInput"])
    assert np.all(out["status"] == 0), out["status"]
    w = np.concatenate([out["xPred"].reshape(12, -1), out["uPred"].reshape(12, -1)], axis=1)
    err = np.abs(w - g["sol_opt"][:, :102]).max()
    print("condensed ltv-mpc |xu - opt| %.2e, iterations %s" % (err, out["iters"]))
    assert err < common.TOL_XU
    ctx.close()
