"""CPU checks of the test infrastructure itself: the vectorised KKT certificate against the dense oracle certificate on the
reference-produced fixture, and the construction of the K1 prefilter edge case."""
import numpy as np

from tests import common, k1_cases, kkt_batch


def _split(z, N=12, S=48):
    x = z[:, :6 * (N + 1)].reshape(-1, N + 1, 6); o = 6 * (N + 1)
    u = z[:, o:o + 2 * N].reshape(-1, N, 2); o += 2 * N
    s = z[:, o:o + 2 * N]; o += 2 * N
    return x, u, s, z[:, o:o + S], z[:, o + S:o + S + 6]


def test_batch_certificate_agrees_with_dense_oracle_certificate():
    """On the 60 recorded QPs: the certified optimum passes the banded certificate at the level of the dense one, and a
    1e-5 perturbation of one input is caught."""
    from oracle import lmpc_oracle as orc
    g = common.load_lmpc_golden()
    par = orc.QPParams.lmpc_default(12)
    x, u, s, lam, sT = _split(g["rec_sol_opt"])
    mu = g["rec_y_opt"][:, :144]
    ss = np.transpose(g["rec_SSsel"], (0, 2, 1))
    c = kkt_batch.certificate(par, g["rec_A"], g["rec_B"], g["rec_C"], g["rec_x0"], g["rec_OldInput"], x, u, s, mu, ssSel=ss, qSel=g["rec_Qsel"], lambd=lam, sTerm=sT)
    assert c["worst"].max() < 1e-8 and c["worst"].max() <= 10 * max(g["rec_cert_opt"].max(), 1e-10)
    u2 = u.copy(); u2[:, 3, 0] += 1e-5
    c2 = kkt_batch.certificate(par, g["rec_A"], g["rec_B"], g["rec_C"], g["rec_x0"], g["rec_OldInput"], x, u2, s, mu, ssSel=ss, qSel=g["rec_Qsel"], lambd=lam, sTerm=sT)
    assert c2["worst"].min() > 1e-6
    # the reference flow's eps = 1e-3 answers (polish failed) do not pass, its polished answers do
    xr, ur, sr, lr, tr = _split(g["rec_sol"])
    c3 = kkt_batch.certificate(par, g["rec_A"], g["rec_B"], g["rec_C"], g["rec_x0"], g["rec_OldInput"], xr, ur, sr, np.maximum(g["rec_y"][:, :144], 0.0),
                               ssSel=ss, qSel=g["rec_Qsel"], lambd=lr, sTerm=tr)
    ok = g["rec_polish"] == 1
    assert c3["worst"][ok].max() < 1e-6 and c3["worst"][~ok].min() > 1e-6


def test_batch_certificate_plain_mpc():
    from oracle import lmpc_oracle as orc
    gl = common.load_ltv_golden()
    par = orc.QPParams.mpc_default(12, 0.8)
    z, y = gl["sol_opt"], gl["y_opt"]
    x = z[:, :78].reshape(-1, 13, 6); u = z[:, 78:102].reshape(-1, 12, 2); s = z[:, 102:126]
    c = kkt_batch.certificate(par, gl["A"], gl["B"], gl["C"], gl["x0"], gl["OldInput"], x, u, s, y[:, :96])
    assert c["worst"].max() < 1e-8


def test_k1_slack_case_construction(monkeypatch):
    """The constructed lap really puts a row of the exact top 7 at integer distance T + 9 (so a prefilter slack of 8 drops it)."""
    from oracle import lmpc_oracle as orc
    case = k1_cases.slack_case()
    emu = k1_cases.emulate_prefilter(case)
    assert emu["T"] == case["E0"] + 6
    assert emu["e"][case["victim"]] == emu["T"] + 9
    assert [int(emu["e"][r]) for r in case["decoys"]] == [case["E0"] + i for i in range(7)]
    monkeypatch.setattr(orc, "H_BAND", case["h"]); monkeypatch.setattr(orc, "SCALING", np.ones(5))
    idx, K = orc.compute_indices(case["x"], case["u"], np.hstack([case["xq"][:3], case["uq"]]))
    assert sorted(idx) == sorted(case["decoys"][:6] + [case["victim"]])
    assert sorted(idx) == sorted(emu["exact_top"])


def test_bench_watchdog_prints_one_line_and_exits_non_zero():
    """bench.py's headline watchdog (a collective that never returns must not leave the job hanging, nor end it with rc 0): while the main thread sits in a call
    that does not come back, rank 0 prints one JSON line with `rccl_error` naming the phase, and the process ends with status 3."""
    import json
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench; "
            "wd = bench.Watchdog(0, 2, lambda m: {'metric': 'QP solves/sec (N=12, nx=6, nu=2)', 'value': None, 'rccl_error': m}); "
            "wd.arm(0.3, 'closing barrier'); time.sleep(30)") % common.ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr[-300:])
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["value"] is None and "closing barrier did not return within 0 s on rank 0" in line["rccl_error"]
    # a single rank arms nothing (no collective to hang in)
    code1 = code.replace("bench.Watchdog(0, 2,", "bench.Watchdog(0, 1,").replace("time.sleep(30)", "time.sleep(0.6); print('alive')")
    r1 = subprocess.run([sys.executable, "-c", code1], capture_output=True, timeout=60)
    assert r1.returncode == 0 and r1.stdout.decode().strip() == "alive"


def test_noise_prefetch_keeps_the_draw_order():
    """BatchedRollouts.prefetch_noise / the per-lap prefetch of the plant noise: the generator is consumed in exactly the order it would be without them,
    a prefetched array of another shape is discarded together with its draws."""
    from racinglmpc_amd import rollout

    def mk():
        r = rollout.BatchedRollouts.__new__(rollout.BatchedRollouts)
        r.rng = np.random.default_rng(5); r._pre = None; r.global_noise = False; r.noise_shard = None
        return r
    ref = np.random.default_rng(5)
    want = [ref.standard_normal((10, 4, 3)), ref.standard_normal((10, 4, 3)), ref.standard_normal((7, 4, 3))]
    a = mk(); got = [a._draw_noise(10, 4), a._draw_noise(10, 4), a._draw_noise(7, 4)]
    assert all(np.array_equal(x, y) for x, y in zip(want, got))
    b = mk(); b.prefetch_noise(10, 4, wait=True); got = [b._draw_noise(10, 4), b._draw_noise(10, 4), b._draw_noise(7, 4)]
    assert all(np.array_equal(x, y) for x, y in zip(want, got))
    c = mk(); c.prefetch_noise(9, 4); assert np.array_equal(c._draw_noise(10, 4), want[0])          # wrong shape prefetched: discarded, state restored


def test_context_is_not_destroyed_by_a_forked_child():
    """A multiprocessing worker inherits the parent's Context objects; when its garbage collector finalises the copies, lmpc_destroy must NOT run there (HIP calls in a
    forked child: a segmentation fault in the worker and a parent waiting for ever -- it happened in the oracle pools of the GPU tests, round 6).  No GPU needed: the
    object is built around a recording stand-in for the library."""
    import ctypes as C
    import os
    from racinglmpc_amd import _capi

    class _Lib:
        def __init__(self):
            self.destroyed = 0

        def lmpc_destroy(self, h):
            self.destroyed += 1

    ctx = _capi.Context.__new__(_capi.Context)
    ctx.lib = _Lib(); ctx._h = C.c_void_p(1234); ctx._pid = os.getpid()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                                       # the child: finalise the inherited copy, report how often the library was called
        os.close(r)
        ctx.close()
        os.write(w, b"%d" % ctx.lib.destroyed)
        os._exit(0)
    os.close(w)
    got = os.read(r, 16); os.waitpid(pid, 0); os.close(r)
    assert got == b"0"
    assert ctx._h.value == 1234 and ctx.lib.destroyed == 0     # the parent's handle is untouched ...
    ctx.close()
    assert ctx.lib.destroyed == 1 and not ctx._h              # ... and the parent destroys it once


def test_context_pool_releases_its_members_when_one_fails(monkeypatch):
    """ContextPool: a member that fails to come up closes the ones before it and the error is the member's own (not a recursion through the pool's attribute hook);
    `with` closes every member.  No GPU needed: Context is replaced by a recording stand-in."""
    import pytest
    from racinglmpc_amd import _capi
    made = []

    class _Ctx:
        def __init__(self, cfg):
            if len(made) == 2:
                raise _capi.LmpcError("no device memory")
            self.closed = 0; self.calls = []; made.append(self)

        def close(self):
            self.closed += 1

        def ss_add_point(self, *a):
            self.calls.append(a); return len(self.calls)

        def ss_num_laps(self):
            return 7

    monkeypatch.setattr(_capi, "Context", _Ctx)
    with pytest.raises(_capi.LmpcError, match="no device memory"):
        _capi.ContextPool(None, depth=3)
    assert [m.closed for m in made] == [1, 1]
    with pytest.raises(ValueError):
        _capi.ContextPool(None, depth=0)
    del made[:]
    with _capi.ContextPool(None, depth=2) as pool:
        assert pool.ss_add_point(1.0, 2.0) == 1 and [m.calls for m in made] == [[(1.0, 2.0)]] * 2      # store edits reach every member
        assert pool.ss_num_laps() == 7                                                                    # queries go to the first
    assert [m.closed for m in made] == [1, 1]
