"""-m gpu: the device lap stores grow the way the reference's Python lists do (PredictiveControllers.py:418-445, 466-474; PredictiveModel.py:35-46):
lmpc_config.max_laps / max_lap_len are initial capacities only."""
import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu


def test_lap_stores_grow_like_the_reference_lists(built):
    g = common.load_lmpc_golden()
    outs = {}
    for name, kw in (("roomy", {}), ("tiny", dict(max_laps=1, max_lap_len=8))):
        ctx, par = common.make_lmpc_ctx(g, 5, max_batch=2, **kw)          # 5 + 6 laps of 1000+ rows through capacity 1 x 8: both stores reallocate repeatedly
        got = []

        def on_record(r):
            o = ctx.step_batch(g["rec_x0"][r][None], g["rec_xLin"][r][None], g["rec_uLin"][r][None], g["rec_OldInput"][r][None], zt=g["rec_zt"][r][None],
                               xPredPrev=g["rec_xPredPrev"][r][None], hasPred=np.array([g["rec_hasPred"][r]]), timeStep=np.array([g["rec_t"][r]]))
            got.append({k: o[k].copy() for k in ("A", "B", "C", "ssSel", "qSel", "xPred", "uPred", "ztNext", "status")})
        common.replay_lap(g, 5, ctx, on_record, max_records=12)             # addPoint between the records: rows appended one by one
        qf = ctx.ss_get_qfun(4)
        outs[name] = (got, qf)
        # more laps than any initial capacity, and a lap longer than any lap so far
        long_x = np.tile(g["xPID"], (3, 1)); long_u = np.tile(g["uPID"], (3, 1))
        for _ in range(70):
            ctx.ss_add_trajectory(g["xPID"], g["uPID"])
        ctx.ss_add_trajectory(long_x, long_u); ctx.model_add_trajectory(long_x, long_u)
        assert np.array_equal(ctx.ss_get_qfun(4), qf)                      # earlier laps survive the moves
        assert ctx.ss_get_qfun(5 + 70).shape[0] == 3000
        ctx.close()
    (a, qa), (b, qb) = outs["roomy"], outs["tiny"]
    assert len(a) == len(b) == 12 and np.array_equal(qa, qb)
    for ra, rb in zip(a, b):
        assert np.all(ra["status"] == 0)
        for k in ra:
            assert np.array_equal(ra[k], rb[k]), k                          # bit-identical answers whatever the initial capacity


def test_lap_stores_checkpoint_and_resume(built, tmp_path):
    """the lap stores as an .npz and back.  A context restored from the file holds the same rows (regression store in its sorted order, safe-set laps
    with the rows addPoint appended and their Q-function) and answers a recorded step bit for bit like the one that was saved."""
    import numpy as np
    from racinglmpc_amd import _capi
    from tests import common
    g = common.load_lmpc_golden()
    ctx, par = common.make_lmpc_ctx(g, 5, max_batch=4)                 # lap 5 of the fixture: five regression laps, a safe-set lap extended by addPoint, one replaced lap
    r = int(np.where(g["rec_lap"] == 5)[0][3])
    for t in range(3):                                                   # a few more addPoint rows on the last lap
        ctx.ss_add_point(g["all_x0"][t], g["all_u0"][t])
    args = (g["rec_x0"][r][None], g["rec_xLin"][r][None], g["rec_uLin"][r][None], g["rec_OldInput"][r][None])
    kw = dict(zt=g["rec_zt"][r][None], xPredPrev=g["rec_xPredPrev"][r][None], hasPred=np.array([g["rec_hasPred"][r]]), timeStep=np.array([g["rec_t"][r]]))
    ref = ctx.step_batch(*args, **kw)
    path = str(tmp_path / "stores.npz")
    ctx.save_stores(path)
    cfg, _ = common.lmpc_config(g, 12, max_batch=4)
    ctx2 = _capi.Context(cfg)
    ctx2.restore_stores(path)
    assert ctx2.ss_num_laps() == ctx.ss_num_laps()
    for i in range(ctx.ss_num_laps()):
        a, b = ctx.store_read_lap(1, i), ctx2.store_read_lap(1, i)
        assert all(np.array_equal(p, q) for p, q in zip(a, b)) and ctx.ss_lap_time(i) == ctx2.ss_lap_time(i)
    for i in range(5):
        a, b = ctx.store_read_lap(0, i), ctx2.store_read_lap(0, i)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    got = ctx2.step_batch(*args, **kw)
    for k in ("xPred", "uPred", "slack", "lambd", "sTerm", "ztNext", "ztuNext", "ssSel", "qSel", "A", "B", "C", "status", "iters"):
        assert np.array_equal(got[k], ref[k]), k
    import pytest
    with pytest.raises(_capi.LmpcError):
        ctx2.restore_stores(path)                                        # only into empty stores
    ctx.close(); ctx2.close()


def test_contexts_and_rollout_sessions_return_their_device_memory(built):
    """Create -> solve a batch -> a closed-loop rollout session -> destroy, a dozen times over: the device's free memory ends where it was after the first cycle
    (lmpc_destroy / lmpc_rollout_end release every buffer of the context, its work space, the host-mapped retry ring and the session; a long-lived service that
    re-creates contexts per (N, numSS_points) must not creep towards 288 GB).  Free memory: lmpc_device_memory (hipMemGetInfo in the library's own HIP runtime)."""
    from racinglmpc_amd import _capi, rollout
    g = common.load_lmpc_golden()
    inp = common.synthetic_inputs(g, 12, 64)

    def cycle(seed):
        ctx, par = common.make_lmpc_ctx(g, 4, max_batch=2048)
        out = ctx.step_batch(**inp)
        assert np.all(out["status"] == 0)
        ro = rollout.BatchedRollouts(ctx, g["track"], seed=seed)
        x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (16, 1)); x0[:, 5] = np.linspace(-0.05, 0.05, 16)
        best = rollout.lap_and_exchange(ro, x0, g["SS0"][1:14], g["uSS0"][1:13], K=2, T_max=400)
        assert len(best) == 2
        alive = _capi.device_memory(0)[0]
        ro.close(); ctx.close()
        return alive

    cycle(0)
    free0 = _capi.device_memory(0)[0]
    alive = [cycle(1 + i) for i in range(12)]
    free1 = _capi.device_memory(0)[0]
    print("free device memory after the first cycle %.1f MB, after twelve more %.1f MB; with a context and a session alive %.1f MB less" % (free0 / 2**20, free1 / 2**20, (free0 - min(alive)) / 2**20))
    assert free0 - max(alive) > 8 * 2**20          # (the reading moves with the allocations: the check below is not vacuous)
    assert free0 - free1 < 4 * 2**20, (free0, free1)


def test_create_that_cannot_get_its_memory_fails_cleanly(built):
    """lmpc_create for a batch whose work buffers exceed the device (hundreds of millions of QPs) returns LMPC_E_HIP with hipMalloc's message and gives back
    what it had allocated before the failing buffer."""
    from racinglmpc_amd import _capi
    g = common.load_lmpc_golden()
    free0, total = _capi.device_memory(0)
    cfg, _ = common.lmpc_config(g, N=12, max_batch=400_000_000)
    with pytest.raises(_capi.LmpcError, match="(?i)memory|hipMalloc"):
        _capi.Context(cfg)
    free1 = _capi.device_memory(0)[0]
    print("total %.0f GB; free before %.1f MB, after the failed create %.1f MB" % (total / 2**30, free0 / 2**20, free1 / 2**20))
    assert abs(free0 - free1) < 4 * 2**20
    ctx, par = common.make_lmpc_ctx(g, 4, max_batch=8)          # ... and the library is as usable as before
    assert ctx.ss_num_laps() == 4
    ctx.close()


@pytest.mark.timeout(600)
def test_hostile_inputs_never_take_the_process_down(built):
    """tools/adversarial_probe.py: NaN / inf / 1e300 in predictions, time steps, QP data, stored laps and rollout start states, empty and 20-row stores -- one process
    per case (a device fault would kill it).  Every case returns: a status bit on the hostile problem or an LmpcError from the argument checks, never a fault or a hang."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "adversarial_probe.py")], capture_output=True, text=True, timeout=560, cwd=root)
    print(r.stdout)
    heads = [l for l in r.stdout.splitlines() if l.startswith("== ")]
    assert len(heads) >= 12 and r.returncode == 0
    assert all(l.endswith("exit status 0") for l in heads), [l for l in heads if not l.endswith("exit status 0")]
