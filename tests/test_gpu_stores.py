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
