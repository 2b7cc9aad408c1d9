"""-m gpu: the reference's actual experiment through the drop-in classes -- main.py:97-121, 40 LMPC laps at N = 14 (Laps = 40 + numSS_it,
initControllerParameters.py:46), Simulator.sim's loop (SysModel.py:22-54) with the oracle's restatement of the plant and seeded noise.

What the reference converges to (tests/golden/reference_flow_laps_n14.json: the EXECUTED reference classes -- LMPC, PredictiveModel,
Simulator.sim -- under np.random.seed(s), tests/golden/make_flow_golden.py; the GPU loop here consumes the same RandomState draws,
tests/closed_loop.noise_source): 203..209 steps in the first LMPC lap, ~100 by lap 7, 68..75 from lap 27 on, vx up to 3.5 m/s, |ey| up to
0.45 (lane slacks active).  That regime is where the round-2 kernels
flagged LMPC_ST_INEXACT once or twice per lap (an active lane row's barrier weight mu / t ~ 1e15 cost the Riccati recursion its accuracy); the
capped weights (LMPC_TH_INV, lmpc_kernels.hip.h) removed that: the tests assert NO status bit on any of the ~3 900 closed-loop QPs of a run.

Behavioural half of the a17 parity statement.  The reference's solver returns an eps = 1e-3 iterate, the GPU path the certified optimum, so the two
closed loops are compared as closed loops:
  * against the oracle flow solved to the certified optimum (same noise): identical lap lengths while round-off has not been amplified
    (the first laps; the loop is chaotic at the scale of single steps from about lap 8 on);
  * against the executed reference's eps = 1e-3 flow: lap lengths scatter by +-5 steps from noise seed to noise seed in EITHER flow (201..212 vs 203..208 in lap 0
    over eight seeds), so seed-by-seed agreement to +-2 steps does not exist even between two seeds of the reference flow itself.  Round 4: the
    yardstick is the executed reference on the SAME RandomState draws; the optimal closed loop is a few steps faster than the eps = 1e-3 one
    (eight-seed means 3.8 / 3.9 / 2.3 steps in laps 0 / 1 / 2), asserted as an interval.
"""
import json
import os

import numpy as np
import pytest

from tests import closed_loop, common

pytestmark = pytest.mark.gpu
_INEXACT = 64       # LMPC_ST_INEXACT (include/lmpc_hip.h)


def _fixture():
    with open(os.path.join(common.GOLDEN, "reference_flow_laps_n14.json")) as f:
        return json.load(f)


def _clean(recs):
    for r in recs:
        assert "error" not in r, r
        assert r["status"] == {0: r["steps"]}, "status bits in lap %d: %s" % (r["lap"], r["status"])


def test_forty_laps_at_main_py_horizon(built):
    g = common.load_lmpc_golden()
    ref = _fixture()["laps40"]
    seeds = sorted(ref)
    runs = {}
    for seed in seeds:
        recs = closed_loop.run_laps(closed_loop.DropinFlow(g, 14), g, 40, seed=int(seed))
        assert len(recs) == 40
        _clean(recs)                                                    # zero NUMERIC / MAXITER / REG_SINGULAR / INEXACT over the whole experiment
        runs[seed] = np.array([r["steps"] for r in recs])
        print("seed %s: %s  (executed reference: %s)  IPM iterations max %d" % (seed, runs[seed].tolist(), ref[seed], max(r["iters_max"] for r in recs)))
        # Iteration gates.  Rounds 1-5: 0-1 lap per seed held a QP above 20 iterations (31 and 24: a two-cycle of the separate / equal step rule).  Round 6 re-tuned three
        # constants of the step rules on every closed-loop QP in the NumPy model (tools/knob_model.py): measured here 17 / 17 / 18 at most over ~3 430 QPs per seed, mean 9.5.
        assert max(r["iters_max"] for r in recs) <= 20
        assert sum(r["iters_max"] > 16 for r in recs) <= 6
        assert np.mean([r["iters_mean"] for r in recs]) <= 10.4
        assert max(r["vx_max"] for r in recs) > 3.0                     # the regime the reference ends up in was actually reached
    gpu = np.mean([runs[s] for s in seeds], axis=0); cpu = np.mean([ref[s] for s in seeds], axis=0)
    print("three-seed means, GPU - executed reference, lap by lap:", np.round(gpu - cpu, 1).tolist())
    # converged lap time.  The reference's solver stops at eps = 1e-3, the GPU path returns the optimum of the same QP: the optimal closed loop is the
    # (slightly) faster one -- about 4 steps in the first lap, 2-3 at the end (same RandomState draws on both sides)
    assert gpu[-10:].mean() <= 75.0 and -6.0 <= gpu[-10:].mean() - cpu[-10:].mean() <= 2.0
    # three-seed means, lap by lap.  The loop is chaotic at the scale of single steps: one flow's laps scatter by +-5 steps from seed to seed, and a change
    # of the last bits of one solve (another summation order in a kernel) moves single late laps by 3-4 steps
    assert np.abs(gpu - cpu).max() <= 8.0, (gpu - cpu)
    assert -5.0 <= (gpu - cpu).mean() <= 1.0 and np.abs(gpu - cpu)[20:].mean() <= 5.0


def test_first_laps_against_both_reference_flows(built):
    g = common.load_lmpc_golden()
    # (a) same noise, certified optimum on both sides: the closed loops coincide
    exact = closed_loop.run_laps(closed_loop.OracleFlow(g, 14, solver="exact"), g, 2, seed=5)
    gpu = closed_loop.run_laps(closed_loop.DropinFlow(g, 14), g, 2, seed=5)
    _clean(gpu)
    assert [r["steps"] for r in gpu] == [r["steps"] for r in exact]
    assert [r["lap_time"] for r in gpu] == [r["lap_time"] for r in exact]                       # Qfun[it][0]
    assert abs(gpu[1]["vx_max"] - exact[1]["vx_max"]) < 1e-4 and abs(gpu[1]["ey_max"] - exact[1]["ey_max"]) < 1e-4
    # (b) the eps = 1e-3 reference flow, eight noise seeds, three laps: means within +-3 steps, every GPU lap inside the band the reference flow spans
    ref = _fixture()["laps3"]
    runs = []
    for seed in sorted(ref, key=int):
        recs = closed_loop.run_laps(closed_loop.DropinFlow(g, 14), g, 3, seed=int(seed))
        _clean(recs)
        runs.append([r["steps"] for r in recs])
    runs = np.array(runs, float); refa = np.array([ref[s] for s in sorted(ref, key=int)], float)
    print("GPU", runs.T.tolist(), "executed reference", refa.T.tolist(), "seed by seed GPU - reference", (runs - refa).T.tolist())
    # Same draws on both sides, so the difference is the solvers': the optimum of each QP (GPU) against the eps = 1e-3 iterate (reference).  Measured:
    # eight-seed means 3.8 / 3.9 / 2.3 steps FASTER in laps 0 / 1 / 2, single seeds between -12 and +4.
    d = runs.mean(axis=0) - refa.mean(axis=0)
    assert np.all(d <= 1.0) and np.all(d >= -6.0), d
    assert np.all(runs - refa >= -15) and np.all(runs - refa <= 8)


# Round 6: the closed-loop QPs against the oracle AT THE GRADED HORIZON, through both kernel routes the closed loop uses, >= 400 QPs per seed x 3 seeds
# each, with margin: worst scaled |xu - z*| <= 3e-7 and |zt - Succ lambda*| <= 5e-7 (the stated tolerances stay 1e-6).  What holds the margin is the termination rule
# (step_bound_ok, lmpc_kernels.hip.h: the a-posteriori bound of a contracting iteration below 3e-7), one rule for every horizon.
MARGIN_XU, MARGIN_ZT = 3e-7, 5e-7


def _report(tag, rec, err, cert):
    ezt = np.array([r["ezt"] for r in rec]); w = int(np.argmax(err))
    print("%s: %d QPs; worst |xu - z*| / (1 + |z*|) %.2e (lap %d, %d iterations), n > 1e-7: %d; worst |zt - Succ lambda*| / (1 + |zt|) %.2e (%d QPs with more than one optimal lambda); "
          "oracle certificates <= %.1e; iterations mean %.2f max %d" % (tag, len(rec), err.max(), rec[w]["lap"], rec[w]["it"], int((err > 1e-7).sum()), ezt.max(), sum(r["indet"] for r in rec),
                                                                      cert.max(), np.mean([r["it"] for r in rec]), max(r["it"] for r in rec)))
    return ezt


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_closed_loop_qps_against_the_oracle_n12_dropin(built, seed):
    """BASELINE's horizon, drop-in route (one QP per step: the four-wave kernel): every 6th QP of the first 24 LMPC laps (230 .. 77 steps: the laps in which round 5's
    sampled probe saw 8.75e-7 with the loose N <= 12 termination pair, and in which the model finds that pair 3.3e-6 off on 8 of 12 442 QPs)."""
    from tests import closed_loop_probe as clp
    rec, err, cert, out, n = clp.probe(seed=seed, stride=6, laps=24, NH=12, fast=True)
    _clean(out)
    ezt = _report("N = 12, drop-in (4 waves per QP), seed %d, %d QPs solved" % (seed, n), rec, err, cert)
    assert len(rec) >= 400 and cert.max() < 1e-8
    assert err.max() <= MARGIN_XU and ezt.max() <= MARGIN_ZT


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_closed_loop_qps_against_the_oracle_n12_rollouts(built, seed):
    """BASELINE's horizon, batched-rollout route at configs[3]'s per-GPU load (1024 device-resident rollouts: the two-wave kernel): three generations, four rollouts' QPs
    sampled per simulated step through lmpc_debug_rollout_qp (generation 0 starts from the PID laps, 1 and 2 continue the four fastest laps with real LMPC laps in the safe set)."""
    from tests import closed_loop_probe as clp
    rec, err, cert, waves, it_max, bits = clp.rollout_probe(seed=seed, NH=12, rollouts=1024, generations=3, per_step=4)
    ezt = _report("N = 12, 1024 rollouts (%d waves per QP), seed %d, iterations max over ALL QPs %d, status bits %#x" % (waves, seed, it_max, bits), rec, err, cert)
    assert waves == 2 and len(rec) >= 400 and cert.max() < 1e-8
    assert all(r["st"] & ~_INEXACT == 0 for r in rec)
    assert err.max() <= MARGIN_XU and ezt.max() <= MARGIN_ZT


def test_closed_loop_qps_against_the_oracle(built):
    """Every 8th QP of the first eight LMPC laps at main.py's horizon (seed 5: the laps in which round 5 found a flat QP -- error = 660 x dual residual -- 2.8e-6 from
    its optimum with every termination test of that time met) against the oracle's certified optimum of the same QP: |xu - z*| <= 1e-6 (1 + |z*|), SURVEY 8(c)-3."""
    from tests import closed_loop_probe as clp
    rec, err, cert, out, n = clp.probe(seed=5, stride=8, laps=8, NH=14, fast=True)      # (the oracle's dense interior-point solver only: its restated ADMM needs up to 20 s on some of these)
    _clean(out)
    ezt = _report("N = 14, drop-in, seed 5, %d QPs solved" % n, rec, err, cert)
    assert len(rec) >= 100 and cert.max() < 1e-8 and err.max() <= MARGIN_XU and ezt.max() <= MARGIN_ZT
