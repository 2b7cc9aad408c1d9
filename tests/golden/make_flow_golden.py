"""Lap-length sequences of the ORACLE's restatement of the reference flow (OracleLMPC: restated OSQP at eps = 1e-3 + polish; oracle plant)
over main.py's experiment -- 40 LMPC laps at N = 14 -- for three noise seeds, and over the first three laps for eight seeds.
Not an output of the executed reference (its Simulator draws from the unseeded global RNG and osqp is not installable); it is the CPU yardstick
the GPU closed loop is compared with in tests/test_gpu_closed_loop.py.  ~4 min.
    python tests/golden/make_flow_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import closed_loop, common

g = common.load_lmpc_golden()
out = dict(horizon=14, flow="oracle-osqp (eps 1e-3, polish)", laps40={}, laps3={})
for seed in (5, 6, 7):
    r = closed_loop.run_laps(closed_loop.OracleFlow(g, 14, solver="osqp"), g, 40, seed=seed)
    out["laps40"][str(seed)] = [x["steps"] for x in r]
    print(seed, out["laps40"][str(seed)], flush=True)
for seed in range(8):
    r = closed_loop.run_laps(closed_loop.OracleFlow(g, 14, solver="osqp"), g, 3, seed=seed)
    out["laps3"][str(seed)] = [x["steps"] for x in r]
with open(os.path.join(HERE, "reference_flow_laps_n14.json"), "w") as f:
    json.dump(out, f, indent=1)
