"""Developer tool: the reference's actual experiment -- main.py:97-121, `Laps - numSS_it = 40` LMPC laps at N = 14 -- through

    --flow dropin   the drop-in classes on the GPU (needs a GPU)
    --flow osqp     the oracle's restatement of the reference flow (restated OSQP, eps = 1e-3, polish), CPU
    --flow exact    the oracle flow with the certified optimum of every QP, CPU
    --flow ipm      the oracle flow with the NumPy model of the kernel's interior-point iteration as QP solver, CPU

with the same plant (oracle restatement of Simulator.dynModel) and the same seeded noise.  Prints one line per lap (steps, status
histogram, iterations, max vx, max |ey|) and writes the per-lap records as JSON.
    python tools/lmpc_40laps.py --flow dropin --laps 40 --horizon 14 --seed 5 --out profiles/r3_40laps_dropin.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import closed_loop, common


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flow", default="dropin", choices=["dropin", "osqp", "exact", "ipm"])
    ap.add_argument("--laps", type=int, default=40)
    ap.add_argument("--horizon", type=int, default=14)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dump-from", type=int, default=None, help="oracle flows: keep the structured QP inputs of every step from this lap on")
    ap.add_argument("--dump", default=None, help="npz file for --dump-from")
    a = ap.parse_args()
    g = common.load_lmpc_golden()
    if a.flow == "dropin":
        flow = closed_loop.DropinFlow(g, a.horizon)
    else:
        flow = closed_loop.OracleFlow(g, a.horizon, solver=a.flow)

    def on_lap(r):
        print(json.dumps(r), flush=True)
    recs = closed_loop.run_laps(flow, g, a.laps, seed=a.seed, on_lap=on_lap, dump_from=a.dump_from)
    laps = [r.get("steps") for r in recs]
    print("flow %s, N = %d, seed %d: lap lengths %s" % (flow.name, a.horizon, a.seed, laps))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(flow=flow.name, horizon=a.horizon, seed=a.seed, laps=recs), f, indent=1)
    if a.dump and getattr(flow, "dump", None):
        d = flow.dump
        np.savez_compressed(a.dump, **{k: np.array([e[k] for e in d]) for k in d[0]})


if __name__ == "__main__":
    main()
