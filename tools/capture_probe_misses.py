"""Developer tool (GPU box): the closed-loop probes of tests/closed_loop_probe.py, keeping every sampled QP whose distance to the oracle's optimum exceeds `thr` (inputs, the
kernel's answer, iterations) in gpurun_out/probe_misses_<route>_N<N>_seed<seed>.npz for analysis on the CPU (tools/analyse_probe_misses.py).
    python tools/capture_probe_misses.py dropin|rollouts [seed] [N] [thr]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import closed_loop_probe as clp          # noqa: E402

route = sys.argv[1]; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5; NH = int(sys.argv[3]) if len(sys.argv) > 3 else 12; thr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-7
if route == "dropin":
    rec, err, cert, out, n = clp.probe(seed=seed, stride=4, laps=24, NH=NH, fast=True)
else:
    rec, err, cert, waves, it_max, bits = clp.rollout_probe(seed=seed, NH=NH, rollouts=1024, generations=3, per_step=4)
ezt = np.array([r["ezt"] for r in rec])
keep = [i for i in range(len(rec)) if err[i] > thr or ezt[i] > thr]
print("%s N = %d seed %d: %d QPs sampled, %d kept (|xu - z*| or zt above %.0e); worst xu %.2e zt %.2e" % (route, NH, seed, len(rec), len(keep), thr, err.max(), ezt.max()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
keys = ("A", "B", "C", "x0", "uOld", "SS", "Qsel", "xu", "it", "lap", "zt", "ztu", "Succ", "SuccU", "ezt", "indet")
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "probe_misses_%s_N%d_seed%d.npz" % (route, NH, seed)), err=err[keep], cert=cert[keep],
                    **{k: np.array([rec[i][k] for i in keep]) for k in keys})
