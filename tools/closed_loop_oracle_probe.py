"""Developer tool (GPU box): the reference's 40-lap experiment on the drop-in classes; every `stride`-th closed-loop QP -- real LMPC laps in the safe set,
lane slacks active in the fast laps -- is solved again by the oracle from the kernel's own A, B, C and selection (both restated on the explicit QP of the
reference, oracle.assemble_lmpc_qp) to its certified optimum, and (x, u) compared.     python tools/closed_loop_oracle_probe.py [seed] [stride] [laps] [N]
       python tools/closed_loop_oracle_probe.py rollouts [seed] [N] [rollouts] [generations] [per_step]      # the batched-rollout route (two waves per QP at 257..1024 rollouts)
(The probe itself is test infrastructure: tests/closed_loop_probe.py; tests/test_gpu_closed_loop.py runs it.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.closed_loop_probe import probe, rollout_probe          # noqa: E402,F401


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rollouts":
        a = [int(v) for v in sys.argv[2:]]
        seed, NH, R, G, P = (a + [5, 12, 1024, 3, 4][len(a):])[:5]
        rec, err, cert, waves, itmax, bits = rollout_probe(seed, NH, R, G, P)
        lap = np.array([r["lap"] for r in rec]); w = int(np.argmax(err))
        print("N = %d, seed %d, %d rollouts x %d generations (%d waves per QP): %d QPs sampled; iterations max %d (all QPs), status bits %#x" % (NH, seed, R, G, waves, len(rec), itmax, bits))
        print("worst |xu - z*| / (1 + |z*|) %.2e (generation %d, step %d, %d iterations), oracle certificates <= %.1e; by generation: %s; n > 3e-7: %d" % (
            err.max(), rec[w]["lap"], rec[w]["t"], rec[w]["it"], cert.max(), [float("%.2e" % err[lap == k].max()) for k in range(G) if (lap == k).any()], int((err > 3e-7).sum())))
        ezt = np.array([r["ezt"] for r in rec]); print("worst |zt - Succ lambda*| / (1 + |zt|) where lambda* is determinate %.2e (%d QPs with more than one optimal lambda)" % (ezt.max(), sum(r["indet"] for r in rec)))
        sys.exit(0)
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    laps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    NH = int(sys.argv[4]) if len(sys.argv) > 4 else 14
    rec, err, cert, out, n = probe(seed, stride, laps, NH)
    lap = np.array([r["lap"] for r in rec]); w = int(np.argmax(err))
    print("N = %d, seed %d: %d laps, %d QPs, %d sampled; last lap %d steps" % (NH, seed, len(out), n, len(rec), out[-1]["steps"]))
    print("worst |xu - z*| / (1 + |z*|) %.2e (lap %d, %d iterations), oracle certificates <= %.1e; by lap thirds: %s; n > 3e-7: %d" % (
        err.max(), rec[w]["lap"], rec[w]["it"], cert.max(), [float("%.2e" % err[(lap >= a) & (lap < b)].max()) for a, b in ((0, 13), (13, 27), (27, 99))], int((err > 3e-7).sum())))
    ezt = np.array([r["ezt"] for r in rec]); print("worst |zt - Succ lambda*| / (1 + |zt|) where lambda* is determinate %.2e (%d QPs with more than one optimal lambda)" % (ezt.max(), sum(r["indet"] for r in rec)))
    bad = np.argsort(-err)[:6]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "closed_loop_probe_worst_N%d_seed%d.npz" % (NH, seed)), err=err[bad],
                        **{k: np.array([rec[i][k] for i in bad]) for k in ("A", "B", "C", "x0", "uOld", "SS", "Qsel", "xu", "it", "lap")})
