"""Developer tool: the figures of one bench.py line (and, optionally, the kernel table / one step's timeline of a rocprofv3 rollout trace) at a glance.
    python tools/show_bench.py gpurun_out/x_bench.json [gpurun_out/x_prof_rollout/ro_results.db]"""
import collections
import json
import sqlite3
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "ipm_iters_mean", "ipm_iters_max", "kernel_ms")})
print("sweep: " + "  ".join("%s: %.0f (%d it, %dw)" % (k, v["solves_per_s"], v["ipm_iters_max"], v["waves_per_qp"]) for k, v in d.get("sweep", {}).items()))
for k in ("config_batch4096_30laps", "config_batch4096_30laps_wide", "config_batch4096_30laps_stress", "config_N40_batch1024"):
    if k in d:
        print(k, "%.0f" % d[k]["solves_per_s"], "iters %.3f / %d" % (d[k]["ipm_iters_mean"], d[k]["ipm_iters_max"]), {a: round(b, 4) for a, b in d[k]["kernel_ms"].items() if b})
for gen in d.get("config_rollouts", {}).get("generations", []):
    print("rollouts generation %d: %.4f s, %d steps, %.0f closed-loop solves/s" % (gen["generation"], gen["seconds"], gen["simulated_steps"], gen["closed_loop_solves_per_s"]))
if len(sys.argv) > 2:
    c = sqlite3.connect(sys.argv[2])
    rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
    t = collections.defaultdict(list)
    for n, s, e, st in rows:
        t[n.split("(")[0][:60]].append((e - s) / 1e3)
    for k, v in t.items():
        if "rocclr" not in k:
            print("%-62s n=%5d mean %.1f us  min %.1f max %.1f" % (k, len(v), sum(v) / len(v), min(v), max(v)))
    ks = [(n.split("(")[0][:40], s, e, st) for n, s, e, st in rows]
    idx = [i for i, k in enumerate(ks) if "regress" in k[0]]
    i0 = idx[100]; t0 = ks[i0][1]
    for k in ks[i0:i0 + 7]:
        print("%-42s start %8.1f end %8.1f dur %6.1f stream %s" % (k[0], (k[1] - t0) / 1e3, (k[2] - t0) / 1e3, (k[2] - k[1]) / 1e3, k[3]))
