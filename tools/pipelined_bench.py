"""Developer tool (GPU box): bench.pipelined_leg at several depths / batch sizes.    python tools/pipelined_bench.py [N]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import __graft_entry__ as ge
ge.build()
g = bench.load_seed(); N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for B in (64, 256, 512, 1024):
    for depth in (1, 2, 3, 4):
        r = bench.pipelined_leg(g, N, B, 0, steps=20, warmup=3, depth=depth)
        print("batch %4d, %d in flight: %9.0f solves/s  (%.4f ms per step, iterations %.2f / %d, ok %d / %d)" % (B, depth, r["solves_per_s"], r["ms_per_step"], r["ipm_iters_mean"], r["ipm_iters_max"], r["solved_ok"], r["solved_of"]), flush=True)
