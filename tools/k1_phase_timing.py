"""Developer tool: per-phase cycle breakdown of the regression kernel (work-group 0) at a given batch size (timing build of the library).
    python tools/k1_phase_timing.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from racinglmpc_amd import build as _build
so = os.environ.get("LMPC_LIB") or _build.build_flavour("timing", ["LMPC_TIMING"])
from racinglmpc_amd import _capi
_capi.LIB_PATH = so
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = bench.load_seed()
ctx = bench.make_ctx(g, 12, B, 0)
inp = bench.synth_batch(g, B, 12)
tb = np.zeros(24, np.int64)
f = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
rc = ctx.lib.lmpc_debug_k1_timing(ctx._h, C.c_int(B), f(inp["xLin"]), f(inp["uLin"]), tb.ctypes.data_as(C.c_void_p))
assert rc == 0, ctx.lib.lmpc_last_error()
names = ["load queries", "scan (computeIndices: prefilter + exact re-rank)", "stage selected points", "normal equations (35 sums per query)", "three 5x5 Cholesky solves per query",
         "analytic rows", "write A, B, C"]
print("regression kernel, batch %d, work-group 0: %d cycles" % (B, tb[7] - tb[0]))
for i, n in enumerate(names):
    print("  %-55s %7d" % (n, tb[i + 1] - tb[i]))
if tb[8]:
    print("  inside the scan of wave 0 (one lap, its share of the queries; loads drained at every stamp):")
    for a, b_, n in ((1, 8, "load the lap's prefilter image"), (8, 9, "step A, first trip (one query; two in the low-occupancy build)"), (9, 10, "step A, remaining trips"),
                     (10, 11, "step B, first trip of four queries: survivors' exact distances"), (11, 12, "step B, first trip: rank and store"), (12, 13, "step B, remaining trips"), (13, 2, "wait for the other waves")):
        print("    %-70s %7d" % (n, tb[b_] - tb[a]))
