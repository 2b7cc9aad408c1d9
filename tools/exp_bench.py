"""Developer tool: quick throughput table of the full step at several batch sizes (LMPC_LIB selects a developer build of the library,
LMPC_FORCE_NW the waves per QP in an LMPC_DEV_FAST build), with a full-batch KKT certificate at one size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

g = bench.load_seed()
NH = int(os.environ.get("EXP_N", "12"))            # horizon
sizes = [int(a) for a in sys.argv[1:]] or [1, 256, 512, 1024, 4096, 8192]
row = []
for B in sizes:
    kw = {"tol_res": float(os.environ["EXP_TOLRES"])} if "EXP_TOLRES" in os.environ else {}
    r = bench.run_config(g, NH, B, 0, steps=20 if B <= 1024 and NH <= 20 else 6, warmup=3, **kw)
    row.append("B=%d: %.0f/s (solve %.3f ms, K1 %.3f ms, ok %d/%d, it max %d)" % (B, r["solves_per_s"], r["kernel_ms"]["lmpc_solve_kernel"], r["kernel_ms"]["lmpc_regress_kernel"] or 0.0,
                                                                                  r["solved_ok"], B, r["ipm_iters_max"]))
print("N=%d NW=%s  " % (NH, os.environ.get("LMPC_FORCE_NW", "auto")) + " | ".join(row))
if os.environ.get("EXP_CERT", "1") == "1":
    from tests import kkt_batch
    from oracle import lmpc_oracle as orc
    B = 1024
    ctx = bench.make_ctx(g, 12, B, 0)
    inp = bench.synth_batch(g, B, 12)
    out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    c = kkt_batch.certificate(orc.QPParams.lmpc_default(12), out["A"], out["B"], out["C"], inp["x0"], inp["uOld"], out["xPred"], out["uPred"], out["slack"], out["mu"],
                              ssSel=out["ssSel"], qSel=out["qSel"], lambd=out["lambd"], sTerm=out["sTerm"])
    print("   certificate B=%d: worst %.2e, status!=0: %d, iters mean %.2f" % (B, c["worst"].max(), int(np.sum(out["status"] != 0)), out["iters"].mean()))
