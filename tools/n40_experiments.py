"""Developer tool: BASELINE configs[4] (N = 40, batch 1024) through developer flavours of the library -- the differential trace against the NumPy model and
the two constructs that rounds 3-4 left "unexplained" at this horizon (gram8_mfma in the one-wave kernel, bound_ctrl cross-lane moves).

    python tools/n40_experiments.py build          # HERE (cross-compile): the flavours below, liblmpc_hip_<name>.so
    python tools/n40_experiments.py run [names]    # on the GPU box: each flavour in its own process -> gpurun_out/r5_n40_<name>.npz + one summary line each
    python tools/n40_experiments.py audit          # on the GPU box: the EXEC-mask audit flavour over N = 40 / 12 / 14 batches on all three kernel routes

Flavours (all: only the N = 40 variants, -DLMPC_TRACE):
    trace40    the product code                         gram40    -DLMPC_FORCE_GRAM8: 4x4x4 Gram matrix of the terminal factor in the one-wave kernel too
    bc40       -DLMPC_DPP_BC: bound_ctrl form of every cross-lane move            gram40nf / bc40nf / trace40nf: the same without -mllvm -amdgpu-mfma-vgpr-form
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BASE = ["LMPC_DEV_FAST", "LMPC_DEV_N=40", "LMPC_TRACE", "LMPC_NO_ISA_CHECK"]      # (these builds are WANTED as the compiler makes them, faulty or not)
FLAVOURS = {"trace40": ([], True), "gram40": (["LMPC_FORCE_GRAM8"], True), "bc40": (["LMPC_DPP_BC"], True),
            "trace40nux": (["LMPC_NO_UNIFORM_EXIT"], True), "gram40nux": (["LMPC_FORCE_GRAM8", "LMPC_NO_UNIFORM_EXIT"], True), "bc40nux": (["LMPC_DPP_BC", "LMPC_NO_UNIFORM_EXIT"], True)}


def build():
    from concurrent.futures import ThreadPoolExecutor
    from racinglmpc_amd import build as b
    with ThreadPoolExecutor(max_workers=4) as ex:
        futs = [ex.submit(b.build_flavour, n, BASE + d, False, (), vf) for n, (d, vf) in FLAVOURS.items()]
        futs.append(ex.submit(b.build_flavour, "audit", ["LMPC_EXEC_AUDIT"]))
        for f in futs:
            print(f.result())


def _one(name, B=1024, N=40):
    """(child process, LMPC_LIB set) the N = 40 batch through one flavour; forces the route the product takes at this batch (one wave per QP, [A_k | B_k] in global memory)."""
    from tests import common
    from tests.test_gpu_certificates import _ctx_pid, _certify
    from tools.n40_model import inputs
    g = common.load_lmpc_golden()
    ctx, par = _ctx_pid(g, N, B)
    inp = inputs(g, N, B)
    tr = ctx.debug_trace_begin(B)
    out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    trace = ctx.debug_trace_fetch(tr, B)
    it = np.asarray(out["iters"]); st = np.asarray(out["status"])
    last = trace[:, -1, :]                                  # (dynamics residual as the kernel sees it | with C, A, B re-read from the inputs | max |c_r - C| | max |AB - (A, B)| | it | converged)
    res = dict(name=name, waves=ctx.solver_waves(B), iters_mean=float(it.mean()), iters_max=int(it.max()), hist=np.bincount(it).tolist(),
               last_row_max=[float(np.nanmax(last[:, c])) for c in range(4)], converged=int(np.nansum(last[:, 5])),
               status=dict(zip(*[a.tolist() for a in np.unique(st, return_counts=True)])))
    try:
        c = _certify(par, out, inp, what=name)
        res["certificate_worst"] = float(c["worst"].max())
    except AssertionError as e:
        res["certificate_error"] = str(e)[:200]
    import time
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"]); ts.append(time.perf_counter() - t0)
    res["host_call_ms_min"] = min(ts) * 1e3
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r5_n40_%s.npz" % name), iters=it, status=st, trace=trace, xPred=out["xPred"], uPred=out["uPred"])
    print("N40 " + json.dumps(res), flush=True)
    ctx.close()


def run(names):
    for n in names:
        lib = os.path.join(ROOT, "racinglmpc_amd", "liblmpc_hip_%s.so" % n)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "_one", n], env=dict(os.environ, LMPC_LIB=lib), capture_output=True, text=True, timeout=600)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("N40 ") or "certificate" in l) or ("N40 %s FAILED rc=%d %s" % (n, r.returncode, r.stderr[-400:])), flush=True)


def _audit():
    import bench
    from tests import common
    from tests.test_gpu_certificates import _ctx_pid
    from tools.n40_model import inputs
    g = common.load_lmpc_golden()
    rows = []
    for N, B in ((40, 1024), (40, 256), (40, 512), (12, 256), (12, 1024), (12, 4096), (14, 300), (14, 2048), (20, 300), (8, 64)):
        ctx, par = _ctx_pid(g, N, B)
        inp = inputs(g, N, B) if N != 12 else bench.synth_batch(g, B, 12)
        ctx.debug_exec_audit(reset=True)
        out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
        partial, calls = ctx.debug_exec_audit(reset=True)
        rows.append(dict(N=N, B=B, waves=ctx.solver_waves(B), status_ok=int(np.sum(out["status"] == 0)), iters_mean=float(np.mean(out["iters"])),
                         partial=[int(v) for v in partial], calls=[int(v) for v in calls]))
        print("AUDIT " + json.dumps(rows[-1]), flush=True)
        ctx.close()
    tot_p = np.sum([r["partial"] for r in rows], axis=0); tot_c = np.sum([r["calls"] for r in rows], axis=0)
    print("AUDIT total: calls per site %s, under an incomplete EXEC mask %s" % (tot_c.tolist(), tot_p.tolist()))
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "r5_exec_audit.json"), "w"))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        build()
    elif cmd == "_one":
        _one(sys.argv[2])
    elif cmd == "audit":
        lib = os.path.join(ROOT, "racinglmpc_amd", "liblmpc_hip_audit.so")
        if os.environ.get("LMPC_LIB") != lib:
            sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), "audit"], env=dict(os.environ, LMPC_LIB=lib)).returncode)
        _audit()
    else:
        run(sys.argv[2:] or list(FLAVOURS))          # (any liblmpc_hip_<name>.so built with LMPC_DEV_N=40 and LMPC_TRACE can be named)
