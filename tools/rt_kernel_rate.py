"""Developer tool (GPU box): the runtime-(N, S) solve kernel against the fast kernels -- full steps per second at a few batch sizes (N = 12) and at horizons nobody built.
    python tools/rt_kernel_rate.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from racinglmpc_amd import _capi
g = bench.load_seed()


def rate(N, B, rt, steps=10):
    cfg = _capi.default_config(); cfg.N = N; cfg.max_batch = B
    for i, v in enumerate(g["track"].reshape(-1)):
        cfg.track[i] = float(v)
    cfg.track_rows = g["track"].shape[0]; cfg.trackLength = g["trackLength"]
    ctx = _capi.Context(cfg, runtime_kernel=rt)
    for _ in range(4):
        ctx.model_add_trajectory(g["xPID"], g["uPID"]); ctx.ss_add_trajectory(g["xPID"], g["uPID"])
    inp = bench.synth_batch(g, B, N)
    a, keep = ctx.step_dev_buffers(inp, diagnostics=False)
    dt, st = bench.time_steps(ctx, B, a, steps, 2)
    it = np.zeros(B, np.int32); ctx.dev_download(a.iters, it)
    r = (B * steps / dt, st.ms_solve / max(st.n_solve_timed, 1), float(it.mean()), int(it.max()), ctx.solver_kind)
    ctx.close()
    return r


for N, B in ((12, 1), (12, 256), (12, 4096), (13, 256), (33, 256)):
    for rt in ((False, True) if N == 12 else (True,)):
        r = rate(N, B, rt)
        print("N = %2d, batch %4d, %-14s: %9.0f steps/s, solve kernel %.3f ms, iterations %.2f / %d" % (N, B, "runtime kernel" if r[4] == 2 else "fast kernels", r[0], r[1], r[2], r[3]), flush=True)
