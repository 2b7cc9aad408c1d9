"""Developer tool (GPU box): the PCIe-inclusive rate of the hot path -- lmpc_step_batch with HOST buffers (inputs copied to the device, outputs copied back,
every optional output requested) against lmpc_step_batch_dev with device-resident buffers (bench.py's `value`), same batch.   python tools/host_path_rate.py [batch ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
g = bench.load_seed()
for B in [int(a) for a in sys.argv[1:]] or [256, 4096]:
    ctx = bench.make_ctx(g, 12, B, 0); inp = bench.synth_batch(g, B, 12)
    for _ in range(5):
        ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    n = 50; t0 = time.perf_counter()
    for _ in range(n):
        out = ctx.step_batch(inp["x0"], inp["xLin"], inp["uLin"], inp["uOld"], zt=inp["zt"], timeStep=inp["timeStep"])
    dt = (time.perf_counter() - t0) / n
    nbytes_in = sum(np.asarray(inp[k]).nbytes for k in ("x0", "xLin", "uLin", "uOld", "zt", "timeStep")); nbytes_out = sum(np.asarray(v).nbytes for v in out.values())
    r = bench.run_config(g, 12, B, 0, steps=20, warmup=3)
    print("batch %d: host buffers %.3f ms per step = %.0f solves/s (%.2f MB in, %.2f MB out per step, every output requested, NumPy allocation of the outputs included); "
          "device-resident %.3f ms = %.0f solves/s" % (B, dt * 1e3, B / dt, nbytes_in / 1e6, nbytes_out / 1e6, 1e3 * B / r["solves_per_s"], r["solves_per_s"]))
    ctx.close()
