#!/usr/bin/env python
"""bench.py -- QP solves/sec of the LMPC hot path on MI355X (driver contract: see task statement).

One "step" = one pass of the full hot path (LTV regression for N horizon points, safe-set selection,
QP assembly-in-structure + solve to the certified optimum, unpack) over one batch of synthetic QPs whose
inputs are already resident in HBM.  Workload at N_gpus=1: BASELINE.json configs[1]
("batch=256 LMPC QPs, N=12, fixed safe-set, 1xMI355X"), generated as SURVEY.md section 8(d) prescribes.

Multi-GPU: one process per GPU.  Either the driver launches the ranks (torch.distributed.run sets RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*), or `python bench.py --gpus N` on its own spawns N ranks of itself with that environment.  Every rank
runs the same batch size on its own GPU (weak scaling, no data-path collective: the QPs are independent); the barrier and the
max-over-ranks time go through RCCL behind the library's C ABI (racinglmpc_amd/parallel.py); value = all ranks' solves /
max-over-ranks time.  No PyTorch anywhere in the measured path.

Besides the contract's keys the line carries (cheap, a few seconds together): the batch sweep 1..8192, BASELINE configs[2]
(batch 4096, safe set from 30 laps), configs[4] (N = 40, batch 1024) and configs[3] (closed-loop rollouts sharded over the ranks
with one all-gather per lap).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP64_VEC_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (spec)
SEED_LAP = os.path.join(ROOT, "tests", "golden", "lmpc_n12.npz")     # PID seed lap + track table recorded from the executed reference (data only)


def load_seed():
    g = np.load(SEED_LAP)
    return dict(xPID=np.array(g["xPID"]), uPID=np.array(g["uPID"]), track=np.array(g["track"]), trackLength=float(g["trackLength"]))


def synth_batch(g, B, N, seed=1234, lap=None):
    """SURVEY 8(d) 'cfg batch=256, fixed safe-set': problem b starts at row t_b = 37 b mod 900 of the PID seed lap
    (of `lap` = (x, u) if given: the 30-lap configuration queries lap 29)."""
    xq, uq = (g["xPID"], g["uPID"]) if lap is None else lap
    rng = np.random.default_rng(seed)
    span = 900 if lap is None else xq.shape[0] - 40 - N - 2
    tb = (37 * np.arange(B)) % span
    eps = rng.normal(size=(B, 6)) * np.array([.02, .01, .02, .01, 0.0, .02])
    return dict(x0=xq[tb] + eps, xLin=np.stack([xq[t + 1:t + N + 2] for t in tb]), uLin=np.stack([uq[t + 1:t + N + 1] for t in tb]),
                uOld=uq[tb].copy(), zt=xq[tb + N + 1].copy(), timeStep=(tb % 300).astype(np.int32), hasPred=np.zeros(B, np.int32),
                xPredPrev=np.zeros((B, N + 1, 6)))


def make_ctx(g, N, B, device, laps=None, pool_depth=0, **kw):
    """Context with the reference's LMPC tuning (lmpc_config_default: initControllerParameters.py:28-59) on the recorded track;
    stores: 4 x PID lap (main.py:102-110) unless `laps` is given."""
    from racinglmpc_amd import _capi
    cfg = _capi.default_config()
    cfg.N = N; cfg.max_batch = max(B, 1); cfg.device = device
    for i, v in enumerate(g["track"].reshape(-1)):
        cfg.track[i] = float(v)
    cfg.track_rows = g["track"].shape[0]; cfg.trackLength = g["trackLength"]
    for k, v in kw.items():
        setattr(cfg, k, v)
    ctx = _capi.Context(cfg) if not pool_depth else _capi.ContextPool(cfg, pool_depth)
    for x, u in (laps if laps is not None else [(g["xPID"], g["uPID"])] * 4):
        ctx.model_add_trajectory(x, u)
        ctx.ss_add_trajectory(x, u)
    return ctx


def pid_laps(ctx, g, n_laps, max_steps=600):
    """SURVEY 8(d) 'safe-set from 30 laps': n single-lap PID trajectories, lap i at target speed 0.6 + 0.02 i (Utilities.PID.solve
    control law, noise seeded), integrated by the plant kernel (lmpc_plant_step_batch = Simulator.dynModel)."""
    TL = g["trackLength"]
    rng = np.random.default_rng(77)
    vt = 0.6 + 0.02 * np.arange(n_laps)
    x = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (n_laps, 1)); xg = x.copy()
    X, U = [], []
    done = -np.ones(n_laps, int)
    for t in range(max_steps):
        u = np.stack([-0.6 * x[:, 5] - 0.9 * x[:, 3] + np.clip(rng.standard_normal(n_laps) * 0.25, -0.9, 0.9),
                      1.5 * (vt - x[:, 0]) + np.clip(rng.standard_normal(n_laps) * 0.10, -0.2, 0.2)], axis=1)
        X.append(x.copy()); U.append(u)
        x, xg, _st = ctx.plant_step_batch(x, xg, u, rng.standard_normal((n_laps, 3)))
        done[(done < 0) & (x[:, 4] > TL)] = t + 1
        if np.all(done > 0):
            break
    X = np.stack(X, 1); U = np.stack(U, 1)
    return [(X[i, :min(done[i] + 20, X.shape[1])], U[i, :min(done[i] + 20, X.shape[1])]) for i in range(n_laps)]


ROLLOUT_WATCHDOG_S = 240   # multi-rank closed-loop leg: seconds before the watchdog gives it up (it takes about one second)
HEADLINE_WATCHDOG_S = 300  # multi-rank headline (communicator set-up, timed region, its closing barrier, the gathers behind it): seconds before the watchdog gives it up
PROGRESS = {}              # what the headline has reached, for the watchdog's line: t0 (timed region entered), t_sync (this rank's device drained after the K steps)


class Watchdog:
    """A collective that one rank never enters would hang the job with nothing on stdout.  While armed, a timer thread waits beside the main thread (which may sit in a
    C call, GIL released); when it fires, rank 0 prints ONE JSON line in the shape of the headline -- this rank's own rate if the timed steps had finished, else null --
    with "rccl_error" saying which phase did not return, and every rank leaves with exit status 3 (a launcher then reports the job as failed, not as rc 0)."""

    def __init__(self, rank, world, make_line):
        self.rank, self.world, self.make_line, self.timer, self.phase = rank, world, make_line, None, ""

    def arm(self, seconds, phase):
        import threading
        self.disarm(); self.phase = phase
        if self.world <= 1 and not os.environ.get("LMPC_BENCH_FORCE_WATCHDOG"):
            return
        self.timer = threading.Timer(seconds, self._fire, args=(seconds,)); self.timer.daemon = True; self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel(); self.timer = None

    def _fire(self, seconds):
        fail(self.rank, self.make_line, "%s did not return within %d s on rank %d (watchdog)" % (self.phase, seconds, self.rank))


def fail(rank, make_line, message, code=3):
    """The headline cannot be completed (a collective hung, the communicator could not be built): rank 0 prints the line it has with "rccl_error", all leave non-zero."""
    try:
        if rank == 0:
            sys.stdout.flush()
            print(json.dumps(make_line(message)), flush=True)
    finally:
        os._exit(code)
PROFILE_EVERY = 5      # HIP events around every 5th launch of each kernel inside the timed region (an event record costs ~4 us of stream time)


def time_steps(ctx, B, a, steps, warmup, sync=None):
    """W untimed + K timed full steps on device-resident buffers; returns (seconds, stats of the timed launches)."""
    for _ in range(warmup):
        ctx.step_batch_dev(B, a)
    (sync or ctx.sync)()
    ctx.reset_stats(); ctx.set_profiling(PROFILE_EVERY)
    t0 = time.perf_counter(); PROGRESS["t0"] = t0; PROGRESS.pop("t_sync", None)
    for _ in range(steps):
        ctx.step_batch_dev(B, a)
    (sync or ctx.sync)()
    dt = time.perf_counter() - t0
    st = ctx.stats(); ctx.set_profiling(False)
    return dt, st


def run_config(g, N, B, device, steps, warmup, laps=None, query_lap=None, **kw):
    """One extra configuration: build, time, check the status of every problem, free."""
    ctx = make_ctx(g, N, B, device, laps=laps, **kw)
    inp = synth_batch(g, B, N, lap=query_lap)
    a, keep = ctx.step_dev_buffers(inp, diagnostics=False)
    dt, st = time_steps(ctx, B, a, steps, warmup)
    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32)
    ctx.dev_download(a.status, status); ctx.dev_download(a.iters, iters)
    out = dict(batch=B, N=N, solves_per_s=B * steps / dt, ms_per_step=dt / steps * 1e3, solved_ok=int(np.sum(status == 0)),
               ipm_iters_mean=float(iters.mean()), ipm_iters_max=int(iters.max()), waves_per_qp=ctx.solver_waves(B),
               kernel_ms={"lmpc_solve_kernel": st.ms_solve / max(st.n_solve_timed, 1), "lmpc_regress_kernel": st.ms_regress / st.n_regress_timed if st.n_regress_timed else None},
               fused_step=st.n_regress == 0)
    for p in keep:
        ctx.dev_free(p)
    ctx.close()
    return out


def pipelined_leg(g, N, B, device, steps, warmup, depth=2):
    """Throughput of back-to-back INDEPENDENT batches with `depth` batches in flight: one context (own HIP stream, own work buffers, own copy of the lap stores)
    per batch in flight, the steps dealt to them in turn.  Inside one launch the CUs whose QP has converged idle until the slowest QP of the batch has (mean 8.6
    against a maximum of 13 interior-point iterations at batch 256: a third of the CU time); with a second batch queued on another stream its work-groups start on
    those CUs.  Every step is still one full pass (regression + selection + QP solve) over one batch of B problems; what changes is that a batch no longer waits
    for the previous batch's slowest QP.  NOT the headline `value` (one batch at a time, comparable across rounds): a separate, labelled figure -- what a caller
    that keeps several requests in flight gets; a closed loop, where step t + 1 needs step t, cannot use it."""
    pool = make_ctx(g, N, B, device, pool_depth=depth)          # racinglmpc_amd._capi.ContextPool: `depth` contexts, the same laps in each
    inp = synth_batch(g, B, N, seed=1234)
    bufs = pool.step_dev_buffers(inp, diagnostics=False)
    for w in range(warmup * depth):
        pool.step_batch_dev(B, bufs)
    pool.sync()
    t0 = time.perf_counter()
    for k in range(steps * depth):
        pool.step_batch_dev(B, bufs)
    pool.sync()
    dt = time.perf_counter() - t0
    ok = 0; it_all = []
    for c, (a, keep) in zip(pool.members, bufs):
        status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32)
        c.dev_download(a.status, status); c.dev_download(a.iters, iters)
        ok += int(np.sum(status == 0)); it_all.append(iters)
        for q in keep:
            c.dev_free(q)
    pool.close()
    return dict(batches_in_flight=depth, batch=B, N=N, steps=steps * depth, solves_per_s=B * steps * depth / dt, ms_per_step=dt / (steps * depth) * 1e3,
                solved_ok=ok, solved_of=B * depth, ipm_iters_mean=float(np.mean(it_all)), ipm_iters_max=int(np.max(it_all)),
                note="independent batches on %d HIP streams (one context each); not the headline value -- see DESIGN 3.3" % depth)


def rollout_leg(g, comm, ctx, rollouts_per_gpu, generations=2, K=4, T_max=400):
    """BASELINE configs[3]: closed-loop LMPC laps, device resident, sharded over the ranks, one all-gather of the fastest laps per lap.
    Runs last on the main context (its stores grow by K laps per generation on every rank, identically)."""
    from racinglmpc_amd import rollout
    world, rank = comm.world, comm.rank
    total = rollouts_per_gpu * world
    ro = rollout.BatchedRollouts(ctx, g["track"], seed=100 + rank)
    x0 = np.tile(np.array([0.5, 0, 0, 0, 0, 0.0]), (total, 1)); x0[:, 5] = np.linspace(-0.1, 0.1, total); x0[:, 0] += np.linspace(0.0, 0.1, total)
    gen = rollout.LmpcGeneration(ro, total, K=K, T_max=T_max, ext=40, comm=comm)
    # The synthetic plant noise of the first lap is input data: drawn BEFORE the timed region (later laps' draws overlap the device work on a worker thread).  The
    # reference draws inside its simulation loop (SysModel.py:139-141), so generation 0's rate excludes this host time: the line says so (`noise_pregenerated`,
    # `noise_host_ms_excluded`; rounds 1-4 drew inside the timed region -- their closed-loop numbers included it).
    t_pre = time.perf_counter(); gen.prepare(); noise_ms = (time.perf_counter() - t_pre) * 1e3
    laps = []
    for it in range(generations):
        comm.barrier(); t0 = time.perf_counter()
        best = gen.run(x0, g["xPID"][1:14], g["uPID"][1:13])
        dt = float(comm.allreduce_max(time.perf_counter() - t0)[0])
        steps = int(ctx._ro_t)
        laps.append(dict(generation=it, seconds=dt, simulated_steps=steps, best_lap_steps=[b[4] for b in best], src_ranks=[int(b[3]) for b in best],
                         closed_loop_solves_per_s=total * steps / dt, allgather_bytes_per_rank=int(gen.last_exchange[0]),
                         exchange_seconds=float(comm.allreduce_max(gen.last_exchange[1])[0])))
    ro.close()                 # (waits for the prefetched draw nobody will use, restores the generator, ends the worker thread)
    info = ctx.comm_info()
    return dict(noise_pregenerated=True, noise_host_ms_excluded=round(noise_ms, 2), rollouts_total=total, rollouts_per_gpu=rollouts_per_gpu, K=K, collective="ncclAllGather (RCCL)" if info[2] else "none (single process)",
                rccl_ranks=info[1] if info[2] else 1, generations=laps)


def _cpu_worker(args):
    """Full reference-algorithm steps (oracle port) on one core for `budget` seconds, cycling over the batch from `start`."""
    start, budget, N, B = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)                         # one BLAS thread per process
    except Exception:
        pass
    from oracle import lmpc_oracle as orc
    g = load_seed()
    inp = synth_batch(g, B, N, seed=1234)
    par = orc.QPParams.lmpc_default(N)
    pt, TL = g["track"], g["trackLength"]
    xS = [g["xPID"]] * 4; uS = [g["uPID"]] * 4
    Qf = [orc.compute_cost(g["xPID"], TL)] * 4
    done = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        b = (start + done) % B
        A, Bm, C = orc.compute_ltv_dynamics(xS, uS, [0, 1, 2, 3], pt, inp["xLin"][b], inp["uLin"][b], N)
        zt = inp["zt"][b].copy()
        if zt[4] - inp["x0"][b][4] > TL / 2:
            zt[4] = np.max([zt[4] - TL, 0])
        SSsel, Qsel, Succ, SuccU = orc.terminal_components(xS, uS, Qf, [1000] * 4, zt, 48, 4, None, 4, int(inp["timeStep"][b]), N, TL)
        P, q, Ao, l, u = orc.assemble_lmpc_qp(par, A, Bm, C, inp["x0"][b], inp["uOld"][b], SSsel, Qsel)
        orc.osqp_solve(P, q, Ao, l, u, polish=True)
        done += 1
    return done, time.perf_counter() - t0


def cpu_baseline(N, B, seconds=6.0, max_procs=None):
    """The reference's algorithm for this path as restated in oracle/ (NumPy regression + selection + assembly, restated
    OSQP at the reference's settings eps=1e-3 + polish -- what main.py does per step), timed on the host cores of this
    box for a fixed wall budget: first ONE core (the reference itself is single-threaded), then one
    single-threaded process per host core (count stated in the line) working through the same batch concurrently."""
    import multiprocessing as mp
    n1, t1 = _cpu_worker((0, min(seconds, 3.0), N, B))
    one = n1 / t1
    ncpu = os.cpu_count() or 1
    legs = []                                        # one single-threaded process per core: first on at most 64 cores, then on every hardware thread of the box
    for cores in sorted({min(ncpu, max_procs or ncpu, 64), min(ncpu, max_procs or ncpu)}):
        budget = seconds if cores <= 64 else min(seconds, 4.0)
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_worker, [(i * 7, budget, N, B) for i in range(cores)])
        done = sum(r[0] for r in res); busy = max(r[1] for r in res)
        legs.append({"value": done / busy, "unit": "solves/s", "cores": cores,
                     "sample": "%d full steps (a3-a19, restated OSQP eps=1e-3 + polish) of the bench batch in %.1f s on %d single-threaded processes" % (done, busy, cores)})
    best = max(legs, key=lambda l: l["value"])       # (round 4: 256 processes on the box's 256 hardware threads run SLOWER than 64 -- both legs are in the line, the better one is `value`)
    out = dict(value=best["value"], unit="solves/s", cores=best["cores"], kind="port", sample=best["sample"] + " (host has %d hardware threads)" % ncpu,
               legs=legs, single_core={"value": one, "unit": "solves/s", "cores": 1, "sample": "%d steps in %.1f s" % (n1, t1)})
    ref = os.path.join(ROOT, "profiles", "cpu_reference.json")
    if os.path.exists(ref):
        try:   # the reference's OWN classes timed in the build container (different box: /root/reference does not exist here)
            out["reference_classes_other_box"] = json.load(open(ref))
        except Exception:
            pass
    return out


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, with the launcher's environment."""
    import __graft_entry__ as ge
    if "--dry-run" not in argv:
        ge.build()                                  # once, before the ranks need the library
    port = _free_port()
    nonce = "%d-%d" % (os.getpid(), int(time.time()))           # names this launch's rendezvous file (parallel._rdzv_file)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LMPC_BENCH_SPAWNED="1",
                   LMPC_RDZV_NONCE=nonce)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else None))
    # rank 0's stdout is collected by a reader thread; the ranks are polled so that one rank dying cannot leave the others (and this
    # process) waiting in a collective for ever: the survivors are terminated, as a launcher would
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.1)
    if failed:
        time.sleep(2.0)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    rcs = [p.wait() for p in procs]
    sys.stdout.write(b"".join(buf).decode())
    sys.stdout.flush()
    if any(rcs):
        sys.exit("bench.py: rank exit codes %s" % rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="QPs per GPU per step (BASELINE configs[1]: 256)")
    ap.add_argument("--horizon", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sweep / extra configurations / rollout leg")
    ap.add_argument("--rollouts-per-gpu", type=int, default=1024)
    ap.add_argument("--rollouts-only", action="store_true", help="of the extras, run the closed-loop rollout leg only (developer runs)")
    ap.add_argument("--dry-run", action="store_true", help="start the ranks and the rendezvous only (no GPU needed): proves the N-rank launch path")
    ap.add_argument("--dry-run-rccl", action="store_true", help="communicator set-up, one barrier and one gather only -- on a box with fewer GPUs than ranks the ranks share "
                    "device 0 and RCCL refuses: the failure path of lmpc_comm_init with world > 1 (one JSON line with rccl_error, exit status 3)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])
    from racinglmpc_amd import parallel
    rank, world, local, addr, port = parallel.env_world()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    if args.dry_run:
        uid = parallel._rendezvous_id(rank, world, addr, port, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        # the host-side wait the ranks without extras use while rank 0 runs its extra configurations (no collective involved)
        if rank == 0:
            time.sleep(0.3); sig = parallel.host_signal("extras", port)
        else:
            parallel.host_wait("extras", port, timeout=60.0)
        if rank == 0:   # rank 0 has served the id to world - 1 peers: every rank started and reached the rendezvous
            time.sleep(0.3)
            try:
                os.remove(sig)
            except OSError:
                pass
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_at_rendezvous": world, "host_wait": "file",
                              "multi_rank_keys": ["per_rank_ms_per_step", "barrier_us", "timed_region_ms", "rccl_ranks", "exchange_seconds"]}))
        return

    import __graft_entry__ as ge
    ge.build()                                       # no-op when the library is current (file-locked against concurrent ranks)
    g = load_seed()
    N, B = args.horizon, args.batch
    def minimal_line(message):                       # what the watchdog / a failed communicator set-up prints: the headline's keys, this rank's own rate if it got that far
        local_dt = PROGRESS["t_sync"] - PROGRESS["t0"] if "t_sync" in PROGRESS and "t0" in PROGRESS else None
        return {"metric": "QP solves/sec (N=%d, nx=6, nu=2)" % N, "value": B * args.steps / local_dt if local_dt else None, "unit": "solves/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": local_dt / args.steps * 1e3 if local_dt else None, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "batch=%d LMPC QPs per GPU, N=%d" % (B, N), "ranks": world},
                "rccl_error": message, "note": "value, if any, is rank 0's own rate over its K steps (no cross-rank barrier closed the region)"}
    wd = Watchdog(rank, world, minimal_line)
    try:
        ctx = make_ctx(g, N, max(B, 1 if args.no_extras else args.rollouts_per_gpu), local)
    except Exception:                                # noqa: BLE001
        if not args.dry_run_rccl or local == 0:
            raise
        ctx = make_ctx(g, N, B, 0)                   # (--dry-run-rccl on a box with fewer GPUs than ranks: share device 0, RCCL will refuse)
    S = ctx.S
    wd.arm(60 if args.dry_run_rccl else HEADLINE_WATCHDOG_S, "communicator set-up (rendezvous, ncclCommInitRank)")
    try:
        comm = parallel.comm_from_env(ctx, force_rccl=os.environ.get("LMPC_BENCH_FORCE_DIST") == "1" or args.dry_run_rccl)
        if args.dry_run_rccl:
            wd.arm(60, "first barrier / gather of the communicator")
            comm.barrier()
            ranks = comm.allgather(np.array([rank], dtype=np.int64)).reshape(-1)
            wd.disarm()
            if rank == 0:
                print(json.dumps({"dry_run_rccl": True, "n_gpus": world, "rccl_ranks": ctx.comm_info()[1], "gathered_ranks": ranks.tolist()}), flush=True)
            comm.close(); ctx.close()
            return
    except Exception as e:                           # noqa: BLE001  (lmpc_comm_init failed: RCCL's own message is in the exception)
        wd.disarm()
        fail(rank, minimal_line, "%s: %s" % (type(e).__name__, str(e)[:400]))
    wd.arm(HEADLINE_WATCHDOG_S, "the timed region or its closing barrier (one all-reduce per rank)")
    if rank == 0 and world > 1:                       # a left-over "extras done" event of a killed launch (same parent, same port) must not release the other ranks early
        try:
            os.remove(parallel._rdzv_file(port) + ".extras")
        except OSError:
            pass
    inp = synth_batch(g, B, N, seed=1234 + rank)
    a, keep = ctx.step_dev_buffers(inp, diagnostics=False)      # the outputs MPC.solve produces (xPred, uPred, slack, lambda, s_T, zt, zt_u, SS_sel); no mu / residual dumps

    def sync_all():                                  # device drained on this rank, then on every rank
        ctx.sync()
        PROGRESS["t_sync"] = time.perf_counter()
        comm.barrier()

    dt, st = time_steps(ctx, B, a, args.steps, args.warmup, sync=sync_all)
    dt_ranks = comm.allgather(np.array([dt], dtype=np.float64)).reshape(-1)         # every rank's own clock around the same K steps
    dt = float(dt_ranks.max())
    tb = time.perf_counter()
    for _ in range(5):
        sync_all()                                     # what ends the timed region: one all-reduce enqueued on the stream + one drain (empty stream here)
    barrier_us = (time.perf_counter() - tb) / 5 * 1e6

    status = np.zeros(B, np.int32); iters = np.zeros(B, np.int32)
    ctx.dev_download(a.status, status); ctx.dev_download(a.iters, iters)
    n_ok = int(comm.allgather(np.array([np.sum(status == 0)], dtype=np.int64)).sum())
    comm_info = ctx.comm_info()
    wd.disarm()                                      # the headline is complete: every collective it needs has returned

    out = None
    if rank == 0:
        ms_solve = st.ms_solve / max(st.n_solve_timed, 1); ms_reg = st.ms_regress / st.n_regress_timed if st.n_regress_timed else None      # (None: fused step, no regression kernel)
        bytes_per_solve = 8 * (18 * N + 92)                     # SURVEY 8(d) B_solve: compulsory in+out per full step
        achieved = B * bytes_per_solve / (ms_solve * 1e-3) / 1e9
        nw = ctx.solver_waves(B)
        kname = "lmpc_solve_kernel_mw<%d,%d,%d>" % (N, S, nw) if nw > 1 else "lmpc_solve_kernel<%d,%d>" % (N, S)
        traffic = None; counters = None; prof_commit = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("lmpc_solve_kernel_bytes_per_launch_B%d_N%d" % (B, N))
                counters = tj.get("lmpc_solve_kernel_counters_B%d_N%d" % (B, N))
                prof_commit = tj.get("commit")
            except Exception:
                traffic = None
        out = {
            "metric": "QP solves/sec (N=%d, nx=6, nu=2)" % N, "value": world * B * args.steps / dt, "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "timed_region_ms": dt * 1e3, "per_rank_ms_per_step": [float(v) / args.steps * 1e3 for v in dt_ranks], "barrier_us": barrier_us,
            "rccl_ranks": comm_info[1] if comm_info[2] else 0,
            "config": {"workload": "batch=%d LMPC QPs per GPU, N=%d, fixed safe-set (4x PID seed lap, 48 points from 4 laps), full step a3-a19" % (B, N),
                       "batch_per_gpu": B, "N": N, "numSS_points": S, "laps_scanned": 4, "rows_per_lap": 1000,
                       "solver": "Riccati-structured primal-dual interior point to certified optimum (gap<1e-11, res<1e-9)",
                       "ranks": world, "collective_backend": "rccl" if comm_info[2] else "none", "rccl_ranks": comm_info[1] if comm_info[2] else 0},
            "solved_ok": n_ok, "solved_of": world * B, "ipm_iters_mean": float(iters.mean()), "ipm_iters_max": int(iters.max()),
            "kernel_ms": {"lmpc_solve_kernel": ms_solve, "lmpc_regress_kernel": ms_reg, "timed_launches": int(st.n_solve_timed),
                          "how": "HIP events on the launch stream around every %d-th launch of each kernel inside the timed region" % PROFILE_EVERY},
            "solver_only_solves_per_s": B / (ms_solve * 1e-3), "regression_only_solves_per_s": B / (ms_reg * 1e-3) if ms_reg else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": kname, "algorithmic_bytes_per_launch": B * bytes_per_solve,
                         "counters_commit": prof_commit,      # commit the PMC passes behind `traffic` / `fp64` were collected on (profiles/traffic.json)
                         # the other fraction SURVEY 8(d) asks for: FP64 work per launch from the rocprofv3 instruction-mix pass
                         # (profiles/, static) over the launch time measured live; plus VALU utilisation and the LDS bank-conflict rate
                         "fp64": None if not counters or "fp64_flop_per_launch" not in counters else {
                             "achieved": counters["fp64_flop_per_launch"] / (ms_solve * 1e-3) / 1e12, "peak": FP64_VEC_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": counters["fp64_flop_per_launch"] / (ms_solve * 1e-3) / 1e12 / FP64_VEC_PEAK_TFLOPS,
                             "valu_utilisation": counters.get("valu_utilisation"), "lds_bank_conflict_rate": counters.get("lds_bank_conflict_rate"),
                             "mfma_busy_cycles_per_launch": counters.get("SQ_VALU_MFMA_BUSY_CYCLES")},
                         "note": "dependent-issue-latency bound path (one Newton recursion per QP); compulsory HBM traffic is 2.46 KB per solve (SURVEY 8(d)); "
                                 "the dominant extra in the measured traffic is the A, B, C hand-over from the regression kernel (5.2 KB per solve at N = 12, written "
                                 "by one kernel and read by the other), then the 288 KB lap store missing once per XCD L2 and the ssSel output; see DESIGN.md"},
        }
    for p in keep:
        ctx.dev_free(p)

    # ---- extra configurations (rank 0's GPU; the other ranks sleep on a host-side event meanwhile: no rank spins in a collective) ----------
    if not args.no_extras:
        host_event = world > 1 and parallel._is_local(addr)
        if rank != 0 and host_event:
            parallel.host_wait("extras", port)
        if rank == 0 and not args.rollouts_only:
            try:                                       # (an extra configuration that fails must not cost the headline line)
                sweep = {}
                for bb in (1, 8, 64, 256, 512, 1024, 2048, 4096, 8192):
                    r = run_config(g, N, bb, local, steps=20 if bb <= 1024 else 8, warmup=3)
                    sweep[str(bb)] = {k: r[k] for k in ("solves_per_s", "solved_ok", "ipm_iters_max", "waves_per_qp")}
                out["sweep"] = sweep
                laps30 = pid_laps(ctx, g, 30)
                out["config_batch4096_30laps"] = dict(run_config(g, N, 4096, local, steps=8, warmup=2, laps=laps30, query_lap=laps30[29], max_laps=40, max_lap_len=1024),
                                                      note="BASELINE configs[2]: 30 PID laps (vt = 0.6 + 0.02 i) in both stores, reference semantics = the 4 fastest are used")
                out["config_batch4096_30laps_wide"] = dict(
                    run_config(g, N, 4096, local, steps=5, warmup=2, laps=laps30, query_lap=laps30[29], max_laps=40, max_lap_len=1024, numSS_it=8, numSS_points=96, trToUse=8),
                    note="an intermediate point of SURVEY 8(d)'s scan-heavy variant (8 laps used by regression and safe set, 96 safe-set points, 30 laps stored); "
                         "more than 58 safe-set points: several terminal-block columns per lane")
                out["config_batch4096_30laps_stress"] = dict(
                    run_config(g, N, 4096, local, steps=3, warmup=1, laps=laps30, query_lap=laps30[29], max_laps=40, max_lap_len=1024, numSS_it=30, numSS_points=360, trToUse=30),
                    note="SURVEY 8(d) scan-heavy stress variant as stated: numSS_it = trToUse = 30 (every stored lap in the regression and in the safe set), "
                         "numSS_Points = 360: 366 terminal-block columns, six per lane")
                out["config_N40_batch1024"] = dict(run_config(g, 40, 1024, local, steps=5, warmup=2), note="BASELINE configs[4]")
                out["pipelined_batches"] = {"2": pipelined_leg(g, N, B, local, steps=20, warmup=3, depth=2), "3": pipelined_leg(g, N, B, local, steps=20, warmup=3, depth=3)}
            except Exception as e:                     # noqa: BLE001
                out["extras_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
        if rank == 0:
            if host_event:
                sig_path = parallel.host_signal("extras", port)
        # The multi-rank form of this leg (one all-gather per lap between different devices) has never run on hardware before the driver's own
        # multi-GPU run: a collective that one rank never enters would hang the job and cost the headline line.  Every rank therefore arms a
        # watchdog; if the leg is not back in time, rank 0 prints the line it has -- still ONE line, still the last thing on stdout -- and all exit.
        watchdog = None
        if world > 1:
            import threading

            def _bail():                              # (the headline itself is complete here: the line is printed in full, but the job still ends non-zero)
                if rank == 0:
                    out["config_rollouts"] = {"error": "multi-rank rollout leg did not return within %d s (watchdog)" % ROLLOUT_WATCHDOG_S}
                    out["rccl_error"] = out["config_rollouts"]["error"]
                    sys.stdout.flush()
                    print(json.dumps(out), flush=True)
                os._exit(3)
            watchdog = threading.Timer(ROLLOUT_WATCHDOG_S, _bail); watchdog.daemon = True; watchdog.start()
        try:                                         # (deterministic failures -- e.g. too few valid laps -- occur on every rank alike)
            leg = rollout_leg(g, comm, ctx, args.rollouts_per_gpu)
        except RuntimeError as e:
            leg = {"error": str(e)[:300]}
        if watchdog is not None:
            watchdog.cancel()
        if rank == 0:
            out["config_rollouts"] = dict(leg, note="BASELINE configs[3]: closed-loop LMPC laps sharded over the ranks, lap stores replicated, one all-gather per lap")
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, B) if world == 1 else None
    def drain_stdio():   # librccl announces its version through C stdio (buffered when stdout is a pipe): push it out NOW
        try:
            import ctypes
            sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    # the JSON line must be the LAST thing on the job's stdout: every rank drains its buffers, THEN the ranks meet at the barrier, then rank 0 prints
    drain_stdio()
    comm.barrier()
    if rank == 0 and not args.no_extras and world > 1 and parallel._is_local(addr):
        try:
            os.remove(sig_path)
        except (OSError, NameError):
            pass
    comm.close()
    ctx.close()
    drain_stdio()
    if rank == 0 and world > 1:
        time.sleep(0.2)                             # the other ranks' exit paths (their stdout is the same pipe)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
