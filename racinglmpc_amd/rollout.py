"""Batched closed-loop LMPC rollouts: B independent cars share one safe set and are advanced in lock-step ON THE DEVICE
(lmpc_rollout_*: regression + solve + plant kernel + bookkeeping per simulated step, no host round trip).  Caller side of
the hot path (reference SysModel.Simulator.sim, SysModel.py:22-54; plant: Simulator.dynModel, SysModel.py:56-147).

Rank-local: each rank owns a contiguous shard of the rollouts (parallel.shard) and its own device context; after a lap the
ranks exchange their fastest VALID laps once (lmpc_rollout_exchange: device-packed records, one RCCL all-gather) and apply
identical inserts.  A lap is valid when the car crossed the finish line and no status bit other than LMPC_ST_INEXACT was
raised up to the crossing step; anything else never reaches a lap store -- neither as a new lap nor as the rows that extend a
stored lap past the finish line (the batched LMPC.addPoint): those are checked for status bits and finiteness first.
"""
import numpy as np

from . import _capi, parallel


class BatchedRollouts:
    """B closed-loop LMPC laps against a shared safe set (one GPU context, one rank)."""

    def __init__(self, ctx, track, seed=0, global_noise=False):
        """global_noise: the plant noise of a lap is drawn for ALL rollouts of the job (same seed on every rank) and this rank keeps the columns
        of its shard (`noise_shard` = (lo, hi, total), set by LmpcGeneration) -- a rollout then sees the same draws however the job is split over
        ranks.  Default: every rank draws for its own shard only (seed per rank)."""
        self.ctx, self.track = ctx, np.asarray(track, float)
        self.TL = float(self.track[-1, 3] + self.track[-1, 4])
        self.rng = np.random.default_rng(seed)
        self.global_noise, self.noise_shard = bool(global_noise), None
        self.last_status = None
        self.last_done = None
        self._pre = None               # (shape key, generator state before the draw, future): the NEXT lap's plant noise, drawn by a worker thread while this lap runs

    def _draw_noise(self, max_steps, B, prefetch_only=False):
        """Plant noise of one lap, (max_steps, B, 3) N(0, 1) draws.  A generation loop asks for the same shape lap after lap, and 1.2 M draws are ~15 ms of one
        host core -- a tenth of a 1024-rollout lap on the GPU: the next lap's array is drawn by a worker thread while the device runs this one (NumPy releases
        the GIL inside the fill).  The generator is consumed in exactly the order it would be without the prefetch: a prefetched array of another shape is
        discarded together with its draws (the generator state from before the draw is restored)."""
        if self.global_noise and self.noise_shard is not None:
            lo, hi, total = self.noise_shard
            assert hi - lo == B, (lo, hi, B)
            key = (max_steps, total, lo, hi)
            draw = lambda: np.ascontiguousarray(self.rng.standard_normal((max_steps, total, 3))[:, lo:hi])
        else:
            key = (max_steps, B)
            draw = lambda: self.rng.standard_normal((max_steps, B, 3))
        if prefetch_only:                          # (prefetch_noise: start the draw of the FIRST lap; the generator is consumed in the same order)
            if self._pre is None:
                from concurrent.futures import ThreadPoolExecutor
                if not hasattr(self, "_pool"):
                    self._pool = ThreadPoolExecutor(max_workers=1)
                self._pre = (key, self.rng.bit_generator.state, self._pool.submit(draw))
            return None
        noise = None
        if self._pre is not None:
            pkey, state, fut = self._pre
            self._pre = None
            got = fut.result()
            if pkey == key:
                noise = got
            else:
                self.rng.bit_generator.state = state
        if noise is None:
            noise = draw()
        from concurrent.futures import ThreadPoolExecutor
        if not hasattr(self, "_pool"):
            self._pool = ThreadPoolExecutor(max_workers=1)
        self._pre = (key, self.rng.bit_generator.state, self._pool.submit(draw))
        return noise

    def close(self):
        """End of a generation loop: wait for the prefetched draw nobody will use, give its draws back (the generator state from before that draw is restored, so
        `rng` is where a run without prefetching leaves it) and shut the worker thread down.  Between begin() calls `rng` belongs to the prefetcher: the worker
        thread is drawing from it -- do not touch it from the caller's thread before close()."""
        if self._pre is not None:
            _, state, fut = self._pre
            self._pre = None
            fut.result()
            self.rng.bit_generator.state = state
        pool = self.__dict__.pop("_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)

    def prefetch_noise(self, max_steps, B, wait=False):
        """Draw the plant noise of the first lap ahead of time (worker thread; same draws, same order as without the call): the synthetic disturbance is input
        data of a closed-loop run, and 1.2 M normal draws are 15-20 ms of host time that would otherwise open the first lap."""
        self._draw_noise(max_steps, B, prefetch_only=True)
        if wait and self._pre is not None:
            self._pre[2].result()

    @staticmethod
    def _per_rollout(a, B):
        a = np.asarray(a, float)
        return np.tile(a[None], (B, 1, 1)) if a.ndim == 2 else a

    def begin(self, x0, xLin0, uLin0, xglob0=None, max_steps=400):
        B = x0.shape[0]
        xl = self._per_rollout(xLin0, B); ul = self._per_rollout(uLin0, B)
        noise = self._draw_noise(max_steps, B)
        self.ctx.rollout_begin(x0, x0 if xglob0 is None else xglob0, xl, ul, noise)

    def run_lap_device(self, x0, xLin0, uLin0, xglob0=None, max_steps=400, ext=0, on_ext=None, keep_invalid=False):
        """The whole lap stays on the GPU: four kernel launches per simulated step.  If ext > 0 the lap pauses after `ext` steps
        and on_ext(X, U) is called with the states/inputs logged so far (n <= ext rows, n x B x .) -- the hook that plays
        LMPC.addPoint for the stored laps.  Returns the VALID laps [(x, u, x_glob, final12, done_at, status)] (final12 = state +
        global state right after the finish line); with keep_invalid the unfinished / flagged ones are returned too, their
        done_at / status telling them apart."""
        B = x0.shape[0]
        self.begin(x0, xLin0, uLin0, xglob0, max_steps)
        if ext > 0:
            t, nd = self.ctx.rollout_run(ext)
            X, U, G, done, st, fx, fg = self.ctx.rollout_fetch(0, t)
            if on_ext is not None:
                on_ext(X, U)
        t, nd = self.ctx.rollout_run(max_steps)
        X, U, G, done, st, fx, fg = self.ctx.rollout_fetch(0, t)
        self.ctx.rollout_end()
        self.last_status, self.last_done = st, done
        laps = []
        for b in range(B):
            valid = done[b] >= 0 and (st[b] & ~_capi.ST_INEXACT) == 0
            if not (valid or keep_invalid):
                continue
            T = int(done[b]) if done[b] >= 0 else X.shape[0]
            laps.append((X[:T, b].copy(), U[:T, b].copy(), G[:T, b].copy(), np.concatenate([fx[b], fg[b]]), int(done[b]), int(st[b])))
        return laps


class LmpcGeneration:
    """Iterated batched LMPC over all ranks.  Generation g: every rank runs its shard of rollouts for one lap; the K
    globally fastest valid laps are exchanged (one all-gather of device-packed records) and appended to the model store and the
    safe set of every rank in identical order.  Generation g+1 starts its rollouts from the states in which those K laps crossed
    the finish line (the reference's xF, SysModel.py:50), and the first `ext` steps of the rollout continuing lap k extend stored
    lap k past the finish line -- the batched form of LMPC.addPoint (:466-474), without which no safe-set point lies beyond the
    line and the terminal constraint would stop the cars in front of it."""

    def __init__(self, rollouts, total_rollouts, K=4, T_max=400, ext=40, comm=None):
        self.ro, self.total, self.K, self.T_max, self.ext = rollouts, total_rollouts, K, T_max, ext
        self.comm = comm or parallel.LocalComm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.lo, self.hi = parallel.shard(total_rollouts, self.rank, self.world)
        if self.hi <= self.lo:
            raise ValueError("every rank needs at least one rollout (total %d, world %d)" % (total_rollouts, self.world))
        rollouts.noise_shard = (self.lo, self.hi, total_rollouts)
        self.parents = None            # [(x, u, x_glob, final12, stored_lap_index)] of the previous generation
        self.last_exchange = None      # (bytes per rank, seconds) of the last all-gather
        self.last_status = self.last_done = None
        self.skipped_extensions = []   # [(k, status bits)]: stored laps NOT extended in the last generation because the continuing rollout was flagged
        self.open_laps = set()         # stored laps that end at the finish line for good (their extension was skipped): kept out of the safe-set selection
        self.step_hook = None          # developer / test hook: called as step_hook(t) after every simulated step (the lap is then advanced one step per call:
                                       # tests/test_gpu_closed_loop.py samples the session's QPs through Context.debug_rollout_qp)

    def _advance(self, n):
        """ctx.rollout_run(n) -- or, with a step hook, n single steps with the hook in between (stops when every rollout has finished, as rollout_run does)."""
        ctx = self.ro.ctx
        if self.step_hook is None:
            return ctx.rollout_run(n)
        t_end = min(self.T_max, ctx._ro_t + n)
        while True:
            t, nd = ctx.rollout_run(1)
            self.step_hook(t)
            if nd >= self.hi - self.lo or t >= t_end:
                return t, nd

    def close(self):
        """After the last run(): BatchedRollouts.close() (prefetched noise returned to the generator, worker thread ended)."""
        self.ro.close()

    def prepare(self, wait=True):
        """Optional, before the first run(): the first lap's plant noise is drawn now (BatchedRollouts.prefetch_noise) instead of inside the lap."""
        self.ro.prefetch_noise(self.T_max, self.hi - self.lo, wait=wait)

    def run(self, x0_all=None, xLin0=None, uLin0=None):
        import time
        ro, ctx, K, N = self.ro, self.ro.ctx, self.K, self.ro.ctx.N
        lo, hi = self.lo, self.hi
        gb = np.arange(lo, hi)
        if self.parents is None:
            x0 = x0_all[lo:hi]; xg0 = x0.copy()
            xl, ul = xLin0, uLin0
            ext = 0
        else:
            par = gb % K
            fin = np.stack([self.parents[k][3] for k in range(K)])
            x0 = fin[par, 0:6].copy(); x0[:, 4] -= ro.TL                           # xF = x_cl[-1] - [0,0,0,0,TrackLength,0]
            xg0 = fin[par, 6:12].copy()
            xl = np.stack([self.parents[k][0][1:N + 2] for k in par]); ul = np.stack([self.parents[k][1][1:N + 1] for k in par])
            ext = self.ext
        ro.begin(x0, xl, ul, xg0, self.T_max)
        undo = []                      # (stored lap, rows before this generation's extension): a generation completes or leaves the safe set as it found it
        open_before = set(self.open_laps)
        try:
            self._apply_selection()    # (the laps added by the previous generation take part from the first step on)
            if ext > 0:
                # the rollout with GLOBAL index k (k < K) continues stored lap k: its first ext points extend that lap on every rank.
                # Each row is taken from the rank whose shard holds rollout k, together with that rollout's accumulated status bits: a row of a
                # flagged rollout (anything but INEXACT) or a non-finite row never reaches a lap store -- decided from the gathered data, hence
                # identically on every rank.
                t, _ = self._advance(ext)
                X, U, G, done, st, fx, fg = ctx.rollout_fetch(0, t)
                n = min(t, ext)
                buf = np.zeros((K, ext, 9)); mask = np.zeros(K, dtype=np.int64)
                for k in range(K):
                    if lo <= k < hi:
                        buf[k, :n, 0:6] = X[:n, k - lo]; buf[k, :n, 6:8] = U[:n, k - lo]; buf[k, :, 8] = float(st[k - lo]); mask[k] = 1
                rows, owned = parallel.gather_owned_rows(buf, mask, self.comm)
                nmin = int(self.comm.allreduce_max(-float(n))[0] * -1)                 # rows every owner really logged
                self.skipped_extensions = []
                for k in range(K):
                    if not owned[k] or nmin <= 0:
                        continue                                                   # (no rows to append: the lap stays open, see below)
                    clean = (int(rows[k, 0, 8]) & ~_capi.ST_INEXACT) == 0 and bool(np.all(np.isfinite(rows[k, :nmin, 0:8])))
                    if not clean:
                        self.skipped_extensions.append((k, int(rows[k, 0, 8])))
                        continue
                    lap = self.parents[k][4]
                    undo.append((lap, ctx.ss_lap_rows(lap)))
                    ctx.ss_extend_lap(lap, rows[k, :nmin, 0:6], rows[k, :nmin, 6:8])
                # A stored lap whose extension was skipped ends AT the finish line: a car approaching the line would find its 13-row window running
                # past the lap's end (LMPC_ST_WINDOW -- the reference's IndexError, :497) on every step there.  Such a lap stays in the store (and in the
                # regression data) but is left out of the safe-set selection from now on: the numSS_it fastest laps among the others are used --
                # decided from the gathered rows, hence identically on every rank.
                self.open_laps |= {self.parents[k][4] for k, _ in self.skipped_extensions} | {self.parents[k][4] for k in range(K) if nmin <= 0 or not owned[k]}
                self._apply_selection()
            self._advance(self.T_max)
            _, _, _, self.last_done, self.last_status, _, _ = ctx.rollout_fetch(0, 0)     # per-rollout finish step / accumulated status bits
            t0 = time.perf_counter()
            # The device-packed exchange needs the context's own RCCL communicator spanning the same world as `comm` (or a single process);
            # any other communicator object (parallel.py: "any object with rank / world / allgather / ...") takes the host-packed path.
            c_rank, c_world, c_rccl = ctx.comm_info()
            if self.world > 1 and c_rccl and c_world != self.world:
                raise RuntimeError("communicator world %d does not match the context's RCCL communicator (world %d)" % (self.world, c_world))
            device_path = self.world == 1 or (c_rccl and c_world == self.world)
            recs, lens, n_valid = ctx.rollout_exchange(K, self.T_max) if device_path else self._host_exchange()
            self.last_exchange = (recs[0].nbytes + lens[0].nbytes, time.perf_counter() - t0)
            ctx.rollout_end()
            best = parallel.top_k(recs, lens, K, self.T_max)
            if len(best) < K:
                raise RuntimeError("only %d valid laps among %d rollouts (need K = %d): the others did not finish within %d steps or were flagged"
                                   % (len(best), self.total, K, self.T_max))
        except Exception:
            for lap, rows_before in undo:
                ctx.ss_truncate_lap(lap, rows_before)
            self.open_laps = open_before
            # ... and the selection as it was: the library's own argsort(LapTime) when no lap was open before, else the choice that goes with the old set
            if open_before:
                self._apply_selection()
            else:
                ctx.ss_set_selected([])
            raise
        self.parents = []
        for x, u, xg, src, T, extra in best:
            ctx.ss_add_trajectory(x, u)
            ctx.model_add_trajectory(x, u)
            self.parents.append((x, u, xg, extra[:12], ctx.ss_num_laps() - 1))
        return best

    def _apply_selection(self):
        """Safe-set selection of this generation: the library's own choice -- the numSS_it fastest stored laps, argsort(LapTime) (:395, 402) -- unless
        some stored lap never got its extension past the finish line (`open_laps`): then the numSS_it fastest of the OTHER laps, handed over explicitly."""
        ctx = self.ro.ctx
        if not self.open_laps:
            return
        L = ctx.cfg.numSS_it
        usable = [l for l in range(ctx.ss_num_laps()) if l not in self.open_laps]
        if len(usable) < L:
            raise RuntimeError("fewer than numSS_it = %d stored laps have their extension past the finish line" % L)
        usable.sort(key=lambda l: (ctx.ss_lap_time(l), l))                  # stable argsort(LapTime)
        ctx.ss_set_selected(usable[:L])

    def _host_exchange(self):
        """Same exchange with host-packed records over a caller-supplied communicator (CPU tests)."""
        ctx = self.ro.ctx
        X, U, G, done, st, fx, fg = ctx.rollout_fetch(0, ctx._ro_t)
        valid = [b for b in range(X.shape[1]) if done[b] >= 0 and (st[b] & ~_capi.ST_INEXACT) == 0]
        laps = [(X[:done[b], b], U[:done[b], b], G[:done[b], b], np.concatenate([fx[b], fg[b]])) for b in valid]
        rec, ln = parallel.pack_laps(laps, self.K, self.T_max, ids=valid)        # (the record carries the rollout's index in the shard, as the device-packed records do)
        return self.comm.allgather(rec), self.comm.allgather(ln), len(laps)


def lap_and_exchange(rollouts, x0_all, xLin0, uLin0, K, T_max, comm=None):
    """One generation over all ranks (first generation form: explicit start states), device-resident."""
    gen = LmpcGeneration(rollouts, x0_all.shape[0], K=K, T_max=T_max, ext=0, comm=comm)
    return gen.run(x0_all, xLin0, uLin0)
