"""Batched closed-loop LMPC rollouts: B independent cars share one safe set and are advanced in lock-step, one
lmpc_step_batch (GPU) per simulated time step.  Caller side of the hot path (reference SysModel.Simulator.sim,
SysModel.py:22-54); the plant (reference dynModel, SysModel.py:56-147) is evaluated on the host, vectorised over
the batch -- moving it into a HIP kernel is the next row of SURVEY 8(f).

Rank-local: each rank owns a contiguous shard of the rollouts (parallel.shard) and its own device context; after a
lap the ranks exchange their fastest laps once (parallel.exchange_laps) and apply identical inserts.
"""
import numpy as np

from . import parallel


def plant_step(track, x, x_glob, u, noise):
    """Vectorised restatement of Simulator.dynModel (SysModel.py:56-147) for a batch: x, x_glob (B,6), u (B,2),
    noise (B,3) ~ N(0,1) draws for (vx, vy, wz).  100 forward-Euler sub-steps of 1 ms."""
    m = 1.98; lf = 0.125; lr = 0.125; Iz = 0.024
    Df = 0.8 * m * 9.81 / 2.0; Cf = 1.25; Bf = 1.0
    Dr = 0.8 * m * 9.81 / 2.0; Cr = 1.25; Br = 1.0
    deltaT = 0.001
    TL = track[-1, 3] + track[-1, 4]
    delta, a = u[:, 0], u[:, 1]
    psi, X, Y = x_glob[:, 3].copy(), x_glob[:, 4].copy(), x_glob[:, 5].copy()
    vx, vy, wz, epsi, s, ey = [x[:, i].copy() for i in range(6)]
    cum, seglen, curv = track[:, 3], track[:, 4], track[:, 5]
    for _ in range(100):
        alpha_f = delta - np.arctan2(vy + lf * wz, vx)
        alpha_r = - np.arctan2(vy - lf * wz, vx)
        Fyf = Df * np.sin(Cf * np.arctan(Bf * alpha_f))
        Fyr = Dr * np.sin(Cr * np.arctan(Br * alpha_r))
        nvx = vx + deltaT * (a - 1 / m * Fyf * np.sin(delta) + wz * vy)
        nvy = vy + deltaT * (1 / m * (Fyf * np.cos(delta) + Fyr) - wz * vx)
        nwz = wz + deltaT * (1 / Iz * (lf * Fyf * np.cos(delta) - lr * Fyr))
        npsi = psi + deltaT * (wz)
        nX = X + deltaT * ((vx * np.cos(psi) - vy * np.sin(psi)))
        nY = Y + deltaT * (vx * np.sin(psi) + vy * np.cos(psi))
        sw = np.where(s > TL, s - TL * np.floor(s / TL), s)               # Map.curvature wrap (Track.py:298-300)
        sw = np.where(sw > TL, sw - TL, sw)
        seg = np.clip(np.searchsorted(cum, sw, side="right") - 1, 0, len(cum) - 1)
        cur = curv[seg]
        nepsi = epsi + deltaT * (wz - (vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey) * cur)
        ns = s + deltaT * ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey))
        ney = ey + deltaT * (vx * np.sin(epsi) + vy * np.cos(epsi))
        vx, vy, wz, epsi, s, ey, psi, X, Y = nvx, nvy, nwz, nepsi, ns, ney, npsi, nX, nY
    nz = np.stack([np.clip(noise[:, 0] * 0.01, -0.05, 0.05), np.clip(noise[:, 1] * 0.01, -0.05, 0.05), np.clip(noise[:, 2] * 0.005, -0.05, 0.05)], axis=1)
    xn = np.stack([vx + 0.01 * nz[:, 0], vy + 0.01 * nz[:, 1], wz + 0.01 * nz[:, 2], epsi, s, ey], axis=1)
    xg = np.stack([vx, vy, wz, psi, X, Y], axis=1)
    return xn, xg


class BatchedRollouts:
    """B closed-loop LMPC laps against a shared safe set (one GPU context, one rank)."""

    def __init__(self, ctx, track, seed=0):
        self.ctx, self.track = ctx, np.asarray(track, float)
        self.TL = float(self.track[-1, 3] + self.track[-1, 4])
        self.rng = np.random.default_rng(seed)
        self.last_status = None

    @staticmethod
    def _per_rollout(a, B):
        a = np.asarray(a, float)
        return np.tile(a[None], (B, 1, 1)) if a.ndim == 2 else a

    def run_lap_device(self, x0, xLin0, uLin0, xglob0=None, max_steps=400, ext=0, on_ext=None):
        """The whole lap (controller steps, plant, bookkeeping) stays on the GPU (lmpc_rollout_*): four kernel launches
        per simulated step, no host round trip.  If ext > 0 the lap pauses after `ext` steps and on_ext(X, U) is called
        with the first ext states/inputs (ext, B, .) -- the hook that plays LMPC.addPoint for the stored laps.
        Returns (laps [(x, u, x_glob, final12)], where final12 = state + global state right after the finish line)."""
        B = x0.shape[0]
        xl = self._per_rollout(xLin0, B); ul = self._per_rollout(uLin0, B)
        noise = self.rng.standard_normal((max_steps, B, 3))
        self.ctx.rollout_begin(x0, x0 if xglob0 is None else xglob0, xl, ul, noise)
        if ext > 0:
            t, nd = self.ctx.rollout_run(ext)
            X, U, G, done, st, fx, fg = self.ctx.rollout_fetch(0, t)
            if on_ext is not None:
                on_ext(X, U)
        t, nd = self.ctx.rollout_run(max_steps)
        X, U, G, done, st, fx, fg = self.ctx.rollout_fetch(0, t)
        self.ctx.rollout_end()
        self.last_status = st
        laps = []
        for b in range(B):
            T = int(done[b]) if done[b] >= 0 else X.shape[0]
            laps.append((X[:T, b].copy(), U[:T, b].copy(), G[:T, b].copy(), np.concatenate([fx[b], fg[b]])))
        return laps

    def run_lap(self, x0, xLin0, uLin0, max_steps=400):
        """Host-driven variant (one lmpc_step_batch per step, NumPy plant); same contract, used to cross-check the device loop.
        x0 (B,6) start states; xLin0 (N+1,6) / uLin0 (N,2) first linearisation trajectory (LMPC.addTrajectory :431-433).
        Returns list of (x (T,6), u (T,2), x_glob (T,6)) per rollout, the lap ending when s > TrackLength (SysModel.py:45)."""
        ctx = self.ctx; N = ctx.N; B = x0.shape[0]
        x = np.array(x0, float); xg = x.copy()
        xLin = self._per_rollout(xLin0, B).copy(); uLin = self._per_rollout(uLin0, B).copy()
        uOld = np.zeros((B, 2)); zt = np.tile(np.array([0.0, 0.0, 0.0, 0.0, 10.0, 0.0]), (B, 1))
        xPP = np.zeros((B, N + 1, 6)); hasPred = np.zeros(B, np.int32)
        hist_x, hist_u, hist_g = [], [], []
        done_at = -np.ones(B, dtype=np.int64)
        for t in range(max_steps):
            out = ctx.step_batch(x, xLin, uLin, uOld, zt=zt, xPredPrev=xPP, hasPred=hasPred, timeStep=np.full(B, t, np.int32))
            u = out["uPred"][:, 0, :].copy()
            hist_x.append(x.copy()); hist_u.append(u); hist_g.append(xg.copy())
            x, xg = plant_step(self.track, x, xg, u, self.rng.standard_normal((B, 3)))
            xPP = out["xPred"]; hasPred[:] = 1
            xLin = np.concatenate([out["xPred"][:, 1:, :], out["ztNext"][:, None, :]], axis=1)
            uLin = np.concatenate([out["uPred"][:, 1:, :], out["ztuNext"][:, None, :]], axis=1)
            uOld = u; zt = out["ztNext"].copy()
            newly = (done_at < 0) & (x[:, 4] > self.TL)
            done_at[newly] = t + 1
            if np.all(done_at >= 0):
                break
        X = np.stack(hist_x, axis=1); U = np.stack(hist_u, axis=1); G = np.stack(hist_g, axis=1)
        laps = []
        for b in range(B):
            T = int(done_at[b]) if done_at[b] >= 0 else X.shape[1]
            laps.append((X[b, :T], U[b, :T], G[b, :T]))
        return laps


class LmpcGeneration:
    """Iterated batched LMPC over all ranks.  Generation g: every rank runs its shard of rollouts for one lap; the K
    globally fastest laps are exchanged (one all-gather) and appended to the model store and the safe set of every rank in
    identical order.  Generation g+1 starts its rollouts from the states in which those K laps crossed the finish line
    (the reference's xF, SysModel.py:50), and the first `ext` steps of the rollout continuing lap k extend stored lap k
    past the finish line -- the batched form of LMPC.addPoint (:466-474), without which no safe-set point lies beyond
    the line and the terminal constraint would stop the cars in front of it."""

    def __init__(self, rollouts, total_rollouts, K=4, T_max=400, ext=40, rank=0, world=1):
        self.ro, self.total, self.K, self.T_max, self.ext, self.rank, self.world = rollouts, total_rollouts, K, T_max, ext, rank, world
        self.lo, self.hi = parallel.shard(total_rollouts, rank, world)
        self.parents = None            # [(x, u, x_glob, final12, stored_lap_index)] of the previous generation

    def run(self, x0_all=None, xLin0=None, uLin0=None):
        ro, ctx, K, N = self.ro, self.ro.ctx, self.K, self.ro.ctx.N
        lo, hi = self.lo, self.hi
        gb = np.arange(lo, hi)
        if self.parents is None:
            x0 = x0_all[lo:hi]; xg0 = x0.copy()
            xl, ul = xLin0, uLin0
            on_ext = None; ext = 0
        else:
            par = gb % K
            fin = np.stack([self.parents[k][3] for k in range(K)])
            x0 = fin[par, 0:6].copy(); x0[:, 4] -= ro.TL                           # xF = x_cl[-1] - [0,0,0,0,TrackLength,0]
            xg0 = fin[par, 6:12].copy()
            xl = np.stack([self.parents[k][0][1:N + 2] for k in par]); ul = np.stack([self.parents[k][1][1:N + 1] for k in par])
            ext = self.ext

            def on_ext(X, U):
                # rollout with global index k continues lap k: its first ext points extend stored lap k (rank 0 owns them)
                buf = np.zeros((K, ext, 8))
                for k in range(K):
                    if lo <= k < hi:
                        buf[k, :, 0:6] = X[:ext, k - lo]; buf[k, :, 6:8] = U[:ext, k - lo]
                buf = parallel.broadcast_array(buf, src=0)
                for k in range(K):
                    ctx.ss_extend_lap(self.parents[k][4], buf[k, :, 0:6], buf[k, :, 6:8])
        laps = ro.run_lap_device(x0, xl, ul, xglob0=xg0, max_steps=self.T_max, ext=ext, on_ext=on_ext) if hi > lo else []
        best = parallel.exchange_laps(laps, K, self.T_max)
        self.parents = []
        for x, u, xg, src, T, extra in best:
            ctx.ss_add_trajectory(x, u)
            ctx.model_add_trajectory(x, u)
            self.parents.append((x, u, xg, extra[:12], self._n_ss() - 1))
        return best

    def _n_ss(self):
        import ctypes as C
        n = C.c_int()
        self.ro.ctx.lib.lmpc_ss_num_laps(self.ro.ctx._h, C.byref(n))
        return n.value


def lap_and_exchange(rollouts, x0_all, xLin0, uLin0, K, T_max, rank=0, world=1, device=False):
    """One generation over all ranks (first generation form: explicit start states)."""
    lo, hi = parallel.shard(x0_all.shape[0], rank, world)
    if device:
        laps = rollouts.run_lap_device(x0_all[lo:hi], xLin0, uLin0, max_steps=T_max) if hi > lo else []
    else:
        laps = rollouts.run_lap(x0_all[lo:hi], xLin0, uLin0, max_steps=T_max) if hi > lo else []
    best = parallel.exchange_laps(laps, K, T_max)
    for x, u, xg, src, T, extra in best:
        rollouts.ctx.ss_add_trajectory(x, u)
        rollouts.ctx.model_add_trajectory(x, u)
    return best
