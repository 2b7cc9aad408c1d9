"""Host-side cross-checks of the device-resident rollout loop (test infrastructure, not product):

* plant_step: vectorised NumPy restatement of Simulator.dynModel (SysModel.py:56-147) for a batch of cars -- used to generate
  many-lap safe sets for the batch tests and to cross-check lmpc_plant_kernel;
* run_lap_host: the closed loop driven from the host (one lmpc_step_batch per simulated step, NumPy plant), same contract as
  BatchedRollouts.run_lap_device.
"""
import numpy as np


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



def run_lap_host(ctx, track, x0, xLin0, uLin0, max_steps=400, seed=0):
    """x0 (B,6) start states; xLin0 (N+1,6) / uLin0 (N,2) first linearisation trajectory (LMPC.addTrajectory :431-433).
    Returns list of (x (T,6), u (T,2), x_glob (T,6)) per rollout, the lap ending when s > TrackLength (SysModel.py:45)."""
    track = np.asarray(track, float); TL = float(track[-1, 3] + track[-1, 4]); rng = np.random.default_rng(seed)
    N = ctx.N; B = x0.shape[0]
    per = lambda a: np.tile(np.asarray(a, float)[None], (B, 1, 1)) if np.asarray(a).ndim == 2 else np.asarray(a, float)
    x = np.array(x0, float); xg = x.copy()
    xLin = per(xLin0).copy(); uLin = per(uLin0).copy()
    uOld = np.zeros((B, 2)); zt = np.tile(np.array([0.0, 0.0, 0.0, 0.0, 10.0, 0.0]), (B, 1))
    xPP = np.zeros((B, N + 1, 6)); hasPred = np.zeros(B, np.int32)
    hist_x, hist_u, hist_g = [], [], []
    done_at = -np.ones(B, dtype=np.int64)
    for t in range(max_steps):
        out = ctx.step_batch(x, xLin, uLin, uOld, zt=zt, xPredPrev=xPP, hasPred=hasPred, timeStep=np.full(B, t, np.int32))
        u = out["uPred"][:, 0, :].copy()
        hist_x.append(x.copy()); hist_u.append(u); hist_g.append(xg.copy())
        x, xg = plant_step(track, x, xg, u, rng.standard_normal((B, 3)))
        xPP = out["xPred"]; hasPred[:] = 1
        xLin = np.concatenate([out["xPred"][:, 1:, :], out["ztNext"][:, None, :]], axis=1)
        uLin = np.concatenate([out["uPred"][:, 1:, :], out["ztuNext"][:, None, :]], axis=1)
        uOld = u; zt = out["ztNext"].copy()
        newly = (done_at < 0) & (x[:, 4] > TL)
        done_at[newly] = t + 1
        if np.all(done_at >= 0):
            break
    X = np.stack(hist_x, axis=1); U = np.stack(hist_u, axis=1); G = np.stack(hist_g, axis=1)
    return [(X[b, :int(done_at[b])], U[b, :int(done_at[b])], G[b, :int(done_at[b])]) for b in range(B) if done_at[b] >= 0]
