"""oracle/lmpc_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (NumPy FP64) restatement of the reference's per-time-step LMPC hot path, function by
function, each citing the reference file:line it follows (paths are under
/root/reference/src/).  The QP itself is handed to oracle/osqp_restated.c (a restatement of
the published OSQP algorithm; the real `osqp` / `cvxopt` packages are third-party, unpinned
(reference README.md:16-20) and not installable here).

Pinning status
  * regression / selection / assembly: pinned bit-for-bit by
    tests/golden/*.npz, which were produced by EXECUTING the reference's own classes
    (tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
  * QP solve (a17): PARITY WITH THE REAL osqp BINARY IS UNPINNED (no osqp, no upstream golden
    vectors).  Pinned instead by the solver-independent optimality certificate
    `kkt_certificate` below.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# Track  (fnc/simulator/Track.py)
# --------------------------------------------------------------------------------------

def _wrap(angle):
    # Track.py:365-373 / Utilities.py:31-39
    if angle < -np.pi:
        return 2 * np.pi + angle
    if angle > np.pi:
        return angle - 2 * np.pi
    return angle


def _sign(a):
    # Track.py:375-381
    return 1 if a >= 0 else -1


def make_track():
    """PointAndTangent table of the L-shaped track and TrackLength.  Track.py:31-133."""
    lengthCurve = 4.5
    spec = np.array([[1.0, 0],
                     [lengthCurve, lengthCurve / np.pi],
                     [lengthCurve / 2, -lengthCurve / np.pi],
                     [lengthCurve, lengthCurve / np.pi],
                     [lengthCurve / np.pi * 2, 0],
                     [lengthCurve / 2, lengthCurve / np.pi]])
    pt = np.zeros((spec.shape[0] + 1, 6))
    for i in range(spec.shape[0]):
        if spec[i, 1] == 0.0:                                   # straight, Track.py:57-76
            l = spec[i, 0]
            if i == 0:
                ang = 0
                x = 0 + l * np.cos(ang)
                y = 0 + l * np.sin(ang)
            else:
                ang = pt[i - 1, 2]
                x = pt[i - 1, 0] + l * np.cos(ang)
                y = pt[i - 1, 1] + l * np.sin(ang)
            psi = ang
            if i == 0:
                pt[i, :] = np.array([x, y, psi, pt[i, 3], l, 0])
            else:
                pt[i, :] = np.array([x, y, psi, pt[i - 1, 3] + pt[i - 1, 4], l, 0])
        else:                                                   # arc, Track.py:77-117
            l = spec[i, 0]
            r = spec[i, 1]
            direction = 1 if r >= 0 else -1
            if i == 0:
                ang = 0
                cx = 0 + np.abs(r) * np.cos(ang + direction * np.pi / 2)
                cy = 0 + np.abs(r) * np.sin(ang + direction * np.pi / 2)
            else:
                ang = pt[i - 1, 2]
                cx = pt[i - 1, 0] + np.abs(r) * np.cos(ang + direction * np.pi / 2)
                cy = pt[i - 1, 1] + np.abs(r) * np.sin(ang + direction * np.pi / 2)
            spanAng = l / np.abs(r)
            psi = _wrap(ang + spanAng * np.sign(r))
            angleNormal = _wrap((direction * np.pi / 2 + ang))
            angle = -(np.pi - np.abs(angleNormal)) * (_sign(angleNormal))
            x = cx + np.abs(r) * np.cos(angle + direction * spanAng)
            y = cy + np.abs(r) * np.sin(angle + direction * spanAng)
            if i == 0:
                pt[i, :] = np.array([x, y, psi, pt[i, 3], l, 1 / r])
            else:
                pt[i, :] = np.array([x, y, psi, pt[i - 1, 3] + pt[i - 1, 4], l, 1 / r])
    xs, ys = pt[-2, 0], pt[-2, 1]                               # Track.py:121-131
    l = np.sqrt((0 - xs) ** 2 + (0 - ys) ** 2)
    pt[-1, :] = np.array([0, 0, 0, pt[-2, 3] + pt[-2, 4], l, 0])
    return pt, pt[-1, 3] + pt[-1, 4]


def curvature(pt, s):
    """Track.py:292-310 (raises if s lands on no segment, as int(np.where(...)) does)."""
    TrackLength = pt[-1, 3] + pt[-1, 4]
    while s > TrackLength:
        s = s - TrackLength
    index = np.all([[s >= pt[:, 3]], [s < pt[:, 3] + pt[:, 4]]], axis=0)
    hit = np.where(np.squeeze(index))[0]
    if hit.size != 1:           # the reference's int(np.where(...)[0]) raises here too
        raise ValueError("curvature: s=%r is on no track segment" % (s,))
    return pt[int(hit[0]), 5]


def _wrap_angle(a):
    """Track.py:367-375."""
    if a < -np.pi:
        return 2 * np.pi + a
    if a > np.pi:
        return a - 2 * np.pi
    return a


def get_global_position(pt, s, ey):
    """Map.getGlobalPosition, Track.py:135-189: curvilinear (s, ey) -> inertial (X, Y).  Raises where the reference raises
    (s on no segment, e.g. s == k * TrackLength after the wrap loop)."""
    TrackLength = pt[-1, 3] + pt[-1, 4]
    while s > TrackLength:
        s = s - TrackLength
    index = np.all([[s >= pt[:, 3]], [s < pt[:, 3] + pt[:, 4]]], axis=0)
    hit = np.where(np.squeeze(index))[0]
    if hit.size != 1:
        raise ValueError("getGlobalPosition: s=%r is on no track segment" % (s,))
    i = int(hit[0])
    if pt[i, 5] == 0.0:                                  # straight segment: linear interpolation + normal offset (:150-163)
        xf, yf, xs, ys, psi = pt[i, 0], pt[i, 1], pt[i - 1, 0], pt[i - 1, 1], pt[i, 2]
        deltaL = pt[i, 4]; reltaL = s - pt[i, 3]
        x = (1 - reltaL / deltaL) * xs + reltaL / deltaL * xf + ey * np.cos(psi + np.pi / 2)
        y = (1 - reltaL / deltaL) * ys + reltaL / deltaL * yf + ey * np.sin(psi + np.pi / 2)
    else:                                                # arc (:164-187)
        r = 1 / pt[i, 5]; ang = pt[i - 1, 2]
        direction = 1 if r >= 0 else -1
        CenterX = pt[i - 1, 0] + np.abs(r) * np.cos(ang + direction * np.pi / 2)
        CenterY = pt[i - 1, 1] + np.abs(r) * np.sin(ang + direction * np.pi / 2)
        spanAng = (s - pt[i, 3]) / (np.pi * np.abs(r)) * np.pi
        angleNormal = _wrap_angle(direction * np.pi / 2 + ang)
        angle = -(np.pi - np.abs(angleNormal)) * (1 if angleNormal >= 0 else -1)
        x = CenterX + (np.abs(r) - direction * ey) * np.cos(angle + direction * spanAng)
        y = CenterY + (np.abs(r) - direction * ey) * np.sin(angle + direction * spanAng)
    return x, y


# --------------------------------------------------------------------------------------
# LTV model regression  (fnc/controller/PredictiveModel.py)
# --------------------------------------------------------------------------------------
MAXNUMPOINT = 7          # PredictiveModel.py:18
H_BAND = 5               # :19
LAMB = 0.0               # :20
DT = 0.1                 # :21
SCALING = np.array([0.1, 1.0, 1.0, 1.0, 1.0])   # diag of :22-26


def lti_regression(x, u, lamb):
    """Utilities.Regression, fnc/Utilities.py:5-28: ridge least squares x_{k+1} ~ A x_k + B u_k over rows 1..T-2 of one lap.
    Returns A (6,6), B (6,2), Error (2,6) = [max; min] over the rows of (X W - Y)."""
    Y = x[2:x.shape[0], :]
    X = np.hstack((x[1:(x.shape[0] - 1), :], u[1:(x.shape[0] - 1), :]))
    Q = np.linalg.inv(np.dot(X.T, X) + lamb * np.eye(X.shape[1]))
    W = np.dot(Q, np.dot(X.T, Y))
    E = np.dot(X, W) - Y
    return W.T[:, 0:6], W.T[:, 6:8], np.vstack((np.max(E, axis=0), np.min(E, axis=0)))


def model_sorted_insert(xStored, uStored, lapTime, x, u):
    """PredictiveModel.addTrajectory, PredictiveModel.py:35-46 (ascending length, ties append)."""
    if lapTime == [] or x.shape[0] >= lapTime[-1]:
        xStored.append(x); uStored.append(u); lapTime.append(x.shape[0])
    else:
        for i in range(len(xStored)):
            if x.shape[0] < lapTime[i]:
                xStored.insert(i, x); uStored.insert(i, u); lapTime.insert(i, x.shape[0])
                break


def compute_indices(xlap, ulap, xu):
    """PredictiveModel.computeIndices, :180-197.  Rows 0..T-2 of the lap are candidates.

    Distance = sum_j |(D_tj - xu_j) * w_j| accumulated in feature order j = 0..4 (what
    la.norm(diff, 1, axis=1) does for a 5-wide row).  Ties are broken towards the lower row
    index (np.argsort's default introsort is not stable; exact ties do not occur in float data).
    """
    D = np.hstack((xlap[0:-1, 0:3], ulap[0:-1, :]))
    diff = (D - xu[None, :]) * SCALING[None, :]
    norm = np.abs(diff[:, 0])
    for j in range(1, 5):
        norm = norm + np.abs(diff[:, j])
    within = np.where(norm < H_BAND)[0]
    if within.shape[0] >= MAXNUMPOINT:
        index = np.argsort(norm, kind='stable')[0:MAXNUMPOINT]
    else:
        index = within
    K = (1 - (norm[index] / H_BAND) ** 2) * 3 / 4
    return index, K


def regression_and_linearization(xStored, uStored, usedIt, pt, x, u):
    """PredictiveModel.regressionAndLinearization, :48-139.  Returns Ai(6,6), Bi(6,2), Ci(6)."""
    n, d = 6, 2
    Ai = np.zeros((n, n)); Bi = np.zeros((n, d)); Ci = np.zeros(n)
    xu = np.hstack((x[0:3], u[:]))
    idx, Ks = [], []
    for it in usedIt:
        i_, k_ = compute_indices(xStored[it], uStored[it], xu)
        idx.append(i_); Ks.append(k_)

    def q_m(inputFeature):           # compute_Q_M, :141-155
        rows = [np.hstack((xStored[it][np.ix_(idx[c], [0, 1, 2])], uStored[it][np.ix_(idx[c], [inputFeature])]))
                for c, it in enumerate(usedIt)]
        X0 = np.vstack(rows) if rows else np.empty((0, 4))
        Ktot = np.concatenate(Ks) if Ks else np.empty(0)
        M = np.hstack((X0, np.ones((X0.shape[0], 1))))
        Q = np.dot(np.dot(M.T, np.diag(Ktot)), M) + LAMB * np.eye(5)
        return Q, M, Ktot

    def b_vec(yIndex, M, Ktot):      # compute_b, :157-168
        y = np.concatenate([xStored[it][idx[c] + 1, yIndex] for c, it in enumerate(usedIt)])
        return -np.dot(np.dot(M.T, np.diag(Ktot)), y)

    def locreg(Q, b):                # LMPC_LocLinReg, :170-178: cvxopt qp(Q,b) unconstrained  <=>  Q theta = -b
        return np.linalg.solve(Q, -b)

    Q_vx, M_vx, Kt = q_m(1)                                     # :66-70 (input feature a = u[1])
    th = locreg(Q_vx, b_vec(0, M_vx, Kt))
    Ai[0, 0:3] = th[0:3]; Bi[0, 1] = th[3]; Ci[0] = th[4]
    Q_lat, M_lat, Kt = q_m(0)                                   # :74-82 (input feature delta = u[0])
    th = locreg(Q_lat, b_vec(1, M_lat, Kt))
    Ai[1, 0:3] = th[0:3]; Bi[1, 0] = th[3]; Ci[1] = th[4]
    th = locreg(Q_lat, b_vec(2, M_lat, Kt))
    Ai[2, 0:3] = th[0:3]; Bi[2, 0] = th[3]; Ci[2] = th[4]

    vx, vy, wz, epsi, s, ey = x                                 # :86-89
    dt = DT
    cur = curvature(pt, s)                                      # :95-96
    den = 1 - cur * ey
    depsi_vx = -dt * np.cos(epsi) / den * cur                   # :102-107
    depsi_vy = dt * np.sin(epsi) / den * cur
    depsi_wz = dt
    depsi_epsi = 1 - dt * (-vx * np.sin(epsi) - vy * np.cos(epsi)) / den * cur
    depsi_s = 0
    depsi_ey = dt * (vx * np.cos(epsi) - vy * np.sin(epsi)) / (den ** 2) * cur * (-cur)
    Ai[3, :] = [depsi_vx, depsi_vy, depsi_wz, depsi_epsi, depsi_s, depsi_ey]
    Ci[3] = epsi + dt * (wz - (vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey) * cur) - np.dot(Ai[3, :], x)
    ds_vx = dt * (np.cos(epsi) / den)                           # :114-119
    ds_vy = -dt * (np.sin(epsi) / den)
    ds_wz = 0
    ds_epsi = dt * (-vx * np.sin(epsi) - vy * np.cos(epsi)) / den
    ds_s = 1
    ds_ey = -dt * (vx * np.cos(epsi) - vy * np.sin(epsi)) / (den ** 2) * (-cur)
    Ai[4, :] = [ds_vx, ds_vy, ds_wz, ds_epsi, ds_s, ds_ey]
    Ci[4] = s + dt * ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey)) - np.dot(Ai[4, :], x)
    dey_vx = dt * np.sin(epsi)                                  # :127-132
    dey_vy = dt * np.cos(epsi)
    dey_wz = 0
    dey_epsi = dt * (vx * np.cos(epsi) - vy * np.sin(epsi))
    dey_s = 0
    dey_ey = 1
    Ai[5, :] = [dey_vx, dey_vy, dey_wz, dey_epsi, dey_s, dey_ey]
    Ci[5] = ey + dt * (vx * np.sin(epsi) + vy * np.cos(epsi)) - np.dot(Ai[5, :], x)
    return Ai, Bi, Ci


def compute_ltv_dynamics(xStored, uStored, usedIt, pt, xLin, uLin, N):
    """MPC.computeLTVdynamics, PredictiveControllers.py:140-145."""
    A, B, C = [], [], []
    for i in range(N):
        Ai, Bi, Ci = regression_and_linearization(xStored, uStored, usedIt, pt, xLin[i], uLin[i])
        A.append(Ai); B.append(Bi); C.append(Ci)
    return np.array(A), np.array(B), np.array(C)


# --------------------------------------------------------------------------------------
# Safe-set selection  (PredictiveControllers.py:386-416, 478-514)
# --------------------------------------------------------------------------------------

def select_points(SS_it, uSS_it, Qfun_it, it, zt, numPoints, xPred, cur_it, timeStep, N, TrackLength):
    """LMPC.selectPoints, :478-514.  `xPred is None` <=> the reference's `self.xPred == []`."""
    x = SS_it
    diff = x - zt[None, :]
    norm = np.abs(diff[:, 0])
    for j in range(1, 6):                       # la.norm(diff, 1, axis=1): |.| summed in column order
        norm = norm + np.abs(diff[:, j])
    MinNorm = int(np.argmin(norm))
    if MinNorm - numPoints / 2 >= 0:
        index = range(-int(numPoints / 2) + MinNorm, int(numPoints / 2) + MinNorm + 1)
    else:
        index = range(MinNorm, MinNorm + int(numPoints))
    index = list(index)
    SS_Points = x[index, :].T
    SSu_Points = uSS_it[index, :].T
    if xPred is None:
        Sel_Qfun = Qfun_it[index]
    elif np.all((xPred[:, 4] > TrackLength) == False):
        Sel_Qfun = Qfun_it[index]
    elif it < cur_it - 1:
        Sel_Qfun = Qfun_it[index] + Qfun_it[0]
    else:
        predCurrLap = N - sum(xPred[:, 4] > TrackLength)
        Sel_Qfun = Qfun_it[index] + timeStep + predCurrLap
    return SS_Points, SSu_Points, Sel_Qfun


def terminal_components(SS, uSS, Qfun, LapTime, zt, numSS_Points, numSS_it, xPred, cur_it, timeStep, N, TrackLength,
                        sortedLapTime=None):
    """Selection part of LMPC.addTerminalComponents, :395-412 (zt wrap at :392-394 is done by the caller)."""
    if sortedLapTime is None:
        sortedLapTime = np.argsort(np.array(LapTime))
    SSsel = np.empty((6, 0)); Succ = np.empty((6, 0)); SuccU = np.empty((2, 0)); Qsel = np.empty(0)
    for jj in sortedLapTime[0:numSS_it]:
        P, U, Qf = select_points(SS[jj], uSS[jj], Qfun[jj], jj, zt, numSS_Points / numSS_it + 1, xPred, cur_it,
                                 timeStep, N, TrackLength)
        Succ = np.append(Succ, P[:, 1:], axis=1)
        SuccU = np.append(SuccU, U[:, 1:], axis=1)
        SSsel = np.append(SSsel, P[:, 0:-1], axis=1)
        Qsel = np.append(Qsel, Qf[0:-1], axis=0)
    return SSsel, Qsel, Succ, SuccU


def compute_cost(x, TrackLength):
    """LMPC.computeCost, :447-464."""
    T = x.shape[0]
    Cost = 10000 * np.ones(T)
    for i in range(T):
        if i == 0:
            Cost[T - 1 - i] = 0
        elif x[T - 1 - i, 4] < TrackLength:
            Cost[T - 1 - i] = Cost[T - 1 - i + 1] + 1
        else:
            Cost[T - 1 - i] = 0
    return Cost


# --------------------------------------------------------------------------------------
# QP assembly in the reference's OSQP form  (PredictiveControllers.py:166-257, 340-362, 259-273)
# --------------------------------------------------------------------------------------

def block_diag(*mats):
    r = sum(m.shape[0] for m in mats); c = sum(m.shape[1] for m in mats)
    out = np.zeros((r, c)); i = j = 0
    for m in mats:
        out[i:i + m.shape[0], j:j + m.shape[1]] = m; i += m.shape[0]; j += m.shape[1]
    return out


class QPParams:
    """Numeric content of MPCParams (PredictiveControllers.py:24-51) + LMPC ctor args (:293)."""

    def __init__(self, N, Q, R, Qf, dR, Qslack, Fx, bx, Fu, bu, xRef, QterminalSlack=None, numSS_Points=0, numSS_it=0, slacks=True):
        self.n, self.d, self.N = 6, 2, N
        self.slacks = bool(slacks)
        self.Q = np.asarray(Q, float); self.R = np.asarray(R, float); self.Qf = np.asarray(Qf, float)
        self.dR = np.asarray(dR, float).reshape(-1); self.Qslack = np.asarray(Qslack, float).reshape(-1)
        self.Fx = np.asarray(Fx, float); self.bx = np.squeeze(np.asarray(bx, float)).reshape(-1)
        self.Fu = np.asarray(Fu, float); self.bu = np.squeeze(np.asarray(bu, float)).reshape(-1)
        self.xRef = np.asarray(xRef, float).reshape(-1)
        self.QterminalSlack = None if QterminalSlack is None else np.asarray(QterminalSlack, float)
        self.numSS_Points, self.numSS_it = numSS_Points, numSS_it

    @staticmethod
    def lmpc_default(N=12, halfWidth=0.4):
        """initLMPCParams, initControllerParameters.py:28-59."""
        Fx = np.array([[0., 0., 0., 0., 0., 1.], [0., 0., 0., 0., 0., -1.]])
        Fu = np.kron(np.eye(2), np.array([1, -1])).T
        return QPParams(N, np.zeros((6, 6)), np.zeros((2, 2)), np.zeros((6, 6)), 5 * np.array([1.0, 10.0]),
                        np.array([5., 25.]), Fx, [halfWidth, halfWidth], Fu, [0.5, 0.5, 10.0, 10.0], np.zeros(6),
                        500 * np.eye(6), 48, 4)

    @staticmethod
    def mpc_default(N=12, vt=0.8):
        """initMPCParams, initControllerParameters.py:4-26."""
        Fx = np.array([[0., 0., 0., 0., 0., 1.], [0., 0., 0., 0., 0., -1.]])
        Fu = np.kron(np.eye(2), np.array([1, -1])).T
        return QPParams(N, np.diag([1.0, 1.0, 1, 1, 0.0, 100.0]), np.diag([1.0, 10.0]), np.zeros((6, 6)), np.zeros(2),
                        np.array([0., 50.]), Fx, [2., 2.], Fu, [0.5, 0.5, 10.0, 10.0], np.array([vt, 0, 0, 0, 0, 0]))


def build_ineq(p):
    """MPC.buildIneqConstr, :166-198 (both branches of `self.slacks`, :184-198)."""
    N = p.N
    Mat = block_diag(*([p.Fx] * N))
    Fxtot = np.hstack((Mat, np.zeros((Mat.shape[0], p.n))))
    bxtot = np.tile(p.bx, N)
    Futot = block_diag(*([p.Fu] * N))
    butot = np.tile(p.bu, N)
    F_hard = block_diag(Fxtot, Futot)
    if not p.slacks:                                   # :196-198
        return F_hard, np.hstack((bxtot, butot))
    nc_x = p.Fx.shape[0]
    addSlack = np.zeros((F_hard.shape[0], nc_x * N))
    addSlack[0:nc_x * N, 0:nc_x * N] = -np.eye(nc_x * N)
    Positivity = np.hstack((np.zeros((nc_x * N, F_hard.shape[1])), -np.eye(nc_x * N)))
    F = np.vstack((np.hstack((F_hard, addSlack)), Positivity))
    b = np.hstack((bxtot, butot, np.zeros(nc_x * N)))
    return F, b


def build_cost(p, OldInput):
    """MPC.buildCost, :228-257 (both branches of `self.slacks`, :248-254)."""
    N, d = p.N, p.d
    Hx = block_diag(*([p.Q] * N))
    Hu = block_diag(*([p.R + 2 * np.diag(p.dR)] * N))
    for i in range(d):
        Hu[i - d, i - d] = Hu[i - d, i - d] - p.dR[i]
    OffDiaf = -np.tile(p.dR, N - 1)
    np.fill_diagonal(Hu[d:], OffDiaf)
    np.fill_diagonal(Hu[:, d:], OffDiaf)
    q = -2 * np.dot(np.append(np.tile(p.xRef, N + 1), np.zeros(p.R.shape[0] * N)), block_diag(Hx, p.Qf, Hu))
    q[p.n * (N + 1):p.n * (N + 1) + d] = -2 * np.dot(np.reshape(OldInput, (-1,))[0:d], np.diag(p.dR))
    if not p.slacks:                                   # :252-254
        return 2 * block_diag(Hx, p.Qf, Hu), q
    nc_x = p.Fx.shape[0]
    H = block_diag(Hx, p.Qf, Hu, p.Qslack[0] * np.eye(nc_x * N))
    q = np.append(q, p.Qslack[1] * np.ones(nc_x * N))
    return 2 * H, q


def build_eq(p, A, B, C):
    """MPC.buildEqConstr, :200-226.  A,B,C: per-stage lists (LTV) or single matrices (LTI, C=None)."""
    N, n, d = p.N, p.n, p.d
    Gx = np.eye(n * (N + 1)); Gu = np.zeros((n * (N + 1), d * N))
    E = np.zeros((n * (N + 1), n)); E[np.arange(n)] = np.eye(n)
    L = np.zeros(n * (N + 1))
    ltv = np.ndim(A) == 3
    for i in range(N):
        Gx[(n + i * n):(n + i * n + n), (i * n):(i * n + n)] = -(A[i] if ltv else A)
        Gu[(n + i * n):(n + i * n + n), (i * d):(i * d + d)] = -(B[i] if ltv else B)
        if ltv:
            L[(n + i * n):(n + i * n + n)] = C[i]
    G = np.hstack((Gx, Gu, np.zeros((Gx.shape[0], p.Fx.shape[0] * N)))) if p.slacks else np.hstack((Gx, Gu))     # :218-221
    return G, E, L


def assemble_lmpc_qp(p, A, B, C, x0, OldInput, SSsel, Qsel):
    """LMPC: buildCost + buildEqConstr + addSafeSetIneqConstr (:340-343) + addSafeSetEqConstr (:345-357)
    + addSafeSetCost (:359-362), then the OSQP stacking of osqp_solve_qp (:269-273).
    Returns dense (P, q, Aosqp, l, u)."""
    n = p.n
    F, b = build_ineq(p)
    H, q = build_cost(p, OldInput)
    G, E, L = build_eq(p, A, B, C)
    S = SSsel.shape[1]
    F_FTOCP = block_diag(F, np.hstack((-np.eye(S), np.zeros((S, n)))))
    b_FTOCP = np.append(b, np.zeros(S))
    xTermCons = np.zeros((n, G.shape[1])); xTermCons[:, p.N * n:(p.N + 1) * n] = np.eye(n)
    G_x_u_slack = np.vstack((G, xTermCons))
    G_lam = np.vstack((np.zeros((G.shape[0], S + n)), np.hstack((-SSsel, np.eye(n)))))
    G_one = np.append(np.append(np.zeros(G.shape[1]), np.ones(S)), np.zeros(n))
    G_FTOCP = np.vstack((np.hstack((G_x_u_slack, G_lam)), G_one))
    E_FTOCP = np.vstack((E, np.zeros((n + 1, n))))
    L_FTOCP = np.append(np.append(L, np.zeros(n)), 1)
    H_FTOCP = block_diag(H, np.zeros((S, S)), 2 * p.QterminalSlack)
    q_FTOCP = np.append(np.append(q, Qsel), np.zeros(n))
    beq = np.add(np.dot(E_FTOCP, x0), L_FTOCP)
    Aosqp = np.vstack((F_FTOCP, G_FTOCP))
    l = np.hstack((-np.inf * np.ones(len(b_FTOCP)), beq))
    u = np.hstack((b_FTOCP, beq))
    return H_FTOCP, q_FTOCP, Aosqp, l, u


def assemble_mpc_qp(p, A, B, C, x0, OldInput):
    """MPC (no terminal set): addTerminalComponents is the identity copy (:147-155)."""
    F, b = build_ineq(p)
    H, q = build_cost(p, OldInput)
    G, E, L = build_eq(p, A, B, C)
    beq = np.add(np.dot(E, x0), L)
    return H, q, np.vstack((F, G)), np.hstack((-np.inf * np.ones(len(b)), beq)), np.hstack((b, beq))


# --------------------------------------------------------------------------------------
# OSQP restatement binding
# --------------------------------------------------------------------------------------

class _Settings(ctypes.Structure):
    _fields_ = [(k, ctypes.c_double) for k in ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf",
                                               "eps_dual_inf", "delta", "adaptive_rho_tolerance")] + \
               [(k, ctypes.c_int) for k in ("max_iter", "check_termination", "scaling", "adaptive_rho",
                                            "adaptive_rho_interval", "polish", "polish_refine_iter")]


class _Info(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("iter", "status", "status_polish", "rho_updates")] + \
               [(k, ctypes.c_double) for k in ("obj_val", "pri_res", "dua_res", "rho_estimate")]


_LIB = None


def build_lib(force=False):
    """gcc-compile oracle/osqp_restated.c -> oracle/libosqp_restated.so."""
    so = os.path.join(_HERE, "libosqp_restated.so")
    src = os.path.join(_HERE, "osqp_restated.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build_lib())
        _LIB.oq_solve.restype = ctypes.c_int
        _LIB.oq_solve_batch.restype = ctypes.c_int
    return _LIB


def min_degree_order(pattern):
    """Greedy minimum-degree ordering of a symmetric sparsity pattern (list of neighbour sets).
    Stands in for the AMD ordering OSQP's QDLDL backend uses; any fill-reducing order gives the
    same solution up to rounding."""
    nn = len(pattern)
    adj = [set(s) - {i} for i, s in enumerate(pattern)]
    alive = [True] * nn
    order = []
    import heapq
    heap = [(len(adj[i]), i) for i in range(nn)]
    heapq.heapify(heap)
    while heap:
        dg, v = heapq.heappop(heap)
        if not alive[v] or dg != len(adj[v]):
            continue
        alive[v] = False; order.append(v)
        nb = list(adj[v])
        for a in nb:
            adj[a].discard(v)
        for ia, a in enumerate(nb):
            for b_ in nb[ia + 1:]:
                if b_ not in adj[a]:
                    adj[a].add(b_); adj[b_].add(a)
        for a in nb:
            heapq.heappush(heap, (len(adj[a]), a))
    return np.array(order, dtype=np.int32)


_PERM_CACHE = {}


def kkt_perm(Pc, Ac):
    """Fill-reducing permutation of [[P,A'],[A,*]] for scipy CSC P (n x n) and A (m x n); cached by pattern."""
    n, m = Pc.shape[0], Ac.shape[0]
    key = (n, m, Pc.indptr.tobytes(), Pc.indices.tobytes(), Ac.indptr.tobytes(), Ac.indices.tobytes())
    if key in _PERM_CACHE:
        return _PERM_CACHE[key]
    pat = [set() for _ in range(n + m)]
    for j in range(n):
        for i in Pc.indices[Pc.indptr[j]:Pc.indptr[j + 1]]:
            pat[i].add(j); pat[j].add(i)
        for i in Ac.indices[Ac.indptr[j]:Ac.indptr[j + 1]]:
            pat[n + i].add(j); pat[j].add(n + i)
    perm = min_degree_order(pat)
    _PERM_CACHE[key] = perm
    return perm


class OSQPResult:
    pass


def osqp_solve(P, q, A, l, u, polish=True, **kw):
    """Solve min 1/2 x'Px + q'x s.t. l <= Ax <= u with the restated OSQP (defaults = OSQP defaults).
    P, A: dense arrays or scipy sparse.  Returns OSQPResult(x, y, z, status, iter, ...)."""
    from scipy import sparse
    Pc = sparse.csc_matrix(P); Pc.sort_indices()
    Ac = sparse.csc_matrix(A); Ac.sort_indices()
    n, m = Pc.shape[0], Ac.shape[0]
    st = _Settings(); lib = _lib(); lib.oq_default_settings(ctypes.byref(st))
    st.polish = 1 if polish else 0
    for k, v in kw.items():
        setattr(st, k, v)
    perm = kkt_perm(Pc, Ac)
    q = np.ascontiguousarray(q, float); l = np.ascontiguousarray(l, float); u = np.ascontiguousarray(u, float)
    x = np.zeros(n); y = np.zeros(m); z = np.zeros(m); info = _Info()
    ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(ctypes.c_void_p)
    dp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    Ppi, Pii, Pxx = np.ascontiguousarray(Pc.indptr, np.int32), np.ascontiguousarray(Pc.indices, np.int32), np.ascontiguousarray(Pc.data, float)
    Api, Aii, Axx = np.ascontiguousarray(Ac.indptr, np.int32), np.ascontiguousarray(Ac.indices, np.int32), np.ascontiguousarray(Ac.data, float)
    permc = np.ascontiguousarray(perm, np.int32)
    status = lib.oq_solve(ctypes.c_int(n), ctypes.c_int(m), ip(Ppi), ip(Pii), dp(Pxx), dp(q), ip(Api), ip(Aii), dp(Axx),
                          dp(l), dp(u), ip(permc), ctypes.byref(st), dp(x), dp(y), dp(z), ctypes.byref(info))
    r = OSQPResult()
    r.interior = False
    r.x, r.y, r.z, r.status, r.iter = x, y, z, status, info.iter
    r.status_polish, r.obj_val, r.pri_res, r.dua_res, r.rho_updates = info.status_polish, info.obj_val, info.pri_res, info.dua_res, info.rho_updates
    return r


def kkt_certificate(P, q, A, l, u, x, y):
    """Solver-independent optimality certificate.  Returns dict of residuals:
    stationarity |Px+q+A'y|_inf, primal violation, dual-sign violation and complementarity."""
    P = np.asarray(P.todense()) if hasattr(P, "todense") else np.asarray(P)
    A = np.asarray(A.todense()) if hasattr(A, "todense") else np.asarray(A)
    Ax = A @ x
    stat = np.max(np.abs(P @ x + q + A.T @ y))
    prim = max(np.max(np.maximum(l - Ax, 0)), np.max(np.maximum(Ax - u, 0)))
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    fin_u, fin_l = np.isfinite(u), np.isfinite(l)
    comp = 0.0
    if fin_u.any():
        comp = max(comp, np.max(np.abs(yp[fin_u] * (u[fin_u] - Ax[fin_u]))))
    if fin_l.any():
        comp = max(comp, np.max(np.abs(ym[fin_l] * (Ax[fin_l] - l[fin_l]))))
    dsign = 0.0
    if (~fin_u).any():
        dsign = max(dsign, np.max(yp[~fin_u], initial=0.0))
    if (~fin_l).any():
        dsign = max(dsign, np.max(-ym[~fin_l], initial=0.0))
    return dict(stationarity=stat, primal=prim, complementarity=comp, dual_sign=dsign)


# --------------------------------------------------------------------------------------
# Controller state machines (restated MPC / LMPC, PredictiveControllers.py:56-137, 286-514)
# --------------------------------------------------------------------------------------

class OracleModel:
    """PredictiveModel.py:11-46 data side."""

    def __init__(self, pt, trToUse):
        self.pt = pt; self.TrackLength = pt[-1, 3] + pt[-1, 4]
        self.xStored, self.uStored, self.lapTime = [], [], []
        self.usedIt = list(range(trToUse))

    def addTrajectory(self, x, u):
        model_sorted_insert(self.xStored, self.uStored, self.lapTime, x, u)


class OracleLMPC:
    """LMPC(MPC) with timeVarying=True, PredictiveControllers.py:286-514 + MPC.solve :110-137."""

    def __init__(self, params, model, solver_kw=None, exact=False):
        self.exact = exact
        self.p = params; self.N = params.N; self.model = model
        self.numSS_Points, self.numSS_it = params.numSS_Points, params.numSS_it
        self.OldInput = np.zeros(2)
        self.xPred = None
        self.LapTime, self.SS, self.uSS, self.Qfun = [], [], [], []
        self.zt = np.array([0.0, 0.0, 0.0, 0.0, 10.0, 0.0])
        self.it = 0; self.timeStep = 0
        self.solver_kw = solver_kw or {}
        # MPC.__init__ :88-91
        self.xLin = model.xStored[-1][0:self.N + 1, :]
        self.uLin = model.uStored[-1][0:self.N, :]

    def addTrajectory(self, x, u):              # :418-445
        self.LapTime.append(x.shape[0]); self.SS.append(x); self.uSS.append(u)
        self.Qfun.append(compute_cost(x, self.model.TrackLength))
        if self.it == 0:
            self.xLin = self.SS[self.it][1:self.N + 2, :]
            self.uLin = self.uSS[self.it][1:self.N + 1, :]
        self.it += 1; self.timeStep = 0

    def addPoint(self, x, u):                   # :466-474
        TL = self.model.TrackLength
        self.SS[self.it - 1] = np.append(self.SS[self.it - 1], np.array([x + np.array([0, 0, 0, 0, TL, 0])]), axis=0)
        self.uSS[self.it - 1] = np.append(self.uSS[self.it - 1], np.array([u]), axis=0)
        self.Qfun[self.it - 1] = np.append(self.Qfun[self.it - 1], self.Qfun[self.it - 1][-1] - 1)

    def solve(self, x0):                        # MPC.solve :110-137
        m, p, N = self.model, self.p, self.N
        TL = m.TrackLength
        self.A, self.B, self.C = compute_ltv_dynamics(m.xStored, m.uStored, m.usedIt, m.pt, self.xLin, self.uLin, N)
        if self.zt[4] - x0[4] > TL / 2:         # addTerminalComponents :392-394
            self.zt[4] = np.max([self.zt[4] - TL, 0])
            # (sic) reference quirk E-2: row 4, LAST COLUMN (ey) -- and IN PLACE: on the first solve xLin is
            # still a view of the stored lap (:432), so the stored lap itself is modified, exactly as upstream.
            self.xLin[4, -1] = self.xLin[4, -1] - TL
        SSsel, Qsel, Succ, SuccU = terminal_components(self.SS, self.uSS, self.Qfun, self.LapTime, self.zt,
                                                       self.numSS_Points, self.numSS_it, self.xPred, self.it,
                                                       self.timeStep, N, TL)
        self.SS_PointSelectedTot, self.Qfun_SelectedTot = SSsel, Qsel
        P, q, A, l, u = assemble_lmpc_qp(p, self.A, self.B, self.C, x0, self.OldInput, SSsel, Qsel)
        self.qp = (P, q, A, l, u)
        if self.exact:
            res, self.cert = osqp_solve_exact(P, q, A, l, u)
        else:
            res = osqp_solve(P, q, A, l, u, polish=True, **self.solver_kw)
        self.res = res; self.feasible = 1 if res.status == 1 else 0
        sol = res.x
        n, d = 6, 2                              # unpackSolution :364-379
        self.xPred = sol[0:n * (N + 1)].reshape(N + 1, n)
        self.uPred = sol[n * (N + 1):n * (N + 1) + d * N].reshape(N, d)
        i0 = n * (N + 1) + d * N; i1 = i0 + 2 * N; i2 = i1 + SSsel.shape[1]
        self.slack, self.lambd, self.slackTerminal = sol[i0:i1], sol[i1:i2], sol[i2:]
        self.zt = np.dot(Succ, self.lambd)       # feasibleStateInput :382-384
        self.zt_u = np.dot(SuccU, self.lambd)
        self.xLin = np.vstack((self.xPred[1:, :], self.zt))      # :131-137
        self.uLin = np.vstack((self.uPred[1:, :], self.zt_u))
        self.OldInput = self.uPred[0, :]
        self.timeStep += 1


# --------------------------------------------------------------------------------------
# Plant + PID (caller side of the path; used only to synthesise laps)
# fnc/simulator/SysModel.py:56-147, fnc/Utilities.py:42-67
# --------------------------------------------------------------------------------------

def dyn_model(pt, x, x_glob, u, rng_randn):
    """Simulator.dynModel, SysModel.py:56-147.  rng_randn() supplies the N(0,1) draws."""
    m = 1.98; lf = 0.125; lr = 0.125; Iz = 0.024
    Df = 0.8 * m * 9.81 / 2.0; Cf = 1.25; Bf = 1.0
    Dr = 0.8 * m * 9.81 / 2.0; Cr = 1.25; Br = 1.0
    deltaT = 0.001; dt = 0.1
    x_next = np.zeros(6); cur_x_next = np.zeros(6)
    delta, a = u[0], u[1]
    psi, X, Y = x_glob[3], x_glob[4], x_glob[5]
    vx, vy, wz, epsi, s, ey = x
    i = 0
    while (i + 1) * deltaT <= dt:
        alpha_f = delta - np.arctan2(vy + lf * wz, vx)
        alpha_r = - np.arctan2(vy - lf * wz, vx)
        Fyf = Df * np.sin(Cf * np.arctan(Bf * alpha_f))
        Fyr = Dr * np.sin(Cr * np.arctan(Br * alpha_r))
        x_next[0] = vx + deltaT * (a - 1 / m * Fyf * np.sin(delta) + wz * vy)
        x_next[1] = vy + deltaT * (1 / m * (Fyf * np.cos(delta) + Fyr) - wz * vx)
        x_next[2] = wz + deltaT * (1 / Iz * (lf * Fyf * np.cos(delta) - lr * Fyr))
        x_next[3] = psi + deltaT * (wz)
        x_next[4] = X + deltaT * ((vx * np.cos(psi) - vy * np.sin(psi)))
        x_next[5] = Y + deltaT * (vx * np.sin(psi) + vy * np.cos(psi))
        cur = curvature(pt, s)
        cur_x_next[0] = x_next[0]; cur_x_next[1] = x_next[1]; cur_x_next[2] = x_next[2]
        cur_x_next[3] = epsi + deltaT * (wz - (vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey) * cur)
        cur_x_next[4] = s + deltaT * ((vx * np.cos(epsi) - vy * np.sin(epsi)) / (1 - cur * ey))
        cur_x_next[5] = ey + deltaT * (vx * np.sin(epsi) + vy * np.cos(epsi))
        psi, X, Y = x_next[3], x_next[4], x_next[5]
        vx, vy, wz, epsi, s, ey = cur_x_next
        i += 1
    noise_vx = np.max([-0.05, np.min([rng_randn() * 0.01, 0.05])])
    noise_vy = np.max([-0.05, np.min([rng_randn() * 0.01, 0.05])])
    noise_wz = np.max([-0.05, np.min([rng_randn() * 0.005, 0.05])])
    cur_x_next[0] += 0.01 * noise_vx; cur_x_next[1] += 0.01 * noise_vy; cur_x_next[2] += 0.01 * noise_wz
    return cur_x_next.copy(), x_next.copy()


def pid_lap(pt, vt, seed, x0=None, multiLap=True, maxSimTime=100):
    """PID seed lap: Simulator.sim (SysModel.py:22-54) driven by PID.solve (Utilities.py:60-67),
    global NumPy RNG seeded with `seed` (draw order identical to the reference)."""
    rs = np.random.RandomState(seed)
    TL = pt[-1, 3] + pt[-1, 4]
    x0 = np.array([0.5, 0, 0, 0, 0, 0]) if x0 is None else x0
    x_cl, x_glob, u_cl = [x0], [x0], []
    i = 0; flagExt = False
    while i < int(maxSimTime / 0.1) and not flagExt:
        xx = x_cl[-1]
        u = np.zeros(2)
        u[0] = - 0.6 * xx[5] - 0.9 * xx[3] + np.max([-0.9, np.min([rs.randn() * 0.25, 0.9])])
        u[1] = 1.5 * (vt - xx[0]) + np.max([-0.2, np.min([rs.randn() * 0.10, 0.2])])
        u_cl.append(u)
        xt, xg = dyn_model(pt, x_cl[-1], x_glob[-1], u, rs.randn)
        x_cl.append(xt); x_glob.append(xg)
        if (not multiLap) and x_cl[-1][4] > TL:
            flagExt = True
        i += 1
    x_cl.pop(); x_glob.pop()
    return np.array(x_cl), np.array(u_cl), np.array(x_glob)


def _active_set_finish(P, q, E, b, G, h, s, lam, scale, rounds=12, delta=1e-9, refine=8):
    """See dense_ipm_solve.  Returns (z, y, lam) of the accepted active set or None."""
    n, me, mi = P.shape[0], E.shape[0], G.shape[0]
    act = lam > s
    for _ in range(rounds):
        Ga, ha = G[act], h[act]; ma = Ga.shape[0]
        K = np.zeros((n + me + ma, n + me + ma))
        K[:n, :n] = P; K[:n, n:n + me] = E.T; K[n:n + me, :n] = E; K[:n, n + me:] = Ga.T; K[n + me:, :n] = Ga
        Kr = K.copy()
        Kr[np.arange(n), np.arange(n)] += delta
        Kr[np.arange(n, n + me + ma), np.arange(n, n + me + ma)] -= delta
        rhs = np.concatenate([-q, b, ha])
        try:
            Ki = np.linalg.inv(Kr)
        except np.linalg.LinAlgError:
            return None
        sol = Ki @ rhs
        for _r in range(refine):
            sol = sol + Ki @ (rhs - K @ sol)
        z, y = sol[:n], sol[n:n + me]
        lam2 = np.zeros(mi); lam2[act] = sol[n + me:]
        bad_p = (~act) & (G @ z - h > 1e-11)
        bad_d = act & (lam2 < -1e-11 * scale)
        if not bad_p.any() and not bad_d.any():
            lam2 = np.maximum(lam2, 0.0)
            if np.abs(K @ sol - rhs).max() > 1e-9 * scale:       # (the refinement did not converge: a singular active set with an inconsistent right-hand side)
                return None
            return z, y, lam2
        act = (act & ~bad_d) | bad_p
    return None


def dense_ipm_solve(P, q, A, l, u, tol=1e-13, max_iter=80, polish=True):
    """Second, independent way to the QP's optimum (test infrastructure; round 5): a textbook dense primal-dual interior-point iteration (Mehrotra) on the
    EXPLICIT reference-form matrices P, q, A, l, u (PredictiveControllers.py:259-283) -- dense LU of the reduced KKT matrix, no block structure, no
    Riccati recursion, nothing shared with the HIP kernels or with tests/ipm_model.py.  Used by osqp_solve_exact where the restated ADMM does not reach the
    certificate (near-degenerate QPs on which ADMM converges sublinearly: a million iterations leave residuals of 1e-3).  Whatever this returns is only
    ever ACCEPTED through kkt_certificate.  Rows with l == u are equalities, rows with a finite bound on one or both sides inequalities.
    tol bounds the complementarity gap as well as the residuals: on a QP without strict complementarity the distance of an interior iterate to the optimum
    goes like the square root of the gap (measured at N = 40: gap 7e-10 -> 1.3e-5 in x, gap 7e-12 -> 1e-7), so the default is far below the 1e-8 the
    certificate of an active-set (polished) point is asked for."""
    P = np.asarray(P.todense()) if hasattr(P, "todense") else np.asarray(P, float)
    A = np.asarray(A.todense()) if hasattr(A, "todense") else np.asarray(A, float)
    q = np.asarray(q, float); l = np.asarray(l, float); u = np.asarray(u, float)
    n = P.shape[0]
    eq = np.isfinite(l) & np.isfinite(u) & (l == u)
    up = np.isfinite(u) & ~eq; lo = np.isfinite(l) & ~eq
    E, b = A[eq], u[eq]
    G = np.vstack([A[up], -A[lo]]); h = np.concatenate([u[up], -l[lo]])
    mi, me = G.shape[0], E.shape[0]
    z = np.zeros(n); y = np.zeros(me)
    s = np.maximum(h - G @ z, 1.0); lam = np.ones(mi)
    scale = max(1.0, np.abs(q).max())
    lam *= scale
    delta = 1e-10

    def solve(D, rd, rp, re, rc):
        # (P + G' D G) dz + E' dy = -rd - G' (D rp - rc / s);  E dz = -re          with D = lam / s
        H = P + (G.T * D) @ G + delta * np.eye(n)
        K = np.block([[H, E.T], [E, -delta * np.eye(me)]])
        rhs = np.concatenate([-rd - G.T @ (D * rp - rc / s), -re])
        sol = np.linalg.solve(K, rhs)
        sol += np.linalg.solve(K, rhs - K @ sol)
        dz, dy = sol[:n], sol[n:]
        ds = -rp - G @ dz
        dlam = -(rc + lam * ds) / s
        return dz, dy, ds, dlam
    for it in range(max_iter):
        rd = P @ z + q + G.T @ lam + E.T @ y
        rp = G @ z + s - h
        re = E @ z - b
        mu = s @ lam / max(mi, 1)
        if max(np.abs(rd).max() / scale, np.abs(rp).max(), np.abs(re).max(), mu / scale) < tol:
            break
        D = lam / s
        dz, dy, ds, dlam = solve(D, rd, rp, re, s * lam)
        def step(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-v[neg] / dv[neg]))) if neg.any() else 1.0
        a_aff = min(step(s, ds), step(lam, dlam))
        mu_aff = (s + a_aff * ds) @ (lam + a_aff * dlam) / max(mi, 1)
        sig = (mu_aff / mu) ** 3 if mu > 0 else 0.0
        dz, dy, ds, dlam = solve(D, rd, rp, re, s * lam + ds * dlam - sig * mu)
        a = 0.99 * min(step(s, ds), step(lam, dlam)); a = min(a, 1.0)
        z = z + a * dz; y = y + a * dy; s = s + a * ds; lam = lam + a * dlam
    interior = True
    if polish:
        # (round 6) active-set finish, the idea of the reference's own `polish=True` (PredictiveControllers.py:275): an interior iterate that meets the residual
        # tolerances can still sit 1e-6..1e-5 from the optimum of a FLAT or degenerate QP (the closed-loop probes of round 6 found this function 7e-6 off where the HIP
        # kernels agreed with a 1e-15 solve to 1e-8).  With the active set A = {rows with lam > s} the optimum solves the equality-constrained system
        #     [P E' G_A'; E 0 0; G_A 0 0] (z, y, lam_A) = (-q, b, h_A)
        # exactly (regularised LU + iterative refinement against the unregularised matrix); the point is accepted only if its signs hold -- inactive rows feasible,
        # active multipliers non-negative -- else rows change sides (primal-dual active-set steps) and the system is solved again.  An accepted point lies ON its active
        # set: its KKT residuals are rounding (1e-12), and so is its distance to the optimum along flat directions.
        pol = _active_set_finish(P, q, E, b, G, h, s, lam, scale)
        if pol is not None:
            z, y, lam = pol; interior = False
    r = OSQPResult()
    yy = np.zeros(A.shape[0]); yy[eq] = y
    nu_ = int(up.sum())
    yy[up] += lam[:nu_]; yy[lo] -= lam[nu_:]
    r.x, r.y, r.z, r.status, r.iter = z, yy, A @ z, 1, it
    r.interior = interior
    r.status_polish = 0; r.obj_val = float(0.5 * z @ P @ z + q @ z); r.pri_res = 0.0; r.dua_res = 0.0; r.rho_updates = 0
    return r


def osqp_solve_exact(P, q, A, l, u, want=1e-9):
    """The QP's optimum to certificate level `want` (max KKT residual), using the restated OSQP at
    increasing accuracy; polish results are only accepted through the solver-independent certificate
    (OSQP's own acceptance rule can accept a polish that drops an equality row whose multiplier is
    exactly 0).  Round 5: where the first two ADMM settings have not certified (sublinear convergence on near-degenerate QPs), a dense interior-point
    iteration on the same explicit matrices (dense_ipm_solve) is tried before the million-iteration ADMM runs.  Returns (result, certificate_max)."""
    best = None
    for kw in (dict(polish=True, eps_abs=1e-6, eps_rel=1e-6),
               dict(polish=True, eps_abs=1e-9, eps_rel=1e-9, max_iter=200000),
               dict(dense_ipm=True),
               dict(polish=False, eps_abs=1e-11, eps_rel=1e-11, max_iter=1000000),
               dict(polish=True, eps_abs=1e-11, eps_rel=1e-11, max_iter=1000000)):
        kw = dict(kw)
        if kw.pop("dense_ipm", False):
            r = dense_ipm_solve(P, q, A, l, u)
        else:
            pol = kw.pop("polish")
            r = osqp_solve(P, q, A, l, u, polish=pol, eps_prim_inf=0.0, eps_dual_inf=0.0, **kw)
        c = max(kkt_certificate(P, q, A, l, u, r.x, r.y).values())
        if best is None or c < best[1]:
            best = (r, c)
        # (an interior point is taken only with a complementarity residual far below `want`, see dense_ipm_solve; polished points sit ON their active set)
        if c <= (min(want, 1e-10) if getattr(r, "interior", False) else want):
            break
    return best
