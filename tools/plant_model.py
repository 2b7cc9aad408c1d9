"""Developer tool (CPU): NumPy model of the plant kernel's range-specialised arithmetic (lmpc_kernels.hip.h: plant_atan2 / plant_atan / plant_sin1 /
plant_sincos, same constants, same evaluation order up to FMA contraction) against the oracle's plant (oracle/lmpc_oracle.py: dyn_model = SysModel.py:56-147),
before any GPU time is spent: quadrant logic, Cody-Waite reduction, quotient correction.      python tools/plant_model.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ATAN = [1.00000000000000000e+00, -3.33333333333333148e-01, 1.99999999999972672e-01, -1.42857142855245423e-01, 1.11111111040167687e-01,
        -9.09090892666328670e-02, 7.69230512832331237e-02, -6.66663809625125253e-02, 5.88211633270929110e-02, -5.26165758832842292e-02, 4.75445443774553250e-02,
        -4.31833735800102661e-02, 3.90564747161603193e-02, -3.45674894142850089e-02, 2.91380929673892217e-02, -2.25878550465188170e-02, 1.54855204521000701e-02,
        -9.01332229968379930e-03, 4.26429640711083104e-03, -1.55763149860080764e-03, 4.09014166942688539e-04, -6.83513646937304115e-05, 5.44016462407589206e-06]
SIN1 = [1.00000000000000000e+00, -1.66666666666666657e-01, 8.33333333333335924e-03, -1.98412698412862263e-04, 2.75573192281394626e-06,
        -2.50521085826905639e-08, 1.60589365566945973e-10, -7.62490938337113395e-13, 1.09951716654846206e-15, 4.72028309461793050e-16]
SK = [-1.66666666666666657e-01, 8.33333333333338699e-03, -1.98412698413160743e-04, 2.75573192401844066e-06, -2.50521105474221276e-08, 1.60589767854145033e-10, -7.60496180966857912e-13]
CK = [4.16666666666671293e-02, -1.38888888890215394e-03, 2.48015874329884951e-05, -2.75573799134182448e-07, 2.08910323522207627e-09, -1.31296120098958296e-11, 8.03487213580045595e-13]


def estrin(c, w):
    v = [np.float64(x) + 0 * w for x in c]; pw = w
    while len(v) > 1:
        n = len(v); v = [(v[2 * i + 1] * pw + v[2 * i]) if 2 * i + 1 < n else v[2 * i] for i in range((n + 1) // 2)]; pw = pw * pw
    return v[0]


def patan(z):
    return np.where(np.abs(z) <= 1.0, z * estrin(ATAN, z * z), np.arctan(z))


def patan2(y, x):
    r = 1 / x; z = y * r; z = (y - z * x) * r + z
    return np.where((x > 0) & (np.abs(y) <= x), z * estrin(ATAN, z * z), np.arctan2(y, x))


def psin1(x):
    return np.where(np.abs(x) <= 1.0, x * estrin(SIN1, x * x), np.sin(x))


def psincos(x):
    k = np.rint(x * 0.63661977236758134)
    LD = np.longdouble                                                   # (the kernel's two FMAs: product exact, one rounding -- emulated in extended precision)
    r = (LD(1) * x - LD(1) * k * LD(1.5707963267948966)).astype(np.float64); r = (LD(1) * r - LD(1) * k * LD(6.123233995736766e-17)).astype(np.float64)
    w = r * r; sr = r * w * estrin(SK, w) + r; cr = w * w * estrin(CK, w) + (1 - 0.5 * w)
    q = k.astype(np.int64) & 3
    s0 = np.where(q & 1, cr, sr); c0 = np.where(q & 1, sr, cr)
    return np.where(q & 2, -s0, s0), np.where((q + 1) & 2, -c0, c0)


def plant(pt, TL, x, xg, u):
    def curv(s):
        out = np.zeros_like(s)
        for j in range(len(s)):
            ss = s[j]
            while ss > TL:
                ss -= TL
            for i in range(pt.shape[0]):
                if pt[i, 3] <= ss < pt[i, 3] + pt[i, 4]:
                    out[j] = pt[i, 5]; break
        return out
    m = 1.98; lf = 0.125; lr = 0.125; Iz = 0.024; Df = 0.8 * m * 9.81 / 2; Cf = 1.25; dT = 0.001
    vx, vy, wz, epsi, s, ey = [x[:, i].copy() for i in range(6)]; psi, X, Y = xg[:, 3].copy(), xg[:, 4].copy(), xg[:, 5].copy()
    delta, a = u[:, 0], u[:, 1]; sd, cd = np.sin(delta), np.cos(delta)
    for _ in range(100):
        af = delta - patan2(vy + lf * wz, vx); ar = -patan2(vy - lf * wz, vx)
        Fyf = Df * psin1(Cf * patan(af)); Fyr = Df * psin1(Cf * patan(ar))
        sp, cp = psincos(psi); se, ce = psincos(epsi)
        nvx = vx + dT * (a - 1 / m * Fyf * sd + wz * vy); nvy = vy + dT * (1 / m * (Fyf * cd + Fyr) - wz * vx); nwz = wz + dT * (1 / Iz * (lf * Fyf * cd - lr * Fyr))
        npsi = psi + dT * wz; nX = X + dT * (vx * cp - vy * sp); nY = Y + dT * (vx * sp + vy * cp)
        cur = curv(s)
        nepsi = epsi + dT * (wz - (vx * ce - vy * se) / (1 - cur * ey) * cur); ns = s + dT * ((vx * ce - vy * se) / (1 - cur * ey)); ney = ey + dT * (vx * se + vy * ce)
        vx, vy, wz, epsi, s, ey, psi, X, Y = nvx, nvy, nwz, nepsi, ns, ney, npsi, nX, nY
    return np.stack([vx, vy, wz, epsi, s, ey], 1), np.stack([vx, vy, wz, psi, X, Y], 1)


if __name__ == "__main__":
    from oracle import lmpc_oracle as orc
    from tests import common
    g = common.load_lmpc_golden()
    pt = np.array(g["track"]); TL = float(g["trackLength"])
    rng = np.random.default_rng(0)
    B = 96
    rows = rng.integers(0, 999, B)
    x = g["xPID"][rows] + rng.normal(size=(B, 6)) * np.array([.3, .1, .5, .05, 0, .05]); x[:, 0] = np.abs(x[:, 0]) + 0.3
    xg = np.array(g["xPID_glob"])[rows].copy(); xg[:, 3] += rng.uniform(-30, 30, B)
    u = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-1, 1, B)], 1)
    x[:8, 0] = 0.05; x[:8, 1] = rng.uniform(-0.3, 0.3, 8)                       # slow, sliding cars: |y / vx| > 1 -> the general routines
    xn, xgn = plant(pt, TL, x, xg, u)
    worst = 0.0
    for b in range(B):
        xr, xgr = orc.dyn_model(pt, x[b], xg[b], u[b], lambda: 0.0)
        worst = max(worst, np.abs(xr - xn[b]).max(), np.abs(xgr - xgn[b]).max())
    print("worst |model of the plant kernel - oracle plant| over %d states (100 sub-steps each): %.2e" % (B, worst))
    xs = rng.uniform(-200, 200, 200000); sn, cs = psincos(xs)
    print("sin / cos by Cody-Waite + 7-term kernels on |x| < 200: max abs err %.2e / %.2e" % (np.abs(sn - np.sin(xs)).max(), np.abs(cs - np.cos(xs)).max()))
