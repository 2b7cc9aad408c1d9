"""Constructed lap stores for the K1 regression kernel's integer prefilter (test infrastructure).

K1 (lmpc_regress_kernel) ranks the rows of a lap on a 16-bit fixed-point image: q = floor((feature * scaling - lo) * sc) per
feature, integer L1 distance e = sum_k |q_row,k - q_query,k|.  With d the exact scaled L1 distance, |e - d| < 5 (five features,
< 1 each).  T = the MaxNumPoint-th smallest distinct lane minimum of e bounds the MaxNumPoint-th smallest d by T + 5, so a row of
the exact top MaxNumPoint can sit as far out as e = T + 9.  slack_case() builds exactly that situation.
"""
import numpy as np

CHUNK = 1024          # K1_CHUNK
WAVE = 64


def slack_case(E0=40, eps=1e-3, T=400, seed=11):
    """One lap (scaling = 1, all five feature ranges [0, 65535] so the fixed-point scale is exactly 1):
         query        integer features m
         decoys i=0..6: m + n_i + 0.999 per feature, sum(n_i) = E0 + i      -> e = E0 + i,  d = E0 + i + 4.995
         victim        m - n + 1 - eps per feature, sum(n) = E0 + 15        -> e = E0 + 15, d = E0 + 10 + 5 eps
       so T = E0 + 6, e_victim = T + 9 and the exact 7 nearest are decoys 0..5 and the victim (d_victim < d_decoy6)."""
    rng = np.random.default_rng(seed)
    m = np.array([100.0, 120.0, 140.0, 160.0, 180.0])
    feat = rng.uniform(30000.0, 60000.0, size=(T, 5))                 # fillers: far from the query
    feat[0] = 0.0; feat[1] = 65535.0                                  # range rows: lo = 0, widest range = 65535 -> sc = 1
    def split(total, lo):
        while True:
            n = rng.multinomial(total - 5 * lo, np.ones(5) / 5) + lo
            if n.min() >= lo:
                return n.astype(float)
    lanes = rng.permutation(np.arange(2, WAVE))[:8]                   # eight distinct lanes (row % 64), none of the range rows
    rows = [int(l + WAVE * rng.integers(0, (T - 2) // WAVE - 1)) for l in lanes]
    decoys = rows[:7]; victim = rows[7]
    for i, r in enumerate(decoys):
        feat[r] = m + split(E0 + i, 0) + 0.999
    feat[victim] = m - split(E0 + 15, 1) + 1.0 - eps
    x = np.zeros((T, 6)); u = np.zeros((T, 2))
    x[:, 0:3] = feat[:, 0:3]; u[:] = feat[:, 3:5]
    x[:, 3] = 0.0; x[:, 4] = np.linspace(0.1, 5.0, T); x[:, 5] = 0.0
    # next-step targets of the candidate rows: moderate, generic values (the regression reads x[t + 1, 0:3])
    for r in decoys + [victim]:
        assert r + 1 not in decoys + [victim] and r + 1 < T - 1
        x[r + 1, 0:3] = rng.uniform(50.0, 250.0, size=3)
    xq = np.array([m[0], m[1], m[2], 0.0, 1.0, 0.0]); uq = np.array([m[3], m[4]])
    return dict(x=x, u=u, xq=xq, uq=uq, decoys=decoys, victim=victim, h=1.0e6, E0=E0)


def emulate_prefilter(case, maxp=7):
    """Host emulation of quantise_lap (lmpc_capi.hip) + step A of lmpc_regress_kernel for a single-chunk lap with scaling = 1:
    returns the integer distances e, the threshold T, and the exact top-maxp rows (stable argsort of the FP64 distances)."""
    x, u = case["x"], case["u"]
    nrows = x.shape[0] - 1
    assert nrows <= CHUNK
    F = np.hstack([x[:nrows, 0:3], u[:nrows]])
    lo = F.min(0); rmax = (F.max(0) - lo).max(); sc = 65535.0 / rmax
    q = np.minimum(np.maximum((F - lo) * sc, 0.0), 65535.0).astype(np.int64)
    xu = np.hstack([case["xq"][0:3], case["uq"]])
    tq = np.minimum(np.maximum((xu - lo) * sc, 0.0), 65535.0).astype(np.int64)
    e = np.abs(q - tq[None]).sum(1)
    lane_min = np.full(WAVE, np.iinfo(np.int64).max)
    for t in range(nrows):
        lane_min[t % WAVE] = min(lane_min[t % WAVE], e[t])
    Tq = np.unique(lane_min)[maxp - 1]
    d = np.abs(F[:, 0] - xu[0])
    for j in range(1, 5):
        d = d + np.abs(F[:, j] - xu[j])
    return dict(e=e, T=int(Tq), exact_top=list(np.argsort(d, kind="stable")[:maxp]), d=d)


def fit_with_rows(case, rows):
    """A[0:3, 0:3] of the local linear regression (PredictiveModel.py:141-178 arithmetic) with a GIVEN row selection."""
    x, u, h = case["x"], case["u"], case["h"]
    xu = np.hstack([case["xq"][0:3], case["uq"]])
    rows = np.asarray(rows)
    F = np.hstack([x[rows, 0:3], u[rows]])
    d = np.abs(F - xu[None]).sum(1)
    K = np.diag((1 - (d / h) ** 2) * 3 / 4)
    out = np.zeros((3, 3))
    for yi, inp in ((0, 4), (1, 3), (2, 3)):
        M = np.hstack([F[:, 0:3], F[:, inp:inp + 1], np.ones((len(rows), 1))])
        th = np.linalg.solve(M.T @ K @ M, M.T @ K @ x[rows + 1, yi])
        out[yi] = th[0:3]
    return out
