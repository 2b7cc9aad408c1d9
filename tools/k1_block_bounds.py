"""Developer tool (CPU): would a coarse index over the lap store let the regression's k-NN scan skip rows?  For the bench batch's queries against the
PID seed lap: the fraction of 16-row / 64-row blocks whose bounding-box lower bound (L1, scaled features) does not exceed the 7th-nearest distance,
i.e. that a scan with such an index would still have to visit.      python tools/k1_block_bounds.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

g = bench.load_seed()
x, u = g["xPID"], g["uPID"]
w = np.array([0.1, 1, 1, 1, 1])                                             # PredictiveModel.py:22-26
F = np.column_stack([x[:-1, :3], u[:x.shape[0] - 1]]) * w                   # rows of the store a query can select (PredictiveModel.py:180-197)
inp = bench.synth_batch(g, 256, 12)
Q = np.column_stack([inp["xLin"][:, :12, :3].reshape(-1, 3), inp["uLin"].reshape(-1, 2)]) * w
for blk in (16, 64):
    nb = (F.shape[0] + blk - 1) // blk
    lo = np.array([F[i * blk:(i + 1) * blk].min(0) for i in range(nb)]); hi = np.array([F[i * blk:(i + 1) * blk].max(0) for i in range(nb)])
    fr = []
    for q in Q:
        T = np.sort(np.abs(F - q).sum(1))[6]
        fr.append(((np.maximum(0, lo - q) + np.maximum(0, q - hi)).sum(1) <= T).mean())
    fr = np.array(fr)
    print("%2d-row blocks: %.0f %% cannot be skipped on average (min %.0f %%, max %.0f %%) over %d queries" % (blk, 100 * fr.mean(), 100 * fr.min(), 100 * fr.max(), len(fr)))
