"""tests/golden/make_track_golden.py -- fixture for Map.getGlobalPosition (Track.py:135-189).

Imports and EXECUTES the reference's own `Map` (read-only import from /root/reference, no bytecode written) on a fixed grid of
(s, ey) samples -- several laps of s so that the wrap loop runs, both sides of the centre line, the segment boundaries -- and
stores inputs and outputs in tests/golden/track_xy.npz.  Run in the build container only (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_track_golden.py
"""
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/src/fnc/simulator"
sys.path.insert(0, REF)
from Track import Map  # noqa: E402

m = Map(0.4)
pt = m.PointAndTangent.copy(); TL = float(m.TrackLength)
rng = np.random.default_rng(20240925)
s = np.concatenate([np.linspace(0.0, 3.2 * TL, 600, endpoint=False), rng.uniform(0, 4 * TL, 300),
                    pt[:, 3] + 1e-9, pt[:, 3] + pt[:, 4] - 1e-9])   # just inside the segment boundaries (exactly on one, rounding
#                                                                      can make two segments match and the reference raises)
s = s[(s == 0) | (np.abs(s / TL - np.round(s / TL)) > 1e-12)]   # s == k TrackLength (k >= 1) wraps to TrackLength exactly: no segment, the reference raises
ey = np.concatenate([np.zeros(200), rng.uniform(-0.4, 0.4, s.size - 200)])
xy = np.array([m.getGlobalPosition(float(a), float(b)) for a, b in zip(s, ey)])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "track_xy.npz")
np.savez_compressed(out, track=pt, trackLength=TL, s=s, ey=ey, xy=xy)
print("wrote", out, xy.shape)
