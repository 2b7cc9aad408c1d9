"""Developer tool (build container only: needs /root/reference, no GPU): the reference's UNCHANGED src/main.py on the drop-in path seam
(racinglmpc_amd/dropin prepended to sys.path) with tests/standin_capi.py -- the oracle's arithmetic -- in place of the ctypes binding.
    python tools/run_reference_main.py            quick run (MPC stages 6 s, two LMPC laps; ~40 s)
    python tools/run_reference_main.py --full     main.py exactly as it is: 100 s MPC stages, 40 LMPC laps (~10 min)
Prints main.py's own output, the modules the bare imports resolved to, and the histogram of binding calls."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import reference_main

if not reference_main.available():
    sys.exit("needs /root/reference (the build container)")
r = reference_main.run(quick="--full" not in sys.argv)
print(r["stdout"])
print("modules:", r["modules"])
print("binding calls:", dict(collections.Counter(c[0] for c in r["calls"])))
print("figures created by plot.py:", r["figures"])
