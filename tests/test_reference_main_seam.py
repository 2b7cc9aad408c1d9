"""CPU, build container only (skipped where /root/reference does not exist, e.g. on the GPU box): the reference's UNCHANGED main.py runs against
the drop-in Python layer through the sys.path seam -- `from PredictiveControllers import MPC, LMPC, MPCParams`, `from PredictiveModel import
PredictiveModel`, `from Utilities import Regression, PID` (main.py:28-32), `from Utilities import wrap` (SysModel.py:4) all resolve to
racinglmpc_amd/dropin -- with tests/standin_capi.py (oracle arithmetic) in place of the ctypes binding.  Every attribute and method main.py,
SysModel.Simulator.sim and plot.py (plotTrajectory, plotClosedLoopLMPC, animation_xy) touch must exist with the reference's shapes, or the run raises."""
import os

import pytest

from tests import reference_main

pytestmark = pytest.mark.skipif(not reference_main.available(), reason="/root/reference is not present (build container only)")


def test_unchanged_main_py_runs_on_the_dropin_seam():
    r = reference_main.run(quick=True, seed=0, lmpc_laps=2)
    seam = os.path.join(reference_main.ROOT, "racinglmpc_amd", "dropin")
    for m in ("PredictiveControllers", "PredictiveModel", "Utilities"):
        assert os.path.dirname(r["modules"][m]) == seam, r["modules"]                       # the three controller-side modules came from the seam ...
    for m in ("SysModel", "plot"):
        assert r["modules"][m].startswith("/root/reference/"), r["modules"]                 # ... the simulator and the plots are the reference's own files
    names = [c[0] for c in r["calls"]]
    assert names.count("lti_regression") == 1                                                # main.py:74 Regression -> lmpc_lti_regression
    ctxs = [c[1] for c in r["calls"] if c[0] == "Context"]
    assert [c["N"] for c in ctxs] == [14, 14, 14] and [c["numSS_it"] for c in ctxs] == [0, 0, 4]   # MPC, TV-MPC, LMPC at main.py's horizon
    assert names.count("qp_solve_batch") == 60 and names.count("ss_add_trajectory") == 4 + 2 and names.count("model_add_trajectory") == 1 + 4 + 2
    n_lmpc = sum(1 for c in r["calls"] if c[0] == "step_batch" and c[1]["lmpc"])
    assert n_lmpc == names.count("ss_add_point") and n_lmpc > 300                            # Simulator.sim: one addPoint per LMPC solve (SysModel.py:37-38)
    out = r["stdout"]
    assert "===== PID terminated" in out and "===== MPC terminated" in out and "===== TV-MPC terminated" in out and "===== LMPC terminated" in out
    laps = [float(l.split(" in ")[1].split()[0]) for l in out.splitlines() if l.startswith("Completed lap")]
    assert len(laps) == 2 and laps[1] < laps[0] < 30.0                                       # main.py:120 printout, seconds
    assert "===== Start Plotting" in out and r["figures"] >= 5                               # plotTrajectory x3, plotClosedLoopLMPC, animation_xy ran
