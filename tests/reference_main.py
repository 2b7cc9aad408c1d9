"""tests/reference_main.py -- run the reference's unchanged src/main.py with racinglmpc_amd/dropin prepended to sys.path (INTEGRATION.md's path
seam) and tests/standin_capi.py in place of the ctypes binding: no GPU needed, needs /root/reference (build container only).

quick=True shortens the run WITHOUT touching the reference's files: the MPC / TV-MPC stages simulate 6 s instead of 100 s (Simulator.sim's
maxSimTime default, through a subclass installed in main's namespace) and initLMPCParams reports Laps = numSS_it + 2."""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
SEAM = os.path.join(ROOT, "racinglmpc_amd", "dropin")
_SHADOWED = ("PredictiveControllers", "PredictiveModel", "Utilities", "SysModel", "Track", "plot", "initControllerParameters",
             "racinglmpc_amd.PredictiveControllers", "racinglmpc_amd.PredictiveModel", "racinglmpc_amd.Utilities", "racinglmpc_amd._capi")


def available():
    return os.path.exists(os.path.join(REF, "main.py"))


def run(quick=True, seed=0, lmpc_laps=2):
    """Returns dict(calls, stdout, main (the module), lmpc_lap_times)."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True                                  # nothing may be written into /root/reference
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from tests import standin_capi
    saved_mods = {k: sys.modules.pop(k) for k in _SHADOWED if k in sys.modules}
    saved_path = list(sys.path)
    import racinglmpc_amd
    saved_attr = getattr(racinglmpc_amd, "_capi", None)
    sys.modules["racinglmpc_amd._capi"] = standin_capi; racinglmpc_amd._capi = standin_capi
    del standin_capi.CALLS[:]
    sys.path[:0] = [SEAM, REF, os.path.join(REF, "fnc", "simulator"), os.path.join(REF, "fnc", "controller"), os.path.join(REF, "fnc")]
    out = io.StringIO()
    try:
        spec = importlib.util.spec_from_file_location("reference_main", os.path.join(REF, "main.py"))
        mod = importlib.util.module_from_spec(spec)
        with contextlib.redirect_stdout(out):
            spec.loader.exec_module(mod)
            if quick:
                RefSim, ref_init = mod.Simulator, mod.initLMPCParams

                class QuickSim(RefSim):                              # the reference's own Simulator; only sim()'s default length changes for the two MPC stages
                    def sim(self, x0, Controller, maxSimTime=100):
                        short = not self.flagLMPC and not isinstance(Controller, mod.PID)
                        return RefSim.sim(self, x0, Controller, 6 if short else maxSimTime)

                def quick_init(map, N):
                    numSS_it, numSS_Points, Laps, TimeLMPC, QterminalSlack, par = ref_init(map, N)
                    return numSS_it, numSS_Points, numSS_it + lmpc_laps, TimeLMPC, QterminalSlack, par
                mod.Simulator, mod.initLMPCParams = QuickSim, quick_init
            np.random.seed(seed)
            mod.main()
        import matplotlib.pyplot as plt
        nfig = len(plt.get_fignums()); plt.close("all")
        res = dict(calls=list(standin_capi.CALLS), stdout=out.getvalue(), main=mod, figures=nfig,
                   modules={k: getattr(sys.modules.get(k), "__file__", None) for k in ("PredictiveControllers", "PredictiveModel", "Utilities", "SysModel", "plot")})
    finally:
        sys.path[:] = saved_path
        for k in _SHADOWED:
            sys.modules.pop(k, None)
        sys.modules.update(saved_mods)
        if saved_attr is not None:
            racinglmpc_amd._capi = saved_attr
        elif hasattr(racinglmpc_amd, "_capi"):
            del racinglmpc_amd._capi
    return res
