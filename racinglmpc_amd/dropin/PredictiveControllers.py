"""sys.path shim: `from PredictiveControllers import MPC, LMPC, MPCParams` resolves to the GPU drop-in."""
from racinglmpc_amd.PredictiveControllers import MPC, LMPC, MPCParams, PythonMsg  # noqa: F401
