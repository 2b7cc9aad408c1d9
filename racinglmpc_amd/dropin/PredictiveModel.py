"""sys.path shim: `from PredictiveModel import PredictiveModel` resolves to the GPU drop-in."""
from racinglmpc_amd.PredictiveModel import PredictiveModel  # noqa: F401
