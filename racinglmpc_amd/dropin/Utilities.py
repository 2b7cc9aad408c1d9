"""sys.path shim: `from Utilities import Regression, PID` (main.py:32) and `from Utilities import wrap` (SysModel.py:4) resolve to the drop-in."""
from racinglmpc_amd.Utilities import Regression, PID, wrap  # noqa: F401
