"""Drop-in for `Regression` of the reference's fnc/Utilities.py (the LTI model of main.py's path-following MPC stage,
main.py:74-77): same signature and return value, the ridge least-squares fit runs in lmpc_lti_regress_kernel on the GPU.
`PID` and `wrap` are caller-side helpers of the reference and stay where they are."""
import numpy as np

from . import _capi


def Regression(x, u, lamb):
    """Estimates linear system dynamics x_{k+1} = A x_k + B u_k from one closed-loop lap (reference Utilities.py:5-28).
    Returns A (6,6), B (6,2), Error (2,6) = [max; min] of the fit residual per state."""
    A, B, Error, status = _capi.lti_regression(np.asarray(x, float), np.asarray(u, float), lamb)
    if status & _capi.ST_REG_SINGULAR:
        raise np.linalg.LinAlgError("Singular matrix")          # what np.linalg.inv raises in the reference
    return A, B, Error
