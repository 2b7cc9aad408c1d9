"""tests/golden/make_regression_golden.py -- fixture for Utilities.Regression by EXECUTING the reference function.

Run where /root/reference exists:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_regression_golden.py
Inputs are laps already held by lmpc_n12.npz (the PID seed lap exactly as main.py:74-77 uses it, and the two recorded LMPC laps);
outputs are what /root/reference/src/fnc/Utilities.py:Regression returns for them, unmodified code, lamb as in main.py:75."""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.append("/root/reference/src/fnc")
from Utilities import Regression  # noqa: E402  (the reference's own function)


def main():
    g = np.load(os.path.join(HERE, "lmpc_n12.npz"))
    out = {}
    cases = [("pid", g["xPID"], g["uPID"], 0.0000001), ("lap0", g["lapx0"], g["lapu0"], 0.0000001), ("lap1", g["lapx1"], g["lapu1"], 1e-3),
             ("short", g["xPID"][:40], g["uPID"][:40], 1e-5)]
    for name, x, u, lamb in cases:
        A, B, Error = Regression(x, u, lamb)
        out[name + "_A"], out[name + "_B"], out[name + "_Error"], out[name + "_lamb"] = A, B, Error, lamb
        X = np.hstack((x[1:-1], u[1:-1]))
        out[name + "_cond"] = np.linalg.cond(X.T @ X + lamb * np.eye(8))
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(HERE, "regression.npz"), **out)
    print({c[0]: float(out[c[0] + "_cond"]) for c in cases})


if __name__ == "__main__":
    main()
