"""Drop-in for the reference's fnc/controller/PredictiveModel.py (class PredictiveModel).

Same constructor, attributes and methods; the arithmetic of regressionAndLinearization
(reference PredictiveModel.py:48-197) runs in lmpc_regress_kernel on the GPU.  The lap data live in the device
store of every controller context attached to this model; the host lists xStored/uStored/lapTime are kept with the
reference's ordering semantics (ascending length, PredictiveModel.py:35-46) because MPC.__init__ reads
xStored[-1] (PredictiveControllers.py:88-91).
"""
import numpy as np

from . import _capi


class PredictiveModel():
    def __init__(self, n, d, map, trToUse):
        self.map = map
        self.n = n
        self.d = d
        self.xStored = []
        self.uStored = []
        self.MaxNumPoint = 7
        self.h = 5
        self.lamb = 0.0
        self.dt = 0.1
        self.scaling = np.diag([0.1, 1.0, 1.0, 1.0, 1.0])
        self.stateFeatures = [0, 1, 2]
        self.inputFeaturesVx = [1]
        self.inputFeaturesLat = [0]
        self.usedIt = [i for i in range(trToUse)]
        self.lapTime = []
        self._calls = []          # addTrajectory history in call order, replayed into contexts created later
        self._sinks = []          # attached lmpc contexts
        self._own_ctx = None

    # -- reference API -----------------------------------------------------------------------------------
    def addTrajectory(self, x, u):
        x = np.asarray(x, dtype=float); u = np.asarray(u, dtype=float)
        if self.lapTime == [] or x.shape[0] >= self.lapTime[-1]:
            self.xStored.append(x); self.uStored.append(u); self.lapTime.append(x.shape[0])
        else:
            for i in range(0, len(self.xStored)):
                if x.shape[0] < self.lapTime[i]:
                    self.xStored.insert(i, x); self.uStored.insert(i, u); self.lapTime.insert(i, x.shape[0])
                    break
        self._calls.append((x, u))
        for ctx in self._sinks:
            ctx.model_add_trajectory(x, u)       # the library repeats the same sorted insert on its slot table

    def regressionAndLinearization(self, x, u):
        """(A_i, B_i, C_i) with x_{k+1} = A_i x_k + B_i u_k + C_i around (x, u).  One GPU launch."""
        ctx = self._any_ctx()
        A, B, C, st = ctx.regress_points(np.asarray(x, float).reshape(1, 6), np.asarray(u, float).reshape(1, 2))      # one query, one work-group (no horizon tiled around it)
        if st[0] & _capi.ST_REG_SINGULAR:
            raise ArithmeticError("local regression is singular (fewer than 5 independent neighbours within h)")
        if st[0] & _capi.ST_NO_SEGMENT:
            raise ValueError("curvature(): s = %r is on no track segment" % (x[4],))
        return A[0], B[0], C[0]

    # -- glue ----------------------------------------------------------------------------------------------
    def _attach(self, ctx):
        """Called by a controller: replay the stored laps into its context and keep it updated."""
        for x, u in self._calls:
            ctx.model_add_trajectory(x, u)
        self._sinks.append(ctx)

    def _any_ctx(self):
        if self._sinks:
            return self._sinks[0]
        if self._own_ctx is None:
            zero6 = np.zeros((6, 6))
            cfg = _capi.config_from(12, zero6, np.zeros((2, 2)), zero6, np.zeros(2), np.array([0., 50.]),
                                    np.array([[0, 0, 0, 0, 0, 1.], [0, 0, 0, 0, 0, -1.]]), [2., 2.],
                                    np.kron(np.eye(2), np.array([1, -1])).T, [0.5, 0.5, 10., 10.], np.zeros(6),
                                    numSS_it=0, trToUse=len(self.usedIt), track=self.map.PointAndTangent,
                                    trackLength=self.map.TrackLength, max_batch=1)
            cfg.maxNumPoint = int(self.MaxNumPoint); cfg.h = float(self.h); cfg.lamb = float(self.lamb); cfg.dt = float(self.dt)
            self._own_ctx = _capi.Context(cfg)
            self._attach(self._own_ctx)
        return self._own_ctx
