"""racinglmpc_amd -- MI355X-native batched solver for the per-time-step LMPC QP of urosolia/RacingLMPC.

Hot path only: LTV model regression, safe-set selection, block-banded QP
assembly and the QP solve run as hand-written HIP kernels (racinglmpc_amd/csrc) behind the C ABI of
include/lmpc_hip.h; this package is the thin Python/NumPy host side (ctypes only):

  racinglmpc_amd._capi                  ctypes binding + Context (batched API)
  racinglmpc_amd.PredictiveControllers  drop-in MPCParams / MPC / LMPC   (reference fnc/controller/PredictiveControllers.py)
  racinglmpc_amd.PredictiveModel        drop-in PredictiveModel          (reference fnc/controller/PredictiveModel.py)
  racinglmpc_amd.rollout                batched closed-loop rollouts + per-lap exchange across ranks
"""
__version__ = "0.1.0"
