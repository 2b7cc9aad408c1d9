"""ctypes binding of liblmpc_hip.so (C ABI: include/lmpc_hip.h).  NumPy in, NumPy out.

The library is the only compute path: importing this module without a built .so, or calling into it
without a working HIP device, raises -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LMPC_LIB") or os.path.join(_HERE, "liblmpc_hip.so")      # LMPC_LIB: developer builds (build.build_flavour)

MAX_TRACK_ROWS = 16
MAX_USED_LAPS = 32
COMM_ID_BYTES = 128
E_VARIANT = -5
CREATE_RUNTIME_KERNEL, CREATE_FORCE_RUNTIME_KERNEL = 1, 2

ST_MAXITER, ST_REG_SINGULAR, ST_NO_SEGMENT, ST_WINDOW, ST_NUMERIC, ST_NOT_INTERIOR, ST_INEXACT, ST_INFEASIBLE = 1, 2, 4, 8, 16, 32, 64, 128


class LmpcConfig(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("numSS_it", C.c_int), ("numSS_points", C.c_int), ("trToUse", C.c_int), ("maxNumPoint", C.c_int),
        ("h", C.c_double), ("lamb", C.c_double), ("dt", C.c_double), ("scaling", C.c_double * 5),
        ("Q", C.c_double * 36), ("R", C.c_double * 4), ("Qf", C.c_double * 36), ("dR", C.c_double * 2), ("Qslack", C.c_double * 2),
        ("QtermSlack", C.c_double * 36), ("xRef", C.c_double * 6),
        ("Fx", C.c_double * 12), ("bx", C.c_double * 2), ("Fu", C.c_double * 8), ("bu", C.c_double * 4),
        ("track", C.c_double * (MAX_TRACK_ROWS * 6)), ("track_rows", C.c_int), ("trackLength", C.c_double),
        ("device", C.c_int), ("max_batch", C.c_int), ("max_laps", C.c_int), ("max_lap_len", C.c_int),
        ("tol_gap", C.c_double), ("tol_res", C.c_double), ("reg_lambda", C.c_double), ("max_iter", C.c_int), ("slacks", C.c_int),
    ]


class LmpcStats(C.Structure):
    _fields_ = [("ms_regress", C.c_double), ("ms_solve", C.c_double), ("n_regress", C.c_longlong), ("n_solve", C.c_longlong),
                ("qp_solved", C.c_longlong), ("ipm_iters", C.c_longlong), ("n_regress_timed", C.c_longlong), ("n_solve_timed", C.c_longlong), ("n_retry", C.c_longlong)]


class StepDevArgs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("x0", "xLin", "uLin", "uOld", "zt", "xPredPrev", "hasPred", "timeStep",
                                          "xPred", "uPred", "slack", "lambda_", "sTerm", "ztNext", "ztuNext", "ssSel",
                                          "A", "Bm", "C", "mu", "resid", "status", "iters", "qSel")]


EXPORTS = [
    "lmpc_config_default", "lmpc_create", "lmpc_create_ex", "lmpc_solver_kind", "lmpc_destroy", "lmpc_last_error", "lmpc_active_knobs", "lmpc_version", "lmpc_device_memory",
    "lmpc_model_add_trajectory", "lmpc_model_num_laps", "lmpc_model_replace_lap",
    "lmpc_ss_add_trajectory", "lmpc_ss_add_point", "lmpc_ss_replace_lap", "lmpc_ss_set_selected", "lmpc_ss_num_laps", "lmpc_ss_get_qfun", "lmpc_ss_get_laptime", "lmpc_store_read_lap",
    "lmpc_regress_batch", "lmpc_regress_points", "lmpc_select_batch", "lmpc_qp_solve_batch", "lmpc_step_batch", "lmpc_assemble_batch", "lmpc_qp_dims",
    "lmpc_dev_alloc", "lmpc_dev_free", "lmpc_dev_upload", "lmpc_dev_download", "lmpc_dev_sync", "lmpc_step_batch_dev",
    "lmpc_lti_regression", "lmpc_comm_unique_id", "lmpc_comm_init", "lmpc_comm_destroy", "lmpc_comm_info", "lmpc_comm_allgather_dev", "lmpc_comm_allgather",
    "lmpc_comm_allreduce_max", "lmpc_comm_barrier", "lmpc_rollout_exchange",
    "lmpc_set_profiling", "lmpc_get_stats", "lmpc_reset_stats", "lmpc_selftest", "lmpc_solver_waves", "lmpc_plant_step_batch", "lmpc_global_position_batch", "lmpc_rollout_begin", "lmpc_rollout_run", "lmpc_rollout_fetch", "lmpc_rollout_end", "lmpc_rollout_release", "lmpc_ss_extend_lap", "lmpc_ss_truncate_lap",
    "lmpc_debug_set_trace", "lmpc_debug_exec_audit", "lmpc_debug_rollout_peek", "lmpc_debug_rollout_capture", "lmpc_debug_rollout_qp",
]

_lib = None


class LmpcError(RuntimeError):
    pass


def load():
    """dlopen liblmpc_hip.so (built by racinglmpc_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LmpcError("liblmpc_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`. "
                            "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(lib, name)          # raises AttributeError if a declared symbol is missing
        lib.lmpc_last_error.restype = C.c_char_p
        lib.lmpc_active_knobs.restype = C.c_char_p
        # (declared argument types let the hot call take plain integers as addresses: no c_void_p object per array)
        lib.lmpc_step_batch.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 24
        _lib = lib
    return _lib


def active_knobs():
    """Developer environment variables the library has acted on in this process ([] = none): route / grid choices, results identical."""
    v = load().lmpc_active_knobs().decode()
    return v.split(";") if v else []


def device_memory(device=0):
    """(free, total) bytes of HBM on `device` (hipMemGetInfo)."""
    f, t = C.c_ulonglong(), C.c_ulonglong()
    _chk(load().lmpc_device_memory(C.c_int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def comm_unique_id():
    """ncclGetUniqueId (call on rank 0, hand the bytes to the other ranks)."""
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _chk(load().lmpc_comm_unique_id(buf))
    return bytes(buf)


def lti_regression(x, u, lamb, device=0):
    """Utilities.Regression on the GPU: returns (A (6,6), B (6,2), Error (2,6), status)."""
    x = np.ascontiguousarray(x, dtype=np.float64); u = np.ascontiguousarray(u, dtype=np.float64)
    assert x.ndim == 2 and x.shape[1] == 6 and u.shape == (x.shape[0], 2), (x.shape, u.shape)
    A = np.zeros((6, 6)); B = np.zeros((6, 2)); E = np.zeros((2, 6)); st = C.c_int()
    _chk(load().lmpc_lti_regression(C.c_int(int(device)), _d(x), _d(u), C.c_int(x.shape[0]), C.c_double(float(lamb)), _d(A), _d(B), _d(E), C.byref(st)))
    return A, B, E, st.value


def _chk(rc):
    if rc != 0:
        raise LmpcError("liblmpc_hip error %d: %s" % (rc, load().lmpc_last_error().decode()))


def _d(a):
    # (c_void_p around the raw address: numpy's data_as goes through ctypes.cast, 3 us per array -- 25 arrays per lmpc_step_batch call)
    return None if a is None else C.c_void_p(a.ctypes.data)


def _p(p):
    """device pointer -> c_void_p (ctypes hands Structure c_void_p fields back as plain ints)."""
    return p if isinstance(p, C.c_void_p) else C.c_void_p(p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def default_config():
    cfg = LmpcConfig()
    _chk(load().lmpc_config_default(C.byref(cfg)))
    return cfg


def set_arr(field, values):
    v = np.asarray(values, dtype=np.float64).reshape(-1)
    assert len(v) == len(field), (len(v), len(field))
    for i, x in enumerate(v):
        field[i] = float(x)


class Context:
    """One lmpc_ctx: device lap stores + batched solver for a fixed (N, safe-set size) configuration."""

    def __init__(self, cfg, runtime_kernel=False):
        self.lib = load()
        self.cfg = cfg
        self.N = cfg.N
        self.S = cfg.numSS_points if cfg.numSS_it > 0 else 0
        self.M = 8 * self.N + self.S
        self._step_plan = {}
        self._h = C.c_void_p()
        self._pid = os.getpid()         # the process that owns the device context (see close)
        if runtime_kernel:              # (tests: the runtime-(N, S) kernel even where a fast one exists)
            rc = self.lib.lmpc_create_ex(C.byref(cfg), C.c_uint(CREATE_FORCE_RUNTIME_KERNEL), C.byref(self._h))
        else:
            rc = self.lib.lmpc_create(C.byref(cfg), C.byref(self._h))
        if rc == E_VARIANT:             # (N, numSS_points) outside the built-in set: compile its shared object once (hipcc, ~20 s), then retry ...
            from . import build
            try:
                build.build_variant(self.N, self.S)
                rc = self.lib.lmpc_create(C.byref(cfg), C.byref(self._h))
            except (OSError, RuntimeError, ValueError, subprocess.SubprocessError) as e:
                # ... and where that is not possible (no hipcc on the box, a failed build): the runtime-(N, S) kernel serves the horizon, several times slower.
                # MPCParams.N is a plain parameter in the reference (PredictiveControllers.py:63-107, main.py:43): LMPC_E_VARIANT never reaches a user.
                warnings.warn("no solve-kernel variant for N = %d, numSS_points = %d (%s): using the runtime-(N, S) kernel" % (self.N, self.S, str(e).splitlines()[0][:120]))
                rc = self.lib.lmpc_create_ex(C.byref(cfg), C.c_uint(CREATE_RUNTIME_KERNEL), C.byref(self._h))
        _chk(rc)
        self.solver_kind = int(self.lib.lmpc_solver_kind(self._h))      # 0 built-in, 1 variant library, 2 runtime-(N, S) kernel

    def close(self):
        # A forked child inherits this object but not a usable HIP runtime: when the child's garbage collector finalises its copy, lmpc_destroy would run HIP calls
        # in a process that must not make any (a segmentation fault in a multiprocessing worker, and a parent waiting for ever on the dead worker: it happened in the
        # worker pools of the GPU tests).  Only the creating process destroys the context.
        if self._h and getattr(self, "_pid", None) == os.getpid():
            self.lib.lmpc_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- stores
    def model_add_trajectory(self, x, u):
        x = _f64(x); u = _f64(u)
        _chk(self.lib.lmpc_model_add_trajectory(self._h, _d(x), _d(u), C.c_int(x.shape[0])))

    def model_replace_lap(self, pos, x, u):
        x = _f64(x); u = _f64(u)
        _chk(self.lib.lmpc_model_replace_lap(self._h, C.c_int(pos), _d(x), _d(u), C.c_int(x.shape[0])))

    def ss_add_trajectory(self, x, u):
        x = _f64(x); u = _f64(u)
        _chk(self.lib.lmpc_ss_add_trajectory(self._h, _d(x), _d(u), C.c_int(x.shape[0])))

    def ss_add_point(self, x, u):
        x = _f64(x); u = _f64(u)
        _chk(self.lib.lmpc_ss_add_point(self._h, _d(x), _d(u)))

    def ss_replace_lap(self, lap, x, u, qfun):
        x = _f64(x); u = _f64(u); q = _f64(qfun)
        _chk(self.lib.lmpc_ss_replace_lap(self._h, C.c_int(lap), _d(x), _d(u), _d(q), C.c_int(x.shape[0])))

    def ss_set_selected(self, laps):
        laps = _i32(laps)
        _chk(self.lib.lmpc_ss_set_selected(self._h, _d(laps), C.c_int(len(laps))))

    def ss_get_qfun(self, lap):
        T = C.c_int()
        _chk(self.lib.lmpc_ss_get_qfun(self._h, C.c_int(lap), None, C.byref(T)))
        q = np.zeros(T.value)
        _chk(self.lib.lmpc_ss_get_qfun(self._h, C.c_int(lap), _d(q), C.byref(T)))
        return q

    # ---- batched compute (host buffers)
    def regress_batch(self, xLin, uLin):
        N = self.N
        xLin = _f64(xLin); uLin = _f64(uLin)
        B = xLin.shape[0]
        stride = xLin.shape[1] * 6
        A = np.zeros((B, N, 6, 6)); Bm = np.zeros((B, N, 6, 2)); Cc = np.zeros((B, N, 6)); st = np.zeros((B, N), np.int32)
        _chk(self.lib.lmpc_regress_batch(self._h, C.c_int(B), _d(xLin), C.c_int(stride), _d(uLin), _d(A), _d(Bm), _d(Cc), _d(st)))
        return A, Bm, Cc, st

    def regress_points(self, x, u):
        """PredictiveModel.regressionAndLinearization for n independent points: (A (n, 6, 6), B (n, 6, 2), C (n, 6), status (n,))."""
        x = _f64(x).reshape(-1, 6); u = _f64(u).reshape(-1, 2); n = x.shape[0]
        A = np.zeros((n, 6, 6)); Bm = np.zeros((n, 6, 2)); Cc = np.zeros((n, 6)); st = np.zeros(n, np.int32)
        _chk(self.lib.lmpc_regress_points(self._h, C.c_int(n), _d(x), _d(u), _d(A), _d(Bm), _d(Cc), _d(st)))
        return A, Bm, Cc, st

    def select_batch(self, x0, zt, xPredPrev=None, hasPred=None, timeStep=None):
        N, S = self.N, self.S
        x0 = _f64(x0); zt = _f64(zt); B = x0.shape[0]
        xpp = None if xPredPrev is None else _f64(xPredPrev)
        hp = None if hasPred is None else _i32(hasPred)
        ts = None if timeStep is None else _i32(timeStep)
        ss = np.zeros((B, S, 6)); q = np.zeros((B, S)); succ = np.zeros((B, S, 6)); succU = np.zeros((B, S, 2)); ztu = np.zeros((B, 6))
        st = np.zeros(B, np.int32); start = np.zeros((B, max(self.cfg.numSS_it, 1)), np.int32)
        _chk(self.lib.lmpc_select_batch(self._h, C.c_int(B), _d(x0), _d(zt), _d(xpp), _d(hp), _d(ts), _d(ss), _d(q), _d(succ), _d(succU), _d(ztu), _d(start), _d(st)))
        return dict(ssSel=ss, qSel=q, succ=succ, succU=succU, ztUsed=ztu, selStart=start, status=st)

    def qp_solve_batch(self, A, Bm, Cc, x0, uOld, ssSel=None, qSel=None):
        N, S, M = self.N, self.S, self.M
        A = _f64(A); Bm = _f64(Bm); Cc = _f64(Cc); x0 = _f64(x0); uOld = _f64(uOld); B = x0.shape[0]
        ss = None if ssSel is None else _f64(ssSel); q = None if qSel is None else _f64(qSel)
        out = dict(xPred=np.zeros((B, N + 1, 6)), uPred=np.zeros((B, N, 2)), slack=np.zeros((B, 2 * N)), lambd=np.zeros((B, S)),
                   sTerm=np.zeros((B, 6)), mu=np.zeros((B, M)), status=np.zeros(B, np.int32), iters=np.zeros(B, np.int32), resid=np.zeros((B, 3)))
        _chk(self.lib.lmpc_qp_solve_batch(self._h, C.c_int(B), _d(A), _d(Bm), _d(Cc), _d(x0), _d(uOld), _d(ss), _d(q),
                                          _d(out["xPred"]), _d(out["uPred"]), _d(out["slack"]), _d(out["lambd"]), _d(out["sTerm"]), _d(out["mu"]),
                                          _d(out["status"]), _d(out["iters"]), _d(out["resid"])))
        return out

    def step_batch(self, x0, xLin, uLin, uOld, zt=None, xPredPrev=None, hasPred=None, timeStep=None):
        N, S = self.N, self.S
        x0 = _f64(x0); xLin = _f64(xLin); uLin = _f64(uLin); uOld = _f64(uOld); B = x0.shape[0]
        assert xLin.shape == (B, N + 1, 6) and uLin.shape == (B, N, 2), (xLin.shape, uLin.shape)
        ztc = None if zt is None else _f64(zt)
        xpp = None if xPredPrev is None else _f64(xPredPrev)
        hp = None if hasPred is None else _i32(hasPred)
        ts = None if timeStep is None else _i32(timeStep)
        # the sixteen outputs are ranges of ONE fresh float64 buffer and one int32 buffer (a drop-in LMPC.solve spends more time in sixteen
        # allocations and twenty-six address look-ups than the library spends outside its two kernels): addresses are base + offset
        plan = self._step_plan.get(B)
        if plan is None:
            shapes = (("xPred", (B, N + 1, 6)), ("uPred", (B, N, 2)), ("slack", (B, 2 * N)), ("lambd", (B, S)), ("sTerm", (B, 6)), ("ztNext", (B, 6)),
                      ("ztuNext", (B, 2)), ("ssSel", (B, S, 6)), ("qSel", (B, S)), ("mu", (B, self.M)), ("A", (B, N, 6, 6)), ("B", (B, N, 6, 2)),
                      ("C", (B, N, 6)), ("resid", (B, 3)))
            offs, o = [], 0
            for k, shp in shapes:
                n = int(np.prod(shp)); offs.append((k, shp, o, n)); o += n
            plan = self._step_plan[B] = (offs, o)
        offs, total = plan
        buf = np.zeros(total); ibuf = np.zeros(2 * B, np.int32)
        base = buf.ctypes.data; ibase = ibuf.ctypes.data
        out = {k: buf[o:o + n].reshape(shp) for k, shp, o, n in offs}
        out["status"] = ibuf[:B]; out["iters"] = ibuf[B:]
        adr = {k: base + 8 * o for k, shp, o, n in offs}
        pa = lambda a: None if a is None else a.ctypes.data
        _chk(self.lib.lmpc_step_batch(self._h, B, pa(x0), pa(xLin), pa(uLin), pa(uOld), pa(ztc), pa(xpp), pa(hp), pa(ts),
                                      adr["xPred"], adr["uPred"], adr["slack"], adr["lambd"], adr["sTerm"], adr["ztNext"], adr["ztuNext"], adr["ssSel"], adr["qSel"], adr["mu"],
                                      adr["A"], adr["B"], adr["C"], ibase, ibase + 4 * B, adr["resid"]))
        return out

    def qp_dims(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _chk(self.lib.lmpc_qp_dims(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def assemble_batch(self, A, Bm, Cc, x0, uOld, ssSel=None, qSel=None):
        nz, mi, me = self.qp_dims(); m = mi + me
        A = _f64(A); Bm = _f64(Bm); Cc = _f64(Cc); x0 = _f64(x0); uOld = _f64(uOld); B = x0.shape[0]
        ss = None if ssSel is None else _f64(ssSel); q = None if qSel is None else _f64(qSel)
        P = np.zeros((B, nz, nz)); qv = np.zeros((B, nz)); Ad = np.zeros((B, m, nz)); l = np.zeros((B, m)); u = np.zeros((B, m))
        _chk(self.lib.lmpc_assemble_batch(self._h, C.c_int(B), _d(A), _d(Bm), _d(Cc), _d(x0), _d(uOld), _d(ss), _d(q), _d(P), _d(qv), _d(Ad), _d(l), _d(u)))
        return P, qv, Ad, l, u

    # ---- device-resident path
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        _chk(self.lib.lmpc_dev_alloc(self._h, C.c_longlong(int(nbytes)), C.byref(p)))
        return p

    def dev_free(self, p):
        _chk(self.lib.lmpc_dev_free(self._h, _p(p)))

    def dev_upload(self, p, arr):
        arr = np.ascontiguousarray(arr)
        _chk(self.lib.lmpc_dev_upload(self._h, _p(p), _d(arr), C.c_longlong(arr.nbytes)))

    def dev_download(self, p, arr):
        assert arr.flags["C_CONTIGUOUS"]
        _chk(self.lib.lmpc_dev_download(self._h, _d(arr), _p(p), C.c_longlong(arr.nbytes)))
        return arr

    def dev_array(self, arr):
        """Allocate HBM for `arr` and upload it; returns the device pointer."""
        arr = np.ascontiguousarray(arr)
        p = self.dev_alloc(max(arr.nbytes, 8))
        if arr.nbytes:
            self.dev_upload(p, arr)
        return p

    def sync(self):
        _chk(self.lib.lmpc_dev_sync(self._h))

    def step_batch_dev(self, B, args):
        _chk(self.lib.lmpc_step_batch_dev(self._h, C.c_int(B), C.byref(args)))

    def step_dev_buffers(self, inp, diagnostics=True):
        """HBM-resident inputs/outputs of lmpc_step_batch_dev for a batch given as host arrays (keys x0, xLin, uLin, uOld, zt,
        xPredPrev, hasPred, timeStep).  Returns (StepDevArgs, device pointers to free with dev_free).  diagnostics=False leaves out what
        the hot path does not need downstream (inequality multipliers mu, residual triple, Q-function values of the selection):
        those pointers stay NULL and the kernels skip the stores."""
        N, S, M = self.N, self.S, self.M
        B = np.asarray(inp["x0"]).shape[0]
        a = StepDevArgs(); keep = []

        def up(arr, dt):
            p = self.dev_array(np.ascontiguousarray(arr, dtype=dt)); keep.append(p); return p

        def alloc(nbytes):
            p = self.dev_alloc(max(int(nbytes), 8)); keep.append(p); return p
        a.x0, a.xLin, a.uLin, a.uOld = up(inp["x0"], np.float64), up(inp["xLin"], np.float64), up(inp["uLin"], np.float64), up(inp["uOld"], np.float64)
        a.zt = up(inp["zt"] if inp.get("zt") is not None else np.zeros((B, 6)), np.float64)
        a.xPredPrev = up(inp["xPredPrev"] if inp.get("xPredPrev") is not None else np.zeros((B, N + 1, 6)), np.float64)
        a.hasPred = up(inp["hasPred"] if inp.get("hasPred") is not None else np.zeros(B), np.int32)
        a.timeStep = up(inp["timeStep"] if inp.get("timeStep") is not None else np.zeros(B), np.int32)
        a.xPred, a.uPred, a.slack = alloc(B * (N + 1) * 6 * 8), alloc(B * N * 2 * 8), alloc(B * N * 2 * 8)
        a.lambda_, a.sTerm, a.ztNext, a.ztuNext = alloc(B * S * 8), alloc(B * 6 * 8), alloc(B * 6 * 8), alloc(B * 2 * 8)
        a.ssSel = alloc(B * S * 6 * 8)
        a.status, a.iters = alloc(B * 4), alloc(B * 4)
        # A_i / B_i / C_i (MPC.A / B / C of the reference): the hand-over from the regression kernel to the solve kernel.  Caller-owned, so that
        # launches can be queued back to back (a launch that used the context's own hand-over buffers is drained before the next one overwrites them)
        a.A, a.Bm, a.C = alloc(B * N * 36 * 8), alloc(B * N * 12 * 8), alloc(B * N * 6 * 8)
        if diagnostics:
            a.qSel, a.mu, a.resid = alloc(B * S * 8), alloc(B * M * 8), alloc(B * 3 * 8)
        return a, keep

    def step_dev_fetch(self, a, B):
        """Download every output of a finished lmpc_step_batch_dev call (same keys as step_batch)."""
        N, S, M = self.N, self.S, self.M
        self.sync()
        out = dict(xPred=np.zeros((B, N + 1, 6)), uPred=np.zeros((B, N, 2)), slack=np.zeros((B, 2 * N)), lambd=np.zeros((B, S)),
                   sTerm=np.zeros((B, 6)), ztNext=np.zeros((B, 6)), ztuNext=np.zeros((B, 2)), ssSel=np.zeros((B, S, 6)), qSel=np.zeros((B, S)),
                   mu=np.zeros((B, M)), A=np.zeros((B, N, 6, 6)), B=np.zeros((B, N, 6, 2)), C=np.zeros((B, N, 6)),
                   status=np.zeros(B, np.int32), iters=np.zeros(B, np.int32), resid=np.zeros((B, 3)))
        src = dict(xPred=a.xPred, uPred=a.uPred, slack=a.slack, lambd=a.lambda_, sTerm=a.sTerm, ztNext=a.ztNext, ztuNext=a.ztuNext, ssSel=a.ssSel,
                   qSel=a.qSel, mu=a.mu, A=a.A, B=a.Bm, C=a.C, status=a.status, iters=a.iters, resid=a.resid)
        for k, arr in out.items():
            if arr.nbytes and src[k]:                      # (diagnostic outputs that were not requested stay zero)
                self.dev_download(src[k], arr)
        return out

    def plant_step_batch(self, x, x_glob, u, noise):
        x = _f64(x); xg = _f64(x_glob); u = _f64(u); nz = _f64(noise); B = x.shape[0]
        xn = np.zeros((B, 6)); xgn = np.zeros((B, 6)); st = np.zeros(B, np.int32)
        _chk(self.lib.lmpc_plant_step_batch(self._h, C.c_int(B), _d(x), _d(xg), _d(u), _d(nz), _d(xn), _d(xgn), _d(st)))
        return xn, xgn, st

    def rollout_begin(self, x0, xglob0, xLin0, uLin0, noise):
        x0 = _f64(x0); xg = _f64(xglob0); xl = _f64(xLin0); ul = _f64(uLin0); nz = _f64(noise)
        B = x0.shape[0]
        assert xl.shape == (B, self.N + 1, 6) and ul.shape == (B, self.N, 2) and nz.shape[1:] == (B, 3)
        self._ro = (B, nz.shape[0]); self._ro_t = 0
        _chk(self.lib.lmpc_rollout_begin(self._h, C.c_int(B), C.c_int(nz.shape[0]), _d(x0), _d(xg), _d(xl), _d(ul), _d(nz)))

    def rollout_run(self, max_steps):
        t = C.c_int(); nd = C.c_int()
        _chk(self.lib.lmpc_rollout_run(self._h, C.c_int(int(max_steps)), C.byref(t), C.byref(nd)))
        self._ro_t = t.value                      # simulated steps so far (upper bound for rollout_fetch)
        return t.value, nd.value

    def rollout_fetch(self, t0, t1):
        B = self._ro[0]; n = t1 - t0
        X = np.zeros((n, B, 6)); U = np.zeros((n, B, 2)); G = np.zeros((n, B, 6))
        done = np.zeros(B, np.int32); st = np.zeros(B, np.int32); fx = np.zeros((B, 6)); fg = np.zeros((B, 6))
        _chk(self.lib.lmpc_rollout_fetch(self._h, C.c_int(t0), C.c_int(t1), _d(X), _d(U), _d(G), _d(done), _d(st), _d(fx), _d(fg)))
        return X, U, G, done, st, fx, fg

    def rollout_end(self):
        _chk(self.lib.lmpc_rollout_end(self._h))

    # ---- checkpoint / resume of the lap stores (optional .npz dump; no reference counterpart beyond an unused `import pickle`, main.py:36) ----
    def store_read_lap(self, store, lap):
        """(x (T, 6), u (T, 2), qfun (T,) or None) of one stored lap: store 0 = regression store in its sorted order, 1 = safe set in addTrajectory order."""
        T = C.c_int()
        _chk(self.lib.lmpc_store_read_lap(self._h, C.c_int(store), C.c_int(int(lap)), None, None, None, C.byref(T)))
        x = np.zeros((T.value, 6)); u = np.zeros((T.value, 2)); q = np.zeros(T.value) if store == 1 else None
        _chk(self.lib.lmpc_store_read_lap(self._h, C.c_int(store), C.c_int(int(lap)), _d(x), _d(u), _d(q), C.byref(T)))
        return x, u, q

    def save_stores(self, path):
        """Both lap stores (and the explicit safe-set selection, if one is set by the caller: not stored -- it is per-step state of the controller) as one .npz."""
        out = {}
        nm = C.c_int(); _chk(self.lib.lmpc_model_num_laps(self._h, C.byref(nm)))
        for i in range(nm.value):
            x, u, _ = self.store_read_lap(0, i); out["model_x%d" % i] = x; out["model_u%d" % i] = u
        ns = self.ss_num_laps()
        for i in range(ns):
            x, u, q = self.store_read_lap(1, i); out["ss_x%d" % i] = x; out["ss_u%d" % i] = u; out["ss_q%d" % i] = q
            out["ss_laptime%d" % i] = np.int64(self.ss_lap_time(i))
        np.savez_compressed(path, n_model=np.int64(nm.value), n_ss=np.int64(ns), N=np.int64(self.N), **out)

    def restore_stores(self, path):
        """Into a context whose stores are EMPTY: the regression laps in their sorted order (a stable sorted insert keeps it), the safe-set laps at their addTrajectory-time
        length followed by the rows addPoint had appended and their Q-function.  The restored context answers every later call bit for bit like the one that was saved."""
        with np.load(path) as d:
            nm = C.c_int(); _chk(self.lib.lmpc_model_num_laps(self._h, C.byref(nm)))
            if nm.value or self.ss_num_laps():
                raise LmpcError("restore_stores needs a context with empty lap stores")
            for i in range(int(d["n_model"])):
                self.model_add_trajectory(d["model_x%d" % i], d["model_u%d" % i])
            for i in range(int(d["n_ss"])):
                x, u, q, T0 = d["ss_x%d" % i], d["ss_u%d" % i], d["ss_q%d" % i], int(d["ss_laptime%d" % i])
                self.ss_add_trajectory(x[:T0], u[:T0])
                self.ss_replace_lap(i, x, u, q)          # (always: the saved rows and Q-function, whatever computeCost makes of the first T0 rows)

    def rollout_release(self):
        """Free the device buffers a finished session keeps for the next lap (lmpc_rollout_release)."""
        _chk(self.lib.lmpc_rollout_release(self._h))

    def debug_rollout_capture(self, on=True):
        """Sessions begun after this call keep the selected safe-set points of every step (lmpc_debug_rollout_capture)."""
        _chk(self.lib.lmpc_debug_rollout_capture(self._h, C.c_int(1 if on else 0)))

    def debug_rollout_qp(self, b0, n, selection=True):
        """The QP of the last simulated step for rollouts b0 .. b0 + n - 1: dict(A, B, C, xPred, uPred, lam, ztNext, ztuNext, iters, status[, ssSel (n, S, 6), qSel, succ (n, S, 6),
        succU (n, S, 2)]) -- lmpc_debug_rollout_qp."""
        N, S = self.N, self.cfg.numSS_points
        o = dict(A=np.zeros((n, N, 6, 6)), B=np.zeros((n, N, 6, 2)), C=np.zeros((n, N, 6)), xPred=np.zeros((n, N + 1, 6)), uPred=np.zeros((n, N, 2)), lam=np.zeros((n, S)),
                 ztNext=np.zeros((n, 6)), ztuNext=np.zeros((n, 2)), iters=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
        if selection:
            o.update(ssSel=np.zeros((n, S, 6)), qSel=np.zeros((n, S)), succ=np.zeros((n, S, 6)), succU=np.zeros((n, S, 2)))
        g = lambda k: _d(o[k]) if k in o else None
        _chk(self.lib.lmpc_debug_rollout_qp(self._h, C.c_int(int(b0)), C.c_int(int(n)), g("A"), g("B"), g("C"), g("xPred"), g("uPred"), g("ssSel"), g("qSel"), g("succ"), g("succU"),
                                            g("lam"), g("ztNext"), g("ztuNext"), g("iters"), g("status")))
        return o

    def ss_extend_lap(self, lap, x, u):
        x = _f64(x); u = _f64(u)
        _chk(self.lib.lmpc_ss_extend_lap(self._h, C.c_int(int(lap)), _d(x), _d(u), C.c_int(x.shape[0])))

    def ss_truncate_lap(self, lap, T):
        _chk(self.lib.lmpc_ss_truncate_lap(self._h, C.c_int(int(lap)), C.c_int(int(T))))

    def ss_num_laps(self):
        n = C.c_int()
        _chk(self.lib.lmpc_ss_num_laps(self._h, C.byref(n)))
        return n.value

    def ss_lap_time(self, lap):
        """LMPC.LapTime[lap]: rows of the stored lap when it was added (addPoint extensions not counted)."""
        T = C.c_int()
        _chk(self.lib.lmpc_ss_get_laptime(self._h, C.c_int(int(lap)), C.byref(T)))
        return T.value

    def ss_lap_rows(self, lap):
        T = C.c_int()
        _chk(self.lib.lmpc_ss_get_qfun(self._h, C.c_int(int(lap)), None, C.byref(T)))
        return T.value

    # ---- multi-GPU exchange (RCCL behind the C ABI; see parallel.py for the rendezvous)
    def comm_init(self, id_bytes, rank, world):
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _chk(self.lib.lmpc_comm_init(self._h, buf, C.c_int(int(rank)), C.c_int(int(world))))

    def comm_destroy(self):
        _chk(self.lib.lmpc_comm_destroy(self._h))

    def comm_info(self):
        r, w, f = C.c_int(), C.c_int(), C.c_int()
        _chk(self.lib.lmpc_comm_info(self._h, C.byref(r), C.byref(w), C.byref(f)))
        return r.value, w.value, bool(f.value)

    def comm_allgather(self, arr):
        """Every rank's `arr` (same shape / dtype everywhere), stacked along a new leading axis of length world."""
        arr = np.ascontiguousarray(arr)
        world = self.comm_info()[1]
        out = np.empty((world,) + arr.shape, dtype=arr.dtype)
        _chk(self.lib.lmpc_comm_allgather(self._h, _d(arr), _d(out), C.c_longlong(arr.nbytes)))
        return out

    def comm_allreduce_max(self, values):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        _chk(self.lib.lmpc_comm_allreduce_max(self._h, _d(v), C.c_int(v.shape[0])))
        return v

    def comm_barrier(self):
        _chk(self.lib.lmpc_comm_barrier(self._h))

    def rollout_exchange(self, K, T_max):
        """Per-lap exchange of the current rollout session (device-packed records, one ncclAllGather).
        Returns (records (world, K, T_max + 1, 14), lens (world, K), number of valid laps on this rank)."""
        world = self.comm_info()[1]
        rec = np.zeros((world, K, T_max + 1, 14)); ln = np.zeros((world, K), dtype=np.int64); nv = C.c_int()
        _chk(self.lib.lmpc_rollout_exchange(self._h, C.c_int(int(K)), C.c_int(int(T_max)), _d(rec), _d(ln), C.byref(nv)))
        return rec, ln, nv.value

    def solver_waves(self, B):
        return int(self.lib.lmpc_solver_waves(self._h, C.c_int(int(B))))

    def global_position_batch(self, s, ey):
        """Map.getGlobalPosition for arrays of (s, ey): returns xy (n, 2) and status (n,)."""
        s = _f64(np.ravel(s)); ey = _f64(np.ravel(ey)); n = s.shape[0]
        xy = np.zeros((n, 2)); st = np.zeros(n, np.int32)
        _chk(self.lib.lmpc_global_position_batch(self._h, C.c_int(n), _d(s), _d(ey), _d(xy), _d(st)))
        return xy, st

    def selftest(self):
        _chk(self.lib.lmpc_selftest(self._h))

    # ---- developer flavours only (build.build_flavour with LMPC_TRACE / LMPC_EXEC_AUDIT; the product library refuses both calls)
    TRACE_ROWS = 48

    def debug_trace_begin(self, B):
        """Device buffer for the per-iteration trace of the next solve launches of up to B problems; returns its handle for debug_trace_fetch."""
        p = self.dev_alloc(B * self.TRACE_ROWS * 6 * 8)
        self.dev_upload(p, np.full((B, self.TRACE_ROWS, 6), np.nan))
        _chk(self.lib.lmpc_debug_set_trace(self._h, _p(p)))
        return p

    def debug_trace_fetch(self, p, B):
        """(B, 48, 6) rows (gap, r_d, r_e, sigma, alpha_p, alpha_d) per iteration, NaN where no iteration ran; switches the trace off and frees the buffer."""
        out = np.zeros((B, self.TRACE_ROWS, 6))
        self.sync(); self.dev_download(p, out)
        _chk(self.lib.lmpc_debug_set_trace(self._h, None)); self.dev_free(p)
        return out

    def debug_rollout_peek(self, B):
        """(xLin (B, N+1, 6), uLin (B, N, 2), status (B,), rstatus (B, N)) of the running rollout session, see lmpc_debug_rollout_peek."""
        N = self.N
        xl = np.zeros((B, N + 1, 6)); ul = np.zeros((B, N, 2)); st = np.zeros(B, np.int32); rs = np.zeros((B, N), np.int32)
        _chk(self.lib.lmpc_debug_rollout_peek(self._h, _d(xl), _d(ul), _d(st), _d(rs)))
        return xl, ul, st, rs

    def debug_exec_audit(self, reset=True):
        """(partial[8], calls[8]) of the cross-lane primitives since the last reset (sites: csrc/lmpc_kernels.hip.h)."""
        out = (C.c_ulonglong * 16)()
        _chk(self.lib.lmpc_debug_exec_audit(self._h, out, C.c_int(1 if reset else 0)))
        v = np.array(list(out), dtype=np.uint64)
        return v[:8], v[8:]

    def set_profiling(self, every):
        """False / 0: off; True / 1: events around every kernel launch; k: around every k-th launch of each kernel (see lmpc_set_profiling)."""
        _chk(self.lib.lmpc_set_profiling(self._h, C.c_int(int(every))))

    def stats(self):
        s = LmpcStats()
        _chk(self.lib.lmpc_get_stats(self._h, C.byref(s)))
        return s

    def reset_stats(self):
        _chk(self.lib.lmpc_reset_stats(self._h))


def config_from(N, Q, R, Qf, dR, Qslack, Fx, bx, Fu, bu, xRef, QterminalSlack=None, numSS_Points=0, numSS_it=0, trToUse=0,
                track=None, trackLength=0.0, max_batch=256, max_laps=64, max_lap_len=2048, device=0, slacks=True, **solver):
    """Build an LmpcConfig from the numeric content of MPCParams / LMPC ctor args / Map."""
    cfg = default_config()
    cfg.N = int(N); cfg.numSS_it = int(numSS_it); cfg.numSS_points = int(numSS_Points) if numSS_it else 0; cfg.trToUse = int(trToUse)
    set_arr(cfg.Q, np.asarray(Q, float)); set_arr(cfg.R, np.asarray(R, float)); set_arr(cfg.Qf, np.asarray(Qf, float))
    set_arr(cfg.dR, np.asarray(dR, float)); set_arr(cfg.Qslack, np.asarray(Qslack, float)); set_arr(cfg.xRef, np.asarray(xRef, float))
    Fx = np.asarray(Fx, float); Fu = np.asarray(Fu, float)
    if Fx.shape != (2, 6) or Fu.shape != (4, 2):
        raise LmpcError("only the reference's constraint shapes are supported: Fx (2,6), Fu (4,2); got %s %s" % (Fx.shape, Fu.shape))
    set_arr(cfg.Fx, Fx); set_arr(cfg.Fu, Fu)
    set_arr(cfg.bx, np.squeeze(np.asarray(bx, float))); set_arr(cfg.bu, np.squeeze(np.asarray(bu, float)))
    if QterminalSlack is not None:
        set_arr(cfg.QtermSlack, np.asarray(QterminalSlack, float))
    if track is not None:
        track = np.asarray(track, float)
        if track.shape[0] > MAX_TRACK_ROWS:
            raise LmpcError("track table too long")
        for i, v in enumerate(track.reshape(-1)):
            cfg.track[i] = float(v)
        cfg.track_rows = track.shape[0]
    cfg.trackLength = float(trackLength)
    cfg.max_batch, cfg.max_laps, cfg.max_lap_len, cfg.device = int(max_batch), int(max_laps), int(max_lap_len), int(device)
    cfg.slacks = 1 if slacks else 0
    for k, v in solver.items():
        setattr(cfg, k, v)
    return cfg


class ContextPool:
    """`depth` contexts on one device -- one HIP stream, one set of work buffers, one copy of the (small) lap stores each -- for callers that keep several INDEPENDENT
    batches in flight.  Inside one launch the CUs whose QP has converged idle until the slowest QP of the batch has (mean 8.6 against a maximum of 13 iterations at batch
    256); with the next batch queued on another stream its work-groups start on those CUs: 1.21 -> 1.66 M solves/s at batch 256 with three in flight, 1.84 -> 3.14 M at
    batch 512 (tools/pipelined_bench.py).  Store edits go to every member; `step_batch_dev` deals the steps to the members in turn and returns the member it used.
    A closed loop -- step t + 1 needs step t -- cannot use it; batched requests can."""

    def __init__(self, cfg, depth=2):
        if int(depth) < 1:
            raise ValueError("ContextPool: depth must be at least 1")
        self.members = []; self._next = 0
        try:
            for _ in range(int(depth)):
                self.members.append(Context(cfg))
        except Exception:
            self.close()                               # (a member that failed to come up does not leave the earlier ones' device memory behind)
            raise

    def __getattr__(self, name):                       # lap-store edits (model_add_trajectory, ss_add_trajectory, ss_add_point, ss_set_selected, ...): the same call on every member
        if name in ("members", "_next"):               # (not set yet: a failed __init__ must not recurse through this hook)
            raise AttributeError(name)
        if name.startswith(("model_", "ss_")) and not name.startswith(("ss_get", "ss_num", "ss_lap", "model_num")):
            def forward(*a, **kw):
                out = None
                for m in self.members:
                    out = getattr(m, name)(*a, **kw)
                return out
            return forward
        return getattr(self.members[0], name)

    def step_dev_buffers(self, inp, diagnostics=True):
        """One set of device buffers per member: [(args, allocations)], in member order."""
        return [m.step_dev_buffers(inp, diagnostics=diagnostics) for m in self.members]

    def step_batch_dev(self, B, args_per_member):
        i = self._next; self._next = (i + 1) % len(self.members)
        self.members[i].step_batch_dev(B, args_per_member[i][0])
        return i

    def sync(self):
        for m in self.members:
            m.sync()

    def close(self):
        for m in self.members:
            m.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
