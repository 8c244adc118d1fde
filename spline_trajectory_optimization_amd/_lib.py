"""ctypes binding of librl_mincurv.so (include/rl_mincurv.h).

The HIP library is the product; there is no CPU fallback anywhere in this package.  If the shared
object is missing, or no MI355X/HIP device is usable, every compute entry point raises.
"""
import ctypes
import os
import threading

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RL_LIB_PATH") or os.path.join(_PKG, "librl_mincurv.so")  # RL_LIB_PATH: dev hook for A/B builds

NCOL = 19
MAX_ITER = 32
BOUNDS_SHARED_RINGS, BOUNDS_WIDTHS, BOUNDS_POINTS = 0, 1, 2
SEARCH_BRUTE, SEARCH_CULLED, SEARCH_WINDOWED = 0, 1, 2
ARITH_DEFAULT, ARITH_FAST, ARITH_REFERENCE, ARITH_BRANCH = -1, 0, 1, 2   # include/rl_mincurv.h: RL_ARITH_*

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_vp = ctypes.c_void_p


class Stats(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_float), ("lds_bytes", ctypes.c_int),
                ("block_threads", ctypes.c_int), ("rings_in_lds", ctypes.c_int),
                ("reserved", ctypes.c_int * 4)]


class RlError(RuntimeError):
    pass


# every symbol include/rl_mincurv.h declares: (restype, argtypes)
_SIGNATURES = {
    "rl_version": (ctypes.c_int, []),
    "rl_last_error": (ctypes.c_char_p, []),
    "rl_debug_dump_enable": (ctypes.c_int, [ctypes.c_int]),
    "rl_debug_read": (ctypes.c_int, [_dp, ctypes.c_longlong]),
    "rl_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    "rl_ctx_destroy": (None, [_vp]),
    "rl_ctx_set_stream": (ctypes.c_int, [_vp, _vp]),
    "rl_ctx_synchronize": (ctypes.c_int, [_vp]),
    "rl_ctx_set_arith": (ctypes.c_int, [_vp, ctypes.c_int]),
    "rl_ctx_get_arith": (ctypes.c_int, [_vp]),
    "rl_ctx_set_numpy_raise": (ctypes.c_int, [_vp, ctypes.c_int]),
    "rl_ctx_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int]),
    "rl_debug_cr_heading": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, _dp]),
    "rl_spline_eval": (ctypes.c_int, [_vp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp,
                                      ctypes.c_int, ctypes.c_int, _dp]),
    "rl_sample_along": (ctypes.c_int, [_vp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                       ctypes.c_double, _dp, ctypes.c_int, _dp]),
    "rl_fill_bounds": (ctypes.c_int, [_vp, _dp, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                      ctypes.c_double]),
    "rl_track_create": (ctypes.c_int, [_vp, _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int,
                                       ctypes.POINTER(_vp)]),
    "rl_track_destroy": (None, [_vp]),
    "rl_track_set_rings": (ctypes.c_int, [_vp, _dp, ctypes.c_int, _dp, ctypes.c_int]),
    "rl_track_set_control_points": (ctypes.c_int, [_vp, _dp, _dp]),
    "rl_track_set_length": (ctypes.c_int, [_vp, ctypes.c_double]),
    "rl_mincurv_cost": (ctypes.c_int, [_vp, _vp, _ip, ctypes.c_int, _dp, _dp, _dp, _ip]),
    "rl_track_constraint": (ctypes.c_int, [_vp, _vp, _dp, ctypes.c_int, _dp, _dp, _dp, _ip]),
    "rl_mincurv_sweep": (ctypes.c_int, [_vp, _vp, _ip, ctypes.c_int, _dp, _dp, _dp, _ip,
                                        ctypes.POINTER(Stats)]),
    "rl_mincurv_sweep_joint": (ctypes.c_int, [_vp, _vp, _ip, ctypes.c_int, _dp, _dp, _dp, _ip,
                                              ctypes.POINTER(Stats)]),
    "rl_mincurv_solve_batch_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _vp, ctypes.c_int, _ip,
                                                  ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp,
                                                  ctypes.POINTER(Stats)]),
    "rl_mincurv_solve_batch_host": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _dp, ctypes.c_int, _ip,
                                                   ctypes.c_int, ctypes.c_int, _dp, _dp, _ip, _ip,
                                                   ctypes.POINTER(Stats)]),
    "rl_dt_eval_nodes": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_double,
                                        ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp]),
    "rl_dt_eval_jac": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_double,
                                      ctypes.c_double, _dp, _dp, _dp, _dp, _dp, _dp]),
    "rl_mintime_solve_batch": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.c_int,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, _dp,
                                              _dp, ctypes.c_int, ctypes.c_double, _dp]),
    "rl_mintime_solve_batch_dev": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int,
                                                  ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp, _vp,
                                                  _vp, ctypes.c_int, ctypes.c_double, _vp]),
    "rl_qss_sim_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp,
                                      ctypes.c_int, _dp, _vp]),
    "rl_mincurv_global_batch_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                   _vp, _vp, _vp, _vp, ctypes.POINTER(Stats)]),
    "rl_mincurv_global_batch_host": (ctypes.c_int, [_vp, _vp, _dp, ctypes.c_int, ctypes.c_double,
                                                    ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.POINTER(Stats)]),
    "rl_qss_sim": (ctypes.c_int, [_vp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_int, _dp, _dp,
                                  ctypes.c_int, _dp, _ip]),
    "rl_mincurv_global_xy_batch_dev": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                                      _vp, _vp, _vp, _vp, ctypes.POINTER(Stats)]),
    "rl_mincurv_global_xy_batch_host": (ctypes.c_int, [_vp, _vp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                       ctypes.c_int, _dp, _dp, _dp, _dp, ctypes.POINTER(Stats)]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_lock = threading.Lock()


def load():
    """Load librl_mincurv.so (no GPU needed just to load and resolve symbols)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RlError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        if os.environ.get("RL_NO_TORCH", "0") != "1":
            try:  # share ONE HIP runtime with torch when both live in the process
                import torch  # noqa: F401
            except Exception:
                pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc):
    if rc != 0:
        raise RlError(f"librl_mincurv error {rc}: {load().rl_last_error().decode()}")


def as_d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def as_i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


_default_device = 0


def set_default_device(device):
    """Device used by the mirror classes / ops when none is given (one process per GPU: LOCAL_RANK)."""
    global _default_device
    _default_device = int(device)


class Context:
    """rl_ctx wrapper: one per (process, device).  Raises if no HIP device is usable."""

    _cache = {}

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        check(self.lib.rl_ctx_create(int(device), ctypes.byref(h)))
        self.h = h
        self.device = device

    @classmethod
    def get(cls, device=None):
        if device is None:
            device = _default_device
        ctx = cls._cache.get(device)
        if ctx is None:
            ctx = cls(device)
            cls._cache[device] = ctx
        return ctx

    def set_stream(self, stream_handle):
        check(self.lib.rl_ctx_set_stream(self.h, _vp(stream_handle or 0)))

    def synchronize(self):
        check(self.lib.rl_ctx_synchronize(self.h))

    def set_arith(self, arith):
        """ARITH_DEFAULT (nothing chosen: the reference-order arithmetic wherever it exists, the fast one elsewhere), ARITH_FAST,
        ARITH_REFERENCE (the reference's operations in the reference's order) or ARITH_BRANCH (include/rl_mincurv.h).
        Returns the previous setting (ARITH_DEFAULT while nothing was chosen)."""
        old = self.lib.rl_ctx_get_arith(self.h)
        check(self.lib.rl_ctx_set_arith(self.h, int(arith)))
        return old

    def set_option(self, name, value):
        """Test hooks (include/rl_mincurv.h: rl_ctx_set_option): "qss_kernel", "qss_df_waves", "qss_df_bail_at"."""
        check(self.lib.rl_ctx_set_option(self.h, name.encode(), int(value)))

    def set_numpy_raise(self, on):
        """Reference-order sweep: np.seterr(all='raise') is already in effect when the driver starts
        (include/rl_mincurv.h: rl_ctx_set_numpy_raise)."""
        check(self.lib.rl_ctx_set_numpy_raise(self.h, 1 if on else 0))

    def arith(self, arith):
        """with ctx.arith(ARITH_REFERENCE): ...  -- the sweep calls inside run in that arithmetic."""
        ctx = self

        class _Scope:
            def __enter__(self_):
                self_.old = ctx.set_arith(arith)
                return ctx

            def __exit__(self_, *exc):
                ctx.set_arith(self_.old)
                return False
        return _Scope()


class Track:
    """rl_track wrapper: device tables for one (knots, degree, sample count, control points)."""

    def __init__(self, ctx, t, cx, cy, k, N):
        self.ctx = ctx
        self.t, tp = as_d(t)
        cx, xp = as_d(cx)
        cy, yp = as_d(cy)
        self.k, self.N = int(k), int(N)
        self.n = len(self.t) - self.k - 1
        assert len(cx) == self.n and len(cy) == self.n
        h = _vp()
        check(ctx.lib.rl_track_create(ctx.h, tp, len(self.t), xp, yp, self.k, self.N, ctypes.byref(h)))
        self.h = h

    def set_rings(self, ringL, ringR):
        ringL, lp = as_d(ringL)
        ringR, rp = as_d(ringR)
        check(self.ctx.lib.rl_track_set_rings(self.h, lp, len(ringL), rp, len(ringR)))

    def set_length(self, length):
        check(self.ctx.lib.rl_track_set_length(self.h, float(length)))

    def set_control_points(self, cx, cy):
        cx, xp = as_d(cx)
        cy, yp = as_d(cy)
        check(self.ctx.lib.rl_track_set_control_points(self.h, xp, yp))

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.ctx.lib.rl_track_destroy(self.h)
                self.h = None
        except Exception:
            pass
