"""ctypes binding of oracle/libmincurv_oracle.so  --  TEST INFRASTRUCTURE ONLY.

Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module.  The product package must never do so.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SO: load another build of the same source (the -fsanitize build of tools/run_oracle_sanitized.sh)
_SO = os.environ.get("ORACLE_SO") or os.path.join(_HERE, "libmincurv_oracle.so")

NCOL = 19
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, "mincurv_oracle.c")
    so = os.path.join(_HERE, "libmincurv_oracle.so")
    others = [os.path.join(_HERE, "libmincurv_oracle_fma.so"), os.path.join(_HERE, "libmincurv_oracle_cr.so")]
    if force or any(not os.path.exists(f) or os.path.getmtime(f) < os.path.getmtime(src) for f in [so] + others):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_lib = None


class _Variant:
    """Context manager: route the wrappers of this module to another build of the same oracle source:
    the FMA-contracted one (noise-sensitivity diagnostics in the tests) or the one with the correctly
    rounded atan2 / cos / sin (oracle/Makefile)."""

    def __init__(self, so_name="libmincurv_oracle_fma.so"):
        self.so_name = so_name

    def __enter__(self):
        global _lib
        build()
        self.saved = _lib
        _lib = None
        self.saved_so = globals()["_SO"]
        globals()["_SO"] = os.path.join(_HERE, self.so_name)
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        globals()["_SO"] = self.saved_so
        return False


def fma_variant():
    return _Variant()


def cr_variant():
    """The strict build whose heading / normal functions are the correctly rounded ones: what the HIP
    kernel's reference-order mode (RL_ARITH_REFERENCE) must reproduce bit for bit."""
    return _Variant("libmincurv_oracle_cr.so")


def lib():
    global _lib
    if _lib is None:
        build()  # no-op unless a .so is missing or older than the source
        _lib = ctypes.CDLL(_SO)
        _lib.orc_arc_gk21.restype = ctypes.c_double
        _lib.orc_arc_gk21.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int,
                                      ctypes.c_double, ctypes.c_double]
        _lib.orc_sample_along.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double,
                                          _dp, ctypes.c_int, _dp]
        _lib.orc_run_min_curvature_qp.argtypes = [
            _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double, _dp, ctypes.c_int,
            _dp, ctypes.c_int, _dp, ctypes.c_int, _ip, ctypes.c_int, _ip]
        _lib.orc_solve_width_batch.argtypes = [
            _dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp,
            ctypes.c_int, _ip, ctypes.c_int, _dp, _dp, _ip, ctypes.c_int, _dp]
        _lib.orc_last_kappa.restype = ctypes.c_double
        _lib.orc_fill_bounds.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                         ctypes.c_double]
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def heading(dx, dy):
    """[n,5] = yaw, cos / sin(yaw + pi/2), cos / sin(yaw - pi/2) as sample_along / fill_bounds take them
    (inside cr_variant(): the correctly rounded values)."""
    dx, xp = _d(dx); dy, yp = _d(dy)
    out = np.empty((len(dx), 5))
    lib().orc_heading(xp, yp, len(dx), out.ctypes.data_as(_dp))
    return out


def bspline_eval(t, c, k, x, der=0):
    t, tp = _d(t); c, cp = _d(c); x, xp = _d(np.atleast_1d(x))
    out = np.empty_like(x)
    lib().orc_bspline_eval(tp, len(t), cp, int(k), xp, len(x), int(der), out.ctypes.data_as(_dp))
    return out


def bspline_derivative(t, c, k, nu=1):
    t, tp = _d(t); c, cp = _d(c)
    to = np.empty(len(t) - 2 * nu); co = np.empty(len(t) - 2 * nu)
    k2 = lib().orc_bspline_derivative(tp, len(t), cp, len(c), int(k), int(nu),
                                      to.ctypes.data_as(_dp), co.ctypes.data_as(_dp))
    return to, co, k2


def basis_element(tk, x, der=0):
    tk, tp = _d(tk); x, xp = _d(np.atleast_1d(x))
    out = np.empty_like(x)
    lib().orc_basis_element(tp, len(tk) - 2, xp, len(x), int(der), out.ctypes.data_as(_dp))
    return out


def arc_gk21(t, cx, cy, k, a, b):
    t, tp = _d(t); cx, xp = _d(cx); cy, yp = _d(cy)
    return lib().orc_arc_gk21(tp, len(t), xp, yp, int(k), float(a), float(b))


def sample_along(t, cx, cy, k, length, u):
    t, tp = _d(t); cx, xp = _d(cx); cy, yp = _d(cy); u, up = _d(u)
    pts = np.empty((len(u), NCOL))
    lib().orc_sample_along(tp, len(t), xp, yp, int(k), float(length), up, len(u),
                           pts.ctypes.data_as(_dp))
    return pts


def fill_bounds(points, ringL, ringR, max_dist=100.0):
    """In place on a C-contiguous float64 [N,19] array."""
    assert points.dtype == np.float64 and points.flags.c_contiguous and points.shape[1] == NCOL
    ringL, lp = _d(ringL); ringR, rp = _d(ringR)
    lib().orc_fill_bounds(points.ctypes.data_as(_dp), len(points), lp, len(ringL), rp, len(ringR),
                          float(max_dist))
    return points


def min_curvature_cost(z, idx, t, cx, cy, k, N):
    t, tp = _d(t); cx, xp = _d(cx); cy, yp = _d(cy); z, zp = _d(z)
    H = np.zeros(4); g = np.zeros(2)
    M = lib().orc_min_curvature_cost(zp, int(idx), tp, len(t), xp, yp, int(k), int(N),
                                     H.ctypes.data_as(_dp), g.ctypes.data_as(_dp))
    return H.reshape(2, 2), g, M


def track_constraint(idx, t, cx, cy, k, points):
    t, tp = _d(t); cx, xp = _d(cx); cy, yp = _d(cy); points, pp = _d(points)
    N = len(points)
    A = np.zeros((2 * N, 2)); lba = np.zeros(2 * N); uba = np.zeros(2 * N)
    M = lib().orc_track_constraint(int(idx), tp, len(t), xp, yp, int(k), pp, N,
                                   A.ctypes.data_as(_dp), lba.ctypes.data_as(_dp),
                                   uba.ctypes.data_as(_dp))
    return A[:2 * M].copy(), lba[:2 * M].copy(), uba[:2 * M].copy()


def qp_solve_separable(H, g, A, lba, uba):
    H, hp = _d(np.asarray(H).reshape(4)); g, gp = _d(g); A, ap = _d(A)
    lba, lp = _d(lba); uba, up = _d(uba)
    x = np.zeros(2)
    st = lib().orc_qp_solve_separable(hp, gp, ap, lp, up, len(lba) // 2, x.ctypes.data_as(_dp))
    return st, x


def run_min_curvature_qp(t, cx, cy, k, length, N, ringL, ringR, i_start, max_iter=None, rerounding=0, numpy_raise=False):
    """Returns (cx, cy, points[N,19], n_success[max_iter,2]).  rerounding: seed of mincurv_oracle.c:
    orc_set_rerounding for this run (0 = the unperturbed oracle).  numpy_raise: np.seterr(all='raise') is in effect
    from the start (mincurv_oracle.c, "numpy's error state"); last_raised() then tells how many steps raised."""
    t, tp = _d(t)
    cx = np.array(cx, dtype=np.float64, copy=True); cy = np.array(cy, dtype=np.float64, copy=True)
    ringL, lp = _d(ringL); ringR, rp = _d(ringR)
    i_start, ip = _i(i_start)
    max_iter = len(i_start) if max_iter is None else max_iter
    pts = np.zeros((N, NCOL))
    lib().orc_trajectory_init(pts.ctypes.data_as(_dp), N)
    ns = np.zeros(2 * max_iter, dtype=np.int32)
    lib().orc_reset_kappa()
    setr = lib().orc_set_rerounding
    setr.argtypes = [ctypes.c_ulonglong]
    setr.restype = None
    setr(int(rerounding))                      # thread-local in the oracle
    lib().orc_set_numpy_raise(1 if numpy_raise else 0)
    try:
        lib().orc_run_min_curvature_qp(tp, len(t), cx.ctypes.data_as(_dp), cy.ctypes.data_as(_dp), int(k),
                                       float(length), pts.ctypes.data_as(_dp), int(N), lp, len(ringL),
                                       rp, len(ringR), ip, int(max_iter), ns.ctypes.data_as(_ip))
    finally:
        setr(0)
        lib().orc_set_numpy_raise(0)
    return cx, cy, pts, ns.reshape(max_iter, 2)


def last_raised():
    """Steps / windows of the last driver call on this thread whose re-sampling raised under numpy's raise mode."""
    return int(lib().orc_last_raised())


def last_kappa():
    """Conditioning indicator of the last run_min_curvature_qp on this thread (see the header)."""
    return lib().orc_last_kappa()


def solve_width_batch(t, cx0, cy0, k, length, N, widths, i_start, max_iter=None, nthreads=1,
                      want_kappa=False, seeds=None):
    """widths [B,N,2] -> (ctrl [B,n,2], xy [B,N,2], n_success [B,max_iter,2][, kappa [B]]).
    seeds [B] (optional): re-rounding seed per instance (mincurv_oracle.c: orc_set_rerounding), 0 = none."""
    t, tp = _d(t); cx0, xp = _d(cx0); cy0, yp = _d(cy0); widths, wp = _d(widths)
    i_start, ip = _i(i_start)
    max_iter = len(i_start) if max_iter is None else max_iter
    B = widths.shape[0]
    n = len(t) - k - 1
    ctrl = np.zeros((B, n, 2)); xy = np.zeros((B, N, 2))
    ns = np.zeros((B, max_iter, 2), dtype=np.int32)
    kappa = np.zeros(B)
    f = lib().orc_solve_width_batch_seeded
    f.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp,
                  ctypes.c_int, _ip, ctypes.c_int, _dp, _dp, _ip, ctypes.c_int, _dp, ctypes.c_void_p]
    f.restype = None
    sd = None
    if seeds is not None:
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert sd.shape == (B,)
    f(tp, len(t), xp, yp, int(k), float(length), int(N), wp, B, ip, int(max_iter), ctrl.ctypes.data_as(_dp),
      xy.ctypes.data_as(_dp), ns.ctypes.data_as(_ip), int(nthreads), kappa.ctypes.data_as(_dp),
      sd.ctypes.data if sd is not None else None)
    if want_kappa:
        return ctrl, xy, ns, kappa
    return ctrl, xy, ns


def qss_sim(points, acc_x, acc_c, dcc_x, dcc_c, params):
    """Simulator.run_simulation on a copy of points [N,19]; returns (points, n_iterations)."""
    pts = np.array(points, dtype=np.float64, copy=True, order="C")
    ax, axp = _d(acc_x); ac, acp = _d(acc_c); dx, dxp = _d(dcc_x); dc, dcp = _d(dcc_c); pr, prp = _d(params)
    it = lib().orc_qss_sim(pts.ctypes.data_as(_dp), len(pts), axp, acp, ac.shape[1], dxp, dcp, dc.shape[1], prp)
    return pts, it


def qp_diag_rows(h, g, A, l, u):
    """Goldfarb-Idnani solve of the diagonal-Hessian QP; returns (status, x, lam)."""
    h, hp = _d(h); g, gp = _d(g); A, ap = _d(A); l, lp = _d(l); u, up = _d(u)
    M, nv = A.shape
    x = np.zeros(nv); lam = np.zeros(M)
    st = lib().orc_qp_diag_rows(int(nv), hp, gp, int(M), ap, lp, up, x.ctypes.data_as(_dp), lam.ctypes.data_as(_dp))
    return st, x, lam


def run_joint_min_curvature_qp(t, cx, cy, k, length, N, ringL, ringR, i_start, max_iter=None, rerounding=0,
                               numpy_raise=False):
    """Returns (cx, cy, points[N,19], n_success[max_iter]).  rerounding: seed of mincurv_oracle.c:
    orc_set_rerounding for this run (0 = the unperturbed oracle).  numpy_raise: as for run_min_curvature_qp."""
    t, tp = _d(t)
    cx = np.array(cx, dtype=np.float64, copy=True); cy = np.array(cy, dtype=np.float64, copy=True)
    ringL, lp = _d(ringL); ringR, rp = _d(ringR)
    i_start, ip = _i(i_start)
    max_iter = len(i_start) if max_iter is None else max_iter
    pts = np.zeros((N, NCOL))
    lib().orc_trajectory_init(pts.ctypes.data_as(_dp), N)
    ns = np.zeros(max_iter, dtype=np.int32)
    f = lib().orc_run_joint_min_curvature_qp
    f.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_double, _dp, ctypes.c_int,
                  _dp, ctypes.c_int, _dp, ctypes.c_int, _ip, ctypes.c_int, _ip]
    setr = lib().orc_set_rerounding
    setr.argtypes = [ctypes.c_ulonglong]
    setr.restype = None
    setr(int(rerounding))
    lib().orc_set_numpy_raise(1 if numpy_raise else 0)
    try:
        f(tp, len(t), cx.ctypes.data_as(_dp), cy.ctypes.data_as(_dp), int(k), float(length),
          pts.ctypes.data_as(_dp), int(N), lp, len(ringL), rp, len(ringR), ip, int(max_iter), ns.ctypes.data_as(_ip))
    finally:
        setr(0)
        lib().orc_set_numpy_raise(0)
    return cx, cy, pts, ns


def global_mincurv(t, cx0, cy0, k, N, w_left, w_right, margin=0.0, n_outer=6):
    """The build's own global min-curvature QP (lateral control-point offsets, interior point);
    returns (cx, cy, xy [N,2], a [n-k], stats[6]: ipm iterations, sum kappa^2 before/after,
    bound violation, last step, active rows)."""
    t, tp = _d(t); cx0, xp = _d(cx0); cy0, yp = _d(cy0); wl, lp = _d(w_left); wr, rp = _d(w_right)
    n = len(cx0)
    cx = np.zeros(n); cy = np.zeros(n); xy = np.zeros((N, 2)); a = np.zeros(n - k); st = np.zeros(6)
    f = lib().orc_global_mincurv
    f.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double,
                  ctypes.c_int, _dp, _dp, _dp, _dp, _dp]
    rc = f(tp, len(t), xp, yp, int(k), int(N), lp, rp, float(margin), int(n_outer), cx.ctypes.data_as(_dp),
           cy.ctypes.data_as(_dp), xy.ctypes.data_as(_dp), a.ctypes.data_as(_dp), st.ctypes.data_as(_dp))
    assert rc == 0
    return cx, cy, xy, a, st


def global_mincurv_xy(t, cx0, cy0, k, N, w_left, w_right, margin=0.0, lon=1.0, n_outer=6):
    """The global QP with both coordinates of every control point free (lateral and +-`lon` m longitudinal rows per sample,
    julia/spline_traj_opt.ipynb L247-281 with SURVEY App. A.6's corrections); returns (cx, cy, xy [N,2], z [n-k, 2],
    stats[8]: ipm iterations, sum kappa^2 before / after, bound violation, last step, active lateral / longitudinal rows,
    halved steps)."""
    t, tp = _d(t); cx0, xp = _d(cx0); cy0, yp = _d(cy0); wl, lp = _d(w_left); wr, rp = _d(w_right)
    n = len(cx0)
    cx = np.zeros(n); cy = np.zeros(n); xy = np.zeros((N, 2)); z = np.zeros((n - k, 2)); st = np.zeros(8)
    f = lib().orc_global_mincurv_xy
    f.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double,
                  ctypes.c_int, _dp, _dp, _dp, _dp, _dp]
    rc = f(tp, len(t), xp, yp, int(k), int(N), lp, rp, float(margin), float(lon), int(n_outer), cx.ctypes.data_as(_dp),
           cy.ctypes.data_as(_dp), xy.ctypes.data_as(_dp), z.ctypes.data_as(_dp), st.ctypes.data_as(_dp))
    assert rc == 0
    return cx, cy, xy, z, st


REPLAY_STRIDE = 20


def width_rings(t, cx0, cy0, k, N, widths):
    """Rings of one width-form instance, built exactly as solve_width_batch builds them."""
    t, tp = _d(t); cx0, xp = _d(cx0); cy0, yp = _d(cy0); widths, wp = _d(widths)
    assert widths.shape == (N, 2)
    rl = np.zeros((N, 2)); rr = np.zeros((N, 2))
    f = lib().orc_width_rings
    f.argtypes = [_dp, ctypes.c_int, _dp, _dp, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
    f.restype = None
    f(tp, len(t), xp, yp, int(k), int(N), wp, rl.ctypes.data_as(_dp), rr.ctypes.data_as(_dp))
    return rl, rr


def replay_steps(t, k, N, ringL, ringR, idx, cxs, cys, nthreads=8):
    """Teacher-forced re-derivation of recorded sweep steps (mincurv_oracle.c: orc_replay_steps).
    idx [S], cxs/cys [S,n] = the control points BEFORE each step.  Returns [S, REPLAY_STRIDE]."""
    t, tp = _d(t); ringL, lp = _d(ringL); ringR, rp = _d(ringR)
    idx, ip = _i(idx); cxs, xp = _d(cxs); cys, yp = _d(cys)
    S = len(idx)
    assert cxs.shape == (S, len(t) - k - 1) and cys.shape == cxs.shape
    out = np.zeros((S, REPLAY_STRIDE))
    f = lib().orc_replay_steps
    f.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                  ctypes.c_int, _ip, _dp, _dp, _dp, ctypes.c_int]
    f.restype = None
    f(tp, len(t), int(k), int(N), lp, len(ringL), rp, len(ringR), S, ip, xp, yp, out.ctypes.data_as(_dp), int(nthreads))
    return out


JREPLAY_HEAD = 48


def replay_joint_windows(t, k, N, ringL, ringR, kks, cxs, cys, nthreads=8):
    """Teacher-forced re-derivation of sliding windows (mincurv_oracle.c: orc_replay_joint_windows).
    kks [W] first control point of each window, cxs/cys [W,n] = the control points BEFORE each window.
    Returns (head [W, JREPLAY_HEAD], rows [W, N, 9])."""
    t, tp = _d(t); ringL, lp = _d(ringL); ringR, rp = _d(ringR)
    kks, ip = _i(kks); cxs, xp = _d(cxs); cys, yp = _d(cys)
    W = len(kks)
    assert cxs.shape == (W, len(t) - k - 1) and cys.shape == cxs.shape
    head = np.zeros((W, JREPLAY_HEAD)); rows = np.zeros((W, N, 9))
    f = lib().orc_replay_joint_windows
    f.argtypes = [_dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int,
                  ctypes.c_int, _ip, _dp, _dp, _dp, _dp, ctypes.c_int]
    f.restype = None
    f(tp, len(t), int(k), int(N), lp, len(ringL), rp, len(ringR), W, ip, xp, yp, head.ctypes.data_as(_dp),
      rows.ctypes.data_as(_dp), int(nthreads))
    return head, rows
