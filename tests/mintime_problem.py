"""Test helper: the min-time problem data of the reference's example (MGKT kart track,
min_time_optm/example/) built on the CPU with scipy + the C oracle -- centre-line samples, boundary
distances, curvature at the nodes, the QSS warm start (entrypoints/traj_opt_double_track.py:24-49)."""
import os
import warnings

import numpy as np
from scipy import integrate, interpolate
from scipy.interpolate import CubicSpline, splprep

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MGKT = os.path.join(ROOT, "spline_trajectory_optimization_amd", "examples", "race_track", "mgkt")


def _load(name):
    return np.loadtxt(os.path.join(MGKT, name), delimiter=",", skiprows=1, usecols=(0, 1))


def _fit(pts, s, k):
    loop = np.vstack([pts, pts[:1]])
    (t, c, kk), _ = splprep([loop[:, 0], loop[:, 1]], s=s, per=True, k=k)
    speed = lambda u: np.hypot(interpolate.splev(u, (t, c[0], kk), der=1), interpolate.splev(u, (t, c[1], kk), der=1))  # noqa: E731
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        L, _ = integrate.quad(speed, 0, 1, limit=1000)
    return t, c[0], c[1], kk, L


def curvature_at_nodes(s0, L, x, y):
    """Signed curvature of the closed line through (x, y) at abscissas s0: periodic cubic splines of x(s), y(s)
    (the build's own stand-in for the CasADi interpolants of race_track.py:58-85)."""
    ss = np.r_[s0, L]
    cx = CubicSpline(ss, np.r_[x, x[0]], bc_type="periodic"); cy = CubicSpline(ss, np.r_[y, y[0]], bc_type="periodic")
    dx, dy, d2x, d2y = cx(s0, 1), cy(s0, 1), cx(s0, 2), cy(s0, 2)
    return (dx * d2y - dy * d2x) / np.sqrt((dx ** 2 + dy ** 2) ** 3)


def mgkt_problem(interval, estimates):
    """Returns dict(s, kappa, left, right, L, speed, seg_time, qss_lap): nodes every ~`interval` metres."""
    centre, left, right = _load("MGKT_CENTER_enu.csv"), _load("MGKT_OUT_BOUND_enu.csv"), _load("MGKT_IN_BOUND_enu.csv")
    tabs = {}
    for name, pts in (("c", centre), ("l", left), ("r", right)):      # RaceTrack(..., s=1.0, interval): k = 3 splines
        t, cx, cy, k, L = _fit(pts, 1.0, 3)
        n = int(L // interval)
        tabs[name] = (orc.sample_along(t, cx, cy, k, L, np.linspace(0, 1, n, endpoint=False)), L)
    pts, L = tabs["c"]
    orc.fill_bounds(pts, np.ascontiguousarray(tabs["l"][0][:, :2]), np.ascontiguousarray(tabs["r"][0][:, :2]), 100.0)
    s0 = pts[:, 6].copy()
    dl = np.hypot(pts[:, 0] - pts[:, 9], pts[:, 1] - pts[:, 10]); dr = -np.hypot(pts[:, 0] - pts[:, 11], pts[:, 1] - pts[:, 12])
    kappa = curvature_at_nodes(s0, L, pts[:, 0], pts[:, 1])
    acc = CubicSpline(*np.array(estimates["acc_speed_loopup"]).T); dcc = CubicSpline(*np.array(estimates["dcc_speed_lookup"]).T)
    params = np.array([estimates["max_lon_acc_mpss"], estimates["max_lon_dcc_mpss"], estimates["max_left_acc_mpss"],
                       estimates["max_right_acc_mpss"], estimates["max_speed_mps"], estimates["max_jerk_mpsc"]])
    sim, it = orc.qss_sim(pts, acc.x, acc.c, dcc.x, dcc.c, params)
    # TIME is stored on the END point of each segment (Trajectory.fill_time): the interval that STARTS at node j
    seg_time = np.roll(sim[:, 16], -1)
    return {"s": s0, "kappa": kappa, "left": dl, "right": dr, "L": L, "speed": sim[:, 4].copy(), "seg_time": seg_time,
            "qss_lap": float(sim[:, 16].sum()), "points": pts}
