"""Host-side mirror of the reference's `spline_traj_optm.models.trajectory`
(models/trajectory.py:11-358): same class names, attributes, argument order and return types, own
implementation.

What runs where
  * the spline FIT (scipy FITPACK `splprep`, reference :219-222) and the total length (scipy `quad`,
    :223) are host-side set-up, as in the reference;
  * evaluation, sampling and boundary filling (`eval`, `sample_along`, `fill_bounds`) are HIP kernels
    behind the C ABI (include/rl_mincurv.h) -- there is no CPU fallback for them.
"""
import copy
import pickle
from dataclasses import dataclass

import numpy as np
from scipy import integrate, interpolate

from .. import ops

# Column layout of the waypoint table (reference :26-44).  CURVATURE holds the turn RADIUS
# 1/|kappa| (reference :281); TIME holds per-segment times (reference :172-180).
_COLUMNS = ("X", "Y", "Z", "YAW", "SPEED", "CURVATURE", "DIST_TO_SF_BWD", "DIST_TO_SF_FWD", "REGION",
            "LEFT_BOUND_X", "LEFT_BOUND_Y", "RIGHT_BOUND_X", "RIGHT_BOUND_Y", "BANK", "LON_ACC",
            "LAT_ACC", "TIME", "IDX", "ITERATION_FLAG")
_NCOL = len(_COLUMNS)
# the 17 columns a TTL row carries, in file order (reference :326-346)
_TTL_COLUMNS = _COLUMNS[:17]


@dataclass
class Region:
    name: str
    code: int
    vertices: np.ndarray  # [n,2]


@dataclass
class Bound:
    name: str
    type: str
    vertices: np.ndarray


def _ring_coords(poly):
    """[n,2] vertex array of a shapely LinearRing or of our own Ring (race_track.py)."""
    if hasattr(poly, "coords"):
        return np.asarray(poly.coords, dtype=np.float64)[:, :2]
    return np.asarray(poly, dtype=np.float64)[:, :2]


class Trajectory:
    """N x 19 float64 waypoint table; the class attributes X, Y, ... are the column indices."""

    def __init__(self, num_point: int, ttl_num: int = 0, origin=None) -> None:
        self.ttl_num = ttl_num
        self.origin = origin
        table = np.zeros((num_point, _NCOL), dtype=np.float64)
        table[:, _COLUMNS.index("IDX")] = np.arange(num_point)
        table[:, _COLUMNS.index("ITERATION_FLAG")] = -1.0
        self.points = table

    # -- container protocol: everything is delegated to the table
    def __getitem__(self, key):
        return self.points[key]

    def __setitem__(self, key, val):
        self.points[key] = val

    def __len__(self):
        return self.points.shape[0]

    def __iter__(self):
        return iter(self.points)

    def copy(self):
        # like the reference (:66-69), ttl_num / origin are not carried over
        twin = Trajectory(len(self))
        twin.points = np.array(self.points, copy=True)
        return twin

    def inc(self, idx: int):
        return (idx + 1) % len(self)

    def dec(self, idx: int):
        return (idx - 1) % len(self)

    def ts(self):
        """Spline parameter of every waypoint: uniform in u, NOT in arc length (reference :199-200)."""
        return np.linspace(0.0, 1.0, len(self), endpoint=False)

    def distance(self, pt1, pt2):
        return float(np.hypot(pt1[Trajectory.X] - pt2[Trajectory.X], pt1[Trajectory.Y] - pt2[Trajectory.Y]))

    # -- derived columns
    def fill_bounds(self, left_poly, right_poly, max_dist=100.0):
        """Closest crossing of each waypoint's normal with the two boundary rings (reference :83-141)
        -> HIP kernel k_fill_bounds through rl_fill_bounds."""
        table = np.ascontiguousarray(self.points, dtype=np.float64)
        ops.fill_bounds(table, _ring_coords(left_poly), _ring_coords(right_poly), max_dist)
        cols = slice(Trajectory.LEFT_BOUND_X, Trajectory.RIGHT_BOUND_Y + 1)
        self.points[:, cols] = table[:, cols]

    def fill_distance(self):
        """Chord-length distance from / to the start-finish line (reference :143-156)."""
        xy = self.points[:, :2]
        chord = np.linalg.norm(np.roll(xy, -1, axis=0) - xy, axis=1)   # waypoint i -> i+1 (cyclic)
        behind = np.concatenate(([0.0], np.cumsum(chord[:-1])))
        self.points[:, Trajectory.DIST_TO_SF_BWD] = behind
        self.points[:, Trajectory.DIST_TO_SF_FWD] = chord.sum() - behind

    def fill_time(self):
        """Time of every segment at the mean of its end speeds, stored on the segment's END point
        (reference :158-180: the running sum is commented out there, so TIME is per segment)."""
        v = self.points[:, Trajectory.SPEED]
        if np.any((v == 0.0) & (self.points[:, Trajectory.LON_ACC] == 0.0)):
            raise Exception("Zero speed and lon_acc encoutered. Cannot fill time.")
        xy = self.points[:, :2]
        nxt = np.roll(np.arange(len(self)), -1)
        seg = np.hypot(*(xy[nxt] - xy).T)
        self.points[0, Trajectory.TIME] = 0.0
        self.points[nxt, Trajectory.TIME] = seg / (0.5 * (v + v[nxt]))

    def fill_region(self, regions: list):
        """Tag waypoints with the code of the first region polygon containing them (reference
        :182-194; file tooling, needs shapely)."""
        from shapely.geometry import Point, Polygon
        shapes = [(Polygon(np.asarray(r.vertices).tolist()), r.code) for r in regions]
        for row in self.points:
            here = Point(row[Trajectory.X], row[Trajectory.Y])
            code = next((c for poly, c in shapes if poly.contains(here)), None)
            if code is not None:
                row[Trajectory.REGION] = code

    # -- plain CSV of the whole table (reference :202-209); called on the class, like the reference
    def save(f, traj):
        np.savetxt(f, traj.points, delimiter=',')

    def load(f):
        table = np.loadtxt(f, np.float64, delimiter=',')
        traj = Trajectory(len(table))
        traj.points = table
        return traj


for _i, _name in enumerate(_COLUMNS):   # Trajectory.X = 0, Trajectory.Y = 1, ...
    setattr(Trajectory, _name, _i)


class BSplineTrajectory:
    """Closed parametric B-spline (x(u), y(u)), u in [0,1), fitted with FITPACK.  `_spl_x`/`_spl_y`
    are scipy BSpline objects sharing one knot vector, as in the reference: the optimiser reaches into
    their `.t`, `.c` (mutable) and `.k`."""

    def __init__(self, coordinates: np.ndarray, s: float, k: int):
        pts = np.asarray(coordinates)
        assert pts.ndim == 2 and pts.shape[1] == 2 and pts.shape[0] >= 3, "coordinates should be N * 2"
        loop = np.vstack([pts, pts[:1]])                       # repeat the first point: closed loop
        (knots, coefs, degree), _ = interpolate.splprep([loop[:, 0], loop[:, 1]], s=s, per=True, k=k)
        self._spl_x = interpolate.BSpline(knots, coefs[0], degree)
        self._spl_y = interpolate.BSpline(knots, coefs[1], degree)
        self._length = self.eval_sectional_length((0.0, 1.0))

    # -- host set-up: arc length by adaptive quadrature of |r'(u)| (reference :225-233)
    def _speed(self, u):
        return np.hypot(interpolate.splev(u, self._spl_x, der=1), interpolate.splev(u, self._spl_y, der=1))

    def eval_sectional_length(self, ts):
        length, _ = integrate.quad(self._speed, ts[0], ts[1], limit=1000)
        return length

    def _second_over_speed(self, spl, lo, hi):
        def f(u):
            return interpolate.splev(u, spl, der=2) / self._speed(u)
        value, _ = integrate.quad(f, lo, hi, limit=200)
        return value

    def eval_dx_sectional_length(self, ts):
        """Integral of x''(u) / |r'(u)| over [ts[0], ts[1]] (reference :235-239; host quadrature like the reference's,
        only reached from its commented-out min-time cost)."""
        return self._second_over_speed(self._spl_x, ts[0], ts[1])

    def eval_dy_sectional_length(self, ts):
        """Integral of y''(u) / |r'(u)| over [ts[0], ts[1]] (reference :241-245)."""
        return self._second_over_speed(self._spl_y, ts[0], ts[1])

    def get_length(self):
        return self._length

    def _tck(self):
        as_c = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        return as_c(self._spl_x.t), as_c(self._spl_x.c), as_c(self._spl_y.c), int(self._spl_x.k)

    # -- HIP-backed evaluation
    def eval(self, t, der=0):
        """(x, y) or their der-th derivatives at t (reference :247-248) -> rl_spline_eval."""
        if der > 2:
            raise ValueError("der > 2 is not on the min-curvature path")
        u = np.atleast_1d(np.asarray(t, dtype=np.float64))
        knots, cx, cy, k = self._tck()
        rows = ops.spline_eval(knots, cx, cy, k, u, der_max=der)
        x, y = rows[2 * der], rows[2 * der + 1]
        return (np.float64(x[0]), np.float64(y[0])) if np.ndim(t) == 0 else (x, y)

    def eval_yaw(self, t):
        dx, dy = self.eval(t, 1)
        return np.arctan2(dy, dx)

    def sample_along(self, interval: float = None, ts=None) -> "Trajectory":
        """Fresh Trajectory with X, Y, YAW, turn radius and cumulative arc length at the given
        parameters, or every ~`interval` metres (reference :268-291) -> rl_sample_along."""
        if interval is not None:
            ts = np.linspace(0.0, 1.0, int(self._length // interval), endpoint=False)
        u = np.ascontiguousarray(ts, dtype=np.float64)
        knots, cx, cy, k = self._tck()
        traj = Trajectory(len(u))
        traj.points = ops.sample_along(knots, cx, cy, k, self._length, u)
        return traj

    def copy(self):
        return copy.deepcopy(self)

    def set_control_point(self, idx, coord):
        self._spl_x.c[idx], self._spl_y.c[idx] = coord[0], coord[1]

    def get_control_point(self, idx):
        return self._spl_x.c[idx], self._spl_y.c[idx]

    # pickle round trip (reference :303-309); called on the class
    def save(f, traj):
        with open(f, "wb") as fh:
            pickle.dump(traj, fh)

    def load(f):
        with open(f, "rb") as fh:
            return pickle.load(fh)


def save_ttl(ttl_path: str, trajectory: Trajectory):
    """TTL file (reference :312-347): header `ttl_num,N,track_length[,origin...]`, then 17 columns per
    waypoint with REGION written as an integer."""
    head = [trajectory.ttl_num, len(trajectory), trajectory[0, Trajectory.DIST_TO_SF_FWD]]
    if trajectory.origin is not None:
        head += list(trajectory.origin)
    region = _TTL_COLUMNS.index("REGION")
    with open(ttl_path, "w") as out:
        out.write(",".join(str(h) for h in head) + "\n")
        for row in trajectory.points:
            cells = [str(int(v)) if c == region else str(v) for c, v in enumerate(row[:len(_TTL_COLUMNS)])]
            out.write(",".join(cells) + "\n")


def load_ttl(ttl_path: str) -> Trajectory:
    """Inverse of save_ttl (reference :350-358); the header must carry the 3-value origin."""
    with open(ttl_path, "r") as src:
        head = src.readline().strip().split(",")
    assert len(head) >= 6
    body = np.atleast_2d(np.loadtxt(ttl_path, dtype=float, delimiter=",", skiprows=1))
    traj = Trajectory(len(body), int(head[0]), tuple(float(h) for h in head[3:6]))
    traj.points[:, :body.shape[1]] = body
    return traj
