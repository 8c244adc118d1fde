"""Host-side mirror of the reference's `spline_traj_optm.models.trajectory`
(models/trajectory.py:11-358): same class names, attributes, argument order and return types.

What runs where
  * the spline FIT (scipy FITPACK splprep, trajectory.py:219-222) and the total length
    (scipy quad, :223) are host-side set-up exactly as in the reference;
  * evaluation, sampling and boundary filling (`eval`, `sample_along`, `fill_bounds`) are HIP
    kernels behind the C ABI (include/rl_mincurv.h) -- there is no CPU fallback for them.
"""
import copy
import pickle
from dataclasses import dataclass

import numpy as np
from scipy import interpolate
from scipy.integrate import quad
from scipy.interpolate import BSpline

from .. import ops


@dataclass
class Region:
    name: str
    code: int
    vertices: np.ndarray  # n * 2


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
    # column layout: models/trajectory.py:26-44
    X = 0
    Y = 1
    Z = 2
    YAW = 3
    SPEED = 4
    CURVATURE = 5
    DIST_TO_SF_BWD = 6
    DIST_TO_SF_FWD = 7
    REGION = 8
    LEFT_BOUND_X = 9
    LEFT_BOUND_Y = 10
    RIGHT_BOUND_X = 11
    RIGHT_BOUND_Y = 12
    BANK = 13
    LON_ACC = 14
    LAT_ACC = 15
    TIME = 16
    IDX = 17
    ITERATION_FLAG = 18

    def __init__(self, num_point: int, ttl_num: int = 0, origin=None) -> None:
        self.ttl_num = ttl_num
        self.origin = origin
        self.points = np.zeros((num_point, 19), dtype=np.float64)
        self.points[:, Trajectory.IDX] = np.arange(0, len(self.points), 1)
        self.points[:, Trajectory.ITERATION_FLAG] = -1

    def __getitem__(self, key):
        return self.points[key]

    def __setitem__(self, key, val):
        self.points[key] = val

    def __len__(self):
        return len(self.points)

    def __iter__(self):
        for pt in self.points:
            yield pt

    def copy(self):
        new_traj = Trajectory(len(self.points))
        new_traj.points = self.points.copy()
        return new_traj

    def inc(self, idx: int):
        return 0 if idx + 1 == len(self.points) else idx + 1

    def dec(self, idx: int):
        return len(self.points) - 1 if idx - 1 < 0 else idx - 1

    def fill_bounds(self, left_poly, right_poly, max_dist=100.0):
        """models/trajectory.py:83-141 -> HIP kernel k_fill_bounds (rl_fill_bounds)."""
        pts = np.ascontiguousarray(self.points, dtype=np.float64)
        ops.fill_bounds(pts, _ring_coords(left_poly), _ring_coords(right_poly), max_dist)
        self.points[:, Trajectory.LEFT_BOUND_X:Trajectory.RIGHT_BOUND_Y + 1] = \
            pts[:, Trajectory.LEFT_BOUND_X:Trajectory.RIGHT_BOUND_Y + 1]

    def fill_distance(self):
        # models/trajectory.py:143-156 (chord lengths; plain host bookkeeping)
        xy = self.points[:, :2]
        dists = np.linalg.norm(xy - np.roll(xy, -1, axis=0), axis=1)
        self.points[0, Trajectory.DIST_TO_SF_BWD] = 0.0
        self.points[1:, Trajectory.DIST_TO_SF_BWD] = dists[:-1]
        self.points[:, Trajectory.DIST_TO_SF_BWD] = np.cumsum(self.points[:, Trajectory.DIST_TO_SF_BWD])
        self.points[:, Trajectory.DIST_TO_SF_FWD] = np.sum(dists) - self.points[:, Trajectory.DIST_TO_SF_BWD]

    def fill_time(self):
        # models/trajectory.py:158-180: stores PER-SEGMENT times (the cumulative sum is commented
        # out in the reference, :179-180) -- reproduced, not fixed.
        for pt in self.points:
            # the reference's guard indexes with a boolean by mistake (:161); the intent is kept
            if pt[Trajectory.SPEED] == 0.0 and pt[Trajectory.LON_ACC] == 0.0:
                raise Exception("Zero speed and lon_acc encoutered. Cannot fill time.")
        self.points[0, Trajectory.TIME] = 0.0
        n = len(self.points)
        for i in range(n):
            nxt = 0 if i + 1 == n else i + 1
            x = self.distance(self.points[i], self.points[nxt])
            self.points[nxt, Trajectory.TIME] = x / (
                0.5 * (self.points[i, Trajectory.SPEED] + self.points[nxt, Trajectory.SPEED]))

    def fill_region(self, regions: list):
        # models/trajectory.py:182-194 (file/region tooling, needs shapely; out of the hot path)
        from shapely.geometry import Point, Polygon
        polygons = [(Polygon(r.vertices.tolist()), r.code) for r in regions]
        for row in self.points:
            p = Point([row[Trajectory.X], row[Trajectory.Y]])
            for polygon, code in polygons:
                if polygon.contains(p):
                    row[Trajectory.REGION] = code
                    break

    def distance(self, pt1, pt2):
        return np.linalg.norm(pt1[Trajectory.X:Trajectory.Y + 1] - pt2[Trajectory.X:Trajectory.Y + 1])

    def ts(self):
        return np.linspace(0.0, 1.0, self.__len__(), endpoint=False)

    def save(f, traj):
        np.savetxt(f, traj.points, delimiter=',')

    def load(f):
        arr = np.loadtxt(f, np.float64, delimiter=',')
        traj = Trajectory(len(arr))
        traj.points = arr
        return traj


class BSplineTrajectory:
    def __init__(self, coordinates: np.ndarray, s: float, k: int):
        assert coordinates.shape[0] >= 3 and coordinates.shape[1] == 2 and len(
            coordinates.shape) == 2, "coordinates should be N * 2"
        # close the loop (models/trajectory.py:217-218) and fit (host set-up, :219-222)
        closed = np.vstack([coordinates, coordinates[0, np.newaxis, :]])
        tck, u = interpolate.splprep([closed[:, 0], closed[:, 1]], s=s, per=True, k=k)
        self._spl_x = BSpline(tck[0], tck[1][0], tck[2])
        self._spl_y = BSpline(tck[0], tck[1][1], tck[2])
        self._length = self.__get_section_length(0.0, 1.0)

    # -- host set-up helpers (scipy, as in the reference :225-245)
    def __integrate_length(self, t: float):
        return np.sqrt(interpolate.splev(t, self._spl_x, der=1) ** 2 +
                       interpolate.splev(t, self._spl_y, der=1) ** 2)

    def __get_section_length(self, t_min: float, t_max: float):
        length, err = quad(self.__integrate_length, t_min, t_max, limit=1000)
        return length

    def eval_sectional_length(self, ts):
        return self.__get_section_length(ts[0], ts[1])

    def _tck(self):
        return (np.ascontiguousarray(self._spl_x.t, dtype=np.float64),
                np.ascontiguousarray(self._spl_x.c, dtype=np.float64),
                np.ascontiguousarray(self._spl_y.c, dtype=np.float64), int(self._spl_x.k))

    # -- HIP-backed evaluation
    def eval(self, t, der=0):
        """models/trajectory.py:247-248 -> rl_spline_eval."""
        t_arr = np.atleast_1d(np.asarray(t, dtype=np.float64))
        tt, cx, cy, k = self._tck()
        if der > 2:
            raise ValueError("der > 2 is not on the min-curvature path")
        out = ops.spline_eval(tt, cx, cy, k, t_arr, der_max=der)
        x, y = out[2 * der], out[2 * der + 1]
        if np.ndim(t) == 0:
            return np.float64(x[0]), np.float64(y[0])
        return x, y

    def eval_yaw(self, t):
        dx, dy = self.eval(t, 1)
        return np.arctan2(dy, dx)

    def get_length(self):
        return self._length

    def sample_along(self, interval: float = None, ts=None) -> Trajectory:
        """models/trajectory.py:268-291 -> rl_sample_along (k_sample_geometry + k_sample_cumsum)."""
        if interval is not None:
            total_length = self.get_length()
            num_sample = int(total_length // interval)
            ts = np.linspace(0.0, 1.0, num_sample, endpoint=False)
        ts = np.ascontiguousarray(ts, dtype=np.float64)
        traj = Trajectory(len(ts))
        tt, cx, cy, k = self._tck()
        traj.points = ops.sample_along(tt, cx, cy, k, self._length, ts)
        return traj

    def copy(self):
        return copy.deepcopy(self)

    def set_control_point(self, idx, coord):
        self._spl_x.c[idx] = coord[0]
        self._spl_y.c[idx] = coord[1]

    def get_control_point(self, idx):
        return self._spl_x.c[idx], self._spl_y.c[idx]

    def save(f, traj):
        with open(f, "wb") as output_file:
            pickle.dump(traj, output_file)

    def load(f):
        with open(f, "rb") as input_file:
            return pickle.load(input_file)


def save_ttl(ttl_path: str, trajectory: Trajectory):
    """models/trajectory.py:312-347 (TTL csv: header ttl_num,N,length[,origin] + 17 columns/row)."""
    with open(ttl_path, "w") as f:
        header = ",".join([str(trajectory.ttl_num), str(len(trajectory)),
                           str(trajectory[0, Trajectory.DIST_TO_SF_FWD])])
        if trajectory.origin is not None:
            header += "," + ",".join([str(x) for x in trajectory.origin])
        f.write(header + "\n")
        for row in trajectory.points:
            vals = [str(row[c]) for c in range(Trajectory.X, Trajectory.REGION)]
            vals.append(str(int(row[Trajectory.REGION])))
            vals += [str(row[c]) for c in range(Trajectory.LEFT_BOUND_X, Trajectory.TIME + 1)]
            f.write(",".join(vals) + "\n")


def load_ttl(ttl_path: str) -> Trajectory:
    """models/trajectory.py:350-358."""
    with open(ttl_path, "r") as f:
        header = f.readline().split(",")
        assert len(header) >= 6
    data = np.loadtxt(ttl_path, dtype=float, delimiter=",", skiprows=1)
    trajectory = Trajectory(len(data), int(header[0]),
                            (float(header[3]), float(header[4]), float(header[5])))
    trajectory.points[:, :data.shape[1]] = data
    return trajectory
