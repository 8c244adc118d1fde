"""Host-side mirror of `spline_traj_optm.models.race_track.RaceTrack` (models/race_track.py:8-104).

The k=3 boundary/centre splines, their discretisations, the rings and `fill_trajectory_boundaries`
(race_track.py:23-37, 98-104) are the min-curvature path.  The interpolants of x, y, yaw, curvature and
boundary distance against the abscissa (race_track.py:39-96, CasADi `interpolant('bspline')` + `hessian`
there) serve the min-time NLP; they are the build's OWN implementation -- periodic cubic splines through the
same samples (scipy, host-side set-up like the spline fits) -- with the reference's attribute names.
"""
import numpy as np

from .trajectory import BSplineTrajectory, Trajectory


class Ring:
    """Minimal stand-in for shapely.geometry.LinearRing: a closed polyline with `.coords`."""

    def __init__(self, coords):
        c = np.asarray(coords, dtype=np.float64)[:, :2]
        if len(c) > 1 and np.array_equal(c[0], c[-1]):
            c = c[:-1]
        self._c = np.ascontiguousarray(c)
        self.name = "ring"

    @property
    def coords(self):
        return np.vstack([self._c, self._c[:1]])  # shapely repeats the first vertex at the end

    @property
    def vertices(self):
        return self._c


def _make_ring(xy):
    try:
        from shapely.geometry import LinearRing  # used when the user's environment has shapely
        return LinearRing(xy)
    except Exception:
        return Ring(xy)


class _DMLike(np.ndarray):
    """ndarray that also answers casadi.DM's `.full()`: the reference's CLI reads `frenet_to_global(...)[:, 2].full()`
    (entrypoints/traj_opt_double_track.py:78)."""

    def full(self):
        return np.asarray(self)


class RaceTrack:
    def __init__(self, name: str, left: np.ndarray, right: np.ndarray, centerline: np.ndarray,
                 s=10.0, interval=2.0) -> None:
        assert left.shape[0] >= 3 and right.shape[0] >= 3
        assert left.shape[1] >= 2 and right.shape[1] >= 2
        self.left_s = BSplineTrajectory(left[:, :2], s, 3)
        self.right_s = BSplineTrajectory(right[:, :2], s, 3)
        self.center_s = BSplineTrajectory(centerline[:, :2], s, 3)

        self.left_d = self.left_s.sample_along(interval)
        self.right_d = self.right_s.sample_along(interval)
        self.center_d = self.center_s.sample_along(interval)

        self.left_r = _make_ring(self.left_d[:, :2])
        self.right_r = _make_ring(self.right_d[:, :2])
        self.center_r = _make_ring(self.center_d[:, :2])

        self.name = name
        self.fill_trajectory_boundaries(self.center_d)
        self._build_interpolants()

    # ---- race_track.py:39-96: interpolants against the abscissa (distance from the start / finish line)
    def _build_interpolants(self):
        from scipy.interpolate import CubicSpline
        cd = self.center_d
        length = self.center_s.get_length()
        self.abscissa = cd[:, Trajectory.DIST_TO_SF_BWD].copy()                                    # :56
        xy = cd[:, Trajectory.X:Trajectory.Y + 1]
        dist_l = np.linalg.norm(xy - cd[:, Trajectory.LEFT_BOUND_X:Trajectory.LEFT_BOUND_Y + 1], axis=1)   # :51
        dist_r = np.linalg.norm(xy - cd[:, Trajectory.RIGHT_BOUND_X:Trajectory.RIGHT_BOUND_Y + 1], axis=1)  # :52
        ss = np.r_[self.abscissa, length]

        def periodic(v):
            return CubicSpline(ss, np.r_[v, v[0]], bc_type="periodic")
        self._sx, self._sy = periodic(xy[:, 0]), periodic(xy[:, 1])
        self._sl, self._sr = periodic(dist_l), periodic(-dist_r)                                  # right distances are negative (:58-59)
        wrap = lambda s_: np.mod(np.asarray(s_, dtype=np.float64), length)  # noqa: E731  (s_mod, :65)
        self.x_intp = lambda s_: self._sx(wrap(s_))
        self.y_intp = lambda s_: self._sy(wrap(s_))
        self.left_intp = lambda s_: self._sl(wrap(s_))
        self.right_intp = lambda s_: self._sr(wrap(s_))
        self.yaw_intp = lambda s_: np.arctan2(self._sy(wrap(s_), 1), self._sx(wrap(s_), 1))       # :73
        def curvature(s_):                                                                        # :74
            q = wrap(s_)
            dx, dy, d2x, d2y = self._sx(q, 1), self._sy(q, 1), self._sx(q, 2), self._sy(q, 2)
            return (dx * d2y - dy * d2x) / np.sqrt((dx ** 2 + dy ** 2) ** 3)
        self.curvature_intp = curvature
        self.fast_curvature_intp = lambda s_: np.interp(wrap(s_), ss, np.r_[curvature(self.abscissa), curvature(self.abscissa[:1])])

    def frenet_to_global(self, s, t, xi):
        """(abscissa, lateral offset, relative heading) -> (x, y, heading)  (race_track.py:87-96)."""
        yaw0 = self.yaw_intp(s)
        phi = yaw0 + np.asarray(xi, dtype=np.float64)
        phi = np.arctan2(np.sin(phi), np.cos(phi))                                                # align_yaw(., 0)
        s, t = np.asarray(s, dtype=np.float64).reshape(-1), np.asarray(t, dtype=np.float64).reshape(-1)
        return np.stack([self.x_intp(s) - np.sin(yaw0).reshape(-1) * t, self.y_intp(s) + np.cos(yaw0).reshape(-1) * t,
                         phi.reshape(-1)], axis=-1).view(_DMLike)

    def fill_trajectory_boundaries(self, traj: Trajectory):
        """Fills the boundary properties of a trajectory in place (race_track.py:98-104)."""
        traj.fill_bounds(self.left_r, self.right_r, max_dist=100.0)
