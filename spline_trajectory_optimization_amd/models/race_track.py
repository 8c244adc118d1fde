"""Host-side mirror of `spline_traj_optm.models.race_track.RaceTrack` (models/race_track.py:8-104).

Only the part on the min-curvature path is built: the k=3 boundary/centre splines, their
discretisations, the rings and `fill_trajectory_boundaries` (race_track.py:23-37, 98-104).  The
CasADi interpolants of race_track.py:39-96 belong to the min-time NLP (out of scope).
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

    def fill_trajectory_boundaries(self, traj: Trajectory):
        """Fills the boundary properties of a trajectory in place (race_track.py:98-104)."""
        traj.fill_bounds(self.left_r, self.right_r, max_dist=100.0)
