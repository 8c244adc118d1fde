"""Host-side mirror of `spline_traj_optm.models.vehicle` (models/vehicle.py:6-47).

`VehicleParams` keeps the reference's field names and order (it is constructed positionally in the
reference's tests); `Vehicle` keeps the attribute / method names the simulator and user code reach
for (`param`, `acc_intp`, `dcc_intp`, `lookup_*`).  The speed-dependent limits are scipy
`CubicSpline` objects exactly like the reference's, because their piecewise-polynomial coefficients
(`.x`, `.c`) are what the GPU simulator kernel consumes (`Simulator._vehicle_tables`).
"""
from dataclasses import dataclass

import numpy as np
from scipy.interpolate import CubicSpline


@dataclass
class VehicleParams:
    acc_speed_lookup: np.ndarray   # [m,2] speed [m/s] -> max longitudinal acceleration [m/s^2]
    dcc_speed_lookup: np.ndarray   # [m,2] speed [m/s] -> max deceleration (negative)
    max_lon_acc_mpss: float        # traction ellipse, forward semi-axis  (> 0)
    max_lon_dcc_mpss: float        # traction ellipse, braking semi-axis  (< 0)
    max_left_acc_mpss: float       # traction ellipse, left semi-axis     (> 0)
    max_right_acc_mpss: float      # traction ellipse, right semi-axis    (< 0)
    max_speed_mps: float
    max_jerk: float


def _limit_curve(table):
    table = np.asarray(table, dtype=np.float64)
    return CubicSpline(table[:, 0], table[:, 1])


class Vehicle:
    def __init__(self, param: VehicleParams):
        self.param = param
        self.acc_intp = _limit_curve(param.acc_speed_lookup)
        self.dcc_intp = _limit_curve(param.dcc_speed_lookup)

    def lookup_acc_from_speed(self, speed_mps):
        return self.acc_intp(speed_mps)

    def lookup_dcc_from_speed(self, speed_mps):
        return self.dcc_intp(speed_mps)

    def lookup_acc_circle(self, lat=None, lon=None, model='ellipse'):
        """Remaining acceleration budget on the other axis of the traction ellipse
        (models/vehicle.py:32-47): give `lat` to get the (forward, braking) pair, or `lon` to get the
        (left, right) pair."""
        if lat is None and lon is None:
            raise AssertionError("give lat or lon")
        if model != 'ellipse':
            return None
        p = self.param
        if lat is not None:
            used, neg, pos = lat, p.max_right_acc_mpss, p.max_left_acc_mpss
            other = (p.max_lon_acc_mpss, p.max_lon_dcc_mpss)
        else:
            used, neg, pos = lon, p.max_lon_dcc_mpss, p.max_lon_acc_mpss
            other = (p.max_left_acc_mpss, p.max_right_acc_mpss)
        used = np.clip(used, neg, pos)
        semi_axis = pos if used > 0.0 else neg
        remaining = np.sqrt(1.0 - used ** 2 / semi_axis ** 2)
        return other[0] * remaining, other[1] * remaining
