"""Mirror of `spline_traj_optm.models.vehicle` (models/vehicle.py:6-47): parameter container and
speed-dependent acceleration lookups used by the QSS simulator."""
from dataclasses import dataclass

import numpy as np
from scipy import interpolate


@dataclass
class VehicleParams:
    acc_speed_lookup: np.ndarray
    dcc_speed_lookup: np.ndarray
    max_lon_acc_mpss: float  # signed acceleration (positive)
    max_lon_dcc_mpss: float  # signed deceleration (negative)
    max_left_acc_mpss: float  # signed acceleration (positive)
    max_right_acc_mpss: float  # signed acceleration (negative)
    max_speed_mps: float
    max_jerk: float


class Vehicle:
    def __init__(self, param: VehicleParams):
        self.param = param
        self.acc_intp = interpolate.CubicSpline(
            self.param.acc_speed_lookup[:, 0], self.param.acc_speed_lookup[:, 1])
        self.dcc_intp = interpolate.CubicSpline(
            self.param.dcc_speed_lookup[:, 0], self.param.dcc_speed_lookup[:, 1])

    def lookup_acc_from_speed(self, speed_mps: float):
        return self.acc_intp(speed_mps)

    def lookup_dcc_from_speed(self, speed_mps: float):
        return self.dcc_intp(speed_mps)

    def lookup_acc_circle(self, lat=None, lon=None, model='ellipse'):
        assert (lat is not None) or (lon is not None)
        if model == 'ellipse':
            p = self.param
            if lat is not None:
                lat = np.clip(lat, p.max_right_acc_mpss, p.max_left_acc_mpss)
                max_lat = p.max_left_acc_mpss if lat > 0.0 else p.max_right_acc_mpss
                return self.__ellipse(lat, max_lat, p.max_lon_acc_mpss, p.max_lon_dcc_mpss)
            lon = np.clip(lon, p.max_lon_dcc_mpss, p.max_lon_acc_mpss)
            max_lon = p.max_lon_acc_mpss if lon > 0.0 else p.max_lon_dcc_mpss
            return self.__ellipse(lon, max_lon, p.max_left_acc_mpss, p.max_right_acc_mpss)

    def __ellipse(self, val, x, y_pos, y_neg):
        r = np.sqrt(1.0 - val ** 2 / x ** 2)
        return y_pos * r, y_neg * r
