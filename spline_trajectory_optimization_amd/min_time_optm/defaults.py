"""Parameter values of the reference's only configuration file
(min_time_optm/example/traj_opt_double_track.yaml:1-69: IPOPT settings, scaling, QSS estimates, the
kart's double-track model) as plain Python data, so that the min-time path runs without a YAML file in
the working directory (the reference's CLI reads one from CWD, entrypoints/traj_opt_double_track.py:17-21)."""

SOLVER = {"max_iter": 500, "tol": 0.1, "constr_viol_tol": 0.1, "speed_cap": 30.0, "average_track_width": 7.0,
          "interval": 1.0}

ESTIMATES = {
    "acc_speed_loopup": [[0.0, 5.0], [15.0, 2.5], [30.0, 0.5]],      # (sic: the reference's key spelling)
    "dcc_speed_lookup": [[0.0, -10.0], [15.0, -10.0], [30.0, -10.0]],
    "max_lon_acc_mpss": 5.0, "max_lon_dcc_mpss": -10.0, "max_left_acc_mpss": 10.0, "max_right_acc_mpss": -10.0,
    "max_speed_mps": 25.0, "max_jerk_mpsc": 10.0,
}

MODEL = {
    "kd_f": 0.0, "kb_f": 0.0, "mass": 209.0, "Jzz": 209.0, "lf": 0.55, "lr": 0.45, "twf": 1.0, "twr": 1.0,
    "delta_max": 0.314159, "vehicle_width": 1.0, "safety_margin": 1.0,
    "fr": 0.01, "hcog": 0.25, "kroll_f": 0.5,
    "cl_f": 0.0, "cl_r": 0.0, "rho": 1.2041, "A": 0.4, "cd": 0.8, "mu": 1.5,
    "Bf": 9.62, "Cf": 2.59, "Ef": 1.0, "Fz0_f": 512.0, "eps_f": -0.0813,
    "Br": 8.62, "Cr": 2.65, "Er": 1.0, "Fz0_r": 512.0, "eps_r": -0.1263,
    "Pmax": 6000.0, "Fd_max": 1000.0, "Fb_max": -2000.0, "Td": 1.0, "Tb": 1.0, "Tdelta": 1.0,
}
