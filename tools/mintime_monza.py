#!/usr/bin/env python3
"""The min-time solve on the Monza track of the reference's examples (kart model of the yaml), from the QSS warm start.
   python tools/mintime_monza.py [interval_m]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spline_trajectory_optimization_amd import batch  # noqa: E402
from spline_trajectory_optimization_amd.min_time_optm import defaults  # noqa: E402
from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import optimise_track  # noqa: E402
from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: E402
from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams  # noqa: E402
interval = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
centre, left, right = batch.load_monza()
est = defaults.ESTIMATES
rt = RaceTrack("Monza", left, right, centre, s=10.0, interval=interval)
veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                            est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"], est["max_speed_mps"],
                            est["max_jerk_mpsc"]))
t0 = time.time()
out, X, U, T, st = optimise_track(rt, veh, defaults.MODEL, 10.0, defaults.SOLVER["speed_cap"], max_iter=400, tol=1e-6)
print(json.dumps({"N": len(out), "iterations": st[0], "status": st[5], "kkt": st[1], "viol": st[2], "compl": st[3], "lap_s": st[4],
                  "wall_s": time.time() - t0, "v_max": float(X[:, 5].max()), "v_min": float(X[:, 5].min())}))
