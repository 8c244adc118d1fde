#!/usr/bin/env python3
"""Which instances of the width-perturbed MGKT batch do not converge, and what their last iterations look like.
   python tools/mintime_stalls.py [B]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from spline_trajectory_optimization_amd.min_time_optm import defaults  # noqa: E402
from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import DoubleTrackProblem  # noqa: E402
from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: E402
from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams  # noqa: E402
from spline_trajectory_optimization_amd.simulator.simulator import Simulator  # noqa: E402
from mintime_problem import _load  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
est = defaults.ESTIMATES
rt = RaceTrack("MGKT", _load("MGKT_OUT_BOUND_enu.csv"), _load("MGKT_IN_BOUND_enu.csv"), _load("MGKT_CENTER_enu.csv"), s=1.0, interval=1.0)
veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                            est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"], est["max_speed_mps"],
                            est["max_jerk_mpsc"]))
traj = rt.center_d.copy(); rt.fill_trajectory_boundaries(traj)
traj = Simulator(veh).run_simulation(traj, False).trajectory
prob = DoubleTrackProblem({"N": len(traj), "model": defaults.MODEL, "race_track": rt, "traj_d": traj,
                           "average_track_width": 7.0, "speed_cap": 30.0})
e = np.random.default_rng(1234).uniform(-0.1, 0.15, size=(B, 1))
left, right = prob.left[None] * (1 + e), prob.right[None] * (1 + e)
X, U, T, st = prob.solve_batch(left, right, max_iter=300, tol=1e-6)
bad = np.where(st[:, 5] != 1)[0]
print(json.dumps({"B": B, "not_converged": bad.tolist(), "scale": (1 + e[bad, 0]).tolist(),
                  "iterations_hist": np.bincount((st[:, 0] // 20).astype(int)).tolist()}))
names = ["iters", "kkt", "viol", "compl", "lap", "status", "mu", "delta", "alpha", "refact", "alpha_ftb", "halvings"]
for b in bad[:3]:
    print(int(b), {n: float(v) for n, v in zip(names, st[b])})
    for k in (20, 30, 40, 50, 60, 80, 100, 150, 300):
        _, _, _, s1 = prob.solve_batch(left[b:b + 1], right[b:b + 1], max_iter=k, tol=1e-6)
        print("   after", k, {n: float(f"{v:.4g}") for n, v in zip(names, s1[0])})
