"""Helper of test_mintime.py: solve the coarse MGKT problem for a few iterations in THIS process (whose environment
selects the Hessian implementation of the library) and save the iterate.   python mintime_run.py out.npz iterations [node spacing, m] [tol]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from mintime_problem import mgkt_problem  # noqa: E402
from oracle import sqp_twin as tw  # noqa: E402
from spline_trajectory_optimization_amd import ops  # noqa: E402
from spline_trajectory_optimization_amd.min_time_optm import defaults  # noqa: E402

d = mgkt_problem(float(sys.argv[3]) if len(sys.argv) > 3 else 8.0, defaults.ESTIMATES)
P = tw.Problem(defaults.MODEL, d["s"], d["kappa"], d["left"], d["right"], d["L"], defaults.SOLVER["average_track_width"],
               defaults.SOLVER["speed_cap"])
X0, U0, T0 = P.unpack(tw.initial_point(P, d["speed"], d["seg_time"]))
X, U, T, st = ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X0[None], U0[None], T0[None],
                                      max_iter=int(sys.argv[2]), tol=float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12)
np.savez(sys.argv[1], X=X, U=U, T=T, st=st)
