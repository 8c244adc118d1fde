#!/usr/bin/env python3
"""Three consecutive timed calls of the batched min-time solve (1024 width-perturbed MGKT tracks): the first call of a fresh
process on a fresh box is ~20 % slower than the following ones.   python tools/mintime_repeat.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem, perturbed_widths
prob = mgkt_problem()
left, right = perturbed_widths(prob, 1024)
prob.solve_batch(left[:2], right[:2], max_iter=8)
for i in range(3):
    t0 = time.perf_counter()
    X, U, T, st = prob.solve_batch(left, right, max_iter=300, tol=1e-6)
    print("call", i, time.perf_counter() - t0, flush=True)
