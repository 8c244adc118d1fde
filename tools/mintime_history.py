#!/usr/bin/env python3
"""Iteration history of one min-time solve (MGKT, 828 nodes) by re-running it with growing iteration limits.
   python tools/mintime_history.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem  # noqa: E402
prob = mgkt_problem()
names = ["iters", "kkt", "viol", "compl", "lap", "status", "mu", "delta", "alpha", "refact", "alpha_ftb", "halvings"]
for k in range(1, 100):
    _, _, _, st = prob.solve_batch(prob.left[None], prob.right[None], max_iter=k, tol=1e-6)
    print(k, " ".join(f"{n}={v:.3g}" for n, v in zip(names, st[0]) if n not in ("iters", "status")))
    if st[0, 5] == 1:
        break
