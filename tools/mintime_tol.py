#!/usr/bin/env python3
"""The batched min-time solve at tighter tolerances than the default 1e-6 (256 width-perturbed MGKT tracks, N = 828; and the
coarse 8 m grid of the cross-check tests): converged count and iterations.   python tools/mintime_tol.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem, perturbed_widths, variant_problem  # noqa: E402
for name, prob in (("mgkt 1 m", mgkt_problem()), ("mgkt 8 m", variant_problem("mgkt", 8.0, {}))):
    left, right = perturbed_widths(prob, 256)
    for tol in (1e-6, 1e-7, 1e-8, 1e-9):
        X, U, T, st = prob.solve_batch(left, right, max_iter=300, tol=tol)
        print(f"{name} N={prob.N} tol {tol:g}: converged {(st[:, 5] == 1).sum()} failed {(st[:, 5] == 2).sum()} of {len(st)}, iterations {st[:, 0].mean():.1f} (max {st[:, 0].max():.0f}), "
              f"kkt max {st[:, 1].max():.2e}", flush=True)
