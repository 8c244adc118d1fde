#!/usr/bin/env python3
"""Harder variants than tools/mintime_robustness.py: wider width range, more model changes, coarser / finer grids."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spline_trajectory_optimization_amd.min_time_optm.example import variant_problem  # noqa: E402
tot = [0, 0]
for track, interval in (("mgkt", 0.25), ("mgkt", 3.0), ("mgkt", 16.0), ("monza", 20.0), ("monza", 3.0)):
    for mod in ({}, {"mu": 0.5}, {"mass": 1.5}, {"Pmax": 4.0}, {"Pmax": 0.5}, {"Cd": 3.0}, {"mu": 1.3}):
        try:
            prob = variant_problem(track, interval, mod)
            B = 32
            e = np.random.default_rng(11).uniform(-0.4, 0.5, size=(B, 1))
            t0 = time.time()
            X, U, T, st = prob.solve_batch(prob.left[None] * (1 + e), prob.right[None] * (1 + e), max_iter=400, tol=1e-6)
            row = {"track": track, "interval": interval, "model": mod, "N": prob.N, "converged": int((st[:, 5] == 1).sum()), "of": B,
                   "failed": int((st[:, 5] == 2).sum()), "it_mean": round(float(st[:, 0].mean()), 1), "it_max": float(st[:, 0].max()),
                   "wall_s": round(time.time() - t0, 2)}
            tot[0] += row["converged"]; tot[1] += B
        except Exception as ex:
            row = {"track": track, "interval": interval, "model": mod, "error": f"{type(ex).__name__}: {ex}"[:160]}
        print(json.dumps(row), flush=True)
print("total", tot)
