#!/usr/bin/env python3
"""Wall time of Simulator.run_simulation on the GPU (rl_qss_sim, host call incl. copies): one Monza trajectory at N = 2000 and
N = 500, and batches of them.   python tools/time_qss.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scipy.interpolate import CubicSpline
from spline_trajectory_optimization_amd import batch, ops

acc = CubicSpline([0.0, 50.0, 100.0], [10.0, 7.0, 0.5]); dcc = CubicSpline([0.0, 50.0, 100.0], [-13.0, -15.0, -20.0])
veh = (acc.x, acc.c, dcc.x, dcc.c, np.array([10.0, -20.0, 15.0, -15.0, 100.0, 30.0]))
line = batch.monza_centerline(100.0, 5)
for N in (2000, 500):
    pts = line.sample_along(ts=np.linspace(0, 1, N, endpoint=False)).points
    for B in (1, 1024):
        P = np.repeat(pts[None], B, axis=0) if B > 1 else pts
        ops.qss_sim(P, *veh)
        t0 = time.perf_counter(); out, it = ops.qss_sim(P, *veh); dt = time.perf_counter() - t0
        print(f"N={N} B={B}: {dt * 1e3:.1f} ms  ({B / dt:.0f} simulations/s)  global iterations {int(np.atleast_1d(it)[0])}  checksum {float(np.asarray(out)[..., 4].sum()):.9f}")
