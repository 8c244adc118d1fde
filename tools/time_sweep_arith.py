#!/usr/bin/env python3
"""Kernel time of the benchmarked batch (Monza N = 2000, 1024 width-perturbed instances, max_iter = 5) in both
arithmetic modes of the sweep, event-timed, interleaved (boxes differ by a few per cent: compare on ONE box).
RL_LIB_PATH selects the build.  usage: tools/time_sweep_arith.py [repeats]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spline_trajectory_optimization_amd import _lib, batch, ops  # noqa: E402
from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: E402

rep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N, B = 2000, 1024
ctx = _lib.Context.get(0)
centre, left, right = batch.load_monza()
line = batch.monza_centerline(100.0, 5)
t, cx, cy, k = line._tck()
traj = line.sample_along(ts=np.linspace(0.0, 1.0, N, endpoint=False))
RaceTrack("Monza", left, right, centre).fill_trajectory_boundaries(traj)
wl, wr = batch.half_widths_from_bounds(traj.points)
w = torch.from_numpy(batch.width_batch(wl, wr, B, seed=1234)).cuda()
trk = _lib.Track(ctx, t, cx, cy, k, N)
ist = batch.default_i_start(len(cx), k, 5, seed=0)
out = {a: ops.solve_batch_torch(trk, _lib.BOUNDS_WIDTHS, w, ist, arith=a) for a in (0, 1)}
torch.cuda.synchronize()
ms = {0: [], 1: []}
for _ in range(rep):
    for a in (0, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.solve_batch_torch(trk, _lib.BOUNDS_WIDTHS, w, ist, out=out[a], arith=a); e1.record()
        torch.cuda.synchronize()
        ms[a].append(e0.elapsed_time(e1))
digest = [float(out[a]["ctrl"].sum().item()) for a in (0, 1)]
print(f"{os.environ.get('RL_LIB_PATH', 'product'):40s} fast {min(ms[0]):.3f} ms (median {np.median(ms[0]):.3f})   reference-order "
      f"{min(ms[1]):.3f} ms (median {np.median(ms[1]):.3f})   ratio {np.median(ms[1]) / np.median(ms[0]):.3f}   digests {digest}")
