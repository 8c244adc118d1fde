#!/usr/bin/env python3
"""The device-resident entry point rl_mintime_solve_batch_dev (no host round trip: all max_iter iterations are enqueued)
against the host entry point (which polls every 8 iterations and stops), same batch of 1024 tracks.
python tools/mintime_dev_path.py [max_iter]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spline_trajectory_optimization_amd import ops
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem, perturbed_widths

max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
prob = mgkt_problem()
B = 1024
left, right = perturbed_widths(prob, B)
rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)
prob.solve_batch(left[:2], right[:2], max_iter=8)
t0 = time.perf_counter(); X, U, T, st = prob.solve_batch(left, right, max_iter=max_iter, tol=1e-6); th = time.perf_counter() - t0
dev = torch.device("cuda:0")
d = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
for rep_i in range(2):
    dX, dU, dT = d(rep(prob.X0)), d(rep(prob.U0)), d(rep(prob.T0))
    args = (prob.model, d(prob.s), d(prob.kappa), d(left), d(right), prob.margin, prob.track_length, dX, dU, dT)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    stats = ops.mintime_solve_torch(*args, average_track_width=prob.average_track_width, speed_cap=prob.speed_cap, max_iter=max_iter, tol=1e-6)
    te = time.perf_counter() - t0
    torch.cuda.synchronize(); td = time.perf_counter() - t0
print(f"max_iter {max_iter}: host entry {th:.3f} s; device entry: enqueue {te:.3f} s, done {td:.3f} s; converged {int((stats[:, 5] == 1).sum())} / {B}, "
      f"same result: {np.array_equal(dT.cpu().numpy(), T)}")
