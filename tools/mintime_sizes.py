#!/usr/bin/env python3
"""Wall time of the batched min-time solve at several batch sizes (MGKT, N = 828, tol 1e-6; second call of each size).
python tools/mintime_sizes.py [sizes ...]      (RL_MT_KKT4=0/1 selects the two- / four-front elimination)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem, perturbed_widths
sizes = [int(a) for a in sys.argv[1:]] or [1, 32, 256, 1024]
prob = mgkt_problem()
prob.solve_batch(*[a[:2] for a in perturbed_widths(prob, 2)], max_iter=8)
out = []
for B in sizes:
    left, right = perturbed_widths(prob, B)
    for rep in range(2):
        t0 = time.perf_counter(); X, U, T, st = prob.solve_batch(left, right, max_iter=300, tol=1e-6); dt = time.perf_counter() - t0
    out.append(f"B={B}: {dt:.3f} s ({int((st[:, 5] == 1).sum())} converged, {st[:, 0].mean():.1f} it)")
print(os.environ.get("RL_MT_KKT4", "default"), " | ".join(out))
