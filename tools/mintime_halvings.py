#!/usr/bin/env python3
"""Histogram of the accepted step lengths of the min-time solve (number of halvings of the fraction-to-the-boundary length)
over all iterations of 1024 instances.  Needs a diagnostic build:  RL_MT_HIST=1 python -c "import __graft_entry__ as g; g.build()"."""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from spline_trajectory_optimization_amd.min_time_optm.example import mgkt_problem, perturbed_widths
from spline_trajectory_optimization_amd import _lib
prob = mgkt_problem()
left, right = perturbed_widths(prob, 1024)
X, U, T, st = prob.solve_batch(left, right, max_iter=300, tol=1e-6)
lib = _lib.load() if hasattr(_lib, "load") else _lib.lib
h = (ctypes.c_ulonglong * 16)()
print("rc", lib.rl_debug_mt_hist(h))
h = np.array(list(h))
print("halvings histogram", h, "fraction first-trial", h[0] / h.sum())
