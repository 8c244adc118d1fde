#!/usr/bin/env python3
"""Throughput of the batched min-time solve (BASELINE config 5): B width-perturbed copies of the reference's
example track (MGKT, interval 1 m, N = 828 nodes), one call.  Prints one JSON line.   python tools/bench_mintime.py [B]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spline_trajectory_optimization_amd.min_time_optm.example import timed_batch_solve  # noqa: E402

print(json.dumps(timed_batch_solve(int(sys.argv[1]) if len(sys.argv) > 1 else 256)))
