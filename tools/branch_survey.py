#!/usr/bin/env python3
"""How far the cheaper arithmetic modes of the sweep end from the reference-order one -- which returns the CPU oracle's bits
(tests/test_reference_order.py), so it serves as the oracle for ALL instances of a batch: per instance the largest
control-point / sample deviation of RL_ARITH_FAST and RL_ARITH_BRANCH from RL_ARITH_REFERENCE on the benchmarked batch
(Monza N = 2000, 1024 width-perturbed instances, max_iter = 5), kernel times interleaved on one box, and the distance of all three
from the reference's own runs (fixtures G7b / G7d).   usage: tools/branch_survey.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spline_trajectory_optimization_amd import _lib, batch, ops  # noqa: E402
from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: E402

N, B = 2000, 1024
ctx = _lib.Context.get(0)
centre, left, right = batch.load_monza()
line = batch.monza_centerline(100.0, 5)
t, cx, cy, k = line._tck()
traj = line.sample_along(ts=np.linspace(0.0, 1.0, N, endpoint=False))
RaceTrack("Monza", left, right, centre).fill_trajectory_boundaries(traj)
wl, wr = batch.half_widths_from_bounds(traj.points)
W = batch.width_batch(wl, wr, B, seed=1234)
ref_runs = {}
for fname in ("G7b_benchmarked_config.npz", "G7d_benchmarked_batch_sample.npz"):
    g = np.load(os.path.join(ROOT, "tests", "golden", fname))
    for key in [str(k_) for k_ in g["cases"] if "bench" in str(k_)]:
        bi = int(key.split("bench")[1])
        W[bi] = g[f"{key}_widths"]
        ref_runs[bi] = (g[f"{key}_cx"], g[f"{key}_cy"])
w = torch.from_numpy(W).cuda()
trk = _lib.Track(ctx, t, cx, cy, k, N)
ist = batch.default_i_start(len(cx), k, 5, seed=0)
modes = {"fast": _lib.ARITH_FAST, "branch": _lib.ARITH_BRANCH, "reference": _lib.ARITH_REFERENCE}
out = {m: ops.solve_batch_torch(trk, _lib.BOUNDS_WIDTHS, w, ist, arith=a) for m, a in modes.items()}
torch.cuda.synchronize()
ms = {m: [] for m in modes}
for _ in range(5):
    for m, a in modes.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.solve_batch_torch(trk, _lib.BOUNDS_WIDTHS, w, ist, out=out[m], arith=a); e1.record()
        torch.cuda.synchronize()
        ms[m].append(e0.elapsed_time(e1))
res = {"workload": f"Monza N={N}, {B} instances, max_iter=5", "kernel_ms_median": {m: float(np.median(v)) for m, v in ms.items()}}
ref = out["reference"]["xy"].cpu().numpy()
ref_ns = out["reference"]["n_success"].cpu().numpy()
for m in ("fast", "branch"):
    xy = out[m]["xy"].cpu().numpy()
    dev = np.abs(xy - ref).reshape(B, -1).max(axis=1)
    ns = out[m]["n_success"].cpu().numpy()
    res[m] = {"dev_vs_reference_order_m": {"median": float(np.median(dev)), "p90": float(np.quantile(dev, 0.9)), "p99": float(np.quantile(dev, 0.99)),
                                           "max": float(dev.max())},
              "instances_beyond_1e-4_m": int((dev > 1e-4).sum()), "instances_beyond_1e-6_m": int((dev > 1e-6).sum()),
              "instances_with_other_success_counts": int((ns.reshape(B, -1) != ref_ns.reshape(B, -1)).any(axis=1).sum()),
              "worst_instances": [int(b) for b in np.argsort(dev)[-5:][::-1]]}
for m in modes:
    c = out[m]["ctrl"].cpu().numpy()
    res.setdefault("vs_the_references_own_runs_m", {})[m] = {int(bi): float(np.hypot(c[bi, :, 0] - rc[0], c[bi, :, 1] - rc[1]).max())
                                                              for bi, rc in sorted(ref_runs.items())}
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
