#!/usr/bin/env python3
"""Where a wave of k_sweep spends its cycles, per phase (diagnostic build -DRL_STAMPS; run on the GPU box):
   python tools/stamp_sweep.py <path to the stamped library>
Stamps are s_memtime reads at the phase boundaries of the benchmarked configuration (Monza N=2000, B=1024,
max_iter=5); the build's run time is not quoted (the stamps fence the schedule), only the SHARES."""
import ctypes, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RL_LIB_PATH"] = sys.argv[1]
from spline_trajectory_optimization_amd import _lib, batch, ops  # noqa: E402
from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: E402
ctx = _lib.Context.get()
centre, left, right = batch.load_monza(); line = batch.monza_centerline(100.0, 5); t, cx, cy, k = line._tck()
tg = RaceTrack("Monza", left, right, centre); traj = line.sample_along(ts=np.linspace(0, 1, 2000, endpoint=False)); tg.fill_trajectory_boundaries(traj)
wl, wr = batch.half_widths_from_bounds(traj.points)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ARITH = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0 fast, 1 reference-order
W = batch.width_batch(wl, wr, B, seed=1234); ist = batch.default_i_start(len(cx), k, 5, seed=0)
trk = _lib.Track(ctx, t, cx, cy, k, 2000)
lib = _lib.load()
_lib.check(lib.rl_debug_dump_enable(1))
for _ in range(2):
    out = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, W, ist, arith=ARITH)
buf = np.zeros(B * 4 * 16)
_lib.check(lib.rl_debug_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(buf)))
s = buf.reshape(B, 4, 16)
tot = s[:, :, 6]
names = ["phase 1 (cost + constraints + reductions)", "barrier A", "phase 2 (QP, wave 0)", "barrier B", "phase 3 (refresh)", "barrier C"]
res = {"kernel_ms": out[4].kernel_ms, "B": B, "arith": ARITH, "cycles_per_wave_mean": float(tot.mean())}
for q, nm in enumerate(names):
    res[nm] = {"share_all_waves": float((s[:, :, q] / tot).mean()), "share_wave0": float((s[:, 0, q] / tot[:, 0]).mean()),
               "cycles_per_step_mean": float(s[:, :, q].mean() / 610)}
fine = ["refresh: curve evaluation + normal", "refresh: left ring search", "refresh: right ring search", "refresh: stores (+ loop)",
        "  of the two searches: window scan (loads, sign pass, exact tests)"]
for q, nm in zip((0, 1, 2, 3, 4), fine):
    res[nm] = {"share_wave0": float((s[:, 0, 8 + q] / tot[:, 0]).mean()), "cycles_per_step_wave0": float(s[:, 0, 8 + q].mean() / 610)}
if ARITH == 1:   # reference-order loop: slots 0 terms, 1 barrier, 2 in-order sums + QP, 3 barrier, 4 refresh, 5 barrier
    res["note"] = "reference-order loop: phase 1 = cost terms and rows; phase 2 = the six in-order sums (wave 0) + QP + derivative window"
    res["refresh: heading (inside evaluation)"] = {"share_wave0": float((s[:, 0, 13] / tot[:, 0]).mean()), "cycles_per_step_wave0": float(s[:, 0, 13].mean() / 610)}
    res["tiles that took the full heading, per instance (all waves)"] = float(s[:, :, 14].sum(axis=1).mean())
print(json.dumps(res, indent=1))
