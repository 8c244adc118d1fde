import ctypes, json, os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
os.environ["RL_LIB_PATH"] = sys.argv[1]
from spline_trajectory_optimization_amd import _lib, batch, ops
from spline_trajectory_optimization_amd.models.race_track import RaceTrack
ctx = _lib.Context.get()
centre, left, right = batch.load_monza(); line = batch.monza_centerline(100.0, 5); t, cx, cy, k = line._tck()
tg = RaceTrack("Monza", left, right, centre); traj = line.sample_along(ts=np.linspace(0, 1, 2000, endpoint=False)); tg.fill_trajectory_boundaries(traj)
wl, wr = batch.half_widths_from_bounds(traj.points)
B = 1024; ARITH = int(sys.argv[2])
W = batch.width_batch(wl, wr, B, seed=1234); ist = batch.default_i_start(len(cx), k, 5, seed=0)
trk = _lib.Track(ctx, t, cx, cy, k, 2000)
lib = _lib.load()
_lib.check(lib.rl_debug_dump_enable(1))
for _ in range(2):
    out = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, W, ist, arith=ARITH)
buf = np.zeros(B * 4 * 16)
_lib.check(lib.rl_debug_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(buf)))
s = buf.reshape(B, 4, 16)
print("kernel ms", out[4].kernel_ms)
print("per wave, cycles per step, slots 0..5 (phase1, barA, phase2, barB, refresh(+ahead), barC), total:")
for w in range(4):
    print(w, (s[:, w, :6].mean(axis=0) / 610).round(0).tolist(), round(s[:, w, 6].mean() / 610))
if ARITH == 1:
    sc, p2, ex = s[:, :, 11].sum(), s[:, :, 12].sum(), s[:, :, 14].sum()
    print("window scans per instance %.0f, pass-2 rounds per scan %.3f, exact sign passes per scan %.4f" % (sc / B, p2 / sc, ex / sc))
