#!/usr/bin/env python3
"""Where the waves of k_sweep run (diagnostic build -DRL_STAMPS): SIMD of every wave, CU / XCC of every workgroup, which
workgroups share a CU.   python tools/wave_placement.py <stamped library> [B] [arith]"""
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
ARITH = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W = batch.width_batch(wl, wr, B, seed=1234); ist = batch.default_i_start(len(cx), k, 5, seed=0)
trk = _lib.Track(ctx, t, cx, cy, k, 2000)
lib = _lib.load()
_lib.check(lib.rl_debug_dump_enable(1))
for _ in range(2):
    out = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, W, ist, arith=ARITH)
buf = np.zeros(B * 4 * 16)
_lib.check(lib.rl_debug_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(buf)))
s = buf.reshape(B, 4, 16)
code = s[:, :, 7].astype(np.int64)
simd = code & 3
cu = code >> 4          # cu | sh | se | xcc
print("SIMD of wave 0..3, first 16 workgroups:\n", simd[:16])
print("share of workgroups whose wave w sits on SIMD w:", float((simd == np.arange(4)[None, :]).all(axis=1).mean()))
print("distinct SIMD patterns:", {str(p): int(c) for p, c in zip(*np.unique(simd, axis=0, return_counts=True))})
cu0 = cu[:, 0]
assert (cu == cu0[:, None]).all()
ids, counts = np.unique(cu0, return_counts=True)
print("CUs used:", len(ids), "workgroups per CU: min", counts.min(), "max", counts.max())
print("XCC of workgroups 0..15:", (cu0[:16] >> 8).tolist())
for c in ids[:4]:
    members = np.nonzero(cu0 == c)[0]
    print("CU code", int(c), "holds workgroups", members.tolist(), "start cycles", (s[members, 0, 15] - s[:, 0, 15].min()).astype(np.int64).tolist())
