"""GPU: for given instances of the benchmarked batch, both formulations with n_outer = 1..6: iteration counts and deviation from the twin."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa
from test_global_qp import monza_widths  # noqa
from oracle import oracle as orc  # noqa
from spline_trajectory_optimization_amd import _lib, ops, batch  # noqa
fits = golden("G1_spline_fits.npz")
t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 2000)
W = batch.width_batch(wl, wr, 1024, seed=1234)
trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, 2000)
for arg in sys.argv[1:]:
    dof, b = [int(x) for x in arg.split(":")]
    for no in range(1, 9):
        if dof == 1:
            r = orc.global_mincurv(t, cx, cy, k, 2000, W[b, :, 0], W[b, :, 1], 0.25, no)
        else:
            r = orc.global_mincurv_xy(t, cx, cy, k, 2000, W[b, :, 0], W[b, :, 1], 0.25, 1.0, no)
        c1, x1, z1, s1, _ = ops.global_batch_host(trk, W[b:b + 1], 0.25, no, dof=dof)
        print(f"dof {dof} instance {b} n_outer {no}: its gpu {int(s1[0, 0])} twin {int(r[4][0])}  |dxy| {np.abs(x1[0] - r[2]).max():.2e}  k2 {s1[0, 2]:.9f}/{r[4][2]:.9f} last step {s1[0, 4]:.2e}", flush=True)
