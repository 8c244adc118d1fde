"""GPU: the instances of the benchmarked batch whose interior-point iteration count differs most from the CPU twin's."""
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
ctrl, xy, z, st, rs = ops.global_batch_host(trk, W, 0.25, 6, dof=2)
its = st[:, 0]
print("gpu its mean", its.mean(), "max", its.max(), "count > 115:", int((its > 115).sum()), np.argsort(its)[-5:], np.sort(its)[-5:])
for b in np.argsort(its)[-3:]:
    b = int(b)
    for no in range(1, 7):
        r = orc.global_mincurv_xy(t, cx, cy, k, 2000, W[b, :, 0], W[b, :, 1], 0.25, 1.0, no)
        c1, x1, z1, s1, _ = ops.global_batch_host(trk, W[b:b + 1], 0.25, no, dof=2)
        print(f"instance {b} n_outer {no}: gpu its {int(s1[0, 0])} twin {int(r[4][0])}  |dxy| {np.abs(x1[0] - r[2]).max():.2e}  k2 {s1[0, 2]:.8f}/{r[4][2]:.8f} halved {int(s1[0, 7])}/{int(r[4][7])}", flush=True)
