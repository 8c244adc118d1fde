"""Dev aid (GPU box): phase times of k_global_qp2 per interior-point iteration, from a -DRL_G2_PROFILE build
(RL_LIB_PATH=...): row waves' laps [I1 rows+flush, wait for the factorisation, assembly, I4/I5, wait for the corrector,
I7+I8] and the linear-algebra wave's factorisation / solves."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from conftest import golden, spline
from oracle import oracle as orc
from spline_trajectory_optimization_amd import batch, ops, _lib
fits = golden("G1_spline_fits.npz"); rg = golden("G1_rings.npz")
t, cx, cy, k, L = spline(fits, "c100"); N = 2000
u = np.linspace(0, 1, N, endpoint=False)
pts = orc.sample_along(t, cx, cy, k, L, u); orc.fill_bounds(pts, rg["ringL"], rg["ringR"])
wl, wr = batch.half_widths_from_bounds(pts)
trk = _lib.Track(_lib.Context.get(None), t, cx, cy, k, N)
W = batch.width_batch(wl, wr, 1024, seed=1234)
for rep in range(2):
    ctrl, xy, a, st, rs = ops.global_batch_host(trk, W, 0.25, 6)
it = st[:, 0]
us = lambda v: (v / it).mean() / 100.0     # wall_clock64 ticks of 10 ns
i45 = np.floor(st[:, 4]); i6 = (st[:, 4] - i45) * 1e3
print("kernel ms", rs.kernel_ms, "iterations", it.mean())
print("per iteration [us]: I1 rows+flush %.2f | assembly %.2f | wait factor+solve %.2f | I4/I5 %.2f | wait corrector %.2f | I7+I8 %.2f | LA: factor %.2f solves %.2f"
      % (us(st[:, 1]), us(st[:, 3]), us(st[:, 2]), us(i45), us(i6), us(st[:, 7]), us(st[:, 5]), us(st[:, 6])))
