"""Dev aid (GPU box): k_global_qp2 on the benchmarked batch; with RL_LIB_PATH pointing at a -DRL_G2_PROFILE build the
per-iteration time of the factorisation and of the solves (linear-algebra wave) is printed as well."""
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
print(os.environ.get("RL_LIB_PATH","default"), "ms", rs.kernel_ms, "its", st[:,0].mean(), "us/iter factor %.2f solves %.2f" % (st[:, 5].mean() / st[:, 0].mean() / 100, st[:, 6].mean() / st[:, 0].mean() / 100))
