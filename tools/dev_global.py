"""Dev aid: global QP kernels (fast path and generic) vs the CPU twin on Monza; run on the GPU box."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import golden, spline
from oracle import oracle as orc
from spline_trajectory_optimization_amd import batch, ops, _lib

fits = golden("G1_spline_fits.npz"); rg = golden("G1_rings.npz")
for name, N in (("c100", 500), ("c100", 2000), ("c30", 2000), ("c0p8", 2000), ("c100", 4000)):
    t, cx, cy, k, L = spline(fits, name)
    u = np.linspace(0, 1, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, L, u); orc.fill_bounds(pts, rg["ringL"], rg["ringR"])
    wl, wr = batch.half_widths_from_bounds(pts)
    trk = _lib.Track(_lib.Context.get(None), t, cx, cy, k, N)
    for no in (1, 6):
        r = orc.global_mincurv(t, cx, cy, k, N, wl, wr, 0.25, no)
        w = np.stack([wl, wr], axis=1)[None]
        ctrl, xy, a, st, rs = ops.global_batch_host(trk, w, 0.25, no)
        print(name, N, "outer", no, "twin", r[4][:5], "\n   gpu ", st[0, :5],
              "\n   |da| %.3e |dxy| %.3e  ms %.3f lds %d block %d" % (np.abs(a[0] - r[3]).max(), np.abs(xy[0] - r[2]).max(), rs.kernel_ms, rs.lds_bytes, rs.block_threads), "us/iter total %.2f factor %.2f solves %.2f | rows: P1+flush %.2f asm+LA1 %.2f P2 %.2f P3+red %.2f LA2 %.2f P4+upd %.2f" % (rs.kernel_ms * 1e3 / st[0, 0], st[0, 5] / st[0, 0] / 100, st[0, 6] / st[0, 0] / 100, st[0, 1] / st[0, 0] / 100, st[0, 2] / st[0, 0] / 100, st[0, 3] / st[0, 0] / 100, int(st[0, 4]) / st[0, 0] / 100, (st[0, 4] % 1) * 1e3 / st[0, 0] / 100, st[0, 7] / st[0, 0] / 100), flush=True)
    if N == 2000 and name == "c100":
        B = 1024
        W = batch.width_batch(wl, wr, B, seed=1234)
        for rep in range(2):
            ctrl, xy, a, st, rs = ops.global_batch_host(trk, W, 0.25, 6)
            print("batch", B, "ms", rs.kernel_ms, "solves/s", B / rs.kernel_ms * 1e3, "ipm its mean", st[:, 0].mean(), "viol max", st[:, 3].max(),
                  "k2", st[:, 2].min(), st[:, 2].max(), "us/iter factor %.2f solves %.2f" % (st[:, 5].mean() / st[:, 0].mean() / 100, st[:, 6].mean() / st[:, 0].mean() / 100), flush=True)
