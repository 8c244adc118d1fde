"""GPU: k_global_xy against its CPU twin on a few cases, then the benchmark batch (timing, properties)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa
from test_global_qp import monza_widths  # noqa
from oracle import oracle as orc  # noqa
from spline_trajectory_optimization_amd import _lib, ops, batch  # noqa

fits = golden("G1_spline_fits.npz")
cases = [("c100", 500, 1), ("c100", 500, 6), ("c100", 2000, 1), ("c100", 2000, 6), ("c100", 2000, 10)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cases = cases[:2]
for tag, N, no in cases:
    t, cx, cy, k, u, wl, wr = monza_widths(fits, tag, N)
    ocx, ocy, oxy, oz, ost = orc.global_mincurv_xy(t, cx, cy, k, N, wl, wr, 0.25, 1.0, no)
    trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, N)
    ctrl, xy, z, st, rs = ops.global_batch_host(trk, np.stack([wl, wr], 1)[None], 0.25, no, dof=2)
    print(f"[{tag} N={N} outer={no}] |dz| {np.abs(z[0] - oz).max():.2e} |dxy| {np.abs(xy[0] - oxy).max():.2e} ipm {int(st[0, 0])}/{int(ost[0])} "
          f"k2 {st[0, 1]:.6f}->{st[0, 2]:.8f} (twin {ost[2]:.8f}) viol {st[0, 3]:.1e} step {st[0, 4]:.2e}/{ost[4]:.2e} act {st[0, 5:8]} / {ost[5:8]} "
          f"{rs.kernel_ms:.2f} ms lds {rs.lds_bytes} block {rs.block_threads}", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 2000)
B = 1024
W = batch.width_batch(wl, wr, B, seed=1234)
trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, 2000)
for rep in range(3):
    ctrl, xy, z, st, rs = ops.global_batch_host(trk, W, 0.25, 6, dof=2)
    print(f"batch {B}: {rs.kernel_ms:.2f} ms, ipm {st[:, 0].mean():.1f} ({st[:, 0].min():.0f}..{st[:, 0].max():.0f}), k2 {st[:, 1].mean():.5f}->{st[:, 2].mean():.5f}, "
          f"viol max {st[:, 3].max():.1e}, halved {st[:, 7].mean():.2f}, finite {np.isfinite(xy).all()}", flush=True)
ctrl2, xy2, z2, st2, _ = ops.global_batch_host(trk, W, 0.25, 6, dof=2)
print("bit-reproducible:", np.array_equal(xy, xy2) and np.array_equal(st, st2))
for b in (0, 511, 1023):
    _, _, oxy, oz, ost = orc.global_mincurv_xy(t, cx, cy, k, 2000, W[b, :, 0], W[b, :, 1], 0.25, 1.0, 6)
    print(f"instance {b}: |dxy| {np.abs(xy[b] - oxy).max():.2e} ipm {int(st[b, 0])}/{int(ost[0])} halved {st[b, 7]}/{ost[7]}")
