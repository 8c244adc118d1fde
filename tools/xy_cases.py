"""GPU vs twin on a list of (tag, N, n_outer, lon) cases given on the command line as tag:N:outer:lon."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, spline  # noqa
from test_global_qp import monza_widths  # noqa
from oracle import oracle as orc  # noqa
from spline_trajectory_optimization_amd import _lib, ops  # noqa
fits = golden("G1_spline_fits.npz")
for arg in sys.argv[1:]:
    tag, N, no, lon = arg.split(":"); N = int(N); no = int(no); lon = float(lon)
    if tag == "l10":
        t, cx, cy, k, L = spline(fits, tag)
        wl = np.full(N, 4.0) + np.sin(np.arange(N) * 0.05); wr = np.full(N, 3.0) + np.cos(np.arange(N) * 0.03)
    else:
        t, cx, cy, k, u, wl, wr = monza_widths(fits, tag, N)
    ocx, ocy, oxy, oz, ost = orc.global_mincurv_xy(t, cx, cy, k, N, wl, wr, 0.25, lon, no)
    trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, N)
    try:
        ctrl, xy, z, st, rs = ops.global_batch_host(trk, np.stack([wl, wr], 1)[None], 0.25, no, dof=2, lon=lon)
    except Exception as e:
        print(arg, "->", type(e).__name__, e); continue
    print(f"[{arg}] |dz| {np.abs(z[0] - oz).max():.2e} |dxy| {np.abs(xy[0] - oxy).max():.2e} ipm {int(st[0, 0])}/{int(ost[0])} "
          f"k2 {st[0, 2]:.8f}/{ost[2]:.8f} halved {int(st[0, 7])}/{int(ost[7])} step {st[0, 4]:.3e}/{ost[4]:.3e} {rs.kernel_ms:.2f} ms", flush=True)
