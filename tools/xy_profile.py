"""Diagnostic build (RL_XY_PROFILE=1 at build time): cycles per phase of k_global_xy, one instance and a batch."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa
from test_global_qp import monza_widths  # noqa
from spline_trajectory_optimization_amd import _lib, ops, batch  # noqa
fits = golden("G1_spline_fits.npz")
t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 2000)
trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, 2000)
names = ["-", "rows A + P x + reduce", "matrix span sums (MFMA)", "gather into band + vector weights", "factor || vector span sums", "rhs gather + check + solve 1", "rows B + reduce", "linearisation", "corrector span sums + gather", "rows C + reduce + update", "solve 2", "corrector weights"]
for B in (1, 1024):
    W = batch.width_batch(wl, wr, B, seed=1234)
    ops.global_batch_host(trk, W, 0.25, 6, dof=2)
    ctrl, xy, z, st, rs = ops.global_batch_host(trk, W, 0.25, 6, dof=2)
    its = st[:, 0].mean()
    pt = z.reshape(B, -1)[:, :12]
    tot = pt.sum(1).mean()
    print(f"B={B}: {rs.kernel_ms:.2f} ms, {its:.1f} its, cycles total {tot:.3e}")
    for q in range(1, 12):
        v = pt[:, q].mean()
        print(f"   {names[q]:24s} {v:12.0f} cycles  {100 * v / tot:5.1f} %   per iteration {v / its:9.0f}")
