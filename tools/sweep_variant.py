#!/usr/bin/env python3
"""One launch configuration of the sweep kernel on the benchmarked batch (Monza, N = 2000, 1024 instances, max_iter = 5):
kernel time and a digest of the results, for A/B runs of diagnostic builds (RL_LIB_PATH, RL_SWEEP_BLOCK, RL_FORCE_RESIDENCY).
python tools/sweep_variant.py [out.npy]"""
import hashlib, os, sys
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
i_start = batch.default_i_start(len(cx), k, 5, seed=0)
ms = []
for rep in range(4):
    ctrl, xy, ns, status, st = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, W, i_start)
    ms.append(st.kernel_ms)
print({v: os.environ.get(v) for v in ("RL_LIB_PATH", "RL_SWEEP_BLOCK", "RL_FORCE_RESIDENCY")}, "kernel ms", [round(m, 3) for m in ms],
      "block", st.block_threads, "lds", st.lds_bytes, "digest", hashlib.sha1(ctrl.tobytes()).hexdigest()[:12])
