#!/usr/bin/env python3
"""ALL 1024 instances of the benchmarked batch (Monza N = 2000, width-perturbed, max_iter = 5, bench.py's sweep order): the
reference-order arithmetic of the sweep kernel against the CPU oracle's correctly rounded build, as bit patterns.
  phase 1 (GPU box):   python tools/full_batch_bits.py gpu  OUT.npz      -> control points + success counts of the HIP run
  phase 2 (any host):  python tools/full_batch_bits.py cpu  OUT.npz  [summary.json]   -> the oracle on every instance (about
                       23 s per instance and core), comparison, summary"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N, B, MAX_ITER = 2000, 1024, 5


def inputs():
    g = np.load(os.path.join(ROOT, "tests", "golden", "G1_spline_fits.npz"))
    r = np.load(os.path.join(ROOT, "tests", "golden", "G1_rings.npz"))
    t, cx, cy, k, length = g["c100_t"], g["c100_cx"], g["c100_cy"], int(g["c100_k"]), float(g["c100_length"])
    from oracle import oracle as orc
    from spline_trajectory_optimization_amd import batch
    pts = orc.sample_along(t, cx, cy, k, length, np.linspace(0.0, 1.0, N, endpoint=False))
    orc.fill_bounds(pts, r["ringL"], r["ringR"], 100.0)
    wl, wr = batch.half_widths_from_bounds(pts)
    return t, cx, cy, k, length, batch.width_batch(wl, wr, B, seed=1234), batch.default_i_start(len(cx), k, MAX_ITER, seed=0)


if sys.argv[1] == "gpu":
    from spline_trajectory_optimization_amd import _lib, ops
    t, cx, cy, k, length, W, ist = inputs()
    trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, N)
    ctrl, xy, ns, status, st = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, W, ist, arith=_lib.ARITH_REFERENCE)
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
    np.savez_compressed(sys.argv[2], ctrl=ctrl, ns=ns, widths_digest=np.float64(W.sum()), kernel_ms=np.float64(st.kernel_ms))
    print("wrote", sys.argv[2], "kernel ms", st.kernel_ms)
else:
    from oracle import oracle as orc
    d = np.load(sys.argv[2])
    t, cx, cy, k, length, W, ist = inputs()
    assert float(W.sum()) == float(d["widths_digest"]), "the two phases built different batches"
    t0 = time.time()
    with orc.cr_variant():
        octrl, oxy, ons = orc.solve_width_batch(t, cx, cy, k, length, N, W, ist, nthreads=os.cpu_count() or 1)
    same = np.array([np.array_equal(d["ctrl"][b], octrl[b]) and np.array_equal(d["ns"][b], ons[b]) for b in range(B)])
    dev = np.abs(d["ctrl"] - octrl).reshape(B, -1).max(axis=1)
    res = {"what": "RL_ARITH_REFERENCE on the benchmarked batch against oracle/libmincurv_oracle_cr.so: control points and per-pass success "
                   "counts as bit patterns, every instance", "instances": B, "bit_identical": int(same.sum()),
           "max_control_point_deviation_m": float(dev.max()), "oracle_seconds": time.time() - t0, "oracle_threads": os.cpu_count(),
           "hip_kernel_ms": float(d["kernel_ms"])}
    print(json.dumps(res))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)
