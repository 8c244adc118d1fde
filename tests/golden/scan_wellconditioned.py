#!/usr/bin/env python3
"""Find WELL-CONDITIONED cases of the two drivers with the oracle alone (no GPU, no reference needed): cases on which the
strict oracle, its FMA build and n_re seeded +-1 ulp re-roundings (oracle/mincurv_oracle.c: orc_set_rerounding) all end
within 1e-6 m of each other -- i.e. on which the reference's own arithmetic defines the answer.  The cases of fixtures
G9b / G7c (tests/golden/make_golden.py) and JOINT_ROBUST_CASES (tests/test_hip_parity.py) were picked from this scan.

usage:  python tests/golden/scan_wellconditioned.py [joint|single] [N ...]
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "G1_spline_fits.npz"))
R = np.load(os.path.join(ROOT, "tests", "golden", "G1_rings.npz"))


def spread(fn, tag, N, ist, n_re=24):
    """(max distance between the strict run and any re-rounding [m], max control-point move [m], success counts)."""
    t, cx, cy, k, length = G[f"{tag}_t"], G[f"{tag}_cx"], G[f"{tag}_cy"], int(G[f"{tag}_k"]), float(G[f"{tag}_length"])

    def run(seed):
        return fn(t, cx, cy, k, length, N, R["ringL"], R["ringR"], ist, rerounding=seed)
    with ThreadPoolExecutor(8) as ex:
        runs = list(ex.map(run, range(n_re + 1)))
    with orc.fma_variant():
        runs.append(run(0))
    o = runs[0]
    sp = max(float(np.hypot(a[0] - o[0], a[1] - o[1]).max()) for a in runs[1:])
    return sp, float(np.hypot(o[0] - cx, o[1] - cy).max()), o[3].ravel().tolist()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "joint"
    sizes = [int(a) for a in sys.argv[2:]] or [200, 300, 400, 2000]
    fn = orc.run_joint_min_curvature_qp if which == "joint" else orc.run_min_curvature_qp
    n = len(G["c100_cx"])
    hi = n - 3 - (5 if which == "joint" else 0)
    for N in sizes:
        robust = []
        for i0 in range(2, hi):
            sp, move, counts = spread(fn, "c100", N, [i0], n_re=8)
            if sp < 1e-6:
                sp, move, counts = spread(fn, "c100", N, [i0], n_re=24)   # confirm with the fixture's re-rounding count
                if sp < 1e-6:
                    robust.append((i0, counts, round(move, 1), float(f"{sp:.1e}")))
        print(f"{which} N={N}: robust start indices (i_start, accepted, max move [m], spread [m]): {robust}", flush=True)
