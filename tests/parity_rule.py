"""The per-instance acceptance rule of the batched parity tests and of bench.py.

For every instance b:
    dev[b]   = max |HIP - oracle| over the sampled line [m]  (the closer of the two oracle builds)
    noise[b] = max |oracle - oracle_fma| : the SAME oracle source compiled with and without FMA
               contraction (oracle/Makefile) -- two equally IEEE-legal roundings of the reference's
               arithmetic.  Where they disagree the reference's formulation itself does not define
               the answer to that accuracy (rows divided by b ~ 1e-10, DESIGN.md "Conditioning").
    accept  <=>  dev <= 1e-4 (north_star)   or   ( noise > 1e-4  and  dev <= 10 * noise )
and, for instances that are not certified ill-conditioned, the per-pass success counts must equal
those of one of the two oracle builds.  No percentile, no majority: one failing instance fails the test.
"""
import os

import numpy as np

from oracle import oracle as orc

TOL_M = 1e-4       # north_star
TIGHT_M = 1e-6     # what is held on the reference-run fixtures
NOISE_M = 2e-2     # documentation only: typical size of a noise-driven deviation


def oracle_pair(t, cx, cy, k, length, N, widths, i_start, nthreads=None):
    """(xy, n_success) of the strict oracle and of its FMA-contracted build on the same instances."""
    nthreads = nthreads or min(16, os.cpu_count() or 1)
    _, oxy, ons = orc.solve_width_batch(t, cx, cy, k, length, N, widths, i_start, nthreads=nthreads)
    with orc.fma_variant():
        _, fxy, fns = orc.solve_width_batch(t, cx, cy, k, length, N, widths, i_start, nthreads=nthreads)
    return oxy, ons, fxy, fns


def classify(xy, ns, pair):
    oxy, ons, fxy, fns = pair
    B = xy.shape[0]
    d_o = np.abs(xy - oxy).reshape(B, -1).max(axis=1)
    d_f = np.abs(xy - fxy).reshape(B, -1).max(axis=1)
    dev = np.minimum(d_o, d_f)
    noise = np.abs(oxy - fxy).reshape(B, -1).max(axis=1)
    within = dev <= TOL_M
    certified = ~within & (noise > TOL_M)
    bounded = dev <= 10.0 * np.maximum(noise, 1e-6)
    same = (ns.reshape(B, -1) == ons.reshape(B, -1)).all(axis=1) | (ns.reshape(B, -1) == fns.reshape(B, -1)).all(axis=1)
    ok = within | (certified & bounded)
    return {"dev": dev, "noise": noise, "within": within, "certified": certified, "ok": ok, "same_counts": same}


def batch_parity(xy, ns, pair, label=""):
    c = classify(xy, ns, pair)
    B = xy.shape[0]
    dev, noise = c["dev"], c["noise"]
    print(f"[parity {label}] B={B}  within_1e-4: {int(c['within'].sum())}  certified_ill_conditioned: "
          f"{int(c['certified'].sum())}  failing: {int((~c['ok']).sum())}  | dev median={np.median(dev):.2e} "
          f"max={dev.max():.2e}  oracle re-rounding spread median={np.median(noise):.2e} max={noise.max():.2e}  "
          f"<=1e-6: {int((dev <= TIGHT_M).sum())}")
    bad = np.where(~c["ok"])[0]
    assert len(bad) == 0, [(int(b), float(dev[b]), float(noise[b])) for b in bad[:10]]
    assert int(c["within"].sum()) + int(c["certified"].sum()) == B
    # success counts: an instance inside the tolerance and not noise-flagged must have made the same
    # accept / skip decisions as one of the two oracle builds
    strict = c["within"] & (noise <= TOL_M)
    badc = np.where(strict & ~c["same_counts"])[0]
    assert len(badc) == 0, ("success counts differ on well-conditioned instances", badc[:10].tolist())
    return c
