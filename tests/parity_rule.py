"""The per-instance acceptance rule of the batched parity tests and of bench.py: NEAREST BRANCH.

The reference forms each constraint row as min(L,R) - (p - b*z_old) (optimizer.py:236-248); the bound
on z it implies divides the ~1e-13 m rounding noise of O(1e3) m coordinates by b, which is ~1e-10 for
the first samples inside a basis function's support.  Whether such a row binds -- and whether a QP is
feasible at all (optimizer.py:291-293) -- is decided by the last bits of the sampled positions and bound
points, so the reference's OWN arithmetic does not define the result of an instance to 1e-4 m unless
those bits happen not to matter (DESIGN_HISTORY.md "Conditioning").  What the reference's arithmetic does define
is a small set of BRANCHES: the lines reached by the legitimate roundings of the same operations.

How the branches are enumerated: RE-ROUNDINGS of the oracle.  Besides the strict build, the same oracle
source is run (a) compiled with FMA contraction (oracle/Makefile) and (b) with the sampled positions,
headings and bound points moved by -1/0/+1 ulp, pseudo-randomly but reproducibly per seed
(oracle/mincurv_oracle.c: orc_set_rerounding) -- each an equally legitimate rounding of the same
arithmetic (another summation order, another libm).

For every instance b:
    dev[b]     = max |HIP - strict oracle| over the sampled line [m]
    nearest[b] = min( dev[b], min over re-roundings r of max |HIP - oracle_r| )
    accept  <=>  nearest[b] <= 1e-4 m (north_star)
i.e. the HIP line must reproduce, to the north_star tolerance, the strict oracle OR one of its
re-roundings.  No multiple of a spread is accepted any more (round 3 accepted 10x the spread, which can be
metres).  An instance that is not within 1e-4 m of any branch seen so far gets more re-roundings (up to
MAX_SEEDS = 24 seeded ones + the FMA build) before it is declared failing.  No percentile, no majority:
one failing instance fails the test.  The HIP kernel's own logic is pinned independently of all this,
step by step, by tests/test_sweep_replay.py.

PARITY_SURVEY=<path>: append one JSON line per classified batch to <path> (diagnostics; the rule still
asserts).
"""
import json
import os

import numpy as np

from oracle import oracle as orc

TOL_M = 1e-4       # north_star
TIGHT_M = 1e-6     # what is held on the reference-run fixtures
MAX_SEEDS = 24


class ParityOracle:
    """Strict oracle + its re-roundings (kept line by line) on one batch of width-form instances."""

    def __init__(self, t, cx, cy, k, length, N, widths, i_start, n_seeds=2, nthreads=None):
        self.args = (t, cx, cy, k, length, N)
        self.widths = np.ascontiguousarray(widths)
        self.i_start = i_start
        self.nthreads = nthreads or min(16, os.cpu_count() or 1)
        B = self.widths.shape[0]
        import time
        t0 = time.perf_counter()
        _, self.xy0, self.ns0 = orc.solve_width_batch(*self.args, self.widths, i_start, nthreads=self.nthreads)
        self.t_strict = time.perf_counter() - t0
        with orc.fma_variant():
            _, fxy, _ = orc.solve_width_batch(*self.args, self.widths, i_start, nthreads=self.nthreads)
        self.noise = np.abs(fxy - self.xy0).reshape(B, -1).max(axis=1)
        self.seeds_used = np.zeros(B, dtype=int)
        # alts[b] = list of (label, line [N,2]): every re-rounding of instance b run so far
        self.alts = [[("fma", fxy[b])] for b in range(B)]
        if n_seeds > 0:
            self.more(np.arange(B), n_seeds)

    def more(self, which, n_seeds):
        """n_seeds further re-roundings of the instances `which`."""
        which = np.asarray(which)
        if len(which) == 0 or n_seeds <= 0:
            return
        rep = np.repeat(self.widths[which], n_seeds, axis=0)
        seeds = np.concatenate([self.seeds_used[b] + 1 + np.arange(n_seeds) for b in which]).astype(np.uint64)
        _, xy, _ = orc.solve_width_batch(*self.args, rep, self.i_start, nthreads=self.nthreads, seeds=seeds)
        xy = xy.reshape(len(which), n_seeds, *self.xy0.shape[1:])
        for j, b in enumerate(which):
            for q in range(n_seeds):
                self.alts[b].append((f"seed{self.seeds_used[b] + 1 + q}", xy[j, q]))
                self.noise[b] = max(self.noise[b], float(np.abs(xy[j, q] - self.xy0[b]).max()))
            self.seeds_used[b] += n_seeds

    def nearest(self, b, line):
        """(distance, label) of the branch of instance b closest to `line`."""
        best, lab = float(np.abs(line - self.xy0[b]).max()), "strict"
        for label, alt in self.alts[b]:
            d = float(np.abs(line - alt).max())
            if d < best:
                best, lab = d, label
        return best, lab


def classify(xy, po):
    B = xy.shape[0]
    dev = np.abs(xy - po.xy0).reshape(B, -1).max(axis=1)
    nearest = np.empty(B)
    branch = [""] * B
    for b in range(B):
        nearest[b], branch[b] = po.nearest(b, xy[b])
    # instances not on any branch seen so far: enumerate more branches
    while True:
        open_ = np.where((nearest > TOL_M) & (po.seeds_used < MAX_SEEDS))[0]
        if len(open_) == 0:
            break
        po.more(open_, 6)
        for b in open_:
            nearest[b], branch[b] = po.nearest(b, xy[b])
    within = dev <= TOL_M
    ok = nearest <= TOL_M
    return {"dev": dev, "nearest": nearest, "branch": branch, "noise": po.noise.copy(), "within": within,
            "on_rerounding_branch": ok & ~within, "ok": ok, "seeds_used": po.seeds_used.copy()}


def summary(c):
    dev, noise, nearest = c["dev"], c["noise"], c["nearest"]
    return {"sample": int(len(dev)), "within_1e-4_of_strict_oracle": int(c["within"].sum()),
            "within_1e-4_of_a_rerounding_branch": int(c["on_rerounding_branch"].sum()), "failing": int((~c["ok"]).sum()),
            "dev_m_median": float(np.median(dev)), "dev_m_max": float(dev.max()),
            "nearest_branch_dev_m_median": float(np.median(nearest)), "nearest_branch_dev_m_max": float(nearest.max()),
            "oracle_rerounding_spread_m_median": float(np.median(noise)),
            "oracle_rerounding_spread_m_max": float(noise.max()),
            "rerounding_seeds_max": int(c["seeds_used"].max()),
            "within_1e-6": int((dev <= TIGHT_M).sum())}


def _survey(label, c):
    path = os.environ.get("PARITY_SURVEY")
    if not path:
        return
    with open(path, "a") as f:
        f.write(json.dumps({"label": label, "dev": c["dev"].tolist(), "nearest": c["nearest"].tolist(), "branch": c["branch"],
                            "noise": c["noise"].tolist(), "seeds": c["seeds_used"].tolist()}) + "\n")


def batch_parity(xy, po, label=""):
    c = classify(xy, po)
    s = summary(c)
    print(f"[parity {label}] {s}")
    _survey(label, c)
    bad = np.where(~c["ok"])[0]
    assert len(bad) == 0, [(int(b), "dev", float(c["dev"][b]), "nearest branch", float(c["nearest"][b]),
                            "oracle spread", float(c["noise"][b])) for b in bad[:10]]
    assert s["within_1e-4_of_strict_oracle"] + s["within_1e-4_of_a_rerounding_branch"] == s["sample"]
    return c
