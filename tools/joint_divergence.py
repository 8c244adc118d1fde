#!/usr/bin/env python3
"""a13 diagnosis: WHERE and WHY the sliding-window kernel and the free-running oracle part company on a G9 case.

The kernel records the control points every window starts from (rl_debug_dump_enable); the oracle is run window by
window through its teacher-forced instrument (orc_replay_joint_windows fed with ITS OWN state, which reproduces
orc_run_joint_min_curvature_qp exactly -- checked).  Printed per case: the first windows at which the two states differ
by more than 1e-13 / 1e-9 / 1e-4 m, what entered that window (state difference), what the window did with it (verdicts,
smallest norm of an active row = the 1/b amplification of DESIGN_HISTORY.md section 5), and what came out.
    python tools/joint_divergence.py [case substring]          (GPU box)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from spline_trajectory_optimization_amd import _lib, ops  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
f = np.load(os.path.join(G, "G1_spline_fits.npz")); r = np.load(os.path.join(G, "G1_rings.npz"))
g9 = np.load(os.path.join(G, "G9_run_joint_min_curvature_qp.npz"))
t, cx, cy, k, length = f["c100_t"], f["c100_cx"], f["c100_cy"], int(f["c100_k"]), float(f["c100_length"])
n = len(cx)
ROWS = 48 + 9 * 3 * 256
STRIDE = ROWS + 2 * n
nwin = (n - 3 - 5) - 2
want = sys.argv[1] if len(sys.argv) > 1 else ""


def wrap(ex, ey):
    ex[0], ey[0] = ex[n - 5], ey[n - 5]; ex[1], ey[1] = ex[n - 4], ey[n - 4]
    ex[n - 3], ey[n - 3] = ex[2], ey[2]; ex[n - 2], ey[n - 2] = ex[3], ey[3]; ex[n - 1], ey[n - 1] = ex[4], ey[4]


for key in [str(k_) for k_ in g9["cases"] if want in str(k_)]:
    N = int(key.split("_")[1][1:])
    ist = g9[f"{key}_i_start"]
    trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, N)
    trk.set_rings(r["ringL"], r["ringR"])
    _lib.check(_lib.load().rl_debug_dump_enable(1))
    hcx, hcy, pts, ns, st = ops.mincurv_sweep_joint(trk, cx, cy, ist)
    buf = np.zeros(len(ist) * nwin * STRIDE)
    _lib.check(_lib.load().rl_debug_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(buf)))
    _lib.check(_lib.load().rl_debug_dump_enable(0))
    dump = buf.reshape(-1, STRIDE)
    kks = dump[:, 0].astype(np.int32)
    # free-running oracle, window by window
    ox, oy = cx.copy(), cy.copy()
    states = np.zeros((len(kks), 2, n)); outs = []
    for w, kk in enumerate(kks):
        states[w, 0], states[w, 1] = ox, oy
        head, rows = orc.replay_joint_windows(t, k, N, r["ringL"], r["ringR"], [kk], ox[None], oy[None], nthreads=1)
        o = head[0]
        ok = o[0] == 0.0 and o[1] == 0.0
        outs.append((ok, o.copy(), rows[0]))
        if ok:
            ox[kk:kk + 5], oy[kk:kk + 5] = o[5:10], o[10:15]
            wrap(ox, oy)
    fcx, fcy, _, fns = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, r["ringL"], r["ringR"], ist)
    assert np.array_equal(fcx, ox) and np.array_equal(fcy, oy), "window-by-window oracle != orc_run_joint_min_curvature_qp"
    ref_dev = np.hypot(ox - g9[f"{key}_cx"], oy - g9[f"{key}_cy"]).max()
    print(f"\n=== {key}: N={N} windows {len(kks)}; oracle vs the reference's run {ref_dev:.1e} m; HIP accepted {int(ns.sum())}, "
          f"oracle {sum(o_[0] for o_ in outs)}, reference {int(g9[f'{key}_n_ok'])}")
    kst = np.stack([dump[:, ROWS:ROWS + n], dump[:, ROWS + n:ROWS + 2 * n]], axis=1)
    diff = np.abs(kst - states).reshape(len(kks), -1).max(axis=1)
    for thr in (0.0, 1e-13, 1e-9, 1e-4):
        idx = np.where(diff > thr)[0]
        print(f"  first window whose START state differs by > {thr:g} m: {int(idx[0]) if len(idx) else None}"
              + (f" (difference {diff[idx[0]]:.2e} m)" if len(idx) else ""))
    # walk the windows where the difference grows by more than 100x or a verdict differs
    shown = 0
    for w in range(len(kks)):
        h = dump[w]
        feas_k, (feas_o, o, orows) = h[4] == 1.0, outs[w]
        d_in = diff[w]
        d_out = diff[w + 1] if w + 1 < len(kks) else np.hypot(hcx - ox, hcy - oy).max()
        if feas_k != feas_o or d_out > 100.0 * max(d_in, 1e-16):
            u0, u1 = int(h[1]), int(h[2])
            kr = h[48:48 + 9 * (u1 - u0)].reshape(u1 - u0, 9)
            act = []
            if feas_k:
                for c, (lo, hi) in enumerate(((5, 6), (7, 8))):
                    ax = kr[:, :5] @ h[5 + 5 * c:10 + 5 * c]
                    nz = np.linalg.norm(kr[:, :5], axis=1) > 0
                    on = nz & ((np.abs(ax - kr[:, lo]) <= 1e-9) | (np.abs(ax - kr[:, hi]) <= 1e-9))
                    act += [float(np.linalg.norm(kr[i, :5])) for i in np.where(on)[0]]
            db = np.abs(kr[:, 5:] - orows[u0:u1, 5:]).max()
            print(f"  window {w:3d} kk={int(h[0]):2d}: state in differs {d_in:.1e} m -> out {d_out:.1e} m; kernel "
                  f"{'accepts' if feas_k else 'rejects'}, oracle {'accepts' if feas_o else 'rejects'} "
                  f"(oracle on ITS state: relaxed {'ok' if o[35] else 'infeasible'}, tightened {'ok' if o[36] else 'infeasible'}); "
                  f"row bounds differ {db:.1e} m; active rows {len(act)}, smallest norm {min(act) if act else float('nan'):.1e}")
            shown += 1
            if shown >= 8:
                break
