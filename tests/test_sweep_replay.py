"""Step-level pin of the sweep kernel (k_sweep) on the benchmarked configuration.

The end-to-end result of the reference's Gauss-Seidel driver is noise-sensitive on some instances
(DESIGN_HISTORY.md "Conditioning"), so besides the per-instance end-to-end rule (tests/parity_rule.py) every
single step the kernel takes is checked on its own, TEACHER-FORCED: the recording instantiation of the
kernel (rl_debug_dump_enable) stores, for the first instances of the batch and for each of their
2 * max_iter * (n - k) steps, the control points the step started from, what it assembled (H, g, the
clamp interval of the box rows) and what it decided (accepted?, the new control point).  The oracle
then re-derives each step from the kernel's own pre-step control points (oracle/mincurv_oracle.c:
orc_replay_steps = optimize_one of optimizer.py:263-293 on that state).  A wrong step cannot hide
behind the noise of the steps before it, and a noise-flipped decision cannot make the rest of the
trajectory incomparable.

Checked per step:
  assembly   H, g <= 1e-9 relative (to the term scale for g); the clamp interval lo/hi within the row's
             own noise radius eps/b (and <= 1e-9 m where that radius is tiny);
  decision   accepted <=> the oracle solved the QP -- unless the oracle's own margin (hi - lo, or the slack
             of a b == 0 row) is inside its noise radius, where either verdict is a legal rounding;
  clamp      new control point == clamp(-g/H, lo, hi) of the KERNEL'S OWN dumped numbers, bit for bit.
"""
import ctypes

import numpy as np
import pytest

from conftest import spline
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

HEAD = 16  # rl_kernels.hpp: kSweepDumpHead


def _record(rl, trk, widths, i_start, n_inst):
    lib = rl.lib.load()
    rl.lib.check(lib.rl_debug_dump_enable(n_inst))
    try:
        ctrl, xy, ns, status, st = rl.ops.solve_batch_host(trk, rl.lib.BOUNDS_WIDTHS, widths, i_start)
        n = trk.n
        steps = 2 * len(i_start) * (n - trk.k)
        buf = np.zeros(n_inst * steps * (HEAD + 2 * n))
        rl.lib.check(lib.rl_debug_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(buf)))
    finally:
        rl.lib.check(lib.rl_debug_dump_enable(0))
    return (ctrl, xy, ns, status), buf.reshape(n_inst, steps, HEAD + 2 * n)


@pytest.fixture(scope="module")
def rl():
    from spline_trajectory_optimization_amd import _lib, batch, ops
    _lib.Context.get(0)

    class NS:
        pass
    ns = NS()
    ns.lib, ns.ops, ns.batch = _lib, ops, batch
    # this module pins the FAST arithmetic (its tolerances, dumps and replays are statements about it); the library's default --
    # the reference-order arithmetic since round 6 -- is the subject of tests/test_reference_order.py
    _lib.Context.get(0).set_arith(_lib.ARITH_FAST)
    yield ns
    _lib.Context.get(0).set_arith(_lib.ARITH_DEFAULT)


def check_steps(rec, rep, n, label):
    """rec [S, HEAD+2n] kernel records, rep [S, REPLAY_STRIDE] oracle re-derivations."""
    S = len(rec)
    H_g, g_g = rec[:, 1:3], rec[:, 3:5]
    lohi_g = rec[:, 5:9]                      # lo_x, hi_x, lo_y, hi_y
    ok_g = rec[:, 10] == 1.0
    z_g = rec[:, 11:13]
    st_o = rep[:, 0].astype(int)
    H_o, g_o, lohi_o, x_o, rad_o, zslack = rep[:, 1:3], rep[:, 3:5], rep[:, 5:9], rep[:, 9:11], rep[:, 11:15], rep[:, 15]
    eps = 8.0 * 2.220446049250313e-16 * 2048.0
    # ---- assembly
    relH = np.abs(H_g - H_o) / np.abs(H_o)
    assert relH.max() <= 1e-9, (label, "H", relH.max())
    # g is a sum of terms of either sign: measure against the scale of H * (coordinate scale)
    gscale = np.abs(H_o) * 2048.0 + np.abs(g_o)
    relg = np.abs(g_g - g_o) / gscale
    assert relg.max() <= 1e-9, (label, "g", relg.max())
    finite = np.isfinite(lohi_o) & np.isfinite(lohi_g)
    assert np.array_equal(np.isfinite(lohi_o), np.isfinite(lohi_g)), (label, "an interval end is infinite on one side only")
    dlohi = np.where(finite, np.abs(np.where(finite, lohi_g, 0.0) - np.where(finite, lohi_o, 0.0)), 0.0)
    tol = np.maximum(4.0 * rad_o, 1e-9)
    worst = (dlohi / tol).max()
    assert (dlohi <= tol).all(), (label, "clamp interval outside the row's noise radius", worst)
    # ---- decision
    ok_o = st_o == 0
    width_o = np.stack([lohi_o[:, 1] - lohi_o[:, 0], lohi_o[:, 3] - lohi_o[:, 2]], axis=1)
    width_noise = np.stack([rad_o[:, 0] + rad_o[:, 1], rad_o[:, 2] + rad_o[:, 3]], axis=1)
    noisy_verdict = (np.abs(width_o) <= 4.0 * width_noise).any(axis=1) | (np.abs(zslack) <= 4.0 * eps)
    differ = ok_g != ok_o
    assert not (differ & ~noisy_verdict).any(), (label, "verdict differs outside the noise bound",
                                                  np.where(differ & ~noisy_verdict)[0][:10].tolist())
    # ---- clamp: the kernel's own numbers, bit for bit (IEEE division, then max, then min)
    with np.errstate(all="ignore"):
        zx = np.minimum(np.maximum(-g_g[:, 0] / H_g[:, 0], lohi_g[:, 0]), lohi_g[:, 1])
        zy = np.minimum(np.maximum(-g_g[:, 1] / H_g[:, 1], lohi_g[:, 2]), lohi_g[:, 3])
        ok_own = (rec[:, 9] == 0.0) & (H_g[:, 0] > 0) & (H_g[:, 1] > 0) & np.isfinite(H_g).all(axis=1) & \
            np.isfinite(g_g).all(axis=1) & (lohi_g[:, 0] <= lohi_g[:, 1]) & (lohi_g[:, 2] <= lohi_g[:, 3])
    np.testing.assert_array_equal(ok_g, ok_own)
    np.testing.assert_array_equal(z_g[ok_g, 0], zx[ok_g])
    np.testing.assert_array_equal(z_g[ok_g, 1], zy[ok_g])
    # ---- where both solved and the binding rows are well conditioned the minimisers agree tightly
    both = ok_g & ok_o
    dz = np.abs(z_g - x_o)
    clamp_noise = np.stack([np.maximum(rad_o[:, 0], rad_o[:, 1]), np.maximum(rad_o[:, 2], rad_o[:, 3])], axis=1)
    ztol = np.maximum(4.0 * clamp_noise, 1e-6)
    assert (dz[both] <= ztol[both]).all(), (label, "new control point", (dz[both] / ztol[both]).max())
    # ---- chaining: the control points a step starts from are those the previous step left
    cx, cy = rec[:, HEAD:HEAD + n], rec[:, HEAD + n:HEAD + 2 * n]
    idx = rec[:, 0].astype(int)
    for s in range(S - 1):
        ex, ey = cx[s].copy(), cy[s].copy()
        if ok_g[s]:
            ex[idx[s]], ey[idx[s]] = z_g[s]
            ex[0], ey[0] = ex[n - 5], ey[n - 5]; ex[1], ey[1] = ex[n - 4], ey[n - 4]          # optimizer.py:281-285
            ex[n - 3], ey[n - 3] = ex[2], ey[2]; ex[n - 2], ey[n - 2] = ex[3], ey[3]; ex[n - 1], ey[n - 1] = ex[4], ey[4]
        assert np.array_equal(ex, cx[s + 1]) and np.array_equal(ey, cy[s + 1]), (label, "state chain broken at step", s)
    return {"steps": S, "accepted": int(ok_g.sum()), "verdict_differs_in_noise": int(differ.sum()),
            "max_rel_H": float(relH.max()), "max_rel_g": float(relg.max()), "max_interval_over_tol": float(worst),
            "noisy_verdict_steps": int(noisy_verdict.sum())}


def _monza_widths(rl, fits, rings, N, B, seed):
    t, cx, cy, k, length = spline(fits, "c100")
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    orc.fill_bounds(pts, rings[0], rings[1], 100.0)
    wl, wr = rl.batch.half_widths_from_bounds(pts)
    return rl.batch.width_batch(wl, wr, B, seed=seed)


@pytest.mark.parametrize("N,B,max_iter,n_inst", [(400, 8, 2, 4), (2000, 1024, 5, 1024)])
def test_sweep_steps_teacher_forced(rl, fits, rings, N, B, max_iter, n_inst):
    """(2000, 1024, 5): the benchmarked configuration itself (Monza widths, B=1024, max_iter=5, the
    global-residency kernel variant bench.py runs): EVERY step of EVERY instance of the batch (1024 x 610 steps).
    The oracle replays run eight at a time (ctypes releases the GIL)."""
    t, cx, cy, k, length = spline(fits, "c100")
    n = len(cx)
    widths = _monza_widths(rl, fits, rings, N, B, seed=1234)
    i_start = rl.batch.default_i_start(n, k, max_iter, seed=0)
    trk = rl.lib.Track(rl.lib.Context.get(0), t, cx, cy, k, N)
    # the product instantiation first: the recording one must reproduce it bit for bit
    ctrl0, xy0, ns0, status0, st0 = rl.ops.solve_batch_host(trk, rl.lib.BOUNDS_WIDTHS, widths, i_start)
    (ctrl, xy, ns, status), rec = _record(rl, trk, widths, i_start, n_inst)
    np.testing.assert_array_equal(ctrl, ctrl0); np.testing.assert_array_equal(xy, xy0)
    np.testing.assert_array_equal(ns, ns0); np.testing.assert_array_equal(status, status0)
    steps = 2 * max_iter * (n - k)
    assert rec.shape == (n_inst, steps, HEAD + 2 * n)
    def one(b):
        r = rec[b]
        # the visiting order is the reference's (optimizer.py:305-324)
        order = []
        for it in range(max_iter):
            for p in range(2):
                for s in range(n - k):
                    i_loop = s if p == 0 else (n - k) - s
                    kk = i_loop + int(i_start[it])
                    if kk >= n - 3:
                        kk = kk - (n - 3) + 2
                    order.append(kk)
        np.testing.assert_array_equal(r[:, 0].astype(int), order)
        np.testing.assert_array_equal(r[0, HEAD:HEAD + n], cx); np.testing.assert_array_equal(r[0, HEAD + n:], cy)
        assert int(r[:, 10].sum()) == int(ns[b].sum())
        ringL, ringR = orc.width_rings(t, cx, cy, k, N, widths[b])
        rep = orc.replay_steps(t, k, N, ringL, ringR, r[:, 0].astype(np.int32), r[:, HEAD:HEAD + n], r[:, HEAD + n:])
        return check_steps(r, rep, n, f"N={N} instance {b}")

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        infos = list(ex.map(one, range(n_inst)))
    for b, info in enumerate(infos[:12]):
        print(f"[replay N={N} B={B} it={max_iter}] instance {b}: {info}")
    tot = {key: sum(i[key] for i in infos) for key in ("steps", "accepted", "verdict_differs_in_noise", "noisy_verdict_steps")}
    worst = {key: max(i[key] for i in infos) for key in ("max_rel_H", "max_rel_g", "max_interval_over_tol")}
    print(f"[replay N={N} B={B} it={max_iter}] ALL {n_inst} instances: {tot} {worst}")
    assert tot["steps"] == n_inst * steps
