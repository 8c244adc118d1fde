"""The global min-curvature QP (include/rl_mincurv.h: rl_mincurv_global_batch_*).

The reference has no working global solver (julia/spline_traj_opt.ipynb cells 6/8/9 do not converge,
SURVEY.md App. A.6), so there is no reference vector to match.  What is checked instead:

* CPU (not gpu): the oracle's twin `orc_global_mincurv` against an INDEPENDENT numpy restatement of
  the same formulation -- Gauss-Newton rows by central finite differences of scipy's BSpline curvature,
  constraint rows by moving one control point at a time -- through the KKT conditions of the QP
  (multipliers by non-negative least squares on the active rows), feasibility and cost decrease.
* GPU: the HIP kernel `k_global_qp` against that twin on the same inputs (control-point offsets and
  line to 1e-6 m, identical interior-point iteration counts), plus size-independent properties at the
  benchmark size (every instance inside its bounds, cost decreased, bit-reproducible, independent of
  the position in the batch)."""
import numpy as np
import pytest
from scipy.interpolate import BSpline
from scipy.optimize import nnls

from conftest import golden, spline
from oracle import oracle as orc

MARGIN = 0.25


def monza_widths(fits, tag, N):
    from spline_trajectory_optimization_amd import batch
    rg = golden("G1_rings.npz")
    t, cx, cy, k, L = spline(fits, tag)
    u = np.linspace(0, 1, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, L, u)
    orc.fill_bounds(pts, rg["ringL"], rg["ringR"])
    wl, wr = batch.half_widths_from_bounds(pts)
    return t, cx, cy, k, u, wl, wr


class NumpyGlobal:
    """Independent statement of the formulation in include/rl_mincurv.h (a15)."""

    def __init__(self, t, cx, cy, k, u):
        self.t, self.k, self.u = t, k, u
        self.n = len(cx)
        self.np = self.n - k
        self.c0 = np.stack([cx, cy], 1)
        g = np.array([t[j + 1:j + k + 1].mean() for j in range(self.np)])
        g -= np.floor(g)
        sx, sy = BSpline(t, cx, k), BSpline(t, cy, k)
        d = np.stack([sx(g, 1), sy(g, 1)], 1)
        self.nu = np.stack([-d[:, 1], d[:, 0]], 1) / np.hypot(d[:, 0], d[:, 1])[:, None]
        self.p0 = np.stack([sx(u), sy(u)], 1)
        d0 = np.stack([sx(u, 1), sy(u, 1)], 1)
        self.n0 = np.stack([-d0[:, 1], d0[:, 0]], 1) / np.hypot(d0[:, 0], d0[:, 1])[:, None]

    def ctrl(self, a):
        c = self.c0.copy()
        c[:self.np] += a[:, None] * self.nu
        c[self.np:] = c[:self.k]
        return c

    def kappa(self, a):
        c = self.ctrl(a)
        bx, by = BSpline(self.t, c[:, 0], self.k), BSpline(self.t, c[:, 1], self.k)
        dx, dy, ddx, ddy = bx(self.u, 1), by(self.u, 1), bx(self.u, 2), by(self.u, 2)
        return (dx * ddy - dy * ddx) / (dx * dx + dy * dy) ** 1.5

    def lateral(self, a):
        c = self.ctrl(a)
        r = np.stack([BSpline(self.t, c[:, 0], self.k)(self.u), BSpline(self.t, c[:, 1], self.k)(self.u)], 1)
        return ((r - self.p0) * self.n0).sum(1)

    def qp_at(self, a, h=1e-3):
        N, m = len(self.u), self.np
        G = np.zeros((N, m))
        A = np.zeros((N, m))
        for j in range(m):
            e = np.zeros(m)
            e[j] = 1.0
            G[:, j] = (self.kappa(a + h * e) - self.kappa(a - h * e)) / (2 * h)
            A[:, j] = self.lateral(e)
        res = self.kappa(a) - G @ a
        P = 2 * G.T @ G
        q = 2 * G.T @ res
        sc = m / np.trace(P)
        return P * sc + 1e-9 * np.eye(m), q * sc, A


@pytest.mark.parametrize("tag,N", [("c100", 500), ("c30", 500), ("c100", 2000)])
def test_twin_solves_the_qp_kkt(fits, tag, N):
    """One linearisation (n_outer=1): the twin's interior-point answer satisfies the KKT conditions of
    the independently assembled QP."""
    t, cx, cy, k, u, wl, wr = monza_widths(fits, tag, N)
    ocx, ocy, oxy, oa, st = orc.global_mincurv(t, cx, cy, k, N, wl, wr, MARGIN, 1)
    ng = NumpyGlobal(t, cx, cy, k, u)
    P, q, A = ng.qp_at(np.zeros(ng.np))
    lo, hi = -(wr - MARGIN), wl - MARGIN
    Aa = A @ oa
    assert (lo - Aa).max() <= 1e-9 and (Aa - hi).max() <= 1e-9          # primal feasibility
    r = P @ oa + q
    au, al = np.where(hi - Aa < 1e-6)[0], np.where(Aa - lo < 1e-6)[0]
    assert len(au) + len(al) > 0                                        # the bounds do bind on Monza
    lam, resid = nnls(np.concatenate([A[au].T, -A[al].T], 1), -r)
    print(f"[global kkt {tag} N={N}] active {len(au)}+{len(al)} of {2 * N}, stationarity residual "
          f"{resid:.2e} (|q|inf {np.abs(q).max():.2e}), ipm iterations {int(st[0])}")
    assert resid <= 1e-6 * np.abs(q).max()                              # stationarity with lambda >= 0
    # the twin's bookkeeping agrees with the independent evaluation
    assert st[1] == pytest.approx((ng.kappa(np.zeros(ng.np)) ** 2).sum(), rel=1e-10)
    assert st[2] == pytest.approx((ng.kappa(oa) ** 2).sum(), rel=1e-10)
    assert np.abs(ng.ctrl(oa) - np.stack([ocx, ocy], 1)).max() <= 1e-9
    assert np.abs(ng.lateral(oa) - Aa).max() <= 1e-9                    # rows are exactly linear in a


def test_twin_outer_iterations_decrease_cost_and_stay_inside(fits):
    t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 2000)
    ng = NumpyGlobal(t, cx, cy, k, u)
    prev = None
    for n_outer in (1, 2, 4, 6):
        ocx, ocy, oxy, oa, st = orc.global_mincurv(t, cx, cy, k, 2000, wl, wr, MARGIN, n_outer)
        lat = ng.lateral(oa)
        assert (lat - (wl - MARGIN)).max() <= 1e-9 and (-(wr - MARGIN) - lat).max() <= 1e-9
        assert st[3] <= 1e-9
        assert st[2] < 0.6 * st[1]                   # sum kappa^2 drops by > 40 % on Monza
        if prev is not None:
            assert st[4] < prev                      # the Gauss-Newton steps shrink
        prev = st[4]
    assert st[4] < 0.5                               # < 0.5 m control-point move on the 6th linearisation


def test_twin_margin_and_degenerate_inputs(fits):
    t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 500)
    # a zero-width track pins the line to the centre line
    z = np.full(500, 1e-3)
    _, _, oxy, oa, st = orc.global_mincurv(t, cx, cy, k, 500, z, z, 0.0, 2)
    ng = NumpyGlobal(t, cx, cy, k, u)
    assert np.abs(ng.lateral(oa)).max() <= 1e-3 + 1e-9
    # a wider margin gives a line that is still inside and costs more
    _, _, _, _, s0 = orc.global_mincurv(t, cx, cy, k, 500, wl, wr, 0.0, 3)
    _, _, _, _, s1 = orc.global_mincurv(t, cx, cy, k, 500, wl, wr, 1.0, 3)
    assert s0[3] <= 1e-9 and s1[3] <= 1e-9 and s1[2] > s0[2]


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def rl():
    from spline_trajectory_optimization_amd import _lib, batch, ops
    _lib.Context.get(0)  # raises loudly when the HIP extension / device is missing

    class NS:
        pass
    ns = NS()
    ns.lib, ns.ops, ns.batch = _lib, ops, batch
    return ns


@pytest.mark.gpu
@pytest.mark.parametrize("tag,N,n_outer", [("c100", 500, 1), ("c100", 500, 6), ("c100", 2000, 6), ("c30", 2000, 3),
                                           ("c0p8", 2000, 2), ("c100", 4000, 6), ("l10", 1000, 2)])
def test_global_kernel_vs_twin(rl, fits, tag, N, n_outer):
    """k_global_qp against orc_global_mincurv: same formulation, different factorisation (banded +
    border L D L' in LDS vs dense Cholesky), different summation order."""
    if tag == "l10":  # degree-3 spline (the track boundary fit) with synthetic widths
        t, cx, cy, k, L = spline(fits, tag)
        wl = np.full(N, 4.0) + np.sin(np.arange(N) * 0.05)
        wr = np.full(N, 3.0) + np.cos(np.arange(N) * 0.03)
    else:
        t, cx, cy, k, u, wl, wr = monza_widths(fits, tag, N)
    ocx, ocy, oxy, oa, ost = orc.global_mincurv(t, cx, cy, k, N, wl, wr, MARGIN, n_outer)
    trk = rl.lib.Track(rl.lib.Context.get(0), t, cx, cy, k, N)
    ctrl, xy, a, st, rs = rl.ops.global_batch_host(trk, np.stack([wl, wr], 1)[None], MARGIN, n_outer)
    da, dxy = np.abs(a[0] - oa).max(), np.abs(xy[0] - oxy).max()
    print(f"[global gpu {tag} N={N} outer={n_outer}] |da| {da:.2e} m |dxy| {dxy:.2e} m  ipm {int(st[0, 0])}/{int(ost[0])}"
          f"  k2 {st[0, 1]:.5f}->{st[0, 2]:.5f}  viol {st[0, 3]:.1e}  {rs.kernel_ms:.2f} ms  lds {rs.lds_bytes}")
    assert da <= 1e-6 and dxy <= 1e-6
    assert np.abs(ctrl[0] - np.stack([ocx, ocy], 1)).max() <= 1e-6
    assert int(st[0, 0]) == int(ost[0])
    assert st[0, 1] == pytest.approx(ost[1], rel=1e-9) and st[0, 2] == pytest.approx(ost[2], rel=1e-7)
    assert st[0, 3] <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["k3_np56", "oval_np27", "k5_few_knots", "k3_few_knots"])
def test_two_front_factorisation_shapes(rl, shape):
    """The two-front factorisation of k_global_qp2 (n_p <= 64) on the shapes the Monza fixtures do not reach: degree 3
    (half-bandwidth 6, TwoFront<6>), few fronts (oval: n_p = 27 -> 3 front columns + a 21-row middle block) and no front
    at all (n_p <= 2 * bandwidth: the whole matrix is the middle block).  Splines are fitted here (host FITPACK, as
    everywhere); against the CPU twin."""
    from spline_trajectory_optimization_amd.models.trajectory import BSplineTrajectory
    N, n_outer = 1000, 3
    if shape == "oval_np27":
        line = rl.batch.oval_centerline(100.0, 5)
        wl, wr = rl.batch.oval_half_widths(N)
        n_outer = 1   # (the Gauss-Newton step of the first linearisation is 21 m on the oval's straights, where the curvature
        #               vanishes; the second linearisation is singular in the formulation itself -- the twin returns NaN too)
    else:
        centre, _, _ = rl.batch.load_monza()
        s_, k_ = {"k3_np56": (100.0, 3), "k5_few_knots": (1e5, 5), "k3_few_knots": (1e5, 3)}[shape]
        line = BSplineTrajectory(centre, s_, k_)
        wl = np.full(N, 5.0) + np.sin(np.arange(N) * 0.02); wr = np.full(N, 4.5) + np.cos(np.arange(N) * 0.03)
    t, cx, cy, k = line._tck()
    n_p = len(cx) - k
    assert n_p <= 64, n_p
    if shape.endswith("few_knots"):
        assert n_p <= 4 * k, n_p      # no front column: T = 0
    ocx, ocy, oxy, oa, ost = orc.global_mincurv(t, cx, cy, k, N, wl, wr, MARGIN, n_outer)
    trk = rl.lib.Track(rl.lib.Context.get(0), t, cx, cy, k, N)
    ctrl, xy, a, st, rs = rl.ops.global_batch_host(trk, np.stack([wl, wr], 1)[None], MARGIN, n_outer)
    da, dxy = np.abs(a[0] - oa).max(), np.abs(xy[0] - oxy).max()
    print(f"[two-front {shape}: k={k} n_p={n_p}] |da| {da:.2e} m |dxy| {dxy:.2e} m  ipm {int(st[0, 0])}/{int(ost[0])}  "
          f"block {rs.block_threads} lds {rs.lds_bytes}  {rs.kernel_ms:.2f} ms")
    assert rs.block_threads > 64 and rs.block_threads % 64 == 0          # the wave-specialised kernel ran
    assert da <= 1e-6 and dxy <= 1e-6 and st[0, 3] <= 1e-9
    assert abs(int(st[0, 0]) - int(ost[0])) <= 1


@pytest.mark.gpu
def test_global_batch_properties_full_size(rl, fits):
    """BASELINE configs[1] shape (Monza N=2000, 1024 width-perturbed instances): properties that need
    no CPU run -- inside the bounds, cost decreased, bit-reproducible, position-independent -- plus a
    spot check of 3 instances against the twin."""
    t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 2000)
    B = 1024
    W = rl.batch.width_batch(wl, wr, B, seed=1234)
    trk = rl.lib.Track(rl.lib.Context.get(0), t, cx, cy, k, 2000)
    ctrl, xy, a, st, rs = rl.ops.global_batch_host(trk, W, MARGIN, 6)
    assert np.isfinite(xy).all() and np.isfinite(a).all()
    assert (st[:, 3] <= 1e-9).all()
    assert (st[:, 2] < 0.7 * st[:, 1]).all()
    # lateral offsets recomputed on the host from the returned line
    p0 = np.stack([BSpline(t, cx, k)(u), BSpline(t, cy, k)(u)], 1)
    d0 = np.stack([BSpline(t, cx, k)(u, 1), BSpline(t, cy, k)(u, 1)], 1)
    n0 = np.stack([-d0[:, 1], d0[:, 0]], 1) / np.hypot(d0[:, 0], d0[:, 1])[:, None]
    lat = ((xy - p0[None]) * n0[None]).sum(2)
    assert (lat - (W[:, :, 0] - MARGIN)).max() <= 1e-8 and (-(W[:, :, 1] - MARGIN) - lat).max() <= 1e-8
    ctrl2, xy2, a2, st2, _ = rl.ops.global_batch_host(trk, W, MARGIN, 6)
    assert np.array_equal(xy, xy2) and np.array_equal(a, a2) and np.array_equal(st, st2)
    perm = np.random.default_rng(7).permutation(B)
    _, xy3, a3, _, _ = rl.ops.global_batch_host(trk, W[perm], MARGIN, 6)
    assert np.array_equal(xy3, xy[perm]) and np.array_equal(a3, a[perm])
    for b in (0, 511, 1023):
        _, _, oxy, oa, ost = orc.global_mincurv(t, cx, cy, k, 2000, W[b, :, 0], W[b, :, 1], MARGIN, 6)
        assert np.abs(xy[b] - oxy).max() <= 1e-6 and int(st[b, 0]) == int(ost[0])
    print(f"[global batch] {B} instances, {rs.kernel_ms:.2f} ms, {B / rs.kernel_ms * 1e3:.0f} 6-linearisation solves/s, "
          f"ipm iterations {st[:, 0].min():.0f}..{st[:, 0].max():.0f}, sum kappa^2 {st[:, 1].mean():.4f} -> {st[:, 2].mean():.4f}")


@pytest.mark.gpu
def test_global_torch_entry_and_errors(rl, fits):
    import torch
    t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", 500)
    trk = rl.lib.Track(rl.lib.Context.get(0), t, cx, cy, k, 500)
    W = rl.batch.width_batch(wl, wr, 8, seed=3)
    ctrl, xy, a, st, _ = rl.ops.global_batch_host(trk, W, MARGIN, 3)
    out = rl.ops.global_batch_torch(trk, torch.from_numpy(W).cuda(), MARGIN, 3)
    torch.cuda.synchronize()
    assert np.array_equal(out["xy"].cpu().numpy(), xy) and np.array_equal(out["a"].cpu().numpy(), a)
    with pytest.raises(Exception):
        rl.ops.global_batch_host(trk, W[:, :100], MARGIN, 3)
    with pytest.raises(rl.lib.RlError):
        rl.ops.global_batch_host(trk, W, MARGIN, -1)
