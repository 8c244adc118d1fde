"""The CPU oracle (oracle/mincurv_oracle.c) against fixtures produced by the reference's own
Python functions (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import golden, spline
from oracle import oracle as orc


def test_g1_periodic_coefficients(fits):
    # models/trajectory.py:219-222: splprep(per=True) -> last k coefficients repeat the first k
    for tag in ("c100", "c30", "c0p8", "l10", "r10"):
        t, cx, cy, k, length = spline(fits, tag)
        assert len(t) == len(cx) + k + 1
        np.testing.assert_allclose(cx[-k:], cx[:k], rtol=0, atol=1e-9)
        np.testing.assert_allclose(cy[-k:], cy[:k], rtol=0, atol=1e-9)
        assert t[k] == 0.0 and t[len(cx)] == 1.0
        assert 5700 < length < 5900


@pytest.mark.parametrize("N", [500, 2000])
def test_g2_sample_along(fits, N):
    t, cx, cy, k, length = spline(fits, "c100")
    g = golden("G2_sample_along.npz")
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    ref = g[f"N{N}_cols"]
    cols = g["cols"]
    # x, y, yaw: same recurrences as scipy -> essentially exact
    np.testing.assert_allclose(pts[:, cols[:3]], ref[:, :3], rtol=0, atol=1e-12)
    # turn radius is 1/|kappa|: relative
    np.testing.assert_allclose(pts[:, 5], ref[:, 3], rtol=1e-12)
    # cumulative arc length: the oracle takes ONE GK21 panel per segment.  scipy.integrate.quad
    # (QUADPACK qagse, epsrel 1.49e-8) accepts that first panel on every 2.9 m segment of the
    # N=2000 grid (bitwise equal); on the 11.6 m segments of the N=500 grid it bisects a few
    # knot-straddling panels, which moves the 5.8 km cumulative sum by ~1e-8 m.
    atol = 1e-9 if N >= 2000 else 1e-7
    np.testing.assert_allclose(pts[:, 6], ref[:, 4], rtol=0, atol=atol)
    np.testing.assert_allclose(pts[:, 7], ref[:, 5], rtol=0, atol=atol)
    assert np.all(pts[:, 17] == np.arange(N)) and np.all(pts[:, 18] == -1)


def test_g2_sample_along_4000(fits):
    t, cx, cy, k, length = spline(fits, "c100")
    g = golden("G2_sample_along.npz")
    u = np.linspace(0.0, 1.0, 4000, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)[:, g["cols"]]
    np.testing.assert_allclose(pts[::16, :3], g["N4000_stride16_cols"][:, :3], rtol=0, atol=1e-12)
    np.testing.assert_allclose(pts[:, [0, 1, 2, 4, 5]].sum(axis=0), g["N4000_colsum"][[0, 1, 2, 4, 5]], rtol=1e-12)


@pytest.mark.parametrize("tag", ["c100", "c30"])
@pytest.mark.parametrize("N", [500, 2000])
def test_g3_min_curvature_cost(fits, tag, N):
    t, cx, cy, k, _ = spline(fits, tag)
    g = golden("G3_min_curvature_cost.npz")
    Href, gref, Mref = g[f"{tag}_N{N}_H"], g[f"{tag}_N{N}_g"], g[f"{tag}_N{N}_M"]
    n = len(cx)
    for j, idx in enumerate(range(2, n - 3)):
        H, gg, M = orc.min_curvature_cost([cx[idx], cy[idx]], idx, t, cx, cy, k, N)
        assert M == Mref[j]
        np.testing.assert_allclose(H, Href[j], rtol=1e-11, atol=0)
        # g is a sum with cancellation: compare against the scale of its terms
        scale = np.abs(gref[j]).max() + 1e-30
        np.testing.assert_allclose(gg, gref[j], rtol=0, atol=1e-9 * scale + 1e-16)
        assert H[0, 1] == 0.0 and H[1, 0] == 0.0  # optimizer.py:82-83


@pytest.mark.parametrize("tag", ["c100", "c30"])
def test_g4_track_constraint(fits, rings, tag):
    t, cx, cy, k, length = spline(fits, tag)
    N = 500
    g = golden("G4_track_constraint.npz")
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    np.testing.assert_allclose(pts[:, [0, 1, 3]], g[f"{tag}_N{N}_xyyaw"], rtol=0, atol=1e-12)
    orc.fill_bounds(pts, rings[0], rings[1], 100.0)
    # same C code made the fixture, but from numpy's arctan2 yaw (1 ulp from libm's in places)
    np.testing.assert_allclose(pts[:, 9:13], g[f"{tag}_N{N}_bounds"], rtol=0, atol=1e-11)
    off = g[f"{tag}_N{N}_off"]
    n = len(cx)
    for j, idx in enumerate(range(2, n - 3)):
        A, lba, uba = orc.track_constraint(idx, t, cx, cy, k, pts)
        sl = slice(off[j], off[j + 1])
        assert len(lba) == off[j + 1] - off[j]
        np.testing.assert_allclose(lba, g[f"{tag}_N{N}_lba"][sl], rtol=0, atol=1e-11)
        np.testing.assert_allclose(uba, g[f"{tag}_N{N}_uba"][sl], rtol=0, atol=1e-11)
        np.testing.assert_allclose(A[0::2, 0], g[f"{tag}_N{N}_b"][off[j] // 2:off[j + 1] // 2], rtol=0, atol=1e-15)
        assert np.array_equal(A[0::2, 0], A[1::2, 1]) and not A[0::2, 1].any() and not A[1::2, 0].any()


def test_fill_bounds_geometry(fits, rings):
    """fill_bounds restates shapely semantics (trajectory.py:83-141): every bound point lies on its
    ring, on the waypoint's normal, and no closer crossing exists (checked by brute force in numpy)."""
    t, cx, cy, k, length = spline(fits, "c100")
    N = 200
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    orc.fill_bounds(pts, rings[0], rings[1], 100.0)
    for ring, (bx, by), sgn in ((rings[0], (9, 10), +1.0), (rings[1], (11, 12), -1.0)):
        P = ring
        Q = np.roll(ring, -1, axis=0)
        for i in range(N):
            p = pts[i, :2]
            b = pts[i, [bx, by]]
            d = b - p
            dist = np.hypot(*d)
            assert 1.0 < dist < 12.0  # Monza half widths (2.6 .. 7.7 m)
            yaw = pts[i, 3]
            nrm = np.array([np.cos(yaw + sgn * np.pi / 2), np.sin(yaw + sgn * np.pi / 2)])
            assert abs(d @ nrm - dist) < 1e-9                       # on the normal, correct side
            # distance from b to the polyline is ~0
            s = Q - P
            w = b - P
            tt = np.clip((w * s).sum(1) / (s * s).sum(1), 0, 1)
            assert np.min(np.hypot(*(w - tt[:, None] * s).T)) < 1e-9
            # brute-force closest crossing of the +-100 m normal segment
            a0 = p - 100 * nrm
            r = 200 * nrm
            den = r[0] * s[:, 1] - r[1] * s[:, 0]
            with np.errstate(all="ignore"):
                tpar = ((P[:, 0] - a0[0]) * s[:, 1] - (P[:, 1] - a0[1]) * s[:, 0]) / den
                upar = ((P[:, 0] - a0[0]) * r[1] - (P[:, 1] - a0[1]) * r[0]) / den
            ok = (den != 0) & (tpar >= 0) & (tpar <= 1) & (upar >= 0) & (upar <= 1)
            hits = a0 + tpar[ok, None] * r
            assert abs(np.min(np.hypot(*(hits - p).T)) - dist) < 1e-9


def test_qp_closed_form_kkt(fits, rings):
    """KKT check of the exact separable QP solve (stands in for casadi.conic/qpOASES,
    optimizer.py:268-277): stationarity, primal feasibility, complementarity."""
    t, cx, cy, k, length = spline(fits, "c100")
    N = 500
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    orc.fill_bounds(pts, rings[0], rings[1], 100.0)
    n = len(cx)
    n_active = 0
    for idx in range(2, n - 3):
        H, g, M = orc.min_curvature_cost([cx[idx], cy[idx]], idx, t, cx, cy, k, N)
        A, lba, uba = orc.track_constraint(idx, t, cx, cy, k, pts)
        st, x = orc.qp_solve_separable(H, g, A, lba, uba)
        assert st == 0
        Ax = A @ x
        tol = 1e-9
        assert np.all(Ax >= lba - tol) and np.all(Ax <= uba + tol)
        grad = H @ x + g
        for c in range(2):
            rows = np.arange(c, 2 * M, 2)
            a = A[rows, c]
            act_lo = np.abs(Ax[rows] - lba[rows]) < 1e-9
            act_hi = np.abs(Ax[rows] - uba[rows]) < 1e-9
            gs = abs(g[c]) + abs(H[c, c] * x[c])
            if abs(grad[c]) <= 1e-9 * gs:
                continue  # unconstrained stationary point
            n_active += 1
            # gradient must be balanced by a multiplier of the right sign on an active row
            if grad[c] > 0:   # wants to decrease x_c -> a lower row (a>0) must be active
                assert np.any(act_lo & (a > 0))
            else:
                assert np.any(act_hi & (a > 0))
    assert n_active > 10  # Monza: most control points end on a bound


def test_qp_infeasible_and_degenerate():
    H = np.diag([2.0, 3.0]); g = np.array([1.0, -1.0])
    A = np.array([[1.0, 0], [0, 1.0], [0.5, 0], [0, 0.5]])
    st, x = orc.qp_solve_separable(H, g, A, np.array([0.0, 0, 1.0, 0]), np.array([1.0, 1, 2.0, 1]))
    assert st == 2                      # x0 in [0,1] and 0.5 x0 in [1,2] -> empty
    st, x = orc.qp_solve_separable(H, g, A, np.array([-1.0, -1, -1, -1]), np.array([1.0, 1, 1, 1]))
    assert st == 0 and np.allclose(x, [-0.5, 1.0 / 3.0])
    A0 = np.array([[0.0, 0], [0, 1.0]])
    st, _ = orc.qp_solve_separable(H, g, A0, np.array([0.5, -1]), np.array([1.0, 1]))
    assert st == 2                      # 0 * x in [0.5, 1] is infeasible
    st, _ = orc.qp_solve_separable(np.diag([0.0, 1.0]), g, A, -np.ones(4), np.ones(4))
    assert st == 3


def test_g7_run_min_curvature_qp(fits, rings):
    """End to end: the reference's run_min_curvature_qp driver (optimizer.py:256-341) with the sweep
    order pinned.  north_star tolerance: 1e-4 m; we hold the oracle to 1e-6 m."""
    g = golden("G7_run_min_curvature_qp.npz")
    for key in g["cases"]:
        key = str(key)
        tag, Ns, its, _ = key.split("_")
        N, max_iter = int(Ns[1:]), int(its[2:])
        if N > 300:
            continue  # covered by the slow test below
        t, cx, cy, k, length = spline(fits, tag)
        ocx, ocy, pts, ns = orc.run_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1],
                                                    g[f"{key}_i_start"], max_iter)
        np.testing.assert_array_equal(ns, g[f"{key}_n_success"])
        dev = np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max()
        assert dev < 1e-6, (key, dev)
        assert np.hypot(ocx - cx, ocy - cy).max() > 1.0  # the line really moved


@pytest.mark.slow
def test_g7_run_min_curvature_qp_large(fits, rings):
    """N = 500, the benchmark size N = 2000 and the reference test's own N = 1929 (sample_along(3.0))."""
    g = golden("G7_run_min_curvature_qp.npz")
    keys = [str(k) for k in g["cases"] if int(str(k).split("_")[1][1:]) > 300]
    if not keys:
        pytest.skip("large fixtures not generated")
    for key in keys:
        tag, Ns, its, _ = key.split("_")
        N, max_iter = int(Ns[1:]), int(its[2:])
        t, cx, cy, k, length = spline(fits, tag)
        ocx, ocy, pts, ns = orc.run_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1],
                                                    g[f"{key}_i_start"], max_iter)
        np.testing.assert_array_equal(ns, g[f"{key}_n_success"])
        dev = np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max()
        print(key, "oracle vs reference run [m]:", dev)
        assert dev < 1e-6


def g7b_rings(g, key, t, cx, cy, k, N, rings):
    """Rings of a G7b case: the real Monza rings, or those of an instance of bench.py's width-perturbed batch."""
    if f"{key}_widths" in g.files:
        return orc.width_rings(t, cx, cy, k, N, g[f"{key}_widths"])
    return rings


@pytest.mark.slow
def test_g7b_benchmarked_configuration(fits, rings):
    """Fixture G7b: the reference's OWN run_min_curvature_qp loop (optimizer.py:256-341) at the benchmarked configuration
    -- N = 2000 (and the reference test's N = 1929), max_iter = 5, three runs on the real Monza rings and instances 0 and 3
    of bench.py's width-perturbed batch in bench.py's sweep order.  The oracle reproduces every one of them: same
    per-pass success counts, lines within 1e-5 m (measured 2e-11 .. 4e-6 m)."""
    from concurrent.futures import ThreadPoolExecutor
    g = golden("G7b_benchmarked_config.npz")
    _check_reference_runs(g, fits, rings, n_cases=5, n_bench=2)


@pytest.mark.slow
def test_g7d_benchmarked_batch_sample(fits, rings):
    """Fixture G7d (round 5): the reference's own loop for instances 1, 2, 4, 5, 6, 7 of bench.py's batch -- with G7b's 0 and 3
    the whole sample bench.py lays beside the oracle.  The oracle reproduces all six lines to <= 1.8e-6 m with the reference's
    success counts in every pass but the last, where on two of the six one QP is decided the other way (numpy's matrix products
    sum in another order than the oracle's loop: the control point then differs in its last bits, DESIGN_HISTORY.md section 5)."""
    _check_reference_runs(golden("G7d_benchmarked_batch_sample.npz"), fits, rings, n_cases=6, n_bench=6)


def _check_reference_runs(g, fits, rings, n_cases, n_bench):
    from concurrent.futures import ThreadPoolExecutor
    t, cx, cy, k, length = spline(fits, "c100")

    def run(key):
        N = int(key.split("_")[1][1:])
        rl_, rr_ = g7b_rings(g, key, t, cx, cy, k, N, rings)
        ocx, ocy, pts, ns = orc.run_min_curvature_qp(t, cx, cy, k, length, N, rl_, rr_, g[f"{key}_i_start"])
        return key, ns, float(np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max())
    keys = [str(k_) for k_ in g["cases"]]
    assert len(keys) == n_cases and sum("bench" in k_ for k_ in keys) == n_bench
    with ThreadPoolExecutor(len(keys)) as ex:       # ctypes releases the GIL: one case per thread
        for key, ns, dev in ex.map(run, keys):
            print(key, "oracle vs the reference's run [m]:", dev, "successes", ns.ravel().tolist())
            if n_cases == 5:     # G7b: identical per-pass success counts on all five
                np.testing.assert_array_equal(ns, g[f"{key}_n_success"])
            else:                # G7d: on two of six (bench1, bench6) ONE QP of the last backward pass flips in the noise
                ref_ns = g[f"{key}_n_success"]
                np.testing.assert_array_equal(ns.ravel()[:-1], ref_ns.ravel()[:-1])
                assert abs(int(ns.ravel()[-1]) - int(ref_ns.ravel()[-1])) <= 1
            assert len(g[f"{key}_i_start"]) == 5 and dev < 1e-5, (key, dev)


def test_g12_numpy_raise_semantics():
    """Fixture G12: the reference's two drivers run under np.seterr(all='raise') -- which the simulator switches on for
    the whole process (simulator.py:164) and the reference's own test does before it calls run_min_curvature_qp
    (tests/test_optimizer.py:35-37) -- on a track with a run of control points exactly on the diagonal y = x: there
    x' y'' - y' x'' == 0 exactly, sample_along raises in 1 / |curvature| (trajectory.py:253-260) INSIDE the try block,
    after set_control_point has written the new control points: the step fails, the spline keeps the update, the table
    stays stale (optimizer.py:276-293, :197-214).  The oracle models that state (orc_set_numpy_raise) and reproduces the
    reference's run: same per-pass success counts, same number of raising windows, same line; without the model it does
    not."""
    g = golden("G12_numpy_raise_semantics.npz")
    t, cx, cy, k, length = g["t"], g["cx"], g["cy"], int(g["k"]), float(g["length"])
    ocx, ocy, _, ns = orc.run_min_curvature_qp(t, cx, cy, k, length, 300, g["ringL"], g["ringR"], g["sweep_N300_i_start"],
                                               numpy_raise=True)
    np.testing.assert_array_equal(ns, g["sweep_N300_n_success"])
    assert orc.last_raised() > 0 and ns[0, 0] < ns[0, 1]          # the stale phase is the start of the first pass
    dev = float(np.hypot(ocx - g["sweep_N300_cx"], ocy - g["sweep_N300_cy"]).max())
    print("G12 sweep: oracle vs the reference's run [m]:", dev, "steps whose re-sampling raised:", orc.last_raised())
    assert dev < 1e-5
    plain = orc.run_min_curvature_qp(t, cx, cy, k, length, 300, g["ringL"], g["ringR"], g["sweep_N300_i_start"])
    assert not np.array_equal(plain[3], g["sweep_N300_n_success"])
    assert np.hypot(plain[0] - g["sweep_N300_cx"], plain[1] - g["sweep_N300_cy"]).max() > 1.0
    jcx, jcy, _, jns = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, 240, g["ringL"], g["ringR"], g["joint_N240_i_start"],
                                                      numpy_raise=True)
    assert orc.last_raised() == int(g["joint_N240_n_raised"]) and int(jns.sum()) == int(g["joint_N240_n_ok"])
    jdev = float(np.hypot(jcx - g["joint_N240_cx"], jcy - g["joint_N240_cy"]).max())
    print("G12 sliding window: oracle vs the reference's run [m]:", jdev, "windows whose re-sampling raised:", orc.last_raised())
    assert jdev < 1e-6


def test_g6_qss_simulator():
    """The oracle's port of Simulator.run_simulation against the reference's own run (fixture G6)."""
    from scipy.interpolate import CubicSpline
    g = golden("G6_simulator.npz")
    N = len(g["speed"])
    pts = np.zeros((N, 19))
    pts[:, [0, 1, 5, 6, 7, 13]] = g["cols_in"]
    pts[:, 17] = np.arange(N); pts[:, 18] = -1
    acc = CubicSpline(g["acc_lookup"][:, 0], g["acc_lookup"][:, 1])
    dcc = CubicSpline(g["dcc_lookup"][:, 0], g["dcc_lookup"][:, 1])
    out, it = orc.qss_sim(pts, acc.x, acc.c, dcc.x, dcc.c, g["params"])
    assert it > 10
    np.testing.assert_array_equal(out[:, 18], g["iter_flag"])
    np.testing.assert_array_equal(out[:, 4], g["speed"])
    np.testing.assert_allclose(out[:, 14], g["lon_acc"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[:, 15], g["lat_acc"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(out[:, 16], g["time"], rtol=0, atol=1e-15)
    # a zero speed makes the reference raise (np.seterr(all='raise'), simulator.py:164-165)
    bad = pts.copy(); bad[7, 5] = 0.0          # zero turn radius -> seed speed 0
    _, it2 = orc.qss_sim(bad, acc.x, acc.c, dcc.x, dcc.c, g["params"])
    assert it2 == -1


def test_qp_diag_rows_kkt_and_feasibility():
    """The Goldfarb-Idnani solver behind the window QP of run_joint_min_curvature_qp
    (optimizer.py:188-197): KKT conditions on random problems, and its infeasibility verdicts against an
    LP feasibility solve (scipy HiGHS)."""
    from scipy.optimize import linprog
    rng = np.random.default_rng(0)
    n_inf = 0
    for trial in range(400):
        nv = int(rng.integers(1, 6)); M = int(rng.integers(1, 40))
        h = rng.uniform(0.1, 5, nv); g = rng.normal(0, 3, nv)
        A = rng.normal(0, 1, (M, nv)) * (rng.random((M, nv)) < 0.6)
        c = A @ rng.normal(0, 1, nv)
        w = rng.uniform(0, 1.0, M) * (rng.random(M) < 0.9)
        l = c - w * rng.random(M); u = c + w * rng.random(M)
        if trial % 7 == 0:
            l = l + rng.normal(0, 2, M); u = np.maximum(u, l + rng.random(M) * 0.2)
        st, x, lam = orc.qp_diag_rows(h, g, A, l, u)
        res = linprog(np.zeros(nv), A_ub=np.vstack([A, -A]), b_ub=np.concatenate([u, -l]), bounds=[(None, None)] * nv)
        assert (res.status == 0) == (st == 0), trial
        if st != 0:
            assert st == 2
            n_inf += 1
            continue
        Ax = A @ x
        assert np.all(Ax >= l - 1e-8) and np.all(Ax <= u + 1e-8)
        assert np.abs(h * x + g - A.T @ lam).max() <= 1e-8 * (1 + np.abs(g).max())      # stationarity
        assert np.all(np.abs(Ax - l)[lam > 1e-12] < 1e-7) and np.all(np.abs(Ax - u)[lam < -1e-12] < 1e-7)
    assert n_inf > 10


def test_g5_joint_assembly(fits, rings):
    """joint_min_curvature_cost / joint_track_constraint (optimizer.py:88-161) as the oracle's joint
    driver assembles them, against the reference's own matrices (fixture G5)."""
    t, cx, cy, k, length = spline(fits, "c100")
    N = 500
    g5 = golden("G5_joint.npz")
    u = np.linspace(0.0, 1.0, N, endpoint=False)
    pts = orc.sample_along(t, cx, cy, k, length, u)
    orc.fill_bounds(pts, rings[0], rings[1], 100.0)
    for st in g5["starts"]:
        st = int(st)
        Hd = []; gg = []
        A = np.zeros((2 * N, 10)); z = np.zeros(10)
        for j in range(5):
            H, g_, M = orc.min_curvature_cost([cx[st + j], cy[st + j]], st + j, t, cx, cy, k, N)
            Hd += [H[0, 0], H[1, 1]]; gg += list(g_)
            Aj, _, _ = orc.track_constraint(st + j, t, cx, cy, k, pts)
            m = len(Aj) // 2
            mask = (u >= t[st + j]) & (u < t[st + j + k + 1])
            s0 = int(np.argmax(mask))
            A[2 * s0:2 * (s0 + m):2, 2 * j] = Aj[0::2, 0]; A[2 * s0 + 1:2 * (s0 + m):2, 2 * j + 1] = Aj[1::2, 1]
            z[2 * j:2 * j + 2] = (cx[st + j], cy[st + j])
        np.testing.assert_allclose(Hd, g5[f"s{st}_Hdiag"], rtol=1e-10)
        assert np.all(np.abs(np.array(gg) - g5[f"s{st}_g"]) <= 1e-8 * np.abs(g5[f"s{st}_g"]).max())
        ref_A = np.zeros_like(A); ref_A[g5[f"s{st}_Arow"], g5[f"s{st}_Acol"]] = g5[f"s{st}_Aval"]
        np.testing.assert_allclose(A, ref_A, rtol=0, atol=1e-14)
        non_z = pts[:, :2] - (A @ z).reshape(-1, 2)
        lba = np.empty(2 * N); uba = np.empty(2 * N)
        lba[0::2] = np.minimum(pts[:, 9], pts[:, 11]) - non_z[:, 0]; lba[1::2] = np.minimum(pts[:, 10], pts[:, 12]) - non_z[:, 1]
        uba[0::2] = np.maximum(pts[:, 9], pts[:, 11]) - non_z[:, 0]; uba[1::2] = np.maximum(pts[:, 10], pts[:, 12]) - non_z[:, 1]
        np.testing.assert_allclose(lba, g5[f"s{st}_lba"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(uba, g5[f"s{st}_uba"], rtol=0, atol=1e-9)


def test_g8_double_track_checker_vs_reference_functions():
    """oracle/dt_checker.py against fixture G8 = the reference's OWN models/double_track.py (dynamics,
    add_constraints), utils/utils.py (align_yaw, align_abscissa) and min_time_optimizer.py
    (set_up_double_track_problem: pairing of node i-1 with node i, variable scaling, cost) executed on
    numbers through a numeric stand-in for CasADi (tests/golden/make_golden.py)."""
    from oracle import dt_checker as dc
    g = golden("G8_double_track.npz")
    m = dict(zip(g["model_keys"].tolist(), g["model_vals"].tolist()))
    xd, (Fx, Fy, Fz) = dc.dynamics(m, g["dyn_X"], g["dyn_U"], g["dyn_k"])
    np.testing.assert_allclose(xd, g["dyn_xdot_frenet"], rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(np.stack([Fx, Fy, Fz], axis=1), g["dyn_tyre_forces"], rtol=1e-13, atol=1e-10)
    xd0, _ = dc.dynamics(m, g["dyn_X"], g["dyn_U"], np.zeros(len(g["dyn_k"])))   # race_track=None: no curvilinear terms
    np.testing.assert_allclose(xd0, g["dyn_xdot_plain"], rtol=1e-13, atol=1e-12)
    eq, ineq, cost = dc.eval_nodes(m, g["nlp_s"], g["nlp_kappa"], g["nlp_left"], g["nlp_right"], float(g["nlp_margin"]),
                                   float(g["nlp_length"]), g["nlp_X"][None], g["nlp_U"][None], g["nlp_T"][None])
    np.testing.assert_allclose(eq[0], g["nlp_eq"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ineq[0], g["nlp_ineq"], rtol=1e-13, atol=1e-12)
    assert abs(cost[0] - float(g["nlp_cost"])) < 1e-12 * abs(cost[0])
    assert abs(g["nlp_eq"][5, 7] - 0.3) < 1e-12 and np.abs(g["nlp_eq"][:, :6]).max() > 1e-3   # a point off the constraints


def test_g9_run_joint_min_curvature_qp(fits, rings):
    """The oracle's sliding-window driver against fixture G9 = the reference's own
    run_joint_min_curvature_qp loop (optimizer.py:163-220; window QPs solved by the oracle's dual active-set
    solver standing in for casadi.conic/qpOASES, simulator called per window as in the reference).  On the
    cases flagged well-conditioned the lines agree to 1e-6 m with equal window counts; the chaotic case is
    only required to be explained by the oracle's own re-roundings."""
    g = golden("G9_run_joint_min_curvature_qp.npz")
    t, cx, cy, k, length = spline(fits, "c100")
    n_good = 0
    for key in g["cases"]:
        key = str(key)
        N = int(key.split("_")[1][1:])
        ist = g[f"{key}_i_start"]
        ocx, ocy, _, ons = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], ist)
        dev = np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max()
        if bool(g[f"{key}_oracle_reproduces_run"]):
            assert dev < 1e-6 and int(ons.sum()) == int(g[f"{key}_n_ok"]), (key, dev, ons)
            n_good += 1
        else:
            spread = 0.0
            for seed in range(1, 7):
                rcx, rcy, _, _ = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], ist, rerounding=seed)
                spread = max(spread, np.hypot(rcx - ocx, rcy - ocy).max())
            # certified chaotic: the oracle's own re-roundings part by more than the tolerance, so neither the oracle nor
            # anything else can be required to reproduce this run (no multiple of the spread is accepted as "agreement")
            assert spread > 1e-4, (key, dev, spread)
    assert n_good >= 5


def test_g9b_g7c_wellconditioned_reference_runs(fits, rings):
    """Fixtures G9b / G7c (round 4): the reference's own run_joint_min_curvature_qp / run_min_curvature_qp loops on cases
    where the oracle's 24 re-roundings + FMA build agree to < 1e-6 m (recorded per case in the fixture).  The strict
    oracle must reproduce every one of these runs to 1e-6 m with equal window / success counts -- no exemption exists
    on a well-conditioned case."""
    g = golden("G9b_joint_wellconditioned.npz")
    assert len(g["cases"]) >= 8
    for key in g["cases"]:
        key = str(key)
        tag, Ns, _ = key.split("_")
        N = int(Ns[1:])
        t, cx, cy, k, length = spline(fits, tag)
        assert float(g[f"{key}_oracle_spread_m"]) < 1e-6
        ocx, ocy, _, ons = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], g[f"{key}_i_start"])
        dev = np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max()
        assert dev < 1e-6 and int(ons.sum()) == int(g[f"{key}_n_ok"]) > 0, (key, dev, ons)
        # and the premise, re-checked on a few re-roundings so that it cannot rot
        for seed in (1, 2, 3):
            rcx, rcy, _, _ = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], g[f"{key}_i_start"], rerounding=seed)
            assert np.hypot(rcx - ocx, rcy - ocy).max() < 1e-6, (key, seed)
    g = golden("G7c_wellconditioned.npz")
    assert len(g["cases"]) >= 4
    for key in g["cases"]:
        key = str(key)
        tag, Ns, _ = key.split("_")
        N = int(Ns[1:])
        t, cx, cy, k, length = spline(fits, tag)
        assert float(g[f"{key}_oracle_spread_m"]) < 1e-6
        ocx, ocy, _, ons = orc.run_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], g[f"{key}_i_start"])
        dev = np.hypot(ocx - g[f"{key}_cx"], ocy - g[f"{key}_cy"]).max()
        assert dev < 1e-6, (key, dev)
        np.testing.assert_array_equal(ons, g[f"{key}_n_success"])


def test_replay_of_the_first_steps_equals_the_plain_pipeline(fits, rings):
    """orc.replay_steps (the teacher-forced single-step re-derivation used by tests/test_sweep_replay.py) on
    the initial control points equals cost + constraint + QP computed through the plain wrappers."""
    t, cx, cy, k, length = spline(fits, "c100")
    N = 400
    idx = np.arange(2, len(cx) - 3, dtype=np.int32)
    rep = orc.replay_steps(t, k, N, rings[0], rings[1], idx, np.tile(cx, (len(idx), 1)), np.tile(cy, (len(idx), 1)), nthreads=4)
    pts = orc.sample_along(t, cx, cy, k, length, np.linspace(0, 1, N, endpoint=False))
    orc.fill_bounds(pts, rings[0], rings[1])
    for j, i in enumerate(idx):
        H, gg, M = orc.min_curvature_cost(np.array([cx[i], cy[i]]), int(i), t, cx, cy, k, N)
        A, l, u = orc.track_constraint(int(i), t, cx, cy, k, pts)
        st, x = orc.qp_solve_separable(H, gg, A, l, u)
        assert st == rep[j, 0] and M == rep[j, 16]
        assert H[0, 0] == rep[j, 1] and H[1, 1] == rep[j, 2] and gg[0] == rep[j, 3] and gg[1] == rep[j, 4]
        if st == 0:
            assert x[0] == rep[j, 9] and x[1] == rep[j, 10]
            assert rep[j, 5] <= x[0] <= rep[j, 6] and rep[j, 7] <= x[1] <= rep[j, 8]
    assert (rep[:, 11:15] > 0).all() and (rep[:, 11:15] < 1.0).all()


def test_joint_window_replay_instrument(fits, rings):
    """orc_replay_joint_windows (the teacher-forced instrument of tests/test_hip_parity.py::test_joint_windows_teacher_forced)
    pinned on the CPU: fed window by window with ITS OWN state it must reproduce orc_run_joint_min_curvature_qp bit for
    bit; relaxing / tightening every bound by eps_b can only make a window easier / harder, and the optimum of the
    window as assembled lies between the two."""
    t, cx, cy, k, length = spline(fits, "c100")
    n, N, ist = len(cx), 200, [28]
    j_max, i_min = n - 3 - 5, 2
    ox, oy = cx.copy(), cy.copy()
    n_ok = n_noise = 0
    for stp in range(j_max - i_min):
        kk = stp + ist[0]
        if kk >= j_max:
            kk = kk - j_max + i_min
        head, rows = orc.replay_joint_windows(t, k, N, rings[0], rings[1], [kk], ox[None], oy[None], nthreads=1)
        o = head[0]
        strict_ok, relaxed_ok, tight_ok = (o[0] == 0.0 and o[1] == 0.0), o[35] == 1.0, o[36] == 1.0
        assert (not tight_ok or strict_ok) and (not strict_ok or relaxed_ok), (stp, o[:2], o[35:37])
        n_noise += relaxed_ok and not tight_ok
        if strict_ok:
            fscale = 1e-12 + abs(o[39])
            assert o[37] <= o[39] + 1e-9 * fscale and (not tight_ok or o[39] <= o[38] + 1e-9 * fscale), (stp, o[37:40])
            ox[kk:kk + 5], oy[kk:kk + 5] = o[5:10], o[10:15]
            ox[0], oy[0] = ox[n - 5], oy[n - 5]; ox[1], oy[1] = ox[n - 4], oy[n - 4]
            ox[n - 3], oy[n - 3] = ox[2], oy[2]; ox[n - 2], oy[n - 2] = ox[3], oy[3]; ox[n - 1], oy[n - 1] = ox[4], oy[4]
            n_ok += 1
        assert rows.shape == (1, N, 9) and int(o[3]) < int(o[4]) <= N
    fcx, fcy, _, fns = orc.run_joint_min_curvature_qp(t, cx, cy, k, length, N, rings[0], rings[1], ist)
    np.testing.assert_array_equal(fcx, ox); np.testing.assert_array_equal(fcy, oy)
    assert int(fns.sum()) == n_ok and n_ok > 5
    print(f"joint replay instrument: {n_ok} windows accepted, {n_noise} windows whose verdict an eps_b change of the bounds reverses")
