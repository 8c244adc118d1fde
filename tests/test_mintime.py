"""f4 / BASELINE config 5: the min-time double-track NLP solve.

CPU: the twin (oracle/sqp_twin.py) -- its pair functions against the G8-pinned checker, convergence on the
reference's example track (MGKT kart circuit) from the QSS warm start, KKT conditions of the result.
GPU: rl_mintime_solve_batch against the twin (solution, iteration count), on a batch with per-instance
track widths, and its argument checks."""
import numpy as np
import pytest

from mintime_problem import mgkt_problem
from oracle import dt_checker as dc
from oracle import sqp_twin as tw
from spline_trajectory_optimization_amd.min_time_optm import defaults


def _problem(interval):
    d = mgkt_problem(interval, defaults.ESTIMATES)
    P = tw.Problem(defaults.MODEL, d["s"], d["kappa"], d["left"], d["right"], d["L"],
                   defaults.SOLVER["average_track_width"], defaults.SOLVER["speed_cap"])
    return d, P, tw.initial_point(P, d["speed"], d["seg_time"])


@pytest.fixture(scope="module")
def coarse():
    return _problem(8.0)


def test_twin_pair_functions_match_the_pinned_checker(coarse):
    d, P, w0 = coarse
    rng = np.random.default_rng(0)
    w = w0 + 0.05 * rng.normal(size=w0.shape)
    eq, g = P.functions(w)
    X, U, T = P.unpack(w)
    oeq, og, _ = dc.eval_nodes(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X[None], U[None], T[None])
    np.testing.assert_allclose(eq / P.se, oeq[0, :, :7], rtol=0, atol=1e-11)
    assert np.abs(oeq[0, :, 7]).max() == 0.0                                  # the abscissa pin holds identically
    sx, su, m = P.scale_x, P.scale_u, P.m
    np.testing.assert_allclose(g[:, :4], og[0, :, :4], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g[:, 4] * m["Pmax"], og[0, :, 4], rtol=0, atol=1e-8)
    np.testing.assert_allclose(g[:, 5] * sx[5], og[0, :, 5], rtol=0, atol=1e-12)
    np.testing.assert_allclose(g[:, 6:8] * su[0], og[0, :, 6:8], rtol=0, atol=1e-9)
    np.testing.assert_allclose(g[:, 8:10] * su[2], og[0, :, 8:10], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.maximum(g[:, 10], g[:, 11]) * su[0], og[0, :, 10], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.maximum(g[:, 12], g[:, 13]) * su[2], og[0, :, 11], rtol=0, atol=1e-11)
    np.testing.assert_allclose(g[:, 14:16] * sx[1], og[0, :, 12:14], rtol=0, atol=1e-12)
    # derivatives: complex-step Jacobian against central differences
    Jo, Jn = P.jacobians(w)
    h = 1e-6
    for a in (0, 4, 5, 8):
        wp, wm = w.copy(), w.copy(); wp[:, a] += h; wm[:, a] -= h
        fp = np.concatenate(P.functions(wp), axis=1); fm = np.concatenate(P.functions(wm), axis=1)
        fd = (fp - fm) / (2 * h)
        # pair j depends on w_j[a] (own) and on w_{j+1}[a] (next): perturbing every node at once adds both
        np.testing.assert_allclose(fd, Jo[:, :, a] + Jn[:, :, a], rtol=1e-5, atol=1e-6)


def test_twin_converges_from_the_qss_warm_start(coarse):
    d, P, w0 = coarse
    w, info = tw.solve(P, w0, max_iter=150, tol=1e-6)
    assert info["status"] == 1, {k: v for k, v in info.items() if k in ("iterations", "kkt", "viol", "compl", "status")}
    assert info["lap_time"] < d["qss_lap"] - 1.0                      # faster than the quasi-steady-state profile
    eq, g = P.functions(w)
    assert np.abs(eq).max() <= 1e-6 and g.max() <= 1e-6               # feasible
    assert info["z"].min() >= 0.0 and (info["z"] * info["s"]).max() <= 1e-6
    X, U, T = P.unpack(w)
    assert X[:, 5].min() >= 1.0 - 1e-6 and T.min() > 0.0
    print(f"twin: N={P.N} iterations {info['iterations']} lap {info['lap_time']:.4f} s (QSS {d['qss_lap']:.4f} s)")


@pytest.mark.gpu
def test_mintime_solve_vs_twin(coarse):
    """rl_mintime_solve_batch against the CPU twin: same iteration (k_mt_derivs / k_mt_kkt / k_mt_step),
    independent derivatives and linear algebra.  First iteration-exact on short runs, then the converged
    solution."""
    from spline_trajectory_optimization_amd import ops
    d, P, w0 = coarse
    X0, U0, T0 = P.unpack(w0)
    for iters in (1, 3):
        w, info = tw.solve(P, w0, max_iter=iters, tol=1e-12)
        X, U, T, st = ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X0[None], U0[None], T0[None],
                                              max_iter=iters, tol=1e-12)
        Xt, Ut, Tt = P.unpack(w)
        dev = max(np.abs((X[0] - Xt) / P.scale_x).max(), np.abs((U[0] - Ut) / P.scale_u).max(), np.abs(T[0] - Tt).max())
        print(f"after {iters} iteration(s): max scaled deviation {dev:.2e}; stats {st[0, :10]}")
        assert dev <= 1e-5, (iters, dev)       # the twin's Hessian is a finite difference (1e-7 .. 1e-8 relative)
    w, info = tw.solve(P, w0, max_iter=150, tol=1e-6)
    X, U, T, st = ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X0[None], U0[None], T0[None],
                                          max_iter=150, tol=1e-6)
    Xt, Ut, Tt = P.unpack(w)
    dev = max(np.abs((X[0] - Xt) / P.scale_x).max(), np.abs((U[0] - Ut) / P.scale_u).max(), np.abs(T[0] - Tt).max())
    print(f"converged: GPU iterations {st[0, 0]:.0f} (twin {info['iterations']}), kkt {st[0, 1]:.2e} viol {st[0, 2]:.2e} "
          f"compl {st[0, 3]:.2e} lap {st[0, 4]:.5f} (twin {info['lap_time']:.5f}, QSS {d['qss_lap']:.4f}); deviation {dev:.2e}")
    assert st[0, 5] == 1.0 and max(st[0, 1], st[0, 2], st[0, 3]) <= 1e-6
    assert st[0, 4] < d["qss_lap"] - 1.0 and abs(st[0, 4] - info["lap_time"]) <= 1e-6
    assert dev <= 1e-6, dev
    # the reference's tolerances (traj_opt_double_track.yaml:8-10: tol 0.1, constr_viol_tol 0.1) are far looser
    assert max(st[0, 1], st[0, 2]) <= defaults.SOLVER["tol"]
    # the solution satisfies the reference's own functions (rl_dt_eval_nodes, pinned by G8)
    eq, g, cost = ops.dt_eval_nodes(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X, U, T)
    assert np.abs(eq[0, :, :6] / P.scale_x).max() <= 1e-6 and np.abs(eq[0, :, 6]).max() / P.scale_u[3] <= 1e-6
    assert np.abs(eq[0, :, 7]).max() == 0.0 and g[0, :, :4].max() <= 1e-6 and g[0, :, 12:].max() <= 1e-6
    # ... ALL of them: 8 equalities and 14 inequality columns of rl_dt_eval_nodes, each in the reference's own scaling
    # (power / Pmax, speed / speed_cap, forces / Fd_max, steering / delta_max; rates as the reference scales u)
    m, sx, su = P.m, P.scale_x, P.scale_u
    gs = np.array([1, 1, 1, 1, m["Pmax"], sx[5], su[0], su[0], su[2], su[2], su[0], su[2], sx[1], sx[1]])
    worst = (g[0] / gs).max(axis=0)
    print("scaled inequality maxima per column:", np.array2string(worst, precision=2))
    assert (worst <= 1e-6).all(), worst
    es = np.r_[P.scale_x, su[3], 1.0]
    assert (np.abs(eq[0] / es).max(axis=0) <= 1e-6).all(), np.abs(eq[0] / es).max(axis=0)
    # and the active set is not empty: power, tyre and track limits are what shapes a minimum-time lap
    assert (worst[:5] >= -1e-3).any() and worst[12:].max() >= -1e-3


@pytest.mark.gpu
def test_hessian_chain_rule_vs_forward_over_forward_sweep(tmp_path):
    """The two Hessian implementations of the library -- the chain rule through the Hermite-Simpson midpoint
    (k_mt_hes_*, default) and forward-over-forward duals through the whole pair function (k_mt_derivs<2>,
    RL_MT_HES_SWEEP=1) -- drive the same iteration: after 12 iterations from the QSS warm start the iterates agree
    to rounding.  (Each library instance reads the switch when its context is created: two processes.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    for mode in ("0", "1"):
        f = str(tmp_path / f"it_{mode}.npz")
        env = dict(os.environ, RL_MT_HES_SWEEP=mode)
        subprocess.run([sys.executable, os.path.join(here, "mintime_run.py"), f, "12"], check=True, env=env, timeout=300)
        out[mode] = np.load(f)
    a, b = out["0"], out["1"]
    assert a["st"][0, 0] == b["st"][0, 0] == 12.0
    su = np.maximum(np.abs(b["U"]).max(axis=(0, 1)), 1.0)      # controls in their own scale (u[1] is identically zero)
    dev = max(np.abs(a["X"] - b["X"]).max(), np.abs(a["T"] - b["T"]).max(), np.abs((a["U"] - b["U"]) / su).max())
    print(f"chain rule vs sweep after 12 iterations: max deviation {dev:.2e}, kkt {a['st'][0, 1]:.6e} / {b['st'][0, 1]:.6e}")
    assert dev <= 1e-8, dev
    assert a["st"][0, 1] == pytest.approx(b["st"][0, 1], rel=1e-6)


@pytest.mark.gpu
def test_fused_node_kernel_vs_separate_assembly_kernels(tmp_path):
    """k_mt_node (Jacobian, Hessian, KKT blocks and right-hand side of the node pairs in one pass, the default) against
    the four kernels it replaces (RL_MT_UNFUSED=1: k_mt_jac_assemble, k_mt_hes_assemble, k_mt_prepare, k_mt_assemble):
    same formulas entry by entry; only the right-hand side is summed in another order (rhs0 - mu r1).  After 1 and after 12
    iterations from the QSS warm start the iterates agree to rounding (1e-10 / 1e-9); run to convergence (KKT 1e-6, the default) both stop after the same
    number of iterations (+-2) at the same point (1e-6).  (Until round 4 the second stage compared the iterates after 80
    iterations at an unreachable tolerance: with the round-4 damping constants the solve has converged long before, and what
    was compared was the rounding-level wandering of two converged iterates.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for iters, tol, ktol in (("1", 1e-10, "1e-12"), ("12", 1e-9, "1e-12"), ("200", 1e-6, "1e-6")):   # 1 iteration: measured 6e-12
        out = {}
        for mode in ("0", "1"):
            f = str(tmp_path / f"fu_{mode}_{iters}.npz")
            env = dict(os.environ, RL_MT_UNFUSED=mode)
            subprocess.run([sys.executable, os.path.join(here, "mintime_run.py"), f, iters, "8.0", ktol], check=True, env=env, timeout=300)
            out[mode] = np.load(f)
        a, b = out["0"], out["1"]
        su = np.maximum(np.abs(b["U"]).max(axis=(0, 1)), 1.0)
        dev = max(np.abs(a["X"] - b["X"]).max(), np.abs(a["T"] - b["T"]).max(), np.abs((a["U"] - b["U"]) / su).max())
        print(f"fused vs separate kernels, max_iter {iters}: iterations {a['st'][0, 0]:.0f} / {b['st'][0, 0]:.0f}, "
              f"max deviation {dev:.2e}, kkt {a['st'][0, 1]:.3e} / {b['st'][0, 1]:.3e}, status {a['st'][0, 5]:.0f} / {b['st'][0, 5]:.0f}")
        assert a["st"][0, 5] == b["st"][0, 5] == (1.0 if iters == "200" else 0.0)    # 1 / 12 iterations: still running; 200: converged
        assert abs(a["st"][0, 0] - b["st"][0, 0]) <= (2 if iters == "200" else 0)
        assert dev <= tol, dev


@pytest.mark.gpu
@pytest.mark.parametrize("spacing", [8.0, 5.0, 3.0])
def test_four_front_vs_two_front_elimination(tmp_path, spacing):
    """k_mt_kkt4 (four fronts per instance, the default for N >= 64) against k_mt_kkt (two fronts, RL_MT_KKT4=0): the same
    block factorisation in another elimination order -- equal refactorisation decisions (inertia), iterates equal to rounding
    after 12 iterations, for node counts that put the cuts q1, m, q3 at different parities."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    # two stages: TIGHT after ONE iteration -- one KKT solve from the same point, before any refactorisation: a real divergence of
    # the two eliminations shows here (measured 2.6e-12 .. 1.6e-11 over the three node counts) -- then after 12 (another elimination
    # order = another rounding of the same directions; the refactorisations of iterations 2 - 3 solve ill-conditioned systems and
    # the round-4 damping constants carry the difference on: measured 1e-9 .. 4.8e-8 from iteration 3 on, not growing after)
    for iters, tol in ((1, 1e-10), (12, 1e-7)):
        out = {}
        for mode in ("0", "1"):
            f = str(tmp_path / f"k4_{mode}_{iters}.npz")
            env = dict(os.environ, RL_MT_KKT4=mode)
            subprocess.run([sys.executable, os.path.join(here, "mintime_run.py"), f, str(iters), str(spacing)], check=True, env=env, timeout=300)
            out[mode] = np.load(f)
        a, b = out["1"], out["0"]
        N = a["X"].shape[1]
        su = np.maximum(np.abs(b["U"]).max(axis=(0, 1)), 1.0)
        dev = max(np.abs(a["X"] - b["X"]).max(), np.abs(a["T"] - b["T"]).max(), np.abs((a["U"] - b["U"]) / su).max())
        print(f"four vs two fronts, N = {N}: max deviation after {iters} iterations {dev:.2e}, kkt {a['st'][0, 1]:.6e} / {b['st'][0, 1]:.6e}, "
              f"refactorisations {a['st'][0, 9]:.0f} / {b['st'][0, 9]:.0f}")
        assert N >= 64 and a["st"][0, 0] == b["st"][0, 0] == float(iters)
        assert a["st"][0, 9] == b["st"][0, 9]
        assert dev <= tol, (iters, dev)


@pytest.mark.gpu
@pytest.mark.parametrize("spacing", [31.0, 22.0, 13.0, 5.0])
def test_fused_node_kernel_tails(tmp_path, spacing):
    """Node counts that leave partial runs / blocks in every kernel of the default path (runs of 9 pairs in k_mt_node, blocks of
    3 / 8 nodes in the derivative kernels, 64 rows in k_mt_dir): two iterations against the separate assembly kernels
    (one thread block per node there: no tails).  The two paths sum the right-hand side in different orders; one KKT solve
    turns that into 1e-11 on every size, and these coarse grids (31 m between nodes on a kart track) amplify it by a decade per
    iteration -- a misplaced tail entry would show as O(1)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    for mode in ("0", "1"):
        f = str(tmp_path / f"tail_{mode}.npz")
        env = dict(os.environ, RL_MT_UNFUSED=mode)
        subprocess.run([sys.executable, os.path.join(here, "mintime_run.py"), f, "2", str(spacing)], check=True, env=env, timeout=300)
        out[mode] = np.load(f)
    a, b = out["0"], out["1"]
    N = a["X"].shape[1]
    su = np.maximum(np.abs(b["U"]).max(axis=(0, 1)), 1.0)
    dev = max(np.abs(a["X"] - b["X"]).max(), np.abs(a["T"] - b["T"]).max(), np.abs((a["U"] - b["U"]) / su).max())
    print(f"N = {N} (mod 9: {N % 9}, mod 3: {N % 3}, mod 8: {N % 8}): max deviation after 2 iterations {dev:.2e}, kkt {a['st'][0, 1]:.3e} / {b['st'][0, 1]:.3e}")
    assert np.isfinite(a["X"]).all() and a["st"][0, 0] == b["st"][0, 0] == 2.0
    assert dev <= 1e-8, dev


@pytest.mark.gpu
def test_mintime_batch_with_per_instance_widths(coarse):
    """A batch of tracks that differ in their widths (BASELINE config 2's perturbation applied to config 5):
    instances are independent (a duplicate agrees bit for bit, instance 0 equals the single solve), wider tracks
    are not slower."""
    from spline_trajectory_optimization_amd import ops
    d, P, w0 = coarse
    X0, U0, T0 = P.unpack(w0)
    B = 6
    scale = np.array([1.0, 0.9, 1.1, 1.25, 1.0, 0.95])
    left = P.left[None] * scale[:, None]; right = P.right[None] * scale[:, None]
    rep = lambda a: np.repeat(a[None], B, axis=0)  # noqa: E731
    X, U, T, st = ops.mintime_solve_batch(P.m, P.s, P.kappa, left, right, P.margin, P.L, rep(X0), rep(U0), rep(T0),
                                          max_iter=200, tol=1e-6)
    print("batch: iterations", st[:, 0], "lap", st[:, 4], "status", st[:, 5])
    assert (st[:, 5] == 1.0).all()
    np.testing.assert_array_equal(X[0], X[4]); np.testing.assert_array_equal(T[0], T[4])
    X1, U1, T1, st1 = ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X0[None], U0[None], T0[None],
                                              max_iter=200, tol=1e-6)
    np.testing.assert_array_equal(X[0], X1[0])
    assert st[3, 4] <= st[2, 4] + 1e-6 <= st[0, 4] + 2e-6 <= st[1, 4] + 3e-6   # wider is not slower
    with pytest.raises(Exception):
        ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left * 0.2, P.right * 0.2, P.margin, P.L, X0[None], U0[None], T0[None])


@pytest.mark.gpu
def test_reference_pipeline_on_the_example_track():
    """The reference CLI's pipeline (entrypoints/traj_opt_double_track.py:24-86) through the mirror classes at the
    example's own resolution (MGKT, interval 1 m, yaml defaults): RaceTrack + interpolants -> QSS warm start on
    the GPU -> NLP solve on the GPU -> TTL-ready table."""
    import os
    import time
    from mintime_problem import MGKT, _load
    from spline_traj_optm.min_time_optm.min_time_optimizer import set_up_double_track_problem
    from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import optimise_track
    from spline_trajectory_optimization_amd.models.race_track import RaceTrack
    from spline_trajectory_optimization_amd.models.trajectory import Trajectory, load_ttl, save_ttl
    from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams
    est = defaults.ESTIMATES
    rt = RaceTrack("Test track", _load("MGKT_OUT_BOUND_enu.csv"), _load("MGKT_IN_BOUND_enu.csv"), _load("MGKT_CENTER_enu.csv"),
                   s=1.0, interval=defaults.SOLVER["interval"])
    veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                                est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                                est["max_speed_mps"], est["max_jerk_mpsc"]))
    assert abs(rt.left_intp(rt.abscissa[5]) - np.hypot(*(rt.center_d[5, 0:2] - rt.center_d[5, 9:11]))) < 1e-9
    assert callable(set_up_double_track_problem)
    t0 = time.time()
    out, X, U, T, st = optimise_track(rt, veh, defaults.MODEL, defaults.SOLVER["average_track_width"],
                                      defaults.SOLVER["speed_cap"], max_iter=300, tol=1e-6)
    N = len(out)
    print(f"MGKT interval 1 m: N={N} iterations {st[0]:.0f} status {st[5]:.0f} kkt {st[1]:.1e} viol {st[2]:.1e} "
          f"lap {st[4]:.4f} s, wall {time.time() - t0:.2f} s (set-up + QSS + solve)")
    assert st[5] == 1.0 and max(st[1], st[2], st[3]) <= 1e-6
    assert isinstance(out, Trajectory) and np.isfinite(out.points).all()
    # the optimised line stays between the boundaries with the vehicle margin, and is faster than the QSS profile
    margin = defaults.MODEL["vehicle_width"] / 2 + defaults.MODEL["safety_margin"]
    assert (X[:, 1] <= rt.left_intp(rt.abscissa) - margin + 1e-6).all() and (X[:, 1] >= rt.right_intp(rt.abscissa) + margin - 1e-6).all()
    assert T.sum() < 56.0 and X[:, 5].max() <= defaults.SOLVER["speed_cap"]
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mgkt_optm.ttl")
    out.ttl_num, out.origin = 1, (0.0, 0.0, 0.0)
    save_ttl(p, out)
    back = load_ttl(p)
    np.testing.assert_allclose(back.points[:, :17], out.points[:, :17], rtol=0, atol=0)


@pytest.mark.gpu
def test_monza_from_the_qss_guess():
    """The same pipeline on the reference's OTHER example track (Monza, 5.8 km; nodes every 5 m -> N = 1158) with the
    yaml's kart model: converges from the QSS guess well inside the yaml's tolerances (traj_opt_double_track.yaml:8-10),
    stays between the boundaries, and is faster than the QSS profile it started from."""
    from spline_trajectory_optimization_amd import batch
    from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import optimise_track
    from spline_trajectory_optimization_amd.models.race_track import RaceTrack
    from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams
    from spline_trajectory_optimization_amd.simulator.simulator import Simulator
    centre, left, right = batch.load_monza()
    est = defaults.ESTIMATES
    rt = RaceTrack("Monza", left, right, centre, s=10.0, interval=5.0)
    veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                                est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                                est["max_speed_mps"], est["max_jerk_mpsc"]))
    qss = rt.center_d.copy(); rt.fill_trajectory_boundaries(qss)
    qss_lap = float(Simulator(veh).run_simulation(qss, False).trajectory[:, 16].sum())
    out, X, U, T, st = optimise_track(rt, veh, defaults.MODEL, 10.0, defaults.SOLVER["speed_cap"], max_iter=300, tol=1e-6)
    print(f"Monza N={len(out)}: iterations {st[0]:.0f} status {st[5]:.0f} kkt {st[1]:.1e} viol {st[2]:.1e} lap {st[4]:.3f} s "
          f"(QSS {qss_lap:.3f} s)")
    assert st[5] == 1.0 and max(st[1], st[2], st[3]) <= 1e-6 <= defaults.SOLVER["tol"]
    assert st[4] < qss_lap and abs(T.sum() - st[4]) <= 1e-9
    margin = defaults.MODEL["vehicle_width"] / 2 + defaults.MODEL["safety_margin"]
    assert (X[:, 1] <= rt.left_intp(rt.abscissa) - margin + 1e-6).all() and (X[:, 1] >= rt.right_intp(rt.abscissa) + margin - 1e-6).all()
    assert np.isfinite(out.points).all() and X[:, 5].min() >= 1.0 - 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("track,interval,changes", [("mgkt", 4.0, {"mu": 0.7}), ("mgkt", 8.0, {"mu": 0.7}),
                                                    ("monza", 10.0, {"Pmax": 2.0}), ("monza", 5.0, {})])
def test_hard_cells_of_the_robustness_matrix(track, interval, changes):
    """Coarse grids, tracks 25 % narrower to 30 % wider, a model that no longer matches the QSS guess (less friction,
    twice the power): the cells of tools/mintime_robustness.py in which the multipliers used to run away while the
    primal step was cut to a few per cent (the dual step length is now tied to the primal one).  All converge."""
    from spline_trajectory_optimization_amd.min_time_optm.example import variant_problem
    prob = variant_problem(track, interval, changes)
    B = 32
    e = np.random.default_rng(7).uniform(-0.25, 0.3, size=(B, 1))
    X, U, T, st = prob.solve_batch(prob.left[None] * (1 + e), prob.right[None] * (1 + e), max_iter=300, tol=1e-6)
    print(f"[{track} {interval} m {changes}] N={prob.N}: iterations {st[:, 0].min():.0f}..{st[:, 0].max():.0f}, "
          f"converged {int((st[:, 5] == 1).sum())}/{B}, lap {st[:, 4].min():.2f}..{st[:, 4].max():.2f} s")
    assert (st[:, 5] == 1.0).all(), np.where(st[:, 5] != 1.0)[0]
    assert max(st[:, 1].max(), st[:, 2].max(), st[:, 3].max()) <= 1e-6


@pytest.mark.gpu
def test_width_perturbed_batch_at_full_resolution_all_converge():
    """BASELINE config 2's perturbation applied to config 5 at the example's own resolution (MGKT, 828 nodes): every
    instance of the batch reaches the 1e-6 tolerance (the globalisation once left 1 - 2 % of such batches creeping or
    cycling next to a feasible point: the floor under the growth cap and the filter memory of k_mt_step)."""
    from mintime_problem import _load
    from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import DoubleTrackProblem
    from spline_trajectory_optimization_amd.models.race_track import RaceTrack
    from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams
    from spline_trajectory_optimization_amd.simulator.simulator import Simulator
    est = defaults.ESTIMATES
    rt = RaceTrack("MGKT", _load("MGKT_OUT_BOUND_enu.csv"), _load("MGKT_IN_BOUND_enu.csv"), _load("MGKT_CENTER_enu.csv"),
                   s=1.0, interval=1.0)
    veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                                est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                                est["max_speed_mps"], est["max_jerk_mpsc"]))
    traj = rt.center_d.copy(); rt.fill_trajectory_boundaries(traj)
    traj = Simulator(veh).run_simulation(traj, False).trajectory
    prob = DoubleTrackProblem({"N": len(traj), "model": defaults.MODEL, "race_track": rt, "traj_d": traj,
                               "average_track_width": 7.0, "speed_cap": 30.0})
    B = 96
    e = np.random.default_rng(1234).uniform(-0.1, 0.15, size=(B, 1))
    X, U, T, st = prob.solve_batch(prob.left[None] * (1 + e), prob.right[None] * (1 + e), max_iter=200, tol=1e-6)
    print(f"[mintime batch N={len(traj)} B={B}] iterations {st[:, 0].min():.0f}..{st[:, 0].max():.0f}, "
          f"kkt max {st[:, 1].max():.1e}, viol max {st[:, 2].max():.1e}, lap {st[:, 4].min():.3f}..{st[:, 4].max():.3f} s")
    assert (st[:, 5] == 1.0).all(), np.where(st[:, 5] != 1.0)[0]
    assert max(st[:, 1].max(), st[:, 2].max(), st[:, 3].max()) <= 1e-6
    # wider tracks are not slower (instances sorted by their width factor)
    order = np.argsort(e[:, 0])
    assert (np.diff(st[order, 4]) <= 1e-6).all()


@pytest.mark.gpu
def test_config5_chain_on_the_device(coarse):
    """BASELINE config 5 without a host round trip: QSS simulator (rl_qss_sim_dev) -> initial guess -> NLP solve
    (rl_mintime_solve_batch_dev) on device tensors, one stream, scratch from the context's arena.  Same result,
    bit for bit, as the host entry points."""
    import torch
    from scipy.interpolate import CubicSpline
    from spline_trajectory_optimization_amd import ops
    d, P, w0 = coarse
    est = defaults.ESTIMATES
    acc = CubicSpline(*np.array(est["acc_speed_loopup"]).T); dcc = CubicSpline(*np.array(est["dcc_speed_lookup"]).T)
    params = np.array([est["max_lon_acc_mpss"], est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                       est["max_speed_mps"], est["max_jerk_mpsc"]])
    B = 3
    pts = np.repeat(d["points"][None], B, axis=0)
    # host path
    sim, it = ops.qss_sim(pts, acc.x, acc.c, dcc.x, dcc.c, params)
    X0 = np.zeros((B, P.N, 6)); X0[:, :, 0] = P.s; X0[:, :, 5] = np.maximum(sim[:, :, 4], 1.5)
    U0 = np.tile(np.array([1.0, 0.0, 0.001, 0.0]), (B, P.N, 1)); T0 = np.maximum(np.roll(sim[:, :, 16], -1, axis=1), 1e-3)
    Xh, Uh, Th, sth = ops.mintime_solve_batch(P.m, P.s, P.kappa, P.left, P.right, P.margin, P.L, X0, U0, T0, max_iter=160, tol=1e-6)
    # device path
    dev = torch.device("cuda", 0)
    tp = torch.from_numpy(pts).to(dev)
    itd = ops.qss_sim_torch(tp, acc.x, acc.c, dcc.x, dcc.c, params)
    X = torch.zeros((B, P.N, 6), dtype=torch.float64, device=dev); X[:, :, 0] = torch.from_numpy(P.s).to(dev)
    X[:, :, 5] = torch.clamp(tp[:, :, 4], min=1.5)
    U = torch.tensor([1.0, 0.0, 0.001, 0.0], dtype=torch.float64, device=dev).repeat(B, P.N, 1).contiguous()
    T = torch.clamp(torch.roll(tp[:, :, 16], -1, dims=1), min=1e-3).contiguous()
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    st = ops.mintime_solve_torch(P.m, g(P.s), g(P.kappa), g(P.left), g(P.right), P.margin, P.L, X, U, T, max_iter=160, tol=1e-6)
    torch.cuda.synchronize()
    assert np.array_equal(itd.cpu().numpy(), it)
    np.testing.assert_array_equal(X.cpu().numpy(), Xh); np.testing.assert_array_equal(T.cpu().numpy(), Th)
    np.testing.assert_array_equal(st.cpu().numpy()[:, :6], sth[:, :6])
    assert (sth[:, 5] == 1.0).all()


@pytest.mark.gpu
def test_cli_lines_against_the_facade(tmp_path):
    """entrypoints/traj_opt_double_track.py:24-86 line by line against the mirror: RaceTrack, QSS warm start,
    `(X, U, T), (scale_x, scale_u, scale_t), opti = set_up_double_track_problem(params)`, `opti.solve()` in try/except,
    read-back through `opti.debug.value(.) * scale + offset`, frenet_to_global, bounds, distances, save_ttl.  (The
    casadi.DM text dumps of :72, :83-86 are the caller's casadi; np.savetxt stands in.)  Coarser nodes than the yaml's
    1 m so that the test stays short; the yaml's solver settings (max_iter 500, tol 0.1) and the reference's own default
    initial guess (:146-151)."""
    from mintime_problem import _load
    from spline_traj_optm.min_time_optm import min_time_optimizer as optm
    from spline_traj_optm.models.race_track import RaceTrack
    from spline_traj_optm.models.trajectory import Trajectory, load_ttl, save_ttl
    from spline_traj_optm.models.vehicle import Vehicle, VehicleParams
    from spline_traj_optm.simulator.simulator import Simulator
    params = {"interval": 4.0, "estimates": defaults.ESTIMATES, "model": defaults.MODEL, "verbose": False,
              "max_iter": defaults.SOLVER["max_iter"], "tol": defaults.SOLVER["tol"],
              "average_track_width": defaults.SOLVER["average_track_width"], "speed_cap": defaults.SOLVER["speed_cap"],
              "output": str(tmp_path / "optm.ttl")}
    interval = params["interval"]                                                                            # :23
    race_track = RaceTrack("Test track", _load("MGKT_OUT_BOUND_enu.csv"), _load("MGKT_IN_BOUND_enu.csv"),
                           _load("MGKT_CENTER_enu.csv"), s=1.0, interval=interval)                          # :24-31
    traj_d = race_track.center_d.copy()
    race_track.fill_trajectory_boundaries(traj_d)                                                            # :32-33
    estimates = params["estimates"]                                                                          # :35-49
    vp = VehicleParams(np.array(estimates["acc_speed_loopup"]), np.array(estimates["dcc_speed_lookup"]),
                       estimates["max_lon_acc_mpss"], estimates["max_lon_dcc_mpss"], estimates["max_left_acc_mpss"],
                       estimates["max_right_acc_mpss"], estimates["max_speed_mps"], estimates["max_jerk_mpsc"])
    result = Simulator(Vehicle(vp)).run_simulation(traj_d, False)
    traj_d = result.trajectory
    params["N"] = len(traj_d); params["traj_d"] = traj_d; params["race_track"] = race_track                  # :54-56
    (X, U, T), (scale_x, scale_u, scale_t), opti = optm.set_up_double_track_problem(params)                  # :58-59
    try:
        sol = opti.solve()                                                                                   # :60-63
    except Exception as e:
        sol = None
        print(e)
    x = opti.debug.value(X) * scale_x + np.hstack([race_track.abscissa[:, np.newaxis], np.zeros((len(traj_d), 5))])  # :65-66
    u = opti.debug.value(U) * scale_u
    t = opti.debug.value(T) * scale_t
    print(f"[Optimal lap time: {np.sum(t) * scale_t}]  stats {opti.stats()}")                                # :70
    opt_traj_d = traj_d.copy()                                                                               # :74-82
    global_pose = race_track.frenet_to_global(x[:, 0].T, x[:, 1].T, x[:, 2].T)
    opt_traj_d[:, 0:2] = global_pose[:, 0:2]
    opt_traj_d[:, Trajectory.YAW] = global_pose[:, 2].full().squeeze()
    opt_traj_d[:, Trajectory.SPEED] = x[:, 5]
    race_track.fill_trajectory_boundaries(opt_traj_d)
    opt_traj_d.fill_distance()
    opt_traj_d.ttl_num, opt_traj_d.origin = 1, (0.0, 0.0, 0.0)
    save_ttl(params["output"], opt_traj_d)
    np.savetxt(str(tmp_path / "x_optm.txt"), x); np.savetxt(str(tmp_path / "u_optm.txt"), u); np.savetxt(str(tmp_path / "t_optm.txt"), t)
    # --- what a user of the reference expects to hold afterwards
    assert sol is not None and opti.stats()["success"], opti.stats()
    np.testing.assert_array_equal(sol.value(X), opti.value(X))
    qss_lap = float(traj_d[:, Trajectory.TIME].sum())
    assert x.shape == (len(traj_d), 6) and u.shape == (len(traj_d), 4) and t.shape == (len(traj_d),)
    np.testing.assert_array_equal(x[:, 0], race_track.abscissa)                     # the abscissa pin (:130)
    assert t.min() > 0.0 and t.sum() < qss_lap
    margin = defaults.MODEL["vehicle_width"] / 2 + defaults.MODEL["safety_margin"]
    assert (x[:, 1] <= race_track.left_intp(race_track.abscissa) - margin + 1e-3).all()
    assert (x[:, 1] >= race_track.right_intp(race_track.abscissa) + margin - 1e-3).all()
    back = load_ttl(params["output"])
    np.testing.assert_allclose(back.points[:, :17], opt_traj_d.points[:, :17], rtol=0, atol=0)
    # the yaml's tolerance (0.1) is IPOPT's SCALED NLP error; the facade maps it onto the solver's absolute KKT tolerance
    # (min(tol, 1e-4)) and says so: a user running the stock yaml gets the optimum, not a lap stopped seconds short of it
    st0 = opti.stats()
    from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import IPOPT_TOL_CAP
    assert st0["requested_tol"] == defaults.SOLVER["tol"] and st0["effective_tol"] == min(defaults.SOLVER["tol"], IPOPT_TOL_CAP)
    assert max(st0["dual_inf"], st0["constr_viol"], st0["compl"]) <= st0["effective_tol"]
    # a second solve from that point at 1e-6: the yaml-settings lap is within 0.1 % of it (measured: 1e-3 s of 46.7 s)
    with pytest.warns(RuntimeWarning, match="without a counterpart"):
        opti.solver("ipopt", {}, {"max_iter": 300, "tol": 1e-6, "constr_viol_tol": 1e-8})
        sol2 = opti.solve()
    assert opti.stats()["ignored_options"] == {"constr_viol_tol": 1e-8}
    lap_yaml, lap_tight = float(np.sum(t)), float(np.sum(sol2.value(T)))
    print(f"lap at the yaml's settings {lap_yaml:.5f} s, at 1e-6 {lap_tight:.5f} s")
    assert abs(lap_yaml - lap_tight) <= 1e-3 * lap_tight
    # the build's own guess variant converges too; the NLP is not convex, so two starting points may end in neighbouring
    # local optima (measured: 46.710 s against 46.697 s on this 4 m grid) -- the laps agree to a tenth of a per cent
    (_, _, Tc), _, opti_c = optm.set_up_double_track_problem(dict(params, initial_guess="clipped", tol=1e-6))
    opti_c.solve()
    assert abs(float(np.sum(opti_c.value(Tc))) - lap_tight) < 0.05


@pytest.mark.gpu
def test_dev_calls_on_two_torch_streams_do_not_share_scratch(coarse):
    """The *_dev entry points carve their work arrays out of ONE arena per context, from offset 0.  Two solves enqueued on
    DIFFERENT torch streams without a host synchronisation in between must still run one after the other on that scratch
    (rl_mincurv.hip: Arena::begin orders a call behind the arena's previous user with an event when the stream has
    changed): both give, bit for bit, what they give when run alone."""
    import torch
    from spline_trajectory_optimization_amd import ops
    d, P, w0 = coarse
    dev = torch.device("cuda", 0)
    X0, U0, T0 = P.unpack(w0)
    g = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    B = 4
    scale = np.array([1.0, 0.92, 1.1, 1.2])
    left = g(P.left[None] * scale[:, None]); right = g(P.right[None] * scale[:, None])
    s_, k_ = g(P.s), g(P.kappa)

    def fresh():
        return (g(np.repeat(X0[None], B, axis=0)), g(np.repeat(U0[None], B, axis=0)), g(np.repeat(T0[None], B, axis=0)))
    # reference: each solve alone
    alone = []
    for sl in (slice(0, 2), slice(2, 4)):
        X, U, T = fresh()
        st = ops.mintime_solve_torch(P.m, s_, k_, left[sl].contiguous(), right[sl].contiguous(), P.margin, P.L,
                                     X[sl].contiguous(), U[sl].contiguous(), T[sl].contiguous(), max_iter=40, tol=1e-6)
        torch.cuda.synchronize()
        alone.append(st.cpu().numpy())
    # the same two solves on two streams, enqueued back to back
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    outs = []
    for stream, sl in ((sa, slice(0, 2)), (sb, slice(2, 4))):
        with torch.cuda.stream(stream):
            X, U, T = fresh()
            st = ops.mintime_solve_torch(P.m, s_, k_, left[sl].contiguous(), right[sl].contiguous(), P.margin, P.L,
                                         X[sl].contiguous(), U[sl].contiguous(), T[sl].contiguous(), max_iter=40, tol=1e-6)
            outs.append(st)
    torch.cuda.synchronize()
    for a_, o_ in zip(alone, outs):
        np.testing.assert_array_equal(o_.cpu().numpy(), a_)
