"""The reference's example problem of the min-time optimiser (min_time_optm/example/: the MGKT kart circuit with
the yaml's model and estimates, entrypoints/traj_opt_double_track.py:24-86) as a BATCH of width-perturbed tracks --
BASELINE config 2's perturbation applied to config 5.  Used by bench.py's `mintime_nlp` leg and tools/bench_mintime.py."""
import os
import time

import numpy as np

from . import defaults
from .min_time_optimizer import DoubleTrackProblem
from ..models.race_track import RaceTrack
from ..models.vehicle import Vehicle, VehicleParams
from ..simulator.simulator import Simulator

MGKT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "race_track", "mgkt")


def load_mgkt(name):
    return np.loadtxt(os.path.join(MGKT_DIR, name), delimiter=",", skiprows=1, usecols=(0, 1))


def mgkt_problem(interval=1.0):
    """RaceTrack + QSS warm start (on the GPU) + the NLP of the example at the yaml's node spacing."""
    est = defaults.ESTIMATES
    rt = RaceTrack("MGKT", load_mgkt("MGKT_OUT_BOUND_enu.csv"), load_mgkt("MGKT_IN_BOUND_enu.csv"),
                   load_mgkt("MGKT_CENTER_enu.csv"), s=1.0, interval=interval)
    veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                                est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                                est["max_speed_mps"], est["max_jerk_mpsc"]))
    traj = rt.center_d.copy()
    rt.fill_trajectory_boundaries(traj)
    traj = Simulator(veh).run_simulation(traj, False).trajectory
    return DoubleTrackProblem({"N": len(traj), "model": defaults.MODEL, "race_track": rt, "traj_d": traj,
                               "average_track_width": 7.0, "speed_cap": 30.0})


def variant_problem(track, interval, model_changes=None):
    """One cell of the robustness matrix (tools/mintime_robustness.py): `track` in ("mgkt", "monza"), node spacing
    [m], factors applied to entries of the yaml's model (e.g. {"mu": 0.7}); the QSS warm start stays the kart's."""
    from .. import batch
    est = defaults.ESTIMATES
    model = dict(defaults.MODEL)
    for key, fac in (model_changes or {}).items():
        model[key] = defaults.MODEL[key] * fac
    if track == "mgkt":
        rt = RaceTrack("MGKT", load_mgkt("MGKT_OUT_BOUND_enu.csv"), load_mgkt("MGKT_IN_BOUND_enu.csv"),
                       load_mgkt("MGKT_CENTER_enu.csv"), s=1.0, interval=interval)
        width = 7.0
    else:
        centre, left, right = batch.load_monza()
        rt = RaceTrack("Monza", left, right, centre, s=10.0, interval=interval)
        width = 10.0
    veh = Vehicle(VehicleParams(np.array(est["acc_speed_loopup"]), np.array(est["dcc_speed_lookup"]), est["max_lon_acc_mpss"],
                                est["max_lon_dcc_mpss"], est["max_left_acc_mpss"], est["max_right_acc_mpss"],
                                est["max_speed_mps"], est["max_jerk_mpsc"]))
    traj = rt.center_d.copy()
    rt.fill_trajectory_boundaries(traj)
    traj = Simulator(veh).run_simulation(traj, False).trajectory
    return DoubleTrackProblem({"N": len(traj), "model": model, "race_track": rt, "traj_d": traj, "average_track_width": width,
                               "speed_cap": 30.0})


def perturbed_widths(prob, B, seed=1234):
    """[B, N] left / right distances: every track scaled by its own factor in [0.9, 1.15)."""
    e = np.random.default_rng(seed).uniform(-0.1, 0.15, size=(B, 1))
    return prob.left[None] * (1 + e), prob.right[None] * (1 + e)


def timed_batch_solve(B, max_iter=300, tol=1e-6, repeats=3):
    """The batched solve timed so that another run reproduces the number: a warm-up call AT THE FULL BATCH SIZE (its time is
    reported separately: the first call at a batch size allocates the context's staging pool and arena), then `repeats`
    timed calls on the host (copies of the initial guess and of the solution included) -- `wall_s` is their MEDIAN --
    and the device-side span of one more solve of the same batch (rl_mintime_solve_batch_dev on torch's stream, events
    around it, exactly the iterations the slowest instance needs, no host copies)."""
    prob = mgkt_problem()
    left, right = perturbed_widths(prob, B)
    t0 = time.perf_counter()
    prob.solve_batch(left[:2], right[:2], max_iter=8)          # module load, kernels resident
    t_small = time.perf_counter() - t0
    t0 = time.perf_counter()
    prob.solve_batch(left, right, max_iter=max_iter, tol=tol)  # warm-up at the timed batch size
    t_first = time.perf_counter() - t0
    walls = []
    for _ in range(max(1, repeats)):
        t0 = time.perf_counter()
        X, U, T, st = prob.solve_batch(left, right, max_iter=max_iter, tol=tol)
        walls.append(time.perf_counter() - t0)
    dt = float(np.median(walls))
    leg = {"metric": "min-time double-track NLP solves/sec (MGKT, N=%d nodes, 9 unknowns + 7 equalities + 17 inequalities "
                     "per node)" % prob.N, "value": B / dt, "unit": "NLP solves/s", "batch": B, "nodes": int(prob.N),
           "wall_s": dt, "wall_s_median": dt, "wall_s_min": float(min(walls)), "wall_s_all": [float(w) for w in walls],
           "timed_calls": len(walls), "first_call_at_this_batch_s": t_first, "warmup_call_b2_s": t_small,
           "converged": int((st[:, 5] == 1).sum()), "iterations_mean": float(st[:, 0].mean()),
           "iterations_max": float(st[:, 0].max()), "kkt_max": float(st[:, 1].max()), "viol_max": float(st[:, 2].max()),
           "lap_s_min_max": [float(st[:, 4].min()), float(st[:, 4].max())], "tol": tol,
           "includes": "value = batch / MEDIAN of the timed host calls, each with the host->device copies of the initial guess and "
                       "device->host of the solution; the warm-up call at the same batch size is not in it"}
    try:   # device-side span: the same solve on device-resident tensors
        import torch
        from .. import ops
        dev = torch.device("cuda", torch.cuda.current_device())
        rep = lambda a: torch.from_numpy(np.ascontiguousarray(np.repeat(np.asarray(a)[None], B, axis=0))).to(dev)  # noqa: E731
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
        iters = int(st[:, 0].max())
        spans = []
        for _ in range(2):
            Xd, Ud, Td = rep(prob.X0), rep(prob.U0), rep(prob.T0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            std = ops.mintime_solve_torch(prob.model, d(prob.s), d(prob.kappa), d(left), d(right), prob.margin, prob.track_length,
                                          Xd, Ud, Td, prob.average_track_width, prob.speed_cap, max_iter=iters, tol=tol)
            e1.record()
            torch.cuda.synchronize()
            spans.append(e0.elapsed_time(e1) * 1e-3)
        leg["device_span_s"] = float(min(spans))
        leg["device_span_note"] = (f"rl_mintime_solve_batch_dev, {iters} iterations enqueued (the slowest instance's count; finished "
                                   f"instances return at once), events on the launch stream, no host copies; converged "
                                   f"{int((std[:, 5] == 1).sum().item())} of {B}")
    except Exception as e:   # the host-timed figure stands on its own
        leg["device_span_s"] = None
        leg["device_span_note"] = f"{type(e).__name__}: {e}"
    return leg
