"""Host-side mirror of `spline_traj_optm.min_time_optm.min_time_optimizer.set_up_double_track_problem`
(min_time_optm/min_time_optimizer.py:93-163) and of the solve the reference's CLI performs on it
(entrypoints/traj_opt_double_track.py:24-86), with the GPU solver behind it.

The reference builds a casadi.Opti object and calls IPOPT; here `set_up_double_track_problem(params)`
takes the same `params` dictionary (N, model, race_track, traj_d, average_track_width, speed_cap, max_iter,
tol, optional x0 / u0 / t0) and returns a `DoubleTrackProblem` whose `solve()` runs
rl_mintime_solve_batch (include/rl_mincurv.h) and returns physical X [N,6], U [N,4], T [N] -- what the CLI
reads back through `opti.debug.value(...) * scale + offset` (:66-69).  `solve_batch` runs many instances
(different boundary widths and / or initial guesses) in one call.
"""
import numpy as np

from .. import ops
from ..models.trajectory import Trajectory


class DoubleTrackProblem:
    def __init__(self, params):
        self.params = params
        self.model = dict(params["model"])
        rt = params["race_track"]
        traj_d = params["traj_d"]
        self.N = int(params.get("N", len(traj_d)))
        assert self.N == len(traj_d) == len(rt.abscissa)
        self.s = np.ascontiguousarray(rt.abscissa, dtype=np.float64)                       # S0, :105
        self.left = np.asarray(rt.left_intp(self.s), dtype=np.float64)                      # BoundL, :110
        self.right = np.asarray(rt.right_intp(self.s), dtype=np.float64)                    # BoundR, :111
        self.kappa = np.asarray(rt.curvature_intp(self.s), dtype=np.float64)                # :139
        self.track_length = float(rt.center_s.get_length())
        self.margin = self.model["vehicle_width"] / 2.0 + self.model["safety_margin"]        # :134
        bad = np.where(~(self.right + self.margin < self.left - self.margin))[0]
        assert len(bad) == 0, f"Track width must be wider than vehicle width plus 2 * safety margin at point {bad[0]}."
        self.average_track_width = float(params["average_track_width"])
        self.speed_cap = float(params["speed_cap"])
        pts = traj_d.points if isinstance(traj_d, Trajectory) else np.asarray(traj_d)
        if "x0" in params:                                                                  # :152-155
            self.X0 = np.array(params["x0"], dtype=np.float64).reshape(self.N, 6)
            self.U0 = np.array(params["u0"], dtype=np.float64).reshape(self.N, 4)
            self.T0 = np.array(params["t0"], dtype=np.float64).reshape(self.N)
        else:                                                                               # :146-151
            self.X0 = np.zeros((self.N, 6)); self.X0[:, 0] = self.s
            self.X0[:, 5] = np.maximum(pts[:, Trajectory.SPEED], 1.5)                       # strictly inside v >= 1 (:188 of double_track.py)
            self.U0 = np.tile(np.array([1.0, 0.0, 0.001, 0.0]), (self.N, 1))                # u[1] is unused; its cost optimum is 0
            # TIME sits on the END point of each segment (Trajectory.fill_time); T[j] is the interval that STARTS at node j
            self.T0 = np.maximum(np.roll(pts[:, Trajectory.TIME], -1), 1e-3)

    def solve(self, max_iter=None, tol=None):
        """-> (X [N,6], U [N,4], T [N], stats [12]); stats as include/rl_mincurv.h: rl_mintime_solve_batch."""
        X, U, T, st = self.solve_batch(self.left[None], self.right[None], max_iter=max_iter, tol=tol)
        return X[0], U[0], T[0], st[0]

    def solve_batch(self, left, right, X0=None, U0=None, T0=None, max_iter=None, tol=None):
        """B instances that differ in their boundary distances left/right [B,N] (and optionally in their initial
        guesses).  The reference's IPOPT tolerance `tol` (yaml :8) is a loose 0.1; the default here is 1e-6."""
        left = np.ascontiguousarray(left, dtype=np.float64); right = np.ascontiguousarray(right, dtype=np.float64)
        B = left.shape[0]
        rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)  # noqa: E731
        X0 = rep(self.X0) if X0 is None else X0
        U0 = rep(self.U0) if U0 is None else U0
        T0 = rep(self.T0) if T0 is None else T0
        return ops.mintime_solve_batch(self.model, self.s, self.kappa, left, right, self.margin, self.track_length,
                                       X0, U0, T0, self.average_track_width, self.speed_cap,
                                       max_iter=int(max_iter or self.params.get("max_iter", 200)),
                                       tol=float(tol if tol is not None else 1e-6))


def set_up_double_track_problem(params):
    return DoubleTrackProblem(params)


def optimise_track(race_track, vehicle, model, average_track_width=7.0, speed_cap=30.0, max_iter=200, tol=1e-6):
    """The reference CLI's pipeline (entrypoints/traj_opt_double_track.py:24-86) as a function: centre-line table
    with bounds -> QSS warm start (Simulator, k_qss_sim) -> NLP solve (k_mt_*) -> optimised Trajectory table with
    X, Y, YAW, SPEED, bounds and distances filled, ready for save_ttl.  Returns (Trajectory, X, U, T, stats)."""
    from ..simulator.simulator import Simulator
    traj_d = race_track.center_d.copy()
    race_track.fill_trajectory_boundaries(traj_d)
    traj_d = Simulator(vehicle).run_simulation(traj_d, False).trajectory                    # :35-49
    prob = DoubleTrackProblem({"N": len(traj_d), "model": model, "race_track": race_track, "traj_d": traj_d,
                               "average_track_width": average_track_width, "speed_cap": speed_cap, "max_iter": max_iter})
    X, U, T, st = prob.solve(tol=tol)
    out = traj_d.copy()
    pose = race_track.frenet_to_global(X[:, 0], X[:, 1], X[:, 2])                           # :76
    out[:, 0:2] = pose[:, 0:2]
    out[:, Trajectory.YAW] = pose[:, 2]
    out[:, Trajectory.SPEED] = X[:, 5]
    race_track.fill_trajectory_boundaries(out)
    out.fill_distance()                                                                     # :81
    return out, X, U, T, st
