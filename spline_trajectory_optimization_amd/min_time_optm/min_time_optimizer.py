"""Host-side mirror of `spline_traj_optm.min_time_optm.min_time_optimizer.set_up_double_track_problem`
(min_time_optm/min_time_optimizer.py:93-163) and of the solve the reference's CLI performs on it
(entrypoints/traj_opt_double_track.py:24-86), with the GPU solver behind it.

The reference builds a casadi.Opti object and calls IPOPT.  `set_up_double_track_problem(params)` here takes the same
`params` dictionary (N, model, race_track, traj_d, average_track_width, speed_cap, max_iter, tol, verbose, optional
x0 / u0 / t0) and returns what the reference returns (:163):

    (X, U, T), (scale_x, scale_u, scale_t), opti

`X`, `U`, `T` are handles of the decision-variable blocks (N x 6, N x 4, N) in the reference's SCALED variables,
`scale_*` the scalings of :109-113, and `opti` a facade with the slice of casadi.Opti's surface the reference's
callers use (entrypoints/traj_opt_double_track.py:58-69, tests/test_min_time_optm.py): `solve()` (raises when the
solver does not reach its tolerance, like casadi), `value(v)`, `debug.value(v)` (the last iterate, also after a
failed solve), `set_initial(v, values)`, `solver(...)`, `stats()`.  So the CLI's

    x = opti.debug.value(X) * scale_x + np.hstack([race_track.abscissa[:, np.newaxis], np.zeros((N, 5))])
    u = opti.debug.value(U) * scale_u;   t = opti.debug.value(T) * scale_t

give the physical solution.  Underneath sits `DoubleTrackProblem` (`opti.problem`), whose `solve()` /
`solve_batch()` run rl_mintime_solve_batch (include/rl_mincurv.h) in physical units -- the batched entry the
reference has no counterpart for.

Deviations from the reference, all in one place:
  * solver: IPOPT is not part of the reference tree nor of this image; the solve is the build's own interior-point
    SQP-type method on the GPU (DESIGN_HISTORY.md 3d).  Its `tol` is an ABSOLUTE bound on the KKT residuals in the reference's
    scaling, whereas IPOPT's `tol` (yaml :8, 0.1) bounds its SCALED NLP error (residuals divided by multiplier-dependent
    factors s_d, s_c >= 1, and only reached after the barrier parameter has come down with it) -- handing 0.1 straight
    through stopped seconds of lap time short of the optimum while reporting success (round-3 advisor finding).  The
    facade therefore maps IPOPT-style tolerances onto the solver's scale: `effective_tol = min(tol, IPOPT_TOL_CAP)`
    (1e-4: measured on the example, the lap is then within 1e-3 s of the 1e-6 lap), warns when it does so, and reports
    `requested_tol`, `effective_tol` and the IPOPT options it has no counterpart for in `stats()`.
    `DoubleTrackProblem.solve(tol=...)` takes the solver's own tolerance unchanged (default 1e-6).
  * interpolants: `race_track.left_intp / right_intp / curvature_intp` are periodic cubic splines through the same
    samples (models/race_track.py), not CasADi's not-a-knot `bspline` interpolants on the padded table: the NLP's
    data agree with the reference's to interpolation accuracy, not bit for bit.
  * initial guess: `params["initial_guess"]` = "reference" (default) is :146-151 as written -- speed of the QSS
    table unfloored, controls (1, -1, 0.001, 0) in physical units, `T[i] = TIME[i]` unshifted; "clipped" is the
    build's variant (speed floored at 1.5 m/s so that v >= 1 holds strictly, the unused brake control at its
    optimum 0, TIME rolled by one sample because fill_time stores a segment's time on its END point, floor 1e-3 s).
    `u[1]` enters only the objective (its optimum is 0) and is eliminated by the solver either way.
"""
import warnings

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
        else:
            mode = params.get("initial_guess", "reference")
            assert mode in ("reference", "clipped"), mode
            self.X0 = np.zeros((self.N, 6)); self.X0[:, 0] = self.s
            if mode == "reference":                                                         # :146-151 as written
                self.X0[:, 5] = pts[:, Trajectory.SPEED]
                self.U0 = np.tile(np.array([1.0, -1.0, 0.001, 0.0]), (self.N, 1))
                self.T0 = pts[:, Trajectory.TIME].copy()
            else:
                self.X0[:, 5] = np.maximum(pts[:, Trajectory.SPEED], 1.5)                   # strictly inside v >= 1 (:188 of double_track.py)
                self.U0 = np.tile(np.array([1.0, 0.0, 0.001, 0.0]), (self.N, 1))            # u[1] is unused; its cost optimum is 0
                # TIME sits on the END point of each segment (Trajectory.fill_time); T[j] is the interval that STARTS at node j
                self.T0 = np.maximum(np.roll(pts[:, Trajectory.TIME], -1), 1e-3)
        # the reference's variable scaling (:109-113)
        self.scale_x = np.array([[1.0, self.average_track_width, 1.0, 1.0, 0.5, self.speed_cap]])
        self.scale_u = np.array([[self.model["Fd_max"], abs(self.model["Fb_max"]), self.model["delta_max"],
                                  self.model["mass"] * 50.0]])
        self.scale_t = 1.0
        self.x_offset = np.zeros((self.N, 6)); self.x_offset[:, 0] = self.s                 # X_OFFSET, :106

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


class OptiVariable:
    """Handle of one block of decision variables -- what `opti.variable(...)` returns in the reference (:101-103)."""

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)

    def __repr__(self):
        return f"OptiVariable({self.name}, {self.shape})"


class OptiSolution:
    """What `opti.solve()` returns: `.value(v)` of the converged point, `.stats()`."""

    def __init__(self, values, stats):
        self._values, self._stats = values, stats

    def value(self, var):
        return self._values[var.name].copy()

    def stats(self):
        return dict(self._stats)


IPOPT_TOL_CAP = 1e-4   # the loosest absolute KKT tolerance an IPOPT-style `tol` is mapped to (module docstring)


class OptiFacade:
    """The slice of casadi.Opti the reference's callers of set_up_double_track_problem touch, over the GPU solve.
    All values are in the reference's SCALED variables (X * scale_x + X_OFFSET, U * scale_u, T * scale_t are physical)."""

    def __init__(self, problem, X, U, T):
        self.problem, self._vars = problem, {"X": X, "U": U, "T": T}
        p = problem
        self._values = {"X": (p.X0 - p.x_offset) / p.scale_x, "U": p.U0 / p.scale_u, "T": p.T0 / p.scale_t}
        self._stats = {"iter_count": 0, "return_status": "not solved", "success": False}
        self._max_iter = int(p.params.get("max_iter", 500))                                  # yaml :7; s_opts of :160
        self._tol = float(p.params.get("tol", 1e-6))
        self._ignored_opts = {}
        self.debug = self          # casadi: opti.debug.value(v) = the latest iterate, solved or not

    def effective_tol(self):
        """The absolute KKT tolerance handed to the GPU solver for the IPOPT-style tolerance set on this object."""
        return min(self._tol, IPOPT_TOL_CAP)

    def solver(self, name="ipopt", p_opts=None, s_opts=None):
        """casadi signature (:161); the plugin name is accepted and ignored -- the solver is the library's."""
        s_opts = s_opts or {}
        self._max_iter = int(s_opts.get("max_iter", self._max_iter))
        self._tol = float(s_opts.get("tol", self._tol))
        # IPOPT options without a counterpart in the library's solver (e.g. constr_viol_tol, the reference passes none
        # besides max_iter / tol, :158-160): kept and reported, never silently dropped
        self._ignored_opts = {k_: v for k_, v in s_opts.items() if k_ not in ("max_iter", "tol", "print_level")}

    def set_initial(self, var, value):
        v = np.asarray(value, dtype=np.float64)
        self._values[var.name] = np.broadcast_to(v.reshape(var.shape) if v.size == int(np.prod(var.shape)) else v,
                                                 var.shape).copy()

    def value(self, var):
        return self._values[var.name].copy()

    def stats(self):
        return dict(self._stats)

    def solve(self):
        p = self.problem
        X0 = self._values["X"] * p.scale_x + p.x_offset
        U0 = self._values["U"] * p.scale_u
        T0 = self._values["T"] * p.scale_t
        tol_eff = self.effective_tol()
        if tol_eff < self._tol:
            warnings.warn(f"IPOPT-style tol {self._tol:g} mapped to the GPU solver's absolute KKT tolerance {tol_eff:g} "
                          f"(min_time_optimizer.IPOPT_TOL_CAP); see stats()['effective_tol']", RuntimeWarning, stacklevel=2)
        if self._ignored_opts:
            warnings.warn(f"solver options without a counterpart in the GPU solver are ignored: {sorted(self._ignored_opts)}",
                          RuntimeWarning, stacklevel=2)
        X, U, T, st = p.solve_batch(p.left[None], p.right[None], X0[None], U0[None], T0[None],
                                    max_iter=self._max_iter, tol=tol_eff)
        self._values = {"X": (X[0] - p.x_offset) / p.scale_x, "U": U[0] / p.scale_u, "T": T[0] / p.scale_t}
        ok = st[0, 5] == 1.0
        self._stats = {"iter_count": int(st[0, 0]), "success": bool(ok),
                       "return_status": "Solve_Succeeded" if ok else
                       ("Maximum_Iterations_Exceeded" if st[0, 5] == 0.0 else "Solver_Failed"),
                       "dual_inf": float(st[0, 1]), "constr_viol": float(st[0, 2]), "compl": float(st[0, 3]),
                       "lap_time": float(st[0, 4]), "requested_tol": self._tol, "effective_tol": tol_eff,
                       "ignored_options": dict(self._ignored_opts), "raw": st[0].copy()}
        if not ok:   # casadi raises on anything but success; the iterate stays readable through opti.debug.value
            raise RuntimeError(f"Error in Opti::solve: solver returned '{self._stats['return_status']}' "
                               f"after {self._stats['iter_count']} iterations")
        return OptiSolution({k_: v.copy() for k_, v in self._values.items()}, self._stats)


def set_up_double_track_problem(params):
    """-> (X, U, T), (scale_x, scale_u, scale_t), opti   (min_time_optimizer.py:93-163, return at :163)."""
    prob = DoubleTrackProblem(params)
    X = OptiVariable("X", (prob.N, 6)); U = OptiVariable("U", (prob.N, 4)); T = OptiVariable("T", (prob.N,))
    opti = OptiFacade(prob, X, U, T)
    return (X, U, T), (prob.scale_x, prob.scale_u, prob.scale_t), opti


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
