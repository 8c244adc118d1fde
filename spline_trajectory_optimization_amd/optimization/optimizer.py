"""Host-side mirror of `spline_traj_optm.optimization.optimizer.TrajectoryOptimizer`
(optimization/optimizer.py:15-341): same constructor, the same six methods with the same argument
order, defaults and return types.  All arithmetic runs in HIP kernels behind the C ABI
(include/rl_mincurv.h); casadi/qpOASES and shapely are not needed.

Differences from the reference, all additive and keyword-only:
  * `i_start=` pins the sweep start indices the reference draws with an UNSEEDED
    np.random.randint (optimizer.py:303).  Without it the same np.random.randint calls are made,
    in the same order, so `np.random.seed(s)` reproduces the reference's order.
  * the per-iteration `Simulator.run_simulation` calls (optimizer.py:333-339) never feed back
    into the returned spline (they only rewrite SPEED/ACC/TIME columns of a table that is
    discarded); they are skipped unless `simulate=True`.
  * the matplotlib visualiser (optimizer.py:261, 331-340) is not constructed.
  * ARITHMETIC.  This class is the drop-in for the reference's, so by default it answers in the REFERENCE-ORDER arithmetic
    (`TrajectoryOptimizer.arith = _lib.ARITH_REFERENCE`: the reference's operations in numpy's / scipy's order, the CPU oracle's
    bits; include/rl_mincurv.h) wherever that exists (degree-5 splines: both drivers and the four assembly methods).  Set
    `optm.arith = _lib.ARITH_FAST` (or pass `arith=` to a driver) for the fast arithmetic; a single solve costs about
    6 ms in either.  (Since round 6 the batched entry points default to the reference-order arithmetic too.)
  * `run_global_min_curvature_qp` is NEW (no reference counterpart that works: the Julia notebook
    prototype): the whole line as ONE banded QP per linearisation, solved by the
    interior-point kernel (include/rl_mincurv.h: rl_mincurv_global_batch_*).
"""
import numpy as np

from .. import _lib, ops
from ..models.race_track import RaceTrack
from ..models.trajectory import BSplineTrajectory, Trajectory, _ring_coords
from ..models.vehicle import Vehicle
from ..simulator.simulator import Simulator


class TrajectoryOptimizer:
    arith = _lib.ARITH_REFERENCE     # see the module docstring; the batched ops' default (ARITH_DEFAULT) is the same arithmetic since round 6

    def _scope(self, k, arith=None):
        """The arithmetic the calls inside run in: `arith` if given, else this object's; the fast one where the
        reference-order kernels do not exist (degree != 5)."""
        a = self.arith if arith is None else arith
        if k != 5:
            a = _lib.ARITH_FAST
        return _lib.Context.get().arith(a)

    def __init__(self, race_track: RaceTrack, center_line: BSplineTrajectory, vehicle: Vehicle) -> None:
        self.track = race_track
        self.left_bound = race_track.left_s
        self.right_bound = race_track.right_s
        self.center_line = center_line
        self.vehicle = vehicle
        self.sim = Simulator(self.vehicle)
        self._trk_cache = {}

    # ---- device tables for (knots, degree, N); control points are refreshed on every call
    def _device_track(self, traj_s: BSplineTrajectory, N: int):
        t, cx, cy, k = traj_s._tck()
        key = (t.tobytes(), k, int(N))
        trk = self._trk_cache.get(key)
        if trk is None:
            trk = _lib.Track(_lib.Context.get(), t, cx, cy, k, N)
            if len(self._trk_cache) > 8:
                self._trk_cache.clear()
            self._trk_cache[key] = trk
        else:
            trk.set_control_points(cx, cy)
        return trk

    def min_curvature_cost(self, z: np.ndarray, idx: int, traj_s: BSplineTrajectory, traj_d: Trajectory):
        """optimizer.py:24-86 -> k_cost.  Returns H (2x2), g (2)."""
        trk = self._device_track(traj_s, len(traj_d))
        with self._scope(traj_s._spl_x.k):
            H, g, _ = ops.mincurv_cost(trk, [idx], z=np.asarray(z, dtype=np.float64).reshape(1, 2))
        return H[0], g[0]

    def joint_min_curvature_cost(self, traj_s: BSplineTrajectory, traj_d: Trajectory, start_idx=None, span=None):
        """optimizer.py:88-110: the 2x2 blocks of the window's control points on the diagonal."""
        num_ctrl_pt = len(traj_s._spl_x.c)
        ignore_front = traj_s._spl_x.k // 2
        ignore_rear = traj_s._spl_x.k - ignore_front
        if start_idx is None:
            i_min, i_max = ignore_front, num_ctrl_pt - ignore_rear
        else:
            i_min, i_max = start_idx, start_idx + span
        nw = i_max - i_min
        trk = self._device_track(traj_s, len(traj_d))
        with self._scope(traj_s._spl_x.k):
            H, g, _ = ops.mincurv_cost(trk, np.arange(i_min, i_max))  # z = current control points
        joint_H = np.zeros((nw * 2, nw * 2), np.float64)
        joint_g = np.zeros(nw * 2, np.float64)
        for j in range(nw):
            joint_H[2 * j:2 * j + 2, 2 * j:2 * j + 2] = H[j]
            joint_g[2 * j:2 * j + 2] = g[j]
        return joint_H, joint_g

    def track_constraint(self, idx: int, traj_s: BSplineTrajectory, traj_d: Trajectory):
        """optimizer.py:222-254 -> k_constraint.  Returns A (2M x 2), lba (2M), uba (2M)."""
        trk = self._device_track(traj_s, len(traj_d))
        with self._scope(traj_s._spl_x.k):
            return ops.track_constraint(trk, traj_d.points, idx)

    def joint_track_constraint(self, traj_s: BSplineTrajectory, traj_d: Trajectory, start_idx=None, span=None):
        """optimizer.py:112-161: dense A over ALL samples for the window's control points."""
        num_ctrl_pt = len(traj_s._spl_x.c)
        k = traj_s._spl_x.k
        ignore_front = k // 2
        ignore_rear = k - ignore_front
        if start_idx is None:
            i_min, i_max = ignore_front, num_ctrl_pt - ignore_rear
        else:
            i_min, i_max = start_idx, start_idx + span
        nw = i_max - i_min
        M = len(traj_d)
        ts = traj_d.ts()
        trk = self._device_track(traj_s, M)
        joint_A = np.zeros((2 * M, 2 * nw), dtype=np.float64)
        z = np.zeros(nw * 2, np.float64)
        for i in range(i_min, i_max):
            j = i - i_min
            with self._scope(k):
                A, _, _ = ops.track_constraint(trk, traj_d.points, i)
            t_mask = (ts >= traj_s._spl_x.t[i]) & (ts < traj_s._spl_x.t[i + k + 1])
            s0 = int(np.argmax(t_mask)) if t_mask.any() else 0
            m = len(A) // 2
            joint_A[2 * s0:2 * (s0 + m):2, 2 * j] = A[0::2, 0]
            joint_A[2 * s0 + 1:2 * (s0 + m):2, 2 * j + 1] = A[1::2, 1]
            z[2 * j:2 * j + 2] = np.array(traj_s.get_control_point(i))
        non_z = traj_d[:, :2] - (joint_A @ z).reshape((-1, 2))
        left_x = traj_d[:, Trajectory.LEFT_BOUND_X]; left_y = traj_d[:, Trajectory.LEFT_BOUND_Y]
        right_x = traj_d[:, Trajectory.RIGHT_BOUND_X]; right_y = traj_d[:, Trajectory.RIGHT_BOUND_Y]
        min_bound = np.empty(2 * M, dtype=np.float64)
        max_bound = np.empty(2 * M, dtype=np.float64)
        min_bound[0::2] = np.minimum(left_x, right_x) - non_z[:, 0]
        min_bound[1::2] = np.minimum(left_y, right_y) - non_z[:, 1]
        max_bound[0::2] = np.maximum(left_x, right_x) - non_z[:, 0]
        max_bound[1::2] = np.maximum(left_y, right_y) - non_z[:, 1]
        return joint_A, min_bound, max_bound

    def run_min_curvature_qp(self, traj_in_s: BSplineTrajectory, traj_in_d: Trajectory, visualize=False,
                             max_iter=5, *, i_start=None, simulate=False, arith=None):
        """optimizer.py:256-341 -> ONE launch of the LDS-resident sweep kernel (k_sweep).

        simulate=True reproduces what the reference prints: the simulator runs on the table after EVERY outer iteration
        ("Iteration j" + the result, :333-336; its speeds never feed back into the spline) and once more at the end
        (:338-339).  The table after outer iteration j is that of a launch over the first j + 1 iterations FROM THE START
        (max_iter launches of growing length): every launch is then a prefix of the one real run, so numpy's error state (raise
        mode from the end of iteration 0 on, simulator.py:164) and a stale table carry over exactly as in the single launch --
        continuing launch by launch from the control points would lose both (fixture G12).  arith: _lib.ARITH_FAST / ARITH_REFERENCE for this call (None = this object's `arith`: the reference-order arithmetic unless changed)."""
        traj_out_s = traj_in_s.copy()
        n = len(traj_out_s._spl_x.c)
        k = traj_out_s._spl_x.k
        i_min, i_max = k // 2, n - (k - k // 2)
        if i_start is None:
            i_start = []
            for _ in range(max_iter):
                i_start.append(int(np.random.randint(i_min, i_max)))  # optimizer.py:303
                print(f'Starting from {i_start[-1]}-th control point.')
        i_start = np.asarray(i_start, dtype=np.int32)
        assert len(i_start) == max_iter
        trk = self._device_track(traj_out_s, len(traj_in_d))
        trk.set_rings(_ring_coords(self.track.left_r), _ring_coords(self.track.right_r))
        trk.set_length(traj_out_s.get_length())
        t, cx, cy, _ = traj_out_s._tck()
        arith_ = (self.arith if arith is None else arith) if k == 5 else _lib.ARITH_FAST

        def table(pts):                                     # the traj_out_d of optimizer.py:286-288
            tab = Trajectory(len(pts))
            tab.points = pts
            return tab
        if simulate:
            ns_all = []
            self.last_sim_results = []
            cx0, cy0 = cx, cy
            for j in range(max_iter):
                cx, cy, pts, ns, stats = ops.mincurv_sweep(trk, cx0, cy0, i_start[:j + 1], want_points=True, arith=arith_)
                ns_all.append(ns[j])
                print(f"Forward pass: number of control points successfully updated: {ns[j, 0]}")
                print(f"Backward pass: number of control points successfully updated: {ns[j, 1]}")
                sim_result = self.sim.run_simulation(table(pts), enable_vis=False)          # :333
                print(f"Iteration {j+1}")
                print(sim_result)
                self.last_sim_results.append(sim_result)
            ns = np.array(ns_all)
            sim_result = self.sim.run_simulation(sim_result.trajectory, enable_vis=False)   # :338-339
            print(sim_result)
            self.last_sim_results.append(sim_result)
        else:
            cx, cy, pts, ns, stats = ops.mincurv_sweep(trk, cx, cy, i_start, want_points=True, arith=arith_)
            for j in range(max_iter):
                print(f"Forward pass: number of control points successfully updated: {ns[j, 0]}")
                print(f"Backward pass: number of control points successfully updated: {ns[j, 1]}")
        traj_out_s._spl_x.c[:] = cx
        traj_out_s._spl_y.c[:] = cy
        self.last_n_success = ns
        self.last_stats = stats
        self.last_table = table(pts)
        return traj_out_s

    def run_joint_min_curvature_qp(self, traj_in_s: BSplineTrajectory, traj_in_d: Trajectory, max_iter=3,
                                   visualize=False, *, i_start=None, arith=None):
        """optimizer.py:163-220 (sliding window, span 5) -> ONE launch of the joint sweep kernel.
        The per-window simulator call (:208-209) never feeds back into the spline and is not run."""
        traj_out_s = traj_in_s.copy()
        n = len(traj_out_s._spl_x.c)
        k = traj_out_s._spl_x.k
        span = 5
        i_min, i_max = k // 2, n - (k - k // 2) - span
        if i_start is None:
            i_start = []
            for _ in range(max_iter):
                i_start.append(int(np.random.randint(i_min, i_max)))  # optimizer.py:178
                print(f'Starting from {i_start[-1]}-th control point.')
        i_start = np.asarray(i_start, dtype=np.int32)
        assert len(i_start) == max_iter
        trk = self._device_track(traj_out_s, len(traj_in_d))
        trk.set_rings(_ring_coords(self.track.left_r), _ring_coords(self.track.right_r))
        trk.set_length(traj_out_s.get_length())
        t, cx, cy, _ = traj_out_s._tck()
        arith_ = (self.arith if arith is None else arith) if k == 5 else _lib.ARITH_FAST
        if arith_ == _lib.ARITH_BRANCH:
            arith_ = _lib.ARITH_FAST      # the branch arithmetic exists for the sweep only
        cx, cy, pts, ns, stats = ops.mincurv_sweep_joint(trk, cx, cy, i_start, want_points=True, arith=arith_)
        traj_out_s._spl_x.c[:] = cx
        traj_out_s._spl_y.c[:] = cy
        self.last_n_success = ns
        self.last_stats = stats
        return traj_out_s

    def run_global_min_curvature_qp(self, traj_in_s: BSplineTrajectory, traj_in_d: Trajectory, margin=0.0,
                                    n_outer=6):
        """NEW, additive: global min-curvature QP.  The control points of
        `traj_in_s` slide along the normals of that line; every sample of the result stays within the
        half-widths |p - L|, |p - R| that `traj_in_d`'s bound columns give it, minus `margin` metres.
        Returns the optimised BSplineTrajectory; `last_global_stats` = [interior-point iterations,
        sum kappa^2 before, after, largest bound violation [m], last Gauss-Newton step [m],
        samples on a bound (active set), 0, 0]."""
        from .. import batch
        traj_out_s = traj_in_s.copy()
        trk = self._device_track(traj_out_s, len(traj_in_d))
        wl, wr = batch.half_widths_from_bounds(traj_in_d.points)
        ctrl, xy, a, st, stats = ops.global_batch_host(trk, np.stack([wl, wr], axis=1)[None], margin, n_outer,
                                                       want_xy=False)
        traj_out_s._spl_x.c[:] = ctrl[0, :, 0]
        traj_out_s._spl_y.c[:] = ctrl[0, :, 1]
        self.last_global_stats = st[0]
        self.last_stats = stats
        return traj_out_s
