"""MI355X-native batched min-curvature racing-line solver.

Drop-in for the min-curvature hot path of HaoruXue/spline-trajectory-optimization: the module layout
mirrors `spline_traj_optm` (models.trajectory, models.race_track, models.vehicle,
optimization.optimizer, simulator.simulator); the arithmetic runs in hand-written HIP kernels for
gfx950 behind the C ABI declared in include/rl_mincurv.h.  No CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib", "ops", "batch"]
