from spline_trajectory_optimization_amd.optimization.optimizer import TrajectoryOptimizer  # noqa: F401
