from spline_trajectory_optimization_amd.models.trajectory import (  # noqa: F401
    Bound, BSplineTrajectory, Region, Trajectory, load_ttl, save_ttl)
