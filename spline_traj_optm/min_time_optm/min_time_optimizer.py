from spline_trajectory_optimization_amd.min_time_optm.min_time_optimizer import (  # noqa: F401
    DoubleTrackProblem, set_up_double_track_problem)
