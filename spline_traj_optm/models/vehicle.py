from spline_trajectory_optimization_amd.models.vehicle import Vehicle, VehicleParams  # noqa: F401
