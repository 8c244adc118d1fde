from spline_trajectory_optimization_amd.models.race_track import RaceTrack  # noqa: F401
