"""Compatibility namespace: the reference's module paths (`spline_traj_optm.models.trajectory`, ...)
re-exported from `spline_trajectory_optimization_amd`, so that user scripts written against the
reference import unchanged and pickles written by the reference (`BSplineTrajectory.save`,
models/trajectory.py:303-305 -- they record the class as spline_traj_optm.models.trajectory.BSplineTrajectory)
load into the MI355X-backed classes.  No code of the reference lives here."""
