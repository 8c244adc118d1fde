"""GPU: every instance of the benchmarked batch (Monza N = 2000, 1024 width-perturbed instances, 6 linearisations) through both
global-QP formulations, against the CPU twins run on the host cores: interior-point iteration counts and the largest deviation of
the line.  Writes gpurun_out/<tag>_global_full_batch.json."""
import json, os, sys, time
import numpy as np
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa
from test_global_qp import monza_widths  # noqa
from oracle import oracle as orc  # noqa

B, N, OUTER, MARGIN, LON = 1024, 2000, 6, 0.25, 1.0
fits = golden("G1_spline_fits.npz")
t, cx, cy, k, u, wl, wr = monza_widths(fits, "c100", N)


def widths():
    from spline_trajectory_optimization_amd import batch
    return batch.width_batch(wl, wr, B, seed=1234)


W = widths()


def twin1(b):
    r = orc.global_mincurv(t, cx, cy, k, N, W[b, :, 0], W[b, :, 1], MARGIN, OUTER)
    return r[2], r[4][0]


def twin2(b):
    r = orc.global_mincurv_xy(t, cx, cy, k, N, W[b, :, 0], W[b, :, 1], MARGIN, LON, OUTER)
    return r[2], r[4][0]


if __name__ == "__main__":
    tag = sys.argv[1] if len(sys.argv) > 1 else "full"
    orc.build()
    t0 = time.time()
    with Pool(min(16, os.cpu_count() or 1)) as p:
        r1 = p.map(twin1, range(B)); r2 = p.map(twin2, range(B))
    print(f"twins: {time.time() - t0:.1f} s", flush=True)
    from spline_trajectory_optimization_amd import _lib, ops
    trk = _lib.Track(_lib.Context.get(0), t, cx, cy, k, N)
    out = {}
    for name, dof, rr in (("one_offset_per_control_point", 1, r1), ("both_coordinates_free", 2, r2)):
        ctrl, xy, a, st, rs = ops.global_batch_host(trk, W, MARGIN, OUTER, dof=dof, lon=LON)
        dev = np.array([np.abs(xy[b] - rr[b][0]).max() for b in range(B)])
        dits = st[:, 0] - np.array([rr[b][1] for b in range(B)])
        out[name] = {"kernel_ms": rs.kernel_ms, "ipm_iterations_gpu_mean_max": [float(st[:, 0].mean()), float(st[:, 0].max())],
                     "ipm_iterations_twin_mean_max": [float(np.mean([x[1] for x in rr])), float(np.max([x[1] for x in rr]))],
                     "instances_with_equal_iteration_count": int((dits == 0).sum()), "largest_iteration_difference": float(np.abs(dits).max()),
                     "line_deviation_m": {"max": float(dev.max()), "median": float(np.median(dev)), "above_1e-6": int((dev > 1e-6).sum()),
                                          "above_2e-6": int((dev > 2e-6).sum()), "worst_instances": [int(i) for i in np.argsort(dev)[-5:]]}}
        print(name, json.dumps(out[name]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_global_full_batch.json"), "w"), indent=1)
