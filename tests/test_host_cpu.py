"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, fails loudly without a device, and the host logic (generators, sharding, gather over
gloo with world_size 2) is right."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_hip()
    from spline_trajectory_optimization_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "rl_mincurv.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rl_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 15
    assert declared == set(built.EXPORTED_SYMBOLS), declared ^ set(built.EXPORTED_SYMBOLS)
    lib = built.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rl_version() == 103   # include/rl_mincurv.h: RL_VERSION


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.RlError, match="no HIP device"):
        built.Context(0)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing in the shipped package may import or include it."""
    pkg = os.path.join(ROOT, "spline_trajectory_optimization_amd")
    pat_py = re.compile(r"^\s*(from|import)\s+\S*oracle", re.M)
    pat_c = re.compile(r"#\s*include\s*[<\"][^>\"]*oracle", re.M)
    n = 0
    for dp, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dp, f)
            if f.endswith(".py"):
                assert not pat_py.search(open(path).read()), path
                n += 1
            elif f.endswith((".hip", ".hpp", ".h", ".cpp")):
                assert not pat_c.search(open(path).read()), path
                n += 1
    assert n >= 8


def test_width_batch_generator():
    from spline_trajectory_optimization_amd import batch
    wl = np.linspace(2.0, 8.0, 50); wr = np.full(50, 5.0)
    a = batch.width_batch(wl, wr, 16, seed=1234)
    b = batch.width_batch(wl, wr, 16, seed=1234)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (16, 50, 2) and a.min() >= 1.5
    ratio = a[:, :, 1] / 5.0
    assert np.all(np.abs(ratio - 1.0) <= 0.15 + 1e-12)
    assert np.allclose(ratio, ratio[:, :1])          # one epsilon per instance and side
    assert not np.array_equal(a, batch.width_batch(wl, wr, 16, seed=5678))


def test_i_start_range_and_shard_ranges():
    from spline_trajectory_optimization_amd import batch
    s = batch.default_i_start(66, 5, 50, seed=3)
    assert s.min() >= 2 and s.max() < 63           # optimizer.py:301-303: randint(2, n-3)
    for B in (8192, 1000, 7):
        for W in (1, 2, 3, 8):
            r = [batch.shard_range(B, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[i][1] == r[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


_GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RL_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RL_ROOT"], "tests"))
from spline_trajectory_optimization_amd import batch
from oracle import oracle as orc
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.path.join(os.environ["RL_ROOT"], "tests", "golden", "G1_spline_fits.npz"))
t, cx, cy, k, L = g["c100_t"], g["c100_cx"], g["c100_cy"], int(g["c100_k"]), float(g["c100_length"])
N, B = 64, 5
widths = batch.width_batch(np.full(N, 5.0), np.full(N, 4.0), B, seed=1)
i_start = batch.default_i_start(len(cx), k, 1, seed=0)
lo, hi = batch.shard_range(B, rank, world)
# the oracle stands in for the GPU solve here: this test covers the sharding + gather plumbing
ctrl, xy, ns = orc.solve_width_batch(t, cx, cy, k, L, N, widths[lo:hi], i_start)
full = batch.gather_to_root(torch.from_numpy(xy), rank, world, dist, total=B)
if rank == 0:
    _, ref, _ = orc.solve_width_batch(t, cx, cy, k, L, N, widths, i_start)
    assert full.shape == (B, N, 2), full.shape
    assert np.array_equal(full.numpy(), ref)
    print("GLOO_OK")
else:
    assert full is None
# the weak-scaling form bench.py uses: equal shards, preallocated destination, asynchronous, double buffered
mine = [torch.full((3, N, 2), float(10 * s_ + rank), dtype=torch.float64) for s_ in range(4)]
dst = [torch.zeros((world * 3, N, 2), dtype=torch.float64) for _ in range(2)] if rank == 0 else [None, None]
works = [None, None]
for s_ in range(4):
    q = s_ & 1
    if works[q] is not None:
        works[q].wait()
        if rank == 0:
            assert all(float(dst[q][3 * r_, 0, 0]) == 10 * (s_ - 2) + r_ for r_ in range(world))
    works[q] = batch.gather_to_root(mine[s_], rank, world, dist, out=dst[q], async_op=True)
for q in range(2):
    works[q].wait()
if rank == 0:
    assert all(float(dst[1][3 * r_, 5, 1]) == 30 + r_ for r_ in range(world))
    assert all(float(dst[0][3 * r_ + 2, 5, 1]) == 20 + r_ for r_ in range(world))
    print("ASYNC_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_solve_and_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, RL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_OK" in out.stdout and "ASYNC_OK" in out.stdout


_GLOO_WORKER8 = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["RL_ROOT"])
from spline_trajectory_optimization_amd import batch
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
# BASELINE configs[2] runs on 8 ranks: an UNEVEN split (B = 1000 over 8 -> shards of 125; B = 1003 -> 126/125) through
# gather_to_root(total=...), every instance tagged with its global index
# ... through dist.gather AND through the grouped isend / irecv form of the same collective (what "auto" falls back to
# when a backend refuses gather)
for method in ("gather", "p2p"):
    for B, N in ((1000, 16), (1003, 8), (5, 4)):
        lo, hi = batch.shard_range(B, rank, world)
        mine = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1, 1).repeat(1, N, 2) if hi > lo else torch.zeros((0, N, 2), dtype=torch.float64)
        full = batch.gather_to_root(mine, rank, world, dist, total=B, method=method)
        if rank == 0:
            assert full.shape == (B, N, 2), full.shape
            assert torch.equal(full[:, 0, 0], torch.arange(B, dtype=torch.float64)) and torch.equal(full[:, N - 1, 1], full[:, 0, 0])
        else:
            assert full is None
    # integer control data (status words) through the same gather
    lo, hi = batch.shard_range(1000, rank, world)
    st = batch.gather_to_root(torch.arange(lo, hi, dtype=torch.int32), rank, world, dist, total=1000, method=method)
    if rank == 0:
        assert torch.equal(st, torch.arange(1000, dtype=torch.int32))
    # equal shards into a preallocated tensor, asynchronously (bench.py's use)
    mine = torch.full((3, 4, 2), float(rank), dtype=torch.float64)
    dst = torch.empty((world * 3, 4, 2), dtype=torch.float64) if rank == 0 else None
    w = batch.gather_to_root(mine, rank, world, dist, out=dst, async_op=True, method=method)
    w.wait()
    if rank == 0:
        assert torch.equal(dst[:, 0, 0], torch.arange(world, dtype=torch.float64).repeat_interleave(3))
# "auto": a backend whose gather raises is served by the grouped form from then on
real_gather = dist.gather
def refusing(*a, **k):
    raise RuntimeError("gather not supported on this backend (test)")
dist.gather = refusing
full = batch.gather_to_root(torch.full((2, 1), float(rank), dtype=torch.float64), rank, world, dist)
dist.gather = real_gather
if rank == 0:
    assert torch.equal(full[:, 0], torch.arange(world, dtype=torch.float64).repeat_interleave(2))
    assert batch._gather_fallback["p2p"]
    print("GLOO8_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_uneven_shards_and_gather_gloo_world8(tmp_path):
    """The world size BASELINE configs[2] names (8 ranks): uneven contiguous shards, empty shards (B < world), float and
    integer payloads, all through batch.gather_to_root on gloo."""
    script = tmp_path / "worker8.py"
    script.write_text(_GLOO_WORKER8)
    env = dict(os.environ, RL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO8_OK" in out.stdout


def test_g11_sectional_lengths():
    """eval_sectional_length / eval_dx_sectional_length / eval_dy_sectional_length (reference trajectory.py:232-245;
    host quadrature in the reference and here) against values the reference's own methods produced (fixture G11)."""
    import pickle
    from spline_trajectory_optimization_amd.models.trajectory import BSplineTrajectory
    from scipy import interpolate
    g = np.load(os.path.join(ROOT, "tests", "golden", "G11_sectional_lengths.npz"))
    f = np.load(os.path.join(ROOT, "tests", "golden", "G1_spline_fits.npz"))
    for tag in ("c100", "l10"):
        sp = BSplineTrajectory.__new__(BSplineTrajectory)
        k = int(f[f"{tag}_k"])
        sp._spl_x = interpolate.BSpline(f[f"{tag}_t"], f[f"{tag}_cx"], k)
        sp._spl_y = interpolate.BSpline(f[f"{tag}_t"], f[f"{tag}_cy"], k)
        for q, iv in enumerate(g["intervals"]):
            assert abs(sp.eval_sectional_length(tuple(iv)) - g[f"{tag}_len"][q]) <= 1e-9 * max(1.0, abs(g[f"{tag}_len"][q]))
            assert abs(sp.eval_dx_sectional_length(tuple(iv)) - g[f"{tag}_dx"][q]) <= 1e-9 * max(1.0, abs(g[f"{tag}_dx"][q]))
            assert abs(sp.eval_dy_sectional_length(tuple(iv)) - g[f"{tag}_dy"][q]) <= 1e-9 * max(1.0, abs(g[f"{tag}_dy"][q]))


def test_ttl_and_csv_formats_roundtrip(tmp_path, built):
    """Wire formats of the reference (models/trajectory.py:202-209, 312-358): Trajectory CSV and the
    TTL file (header ttl_num,N,length,origin + 17 columns per row) -- host-side, no GPU involved."""
    from spline_trajectory_optimization_amd.models.trajectory import Trajectory, load_ttl, save_ttl
    rng = np.random.default_rng(0)
    tr = Trajectory(7, ttl_num=3, origin=(45.6, 9.28, 180.0))
    tr.points[:, :17] = rng.normal(size=(7, 17))
    tr.points[:, Trajectory.REGION] = [0, 1, 2, 0, 1, 2, 3]
    p = tmp_path / "a.csv"
    Trajectory.save(str(p), tr)
    back = Trajectory.load(str(p))
    np.testing.assert_allclose(back.points, tr.points, rtol=0, atol=0)
    q = tmp_path / "a.ttl"
    save_ttl(str(q), tr)
    lines = open(q).read().splitlines()
    assert lines[0].split(",")[:2] == ["3", "7"] and len(lines) == 8 and len(lines[1].split(",")) == 17
    ttl = load_ttl(str(q))
    assert ttl.ttl_num == 3 and ttl.origin == (45.6, 9.28, 180.0) and len(ttl) == 7
    np.testing.assert_allclose(ttl.points[:, :17], tr.points[:, :17], rtol=0, atol=0)
    assert np.all(ttl.points[:, 17] == np.arange(7)) and np.all(ttl.points[:, 18] == -1)
    # ring helper: a closed polyline keeps one copy of the closing vertex, like shapely's LinearRing
    from spline_trajectory_optimization_amd.models.race_track import Ring
    r = Ring(np.array([[0.0, 0], [1, 0], [1, 1], [0, 0]]))
    assert r.vertices.shape == (3, 2) and r.coords.shape == (4, 2)


def test_g10_files_written_by_the_reference(tmp_path, built):
    """f3 against REFERENCE-WRITTEN files (fixture G10: save_ttl, Trajectory.save and BSplineTrajectory.save
    of models/trajectory.py:202-209, 303-358 executed in the build container): the mirror writes the same
    bytes, reads the same values, and the reference's pickle loads through the `spline_traj_optm`
    compatibility namespace into the MI355X-backed class."""
    import os
    from conftest import GOLDEN, golden
    from spline_traj_optm.models.trajectory import BSplineTrajectory, Trajectory, load_ttl, save_ttl
    import spline_trajectory_optimization_amd.models.trajectory as mirror
    assert BSplineTrajectory is mirror.BSplineTrajectory and Trajectory is mirror.Trajectory
    g = golden("G10_formats.npz")
    tab = Trajectory(len(g["table"]), int(g["ttl_num"]), tuple(g["origin"]))
    tab.points = g["table"].copy()
    save_ttl(str(tmp_path / "a.ttl"), tab)
    Trajectory.save(str(tmp_path / "a.csv"), tab)
    assert open(tmp_path / "a.ttl", "rb").read() == open(os.path.join(GOLDEN, "G10_ref.ttl"), "rb").read()
    assert open(tmp_path / "a.csv", "rb").read() == open(os.path.join(GOLDEN, "G10_ref_table.csv"), "rb").read()
    back = load_ttl(os.path.join(GOLDEN, "G10_ref.ttl"))
    np.testing.assert_array_equal(back.points, g["ttl_loaded"])
    assert back.ttl_num == int(g["ttl_loaded_num"]) and tuple(back.origin) == tuple(g["ttl_loaded_origin"])
    np.testing.assert_array_equal(Trajectory.load(os.path.join(GOLDEN, "G10_ref_table.csv")).points, g["table"])
    sp = BSplineTrajectory.load(os.path.join(GOLDEN, "G10_ref_spline.pkl"))     # pickled by the reference's class
    assert isinstance(sp, mirror.BSplineTrajectory)
    f = golden("G1_spline_fits.npz")
    np.testing.assert_array_equal(sp._spl_x.t, f["c100_t"]); np.testing.assert_array_equal(sp._spl_x.c, f["c100_cx"])
    np.testing.assert_array_equal(sp._spl_y.c, f["c100_cy"]); assert sp.get_length() == float(f["c100_length"])
    cp = sp.copy(); cp.set_control_point(7, (1.0, 2.0))
    assert cp.get_control_point(7) == (1.0, 2.0) and sp.get_control_point(7) != (1.0, 2.0)
    # the other module paths of the reference resolve too
    from spline_traj_optm.models.race_track import RaceTrack  # noqa: F401
    from spline_traj_optm.models.vehicle import Vehicle, VehicleParams  # noqa: F401
    from spline_traj_optm.optimization.optimizer import TrajectoryOptimizer  # noqa: F401
    from spline_traj_optm.simulator.simulator import Simulator  # noqa: F401


def test_set_up_double_track_problem_returns_the_reference_contract():
    """`(X, U, T), (scale_x, scale_u, scale_t), opti = set_up_double_track_problem(params)`
    (min_time_optm/min_time_optimizer.py:163; consumed at entrypoints/traj_opt_double_track.py:56-69): handles,
    scalings of :109-113, and an opti facade whose values are the reference's SCALED variables; the default initial
    guess is :146-151 as written.  No GPU: solve() must fail loudly (no CPU fallback), the guess stays readable."""
    import types
    from spline_traj_optm.min_time_optm.min_time_optimizer import set_up_double_track_problem
    from spline_trajectory_optimization_amd.min_time_optm import defaults
    from spline_trajectory_optimization_amd.models.trajectory import Trajectory
    N, L = 40, 200.0
    s0 = np.arange(N) * (L / N)
    rt = types.SimpleNamespace(abscissa=s0, center_s=types.SimpleNamespace(get_length=lambda: L),
                               left_intp=lambda s: 4.0 + 0 * np.asarray(s), right_intp=lambda s: -3.5 + 0 * np.asarray(s),
                               curvature_intp=lambda s: 2 * np.pi / L + 0 * np.asarray(s))
    traj = Trajectory(N)
    rng = np.random.default_rng(5)
    traj[:, Trajectory.SPEED] = rng.uniform(5, 20, N); traj[:, Trajectory.TIME] = rng.uniform(0.2, 0.9, N)
    params = {"N": N, "model": defaults.MODEL, "race_track": rt, "traj_d": traj, "average_track_width": 7.0,
              "speed_cap": 30.0, "verbose": False, "max_iter": 500, "tol": 0.1}
    (X, U, T), (scale_x, scale_u, scale_t), opti = set_up_double_track_problem(params)
    assert X.shape == (N, 6) and U.shape == (N, 4) and T.shape == (N,)
    m = defaults.MODEL
    np.testing.assert_array_equal(np.ravel(scale_x), [1.0, 7.0, 1.0, 1.0, 0.5, 30.0])                     # :109
    np.testing.assert_array_equal(np.ravel(scale_u), [m["Fd_max"], abs(m["Fb_max"]), m["delta_max"], m["mass"] * 50.0])
    assert scale_t == 1.0
    # the CLI's read-back (entrypoints/traj_opt_double_track.py:66-69) of the initial point = :146-151 as written
    x = opti.debug.value(X) * scale_x + np.hstack([s0[:, np.newaxis], np.zeros((N, 5))])
    u = opti.debug.value(U) * scale_u
    t = opti.debug.value(T) * scale_t
    np.testing.assert_allclose(x[:, 0], s0, rtol=0, atol=0)
    assert not x[:, 1:5].any()
    np.testing.assert_allclose(x[:, 5], traj[:, Trajectory.SPEED], rtol=1e-15)
    np.testing.assert_allclose(u, np.tile([1.0, -1.0, 0.001, 0.0], (N, 1)), rtol=1e-15)
    np.testing.assert_array_equal(t, traj[:, Trajectory.TIME])
    # the build's own variant stays available as an explicit option
    (_, _, T2), _, opti2 = set_up_double_track_problem(dict(params, initial_guess="clipped"))
    np.testing.assert_array_equal(opti2.debug.value(T2), np.roll(traj[:, Trajectory.TIME], -1))
    # x0 / u0 / t0 (:152-155)
    x0 = x.copy(); x0[:, 1] = 0.3
    (X3, _, _), _, opti3 = set_up_double_track_problem(dict(params, x0=x0, u0=u, t0=t.reshape(-1, 1)))
    np.testing.assert_allclose(opti3.debug.value(X3)[:, 1], 0.3 / 7.0, rtol=1e-15)
    opti.set_initial(T, 0.5 * np.ones(N))
    assert (opti.value(T) == 0.5).all()
    opti.solver("ipopt", {"expand": True}, {"max_iter": 7, "tol": 1e-3, "print_level": 0})                 # :158-161
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception, match="no HIP device|librl_mincurv"):
            opti.solve()
        assert opti.stats()["success"] is False


def test_qss_dataflow_schedule_model():
    """tests/qss_schedule_model.py: the readiness rules of k_qss_dfw replayed on step logs of the sequential oracle -- every step
    released only after all its true dependencies, progress in every pass, spawned fronts numbered in list order (the small
    cases here; 24 trajectories up to N = 2000 when the kernel was written)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "qss_schedule_model.py"), "4", "24", "small"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "no violation" in out.stdout and out.stdout.count("case ") >= 2, out.stdout[-1500:] + out.stderr[-1500:]
    # the rules alone, iteration 0 included (the kernel runs that one in list order: start = 1 above)
    out = subprocess.run([sys.executable, os.path.join(here, "qss_schedule_model.py"), "1", "24", "small", "0"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0 and "no violation" in out.stdout and out.stdout.count("case ") >= 1, out.stdout[-1500:] + out.stderr[-1500:]


def test_spawn_ranks_propagates_a_failed_rank():
    """`python bench.py --gpus N` starts its own ranks (bench.py: spawn_ranks).  A rank that exits non-zero must end the run:
    the other ranks are terminated and the failing rank's exit code is the parent's, well inside the deadline -- not a hang
    until the 50-minute limit, and not a zero exit with a missing JSON line."""
    import subprocess
    import sys
    import time
    env = dict(os.environ, RL_BENCH_FAIL_RANK="1", RL_BENCH_HOLD_S="600")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert p.returncode == 7, (p.returncode, p.stdout[-500:], p.stderr[-500:])
    assert took < 200, took              # rank 0 was holding for 600 s: it was terminated, not waited for
    assert p.stdout.strip() == ""        # no line was printed
