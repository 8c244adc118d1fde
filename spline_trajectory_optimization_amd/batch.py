"""Batched min-curvature solves: workload generators for the BASELINE.json configs and the
rank-sharding helpers (one process per GPU, independent track instances, one gather).

No arithmetic of the hot path lives here: instances are solved by rl_mincurv_solve_batch_* (HIP).
"""
import os

import numpy as np

from . import _lib, ops
from .models.trajectory import BSplineTrajectory

_PKG = os.path.dirname(os.path.abspath(__file__))
MONZA_DIR = os.path.join(_PKG, "examples", "race_track", "monza")


def load_track_csv(path):
    """x,y columns of a track CSV (tests/test_trajectory.py:9-15 reads usecols=(0,1))."""
    return np.loadtxt(path, dtype=np.float64, delimiter=",", skiprows=1, usecols=(0, 1))


def load_track_rows(path):
    """All numeric columns of a track CSV: 3 columns x,y,z or 4 columns x,y,z,bank (README.md:28-35;
    the reference itself only ever reads columns 0-1, race_track.py:23-25)."""
    rows = np.loadtxt(path, dtype=np.float64, delimiter=",", skiprows=1, ndmin=2)
    assert rows.shape[1] in (2, 3, 4), rows.shape
    return rows


def save_track_rows(path, rows):
    rows = np.asarray(rows, dtype=np.float64)
    header = ",".join(["x", "y", "z", "bank"][:rows.shape[1]])
    np.savetxt(path, rows, delimiter=",", header=header, comments="", fmt="%.17g")


def bank_on_samples(raw_xy, raw_bank, points):
    """BASELINE config 4: bank angle [rad] of the raw 4-column track rows carried onto the samples of
    a trajectory table -- for every sample the two nearest consecutive raw points, linear in the
    projection onto their segment.  Returns float64 [N] (what goes into Trajectory.BANK, column 13;
    only the simulator reads it, simulator.py:51-58)."""
    raw_xy = np.asarray(raw_xy, dtype=np.float64); raw_bank = np.asarray(raw_bank, dtype=np.float64)
    p = np.asarray(points)[:, :2]
    d2 = ((p[:, None, :] - raw_xy[None, :, :]) ** 2).sum(axis=2)
    j = d2.argmin(axis=1)
    P = len(raw_xy)
    jn, jp = (j + 1) % P, (j - 1) % P
    use_next = d2[np.arange(len(p)), jn] <= d2[np.arange(len(p)), jp]
    j0 = np.where(use_next, j, jp); j1 = np.where(use_next, jn, j)
    seg = raw_xy[j1] - raw_xy[j0]
    tpar = ((p - raw_xy[j0]) * seg).sum(axis=1) / np.maximum((seg * seg).sum(axis=1), 1e-300)
    tpar = np.clip(tpar, 0.0, 1.0)
    return raw_bank[j0] * (1.0 - tpar) + raw_bank[j1] * tpar


def load_monza():
    return (load_track_csv(os.path.join(MONZA_DIR, "MONZA_UNOPTIMIZED_LINE_enu.csv")),
            load_track_csv(os.path.join(MONZA_DIR, "MONZA_LEFT_BOUNDARY_enu.csv")),
            load_track_csv(os.path.join(MONZA_DIR, "MONZA_RIGHT_BOUNDARY_enu.csv")))


def monza_centerline(s=100.0, k=5):
    """The centre-line spline of tests/test_optimizer.py:16-17 (host FITPACK fit)."""
    centre, _, _ = load_monza()
    return BSplineTrajectory(centre, s, k)


def half_widths_from_bounds(points):
    """|p - L|, |p - R| per sample from a bounds-filled [N,19] table."""
    wl = np.hypot(points[:, 9] - points[:, 0], points[:, 10] - points[:, 1])
    wr = np.hypot(points[:, 11] - points[:, 0], points[:, 12] - points[:, 1])
    return wl, wr


def width_batch(w_left, w_right, B, seed=1234, eps=0.15, floor=1.5):
    """per-instance widths w[b,i] = w[i] * (1 + e[b]), e ~ U(-eps, eps)
    from numpy.random.default_rng(seed), floored at `floor` metres.  Returns float64 [B,N,2]."""
    rng = np.random.default_rng(seed)
    e = rng.uniform(-eps, eps, size=(B, 2))
    w = np.empty((B, len(w_left), 2))
    w[:, :, 0] = np.maximum(w_left[None, :] * (1.0 + e[:, 0:1]), floor)
    w[:, :, 1] = np.maximum(w_right[None, :] * (1.0 + e[:, 1:2]), floor)
    return w


def default_i_start(n, k, max_iter, seed=0):
    """A pinned stand-in for the unseeded np.random.randint of optimizer.py:303."""
    rng = np.random.RandomState(seed)
    return np.array([rng.randint(k // 2, n - (k - k // 2)) for _ in range(max_iter)], dtype=np.int32)


def make_track(spline: BSplineTrajectory, N: int, device=None):
    t, cx, cy, k = spline._tck()
    return _lib.Track(_lib.Context.get(device), t, cx, cy, k, N)


# ---------------------------------------------------------------- min-time NLP plumbing (config 5)
def min_time_initial_guess(points):
    """The initial guess the reference hands to its min-time NLP when no previous solution is given
    (min_time_optimizer.py:146-151): the QSS-simulated table's abscissa, speed and segment times, zero
    lateral offset / relative yaw / yaw rate / slip, controls (1, -1, 0.001, 0) in physical units.
    `points` = [N,19] after Simulator.run_simulation (rl_qss_sim).  Returns (s [N], X [N,6], U [N,4], T [N])
    ready for ops.dt_eval_nodes / ops.dt_eval_jac."""
    pts = np.asarray(points, dtype=np.float64)
    N = len(pts)
    s = pts[:, 6].copy()                      # DIST_TO_SF_BWD: race_track.abscissa (race_track.py:56)
    X = np.zeros((N, 6)); X[:, 0] = s; X[:, 5] = pts[:, 4]
    U = np.tile(np.array([1.0, -1.0, 0.001, 0.0]), (N, 1))
    T = pts[:, 16].copy()
    return s, X, U, T


def signed_curvature(points):
    """Centre-line curvature with sign at the samples of a sampled table (the turn radius of column 5 is
    unsigned): heading change per arc length, central differences over the closed line."""
    pts = np.asarray(points, dtype=np.float64)
    yaw = np.unwrap(pts[:, 3])
    dyaw = np.roll(yaw, -1) - np.roll(yaw, 1)
    dyaw[0] += 2 * np.pi * round((yaw[-1] - yaw[0]) / (2 * np.pi)); dyaw[-1] += 2 * np.pi * round((yaw[-1] - yaw[0]) / (2 * np.pi))
    L = pts[0, 7] if pts[0, 7] > 0 else pts[-1, 6]
    ds = np.roll(pts[:, 6], -1) - np.roll(pts[:, 6], 1)
    ds[0] += L; ds[-1] += L
    return dyaw / ds


# ---------------------------------------------------------------- rank sharding
def shard_range(B, rank, world):
    """Contiguous block partition: rank r takes instances [lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _Works:
    """wait() on every work handle of a grouped send / receive (what dist.gather's single handle offers)."""

    def __init__(self, works):
        self.works = list(works or [])

    def wait(self):
        for w in self.works:
            w.wait()
        return True


_gather_fallback = {"p2p": os.environ.get("RL_GATHER", "") == "p2p"}   # RL_GATHER=p2p forces the grouped send / receive


def _gather_p2p(pad, bufs, rank, world, dist, async_op):
    """The same collective as grouped point-to-point transfers ("ncclGather-equivalent via grouped
    ncclSend / ncclRecv"): rank 0 posts one receive per peer, every other rank one send, in ONE batch
    (dist.batch_isend_irecv = ncclGroupStart ... ncclGroupEnd on RCCL); rank 0's own shard is a local copy."""
    if rank == 0:
        bufs[0].copy_(pad)
        ops_ = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(1, world)]
    else:
        ops_ = [dist.P2POp(dist.isend, pad, 0)]
    works = _Works(dist.batch_isend_irecv(ops_) if ops_ else [])
    if async_op:
        return works
    works.wait()
    return None


def gather_to_root(tensor, rank, world, dist, total=None, out=None, async_op=False, method="auto"):
    """The single collective of the path: every rank's result shard to rank 0 (RCCL `gather` over
    xGMI; gloo in the CPU tests).  `total` = global instance count when the shards come from
    shard_range(total, r, world) (they may then differ by one instance and are padded to the
    largest); None = every rank holds the same number of instances (weak scaling).
    `out` (rank 0, equal shards only): preallocated [world * B, ...] tensor the shards land in directly,
    no concatenation.  `async_op=True` returns the collective's work handle instead of the result: the
    gather then runs on the process group's own stream, ordered after what is already enqueued on the
    current stream, and overlaps with whatever the caller enqueues next (bench.py: the next launch).
    `method`: "gather" = dist.gather; "p2p" = the same transfer as one batch of grouped isend / irecv; "auto" =
    dist.gather, and from the first time the backend REFUSES it (NotImplementedError / "not supported": every rank is
    refused alike, before anything is sent) the grouped form for the rest of the process; any other exception is re-raised.
    RL_GATHER=p2p decides for the grouped form up front."""
    import torch
    if world == 1:
        return None if async_op else tensor
    if total is None:
        counts = [tensor.shape[0]] * world
    else:
        counts = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    mx = max(counts)
    pad = tensor
    if tensor.shape[0] < mx:
        pad = torch.cat([tensor, tensor.new_zeros((mx - tensor.shape[0],) + tuple(tensor.shape[1:]))])
    pad = pad.contiguous()
    bufs = None
    if rank == 0:
        if out is not None:
            assert total is None and out.shape[0] == world * mx and out.is_contiguous()
            bufs = list(out.view((world, mx) + tuple(tensor.shape[1:])).unbind(0))
        else:
            bufs = [torch.empty_like(pad) for _ in range(world)]
    use_p2p = method == "p2p" or (method == "auto" and _gather_fallback["p2p"])
    work = None
    if not use_p2p:
        try:
            work = dist.gather(pad, bufs, dst=0, async_op=async_op)
        except (RuntimeError, NotImplementedError) as e:
            # only a REFUSED collective switches forms: a backend that does not implement gather says so on every rank alike,
            # before anything is sent.  Any other failure (a shape error on one rank, out of memory, a communicator error)
            # is that rank's alone -- falling back there would leave it in send / receive while the others sit in gather.
            refused = isinstance(e, NotImplementedError) or any(
                w in str(e).lower() for w in ("not support", "unsupported", "not implemented", "no backend type"))
            if method != "auto" or not refused:
                raise
            print(f"gather_to_root: dist.gather raised on backend '{dist.get_backend()}' ({type(e).__name__}: {e}); "
                  f"using grouped isend / irecv from here on", flush=True)
            _gather_fallback["p2p"] = True
            use_p2p = True
    if use_p2p:
        work = _gather_p2p(pad, bufs, rank, world, dist, async_op)
    if async_op:
        return work
    if rank != 0:
        return None
    if out is not None:
        return out
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])


# ---------------------------------------------------------------- synthetic oval (BASELINE config 3)
def oval_points(spacing=10.0, heading0=np.deg2rad(17.0)):
    """Centre line and half-widths of the synthetic Indy-style oval of SURVEY.md 8(d) config 3 (the
    reference ships no Indy data): two 1006 m straights, two 201 m short chutes, four 402 m
    quarter-turns (4023 m in all); half-width 7.6 m on the straights, 9.1 m in the turns.
    The front straight heads `heading0` off the x axis: the reference's constraints are axis-aligned
    boxes between the two bound points (optimizer.py:243-248), which collapse to zero width on an
    exactly axis-parallel straight and leave every QP there feasible or not by rounding noise.
    Returns (xy [P,2], half_width [P]) sampled every ~`spacing` metres, counter-clockwise."""
    R = 402.0 / (np.pi / 2.0)
    segs = [("s", 1006.0), ("t", 402.0), ("s", 201.0), ("t", 402.0),
            ("s", 1006.0), ("t", 402.0), ("s", 201.0), ("t", 402.0)]
    pos = np.array([0.0, 0.0]); heading = float(heading0)
    pts, hw = [], []
    for kind, length in segs:
        m = max(2, int(round(length / spacing)))
        for j in range(m):
            s = length * j / m
            if kind == "s":
                p = pos + s * np.array([np.cos(heading), np.sin(heading)])
                w = 7.6
            else:
                ang = s / R
                c = pos + R * np.array([-np.sin(heading), np.cos(heading)])      # centre of the left turn
                p = c + R * np.array([np.sin(heading + ang), -np.cos(heading + ang)])
                w = 9.1
            pts.append(p); hw.append(w)
        if kind == "s":
            pos = pos + length * np.array([np.cos(heading), np.sin(heading)])
        else:
            c = pos + R * np.array([-np.sin(heading), np.cos(heading)])
            heading += np.pi / 2.0
            pos = c + R * np.array([np.sin(heading), -np.cos(heading)])
    return np.array(pts), np.array(hw)


def oval_track_rows(bank_turn_deg=9.2, spacing=10.0):
    """The oval as 4-column track rows x,y,z,bank (BASELINE config 4): turns banked `bank_turn_deg`
    (Indianapolis: 9.2 degrees), straights flat, z = 0."""
    xy, hw = oval_points(spacing)
    bank = np.where(hw > 8.0, np.deg2rad(bank_turn_deg), 0.0)
    return np.column_stack([xy, np.zeros(len(xy)), bank])


def oval_centerline(s=100.0, k=5):
    xy, _ = oval_points()
    return BSplineTrajectory(xy, s, k)


def oval_half_widths(N):
    """Half-widths of the oval resampled on N uniform-in-parameter samples (nearest raw point)."""
    xy, hw = oval_points()
    idx = np.minimum((np.arange(N) * (len(hw) / N)).astype(int), len(hw) - 1)
    return hw[idx].copy(), hw[idx].copy()


def solve_grouped(groups, i_starts, search=_lib.SEARCH_WINDOWED):
    """Mixed batches (BASELINE config 3): instances are grouped by track so that every workgroup of
    one launch sees one knot layout.  groups = [(Track, widths [B_g,N,2]), ...], i_starts = one pinned
    start-index array per group (control-point counts differ between tracks).
    Returns a list of (ctrl, xy, n_success, status) per group, in order."""
    out = []
    for (trk, widths), ist in zip(groups, i_starts):
        ctrl, xy, ns, status, _ = ops.solve_batch_host(trk, _lib.BOUNDS_WIDTHS, widths, ist, search=search)
        out.append((ctrl, xy, ns, status))
    return out
