"""Frame-to-frame odometry over a sequence: every consecutive pair (frame k as
target, frame k+1 as source) is one independent registration, so the pairs of a
sequence are sharded over the ranks of a node and, on each rank, run as batches.

This is the refinement half of the reference's per-frame loop
(EnvironmentConstructor.getRefinedTransition, :116-132: full_pt2pl of the new
frame against the reference cloud) without its IMU prediction and voxel feed;
initial guesses are identity or supplied by the caller.
"""
import numpy as np

from . import sharding
from .compute_shader import MODE_PLANE, MODE_POINT, ComputeShader, IcpBatch
from .icp_container import PointcloudIcpContainer


def register_pairs(targets, sources, origin=None, point_to_plane=True, iterations=10, device=0,
                   batch_size=16, guesses=None):
    """Registers sources[i] onto targets[i] for every i, `batch_size` pairs at a time.
    Returns [(t_opt, R_opt)] with the reference's convention: p <- R_opt p - t_opt maps the
    source into the target's frame (EnvironmentConstructor.py:101-102).  guesses[i] = (t, R)
    pre-aligns source i the same way before the registration."""
    origin = np.zeros(3) if origin is None else np.asarray(origin, dtype=np.float64)
    mode = MODE_PLANE if point_to_plane else MODE_POINT
    n = len(targets)
    out = [None] * n
    size = max(max(len(a), len(b)) for a, b in zip(targets, sources)) if n else 1
    computes = [ComputeShader(maxPointsPerCloud=size, maxScans=100, device=device)
                for _ in range(min(batch_size, n))]
    icps = [PointcloudIcpContainer(c) for c in computes]
    batch = IcpBatch(computes) if computes else None
    try:
        for lo in range(0, n, max(1, len(computes))):
            idx = list(range(lo, min(n, lo + len(computes))))
            moved = []
            for slot, k in enumerate(idx):
                src = np.asarray(sources[k], dtype=np.float64)
                if guesses is not None and guesses[k] is not None:
                    t0, R0 = guesses[k]
                    src = (R0 @ src.T).T - t0
                moved.append(src)
                if point_to_plane:
                    icps[slot].preparePointToPlane(targets[k], origin, src)
                else:
                    icps[slot].preparePointToPoint(targets[k], origin, src)
            if len(idx) == len(computes):
                batch.upload_sources(moved)
                batch.prepare(point_to_plane)
                batch.launch(mode, iterations)
                results = batch.results(iterations)
            else:       # last, short batch: pair by pair
                results = [computes[s].icpRun(moved[s], mode, iterations)[:2] + (None, None) for s in range(len(idx))]
            for slot, k in enumerate(idx):
                dx, converged = results[slot][0], results[slot][1]
                t, R = computes[slot].fetchPose()          # tracking grid + Kabsch behind the loop, on the device
                if guesses is not None and guesses[k] is not None:
                    t0, R0 = guesses[k]                      # p <- R (R0 p - t0) - t
                    t, R = R @ t0 + t, R @ R0
                out[k] = (t, R)
    finally:
        if batch is not None:
            batch.close()
        for c in computes:
            c.cleanup()
    return out


def sequence_odometry(frames, origin=None, point_to_plane=True, iterations=10, rank=0, world=1, device=0,
                      batch_size=16, gather_device="cpu"):
    """Poses of every frame of `frames` in the frame of frames[0]: [(R_w, t_w)] with p0 = R_w p + t_w.
    Pair k (frame k <- frame k+1) is registered by rank k mod world; all ranks return all poses."""
    n_pairs = len(frames) - 1
    mine = sharding.pairs_of_rank(n_pairs, rank, world)
    res = register_pairs([frames[k] for k in mine], [frames[k + 1] for k in mine], origin, point_to_plane,
                         iterations, device, batch_size)
    local = {k: r for k, r in zip(mine, res)}
    poses = sharding.gather_poses(local, n_pairs, rank, world, device=gather_device)
    return sharding.compose_odometry(poses)
