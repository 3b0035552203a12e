"""Frame-pair sharding over the GPUs of one node (one process per GPU).

Consecutive frame pairs of a sequence are independent registration problems
once each has its own initial guess, so they shard with no collective in the
data path: pair k goes to rank k mod world.  The only exchange is the gather of
the results (12 float64 per pair) at the end, over torch.distributed (RCCL on
GPUs, gloo in CPU tests).
"""
import numpy as np


def pairs_of_rank(n_pairs, rank, world):
    """Indices of the pairs rank `rank` registers (round robin)."""
    return list(range(rank, n_pairs, world))


def pack_pose(t, R):
    return np.concatenate([np.asarray(t, dtype=np.float64).reshape(3), np.asarray(R, dtype=np.float64).reshape(9)])


def unpack_pose(v):
    v = np.asarray(v, dtype=np.float64)
    return v[:3].copy(), v[3:12].reshape(3, 3).copy()


def gather_poses(local, n_pairs, rank, world, device="cpu"):
    """local: {pair index: (t, R)} computed by this rank.  Returns the list of all
    n_pairs poses in pair order on every rank (one all_gather of a padded block)."""
    if world == 1:
        return [local[k] for k in range(n_pairs)]
    import torch
    import torch.distributed as dist
    per_rank = -(-n_pairs // world)
    block = torch.full((per_rank, 13), -1.0, dtype=torch.float64, device=device)
    if len(local) > per_rank:
        raise ValueError("a rank holds more pairs than ceil(n_pairs / world)")
    for slot, k in enumerate(sorted(local)):            # round robin or contiguous blocks: any split that is even
        block[slot, 0] = float(k)
        block[slot, 1:] = torch.from_numpy(pack_pose(*local[k])).to(device)
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(blocks, block)
    out = [None] * n_pairs
    for b in blocks:
        for row in b.cpu().numpy():
            if row[0] >= 0:
                out[int(row[0])] = unpack_pose(row[1:])
    return out


def compose_odometry(poses):
    """Chains per-pair registrations (t_k, R_k), each mapping frame k+1 into frame k
    by p <- R_k p - t_k (EnvironmentConstructor.py:101-102), into the pose of every
    frame in the frame of frame 0: p0 = Rw p + tw."""
    Rw, tw = np.eye(3), np.zeros(3)
    out = [(Rw.copy(), tw.copy())]
    for t, R in poses:
        tw = tw - Rw @ t
        Rw = Rw @ R
        out.append((Rw.copy(), tw.copy()))
    return out
