"""CPU checker of the exact-neighbour path (test infrastructure only: imported by
tests/, never by the product).

The reference's exact search IS scipy's KD-tree (PointcloudIcpContainer.py:7, :467-472),
so the checker calls the same class; brute force covers the tie rule on small inputs.
icp_step_kd / align_kd restate PointcloudIcpContainer.py:474-515; map_ls_tile combines
the exact neighbour with the tile arithmetic of np_oracle (LeastSquaresPoint/Plane.glsl).
"""
import numpy as np
from scipy.spatial import KDTree

from . import np_oracle as O


def exact_nn(target, query, bound=np.inf):
    """(dist, idx) like KDTree.query(query, distance_upper_bound=bound)."""
    return KDTree(np.asarray(target, dtype=np.float64)).query(np.asarray(query, dtype=np.float64),
                                                              distance_upper_bound=bound)


def exact_nn_lowest(target, query, bound=np.inf, k=8):
    """exact_nn with the product's tie rule: among the (up to k) nearest candidates the smallest
    ((dx^2 + dy^2) + dz^2) wins, the lowest index on equality.  Misses give (inf, n)."""
    t = np.asarray(target, dtype=np.float64)
    q = np.asarray(query, dtype=np.float64)
    k = min(k, len(t))
    dist, idx = KDTree(t).query(q, k=k, distance_upper_bound=bound)
    if k == 1:
        dist, idx = dist[:, None], idx[:, None]
    ok = idx < len(t)
    d = q[:, None, :] - t[np.where(ok, idx, 0)]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    d2 = np.where(ok, d2, np.inf)
    best = d2.min(axis=1)
    cand = np.where(d2 == best[:, None], idx, np.iinfo(np.int64).max)
    out_idx = cand.min(axis=1)
    miss = ~ok.any(axis=1)
    out_idx[miss] = len(t)
    return np.where(miss, np.inf, np.sqrt(best)), out_idx


def brute_nn(target, query):
    """Exact float64 minimum of ((dx^2 + dy^2) + dz^2), lowest index on ties."""
    t = np.asarray(target, dtype=np.float64)
    q = np.asarray(query, dtype=np.float64)
    idx = np.zeros(len(q), dtype=np.int64)
    d2 = np.zeros(len(q))
    for i in range(len(q)):
        d = q[i] - t
        s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        idx[i] = int(np.argmin(s))           # first minimum
        d2[i] = s[idx[i]]
    return np.sqrt(d2), idx


def icp_step_kd(tree, pc1, pc2, keep=0.8):
    """PointcloudIcpContainer.icp_step_KD (:474-498) with a stable sort at the cut."""
    dists, closest = tree.query(pc2)
    taken = int(len(dists) * keep)
    order = np.argsort(dists, kind="stable")[:taken]
    reference, other = pc1[closest][order], pc2[order]
    mean_ref = np.mean(reference, axis=0)
    mean_other = np.mean(other, axis=0)
    W = (reference - mean_ref).T @ (other - mean_other)
    U, S, Vh = np.linalg.svd(W)
    R = U @ Vh
    return R, mean_ref - R @ mean_other


def align_kd(reference_points, points_to_align, iterations=20):
    """PointcloudIcpContainer.align_KD (:500-515)."""
    tree = KDTree(reference_points)
    grid0 = O.uniform_grid(10)
    grid = grid0.copy()
    pts = np.asarray(points_to_align, dtype=np.float64).copy()
    for _ in range(iterations):
        R, t = icp_step_kd(tree, reference_points, pts)
        pts = (R @ pts.T).T + t
        grid = (R @ grid.T).T + t
    R_opt, t_opt = O.uniform_optimal_icp(grid, grid0)
    return pts, R_opt, t_opt


def map_ls_tile(target, normals, query, mode, radius, owned=None):
    """8x8 tile of one scan-to-map dispatch: exact neighbour strictly within `radius`, tile values
    of LeastSquaresPoint.glsl (mode 0) / LeastSquaresPlane.glsl (mode 1) on the float32 casts, summed
    by the 64-ary tree over the queries in order; queries without neighbour (or not owned) give zeros."""
    target = np.asarray(target, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    dist, idx = exact_nn_lowest(target, query, radius)
    hit = np.isfinite(dist)
    if owned is not None:
        hit &= owned
    safe = np.where(hit, idx, 0)
    a4 = O.pad_points(target.astype(np.float32))
    pb = O.pad_points(query.astype(np.float32))
    a4[:, 3] = 0
    pb[:, 3] = 0
    diff = pb - a4[safe]
    d = (diff[:, 0] * diff[:, 0]).astype(np.float32)
    d = (d + diff[:, 1] * diff[:, 1]).astype(np.float32)
    d = (d + diff[:, 2] * diff[:, 2]).astype(np.float32)
    d = (d + diff[:, 3] * diff[:, 3]).astype(np.float32)
    if mode == 0:
        tiles = O.ls_point_tiles(a4, pb, safe, d)
    else:
        nrm = np.asarray(normals, dtype=np.float32)
        n4 = np.zeros((len(nrm), 4), dtype=np.float32)
        n4[:, :nrm.shape[1]] = nrm
        tiles = O.ls_plane_tiles(a4, pb, n4, safe, d)
    tiles[~hit] = 0
    return O.hb_tree_sum(tiles), hit, idx


def pca_normals(points, radius, min_points=3, viewpoint=None, only=None):
    """Normals of a ring-less cloud as the product defines them (no counterpart in the reference, whose
    NormalShader.glsl needs scan rings): smallest eigenvector (numpy.linalg.eigh) of the scatter matrix of
    the points with ((dx^2 + dy^2) + dz^2) < radius^2 around each point, itself included; turned towards
    `viewpoint` like normalTowardsOrigin (NormalShader.glsl:62-71), else largest component positive; zero
    for fewer than max(min_points, 3) points.  Returns (normals float64 (n,3), counts, gap) where gap =
    (second smallest - smallest eigenvalue) / largest says how well defined the direction is.
    `only` = indices of the points to evaluate (rows of the others stay zero)."""
    p = np.asarray(points, dtype=np.float64)
    tree = KDTree(p)
    rows = np.arange(len(p)) if only is None else np.asarray(only, dtype=np.int64)
    near = tree.query_ball_point(p[rows], radius * (1.0 + 1e-9))
    normals = np.zeros((len(p), 3))
    counts = np.zeros(len(p), dtype=np.int64)
    gap = np.zeros(len(p))
    r2 = radius * radius
    for i, cand in zip(rows, near):
        d = p[np.asarray(cand, dtype=np.int64)] - p[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        d = d[d2 < r2]
        counts[i] = len(d)
        if len(d) < max(min_points, 3):
            continue
        s1 = d.sum(axis=0)
        C = d.T @ d - np.outer(s1, s1) / len(d)
        w, V = np.linalg.eigh(C)
        n = V[:, 0]
        if viewpoint is not None:
            flip = np.dot(np.asarray(viewpoint, dtype=np.float64) - p[i], n) < 0.0
        else:
            flip = n[int(np.argmax(np.abs(n)))] < 0.0
        normals[i] = -n if flip else n
        gap[i] = (w[1] - w[0]) / w[2] if w[2] > 0 else 0.0
    return normals, counts, gap
