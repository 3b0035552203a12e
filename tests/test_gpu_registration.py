"""Parity of the HIP registration kernels (through the C ABI, via the
ComputeShader-compatible class) with the NumPy restatement in oracle/.

Bar: bit-exact for indices, distances, ring hints, normals, per-point tiles and
the tree-summed H/b (all float32 with a fixed evaluation order); the
device-resident ICP loop is held to 1e-5 rotation / 1e-4 m translation and, in
practice, is bit-equal step by step.
"""
import numpy as np
import pytest

from oracle import np_oracle as O
from pointcloudprocessing_amd import MODE_PLANE, MODE_POINT, PointcloudIcpContainer
from tests._data import bits, golden, kitti_pair, small_pair

pytestmark = pytest.mark.gpu
ORIGIN = np.zeros(3)


def _oracle_state(a, origin, kind):
    st = O.RingIcp(100)
    lines = O.get_scan_lines(a, origin)
    {"nns": st.prepare_nns, "ls": st.prepare_ls, "plane": st.prepare_plane}[kind](a, lines, origin)
    return st, lines


# ------------------------------------------------------------------------------ NN (a-2)
@pytest.mark.parametrize("origin", [np.zeros(3), np.array([0.3, -0.1, 0.25])])
def test_nn_small_matches_oracle_bitwise(compute, origin):
    a, b = small_pair()
    st, lines = _oracle_state(a, origin, "nns")
    compute.prepareNNS(a, lines, b, origin)
    for call in range(3):          # first call scans all 100 rings, later ones follow the hints
        idx, dist = compute.dispatchNNS(b)
        ridx, rdist = st.dispatch_nns(b)
        assert np.array_equal(idx, ridx), f"call {call}"
        assert np.array_equal(bits(dist), bits(rdist)), f"call {call}"
        assert np.array_equal(bits(compute.corr_out[:, 3]), bits(st.hint)), f"call {call}"
        b = b + np.array([0.01, 0.0, -0.005])


def test_nn_kitti_matches_golden(compute):
    a, b = kitti_pair()
    g = golden("oracle_kitti.npz")
    icp = PointcloudIcpContainer(compute)
    icp.prepareNNS(a, ORIGIN, b)
    idx, dist = icp.dispatchNNS(b)
    assert np.array_equal(idx, g["nn_idx"])
    assert np.array_equal(bits(dist), bits(g["nn_dist"]))
    assert np.array_equal(compute.corr_out[:, 3].astype(np.int8), g["nn_hint"])
    idx2, dist2 = icp.dispatchNNS(b)   # hint-guided second pass
    assert O_sha(idx2) == str(g["nn2_idx_sha"])
    assert O_sha(dist2) == str(g["nn2_dist_sha"])


def test_nn_kitti_nonzero_origin_matches_golden(compute):
    a, b = kitti_pair()
    g = golden("oracle_kitti.npz")
    o = golden("ref_kat.npz")["scan_origin"]
    icp = PointcloudIcpContainer(compute)
    icp.prepareNNS(a, o, b[::8])
    idx, dist = icp.dispatchNNS(b[::8])
    assert np.array_equal(idx, g["nn_o_idx"])
    assert np.array_equal(bits(dist), bits(g["nn_o_dist"]))


def O_sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------ normals (a-3)
def test_normals_small_matches_oracle_bitwise(compute):
    a, _ = small_pair()
    lines = O.get_scan_lines(a, ORIGIN)
    compute.prepareDispatchNormals(a, lines, ORIGIN)
    ref = O.normals(O.pad_points(a), O.origin4(ORIGIN), O.pad_scan_lines(lines).astype(np.int64), 100)
    assert compute.normals_out_a.shape == (len(a), 4)
    assert np.array_equal(bits(compute.normals_out_a), bits(ref))


def test_normals_kitti_matches_golden(compute):
    a, _ = kitti_pair()
    g = golden("oracle_kitti.npz")
    lines = O.get_scan_lines(a, ORIGIN)
    compute.prepareDispatchNormals(a, lines, ORIGIN)
    assert np.array_equal(bits(compute.normals_out_a[::16]), bits(g["normals_sub"]))
    assert O_sha(compute.normals_out_a) == str(g["normals_sha"])


def test_normals_equal_the_reference_getNormals(compute):
    """tests/GpuTest.py:297-323 as the reference runs it: prepareDispatchNormals on frame 0, the first
    50,000 rows of normals_out_a against GpuTest.getNormals (fixture ref_normals.npz), > 99.9 % within 1e-3."""
    ref = golden("ref_normals.npz")
    ref32 = np.insert(ref["normals64"].astype(np.float32), 3, 0, axis=1)      # as tests/GpuTest.py:308-309 casts it
    a, _ = kitti_pair()
    icp = PointcloudIcpContainer(compute)
    lines = icp.getScanLines(a, ORIGIN)
    compute.prepareDispatchNormals(a, lines, ORIGIN)
    normals = compute.normals_out_a[:50000]
    diff = normals - ref32
    assert (np.linalg.norm(diff, axis=1) < 1e-3).mean() > 0.999
    same = np.linalg.norm(normals[:, :3].astype(np.float64) - ref["normals64"], axis=1) < 1e-5
    assert same.mean() > 0.9997        # same neighbours picked as the reference's search, see the CPU test


# ------------------------------------------------------------------------------ LS tiles + tree sum (a-4, a-5, a-6)
def test_ls_point_tiles_and_sum_match_oracle_bitwise(compute):
    a, b = small_pair()
    st, lines = _oracle_state(a, ORIGIN, "ls")
    compute.prepareLS(a, lines, b, ORIGIN)
    tiles = compute.lsTiles(b, MODE_POINT).copy()
    nn, d = st._nn(O.pad_points(b))
    ref = O.ls_point_tiles(st.a4, O.pad_points(b), nn, d)
    assert np.array_equal(bits(tiles), bits(ref))
    # second dispatch (hints now set on both sides) through the fused sum
    H, bb = compute.dispatchLS(b)
    rH, rb = st.dispatch_ls(b)
    assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb))


def test_ls_plane_tiles_and_sum_match_oracle_bitwise(compute):
    a, b = small_pair()
    st, lines = _oracle_state(a, ORIGIN, "plane")
    compute.preparePointPlane(a, lines, b, ORIGIN)
    tiles = compute.lsTiles(b, MODE_PLANE).copy()
    nn, d = st._nn(O.pad_points(b))
    ref = O.ls_plane_tiles(st.a4, O.pad_points(b), st.normals_a, nn, d)
    assert np.array_equal(bits(tiles), bits(ref))
    H, bb = compute.dispatchPointPlane(b)
    rH, rb = st.dispatch_plane(b)
    assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb))


def test_identity_correspondence_grid_matches_reference_numpy(compute):
    """tests/GpuTest.py:561-582 (test_uniformGridComputeEqualNumpy), atol 1e-3."""
    k = golden("ref_kat.npz")
    compute.prepareLS(k["grid1"], [(0, 0)], k["grid2"], np.array([0, 0, 0]), True)
    H, b = compute.dispatchLS(k["grid2"])
    assert np.allclose(H, k["grid_H"], atol=1e-3)
    assert np.allclose(b, k["grid_b"].T, atol=1e-3)
    st = O.RingIcp(100)
    st.prepare_ls(k["grid1"], [(0, 0)], np.zeros(3), debug=True)
    rH, rb = st.dispatch_ls(k["grid2"])
    assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(b), bits(rb))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 4095, 4096, 4097, 123645, 200000])
def test_hb_sum_matches_tree_bitwise(compute, n):
    rng = np.random.default_rng(n)
    tiles = (rng.standard_normal((n, 8, 8)) * rng.choice([1e-3, 1.0, 1e3], (n, 1, 1))).astype(np.float32)
    out = compute.hbSum(tiles)
    assert np.array_equal(bits(out), bits(O.hb_tree_sum(tiles)))


# ------------------------------------------------------------------------------ device-resident loop (a-7, a-8)
def _pose_close(t, R, rt, rR):
    assert np.abs(R - rR).max() < 1e-5
    assert np.abs(t - rt).max() < 1e-4


def test_icp_point_small_matches_oracle(compute):
    a, b = small_pair()
    lines = O.get_scan_lines(a, ORIGIN)
    compute.prepareLS(a, lines, b, ORIGIN)
    dx, conv, hb, moved = compute.icpRun(b, MODE_POINT, 8, want_points=True, want_hb=True)
    tr = []
    rt, rR = O.full_pt2pt(a, b, ORIGIN, iterations=8, trace=tr)
    assert len(dx) == 8 and not conv
    for i, (H, bb, d) in enumerate(tr):
        assert np.array_equal(bits(hb[i][:6, :6]), bits(H)), f"H at iteration {i}"
        assert np.array_equal(bits(hb[i][:6, 6]), bits(bb)), f"b at iteration {i}"
        assert np.array_equal(bits(dx[i]), bits(d)), f"dx at iteration {i}"
    t, R = PointcloudIcpContainer(compute).full_pt2pt(a, b, ORIGIN, iterations=8)
    _pose_close(t, R, rt, rR)


def test_icp_plane_small_matches_oracle(compute):
    a, b = small_pair()
    icp = PointcloudIcpContainer(compute)
    for full_iter in (False, True):
        tr = []
        rt, rR = O.full_pt2pl(a, b, ORIGIN, iterations=12, full_iter=full_iter, trace=tr)
        icp.preparePointToPlane(a, ORIGIN, b)
        dx, conv, hb, _ = compute.icpRun(b, MODE_PLANE, 12, full_iter=full_iter, want_hb=True)
        assert len(dx) == len(tr)
        for i, (H, bb, d) in enumerate(tr):
            assert np.array_equal(bits(hb[i][:6, :6]), bits(H)), f"H at iteration {i}"
            assert np.array_equal(bits(dx[i]), bits(d)), f"dx at iteration {i}"
        t, R = icp.full_pt2pl(a, b, ORIGIN, iterations=12, full_iter=full_iter)
        _pose_close(t, R, rt, rR)


def test_icp_kitti_matches_golden(compute):
    a, b = kitti_pair()
    g = golden("oracle_kitti.npz")
    icp = PointcloudIcpContainer(compute)
    t, R = icp.full_pt2pt(a, b, ORIGIN, iterations=20)
    _pose_close(t, R, g["p2p_t"], g["p2p_R"])
    icp.preparePointToPoint(a, ORIGIN, b)
    dx, _, hb, _ = compute.icpRun(b, MODE_POINT, 20, want_hb=True)
    assert np.array_equal(bits(hb[:, :6, :7]), bits(g["p2p_hb"]))
    assert np.array_equal(bits(dx), bits(g["p2p_dx"]))
    t, R = icp.full_pt2pl(a, b, ORIGIN, iterations=10)
    _pose_close(t, R, g["p2l_t"], g["p2l_R"])
    icp.preparePointToPlane(a, ORIGIN, b)
    dx, _, hb, _ = compute.icpRun(b, MODE_PLANE, 10, want_hb=True)
    assert np.array_equal(bits(hb[:, :6, :7]), bits(g["p2l_hb"]))
    assert np.array_equal(bits(dx), bits(g["p2l_dx"]))


def test_stepwise_protocol_equals_device_loop(compute):
    """dispatchPointToPlane per iteration on the host (the reference's loop) and
    the device-resident loop land on the same pose."""
    a, b = small_pair()
    icp = PointcloudIcpContainer(compute)
    icp.preparePointToPlane(a, ORIGIN, b)
    grid = icp.getUniformGrid(10)
    cur = b.copy()
    for _ in range(6):
        t, R, delta = icp.dispatchPointToPlane(cur)
        cur = icp.apply_iteration_of_ls(cur, t, R)
        grid = icp.applyIcpStep(grid, R, t)
    R1, t1 = icp.uniform_optimal_icp(grid, icp.getUniformGrid(10))
    t2, R2 = icp.full_pt2pl(a, b, ORIGIN, iterations=6, full_iter=True)
    _pose_close(t1, R1, t2, R2)


def test_kept_target_gives_the_poses_of_a_fresh_prepare(compute):
    """keep_prepared_target: a second registration against the same target array skips the upload
    and must still equal a registration that prepared from scratch; anything else that uploads a
    target, another array or another origin ends the reuse."""
    a, b = small_pair()
    _, b2 = small_pair(seed=3)
    fresh = PointcloudIcpContainer(compute)
    want = [fresh.full_pt2pl(a, s, ORIGIN, iterations=8) for s in (b, b2, b)]
    want_pt = fresh.full_pt2pt(a, b2, ORIGIN, iterations=5)
    icp = PointcloudIcpContainer(compute)
    icp.keep_prepared_target = True
    serial0 = compute.target_serial
    got = [icp.full_pt2pl(a, s, ORIGIN, iterations=8) for s in (b, b2, b)]
    assert compute.target_serial == serial0 + 1                      # one upload for the three runs
    for (t, R), (wt, wR) in zip(got, want):
        assert np.array_equal(t, wt) and np.array_equal(R, wR)
    t, R = icp.full_pt2pt(a, b2, ORIGIN, iterations=5)               # another mode prepares again
    assert compute.target_serial == serial0 + 2
    assert np.array_equal(t, want_pt[0]) and np.array_equal(R, want_pt[1])
    icp.full_pt2pt(a.copy(), b2, ORIGIN, iterations=1)               # an equal but different array too
    assert compute.target_serial == serial0 + 3
    a2 = a.copy()
    icp.full_pt2pl(a2, b, ORIGIN, iterations=2)
    fresh.preparePointToPlane(a, ORIGIN, b)                          # somebody else used the compute object
    n = compute.target_serial
    t, R = icp.full_pt2pl(a2, b, ORIGIN, iterations=8)
    assert compute.target_serial == n + 1
    assert np.array_equal(t, want[0][0]) and np.array_equal(R, want[0][1])
    icp.full_pt2pl(a2, b, np.array([0.0, 0.0, 0.1]), iterations=2)   # another origin
    assert compute.target_serial == n + 2


def test_batched_pairs_equal_single_pair_runs():
    """pcp_batch_icp_launch on three pairs of different sizes (one of them converging early)
    reproduces pcp_icp_run pair by pair, bit for bit."""
    from pointcloudprocessing_amd import ComputeShader, IcpBatch, lidar
    a, b = small_pair()
    a2 = lidar.synthetic_scan(5, rings=21, steps=400)
    pairs = [(a, b), (a2, (lidar.rotation_xyz(0.003, 0.004, -0.002) @ a2.T).T + 0.02), (a, a + 1e-9)]
    for mode, iters in ((MODE_PLANE, 9), (MODE_POINT, 5)):
        singles, computes = [], []
        for ta, tb in pairs:
            cs = ComputeShader(maxPointsPerCloud=max(len(ta), len(tb)), maxScans=100)
            icp = PointcloudIcpContainer(cs)
            (icp.preparePointToPlane if mode == MODE_PLANE else icp.preparePointToPoint)(ta, ORIGIN, tb)
            singles.append(cs.icpRun(tb, mode, iters, want_hb=True, want_points=True))
            computes.append(cs)
        batch = IcpBatch(computes)
        batch.upload_sources([tb for _, tb in pairs])
        batch.prepare(mode == MODE_PLANE)
        batch.launch(mode, iters)
        got = batch.results(iters, want_hb=True, want_points=True)
        for (dx, conv, hb, pts), (rdx, rconv, rhb, rpts) in zip(got, singles):
            assert conv == rconv and len(dx) == len(rdx)
            assert np.array_equal(bits(dx), bits(rdx)) and np.array_equal(bits(hb), bits(rhb))
            assert np.array_equal(pts, rpts)
        if mode == MODE_PLANE:
            assert got[2][1] and len(got[2][0]) < iters          # the identical pair stops early
        batch.close()
        for cs in computes:
            cs.cleanup()


def test_a_context_may_be_destroyed_before_its_batch():
    """pcp_destroy of an adopted context takes it out of the batch; the batch keeps working with the rest."""
    from pointcloudprocessing_amd import ComputeShader, IcpBatch
    a, b = small_pair()
    computes = []
    for _ in range(3):
        cs = ComputeShader(maxPointsPerCloud=max(len(a), len(b)), maxScans=100)
        PointcloudIcpContainer(cs).preparePointToPlane(a, ORIGIN, b)
        computes.append(cs)
    want = computes[0].icpRun(b, MODE_PLANE, 4)
    batch = IcpBatch(computes)
    lib = computes[1]._lib
    assert lib.pcp_set_stream(computes[1]._ctx, None) != 0 and b"batch" in lib.pcp_last_error()
    batch.upload_sources([b, b, b])
    computes[1].cleanup()                   # in the middle of the batch's life
    batch.computes = [computes[0], computes[2]]
    batch.prepare(True)
    batch.launch(MODE_PLANE, 4)
    for dx, conv, _, _ in batch.results(4):
        assert np.array_equal(bits(dx), bits(want[0])) and conv == want[1]
    batch.close()
    computes[0].synchronize()               # on its own stream again
    for cs in (computes[0], computes[2]):
        cs.cleanup()


# ------------------------------------------------------------------------------ errors
def test_error_paths(compute):
    from pointcloudprocessing_amd import PcpError
    a, b = small_pair()
    with pytest.raises(PcpError):
        compute.prepareNNS(a, [(0, len(a) + 5)], b, ORIGIN)        # ring past the cloud
    with pytest.raises(PcpError):
        compute.prepareLS(a, [(i, i) for i in range(101)], b, ORIGIN)   # > maxScans
    lines = O.get_scan_lines(a, ORIGIN)
    compute.prepareLS(a, lines, b, ORIGIN)
    with pytest.raises(PcpError):
        compute.dispatchPointPlane(b)                              # no normals for this target
    with pytest.raises(PcpError):
        compute.dispatchLS(np.zeros((0, 3)))


# ------------------------------------------------------------------------------ sequences (pairs sharded / batched)
def test_sequence_odometry_recovers_a_synthetic_trajectory():
    """Five sweeps of a sensor driving through the synthetic scene: the chained pair
    registrations (batched on one GPU) reproduce the trajectory, and batch size does not
    change any pose."""
    from pointcloudprocessing_amd import lidar, odometry
    poses = [(lidar.rotation_xyz(0.0, 0.004 * k, 0.0), np.array([0.25 * k, 0.0, 0.01 * k])) for k in range(5)]
    frames = [lidar.synthetic_scan(10 + k, pose_R=R, pose_t=t, rings=25, steps=420) for k, (R, t) in enumerate(poses)]
    got = odometry.sequence_odometry(frames, iterations=12, batch_size=4)
    again = odometry.sequence_odometry(frames, iterations=12, batch_size=3)      # 3 + 1: exercises the short batch
    for (Rw, tw), (Rw2, tw2), (R, t) in zip(got, again, poses):
        assert np.abs(Rw - R).max() < 5e-3 and np.abs(tw - t).max() < 5e-2
        assert np.array_equal(Rw, Rw2) and np.array_equal(tw, tw2)
    # an initial guess that already explains most of the motion is composed with the refinement
    # (point-to-point is not asserted here: on ground-dominated sweeps its optimum is biased
    # towards zero motion, ring matching ring)
    from pointcloudprocessing_amd.odometry import register_pairs
    t_rel, R_rel = poses[1][1] - poses[0][1], poses[1][0]
    guess = (-(t_rel * 0.9), np.eye(3))
    t_opt, R_opt = register_pairs([frames[0]], [frames[1]], iterations=12, guesses=[guess])[0]
    assert np.abs(R_opt - R_rel).max() < 5e-3 and np.abs(-t_opt - t_rel).max() < 5e-2
    t_pp, R_pp = register_pairs([frames[0]], [frames[1]], point_to_plane=False, iterations=3, guesses=[guess])[0]
    assert np.isfinite(t_pp).all() and np.abs(R_pp @ R_pp.T - np.eye(3)).max() < 1e-9


@pytest.mark.parametrize("plane", [False, True])
def test_perturbed_kitti_frame_converges_like_the_reference_scenario(compute, plane):
    """tests/FunctionalTesting.py:348-365 / :456-475: a frame against a copy of itself moved by
    randomRotation(0.01), randomTranslation1() * 0.01 (random.seed(0)); the reference only displays the
    result, here the recovered pose must undo the perturbation."""
    import random
    from pointcloudprocessing_amd import PointcloudAlignment
    a, _ = kitti_pair()
    random.seed(0)
    randrot = PointcloudAlignment.randomRotation(0.01)
    randtrans = PointcloudAlignment.randomTranslation1() * 0.01
    moved = (randrot @ a.T).T + randtrans
    icp = PointcloudIcpContainer(compute)
    if plane:
        t_opt, R_opt = icp.full_pt2pl(a, moved, np.zeros(3), iterations=20)
    else:
        t_opt, R_opt = icp.full_pt2pt(a, moved, np.zeros(3), iterations=40)
    back = (R_opt @ moved.T).T - t_opt
    err = np.linalg.norm(back - a, axis=1)
    tol_R, tol_t, tol_p = (2e-4, 2e-3, 5e-3) if plane else (3e-3, 3e-2, 0.15)    # point-to-point creeps
    assert np.abs(R_opt @ randrot - np.eye(3)).max() < tol_R
    assert np.abs(R_opt @ randtrans - t_opt).max() < tol_t
    assert np.median(err) < tol_p


# ------------------------------------------------------------------------------ ragged ring layouts
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7])
def test_ragged_ring_layouts_match_oracle_bitwise(compute, seed):
    """Arbitrary ring tables, not a LiDAR's: rings of 1 to 60 points (shorter than the +-10 window, of
    length 1, of length a power of two), points the rings do not cover, points near the sensor's pole,
    a non-zero origin.  NN, hints, normals and both tile sums against the restatement, bit for bit."""
    rng = np.random.default_rng(100 + seed)
    lengths = list(rng.integers(1, 61, 30)) + [1, 2, 8, 16, 32, 64, 19, 20, 21]
    rng.shuffle(lengths)
    gaps = rng.integers(0, 3, len(lengths))                       # a few points between rings belong to none
    lines, pos = [], 0
    for L, gp in zip(lengths, gaps):
        pos += int(gp)
        lines.append((pos, pos + int(L) - 1))
        pos += int(L)
    n_a = pos + 5
    a = rng.normal(0, 6, (n_a, 3))
    a[::37] = [0.0, 9.0, 0.0] + rng.normal(0, 0.02, (len(a[::37]), 3))      # almost straight above the sensor
    a[5] = a[4]                                                               # duplicate neighbours
    b = np.vstack([a[rng.integers(0, n_a, 300)] + rng.normal(0, 0.05, (300, 3)), rng.normal(0, 8, (150, 3)),
                   [[0.0, 0.127, 0.0]], [[1e-7, 40.0, -1e-7]]])
    origin = np.array([0.4, -0.2, 0.1]) if seed % 2 else np.zeros(3)
    # nearest neighbour, three passes (all rings, then hinted)
    st = O.RingIcp(100)
    st.prepare_nns(a, lines, origin)
    compute.prepareNNS(a, lines, b, origin)
    for call in range(3):
        idx, dist = compute.dispatchNNS(b)
        ridx, rdist = st.dispatch_nns(b)
        assert np.array_equal(idx, ridx), f"call {call}"
        assert np.array_equal(bits(dist), bits(rdist)), f"call {call}"
        assert np.array_equal(bits(compute.corr_out[:, 3]), bits(st.hint)), f"call {call}"
    # normals
    compute.prepareDispatchNormals(a, lines, origin)
    a4, o4 = O.pad_points(a), O.origin4(origin)
    ref_n = O.normals(a4, o4, O.pad_scan_lines(lines).astype(np.int64), 100)
    assert np.array_equal(bits(compute.normals_out_a[:n_a]), bits(ref_n))
    # tile sums, unpadded (point) and padded (plane) ring tables
    sp = O.RingIcp(100)
    sp.prepare_ls(a, lines, origin)
    compute.prepareLS(a, lines, b, origin)
    H, bb = compute.dispatchLS(b)
    rH, rb = sp.dispatch_ls(b)
    assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb))
    spl = O.RingIcp(100)
    spl.prepare_plane(a, lines, origin)
    compute.preparePointPlane(a, lines, b, origin)
    H, bb = compute.dispatchPointPlane(b)
    rH, rb = spl.dispatch_plane(b)
    assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb))
    # the device loop, whose first iteration finds its rings with several lanes per query, starts from that tile
    # (a pre-pass over all rings, then four waves per 64 points on the hinted rings), starts from that tile and
    # goes on exactly like the batched launches, which search with one lane per point
    from pointcloudprocessing_amd import IcpBatch
    dx, cv, hb, moved = compute.icpRun(b, MODE_PLANE, 3, full_iter=True, want_hb=True, want_points=True)
    assert np.array_equal(bits(hb[0][:6, :6]), bits(rH)) and np.array_equal(bits(hb[0][:6, 6]), bits(rb))
    batch = IcpBatch([compute])
    try:
        batch.upload_sources([b])
        batch.prepare(True)
        batch.launch(MODE_PLANE, 3, full_iter=True)
        (dx2, cv2, hb2, moved2), = batch.results(3, want_hb=True, want_points=True)
    finally:
        batch.close()
    assert len(dx) == len(dx2) and np.array_equal(bits(dx), bits(dx2)) and np.array_equal(bits(hb), bits(hb2))
    assert np.array_equal(moved.view(np.uint64), moved2.view(np.uint64))


# ------------------------------------------------------------------------------ largest ring table
def test_two_thousand_rings_match_oracle_bitwise():
    """maxScans at the library's limit (2048 table rows, 1,500 of them real rings of 3 to 40 points): the
    ring table fills most of a block's LDS.  NN over all rings and hinted, and the plane tile sum."""
    from pointcloudprocessing_amd import ComputeShader, PcpError
    rng = np.random.default_rng(7)
    lengths = rng.integers(3, 41, 1500)
    ends = np.cumsum(lengths)
    lines = [(int(e - L), int(e - 1)) for L, e in zip(lengths, ends)]
    n_a = int(ends[-1])
    el = np.repeat(np.linspace(-0.4, 0.3, len(lengths)), lengths) + rng.normal(0, 1e-3, n_a)
    az = np.concatenate([np.sort(rng.uniform(-np.pi, np.pi, L)) for L in lengths])
    rg = rng.uniform(4.0, 30.0, n_a)
    a = np.column_stack([rg * np.cos(el) * np.cos(az), rg * np.sin(el) + 0.127, rg * np.cos(el) * np.sin(az)])
    b = a[rng.integers(0, n_a, 700)] + rng.normal(0, 0.03, (700, 3))
    with pytest.raises(PcpError):
        ComputeShader(maxPointsPerCloud=n_a, maxScans=2049)
    cs = ComputeShader(maxPointsPerCloud=n_a, maxScans=2048)
    try:
        st = O.RingIcp(2048)
        st.prepare_nns(a, lines, ORIGIN)
        cs.prepareNNS(a, lines, b, ORIGIN)
        for call in range(2):
            idx, dist = cs.dispatchNNS(b)
            ridx, rdist = st.dispatch_nns(b)
            assert np.array_equal(idx, ridx), f"call {call}"
            assert np.array_equal(bits(dist), bits(rdist)), f"call {call}"
            assert np.array_equal(bits(cs.corr_out[:, 3]), bits(st.hint)), f"call {call}"
        spl = O.RingIcp(2048)
        spl.prepare_plane(a, lines, ORIGIN)
        cs.preparePointPlane(a, lines, b, ORIGIN)
        H, bb = cs.dispatchPointPlane(b)
        rH, rb = spl.dispatch_plane(b)
        assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb))
        # more rings than the batched launches' shared list can name: they fall back to the lock-step walk
        from pointcloudprocessing_amd import IcpBatch
        dx, cv, hb, moved = cs.icpRun(b, MODE_PLANE, 3, full_iter=True, want_hb=True, want_points=True)
        assert np.array_equal(bits(hb[0][:6, :6]), bits(rH)) and np.array_equal(bits(hb[0][:6, 6]), bits(rb))
        batch = IcpBatch([cs])
        try:
            batch.upload_sources([b])
            batch.prepare(True)
            batch.launch(MODE_PLANE, 3, full_iter=True)
            (dx2, cv2, hb2, moved2), = batch.results(3, want_hb=True, want_points=True)
        finally:
            batch.close()
        assert np.array_equal(bits(dx), bits(dx2)) and np.array_equal(bits(hb), bits(hb2))
        assert np.array_equal(moved.view(np.uint64), moved2.view(np.uint64))
    finally:
        cs.cleanup()


# ------------------------------------------------------------------------------ non-finite and degenerate points
def test_non_finite_and_degenerate_points_match_oracle_bitwise(compute):
    """Source points that are NaN, infinite, huge, the sensor position itself (zero-length direction) or
    copies of target points, and a target with runs of identical points (undefined normals, which the plane
    kernel zeroes like nan_check, LeastSquaresPlane.glsl:18-24): indices, distances, hints, normals and both
    tile sums against the restatement, bit for bit."""
    a, b = small_pair()
    a = a.copy()
    a[100:104] = a[100]                      # four identical neighbours on one ring
    a[2000] = a[1999]
    b = b.copy()
    b[0] = [np.nan, 1.0, 2.0]
    b[1] = [np.inf, 0.0, 0.0]
    b[2] = [-np.inf, np.inf, 0.0]
    b[3] = [1e20, -1e20, 1e19]
    b[4] = [0.0, 0.127, 0.0]                 # the scanner's centre: direction 0/0
    b[5] = a[100]
    b[6] = a[2000]
    b[7] = [1e-30, 0.127, -1e-30]
    with np.errstate(all="ignore"):
        st, lines = _oracle_state(a, ORIGIN, "nns")
        compute.prepareNNS(a, lines, b, ORIGIN)
        for call in range(2):
            idx, dist = compute.dispatchNNS(b)
            ridx, rdist = st.dispatch_nns(b)
            assert np.array_equal(idx, ridx), f"call {call}"
            assert np.array_equal(bits(dist), bits(rdist)), f"call {call}"
            assert np.array_equal(bits(compute.corr_out[:, 3]), bits(st.hint)), f"call {call}"
        compute.prepareDispatchNormals(a, lines, ORIGIN)
        ref_n = O.normals(O.pad_points(a), O.origin4(ORIGIN), O.pad_scan_lines(lines).astype(np.int64), 100)
        assert np.array_equal(bits(compute.normals_out_a[:len(a)]), bits(ref_n))
        for kind, prep, disp, rdisp in (("ls", compute.prepareLS, compute.dispatchLS, "dispatch_ls"),
                                        ("plane", compute.preparePointPlane, compute.dispatchPointPlane, "dispatch_plane")):
            sp, _ = _oracle_state(a, ORIGIN, kind)
            prep(a, lines, b, ORIGIN)
            H, bb = disp(b)
            rH, rb = getattr(sp, rdisp)(b)
            assert np.array_equal(bits(H), bits(rH)) and np.array_equal(bits(bb), bits(rb)), kind


def test_device_loop_with_non_finite_source_points_matches_oracle(compute):
    """The whole point-to-plane loop with NaN / infinite / far-away source points among the good ones: they
    contribute nothing (nan0, the distance gate) and stay what they are under the float64 transform; every
    iteration's H, b and dx equal the restatement's bit for bit."""
    a, b = small_pair()
    b = b.copy()
    b[10] = [np.nan, np.nan, np.nan]
    b[500] = [np.inf, 1.0, -np.inf]
    b[900] = [4e18, 1.0, 2.0]
    b[1300] = [0.0, 0.127, 0.0]
    icp = PointcloudIcpContainer(compute)
    with np.errstate(all="ignore"):
        tr = []
        O.full_pt2pl(a, b, ORIGIN, iterations=6, trace=tr)
        icp.preparePointToPlane(a, ORIGIN, b)
        dx, conv, hb, moved = compute.icpRun(b, MODE_PLANE, 6, want_hb=True, want_points=True)
    assert len(dx) == len(tr) and len(tr) >= 2
    for i, (H, bb, d) in enumerate(tr):
        assert np.array_equal(bits(hb[i][:6, :6]), bits(H)), f"H at iteration {i}"
        assert np.array_equal(bits(hb[i][:6, 6]), bits(bb)), f"b at iteration {i}"
        assert np.array_equal(bits(dx[i]), bits(d)), f"dx at iteration {i}"
    assert np.isnan(moved[10]).all() and np.isfinite(moved[11]).all()
    # the batched launches (blocks of queries sharing their ring searches) on the same pair: the same steps, tiles and points
    from pointcloudprocessing_amd import IcpBatch
    batch = IcpBatch([compute])
    try:
        batch.upload_sources([b])
        batch.prepare(True)
        batch.launch(MODE_PLANE, 6)
        (dx2, cv2, hb2, moved2), = batch.results(6, want_hb=True, want_points=True)
    finally:
        batch.close()
    assert len(dx2) == len(dx) and np.array_equal(bits(dx2), bits(dx)) and np.array_equal(bits(hb2[:len(dx)]), bits(hb[:len(dx)]))
    assert np.array_equal(moved.view(np.uint64), moved2.view(np.uint64))


def test_transposed_source_arrays_take_the_row_upload_and_change_nothing(compute):
    """A source cloud that is a transposed (3, N) array in memory -- what (R @ p.T).T leaves -- goes up as three
    coordinate rows (pcp_upload_source_soa); steps, tiles and moved points equal those of the (N, 3) copy."""
    a, b = small_pair()
    bt = np.asfortranarray(b)                     # same values, column-major
    assert bt[:, :3].T.flags.c_contiguous and not bt.flags.c_contiguous
    wide = np.asfortranarray(np.hstack([b, np.ones((len(b), 1))]))      # (N, 4), fourth column ignored
    icp = PointcloudIcpContainer(compute)
    icp.preparePointToPlane(a, ORIGIN, b)
    ref = compute.icpRun(np.ascontiguousarray(b), MODE_PLANE, 6, want_hb=True, want_points=True)
    for arr in (bt, wide):
        icp.preparePointToPlane(a, ORIGIN, b)
        got = compute.icpRun(arr, MODE_PLANE, 6, want_hb=True, want_points=True)
        assert np.array_equal(bits(got[0]), bits(ref[0])) and got[1] == ref[1]
        assert np.array_equal(bits(got[2]), bits(ref[2]))
        assert np.array_equal(got[3].view(np.uint64), ref[3].view(np.uint64))


def test_device_tiles_equal_the_scalar_loop_per_point(compute):
    """Identity-correspondence mode (debug_info.x >= 1): every 8 x 8 tile of the point kernel against the
    reference's scalar statement of it, getH_i / getB of to_align[i] and aligned[i] - to_align[i], evaluated in
    float32 like the shader (bit-equal) and in float64 (1e-3, the bar of tests/GpuTest.py:561-582)."""
    icp = PointcloudIcpContainer(compute)
    a = icp.getUniformShape().astype(np.float32).astype(np.float64)
    b = a + np.random.default_rng(2).normal(0, 0.05, a.shape).astype(np.float32)
    b = b.astype(np.float32).astype(np.float64)
    compute.prepareLS(a, [(0, len(a) - 1)], b, ORIGIN, debug_mode=True)     # rings play no part in this mode
    tiles = compute.lsTiles(b, MODE_POINT, debug=True)
    Hs, Bs = icp.getHsBsLoop(a[:300], b[:300])
    assert np.allclose(tiles[:300, :6, :6], Hs, atol=1e-3) and np.allclose(tiles[:300, :6, 6], Bs, atol=1e-3)
    f = np.float32
    for i in (0, 17, 299):
        J = icp.getJ_i(b[i]).astype(f)
        e = (a[i].astype(f) - b[i].astype(f)).astype(f)
        H32 = np.zeros((6, 6), f)
        b32 = np.zeros(6, f)
        for r in range(6):
            for c in range(6):
                acc = f(J[0][r] * J[0][c])
                acc = f(acc + f(J[1][r] * J[1][c]))
                H32[r][c] = f(acc + f(J[2][r] * J[2][c]))
            acc = f(J[0][r] * e[0])
            acc = f(acc + f(J[1][r] * e[1]))
            b32[r] = f(acc + f(J[2][r] * e[2]))
        assert np.array_equal(bits(tiles[i, :6, :6]), bits(H32)) and np.array_equal(bits(tiles[i, :6, 6]), bits(b32)), i
