"""Exact-neighbour path (cell grid) against the reference's KD-tree outputs (tests/golden/ref_kd.npz,
ref_kat.npz), scipy's KDTree itself, and brute force where ties matter."""
import numpy as np
import pytest

from oracle import np_grid as NG
from oracle import np_oracle as O
from pointcloudprocessing_amd import PointcloudIcpContainer, lidar
from pointcloudprocessing_amd.cell_grid import CellGrid, cell_keys
from tests._data import bits, free_port, golden, kitti_pair

pytestmark = pytest.mark.gpu


def _spread(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    for sh, mask in ((32, 0x1F00000000FFFF), (16, 0x1F0000FF0000FF), (8, 0x100F00F00F00F00F),
                     (4, 0x10C30C30C30C30C3), (2, 0x1249249249249249)):
        v = (v | (v << np.uint64(sh))) & np.uint64(mask)
    return v


def morton(c):
    return _spread(c[:, 0]) | (_spread(c[:, 1]) << np.uint64(1)) | (_spread(c[:, 2]) << np.uint64(2))


def cell_keys_abs(points, cell):
    """The grid's cell key in NumPy: absolute cell coordinates floor(p / cell), zig-zag per axis
    (0, -1, 1, -2, ... -> 0, 1, 2, 3, ...), bits interleaved."""
    c = np.floor(np.asarray(points) / cell).astype(np.int64)
    z = np.where(c >= 0, 2 * c, -2 * c - 1).astype(np.uint64)
    return morton(z), c


def assert_same_neighbours(a, q, idx, dist, ref_idx, ref_dist):
    """Equal distances everywhere; indices may differ from the KD-tree's only on exact ties, where
    the grid returns the lowest index."""
    ok = np.isfinite(ref_dist)
    assert np.array_equal(np.isfinite(dist), ok)
    assert np.allclose(dist[ok], ref_dist[ok], rtol=1e-15, atol=0)
    assert np.array_equal(idx[~ok], ref_idx[~ok])
    diff = np.nonzero(ok & (idx != ref_idx))[0]
    assert len(diff) <= 0.01 * len(q) + 1
    if len(diff):
        d_ours = np.linalg.norm(q[diff] - a[idx[diff]], axis=1)
        d_ref = np.linalg.norm(q[diff] - a[ref_idx[diff]], axis=1)
        assert np.array_equal(d_ours, d_ref) and np.all(idx[diff] < ref_idx[diff])


@pytest.fixture(scope="module")
def frames():
    return kitti_pair()


@pytest.fixture(scope="module")
def grid0(frames):
    g = CellGrid(frames[0], cell=0.5)
    yield g
    g.cleanup()


def test_build_sorts_by_cell_key_and_keeps_every_point(frames, grid0):
    a = frames[0]
    info = grid0.info()
    ids, xyz = grid0.sorted_points()
    assert info["points"] == len(a) and np.array_equal(np.sort(ids), np.arange(len(a)))
    assert np.array_equal(xyz, a[ids])
    keys, c = cell_keys_abs(a[ids], 0.5)
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)
    same = np.diff(keys.astype(np.int64)) == 0
    assert np.all(np.diff(ids)[same] > 0)                    # stable: original order inside a cell
    assert info["cells"] == len(np.unique(keys))
    assert info["extent"] == (c.max(axis=0) - c.min(axis=0) + 1).tolist()
    # the frame straddles the origin in every axis: zig-zag values up to 2 * max|c|, their bits interleaved,
    # one radix pass per 8 key bits that can vary
    bits_needed = int(np.ceil(np.log2(2 * np.abs(c).max() + 2)))
    assert info["sort_passes"] == -(-3 * bits_needed // 8)


def test_query_equals_reference_kdtree_on_kitti(frames, grid0):
    g = golden("ref_kd.npz")
    q = frames[1][::7]
    dist, idx = grid0.query(q)
    assert_same_neighbours(frames[0], q, idx, dist, g["kd7_idx"].astype(np.int64), g["kd7_dist"])
    k = golden("ref_kat.npz")
    q = frames[1][::int(k["kd_query_stride"])]
    dist, idx = grid0.query(q)
    assert_same_neighbours(frames[0], q, idx, dist, k["kd_idx"], k["kd_dist"])


@pytest.mark.parametrize("cell", [0.05, 0.3, 1.7, 40.0, 1e4])
def test_query_is_independent_of_the_cell_size(frames, cell):
    a, b = frames[0][::5], frames[1][::11]
    ref_d, ref_i = NG.exact_nn(a, b)
    g = CellGrid(a, cell=cell)
    try:
        dist, idx = g.query(b)
    finally:
        g.cleanup()
    assert_same_neighbours(a, b, idx, dist, ref_i, ref_d)


def test_refined_grid_gives_the_same_neighbours(frames):
    """CellGrid(refine=N) halves the cell edge until an occupied cell holds at most N points: faster, same answers."""
    a, b = frames[0], frames[1][::11]
    coarse, fine = CellGrid(a, cell=4.0), CellGrid(a, cell=4.0, refine=16)
    try:
        assert fine.cell in (2.0, 1.0, 0.5, 0.25) and fine.info()["points"] <= 16 * fine.info()["cells"]
        assert coarse.info()["points"] > 16 * coarse.info()["cells"]
        d0, i0 = coarse.query(b)
        d1, i1 = fine.query(b)
        assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    finally:
        coarse.cleanup()
        fine.cleanup()


def test_ties_go_to_the_lowest_index_and_distances_are_bitwise():
    rng = np.random.default_rng(5)
    a = rng.integers(-6, 7, (3000, 3)).astype(np.float64) * 0.25       # many duplicates and ties
    q = rng.integers(-8, 9, (700, 3)).astype(np.float64) * 0.25 + 0.125
    ref_d, ref_i = NG.brute_nn(a, q)
    for cell in (0.25, 0.6, 3.0):
        g = CellGrid(a, cell=cell)
        try:
            dist, idx = g.query(q)
        finally:
            g.cleanup()
        assert np.array_equal(idx, ref_i) and np.array_equal(bits64(dist), bits64(ref_d))


def bits64(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def test_distance_upper_bound_and_far_queries(frames):
    a = frames[0][::3]
    rng = np.random.default_rng(9)
    q = np.vstack([frames[1][::50], rng.uniform(-500, 500, (300, 3)), [[1e6, -1e6, 3e5]], [[0, 0, 0]]])
    g = CellGrid(a, cell=0.5)
    try:
        for bound in (0.05, 0.4, 3.0):
            ref_d, ref_i = NG.exact_nn(a, q, bound)
            dist, idx = g.query(q, distance_upper_bound=bound)
            assert_same_neighbours(a, q, idx, dist, ref_i, ref_d)          # misses are (inf, n), like scipy
        ref_d, ref_i = NG.exact_nn(a, q)
        dist, idx = g.query(q)
        assert_same_neighbours(a, q, idx, dist, ref_i, ref_d)
        # nothing is strictly nearer than a bound of 0 (or below): all misses, as scipy's KDTree.query answers
        for bound in (0.0, -1.0, float("nan")):
            dist, idx = g.query(q, distance_upper_bound=bound)
            assert np.all(np.isinf(dist)) and np.all(idx == len(a))
    finally:
        g.cleanup()


def test_degenerate_maps():
    q = np.array([[0.0, 0.0, 0.0], [5.0, -2.0, 1.0]])
    g = CellGrid(np.zeros((0, 3)), cell=1.0)
    try:
        dist, idx = g.query(q)
        assert np.all(np.isinf(dist)) and np.all(idx == 0)
        g.rebuild(np.array([[1.0, 2.0, 3.0]]), cell=0.1)
        dist, idx = g.query(q)
        assert np.array_equal(idx, [0, 0]) and np.allclose(dist, np.linalg.norm(q - [1, 2, 3], axis=1), rtol=1e-15)
        same = np.tile([[0.5, 0.5, 0.5]], (5000, 1))                 # one cell, 5000 duplicates
        g.rebuild(same, cell=1.0)
        dist, idx = g.query(q)
        assert np.array_equal(idx, [0, 0])
        with pytest.raises(RuntimeError):
            g.rebuild(np.array([[np.nan, 0, 0]]), cell=1.0)
        with pytest.raises(RuntimeError):
            g.rebuild(np.array([[1e9, 0, 0]]), cell=0.5)
    finally:
        g.cleanup()


def test_kd_step_and_align_kd_equal_the_reference(frames):
    g = golden("ref_kd.npz")
    icp = PointcloudIcpContainer(None)
    tree = icp.getKDTree(frames[0], cell=0.5)
    try:
        R, t = icp.icp_step_KD(tree, frames[0], frames[1])
        # KITTI coordinates are millimetre-quantised: a few hundred queries have two DIFFERENT map points
        # at exactly the same distance, where the KD-tree's pick is arbitrary; each moves R by ~1e-9
        assert np.abs(R - g["step_R"]).max() < 1e-6 and np.abs(t - g["step_t"]).max() < 1e-5
        idx, dist, sel = tree.nn_results(selected=True)
        ref, other = frames[0][idx[sel]], frames[1][sel]             # Kabsch of exactly this trimmed set
        mr, mo = ref.mean(axis=0), other.mean(axis=0)
        U, S, Vh = np.linalg.svd((ref - mr).T @ (other - mo))
        assert np.abs(R - U @ Vh).max() < 1e-12 and np.abs(t - (mr - U @ Vh @ mo)).max() < 1e-10
        ref_d, ref_i = NG.exact_nn(frames[0], frames[1])
        assert_same_neighbours(frames[0], frames[1], idx, dist, ref_i, ref_d)
        taken = int(len(ref_d) * 0.8)
        assert sel.sum() == taken and dist[sel].max() <= dist[~sel].min()
    finally:
        tree.cleanup()
    moved, R_opt, t_opt = icp.align_KD(frames[0], frames[1])
    assert np.abs(R_opt - g["align_R"]).max() < 1e-5 and np.abs(t_opt - g["align_t"]).max() < 1e-4
    assert np.abs(moved[::64] - g["align_moved_sub"]).max() < 1e-3


def test_trimmed_selection_breaks_ties_by_index():
    a = np.array([[0.0, 0, 0], [10.0, 0, 0]])
    q = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0], [-1.0, 0, 0], [11.0, 0, 0], [12.0, 0, 0], [0, 2.0, 0],
                  [0, -1.0, 0], [3.0, 0, 0], [10.0, 1.0, 0]])
    g = CellGrid(a, cell=1.0)
    try:
        g.set_source(q)
        g.kd_step(0.4)                                            # 4 of the 6 unit distances
        idx, dist, sel = g.nn_results(selected=True)
        assert np.array_equal(np.nonzero(sel)[0], [0, 1, 2, 3])
        g.kd_step(1.0)
        assert g.nn_results(selected=True)[2].all()
    finally:
        g.cleanup()


@pytest.mark.parametrize("mode", [0, 1])
def test_map_ls_tile_equals_checker_bitwise(frames, mode):
    a, b = frames
    rng = np.random.default_rng(2)
    nrm = rng.normal(size=(len(a), 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    g = CellGrid(a, cell=0.5, normals=nrm)
    try:
        g.set_source(b)
        H, bb = g.ls(mode, 0.5)
        ref, hit, ref_idx = NG.map_ls_tile(a, nrm, b, mode, 0.5)
        assert np.array_equal(bits(H), bits(ref[:6, :6])) and np.array_equal(bits(bb), bits(ref[:6, 6]))
        idx, dist = g.nn_results()
        assert np.array_equal(idx >= 0, hit) and np.array_equal(idx[hit], ref_idx[hit])
        # split of the key space: the two partial tiles are those of the masked checker, their sum the whole
        keys = cell_keys(b, 0.5)
        cut = int(np.sort(keys)[len(keys) // 2])
        parts = []
        for lo, hi in ((0, cut), (cut, 1 << 63)):
            g.set_owned_keys(lo, hi)
            Hp, bp = g.ls(mode, 0.5)
            owned = (keys >= np.uint64(lo)) & (keys < np.uint64(hi))
            refp, _, _ = NG.map_ls_tile(a, nrm, b, mode, 0.5, owned)
            assert np.array_equal(bits(Hp), bits(refp[:6, :6])) and np.array_equal(bits(bp), bits(refp[:6, 6]))
            parts.append(np.column_stack([Hp, bp]))
        assert np.allclose(parts[0] + parts[1], np.column_stack([H, bb]), rtol=2e-5, atol=1e-3)
    finally:
        g.cleanup()


def test_scan_to_map_registration_recovers_a_known_motion():
    scans = [lidar.synthetic_scan(s, rings=33, steps=600) for s in (1, 2, 3)]
    world = np.vstack(scans)                                       # a small "map" of three sweeps
    R = lidar.rotation_xyz(0.01, -0.02, 0.015)
    t = np.array([0.06, -0.03, 0.04])
    rng = np.random.default_rng(4)
    src = (R @ scans[1].T).T + t + rng.normal(0, 0.003, scans[1].shape)
    g = CellGrid(world, cell=0.5)
    icp = PointcloudIcpContainer(None)
    try:
        g.set_source(src)
        dx, conv, hb = g.icp_run(0, 0.4, max_iter=60)
        t_opt, R_opt = icp._replay(dx, len(dx))
        # point-to-point converges linearly (points slide along the rings): 20 mrad of yaw come down to < 5
        assert np.abs(R_opt @ R - np.eye(3)).max() < 5e-3 and np.abs(R_opt @ t - t_opt).max() < 3e-2
        moved = g.read_source()
        assert np.abs(moved - ((R_opt @ src.T).T - t_opt)).max() < 1e-4
        # the loop equals stepwise dispatches: first tile of the trace = one dispatch on the fresh cloud
        g.rewind_source()
        H, b = g.ls(0, 0.4)
        assert np.array_equal(bits(H), bits(hb[0][:6, :6])) and np.array_equal(bits(b), bits(hb[0][:6, 6]))
        dx0 = O.solve6(H, b)
        assert np.array_equal(bits(dx0), bits(dx[0]))
    finally:
        g.cleanup()


def test_million_point_map_self_query():
    rng = np.random.default_rng(1)
    cloud = lidar.synthetic_scan(3, rings=65, steps=19000)
    q = cloud + rng.normal(0, 0.05, cloud.shape)                  # config 3: self-query with 5 cm jitter
    g = CellGrid(cloud, cell=0.5)
    try:
        info = g.info()
        dist, idx, ms = g.query(q, return_kernel_ms=True)
        sel = np.arange(0, len(q), 50)
        ref_d, ref_i = NG.exact_nn(cloud, q[sel])
        assert_same_neighbours(cloud, q[sel], idx[sel], dist[sel], ref_i, ref_d)
        own = np.linalg.norm(q - cloud, axis=1)
        assert np.all(dist <= own * (1 + 1e-12))
        print(f"\n1M map: build {info['build_ms']:.2f} ms (sort {info['sort_ms']:.2f} ms, {info['sort_passes']} passes, "
              f"{info['cells']} cells), query {ms:.2f} ms = {len(q) / ms / 1e3:.1f} M queries/s")
    finally:
        g.cleanup()


def _map_scene():
    scans = [lidar.synthetic_scan(s, rings=33, steps=600) for s in (1, 2, 3)]
    world = np.vstack(scans)
    R = lidar.rotation_xyz(0.004, -0.006, 0.005)
    t = np.array([0.03, -0.01, 0.02])
    rng = np.random.default_rng(4)
    src = (R @ scans[1].T).T + t + rng.normal(0, 0.002, scans[1].shape)
    nrm = rng.normal(size=(len(world), 3)).astype(np.float32)
    nrm[:, 1] += 3.0                                              # mostly "up": a well-conditioned plane system
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    return world, nrm, src


@pytest.mark.parametrize("mode", [0, 1])
def test_map_registration_driver_equals_the_fused_loop_bitwise(mode):
    from pointcloudprocessing_amd import scan_to_map as S
    world, nrm, src = _map_scene()
    g = CellGrid(world, cell=0.5, normals=nrm)
    reg = S.MapRegistration(world, 0.5, normals=nrm)
    try:
        g.set_source(src)
        dx_ref, conv_ref, hb_ref = g.icp_run(mode, 0.4, max_iter=8)
        t_opt, R_opt, dx, conv = reg.register(src, mode, 0.4, max_iter=8)
        assert conv == conv_ref and np.array_equal(bits(dx), bits(dx_ref))
        assert np.array_equal(bits(reg.last_tiles), bits(hb_ref))
        assert np.array_equal(reg.moved_scan(), g.read_source())
        # and the first tile is the checker's
        ref, _, _ = NG.map_ls_tile(world, nrm, src, mode, 0.4)
        assert np.array_equal(bits(hb_ref[0]), bits(ref))
    finally:
        g.cleanup()
        reg.close()


# ------------------------------------------------------------------------------ a map that accumulates (pcp_grid_append)
def _same_grid(g_inc, g_all, queries):
    ids_i, xyz_i = g_inc.sorted_points()
    ids_a, xyz_a = g_all.sorted_points()
    assert np.array_equal(ids_i, ids_a) and np.array_equal(xyz_i, xyz_a)            # same order, entry by entry
    ii, ia = g_inc.info(), g_all.info()
    assert (ii["points"], ii["cells"], ii["extent"]) == (ia["points"], ia["cells"], ia["extent"])
    d_i, i_i = g_inc.query(queries)
    d_a, i_a = g_all.query(queries)
    assert np.array_equal(i_i, i_a) and np.array_equal(bits64(d_i), bits64(d_a))
    d_i, i_i = g_inc.query(queries, distance_upper_bound=0.3)
    d_a, i_a = g_all.query(queries, distance_upper_bound=0.3)
    assert np.array_equal(i_i, i_a) and np.array_equal(bits64(d_i), bits64(d_a))


def test_appending_frames_equals_building_the_whole(frames):
    """Frame 1 ingested into the grid of frame 0 (moved a little, and far away once so that the bounding box
    grows on the negative and the positive side) = the grid built from all points at once."""
    a, b = frames
    far = b[:5000] + np.array([-700.0, 35.0, 900.0])
    batches = [a, b + np.array([0.3, 0.0, 0.1]), far, a[::3] * np.array([1.0, 1.0, -1.0])]
    g_inc = CellGrid(batches[0], cell=0.5)
    g_all = CellGrid(np.concatenate(batches), cell=0.5)
    try:
        for extra in batches[1:]:
            g_inc.append(extra)
        _same_grid(g_inc, g_all, np.concatenate([b[::11], far[::7] + 0.05]))
        # starting from an empty map, and a batch of one point
        g_inc.rebuild(np.zeros((0, 3)), cell=0.5)
        for extra in batches + [np.array([[1e3, -2.0, 1e3]])]:
            g_inc.append(extra)
        g_all.rebuild(np.concatenate(batches + [np.array([[1e3, -2.0, 1e3]])]), cell=0.5)
        _same_grid(g_inc, g_all, b[::29])
    finally:
        g_inc.cleanup()
        g_all.cleanup()


def test_collected_appends_equal_appends_one_by_one(frames):
    """pcp_grid_collect_begin / _end: batches handed over inside are merged into the map together at the end -- the same
    map (ids, order, cells, answers) as an append per batch; nothing may read the map in between."""
    from pointcloudprocessing_amd._lib import PcpError
    a, b = frames
    batches = [b[:40000] + np.array([0.3, 0.0, 0.1]), b[40000:40007], a[::3] * np.array([1.0, 1.0, -1.0]),
               b[:5000] + np.array([-700.0, 35.0, 900.0])]
    g_one, g_col = CellGrid(a, cell=0.5), CellGrid(a, cell=0.5)
    try:
        for extra in batches:
            g_one.append(extra)
        with g_col.collect():
            for extra in batches:
                g_col.append(extra)
            with pytest.raises(PcpError, match="being collected"):
                g_col.query(b[:10])
            with pytest.raises(PcpError, match="being collected"):
                g_col.info()
        _same_grid(g_one, g_col, np.concatenate([b[::11], batches[3][::7] + 0.05]))
        with g_col.collect():                      # nothing handed over: the map stays as it is
            pass
        _same_grid(g_one, g_col, b[::29])
        with pytest.raises(PcpError, match="normals"):
            with g_col.collect():
                g_col.append(b[:100], normals=np.ones((100, 3), np.float32))
    finally:
        g_one.cleanup()
        g_col.cleanup()


def test_deferred_appends_join_the_map_when_it_is_read(frames):
    """pcp_grid_defer_appends: batches wait on the device; the first reader takes them in, all at once -- the map an append per
    batch leaves.  A collection the caller opens adopts what waits; a rebuild drops it with the map."""
    a, b = frames
    batches = [b[:40000] + np.array([0.3, 0.0, 0.1]), b[40000:40007], a[::3] * np.array([1.0, 1.0, -1.0]),
               b[:5000] + np.array([-700.0, 35.0, 900.0])]
    g_one, g_def = CellGrid(a, cell=0.5), CellGrid(a, cell=0.5)
    try:
        g_def.defer_appends(True)
        for extra in batches[:2]:
            g_one.append(extra)
            g_def.append(extra)
        _same_grid(g_one, g_def, b[::11])                          # info() / query() inside are the readers that flush
        g_one.append(batches[2])
        g_def.append(batches[2])
        with g_def.collect():                                      # adopts the batch that waits
            g_def.append(batches[3])
        g_one.append(batches[3])
        _same_grid(g_one, g_def, np.concatenate([b[::13], batches[3][::7] + 0.05]))
        g_def.append(b[:100] + 5.0)                                # waits ...
        g_def.rebuild(a, cell=0.5)                                 # ... and goes with the map it was meant for
        g_one.rebuild(a, cell=0.5)
        _same_grid(g_one, g_def, b[::17])
        g_def.append(batches[0])
        g_def.defer_appends(False)                                 # switching off flushes
        g_one.append(batches[0])
        assert g_def.info()["points"] == g_one.info()["points"] == len(a) + len(batches[0])
        _same_grid(g_one, g_def, b[::19])
    finally:
        g_one.cleanup()
        g_def.cleanup()


def test_appending_with_normals_and_the_rules_around_them(frames):
    a, b = frames[0][::5], frames[1][::5]
    rng = np.random.default_rng(2)
    na, nb = rng.normal(size=(len(a), 3)).astype(np.float32), rng.normal(size=(len(b), 3)).astype(np.float32)
    g_inc = CellGrid(a, cell=0.5, normals=na)
    g_all = CellGrid(np.concatenate([a, b]), cell=0.5, normals=np.concatenate([na, nb]))
    try:
        with pytest.raises(RuntimeError, match="normals"):
            g_inc.append(b)                                    # the caller's normals cannot be left out
        g_inc.append(b, normals=nb)
        src = b[::3] + 0.01
        g_inc.set_source(src)
        g_all.set_source(src)
        Hi, bi = g_inc.ls(1, 0.4)
        Ha, ba = g_all.ls(1, 0.4)
        assert np.array_equal(bits(Hi), bits(Ha)) and np.array_equal(bits(bi), bits(ba))     # the plane tiles see the same normals
        # estimated normals are of the map as it was: an append drops them
        g_inc.rebuild(a, cell=0.5)
        g_inc.estimate_normals(0.5)
        g_inc.append(b)
        g_inc.set_source(src)
        with pytest.raises(RuntimeError, match="normals"):
            g_inc.ls(1, 0.4)
        # ids go up from batch to batch (equal cells keep resident points in front of new ones)
        with pytest.raises(RuntimeError, match="below the ids"):
            g_inc.append(b, first_id=3)
    finally:
        g_inc.cleanup()
        g_all.cleanup()


def test_ten_million_point_map_grown_by_ingests_equals_its_build(frames):
    """One GPU's share of the accumulated-map configuration: 100 sweeps along a drive (12.3 M points) fed in ten
    ingests of ten sweeps, against the build of the concatenation; then a scan registered on both."""
    a, b = frames
    rng = np.random.default_rng(7)
    sweeps = [a + np.array([0.05 * k, 0.0, -0.33 * k]) + rng.normal(0, 0.004, a.shape) for k in range(100)]
    batches = [np.concatenate(sweeps[k:k + 10]) for k in range(0, 100, 10)]
    whole = np.concatenate(batches)
    assert len(whole) > 10_000_000
    g_inc = CellGrid(batches[0], cell=0.25)
    g_all = CellGrid(whole, cell=0.25)
    try:
        for extra in batches[1:]:
            g_inc.append(extra)
        scan = b + np.array([0.02, 0.0, -16.5])
        _same_grid(g_inc, g_all, scan[::3])
        runs = []
        for g in (g_inc, g_all):
            g.set_source(scan)
            runs.append(g.icp_run(0, 0.25, max_iter=5))
        assert np.array_equal(bits(runs[0][0]), bits(runs[1][0])) and np.array_equal(bits(runs[0][2]), bits(runs[1][2]))
    finally:
        g_inc.cleanup()
        g_all.cleanup()


def test_staged_voxels_of_the_voxelizer_accumulate_in_the_map(frames, tmp_path):
    """Voxelizer -> MapSink -> CellGrid: what the reference's Voxelizer hands to its VoxelManager for drawing
    (Voxelizer.py:130-133) is the map here.  Every staged point is in the grid afterwards, nothing else is."""
    from pointcloudprocessing_amd import ComputeShader, MapSink, Voxelizer
    a, b = frames
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    grid = CellGrid(np.zeros((0, 3)), cell=0.5)
    sink = MapSink(grid)
    seen = []

    class Recording(MapSink):
        def frameVoxelized(self, pts, colors, separate):
            seen.append(np.asarray(pts, dtype=np.float64))
            MapSink.frameVoxelized(self, pts, colors, separate)

    sink = Recording(grid)
    vox = Voxelizer(cs, None, voxman=sink)
    try:
        vox.init(0.5)
        vox.separate_colors, vox.filter_outliers = False, False
        for k in range(4):                                     # the same street seen four times: voxels fill up and get staged
            vox.addPoints(a + np.array([0.0, 0.0, -0.33 * k]), None)
        vox.stageAllRemaningVoxels()
        staged = np.concatenate(seen)
        assert sink.points_fed == len(staged) > 100000 and grid.info()["points"] == len(staged)
        ids, xyz = grid.sorted_points()
        assert np.array_equal(np.sort(ids), np.arange(len(staged))) and np.array_equal(xyz, staged[ids])
        d, i = grid.query(staged[::17])
        assert np.all(d == 0.0)
        # the map leaves the device in the reference's PLY format, in feeding order
        from pointcloudprocessing_amd.map_output import read_ply_ascii, save_grid_ply
        back = read_ply_ascii(save_grid_ply(str(tmp_path / "map"), grid))
        assert back.shape == (len(staged), 6) and np.allclose(back[:, :3], staged, atol=6e-7) and np.all(back[:, 3:] == 1.0)
        # and a scan registers against the accumulated map
        grid.set_source(b[::2])
        dx, conv, _ = grid.icp_run(0, 0.5, max_iter=6)
        assert len(dx) == 6 and np.abs(dx[-1]).max() < np.abs(dx[0]).max()
    finally:
        vox.cleanup()
        grid.cleanup()
        cs.cleanup()


def test_staged_voxels_reach_the_map_without_the_host(frames):
    """The device copy of the point store: the rows a stager dispatch releases are resolved to points and appended
    on the device (pcp_voxel_staged_points + pcp_grid_append_dev / _slab_dev).  Same map as the host gather of the
    reference's hand-off (Voxelizer.py:130-133, :396-399) feeds -- ids, order, coordinates -- whole and as one
    rank's slab of a split map."""
    from pointcloudprocessing_amd import ComputeShader, MapSink, Voxelizer
    from pointcloudprocessing_amd import scan_to_map as S
    a, b = frames
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    grid_host, grid_dev = CellGrid(np.zeros((0, 3)), cell=0.5), CellGrid(np.zeros((0, 3)), cell=0.5)
    seed = a[::50]
    regs = [S.MapRegistration(seed, 1.0, rank=1, world=2, nn_cell=0.5) for _ in range(2)]
    host_sinks = [MapSink(grid_host), MapSink(regs[0])]
    dev_sinks = [MapSink(grid_dev, compute=cs), MapSink(regs[1], compute=cs)]
    vox = Voxelizer(cs, None, voxman=dev_sinks[0])
    plain = vox._hand_off

    def both(rows):
        for v_slice in rows:                                   # the reference's hand-off, on the host
            pts, cols = vox.getStoredPointsFromVoxelDataSlice(v_slice)
            for sink in host_sinks:
                sink.frameVoxelized(pts, cols, False)
        for sink in host_sinks:
            sink.flush()
        plain(rows)                                            # the device path into the whole map ...
        if len(rows):
            dev_sinks[1].framesStagedOnDevice()                # ... and into the rank's slab

    vox._hand_off = both
    try:
        vox.init(0.5)
        vox.separate_colors, vox.filter_outliers = False, False
        for k in range(4):
            vox.addPoints(a + np.array([0.0, 0.0, -0.33 * k]), None)
        vox.stageAllRemaningVoxels()
        assert dev_sinks[0].points_fed == host_sinks[0].points_fed > 100000
        assert dev_sinks[0].voxels_fed >= host_sinks[0].voxels_fed > 1000      # the host sink skips empty rows
        ih, xh = grid_host.sorted_points()
        idv, xd = grid_dev.sorted_points()
        assert np.array_equal(ih, idv) and np.array_equal(xh, xd)
        assert grid_host.info()["cells"] == grid_dev.info()["cells"]
        assert 0 < regs[0].slab_points == regs[1].slab_points < regs[0].total_points == regs[1].total_points
        ih, xh = regs[0].backend.grid.sorted_points()
        idv, xd = regs[1].backend.grid.sorted_points()
        assert np.array_equal(ih, idv) and np.array_equal(xh, xd)
        # a store that was not kept cannot resolve anything
        cs2 = ComputeShader(maxPointsPerCloud=1000)
        with pytest.raises(Exception, match="point store is off"):
            cs2.stagedPointsDevice()
        cs2.cleanup()
    finally:
        vox.cleanup()
        for g in (grid_host, grid_dev):
            g.cleanup()
        for r in regs:
            r.close()
        cs.cleanup()


def test_ingest_routed_by_key_range_equals_the_slab_of_the_whole(frames):
    """Two ranks' shares, one after the other on this GPU: each rank's slab grown by ingests (every rank is handed
    the same batch and keeps its part) equals its slab of the whole map."""
    from pointcloudprocessing_amd import scan_to_map as S
    a, b = frames
    first, more = a, [b + np.array([0.3, 0.0, 0.1]), a[::2] + np.array([25.0, 0.0, -40.0])]
    whole = np.concatenate([first] + more)
    ranges = S.HipMapBackend(0).partition(whole, 1.0, 2)
    for rank in range(2):
        reg_inc = S.MapRegistration(first, 1.0, rank=rank, world=2, nn_cell=0.5)
        reg_inc.ranges = ranges                      # the cut of the WHOLE map: fixed before the first ingest
        reg_inc.backend.build(first, None, 1.0, 0.5, slab=ranges[rank])
        reg_inc.backend.set_owned(*ranges[rank])
        reg_inc.slab_points = reg_inc.backend.held
        reg_all = S.MapRegistration(whole, 1.0, rank=rank, world=2, nn_cell=0.5)
        try:
            assert reg_all.ranges == ranges
            for extra in more:
                reg_inc.ingest(extra)
            assert reg_inc.slab_points == reg_all.slab_points and reg_inc.total_points == len(whole)
            _same_grid(reg_inc.backend.grid, reg_all.backend.grid, b[::13])
        finally:
            reg_inc.close()
            reg_all.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_exchange_step_inside_the_library_single_rank_communicator(mode):
    """pcp_comm_* / pcp_grid_icp_run_comm with a world of one: the RCCL all-reduce of one tile is the identity,
    so both the one-call loop and the per-iteration driver must reproduce pcp_grid_icp_run bit for bit.  (A world
    of two needs two GPUs: RCCL refuses two ranks on one device.)"""
    from pointcloudprocessing_amd import scan_to_map as S
    world, nrm, src = _map_scene()
    g = CellGrid(world, cell=0.5, normals=nrm)
    comm = S.RcclComm(rank=0, world=1, device=0)
    reg = S.MapRegistration(world, 0.5, normals=nrm, comm=comm)
    try:
        g.set_source(src)
        dx_ref, conv_ref, hb_ref = g.icp_run(mode, 0.4, max_iter=8)
        for check_every in (None, 1):                       # one ABI call / stepwise with the all-reduce per iteration
            t_opt, R_opt, dx, conv = reg.register(src, mode, 0.4, max_iter=8, check_every=check_every)
            assert conv == conv_ref and np.array_equal(bits(dx), bits(dx_ref)), check_every
            assert np.array_equal(bits(reg.last_tiles), bits(hb_ref))
            assert np.array_equal(reg.moved_scan(), g.read_source())
    finally:
        g.cleanup()
        reg.close()
        comm.close()


MAP_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pointcloudprocessing_amd import scan_to_map as S
from tests.test_gpu_grid import _map_scene
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
pts, nrm, src = _map_scene()
reg = S.MapRegistration(pts, 0.5, normals=nrm, rank=rank, world=world, device=0, nn_cell=0.2)
out = {"slab": reg.slab_points}
for mode in (0, 1):
    t_opt, R_opt, dx, conv = reg.register(src, mode, 0.4, max_iter=8)
    out.update({f"t{mode}": t_opt, f"R{mode}": R_opt, f"dx{mode}": dx, f"moved{mode}": reg.moved_scan()})
np.savez(os.path.join(sys.argv[2], f"map_{rank}.npz"), **out)
dist.barrier()
reg.close()
dist.destroy_process_group()
'''


def test_two_ranks_sharing_the_gpu_register_like_one(tmp_path):
    """The split map on real kernels: 2 processes on this GPU, tiles exchanged over gloo."""
    import os
    import subprocess
    import sys
    from pointcloudprocessing_amd import scan_to_map as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pts, nrm, src = _map_scene()
    single = S.MapRegistration(pts, 0.5, normals=nrm)
    script = tmp_path / "worker.py"
    script.write_text(MAP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=200) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    r0, r1 = np.load(tmp_path / "map_0.npz"), np.load(tmp_path / "map_1.npz")
    assert int(r0["slab"]) < len(pts) and int(r1["slab"]) < len(pts)
    try:
        for mode in (0, 1):
            t1, R1, dx1, _ = single.register(src, mode, 0.4, max_iter=8)
            assert np.array_equal(r0[f"dx{mode}"], r1[f"dx{mode}"])            # both ranks solve the same system
            assert np.array_equal(r0[f"moved{mode}"], r1[f"moved{mode}"])
            assert r0[f"dx{mode}"].shape == dx1.shape and np.abs(r0[f"dx{mode}"] - dx1).max() < 5e-6
            assert np.abs(r0[f"R{mode}"] - R1).max() < 1e-5 and np.abs(r0[f"t{mode}"] - t1).max() < 1e-4
            assert np.abs(r0[f"moved{mode}"] - single.moved_scan()).max() < 1e-4
    finally:
        single.close()


def test_device_partition_and_slab_equal_the_numpy_statement():
    from pointcloudprocessing_amd import scan_to_map as S
    from pointcloudprocessing_amd.cell_grid import partition_keys
    pts, nrm, src = _map_scene()
    for world in (1, 2, 3, 5):
        ranges = partition_keys(pts, 0.5, world)
        assert ranges == S.split_key_space(pts, 0.5, world)
        for lo, hi in ranges:
            mask = S.slab_mask(pts, 0.5, lo, hi)
            g = CellGrid(pts, cell=0.2, normals=nrm, slab=(0.5, lo, hi))
            try:
                ids, xyz = g.sorted_points()
                assert g.held == mask.sum() == len(ids) and np.array_equal(np.sort(ids), np.nonzero(mask)[0])
                assert np.array_equal(xyz, pts[ids])
                # neighbours are reported by their index in the whole map
                q = src[::9]
                dist, idx = g.query(q, distance_upper_bound=0.5)
                ref_d, ref_i = NG.exact_nn_lowest(pts[mask], q, 0.5)
                hit = np.isfinite(ref_d)
                assert np.array_equal(np.isfinite(dist), hit)
                assert np.array_equal(idx[hit], np.nonzero(mask)[0][ref_i[hit]]) and np.array_equal(dist[hit], ref_d[hit])
            finally:
                g.cleanup()
    with pytest.raises(RuntimeError):
        partition_keys(np.array([[0.0, 0, 0], [1e9, 0, 0]]), 0.5, 2)


def _torus(n, seed, noise=0.002):
    rng = np.random.default_rng(seed)
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    R0, r0 = 1.0, 0.35
    p = np.column_stack([(R0 + r0 * np.cos(v)) * np.cos(u), (R0 + r0 * np.cos(v)) * np.sin(u), r0 * np.sin(v)])
    outward = np.column_stack([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)])
    return p + noise * rng.normal(size=p.shape), outward


def _assert_normals_equal_checker(p, nrm, cnt, radius, viewpoint, only=None):
    ref, ref_cnt, gap = NG.pca_normals(p, radius, viewpoint=viewpoint, only=only)
    rows = np.arange(len(p)) if only is None else only
    assert np.array_equal(cnt[rows], ref_cnt[rows])                       # the same ball of neighbours, exactly
    none = rows[ref_cnt[rows] < 3]
    assert not np.any(nrm[none])
    good = rows[(ref_cnt[rows] >= 3) & (gap[rows] > 1e-3)]
    assert len(good) > 0.8 * len(rows)
    n = nrm[good, :3].astype(np.float64)
    assert np.all(nrm[good, 3] == 0) and np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    # float32 rows of a float64 eigenvector: same direction AND same sense as numpy's eigh (the sense
    # may differ only where the normal is perpendicular to the line of sight / its largest components tie)
    dots = np.einsum("ij,ij->i", n, ref[good])
    assert np.all(np.abs(dots) > 1.0 - 1e-6)
    if viewpoint is not None:
        sight = np.abs(np.einsum("ij,ij->i", ref[good], np.asarray(viewpoint) - p[good]))
        sure = sight > 1e-6
    else:
        a = np.sort(np.abs(ref[good]), axis=1)
        sure = a[:, 2] - a[:, 1] > 1e-6
    assert np.all(dots[sure] > 0)
    return good


@pytest.mark.parametrize("cell,viewpoint", [(0.08, None), (0.08, (0.0, 0.0, 5.0)), (0.03, (3.0, 1.0, 0.5))])
def test_ringless_normals_equal_the_eigh_checker(cell, viewpoint):
    """No counterpart in the reference (its normals need scan rings): parity unpinned, the checker is
    numpy.linalg.eigh on the same ball of neighbours."""
    p, outward = _torus(20000, 5)
    p[:40] += 50.0 + 10.0 * np.arange(40)[:, None]                        # stray points: fewer than 3 in the ball
    g = CellGrid(p, cell=cell)
    try:
        nrm, cnt = g.estimate_normals(0.08, viewpoint=viewpoint, return_counts=True)
        good = _assert_normals_equal_checker(p, nrm, cnt, 0.08, viewpoint)
        assert not np.any(nrm[:40]) and np.all(cnt[:40] == 1)
        # and they are the surface's normals: within a few degrees of the torus' own
        on = good[good >= 40]
        assert np.median(np.abs(np.einsum("ij,ij->i", nrm[on, :3], outward[on]))) > 0.995
        with pytest.raises(RuntimeError):
            g.estimate_normals(cell * 3.5)                                 # ball wider than 3 cells
    finally:
        g.cleanup()


def test_ringless_normals_on_a_kitti_frame(frames):
    a = frames[0]
    g = CellGrid(a, cell=0.25)
    try:
        nrm, cnt = g.estimate_normals(0.5, viewpoint=(0.0, 0.0, 0.0), return_counts=True)
        only = np.arange(0, len(a), 23)
        _assert_normals_equal_checker(a, nrm, cnt, 0.5, (0.0, 0.0, 0.0), only)
    finally:
        g.cleanup()


def test_ringless_point_to_plane_registration_recovers_a_known_motion():
    """A cloud without scan rings against its rigidly moved, noisy copy: normals from the cloud itself,
    exact neighbours, the plane tiles of LeastSquaresPlane.glsl."""
    a, _ = _torus(60000, 7, noise=0.001)
    a = np.vstack([a, np.column_stack([np.random.default_rng(8).uniform(-2, 2, (20000, 2)), np.full(20000, -0.4)])])
    R = lidar.rotation_xyz(0.02, -0.01, 0.015)
    t = np.array([0.05, -0.02, 0.03])
    rng = np.random.default_rng(9)
    pick = rng.permutation(len(a))[:50000]
    src = (R @ a[pick].T).T + t + rng.normal(0, 0.001, (len(pick), 3))
    g = CellGrid(a, cell=0.06)
    icp = PointcloudIcpContainer(None)
    try:
        g.estimate_normals(0.06, viewpoint=(0.0, 0.0, 10.0))
        g.set_source(src)
        dx, conv, hb = g.icp_run(1, 0.15, max_iter=30)
        t_opt, R_opt = icp._replay(dx, len(dx) - (1 if conv else 0))
        # moved = R_opt src - t_opt must undo src = R a + t
        assert np.abs(R_opt @ R - np.eye(3)).max() < 1e-3 and np.abs(R_opt @ t - t_opt).max() < 2e-3
        moved_loop = g.read_source()
    finally:
        g.cleanup()
    # the container's driver is that loop
    moved, t2, R2 = icp.align_ringless_pt2pl(a, src, 0.15, iterations=30, viewpoint=(0.0, 0.0, 10.0),
                                             normal_radius=0.06, cell=0.06)
    assert np.array_equal(moved, moved_loop) and np.array_equal(t2, t_opt) and np.array_equal(R2, R_opt)
    assert np.abs(moved - a[pick]).max() < 0.01
