"""CPU suite: pins the NumPy restatement (oracle/np_oracle.py) and the host
logic of the product against the reference's own known answers
(tests/golden/ref_kat.npz, written by oracle/gen_golden.py from the importable
reference functions) and against the tolerances the reference's tests use."""
import numpy as np
import pytest

from oracle import np_oracle as O
from pointcloudprocessing_amd import PointcloudAlignment, PointcloudIcpContainer
from tests._data import bits, golden, kitti_pair, small_pair

K = golden("ref_kat.npz")
ZERO = np.zeros(3)


def test_fixture_frames_are_the_reference_frames():
    import hashlib
    for k, f in enumerate(kitti_pair()):
        assert len(f) == int(K[f"frame{k}_n"])
        assert hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(K[f"frame{k}_sha"])


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_scan_lines_equal_reference(impl):
    """PointcloudIcpContainer.getScanLines (:362-392): 64 rings, tail points in no ring."""
    get = O.get_scan_lines if impl == "oracle" else PointcloudIcpContainer(None).getScanLines
    a, b = kitti_pair()
    for pts, origin, ref in ((a, ZERO, K["scan_lines0"]), (b, ZERO, K["scan_lines1"]),
                             (a, K["scan_origin"], K["scan_lines0_o"])):
        lines = np.array(get(pts, origin), dtype=np.int64)
        assert np.array_equal(lines, ref)
    assert len(K["scan_lines0"]) == 64 and K["scan_lines0"][-1, 1] < len(a) - 1


def test_scan_lines_edge_cases():
    assert O.get_scan_lines(np.array([[5.0, 0, 0], [5.0, 0, 0.1]]), ZERO) == []     # no wrap: no ring at all
    ring = np.array([[np.cos(t), -1.0, np.sin(t)] for t in np.linspace(0.01, 2 * np.pi - 0.01, 50)]) * 10
    two = np.vstack([ring, ring * 1.1, ring[:5]])
    assert O.get_scan_lines(two, ZERO) == [(0, 49), (50, 99)]


def test_rotation_and_pose_algebra_equal_reference():
    for ang, R in zip(K["rot_angles"], K["rot_R"]):
        assert np.allclose(PointcloudAlignment.rotation(ang[0], ang[1], ang[2]), R, atol=0, rtol=0)
        assert np.allclose(PointcloudAlignment.rotation(ang), R, atol=0, rtol=0)
        # the restatement evaluates the float32 trig through float64: within an ulp of NumPy's
        assert np.abs(O.rotation(ang[0], ang[1], ang[2]) - R).max() < 2.5e-7
    icp = PointcloudIcpContainer(None)
    pts, t, R = K["alg_pts"], K["alg_t"], K["alg_R"]
    assert np.abs(O.apply_ls(pts, t, R) - K["alg_apply_ls"]).max() < 1e-12
    assert np.abs(O.apply_icp_step(pts, R, t) - K["alg_apply_step"]).max() < 1e-12
    assert np.array_equal(icp.apply_iteration_of_ls(pts, t, R), K["alg_apply_ls"])
    assert np.array_equal(icp.applyIcpStep(pts, R, t), K["alg_apply_step"])
    assert np.array_equal(O.uniform_grid(10), K["uniform_grid"])
    assert np.array_equal(icp.getUniformGrid(10), K["uniform_grid"])
    moved = icp.applyIcpStep(icp.getUniformGrid(10), K["rot_R"][1], t)
    for f in (O.uniform_optimal_icp, icp.uniform_optimal_icp):
        Ropt, topt = f(moved, K["uniform_grid"])
        assert np.abs(Ropt - K["alg_Ropt"]).max() < 1e-12 and np.abs(topt - K["alg_topt"]).max() < 1e-12


def test_solve6_equals_numpy_solve_on_float32():
    """np.linalg.solve(float32, float32) (PointcloudIcpContainer.py:27) = float64 LU rounded to float32."""
    for H, b, dx in zip(K["solve_H"], K["solve_b"], K["solve_dx"]):
        got = O.solve6(H, b)
        assert got.dtype == np.float32
        assert np.abs(got.view(np.int32).astype(np.int64) - dx.view(np.int32)).max() <= 1   # <= 1 ulp
    assert O.solve6(np.zeros((6, 6), np.float32), np.ones(6, np.float32)) is None
    d = O.mean_abs6(K["solve_dx"][0])
    assert d == np.mean(np.abs(K["solve_dx"][0]))


def test_identity_correspondence_H_b_equal_reference_numpy():
    """tests/GpuTest.py:561-582: debug-mode dispatchLS vs getHsBsNumpy sums, atol 1e-3."""
    st = O.RingIcp(100)
    st.prepare_ls(K["grid1"], [(0, 0)], ZERO, debug=True)
    H, b = st.dispatch_ls(K["grid2"])
    assert np.allclose(H, K["grid_H"], atol=1e-3) and np.allclose(b, K["grid_b"], atol=1e-3)
    icp = PointcloudIcpContainer(None)
    Hs, Bs = icp.getHsBsNumpy(K["grid1"], K["grid2"])
    Hn, bn = icp.sumHsBsNumpy(Hs, Bs)
    assert np.allclose(Hn, K["grid_H"], atol=1e-4) and np.allclose(bn, K["grid_b"], atol=1e-4)


def test_known_correspondence_icp_converges_like_reference():
    """tests/FunctionalTesting.py:250-277 scenario through the product's NumPy helpers."""
    icp = PointcloudIcpContainer(None)
    assert np.array_equal(icp.getUniformShape(), K["shape"])
    cur = K["shape_moved"].copy()
    for _ in range(5):
        t, r = icp.icp_step_LS_vector(K["shape"], cur)
        cur = icp.apply_iteration_of_ls(cur, t, PointcloudAlignment.rotation(r[0], r[1], r[2]))
    assert np.abs(cur - K["shape_after5"]).max() < 1e-9
    assert np.abs(cur - K["shape"]).max() < 1e-6


def test_hb_tree_sum_structure():
    ones = np.ones((130, 8, 8), np.float32)
    out = O.hb_tree_sum(ones)
    assert np.all(out[:6, :7] == 130) and np.all(out[6:] == 0) and np.all(out[:, 7] == 0)
    # the fixed tree, not a left-to-right sum: tiles 1 and 33 meet first (1 + 1 = 2), then join
    # tile 0; summed one after the other both ones would be lost against 2^24
    t = np.zeros((64, 8, 8), np.float32)
    t[0], t[1], t[33] = 2.0 ** 24, 1.0, 1.0
    assert O.hb_tree_sum(t)[0, 0] == np.float32(2.0 ** 24 + 2.0)
    rng = np.random.default_rng(0)
    tiles = rng.standard_normal((5000, 8, 8)).astype(np.float32)
    assert np.allclose(O.hb_tree_sum(tiles)[:6, :7], tiles[:, :6, :7].astype(np.float64).sum(0), atol=1e-3)


def test_ring_nn_is_consistent_with_exact_nn():
    """The ring search is approximate: it can only be as good as the exact neighbour
    (legacy KD-tree path, PointcloudIcpContainer.py:467-472), and it mostly is."""
    a, b = kitti_pair()
    sub = b[::int(K["kd_query_stride"])]
    lines = O.get_scan_lines(a, ZERO)
    st = O.RingIcp(100)
    st.prepare_nns(a, lines, ZERO)
    idx, d2 = st.dispatch_nns(sub)
    exact = (K["kd_dist"] ** 2)
    assert np.all(d2 >= exact * (1 - 1e-4) - 1e-6)
    assert np.mean(idx == K["kd_idx"]) > 0.9
    # hints were recorded and a second, hint-guided pass reproduces the result
    assert np.all(st.hint >= 0)
    idx2, d22 = st.dispatch_nns(sub)
    assert np.array_equal(idx, idx2) and np.array_equal(bits(d2), bits(d22))


def test_nn_pad_rings_and_origin_quirk():
    a, b = small_pair()
    lines = O.get_scan_lines(a, ZERO)
    a4, b4 = O.pad_points(a), O.pad_points(b[:300])
    pad = O.pad_scan_lines(lines).astype(np.int64)
    assert pad.shape == (100, 4) and tuple(pad[-1]) == (1, 1, 1, 1)
    hint = np.full(len(b4), -1, np.float32)
    i100, d100, h100 = O.nearest_neighbour_scan(a4, b4, O.origin4(ZERO), pad, 100, hint)
    unp = O.ls_scan_lines(lines).astype(np.int64)
    iz, dz, hz = O.nearest_neighbour_scan(a4, b4, O.origin4(ZERO), unp, len(unp), hint)
    # pad rings add point #1 as a candidate and nothing else
    diff = i100 != iz
    assert np.all(i100[diff] == 1)
    # origin is uploaded as [ox, oy, oz, ox]: w = -ox leaks into the unit vectors
    o4 = O.origin4(np.array([0.5, 0.0, 0.0]))
    assert tuple(o4) == (0.5, 0.0, 0.0, 0.5)
    u = O.unit_sphere(a4[:4], o4)
    assert np.all(u[:, 3] != 0)


def test_normals_equal_the_reference_getNormals():
    """tests/GpuTest.py:297-323, test_shouldGetSameNormalsFromGpu: shader normals (float32, rings padded to
    100 rows) against the reference's own pure-Python getNormals (:325-455; float64, unpadded rings) on the
    first 50,000 points of frame 0: more than 99.9 % must agree to |delta| < 1e-3.  ref_normals.npz holds
    what GpuTest.getNormals returned for that call (oracle/gen_golden.py: ref_normals)."""
    ref = golden("ref_normals.npz")
    ref32 = np.insert(ref["normals64"].astype(np.float32), 3, 0, axis=1)      # as tests/GpuTest.py:308-309 casts it
    a, _ = kitti_pair()
    n = int(ref["n_points"])
    assert n == 50000 and str(ref["frame_sha"]) == str(K["frame0_sha"])
    lines = O.get_scan_lines(a, ZERO)
    n32 = O.normals(O.pad_points(a), O.origin4(ZERO), O.pad_scan_lines(lines).astype(np.int64), 100)[:n]
    # the reference's assertion, as it states it (float32 difference, 4 components)
    ok = np.linalg.norm(n32 - ref32, axis=1) < 1e-3
    assert ok.mean() > 0.999
    # and tighter than the reference asks: a normal built from the SAME two neighbours differs only by
    # float32 rounding, one built from other neighbours by O(1); so this counts the points on which the
    # restated ring search (circular binary search + window + above/below choice) picked the reference's
    # neighbours.  The rest (12 points) sit on float32-vs-float64 ties of the cosine comparison.
    same = np.linalg.norm(n32[:, :3].astype(np.float64) - ref["normals64"], axis=1) < 1e-5
    assert same.mean() > 0.9997
    # unit length, w = 0, pointing towards the sensor
    good = ~np.isnan(n32).any(axis=1)
    assert np.allclose(np.linalg.norm(n32[good, :3], axis=1), 1.0, atol=1e-5) and np.all(n32[:, 3] == 0)
    assert np.mean(np.einsum("ij,ij->i", n32[good, :3], -a[:n][good]) >= 0) > 0.999


def test_oracle_icp_recovers_a_known_motion():
    a, _ = small_pair()
    from pointcloudprocessing_amd import lidar
    R = lidar.rotation_xyz(0.004, -0.006, 0.003)
    t = np.array([0.05, -0.02, 0.03])
    b = (R @ a.T).T + t
    for full in (O.full_pt2pt, O.full_pt2pl):
        t_opt, R_opt = full(a, b, ZERO, iterations=15)
        assert np.abs(R_opt @ R - np.eye(3)).max() < 2e-3
        assert np.abs(R_opt @ t - t_opt).max() < 2e-2


# ------------------------------------------------------------------ legacy exact-neighbour ICP
def test_kd_restatement_equals_reference_step_and_alignment():
    """np_grid.icp_step_kd / align_kd against the reference's icp_step_KD / align_KD outputs
    (tests/golden/ref_kd.npz, written by oracle/gen_golden.py from the reference itself)."""
    from scipy.spatial import KDTree
    from oracle import np_grid as NG
    from tests._data import golden, kitti_pair
    g = golden("ref_kd.npz")
    a, b = kitti_pair()
    tree = KDTree(a)
    dd, ii = tree.query(b[::7])
    assert np.array_equal(ii, g["kd7_idx"]) and np.array_equal(dd, g["kd7_dist"])
    R, t = NG.icp_step_kd(tree, a, b)
    assert np.abs(R - g["step_R"]).max() < 1e-9 and np.abs(t - g["step_t"]).max() < 1e-8
    d_low, i_low = NG.exact_nn_lowest(a, b[::7])
    assert np.array_equal(d_low, dd) and np.all(i_low <= ii)
    same = i_low == ii
    assert same.mean() > 0.99 and np.array_equal(np.linalg.norm(b[::7][~same] - a[i_low[~same]], axis=1), dd[~same])


def test_kd_alignment_restatement_equals_reference():
    from oracle import np_grid as NG
    from tests._data import golden, kitti_pair
    g = golden("ref_kd.npz")
    a, b = kitti_pair()
    moved, R_opt, t_opt = NG.align_kd(a[::4], b[::4], iterations=3)      # the restated loop runs
    assert moved.shape == b[::4].shape and np.abs(R_opt @ R_opt.T - np.eye(3)).max() < 1e-12
    moved, R_opt, t_opt = NG.align_kd(a, b)
    assert np.abs(R_opt - g["align_R"]).max() < 1e-7 and np.abs(t_opt - g["align_t"]).max() < 1e-6
    assert np.abs(moved[::64] - g["align_moved_sub"]).max() < 1e-5


# ------------------------------------------------------------------------------ the reference's scalar KATs
def test_unrolled_H_and_b_equal_the_matrix_products():
    """tests/GpuTest.py:521-544 (test_H_unrollEqualsNumpyBehaviour, test_B_unrollEqualsNumpyBehaviour), same
    point, same error vector, same 1e-3 bar."""
    icp = PointcloudIcpContainer(None)
    point = np.array([12.23, -12.82, 0.001])
    error = np.array([1234.123, 12, -0.001])
    j = icp.getJ_i(point)
    assert np.allclose(j.T @ j, icp.getH_i(j), atol=1e-3)
    assert np.array_equal(icp.getH_i_vector(j), j.T @ j)
    assert np.allclose(j.T @ error, icp.getB(j, error), atol=1e-3)


def test_uniform_grid_loop_equals_numpy():
    """tests/GpuTest.py:546-559 (test_uniformGridNumpyEqualLoop): the per-point loop and the vectorised
    getHsBsNumpy on a rotated, shifted 10 x 10 x 10 grid, 1e-3; and the running sums against np.sum."""
    import random
    random.seed(3)
    icp = PointcloudIcpContainer(None)
    grid1 = icp.getUniformGrid(10)
    R = PointcloudAlignment.randomRotation(0.4)
    t = PointcloudAlignment.randomTranslation1() * 10
    grid2 = (R @ icp.getUniformGrid(10).T).T + t
    Hs_loop, Bs_loop = icp.getHsBsLoop(grid1, grid2)
    Hs_numpy, Bs_numpy = icp.getHsBsNumpy(grid1, grid2)
    assert np.allclose(Hs_loop, Hs_numpy, atol=1e-3) and np.allclose(Bs_loop, Bs_numpy, atol=1e-3)
    H, b = icp.sumHsBsLoop(Hs_loop[:100], Bs_loop[:100])
    assert np.allclose(H, Hs_loop[:100].sum(axis=0), atol=1e-9) and np.allclose(b, Bs_loop[:100].sum(axis=0), atol=1e-9)
