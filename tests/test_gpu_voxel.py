"""Parity of the HIP voxel kernels (through the C ABI) with the reference's own
Voxelizer output on the KITTI fixture frames, with the CPU restatement, and with
the known answers of the reference's tests/VoxelizerTest.py.  Voxel coordinates,
table positions, counts and counters are exact; point ids inside a row and rows
in the staging buffer are compared as sets (append order is not defined by the
reference either, tests/VoxelizerTest.py:325-331)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from oracle import np_voxel as V
from pointcloudprocessing_amd import PcpError, Voxelizer
from tests._data import golden, kitti_pair, small_pair

pytestmark = pytest.mark.gpu
G = golden("ref_voxel.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_voxelizer_reproduces_reference_map_on_kitti(compute):
    v = Voxelizer(compute, None)
    v.init(0.5)
    v.filter_outliers = True
    for k, frame in enumerate(kitti_pair()):
        v.addPoints(frame, None)
        assert v.stored_voxels == int(G[f"stored{k}"]) and len(v.voxel_index) == int(G[f"capacity{k}"])
    v.refreshVoxelData()
    s = v.stored_voxels
    assert np.array_equal(v.voxel_index[:s], G["voxel_index"])
    assert np.array_equal(v.voxel_data[:s, 0], G["counts"])
    # id sets: exact for every row that has not overflowed; a row with more than max_points-1
    # points keeps whichever ids were appended first, which no implementation defines
    # (the reference's own test skips them, tests/VoxelizerTest.py:327-331)
    m = V.VoxelMap(0.5)
    for frame in kitti_pair():
        m.add_points(frame)
    assert sha(np.concatenate([np.sort(m.voxel_data[r, 1:min(m.voxel_data[r, 0] + 1, 1024)]) for r in range(s)])) \
        == str(G["ids_sha"])                                  # the restatement IS the reference here
    full = v.voxel_data[:s, 0] >= v.max_points
    assert 0 < full.sum() < s
    coords = v.vectorizedGetVoxelCoord(v.added_points[:v.last_added_point, :3])
    for r in range(s):
        k = min(v.voxel_data[r, 0] + 1, v.max_points)
        ids = v.voxel_data[r, 1:k]
        if not full[r]:
            assert np.array_equal(np.sort(ids), np.sort(m.voxel_data[r, 1:k]))
        else:
            pos = np.nonzero(v.voxel_index[:s, 3] == r)[0][0]
            assert len(np.unique(ids)) == v.max_points - 1
            assert np.all(coords[ids] == v.voxel_index[pos, :3])


def test_dispatch_matches_reference_and_restatement(compute):
    """tests/VoxelizerTest.py:263-342 (test_gpuCanVoxelize) and :192-261 (test_sameUnknownPoints)."""
    a, b = kitti_pair()
    m = V.VoxelMap(0.5)
    m.add_points(a)
    ref_data = m.voxel_data.copy()
    begin = len(a)
    unknown, dbg = compute.prepareDispatchVoxelizer(
        np_points=b, voxel_index=m.voxel_index, voxel_data=m.voxel_data, voxel_size=np.array([0.5], np.float32),
        stored_voxel_num=m.stored, begin_index=begin, max_points_to_store=1024, realloc_needed=True,
        prev_stored_voxels=-1, debug=True)
    r_unknown, r_coords, r_entry = V.voxelize_dispatch(b, m.voxel_index, ref_data, 0.5, m.stored, begin, 1024)
    assert sha(dbg[:, :3].astype(np.int32)) == str(G["coords_sha1"])          # voxel coordinates, exact
    assert sha(dbg[:, 3].astype(np.int32)) == str(G["entries_sha1"])          # table positions, exact
    assert np.array_equal(dbg[:, :3].astype(np.int32), r_coords) and np.array_equal(dbg[:, 3].astype(np.int64), r_entry)
    assert unknown[0] == r_unknown[0] == int(G["unknown_count1"])
    assert len(unknown) == len(r_unknown) and len(unknown) % 16 == 0 and not unknown[1 + unknown[0]:].any()
    assert sha(np.sort(unknown[1:1 + unknown[0]])) == str(G["unknown_sorted_sha1"])
    got = np.zeros_like(ref_data)
    compute.getFullVoxelData(m.stored, 1024, got)
    assert np.array_equal(got[:m.stored, 0], ref_data[:m.stored, 0])            # counts
    for r in range(0, m.stored, 7):
        k = min(got[r, 0] + 1, 1024)
        assert np.array_equal(np.sort(got[r, 1:k]), np.sort(ref_data[r, 1:k]))


def test_row_overflow_keeps_counting(compute):
    """More points than a row holds: the count keeps growing, ids stop at max_points - 1."""
    pts = np.tile(np.array([[0.1, 0.1, 0.1]]), (3000, 1)) + np.linspace(0, 0.3, 3000)[:, None]
    v = Voxelizer(compute, None)
    v.init(1.0)
    v.filter_outliers = False
    v.addPoints(pts[:5], None)
    v.addPoints(pts, None)
    v.refreshVoxelData()
    assert v.stored_voxels == 1 and v.voxel_data[0, 0] == 3005
    assert len(np.unique(v.voxel_data[0, 1:1024])) == 1023
    v.addPoints(pts[:10], None)                                  # the voxel is now ready to stage (0.8)
    assert v.voxel_stats[0, 2] == np.float32(0.8)


def _two_voxels(max_points=1024):
    voxel_index = np.array([[0, 0, 0, 0], [0, 2, 0, 1]], dtype=np.int32)
    voxel_data = np.zeros((2, max_points), dtype=np.int32)
    voxel_data[0, :2] = [1, 1]
    voxel_data[1, :3] = [2, 2, 3]
    return voxel_index, voxel_data, np.zeros((2, 4), dtype=np.float32)


def _stage(compute, vi, vd, st, stored, **kw):
    compute.bufferVoxelIndex(vi)
    compute.fullBufferVoxelData(vd)
    return compute.prepareDispatchVoxelStager(voxel_index=None, stored_voxel_num=stored, max_points=vd.shape[1],
                                              voxel_statistics=st, debug=True, **kw)


def test_stager_known_answers(compute):
    """tests/VoxelizerTest.py:345-676"""
    vi, vd, st = _two_voxels()
    st[:, 3] = 1.0
    before = st.copy()
    st, stage, counter = _stage(compute, vi, vd, st, 2, filter_outliers=True, stage_everything=True, max_staging=512)
    assert list(counter[0][:6]) == [0, 512, 1024, 2, 1, 0] and np.array_equal(st, before)
    assert stage.shape == (512, 1024)

    vi, vd, st = _two_voxels()
    st[0, 0], st[1, 0] = 1.0, 0.91
    st, stage, counter = _stage(compute, vi, vd, st, 2, filter_outliers=False, stage_everything=False, max_staging=512)
    assert list(counter[0][:6]) == [0, 512, 1024, 2, 0, 1]
    assert st[0, 2] == np.float32(0.2) and st[1, 2] == np.float32(0.2)
    assert np.isclose(st[0, 1], 1.999, atol=0.001) and 2.9 <= st[1, 1] <= 3.0

    vi, vd, st = _two_voxels()
    vd[0, :4] = [3, 1, 77, 11]
    st[0, 0], st[1, 0] = 1.0, 0.91
    st[0, 2] = 0.8
    st, stage, counter = _stage(compute, vi, vd, st, 2, filter_outliers=False, stage_everything=False, max_staging=512)
    assert counter[0][0] == 1 and st[0, 3] == 1.0 and st[1, 3] == 0.0
    assert list(stage[0, :4]) == [3, 1, 77, 11]

    vi, vd, st = _two_voxels()
    vd[0, :4] = [3, 1, 77, 11]
    st[:, 2] = 0.8
    st, stage, counter = _stage(compute, vi, vd, st, 2, filter_outliers=False, stage_everything=False, max_staging=512)
    assert counter[0][0] == 2 and np.all(st[:, 3] == 1.0)
    assert list(stage[0, :4]) == [3, 1, 77, 11] and list(stage[1, :3]) == [2, 2, 3]   # same wave: lane order


def test_stager_many_voxels(compute):
    """tests/VoxelizerTest.py:680-794"""
    n = 300
    vi = np.zeros((2 * n, 4), dtype=np.int32)
    vd = np.zeros((2 * n, 1024), dtype=np.int32)
    st = np.zeros((2 * n, 4), dtype=np.float32)
    for i in range(n):
        vi[i, 3] = i
        vd[i, 0] = i
        vd[i, 1:1 + i] = 10 + np.arange(i)
        if i % 5 == 0:
            vd[i, 1:1024] = i * 100
            vd[i, 1023] = i
            vd[i, 0] = 11234
        vd[i, 1] = i
        st[i, 2] = 0.8
    st, stage, counter = _stage(compute, vi, vd, st, n, filter_outliers=False, stage_everything=False, max_staging=512)
    assert counter[0][0] == n and np.all(st[:n, 3] == 1.0) and np.all(st[n:, 3] == 0.0)
    seen = set()
    for i in range(n):
        src = stage[i][1]
        seen.add(int(src))
        k = min(stage[i][0] + 1, 1024)
        assert stage[i][0] == vd[src][0] and np.array_equal(stage[i][1:k], vd[src][1:k])
    assert seen == set(range(n))


def test_stager_matches_restatement_on_random_states(compute):
    rng = np.random.default_rng(9)
    for trial, (filt, everything, max_staging) in enumerate([(True, False, 64), (False, False, 1024), (True, True, 1024),
                                                             (True, True, 40), (False, True, 1024)]):
        coords = np.unique(rng.integers(-6, 6, size=(700, 3)), axis=0)
        n = len(coords)
        vi = np.column_stack([coords, rng.permutation(n)]).astype(np.int32)
        vd = np.zeros((n, 64), dtype=np.int32)
        vd[:, 0] = rng.integers(0, 90, n)
        vd[:, 1:] = rng.integers(0, 100000, (n, 63))
        st = np.zeros((n, 4), dtype=np.float32)
        st[:, 0] = rng.uniform(0.05, 0.99, n)
        prev = np.where(rng.random(n) < 0.5, vd[:, 0], rng.integers(0, 90, n))
        st[:, 1] = prev + rng.uniform(0.0, 0.98, n).astype(np.float32)
        st[:, 2] = rng.choice([0.0, 0.2, 0.8], n, p=[0.6, 0.25, 0.15])
        st[:, 3] = rng.choice([0.0, 1.0], n, p=[0.85, 0.15])
        ref = st.copy()
        r_counter, r_stage = V.voxel_stage(vi, vd, ref, n, 64, filt, everything, max_staging)
        got = st.copy()
        got, stage, counter = _stage(compute, vi, vd, got, n, filter_outliers=filt, stage_everything=everything,
                                     max_staging=max_staging)
        assert np.array_equal(counter[0][:6], r_counter[:6]), trial
        k = counter[0][0]
        if k < max_staging:                                    # no overflow: outcome is fully determined
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), trial
            key = lambda rows: sorted(tuple(r[:min(r[0] + 1, 64)]) for r in rows)
            assert key(stage[:k]) == key(r_stage[:k]), trial
        else:                                                  # which voxels won the race is not defined
            assert got[:, 3].sum() == ref[:, 3].sum()
            unstaged = got[:, 3] == ref[:, 3]
            assert np.array_equal(got[unstaged][:, :3].view(np.uint32), ref[unstaged][:, :3].view(np.uint32))


def test_stage_all_remaining_voxels_terminates(compute):
    a, b = small_pair()
    v = Voxelizer(compute, None)
    v.init(0.5)
    v.filter_outliers = True
    for f in (a, b, a):
        v.addPoints(f, None)
    v.stageAllRemaningVoxels()
    staged = v.voxel_stats[:v.stored_voxels, 3] == 1.0
    ready = v.voxel_stats[:v.stored_voxels, 2] > 0.5
    assert np.array_equal(staged, ready)                       # everything marked ready has been staged


def test_voxel_error_paths(compute):
    with pytest.raises(PcpError):
        compute.prepareDispatchVoxelStager(None, 5, 17, np.zeros((8, 4), np.float32), True)   # row width mismatch
    vi, vd, st = _two_voxels()
    compute.bufferVoxelIndex(vi)
    compute.fullBufferVoxelData(vd)
    with pytest.raises(PcpError):
        compute.prepareDispatchVoxelStager(None, 3, 1024, st, True)                          # stored > table rows
    with pytest.raises(PcpError):
        compute.prepareDispatchVoxelizer(np.zeros((0, 3)), vi, vd, np.array([0.5], np.float32), 2, 0, 1024, True, -1)


class _HostMergeCompute:
    """This package's ComputeShader minus the device-side voxel creation: the Voxelizer then merges new
    voxels on the host (the path taken with a foreign compute object)."""
    def __init__(self, compute):
        self._c = compute

    def __getattr__(self, name):
        if name in ("insertUnknownVoxels", "voxelReserve", "readVoxelIndex", "readVoxelCounts"):
            raise AttributeError(name)
        return getattr(self._c, name)


def test_device_side_voxel_creation_equals_the_host_merge(compute):
    """Three clouds: growth past the initial 4096 rows, voxels with more than 1023 points (the lowest ids
    are kept), later clouds that add to old voxels and create new ones between them."""
    rng = np.random.default_rng(8)
    clouds = [rng.uniform(-22, 22, (60000, 3)) * [1, 0.1, 1],
              np.vstack([rng.normal(0, 0.1, (3000, 3)) + [5.2, 0.2, -3.2], rng.uniform(-30, 30, (20000, 3)) * [1, 0.12, 1]]),
              rng.uniform(-35, 35, (50000, 3)) * [1, 0.15, 1]]
    maps = []
    for comp in (compute, _HostMergeCompute(compute)):
        v = Voxelizer(comp, None)
        v.init(0.5)
        v.filter_outliers = False
        for c in clouds:
            v.addPoints(c, None)
        v.refreshVoxelData()
        maps.append(v)
    dev, host = maps
    s = dev.stored_voxels
    assert s == host.stored_voxels and s > 8192 and len(dev.voxel_index) == len(host.voxel_index)
    assert np.array_equal(dev.voxel_index[:s], host.voxel_index[:s])
    assert np.array_equal(dev.voxel_data[:s, 0], host.voxel_data[:s, 0])
    assert dev.voxel_data[:s, 0].max() > 1023 and dev.voxel_data[:s, 0].sum() == sum(len(c) for c in clouds)
    for r in range(s):
        k = min(dev.voxel_data[r, 0], dev.max_points - 1)
        a, b = np.sort(dev.voxel_data[r, 1:1 + k]), np.sort(host.voxel_data[r, 1:1 + k])
        created_whole = dev.voxel_data[r, 0] < dev.max_points          # rows that overflowed later keep arbitrary ids
        if created_whole:
            assert np.array_equal(a, b)
    big = int(np.argmax(dev.voxel_data[:s, 0]))
    assert dev.getStoredPoints(big).shape == (dev.max_points - 1, 3)
    # a voxel CREATED with more points than a row holds keeps the lowest ids, in order (both paths)
    blob = np.vstack([rng.uniform(-3, 3, (500, 3)), rng.uniform(0.01, 0.49, (3000, 3)) + [4.0, 1.0, -2.0]])
    blob = blob[rng.permutation(len(blob))]
    rows = []
    for comp in (compute, _HostMergeCompute(compute)):
        v = Voxelizer(comp, None)
        v.init(0.5)
        v.filter_outliers = False
        v.addPoints(blob, None)
        if comp is compute:
            v.refreshVoxelData()      # host-merged rows are only on the host until the next dispatch uploads them
        r = int(np.argmax(v.voxel_data[:v.stored_voxels, 0]))
        assert v.voxel_data[r, 0] == 3000
        rows.append(v.voxel_data[r, 1:v.max_points].copy())
    inside = np.nonzero(np.all(np.ceil(blob.astype(np.float32) / np.float32(0.5)) == [9, 3, -3], axis=1))[0]
    assert np.array_equal(rows[0], inside[:1023]) and np.array_equal(rows[1], inside[:1023])


def test_voxel_insertion_error_paths(compute):
    lib, ctx = compute._lib, compute._ctx
    v = Voxelizer(compute, None)
    v.init(1.0)
    stored, cap = C.c_int(0), C.c_int(0)
    assert lib.pcp_voxel_insert_unknown(ctx, 1024, C.byref(stored), C.byref(cap)) != 0      # no dispatch yet
    v.addPoints(np.array([[0.5, 0.5, 0.5], [1.5, 0.5, 0.5], [0.5, 0.4, 0.3]]), None)
    assert v.stored_voxels == 2 and sorted(v.voxel_data[:2, 0].tolist()) == [1, 2]
    assert lib.pcp_voxel_insert_unknown(ctx, 1024, C.byref(stored), C.byref(cap)) != 0      # already consumed
    with pytest.raises(RuntimeError):
        v.addPoints(np.array([[3e6, 0.0, 0.0]]), None)                                        # beyond +-2^20 voxels
    # the boundary itself: |coordinate| = 2^20 - 1 is the last one three 21-bit fields can key apart
    edge = float(2 ** 20 - 1)
    v2 = Voxelizer(compute, None)
    v2.init(1.0)
    v2.addPoints(np.array([[edge, 0.5, 0.5], [-edge + 0.5, 0.5, 0.5], [edge, 1.5, 0.5]]), None)
    assert v2.stored_voxels == 3
    assert sorted(v2.voxel_index[:3, 0].tolist()) == [-(2 ** 20 - 1) + 1, 2 ** 20 - 1, 2 ** 20 - 1]
    for bad in (float(2 ** 20), -float(2 ** 20) - 0.5):
        v3 = Voxelizer(compute, None)
        v3.init(1.0)
        with pytest.raises(RuntimeError, match="beyond"):
            v3.addPoints(np.array([[bad, 0.5, 0.5], [0.5, 0.5, 0.5]]), None)


def test_transposed_clouds_are_fed_as_coordinate_rows(compute):
    """The reference hands over (R @ p.T).T (LidarDataReader.py:66-67), a transposed array: its float32 copy keeps
    the coordinate rows and the device interleaves them.  Same map, same point store as with an (n, 3)-ordered copy."""
    maps = []
    for transposed in (True, False):
        v = Voxelizer(compute, None)
        v.init(0.5)
        for frame in kitti_pair():
            rows = np.ascontiguousarray(np.asarray(frame).T)             # (3, n) in memory
            cloud = rows.T if transposed else np.ascontiguousarray(rows.T)
            assert cloud.flags.c_contiguous != transposed
            v.addPoints(cloud, None)
            assert v._pending[-1][1].flags.c_contiguous != transposed
        s = v.stored_voxels
        maps.append((v.voxel_index[:s].copy(), v.voxel_data[:s, 0].copy(),
                     [np.sort(v.voxel_data[r, 1:min(v.voxel_data[r, 0] + 1, v.max_points)]) for r in range(0, s, 97)],
                     v.added_points[:v.last_added_point, :3].copy(), v.added_colors[:v.last_added_point].copy()))
    a, b = maps
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], G["voxel_index"][:len(a[0])])
    assert all(np.array_equal(x, y) for x, y in zip(a[2], b[2]))
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.all(a[4] == 255.0)
    assert np.array_equal(a[3], np.vstack(kitti_pair()).astype(np.float32))


def test_statistics_stay_on_the_device_and_are_fetched_on_access(compute):
    """The stager's statistics are not uploaded and read back per frame when the map lives on the device: the host
    array is filled when it is looked at, equals what the per-frame transfers of the reference's protocol give, and
    a mirror whose map has been replaced on the device refuses to be read."""
    rng = np.random.default_rng(3)
    clouds = [rng.uniform(-12, 12, (30000, 3)) * [1, 0.1, 1] for _ in range(4)]
    lazy = Voxelizer(compute, None)
    lazy.init(0.5)
    lazy.filter_outliers = True
    for c in clouds:
        lazy.addPoints(c, None)
    assert compute.resident_statistics and lazy._stats_stale
    got = lazy.voxel_stats.copy()                       # fetched here
    assert not lazy._stats_stale and got.shape[1] == 4
    n = lazy.stored_voxels
    eager = Voxelizer(compute, None)
    eager.init(0.5)
    compute.resident_statistics = False                 # the reference's protocol: up and down with every dispatch
    eager.filter_outliers = True
    for c in clouds:
        eager.addPoints(c, None)
    assert eager.stored_voxels == n and not eager._stats_stale
    assert np.array_equal(got[:n].view(np.uint32), eager.voxel_stats[:n].view(np.uint32)) and np.all(got[n:, 0] == 0.5)
    assert np.any(got[:n, 1] > 1.0)                     # counts and running maxima have been written
    lazy._stats_stale = True                            # as if it had not been read before the second map was made
    with pytest.raises(RuntimeError, match="has been replaced"):
        lazy.voxel_stats
    eager.cleanup()


# ------------------------------------------------------------------------------ whole frames as one enqueue (pcp_voxel_add*)
def _map_state(v):
    s = v.stored_voxels
    v.refreshVoxelData()
    data = v.voxel_data[:s].copy()
    rows = [np.sort(data[r, 1:min(data[r, 0] + 1, v.max_points)]) for r in range(s)]
    return v.voxel_index[:s].copy(), data[:, 0].copy(), rows, v.voxel_stats[:s].copy()


def _raw_frames(n):
    from pointcloudprocessing_amd import lidar
    from tests._data import kitti_sequence_raw
    return [lidar.frame_from_raw(r) for r in kitti_sequence_raw()[:n]]


def test_fed_frames_equal_the_step_by_step_path(compute):
    """Voxelizer.addPoints as one enqueue and one wait (dispatch + stager + creation of the new voxels with every count
    left on the device) against the same frames through the step-by-step entry points (five host round trips per
    frame): table, counts, id sets, statistics bit for bit, stored counts and staged-row counts after every frame."""
    from pointcloudprocessing_amd import ComputeShader
    frames = _raw_frames(6)
    other = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    try:
        fed, steps = Voxelizer(compute, None), Voxelizer(other, None)
        steps.feed_frames = False
        for v in (fed, steps):
            v.init(0.5)
            v.filter_outliers = True
        for k, f in enumerate(frames):
            fed.addPoints(f, None)
            steps.addPoints(f, None)
            assert fed.stored_voxels == steps.stored_voxels and len(fed.voxel_index) == len(steps.voxel_index), k
        ti, tc, tr, ts = _map_state(fed)
        wi, wc, wr, ws = _map_state(steps)
        assert np.array_equal(ti, wi) and np.array_equal(tc, wc)
        assert np.array_equal(ts.view(np.uint32), ws.view(np.uint32))
        full = tc >= fed.max_points
        for r in range(len(tr)):
            if not full[r]:
                assert np.array_equal(tr[r], wr[r]), r
        # a new voxel's row holds its ids ascending, as the host statement writes them
        created_last = np.arange(fed.prev_stored_voxels, fed.stored_voxels)
        for r in created_last[:50]:
            k = min(fed.voxel_data[r, 0] + 1, fed.max_points)
            assert np.all(np.diff(fed.voxel_data[r, 1:k]) > 0)
        # the global point store the rows index into
        assert np.array_equal(fed.added_points[:fed.last_added_point, :3], steps.added_points[:steps.last_added_point, :3])
        # the final full staging works on the fed map as on the other
        fed.stageAllRemaningVoxels()
        steps.stageAllRemaningVoxels()
        assert np.array_equal(fed.voxel_stats[:fed.stored_voxels].view(np.uint32), steps.voxel_stats[:steps.stored_voxels].view(np.uint32))
    finally:
        other.cleanup()


def test_frames_fed_from_device_memory_and_without_waiting(compute):
    """The raw entry points: frames in device memory as float64 (cast on the device like handleAddingPoints), four of
    them enqueued before the first wait, the device's own voxel count carried from frame to frame (stored = -1); the
    records and the map equal those of the Voxelizer fed the same clouds from the host one at a time."""
    from pointcloudprocessing_amd import ComputeShader, _lib
    from pointcloudprocessing_amd.frame import DeviceFrame
    lib = compute._lib
    frames = _raw_frames(4)
    other = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    dev = [DeviceFrame(200000) for _ in frames]
    try:
        want = Voxelizer(other, None)
        want.init(0.5)
        want.filter_outliers = False
        recs_want = []
        for f in frames:
            before = want.stored_voxels
            want.addPoints(f, None)
            recs_want.append((before, want.stored_voxels))
        v = Voxelizer(compute, None)
        v.init(0.5)
        compute.keepPointStore(True)
        tickets, begin = [], 0
        for d, f in zip(dev, frames):
            d.upload(f)
            ptr, n = d.points_dev()
            t = C.c_int(-1)
            _lib.check(lib.pcp_voxel_add_dev(compute._ctx, C.c_void_p(ptr), 8, n, 0.5, 0 if not tickets else -1, begin, 1024, 0, 1024,
                                             None, C.byref(t)))
            tickets.append(t.value)
            begin += n
        for k, t in enumerate(tickets):
            rec = _lib.VoxelFrame()
            _lib.check(lib.pcp_voxel_add_wait(compute._ctx, t, C.byref(rec)))
            assert (rec.stored_before, rec.stored) == recs_want[k] and rec.created == rec.stored - rec.stored_before and rec.flags == 0
            assert rec.unknown >= rec.created and rec.serial == t
        s = want.stored_voxels
        table = np.zeros((s, 4), np.int32)
        compute.readVoxelIndex(s, table)
        assert np.array_equal(table, want.voxel_index[:s])
        assert np.array_equal(compute.readVoxelCounts(0, s), want.voxel_data[:s, 0])
        # the unknown points of the last frame, on request: the ids the host's np.unique grouping would take
        ids = compute.readUnknownOfLastFrame()
        coords = want.vectorizedGetVoxelCoord(want.added_points[ids][:, :3])
        new_rows = want.voxel_index[:s][want.voxel_index[:s, 3] >= recs_want[-1][0]]
        assert len(ids) == rec.unknown and len(np.unique(coords, axis=0)) == rec.created == len(new_rows)
        assert np.array_equal(np.unique(coords, axis=0), new_rows[:, :3])
        # the device point store holds what handleAddingPoints casts
        got = compute.readPointStore(0, begin)
        assert np.array_equal(got[:, :3], want.added_points[:begin, :3])
    finally:
        for d in dev:
            d.cleanup()
        other.cleanup()


def _rows_sorted(xyz):
    return xyz[np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))]


def test_the_log_holds_what_every_fed_frame_staged(compute):
    """pcp_voxel_log_*: while a log is open every fed frame appends the points of the rows it staged.  Here the frames
    are fed one at a time, so the per-frame hand-off (pcp_voxel_staged_points after each wait, Voxelizer.py:130-133)
    sees the same rows: the log taken at the end holds exactly the points those hand-offs put into a map, and as many
    rows."""
    from pointcloudprocessing_amd import CellGrid, MapSink
    frames = _raw_frames(10)
    per_frame, whole = CellGrid(np.zeros((0, 3)), cell=1.0), CellGrid(np.zeros((0, 3)), cell=1.0)
    try:
        sink = MapSink(per_frame, compute=compute)
        v = Voxelizer(compute, None, voxman=sink)
        v.init(2.0)
        v.filter_outliers = False
        with pytest.raises(PcpError, match="not been ended"):
            compute.stagedLogTake(0)
        compute.stagedLogBegin(0, 4 << 20)
        with pytest.raises(PcpError, match="is open"):
            compute.stagedLogBegin(1, 1024)
        for f in frames[:6]:
            v.addPoints(f, None)
        compute.stagedLogEnd(0)
        with pytest.raises(PcpError, match="not open"):
            compute.stagedLogEnd(0)
        with pytest.raises(PcpError, match="not been taken"):
            compute.stagedLogBegin(0, 1024)
        fed_before = sink.points_fed
        # the other log takes the next frames while the first waits to be taken
        compute.stagedLogBegin(1, 4 << 20)
        for f in frames[6:]:
            v.addPoints(f, None)
        compute.stagedLogEnd(1)
        ptr0, n0, rows0 = compute.stagedLogTake(0)
        ptr1, n1, rows1 = compute.stagedLogTake(1)
        assert n0 == fed_before > 0 and n0 + n1 == sink.points_fed and rows0 + rows1 == sink.voxels_fed
        assert n1 > 0 and ptr0 != ptr1
        whole.append_device(ptr0, n0)
        whole.append_device(ptr1, n1)
        got, want = whole.sorted_points()[1], per_frame.sorted_points()[1]
        assert len(got) == n0 + n1 and np.array_equal(_rows_sorted(got), _rows_sorted(want))
        # every logged point is a point of the store
        store = compute.readPointStore(0, v.last_added_point)[:, :3].astype(np.float64)
        assert len(np.unique(np.concatenate([_rows_sorted(got), store]), axis=0)) == len(np.unique(store, axis=0))
    finally:
        per_frame.cleanup()
        whole.cleanup()


def test_the_log_needs_the_point_store_and_room():
    from pointcloudprocessing_amd import ComputeShader
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=10)
    try:
        v = Voxelizer(cs, None)
        v.init(2.0)
        v.filter_outliers = False
        with pytest.raises(PcpError, match="point store"):
            cs.stagedLogBegin(0, 1024)
        with pytest.raises(PcpError, match="two logs"):
            cs.stagedLogBegin(2, 1024)
        # a log too small for a frame's staged points says so when it is taken; the slot is usable afterwards
        cs.keepPointStore(True)
        v.init(2.0)
        cs.stagedLogBegin(0, 16)
        for f in _raw_frames(8):
            v.addPoints(f, None)
        cs.stagedLogEnd(0)
        with pytest.raises(PcpError, match="overflowed"):
            cs.stagedLogTake(0)
        cs.stagedLogBegin(0, 16)
        cs.stagedLogEnd(0)
        assert cs.stagedLogTake(0)[1:] == (0, 0)
    finally:
        cs.cleanup()


def test_a_frame_too_wide_for_the_keys_is_refused(compute):
    v = Voxelizer(compute, None)
    v.init(0.5)
    pts = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [40000.0, 0.1, 0.1], [0.1, 39000.0, 0.3], [0.3, 0.1, 38000.0]])
    with pytest.raises(PcpError, match="span more than 2\\^16 voxels"):
        v.addPoints(pts, None)
    # the map is unchanged and usable
    v2 = Voxelizer(compute, None)
    v2.init(0.5)
    v2.addPoints(pts[:2], None)
    assert v2.stored_voxels == 1
    with pytest.raises(PcpError, match="beyond \\+-2\\^20"):
        v2.addPoints(np.array([[0.1, 0.1, 0.1], [600000.0, 0.0, 0.0]]), None)
