"""The driver loop of a drive kept on the device (pcp_chain_*) against the same drive run frame by frame through
EnvironmentConstructor.calculateTransition_imu on the host (EnvironmentConstructor.py:154-265) with the device loop
underneath: all thirty real sweeps of sample_data with their OXTS records, i.e. two replacements of the reference
cloud, 15-iteration closing frames, non-zero origins."""
import numpy as np
import pytest

from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, Voxelizer, lidar
from pointcloudprocessing_amd.chain import OdometryChain, chain_odometry, imu_rows
from pointcloudprocessing_amd.environment import EnvironmentConstructor
from pointcloudprocessing_amd.sequence import PinnedFrames
from tests._data import bits, golden, kitti_all_raw
from tests.test_environment_cpu import reader_from_golden

pytestmark = pytest.mark.gpu


def _raw4(raw3):
    return np.ascontiguousarray(np.column_stack([raw3, np.zeros(len(raw3), np.float32)]), dtype=np.float32)


def _host_driven(n_frames, voxel=0.5, sink=None):
    """The frame-by-frame driver -> (prev_poses, [(placed, iterations, full_iter, (t_opt, R_opt), steps)], clouds fed to
    the voxel map, voxelizer, compute shader)."""
    ox = reader_from_golden(golden("ref_imu.npz"))
    reader = lidar.LidarDataReader()
    for k, raw in enumerate(kitti_all_raw()[:n_frames]):
        reader.add_frame(f"{k:010d}", _raw4(raw))
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    vox = Voxelizer(cs, None, voxman=sink(cs) if sink else None)
    env = EnvironmentConstructor(None, ox, reader, icp, cs, voxelizer=vox)
    calls, fed, steps = [], [], []
    plain_refine, plain_add, plain_run = env.getRefinedTransition, vox.addPoints, cs.icpRun

    def refine(**kw):
        out = plain_refine(**kw)
        calls.append((kw["points_to_refine"], kw["iterations"], kw["full_iter"], out, steps[-1]))
        return out

    def add(points, colors, printt=False):
        fed.append(np.array(points))
        return plain_add(points, colors, printt)

    def run(*a, **kw):
        out = plain_run(*a, **kw)
        steps.append((out[0].copy(), out[1]))
        return out

    env.getRefinedTransition, vox.addPoints, cs.icpRun = refine, add, run
    env.init(voxel)
    for _ in range(n_frames):
        frame, rec = env.getNextFrameData(0)
        env.calculateTransition_imu(frame, rec, point_to_plane=True, iterations=10)       # UI.py:400
    poses = list(env.prev_poses)
    reader.cleanup()
    return poses, calls, fed, vox, cs, env


def test_chain_equals_the_frame_by_frame_driver():
    n_frames = 30
    poses, calls, fed, vox_h, cs_h, env = _host_driven(n_frames)
    ox = reader_from_golden(golden("ref_imu.npz"))
    records = [ox.getOx(k) for k in ox.getfilenames()[:n_frames]]
    rows, rots, stepv, _ = imu_rows(records)
    raws = [_raw4(r) for r in kitti_all_raw()[:n_frames]]
    pinned = PinnedFrames(raws)
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    cs.keepPointStore(True)
    vox = Voxelizer(cs, None)
    vox.separate_colors, vox.filter_outliers = False, False
    vox.init(0.5)
    chain = OdometryChain(max_frames=16, max_points=131072)
    try:
        # two runs in flight: 16 + 14 frames, the second run continues the drive of the first
        t0 = chain.submit([pinned[k] for k in range(16)], rows[:16], voxelizer=vox)
        t1 = chain.submit([pinned[k] for k in range(16, n_frames)], rows[16:], voxelizer=vox)
        with pytest.raises(Exception, match="in flight"):
            chain.submit([pinned[0]], rows[:1])
        first = chain.collect(t0)
        kept0 = chain.last_points.copy()
        clouds = {k: chain.read_refined(t0, k, kept0[k]) for k in (0, 3, 10, 15)}
        got = first + chain.collect(t1)
        kept1 = chain.last_points.copy()
        clouds[16] = chain.read_refined(t1, 0, kept1[0])
        clouds[29] = chain.read_refined(t1, 13, kept1[13])
        assert len(got) == n_frames
        # frame 0: the first reference, nothing registered
        assert np.array_equal(got[0][1], np.zeros(3)) and np.array_equal(got[0][3], np.eye(3)) and len(got[0][4]) == 0
        closing = 0
        for k in range(1, n_frames):
            guess, pos, t_opt, R_opt, dx, conv = got[k]
            placed, iters, full_iter, (t_h, R_h), (dx_h, conv_h) = calls[k - 1]
            # the loop itself: every step bit for bit, hence the same pose
            assert conv == conv_h and np.array_equal(bits(dx), bits(dx_h)), k
            assert np.array_equal(t_opt, t_h) and np.array_equal(R_opt, R_h), k
            assert len(dx) <= iters and (not full_iter or len(dx) == 15)
            closing += int(full_iter)
            # the chain's state: positions as the driver's NumPy expressions give them
            assert np.array_equal(pos, poses[k][1]), (k, pos - poses[k][1])
            assert np.array_equal(guess, poses[k - 1][1] + stepv[k]), k
            assert np.array_equal(R_opt @ rots[k], poses[k][3])
        assert closing == 2                                    # frames 10 and 20 closed their blocks
        for k, cloud in clouds.items():
            assert np.array_equal(cloud, fed[k]), k            # what went into the voxel map
        # the voxel map: same table, counts, id sets as the host-driven feed
        assert vox.stored_voxels == vox_h.stored_voxels and vox.last_added_point == vox_h.last_added_point
        n = vox.stored_voxels
        assert np.array_equal(vox.voxel_index[:n], vox_h.voxel_index[:n])
        got_rows, want_rows = vox.voxel_data[:n], vox_h.voxel_data[:n]
        assert np.array_equal(got_rows[:, 0], want_rows[:, 0])
        whole = want_rows[:, 0] < vox.max_points           # appends race: id order is free, and so is which ids an overfull row keeps
        assert np.array_equal(np.sort(got_rows[whole, 1:], axis=1), np.sort(want_rows[whole, 1:], axis=1))
        assert np.array_equal(vox.voxel_stats[:n], vox_h.voxel_stats[:n])
        assert np.array_equal(vox.added_points[:vox.last_added_point], vox_h.added_points[:vox.last_added_point])
        # and the whole drive through the convenience driver, in runs of 7 frames
        cs2 = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
        try:
            track = chain_odometry(raws, records, run=7)
        finally:
            cs2.cleanup()
        for k in range(n_frames):
            assert np.array_equal(track[k][0], poses[k][0]) and np.array_equal(track[k][1], poses[k][1])
            assert np.array_equal(track[k][2], poses[k][2]) and np.array_equal(track[k][3], poses[k][3])
    finally:
        chain.close()
        pinned.close()
        env.cleanup()
        cs.cleanup()
        cs_h.cleanup()


def test_chain_hands_every_staged_voxel_to_the_map():
    """Voxelizer.addPoints hands the points of the voxels each frame stages to the voxel manager (Voxelizer.py:130-133).
    The frame-by-frame driver does so after every frame; the chain keeps runs of frames in flight and delivers what all
    of them staged when a run is collected.  Both maps receive the same voxels with the same number of points each (which
    ids a row keeps once more points than it holds have arrived is decided by the race of the appends, in the reference as
    here: VoxelShader.glsl:141-145), and every point the chain's map receives is a point of its store."""
    from pointcloudprocessing_amd import CellGrid, MapSink
    n_frames = 30
    grids = [CellGrid(np.zeros((0, 3)), cell=1.0) for _ in range(2)]
    sinks = []

    def host_sink(cs):
        sinks.append(MapSink(grids[0], compute=cs))
        return sinks[0]

    poses, calls, fed, vox_h, cs_h, env = _host_driven(n_frames, voxel=2.0, sink=host_sink)
    ox = reader_from_golden(golden("ref_imu.npz"))
    rows, _, _, _ = imu_rows([ox.getOx(k) for k in ox.getfilenames()[:n_frames]])
    pinned = PinnedFrames([_raw4(r) for r in kitti_all_raw()[:n_frames]])
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    sink = MapSink(grids[1], compute=cs)
    vox = Voxelizer(cs, None, voxman=sink)
    vox.separate_colors, vox.filter_outliers = vox_h.separate_colors, vox_h.filter_outliers
    vox.init(2.0)
    chain = OdometryChain(max_frames=7, max_points=131072)
    try:
        tickets, got = [], []
        for lo in range(0, n_frames, 7):                       # runs of 7, 7, 7, 7, 2 frames, two in flight
            hi = min(n_frames, lo + 7)
            tickets.append(chain.submit([pinned[k] for k in range(lo, hi)], rows[lo:hi], voxelizer=vox))
            if len(tickets) == 2:
                got += chain.collect(tickets.pop(0))
        got += chain.collect(tickets.pop(0))
        for k in range(n_frames):
            assert np.array_equal(got[k][1], poses[k][1]), k
        assert vox.stored_voxels == vox_h.stored_voxels
        assert sinks[0].points_fed > 100000                    # the drive does stage voxels
        assert (sink.points_fed, sink.voxels_fed) == (sinks[0].points_fed, sinks[0].voxels_fed)
        a, b = grids[1].sorted_points()[1], grids[0].sorted_points()[1]
        assert len(a) == sink.points_fed == len(b)
        per_voxel = lambda x: np.unique(vox.vectorizedGetVoxelCoord(x), axis=0, return_counts=True)
        (va, ca), (vb, cb) = per_voxel(a), per_voxel(b)
        assert len(va) == sink.voxels_fed and np.array_equal(va, vb) and np.array_equal(ca, cb)
        store = vox.added_points[:vox.last_added_point, :3].astype(np.float64)
        assert len(np.unique(np.concatenate([a, store]), axis=0)) == len(np.unique(store, axis=0))
    finally:
        chain.close()
        pinned.close()
        env.cleanup()
        cs.cleanup()
        cs_h.cleanup()
        for g in grids:
            g.cleanup()


def test_chain_reports_a_reference_it_cannot_prepare():
    from pointcloudprocessing_amd import PcpError
    raws = [_raw4(r) for r in kitti_all_raw()[:3]]
    rows = np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (3, 1))
    few = OdometryChain(max_frames=3, max_points=131072, max_scans=32)
    try:
        with pytest.raises(PcpError, match="64 scan lines do not fit maxScans = 32"):
            few.collect(few.submit(raws, rows))
        few.reset()
        with pytest.raises(PcpError, match="keeps 0 points"):
            few.submit([np.zeros((5, 4), np.float32)], rows[:1])
    finally:
        few.close()
    point = OdometryChain(max_frames=3, max_points=131072, reset_every=2)
    try:
        # point-to-point mode, reference replaced every second frame; a reset starts the same drive again
        a = point.collect(point.submit(raws, rows, point_to_plane=False, iterations=4, closing_iterations=5))
        assert [len(f[4]) for f in a] == [0, 4, 5]
        point.reset()
        b = point.collect(point.submit(raws, rows, point_to_plane=False, iterations=4, closing_iterations=5))
        for fa, fb in zip(a, b):
            assert np.array_equal(fa[1], fb[1]) and np.array_equal(bits(fa[4]), bits(fb[4]))
    finally:
        point.close()


def test_a_submission_that_fails_half_way_resets_the_drive():
    """The voxel feed refuses the first frame (a voxel count the map does not hold): the frames already enqueued are
    drained, the drive starts over with the next submission, and that one gives what a fresh chain gives."""
    from pointcloudprocessing_amd import PcpError
    raws = [_raw4(r) for r in kitti_all_raw()[:3]]
    rows = np.tile(np.concatenate([np.eye(3).reshape(9), np.zeros(3)]), (3, 1))
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    vox = Voxelizer(cs, None)
    vox.separate_colors, vox.filter_outliers = False, False
    vox.init(0.5)

    class WrongCount:                       # a voxelizer that claims five voxels the resident table does not have
        def _feed_begin(self):
            ctx, size, stored, begin, width, filt = vox._feed_begin()
            return ctx, size, 5, begin, width, filt

    chain, fresh = OdometryChain(max_frames=3), OdometryChain(max_frames=3)
    try:
        good = chain.collect(chain.submit(raws[:1], rows[:1]))                    # frame 0 of a drive
        with pytest.raises(PcpError, match="resident table has 0 rows.*the drive has been reset"):
            chain.submit(raws[1:], rows[1:], voxelizer=WrongCount())
        again = chain.collect(chain.submit(raws, rows))                          # frame 0 of a new drive
        want = fresh.collect(fresh.submit(raws, rows))
        assert len(good) == 1 and len(again) == 3
        for fa, fb in zip(again, want):
            assert np.array_equal(fa[1], fb[1]) and np.array_equal(bits(fa[4]), bits(fb[4]))
    finally:
        chain.close()
        fresh.close()
        vox.cleanup()
        cs.cleanup()
