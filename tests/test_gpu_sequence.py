"""Frame-to-frame odometry over the first twelve real frames of sample_data (2011_09_26 drive): far enough to
cross the reference's 10-frame replacement of the reference cloud (EnvironmentConstructor.py:35, :250-254),
after which the registration origin is no longer zero and the [ox, oy, oz, ox] upload (ComputeShader.py:436-437)
shows in every unit vector."""
import hashlib

import numpy as np
import pytest

from oracle import np_oracle as O
from pointcloudprocessing_amd import MODE_PLANE, ComputeShader, PointcloudIcpContainer, Voxelizer, lidar, odometry
from pointcloudprocessing_amd.environment import EnvironmentConstructor
from tests._data import bits, golden, kitti_sequence_raw
from tests.test_environment_cpu import reader_from_golden

pytestmark = pytest.mark.gpu

N_FRAMES = 12
SAMPLED = (3, 11)         # a frame against the first reference cloud, and the first one after the reset


def test_packed_sequence_fixture_is_lossless():
    seq = golden("kitti_seq.npz")
    for raw, want in zip(kitti_sequence_raw()[2:], seq["raw_sha"]):
        assert hashlib.sha256(np.ascontiguousarray(raw).tobytes()).hexdigest() == str(want)


def test_driver_over_real_frames_crosses_the_reset():
    imu = golden("ref_imu.npz")
    ox = reader_from_golden(imu)
    reader = lidar.LidarDataReader()
    for k, raw in enumerate(kitti_sequence_raw()):
        reader.add_frame(f"{k:010d}", np.column_stack([raw, np.zeros(len(raw), np.float32)]))     # decoded on the device
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    env = EnvironmentConstructor(None, ox, reader, icp, cs, voxelizer=Voxelizer(cs, None))
    calls = []
    plain = env.getRefinedTransition

    def recording(**kw):
        out = plain(**kw)
        calls.append((kw["reference_points"], np.array(kw["reference_origin"], dtype=np.float64), kw["points_to_refine"],
                      kw["iterations"], kw["full_iter"], out))
        return out

    env.getRefinedTransition = recording
    try:
        env.init(0.5)
        for _ in range(N_FRAMES):
            frame, rec = env.getNextFrameData(0)
            env.calculateTransition_imu(frame, rec, point_to_plane=True, iterations=10)       # UI.py:400
        poses = env.prev_poses
        # IMU half: the reference's own run (pure_imu) over the same records
        assert np.array_equal(np.stack([p[0] for p in poses]), imu["imu_pos"][:N_FRAMES])
        assert np.array_equal(np.stack([p[2] for p in poses]), imu["rotation"][:N_FRAMES])
        # the frame that closes the block runs 15 full iterations, then its refined cloud is the reference
        assert [c[3] for c in calls] == [10] * 9 + [15, 10] and [c[4] for c in calls] == [False] * 9 + [True, False]
        assert calls[10][0] is env.reference_points and np.array_equal(calls[10][1], poses[10][1])
        assert np.abs(calls[10][1]).max() > 1.0                   # origin 3 m from the start: the quirk is live
        assert env.local_frame_counter == 1
        # refined against dead-reckoned trajectory: ICP corrects the IMU path by centimetres per frame, not metres
        refined = np.stack([p[1] for p in poses])
        drift = np.linalg.norm(refined - imu["imu_pos"][:N_FRAMES], axis=1)
        assert drift.max() < 0.5, drift
        steps = np.linalg.norm(np.diff(refined, axis=0), axis=1)
        assert np.all((steps > 0.25) & (steps < 0.45)), steps      # the car rolls about a third of a metre per sweep
        assert env.voxelizer.stored_voxels > 12082                # more than frame 0 alone holds at 0.5 m
        # two sampled registrations, every iteration of the device loop against the restatement, bit for bit
        for k in SAMPLED:
            ref_pts, origin, placed, iters, full_iter, (t_dev, R_dev) = calls[k - 1]
            icp.keep_prepared_target = False
            icp.preparePointToPlane(ref_pts, origin, placed)
            dx, conv, hb, _ = cs.icpRun(placed, MODE_PLANE, iters, tol=1e-6, full_iter=full_iter, want_hb=True)
            trace = []
            t_ref, R_ref = O.full_pt2pl(ref_pts, placed, origin, iterations=iters, full_iter=full_iter, trace=trace)
            assert len(dx) == len(trace)
            for i, (H, b, d) in enumerate(trace):
                assert np.array_equal(bits(hb[i][:6, :6]), bits(H)) and np.array_equal(bits(hb[i][:6, 6]), bits(b)), (k, i)
                assert np.array_equal(bits(dx[i]), bits(d)), (k, i)
            assert np.abs(R_dev - R_ref).max() < 1e-5 and np.abs(t_dev - t_ref).max() < 1e-4
    finally:
        env.cleanup()
        cs.cleanup()
        reader.cleanup()


def test_batched_sequence_odometry_equals_pair_by_pair():
    frames = [lidar.frame_from_raw(r) for r in kitti_sequence_raw()]
    poses = odometry.sequence_odometry(frames, point_to_plane=True, iterations=10, batch_size=4)
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    try:
        single = [icp.full_pt2pl(frames[k], frames[k + 1], np.zeros(3), iterations=10) for k in range(N_FRAMES - 1)]
    finally:
        cs.cleanup()
    from pointcloudprocessing_amd import sharding
    want = sharding.compose_odometry(single)
    for (R, t), (Rw, tw) in zip(poses, want):
        assert np.array_equal(R, Rw) and np.array_equal(t, tw)
    # the drive: a third of a metre per sweep
    length = np.linalg.norm(poses[-1][1])
    assert 3.0 < length < 4.5, length


# ------------------------------------------------------------------------------ the one-stream pipeline (pcp_seq_*)
def _raw4(raw3):
    return np.ascontiguousarray(np.column_stack([raw3, np.zeros(len(raw3), np.float32)]), dtype=np.float32)


def _oracle_replay(dx, applied):
    """(t_opt, R_opt) of a step trace by the restatement: grid <- R grid + t per applied step, then Kabsch."""
    grid = O.uniform_grid(10)
    for d in dx[:applied]:
        grid = O.apply_icp_step(grid, O.rotation(d[3], d[4], d[5]), d[:3])
    R_opt, t_opt = O.uniform_optimal_icp(grid, O.uniform_grid(10))
    return t_opt, R_opt


def test_device_pose_extraction_against_the_reference_known_answer():
    """uniform_optimal_icp of the reference itself (ref_kat: one step applied to the 10^3 grid, then its Kabsch
    extraction, PointcloudIcpContainer.py:333-347) against the device's extraction of the same step; random step
    lists against the restatement's replay; no step = identity."""
    K = golden("ref_kat.npz")
    cs = ComputeShader(maxPointsPerCloud=1000, maxScans=4)
    try:
        step = np.concatenate([K["alg_t"], K["rot_angles"][1]]).astype(np.float32)
        t_opt, R_opt = cs.poseFromSteps([step])
        # the reference's rotation() took NumPy's float32 cosines; the device rounds float64 ones (an ulp apart at most)
        assert np.abs(R_opt - K["alg_Ropt"]).max() < 2.5e-7 and np.abs(t_opt - K["alg_topt"]).max() < 2.5e-7
        t_o, R_o = _oracle_replay(step[None, :], 1)
        assert np.abs(R_opt - R_o).max() < 1e-13 and np.abs(t_opt - t_o).max() < 1e-13
        t_opt, R_opt = cs.poseFromSteps(np.zeros((0, 6), np.float32))
        assert np.abs(R_opt - np.eye(3)).max() < 1e-15 and np.abs(t_opt).max() < 1e-15
        rng = np.random.default_rng(5)
        for n in (1, 2, 7, 20, 32):
            steps = np.concatenate([rng.uniform(-0.4, 0.4, (n, 3)), rng.uniform(-0.2, 0.2, (n, 3))], axis=1).astype(np.float32)
            t_opt, R_opt = cs.poseFromSteps(steps)
            t_o, R_o = _oracle_replay(steps, n)
            assert np.abs(R_opt - R_o).max() < 1e-12 and np.abs(t_opt - t_o).max() < 1e-11, n
    finally:
        cs.cleanup()


@pytest.mark.parametrize("plane,iterations", [(True, 10), (False, 6)])
def test_pipeline_equals_the_pair_by_pair_device_path(plane, iterations):
    """Raw bytes in, steps out: every pair's dx trace, convergence flag and point counts equal those of
    DeviceFrame.decode + scan_lines + as_target + pcp_icp_launch on the same frames, bit for bit."""
    from pointcloudprocessing_amd import PinnedFrames, SequencePipeline
    from pointcloudprocessing_amd.frame import DeviceFrame, register_frames
    raws = [_raw4(r) for r in kitti_sequence_raw()]
    pinned = PinnedFrames(raws)
    pipe = SequencePipeline(max_frames=N_FRAMES, max_points=131072)
    cs = ComputeShader(maxPointsPerCloud=131072, maxScans=100)
    fa, fb = DeviceFrame(131072), DeviceFrame(131072)
    try:
        # two submissions in flight: a short run first (5 frames), then all twelve
        t_short = pipe.submit([pinned[k] for k in range(5)], point_to_plane=plane, iterations=iterations)
        t_all = pipe.submit(pinned, point_to_plane=plane, iterations=iterations)
        with pytest.raises(Exception, match="in flight"):
            pipe.submit(pinned, point_to_plane=plane, iterations=iterations)
        short, n_short = pipe.collect(t_short)
        steps, n_pts, poses = pipe.collect_with_poses(t_all)
        assert len(short) == 4 and len(steps) == N_FRAMES - 1
        for p in range(4):
            assert np.array_equal(bits(short[p][0]), bits(steps[p][0])) and short[p][1] == steps[p][1]
        # the poses come from the device (tracking grid + Kabsch behind the loops): equal to the restatement's replay of
        # the same steps to rounding, and to the NumPy replay of the reference's own functions (whose float32 cosines
        # may differ by an ulp) far inside the 1e-5 / 1e-4 the poses have to meet
        for (t_dev, R_dev), (t_np, R_np), (dx, conv) in zip(poses, pipe.poses(steps, plane), steps):
            t_o, R_o = _oracle_replay(dx, len(dx) - 1 if (conv and plane) else len(dx))
            assert np.abs(R_dev - R_o).max() < 1e-12 and np.abs(t_dev - t_o).max() < 1e-12
            assert np.abs(R_dev - R_np).max() < 1e-6 and np.abs(t_dev - t_np).max() < 1e-6
            assert abs(np.linalg.det(R_dev) - 1.0) < 1e-12 and np.abs(R_dev @ R_dev.T - np.eye(3)).max() < 1e-14
        for p in range(N_FRAMES - 1):
            assert fa.decode(raws[p]) == n_pts[p] and fb.decode(raws[p + 1]) == n_pts[p + 1]
            (t_opt, R_opt), dx, conv = register_frames(cs, fa, fb, np.zeros(3), point_to_plane=plane, iterations=iterations)
            assert conv == steps[p][1] and np.array_equal(bits(dx), bits(steps[p][0])), p
            assert np.array_equal(t_opt, poses[p][0]) and np.array_equal(R_opt, poses[p][1])
    finally:
        fa.cleanup()
        fb.cleanup()
        cs.cleanup()
        pipe.close()
        pinned.close()


def test_pipeline_about_a_non_zero_origin():
    """Scan lines and unit vectors about an origin with x != 0: the [ox, oy, oz, ox] upload (ComputeShader.py:436-437)
    is live in every dot product; the pipeline still equals the pair-by-pair path bit for bit."""
    from pointcloudprocessing_amd import PinnedFrames, SequencePipeline
    from pointcloudprocessing_amd.frame import DeviceFrame, register_frames
    raws = [_raw4(r) for r in kitti_sequence_raw()[:4]]
    origin = np.array([0.35, -0.05, 0.2])
    pipe = SequencePipeline(max_frames=4, max_points=131072)
    cs = ComputeShader(maxPointsPerCloud=131072, maxScans=100)
    fa, fb = DeviceFrame(131072), DeviceFrame(131072)
    try:
        steps, n_pts = pipe.collect(pipe.submit(raws, origin=origin, iterations=5))
        for p in range(3):
            fa.decode(raws[p]); fb.decode(raws[p + 1])
            _, dx, conv = register_frames(cs, fa, fb, origin, point_to_plane=True, iterations=5)
            assert conv == steps[p][1] and np.array_equal(bits(dx), bits(steps[p][0])), p
        # and back to the zero origin on the same pipeline (the descriptors' origin is refreshed)
        steps0, _ = pipe.collect(pipe.submit(raws, iterations=5))
        fa.decode(raws[0]); fb.decode(raws[1])
        _, dx, conv = register_frames(cs, fa, fb, np.zeros(3), point_to_plane=True, iterations=5)
        assert np.array_equal(bits(dx), bits(steps0[0][0])) and not np.array_equal(bits(dx), bits(steps[0][0]))
    finally:
        fa.cleanup()
        fb.cleanup()
        cs.cleanup()
        pipe.close()


def test_pipeline_reports_pairs_that_cannot_run():
    from pointcloudprocessing_amd import PcpError, SequencePipeline
    raws = [_raw4(r) for r in kitti_sequence_raw()[:2]]
    rng = np.random.default_rng(0)
    # a frame of 5 points that never wrap in azimuth has no ring: it cannot be a target, and as a source it
    # leaves the normal equations singular
    flat = np.column_stack([rng.uniform(5, 6, 5), rng.uniform(0.1, 0.2, 5), rng.uniform(-1, 1, 5), np.zeros(5)]).astype(np.float32)
    pipe = SequencePipeline(max_frames=4, max_points=131072)
    try:
        with pytest.raises(PcpError, match="pair 0.*no scan line"):
            pipe.collect(pipe.submit([flat, raws[0], raws[1]], iterations=3))
        with pytest.raises(PcpError, match="pair 0: singular normal equations at iteration 0"):
            pipe.collect(pipe.submit([raws[0], flat], iterations=3))
        with pytest.raises(PcpError, match="source frame is empty"):
            pipe.collect(pipe.submit([raws[0], np.zeros((0, 4), np.float32)], iterations=3))
        # and the pipeline is usable afterwards
        steps, _ = pipe.collect(pipe.submit(raws, iterations=3))
        assert len(steps[0][0]) == 3
    finally:
        pipe.close()
    few = SequencePipeline(max_frames=2, max_points=131072, max_scans=32)
    try:
        with pytest.raises(PcpError, match="64 scan lines do not fit maxScans = 32"):
            few.collect(few.submit(raws, iterations=3))
    finally:
        few.close()


def test_raw_sequence_odometry_equals_the_float64_driver():
    from pointcloudprocessing_amd.sequence import sequence_odometry_raw
    raws = [_raw4(r) for r in kitti_sequence_raw()]
    got = sequence_odometry_raw(raws, iterations=10, chunk=4)             # 11 pairs in runs of 4, 4, 3
    want = odometry.sequence_odometry([lidar.frame_from_raw(r[:, :3]) for r in raws], iterations=10, batch_size=4)
    for (R, t), (Rw, tw) in zip(got, want):
        assert np.array_equal(R, Rw) and np.array_equal(t, tw)


def test_pipeline_with_imu_placements_equals_the_host_driven_pairs():
    """Independent pairs whose initial guess comes from the IMU alone: frame k turned by its IMU rotation is the
    target, frame k + 1 turned by its own and shifted by the dead-reckoned step the source -- the arithmetic of
    calculateTransition_imu (EnvironmentConstructor.py:169, :200-209) in NumPy, registered pair by pair through the
    device loop, against the same run of raw frames submitted once with placements: steps bit for bit."""
    from pointcloudprocessing_amd import PinnedFrames, SequencePipeline
    from pointcloudprocessing_amd.sequence import imu_placements, sequence_odometry_imu
    imu = golden("ref_imu.npz")
    ox = reader_from_golden(imu)
    records = [ox.getOx(k) for k in ox.getfilenames()[:N_FRAMES]]
    place = imu_placements(records)
    # the placements are the driver's own numbers: rotation and dead-reckoned path of the reference's pure-IMU run
    assert np.array_equal(np.stack([q for q, _ in place]), imu["rotation"][:N_FRAMES])
    assert np.allclose(np.cumsum([g for _, g in place], axis=0), imu["imu_pos"][:N_FRAMES], atol=1e-12)
    raws = [_raw4(r) for r in kitti_sequence_raw()]
    frames = [lidar.frame_from_raw(r[:, :3]) for r in raws]
    pinned = PinnedFrames(raws)
    pipe = SequencePipeline(max_frames=N_FRAMES, max_points=131072)
    cs = ComputeShader(maxPointsPerCloud=131072, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    try:
        steps, n_pts, poses = pipe.collect_with_poses(pipe.submit(pinned, point_to_plane=True, iterations=10, placements=place))
        plain, _ = pipe.collect(pipe.submit(pinned, point_to_plane=True, iterations=10))
        first_steps, first_plain = [], []
        for p in range(N_FRAMES - 1):
            target = (place[p][0] @ frames[p].T).T
            source = (place[p + 1][0] @ frames[p + 1].T).T + place[p + 1][1]
            icp.preparePointToPlane(target, np.zeros(3), source)
            dx, conv, _, _ = cs.icpRun(source, MODE_PLANE, 10, tol=1e-6, full_iter=False, want_hb=False)
            assert conv == steps[p][1] and np.array_equal(bits(dx), bits(steps[p][0])), p
            t_opt, R_opt = icp.full_pt2pl(target, source, np.zeros(3), iterations=10)
            assert np.array_equal(t_opt, poses[p][0]) and np.array_equal(R_opt, poses[p][1])
            first_steps.append(np.linalg.norm(steps[p][0][0][:3]))
            first_plain.append(np.linalg.norm(plain[p][0][0][:3]))
        # the guess does its work: the first iteration has centimetres left to correct instead of decimetres
        assert np.median(first_steps) < 0.25 * np.median(first_plain), (first_steps, first_plain)
        # the chained trajectory stays within decimetres of the dead-reckoned one and moves a third of a metre per sweep
        track = sequence_odometry_imu(raws, records, point_to_plane=True, iterations=10, chunk=5)
        refined = np.stack([t[1] for t in track])
        assert np.array_equal(np.stack([t[2] for t in track]), imu["rotation"][:N_FRAMES])
        assert np.allclose(np.stack([t[0] for t in track]), imu["imu_pos"][:N_FRAMES], atol=1e-12)
        assert np.linalg.norm(refined - imu["imu_pos"][:N_FRAMES], axis=1).max() < 0.5
        hops = np.linalg.norm(np.diff(refined, axis=0), axis=1)
        assert np.all((hops > 0.25) & (hops < 0.45)), hops
        for k in range(N_FRAMES - 1):           # chunked submissions (5 pairs at a time) give the same pairs
            assert np.array_equal(track[k + 1][3], poses[k][1] @ place[k + 1][0])
    finally:
        cs.cleanup()
        pipe.close()
        pinned.close()


def _synthetic_raw(seed, rings, steps, n_exact=None):
    """A raw record (sensor axes, float32 rows x, y, z, 0) of a synthetic sweep; n_exact pads or cuts it to a given
    number of raw rows (padding repeats the last ring's points, which keeps the ring order)."""
    back = lidar.aligning_rotation().T
    cloud = lidar.synthetic_scan(seed, rings=rings, steps=steps, pose_t=np.array([0.2 * seed, 0.0, 0.0]))
    raw = (back @ cloud.T).T.astype(np.float32)
    if n_exact is not None:
        while len(raw) < n_exact:
            raw = np.concatenate([raw, raw[-min(len(raw), n_exact - len(raw)):]])
        raw = raw[:n_exact]
    return _raw4(raw)


@pytest.mark.parametrize("plane", [True, False])
def test_pipeline_on_ragged_frames_equals_the_pair_by_pair_path(plane):
    """Frames of very different sizes and ring counts in one submission -- a full-capacity frame, one that ends on a
    compaction tile boundary, one of three short rings, one a tile plus one point long: every pair equals the
    pair-by-pair device path bit for bit; identity placements change nothing."""
    from pointcloudprocessing_amd import PinnedFrames, SequencePipeline
    from pointcloudprocessing_amd.frame import DeviceFrame, register_frames
    cap = 16384
    raws = [_synthetic_raw(0, 32, 400), _synthetic_raw(1, 64, 300, n_exact=cap), _synthetic_raw(2, 3, 60),
            _synthetic_raw(3, 16, 128, n_exact=2048), _synthetic_raw(4, 64, 40), _synthetic_raw(5, 8, 300, n_exact=2049),
            _synthetic_raw(6, 40, 333)]
    pinned = PinnedFrames(raws)
    pipe = SequencePipeline(max_frames=len(raws), max_points=cap)
    cs = ComputeShader(maxPointsPerCloud=cap, maxScans=100)
    fa, fb = DeviceFrame(cap), DeviceFrame(cap)
    try:
        steps, n_pts = pipe.collect(pipe.submit(pinned, point_to_plane=plane, iterations=4))
        ident = [(np.eye(3), np.zeros(3))] * len(raws)
        same, n_same = pipe.collect(pipe.submit(pinned, point_to_plane=plane, iterations=4, placements=ident))
        assert np.array_equal(n_pts, n_same)
        for p in range(len(raws) - 1):
            assert fa.decode(raws[p]) == n_pts[p] and fb.decode(raws[p + 1]) == n_pts[p + 1]
            _, dx, conv = register_frames(cs, fa, fb, np.zeros(3), point_to_plane=plane, iterations=4)
            assert conv == steps[p][1] and np.array_equal(bits(dx), bits(steps[p][0])), p
            assert np.array_equal(bits(steps[p][0]), bits(same[p][0])), p
        assert len({int(n) for n in n_pts}) == len(raws)          # really ragged
    finally:
        fa.cleanup()
        fb.cleanup()
        cs.cleanup()
        pipe.close()
        pinned.close()


ODO_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pointcloudprocessing_amd.sequence import sequence_odometry_imu, sequence_odometry_raw
from tests._data import golden, kitti_sequence_raw
from tests.test_environment_cpu import reader_from_golden
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
raws = [np.ascontiguousarray(np.column_stack([r, np.zeros(len(r), np.float32)]), dtype=np.float32) for r in kitti_sequence_raw()[:8]]
ox = reader_from_golden(golden("ref_imu.npz"))
records = [ox.getOx(k) for k in ox.getfilenames()[:8]]
track = sequence_odometry_imu(raws, records, iterations=6, rank=rank, world=world, chunk=3)
plain = sequence_odometry_raw(raws, iterations=6, rank=rank, world=world, chunk=3)
np.savez(os.path.join(sys.argv[2], f"odo_{rank}.npz"), pos=np.stack([t[1] for t in track]), rot=np.stack([t[3] for t in track]),
         R=np.stack([R for R, _ in plain]), t=np.stack([t for _, t in plain]))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_share_a_run_of_frames(tmp_path):
    """Pairs dealt to two processes in contiguous blocks (both on this GPU, poses gathered over gloo): every rank ends
    with the trajectory one process computes alone, with and without the IMU's guesses."""
    import os
    import subprocess
    import sys
    from pointcloudprocessing_amd.sequence import sequence_odometry_imu, sequence_odometry_raw
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(ODO_WORKER)
    import socket
    with socket.socket() as sk:           # a rendezvous port nobody holds
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path)], env=dict(env, RANK=str(r))) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=200) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raws = [_raw4(r) for r in kitti_sequence_raw()[:8]]
    ox = reader_from_golden(golden("ref_imu.npz"))
    records = [ox.getOx(k) for k in ox.getfilenames()[:8]]
    track = sequence_odometry_imu(raws, records, iterations=6)
    plain = sequence_odometry_raw(raws, iterations=6)
    for r in range(2):
        got = np.load(tmp_path / f"odo_{r}.npz")
        assert np.array_equal(got["pos"], np.stack([t[1] for t in track])) and np.array_equal(got["rot"], np.stack([t[3] for t in track]))
        assert np.array_equal(got["R"], np.stack([R for R, _ in plain])) and np.array_equal(got["t"], np.stack([t for _, t in plain]))
