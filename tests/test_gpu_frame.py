"""Device-side frame preparation (decode, ego filter, rotation, pre-alignment, ring segmentation)
against the reference's outputs in tests/golden/ref_kat.npz, and the odometry driver end to end."""
import datetime
import hashlib

import numpy as np
import pytest

from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, Voxelizer, lidar
from pointcloudprocessing_amd.environment import EnvironmentConstructor
from pointcloudprocessing_amd.frame import DeviceFrame, register_frames
from pointcloudprocessing_amd.oxts import OxtsDataReader
from tests._data import bits, golden, kitti_pair

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def raw4(raw3, seed=0):
    rng = np.random.default_rng(seed)
    return np.column_stack([raw3, rng.uniform(0, 1, len(raw3)).astype(np.float32)]).astype(np.float32)


@pytest.fixture(scope="module")
def frame():
    f = DeviceFrame(max_points=200000)
    yield f
    f.cleanup()


def test_decode_reproduces_the_reference_frames_bit_for_bit(frame):
    k, pair = golden("ref_kat.npz"), golden("kitti_pair.npz")
    for i in (0, 1):
        raw = raw4(pair[f"raw{i}"], i)
        n = frame.decode(raw)
        xyz, inten = frame.read(intensities=True)
        assert n == int(k[f"frame{i}_n"]) and sha(xyz) == str(k[f"frame{i}_sha"])
        keep = np.linalg.norm(raw[:, :3], axis=1) > 2.5
        assert np.array_equal(inten, raw[keep, 3])
        assert frame.scan_lines(np.zeros(3)) == [tuple(r) for r in k[f"scan_lines{i}"].tolist()]
    frame.decode(raw4(pair["raw0"]))
    o = k["scan_origin"]
    assert frame.scan_lines(o) == [tuple(r) for r in k["scan_lines0_o"].tolist()]


def test_decode_edge_cases(frame):
    assert frame.decode(np.zeros((0, 4), np.float32)) == 0 and frame.scan_lines(np.zeros(3)) == []
    near = np.array([[1.0, 1.0, 1.0, 0.5], [2.5, 0.0, 0.0, 0.1], [0.0, 2.5000002, 0.0, 0.2], [3.0, 4.0, 0.0, 0.3]], np.float32)
    assert frame.decode(near) == 2                                     # |p| > 2.5 is strict, in float32
    assert np.array_equal(frame.read(), lidar.frame_from_raw(near[:, :3]))
    rng = np.random.default_rng(1)
    big = rng.normal(0, 20, (150001, 4)).astype(np.float32)           # not a multiple of the tile size
    assert frame.decode(big, min_range=10.0, rotation=np.eye(3)) == int((np.linalg.norm(big[:, :3], axis=1) > 10).sum())
    assert np.array_equal(frame.read(), big[np.linalg.norm(big[:, :3], axis=1) > 10, :3].astype(np.float64))
    with pytest.raises(RuntimeError):
        frame.decode(np.zeros((200001, 4), np.float32))


def test_lidar_reader_decodes_on_the_device_and_keeps_the_reference_interface(tmp_path):
    pair, k = golden("kitti_pair.npz"), golden("ref_kat.npz")
    folder = tmp_path / "velodyne_points" / "data"
    folder.mkdir(parents=True)
    for i in (1, 0):                                                  # written out of order, read sorted
        raw4(pair[f"raw{i}"], i).tofile(folder / f"{i:010d}.bin")
    rd = lidar.LidarDataReader()
    rd.init(str(tmp_path), max_read=10)
    try:
        assert rd.getfilenames() == ["0000000000", "0000000001"] and rd.count == 2
        for i, key in enumerate(rd.getfilenames()):
            pts, colors, inten = rd.getPointsWithIntensities(key)
            assert sha(pts) == str(k[f"frame{i}_sha"]) and pts.flags["C_CONTIGUOUS"]
            assert inten.shape == (len(pts), 3) and np.array_equal(inten[:, 0], inten[:, 2])
            assert np.array_equal(colors, inten) and np.array_equal(rd.getPoints(key)[0], pts)
    finally:
        rd.cleanup()


def test_move_and_scan_lines_of_an_uploaded_cloud(frame):
    a, _ = kitti_pair()
    R = lidar.rotation_xyz(0.01, 0.3, -0.02)
    t = np.array([1.5, -0.2, 3.0])
    frame.upload(a)
    frame.move(R, t)
    moved = (R @ a.T).T + t
    assert np.abs(frame.read() - moved).max() < 1e-12
    icp = PointcloudIcpContainer(None)
    assert frame.scan_lines(t) == [(int(b), int(e)) for b, e in icp.getScanLines(frame.read(), t)]
    small = lidar.synthetic_scan(2, rings=9, steps=100)
    frame.upload(small)
    assert frame.scan_lines(np.zeros(3)) == [(int(b), int(e)) for b, e in icp.getScanLines(small, np.zeros(3))]


@pytest.mark.parametrize("plane", [True, False])
def test_registration_of_device_frames_equals_the_host_path(plane):
    pair = golden("kitti_pair.npz")
    a, b = kitti_pair()
    cs = ComputeShader(maxPointsPerCloud=130000, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    fa, fb = DeviceFrame(130000), DeviceFrame(130000)
    try:
        fa.decode(raw4(pair["raw0"]))
        fb.decode(raw4(pair["raw1"]))
        (t_opt, R_opt), dx, conv = register_frames(cs, fa, fb, np.zeros(3), point_to_plane=plane,
                                                   iterations=10 if plane else 6)
        if plane:
            icp.preparePointToPlane(a, np.zeros(3), b)
        else:
            icp.preparePointToPoint(a, np.zeros(3), b)
        dx_h, conv_h, _, _ = cs.icpRun(b, 1 if plane else 0, 10 if plane else 6)
        assert conv == conv_h and np.array_equal(bits(dx), bits(dx_h))
        t_h, R_h = (icp.full_pt2pl if plane else icp.full_pt2pt)(a, b, np.zeros(3), iterations=10 if plane else 6)
        assert np.array_equal(t_opt, t_h) and np.array_equal(R_opt, R_h)
    finally:
        fa.cleanup()
        fb.cleanup()
        cs.cleanup()


def test_odometry_driver_follows_a_synthetic_drive():
    """EnvironmentConstructor end to end: 5 sweeps of a vehicle moving 0.5 m per frame, OXTS velocities
    off by 10 %, refinement by point-to-plane ICP, voxel map fed on the way."""
    n_frames, step = 5, np.array([0.12, 0.0, -0.5])
    truth = [k * step for k in range(n_frames)]
    reader = lidar.LidarDataReader()
    ox = OxtsDataReader()
    t0 = datetime.datetime(2011, 9, 26, 13, 4, 32)
    for k in range(n_frames):
        scan = lidar.synthetic_scan(10 + k, pose_t=truth[k], rings=25, steps=420)
        reader.datamap[f"{k:010d}"] = (scan, np.zeros_like(scan), np.zeros_like(scan))
        vals = np.zeros(30)
        v = -step / 0.1 * 1.1                                         # position moves by -velocity * dt
        vals[9], vals[10], vals[8] = v[0], v[1], v[2]                 # velLeft, velUp, velForward
        ox.add_record(f"{k:010d}", vals, t0 + datetime.timedelta(seconds=0.1 * k))
    ox.finish()
    cs = ComputeShader(maxPointsPerCloud=20000, maxScans=100)
    icp = PointcloudIcpContainer(cs)
    env = EnvironmentConstructor(None, ox, reader, icp, cs, voxelizer=Voxelizer(cs, None))
    try:
        env.init(0.5)
        for k in range(n_frames):
            lidar_points, rec = env.getNextFrameData(0)
            env.calculateTransition_imu(lidar_points, rec, point_to_plane=True, iterations=12)
        imu = np.stack([p[0] for p in env.prev_poses])
        refined = np.stack([p[1] for p in env.prev_poses])
        err_imu = np.linalg.norm(imu - np.stack(truth), axis=1)
        err_ref = np.linalg.norm(refined - np.stack(truth), axis=1)
        assert err_imu[-1] > 0.15                                     # the IMU alone drifts by 10 % of 2 m
        assert err_ref.max() < 0.05 and err_ref[-1] < 0.3 * err_imu[-1]
        assert env.voxelizer.stored_voxels > 500
        assert env.voxelizer.last_added_point == sum(len(reader.datamap[k][0]) for k in reader.getfilenames())
    finally:
        env.cleanup()
        cs.cleanup()


def test_transposed_clouds_go_up_as_coordinate_rows():
    """(R @ p.T).T is a transposed array: DeviceFrame.upload sends its rows as they lie (pcp_frame_upload_rows);
    the resident cloud, its scan lines and its float32 copy are those of an (n, 3)-ordered upload."""
    from pointcloudprocessing_amd.frame import DeviceFrame
    from tests._data import kitti_pair
    a = kitti_pair()[0]
    rows = np.ascontiguousarray(a.T)
    f1, f2 = DeviceFrame(max_points=len(a)), DeviceFrame(max_points=len(a))
    try:
        f1.upload(rows.T)
        f2.upload(np.ascontiguousarray(a))
        assert np.array_equal(f1.read(), a) and np.array_equal(f2.read(), a)
        assert f1.scan_lines(np.zeros(3)) == f2.scan_lines(np.zeros(3))
        f1.move(np.eye(3), np.array([0.5, 0.0, -0.25]))
        assert np.array_equal(f1.read(), a + np.array([0.5, 0.0, -0.25]))
    finally:
        f1.cleanup()
        f2.cleanup()
