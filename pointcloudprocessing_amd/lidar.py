"""Input clouds.

* frame_from_raw / load_velodyne_bin: the three numeric steps of the reference's KITTI reader that
  shape a frame before registration (LidarDataReader.py:144-150 decode, :59-61 ego filter,
  :66-67 + :139-142 axis-aligning rotation) in NumPy, for test and benchmark inputs; the product's
  reader (class LidarDataReader below) runs them on the device.
* A deterministic synthetic 64-ring spinning-LiDAR scan of a box scene with the
  size and ring structure of a KITTI sweep (about 120k points), for machines
  that hold no KITTI data.
"""
import numpy as np


def rotation_xyz(x, y, z):
    """Rx @ Ry @ Rz (LidarDataReader.py:176-202)."""
    cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
    R_z = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    R_y = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    R_x = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return R_x @ R_y @ R_z


def aligning_rotation():
    """LidarDataReader.py:139-142"""
    return rotation_xyz(-np.pi / 2, 0, np.pi / 2)


def frame_from_raw(xyz32):
    """Raw velodyne xyz (float32) -> the float64 cloud the reference registers:
    drop points within 2.5 m of the sensor, then rotate (LidarDataReader.py:59-67)."""
    p = np.asarray(xyz32, dtype=np.float32)
    p = p[np.linalg.norm(p, axis=1) > 2.5]
    return (aligning_rotation() @ p.T).T


def load_velodyne_bin(path):
    """LidarDataReader.eatBytes (:144-150) + frame_from_raw."""
    raw = np.fromfile(path, dtype=np.float32).reshape(-1, 4)
    return frame_from_raw(raw[:, :3])


def pack_raw_frames(frames):
    """Lossless compact container for raw velodyne xyz (float32, millimetre-quantised as KITTI stores
    them): per frame k the first differences of the integer millimetres, coordinate by coordinate
    (`d<k>`, int32 (3, n)), and the flat positions of negative zeros (`z<k>`).  Dict for np.savez."""
    out = {"frames": np.int64(len(frames))}
    for k, xyz in enumerate(frames):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        mm = np.rint(xyz.astype(np.float64) * 1000.0).astype(np.int64)
        out[f"d{k}"] = np.diff(mm, axis=0, prepend=0).T.astype(np.int32)
        out[f"z{k}"] = np.flatnonzero(np.signbit(xyz) & (xyz == 0)).astype(np.int64)
    return out


def unpack_raw_frames(npz):
    """Inverse of pack_raw_frames: list of (n, 3) float32 arrays, bit for bit the packed ones."""
    frames = []
    for k in range(int(npz["frames"])):
        mm = np.cumsum(npz[f"d{k}"].astype(np.int64), axis=1).T
        xyz = np.ascontiguousarray((mm.astype(np.float64) / 1000.0).astype(np.float32))
        xyz.reshape(-1)[npz[f"z{k}"]] = -0.0
        frames.append(xyz)
    return frames


# ----------------------------------------------------------------------------- synthetic scene
_BOXES = np.array([   # axis-aligned boxes (xmin, ymin, zmin, xmax, ymax, zmax), y is up
    [6.0, -1.73, -2.0, 10.5, 0.0, 0.2],
    [-14.0, -1.73, 5.0, -9.0, 1.2, 7.5],
    [12.0, -1.73, 9.0, 30.0, 6.0, 25.0],
    [-35.0, -1.73, -30.0, -20.0, 9.0, -12.0],
    [3.0, -1.73, -18.0, 8.0, 0.3, -14.5],
    [-6.0, -1.73, 14.0, -2.0, 2.5, 16.0],
    [20.0, -1.73, -22.0, 24.0, 1.6, -9.0],
    [-3.0, -1.73, -45.0, 18.0, 12.0, -38.0],
    [-50.0, -1.73, 20.0, -30.0, 7.0, 45.0],
    [35.0, -1.73, -5.0, 38.0, 3.0, 4.0],
])
_GROUND_Y = -1.73
_MAX_RANGE = 80.0
_SENSOR_OFF = np.array([0.0, 0.254 / 2, 0.0])   # the rings' apex, FunctionalBase.glsl:35-37


def _cast(origin, dirs):
    """Range of the first hit of each ray with the ground plane or a box (inf = none)."""
    n = len(dirs)
    best = np.full(n, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (_GROUND_Y - origin[1]) / dirs[:, 1]
        best = np.where((t > 0), t, best)
        for b in _BOXES:
            lo = (b[:3] - origin) / dirs
            hi = (b[3:] - origin) / dirs
            tmin = np.nanmax(np.minimum(lo, hi), axis=1)
            tmax = np.nanmin(np.maximum(lo, hi), axis=1)
            hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0)
            best = np.where(hit & (tmin < best), tmin, best)
    return best


def synthetic_scan(seed=0, pose_R=None, pose_t=None, rings=65, steps=2300, noise=0.008):
    """One sweep of a 64-ring sensor standing at pose (R, t) in the scene, as an
    (N, 3) float64 cloud in the SENSOR frame, ring after ring (top ring first),
    each ring one monotonic azimuth turn -- the order a KITTI .bin has after
    frame_from_raw.  About 120k points survive with the defaults."""
    rng = np.random.default_rng(seed)
    R = np.eye(3) if pose_R is None else np.asarray(pose_R, dtype=np.float64)
    t = np.zeros(3) if pose_t is None else np.asarray(pose_t, dtype=np.float64)
    elev = np.deg2rad(np.linspace(2.0, -24.8, rings))
    out = []
    for r in range(rings):
        # ragged rings: every ring drops a different random subset of its firings
        az = np.linspace(0.0, 2 * np.pi, steps, endpoint=False) + rng.uniform(0, 2 * np.pi / steps)
        keep = rng.random(steps) > 0.04 + 0.003 * r
        az = az[keep]
        d = np.stack([np.cos(elev[r]) * np.cos(az), np.full(len(az), np.sin(elev[r])),
                      np.cos(elev[r]) * np.sin(az)], axis=1)
        rng_m = _cast(t + R @ _SENSOR_OFF, d @ R.T)
        ok = np.isfinite(rng_m) & (rng_m < _MAX_RANGE)
        rng_m = rng_m[ok] + rng.normal(0.0, noise, ok.sum())
        pts = _SENSOR_OFF + d[ok] * rng_m[:, None]
        out.append(pts[np.linalg.norm(pts, axis=1) > 2.5])
    return np.concatenate(out, axis=0)


def synthetic_pair(seed=0, yaw=0.012, forward=0.85):
    """Two consecutive sweeps of a sensor that moved `forward` metres along x and
    turned `yaw` radians about the up axis between them."""
    a = synthetic_scan(seed)
    R = rotation_xyz(0.0, yaw, 0.0)
    b = synthetic_scan(seed + 1, pose_R=R, pose_t=np.array([forward, 0.0, 0.03]))
    return a, b


class LidarDataReader:
    """The velodyne half of the reference's reader (LidarDataReader.py:9-260): frames keyed by file
    stem, decoded, ego-filtered and rotated -- on the device (frame.DeviceFrame), read back once because
    the reader's interface hands out host arrays.  Camera colouring is not reproduced: colours are the
    intensities, which is what createColors (:96-101) starts from.  Files are read in sorted order."""

    VELODYNE_DATA = "velodyne_points/data"
    DEFAULT_COLOR = np.array([255, 178, 102])

    def __init__(self, device=0):
        self.datamap = {}
        self.root = None
        self.count = 0
        self.inited = False
        self.device = device
        self._frame = None

    def init(self, path, calibration=None, targetCamera="02", max_read=10, ui=None):
        import os
        self.root = path
        folder = os.path.join(path, self.VELODYNE_DATA)
        for name in sorted(n for n in os.listdir(folder) if n.endswith(".bin")):
            raw = np.fromfile(os.path.join(folder, name), dtype=np.float32).reshape(-1, 4)
            self.add_frame(name[:-4], raw)
            if ui is not None:
                ui.setFrameCounter(self.count, max_read)
            if self.count >= max_read:
                break
        self.inited = True

    def add_frame(self, key, raw_xyzi):
        from .frame import DeviceFrame
        raw = np.ascontiguousarray(raw_xyzi, dtype=np.float32)
        if self._frame is None or self._frame.max_points < len(raw):
            if self._frame is not None:
                self._frame.cleanup()
            self._frame = DeviceFrame(max_points=max(len(raw), 200000), device=self.device)
        self._frame.decode(raw, min_range=2.5, rotation=aligning_rotation())
        points, intensities = self._frame.read(intensities=True)
        int_cls = np.tile(intensities.T.reshape(-1, 1), 3)
        self.datamap[key] = (points, int_cls.copy(), int_cls)
        self.count += 1

    def cleanup(self):
        self.inited = False
        self.datamap.clear()
        self.count = 0
        if self._frame is not None:
            self._frame.cleanup()
            self._frame = None

    def getPoints(self, filename):
        d = self.datamap[filename]
        return np.ascontiguousarray(d[0]), np.ascontiguousarray(d[1])

    def getPointsWithIntensities(self, filename):
        d = self.datamap[filename]
        return np.ascontiguousarray(d[0]), np.ascontiguousarray(d[1]), d[2]

    def getfilenames(self):
        return list(self.datamap.keys())

    @staticmethod
    def aligningRotation():
        return aligning_rotation()

    @staticmethod
    def rotation(x, y, z):
        return rotation_xyz(x, y, z)
