"""A LiDAR frame that lives on the GPU from its file bytes to the registration kernels.

DeviceFrame runs the reference's frame preparation on the device:
  decode()      LidarDataReader.eatBytes + ego filter + axis rotation (LidarDataReader.py:144-150, :59-67)
  move()        (R @ p.T).T + t, the IMU pre-alignment (EnvironmentConstructor.py:169, :209)
  scan_lines()  PointcloudIcpContainer.getScanLines (:362-392)
and hands the cloud to a ComputeShader context as registration target or source
without a host copy (as_target / as_source).
"""
import ctypes as C

import numpy as np

from . import _lib, lidar
from ._lib import check, f32p, f64p, u32p
from .compute_shader import MODE_PLANE, MODE_POINT


class DeviceFrame:
    def __init__(self, max_points=200000, device=0, max_scans=100):
        self._lib = _lib.load()
        self._f = C.c_void_p()
        check(self._lib.pcp_frame_create(int(device), int(max_points), C.byref(self._f)))
        self.device, self.max_points, self.max_scans = int(device), int(max_points), int(max_scans)
        self.n = 0

    def decode(self, raw_xyzi, min_range=2.5, rotation=None):
        """raw_xyzi: (N, 4) float32 rows as np.fromfile(path, float32).reshape(-1, 4); returns points kept."""
        raw = np.ascontiguousarray(raw_xyzi, dtype=np.float32)
        if raw.ndim != 2 or raw.shape[1] != 4:
            raise ValueError("raw_xyzi must have shape (N, 4)")
        R = np.ascontiguousarray(lidar.aligning_rotation() if rotation is None else rotation, dtype=np.float64)
        n = C.c_size_t(0)
        check(self._lib.pcp_frame_decode(self._f, f32p(raw), len(raw), float(min_range), f64p(R), C.byref(n)))
        self.n = int(n.value)
        return self.n

    def decode_file(self, path, min_range=2.5, rotation=None):
        return self.decode(np.fromfile(path, dtype=np.float32).reshape(-1, 4), min_range, rotation)

    def upload(self, np_points):
        pts = np.asarray(np_points)
        cols = pts[:, :3].T if pts.ndim == 2 else None
        if cols is not None and pts.dtype == np.float64 and len(pts) > 1 and cols.flags.c_contiguous:
            # a transposed cloud: its three coordinate rows go up as they lie
            row, base = cols.strides[0], cols.ctypes.data
            check(self._lib.pcp_frame_upload_rows(self._f, base, base + row, base + 2 * row, len(pts)))
            self.n = len(pts)
            return
        p = np.ascontiguousarray(np.asarray(np_points, dtype=np.float64)[:, :3])
        check(self._lib.pcp_frame_upload(self._f, f64p(p), len(p)))
        self.n = len(p)

    def move(self, R, t):
        R = np.ascontiguousarray(R, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        check(self._lib.pcp_frame_move(self._f, f64p(R), f64p(t)))

    def scan_lines(self, origin):
        o = np.ascontiguousarray(origin, dtype=np.float64)
        lines = np.zeros((self.max_scans * 64, 2), dtype=np.uint32)
        k = C.c_int(0)
        check(self._lib.pcp_frame_scan_lines(self._f, f64p(o), len(lines), u32p(lines), C.byref(k)))
        return [(int(b), int(e)) for b, e in lines[:k.value]]

    def read(self, intensities=False):
        xyz = np.zeros((self.n, 3), dtype=np.float64)
        inten = np.zeros(self.n, dtype=np.float32) if intensities else None
        check(self._lib.pcp_frame_read(self._f, f64p(xyz), f32p(inten)))
        return (xyz, inten) if intensities else xyz

    def points_dev(self):
        """(device address of the resident float64 cloud, n rows of 3; number of points).  The frame's work is complete
        when this returns; the address is valid until the next call that changes the frame."""
        ptr, n = C.c_void_p(), C.c_size_t(0)
        check(self._lib.pcp_frame_points_dev(self._f, C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    # -- hand-over to a registration context ------------------------------------------------
    def as_target(self, compute, scan_lines, origin, point_to_plane=True):
        """prepareLS / preparePointPlane of `compute` with this frame as points_a."""
        ptr, n = C.c_void_p(), C.c_size_t(0)
        check(self._lib.pcp_frame_padded_dev(self._f, C.byref(ptr), C.byref(n)))
        if point_to_plane:
            lines4 = compute.padScanlines(scan_lines)
        else:
            lines4 = np.ones((len(scan_lines), 4), dtype=np.uint32)
            lines4[:, :2] = np.asarray(scan_lines, dtype=np.uint32).reshape(-1, 2)
        lines4 = np.ascontiguousarray(lines4, dtype=np.uint32)
        o4 = np.resize(np.asarray(origin, dtype=np.float64), (4,)).astype(np.float32)
        check(self._lib.pcp_set_target_dev(compute._ctx, ptr, int(n.value), u32p(lines4), len(lines4), f32p(o4)))
        if point_to_plane:
            check(self._lib.pcp_normals(compute._ctx, None))
        compute._active = "plane" if point_to_plane else "ls"

    def as_source(self, compute):
        ptr, n = C.c_void_p(), C.c_size_t(0)
        check(self._lib.pcp_frame_points_dev(self._f, C.byref(ptr), C.byref(n)))
        check(self._lib.pcp_upload_source_dev(compute._ctx, ptr, int(n.value)))
        check(self._lib.pcp_synchronize(compute._ctx))      # the copy is done before this frame may change

    def cleanup(self):
        if self._f:
            check(self._lib.pcp_frame_destroy(self._f))
            self._f = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def register_frames(compute, target, source, origin, point_to_plane=True, iterations=10, tol=1e-6, full_iter=False):
    """full_pt2pl / full_pt2pt (PointcloudIcpContainer.py:71-144) between two DeviceFrames: returns
    (t_opt, R_opt), dx trace and the convergence flag; the source frame itself is not modified."""
    lib = compute._lib
    lines = target.scan_lines(origin)
    target.as_target(compute, lines, origin, point_to_plane)
    source.as_source(compute)
    mode = MODE_PLANE if point_to_plane else MODE_POINT
    check(lib.pcp_icp_launch(compute._ctx, mode, int(iterations), float(tol), 1 if full_iter else 0))
    dx = np.zeros((iterations, 6), dtype=np.float32)
    it, cv = C.c_int(0), C.c_int(0)
    check(lib.pcp_icp_fetch(compute._ctx, f32p(dx), None, C.byref(it), C.byref(cv), None))
    dx = dx[:it.value]
    conv = bool(cv.value)
    t_opt, R_opt = compute.fetchPose()
    return (t_opt, R_opt), dx, conv
