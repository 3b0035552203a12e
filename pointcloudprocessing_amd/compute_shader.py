"""Drop-in for the reference's GL compute runtime (ComputeShader.py, class
ComputeShader) over libpcp.so's HIP kernels.

Same method names, argument orders, return shapes and staging-array
attributes as the reference; file:line citations point at
/root/reference/ComputeShader.py.  No GL context, no shader compilation:
the kernels are precompiled gfx950 code objects inside libpcp.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PcpError, check, f32p, f64p, i32p, u32p

MODE_POINT = 0
MODE_PLANE = 1


class ComputeShader:
    """ComputeShader.py:11-98.  One instance owns one device context."""

    def __init__(self, maxPointsPerCloud=200000, maxScans=100, device=0):
        self._lib = _lib.load()
        self.bigint = 9999999
        self.float_size = 4
        self.max_points_per_voxel = 4096
        self.maxPointsPerCloud = maxPointsPerCloud
        self.maxScans = maxScans
        self.device = device
        self.main_shadercode_regex = "void main(){}"

        self.origin = None
        self.debug_info = None
        self.lens_data = None
        self.hs_out = None
        self.corr_out = None
        self.normals_out_a = None
        self.prev_out_a = 0
        self.prev_out_b = 0
        self.voxel_stage_out = None

        self._ctx = C.c_void_p()
        check(self._lib.pcp_create(int(device), int(maxPointsPerCloud), int(maxScans), C.byref(self._ctx)))
        self._active = None          # which prepare* ran last
        self._frame = None           # device frame of prepareTargetOnDevice, made on first use
        self._n_source = None        # points of the float64 source IcpBatch.upload_sources sent last
        self._voxel_resident = None  # voxel count of a map whose device copy is the authority, else None
        self._feed_clean = False     # the last voxel call was a fed frame: the device's own voxel count is current
        self._feed_keep = None
        self.resident_statistics = False  # the stager's statistics stay on the device between dispatches
        self.reallocate_out(np.array([maxPointsPerCloud, maxPointsPerCloud, 0, 0]))

    # ------------------------------------------------------------------ helpers
    def reallocate_out(self, lens_data):
        """:231-241 -- backend-owned staging arrays, reused by the next call."""
        len_a, len_b = int(lens_data[0]), int(lens_data[1])
        if len_b != self.prev_out_b:
            self.corr_out = np.zeros((len_b, 4), dtype=np.float32)
            self.prev_out_b = len_b
        if len_a != self.prev_out_a:
            self.normals_out_a = np.zeros((len_a, 4), dtype=np.float32)
            self.prev_out_a = len_a

    def pad_points(self, points):
        """:956-961"""
        points = np.asarray(points)
        if points.shape[1] == 4:
            return np.ascontiguousarray(points.astype(np.float32))
        padded = np.zeros((points.shape[0], 4), dtype=np.float32)
        padded[:, :3] = points.astype(np.float32)
        return padded

    def padScanlines(self, scan_lines):
        """:531-539 -- pad rows are (1, 1, 1, 1)."""
        scan_lines = np.array(scan_lines).astype(np.uint32)
        if scan_lines.shape != (self.maxScans, 4):
            if scan_lines.shape[0] > self.maxScans:
                # the reference overflows its SSBO here and re-raises the GL error (:549-555)
                raise PcpError(f"{scan_lines.shape[0]} scan lines do not fit maxScans = {self.maxScans}")
            padded = np.ones((self.maxScans, 4), dtype=np.uint32)
            padded[:scan_lines.shape[0], :scan_lines.shape[1]] = scan_lines
            return padded
        return scan_lines

    def saveUniformInfo(self, points_a, points_b, origin, scan_lines, debug_mode):
        """:435-445 -- origin becomes [ox, oy, oz, ox] (np.resize repeats)."""
        self.origin = np.resize(origin, (4,)).astype(np.float32)
        self.debug_info = np.array([0, 0, 0, 0]).astype(np.float32)
        if debug_mode:
            self.debug_info[0] = 1.123
        self.lens_data = np.array([len(points_a), len(points_b), len(scan_lines), 0]).astype(np.uint32)

    def _set_target(self, points_a, scan_lines4, origin):
        self.target_serial = getattr(self, "target_serial", 0) + 1      # every upload of a target counts
        a4 = self.pad_points(points_a)
        lines = np.ascontiguousarray(scan_lines4, dtype=np.uint32)
        o4 = np.resize(origin, (4,)).astype(np.float32)
        check(self._lib.pcp_set_target(self._ctx, f32p(a4), len(a4), u32p(lines), len(lines), f32p(o4)))
        # pcp_set_target only enqueues the copy of the points: the padded array has to outlive it, i.e. stay referenced
        # until the next upload replaces it (every call that returns results waits for the stream in between)
        self._target_keep = a4

    # ------------------------------------------------------------------ point to point
    def prepareLS(self, points_a, scan_lines, points_b, origin, debug_mode=False):
        """:491-515 -- scan lines are NOT padded to maxScans on this path."""
        scan_lines = np.array(scan_lines).astype(np.uint32)
        padded = np.ones((scan_lines.shape[0], 4), dtype=np.uint32)
        padded[:, :2] = scan_lines
        self._set_target(points_a, padded, origin)
        self.saveUniformInfo(points_a, points_b, origin, padded, debug_mode)
        self._active = "ls"

    @staticmethod
    def _source_layout(points):
        """(layout code of include/pcp.h, address) when the library can take the array as it lies -- float64 (n, 3),
        float32 (n, 3) or (n, 4), C-contiguous -- so that no padded float32 copy (pad_points, :956-961) is made on the
        host first; None otherwise."""
        a = points
        if (isinstance(a, np.ndarray) and a.ndim == 2 and a.dtype == np.float64 and a.shape[1] == 3 and len(a) > 1
                and a.T.flags.c_contiguous and not a.flags.c_contiguous):
            return ComputeShader.PTS_F64_ROWS, a.T.ctypes.data      # a transposed cloud: its three coordinate rows as they lie
        if not (isinstance(a, np.ndarray) and a.ndim == 2 and a.flags.c_contiguous and len(a)):
            return None
        if a.dtype == np.float64 and a.shape[1] == 3:
            return ComputeShader.PTS_F64_XYZ, a.ctypes.data
        if a.dtype == np.float32 and a.shape[1] == 3:
            return ComputeShader.PTS_F32_XYZ, a.ctypes.data
        if a.dtype == np.float32 and a.shape[1] == 4:
            return ComputeShader.PTS_F32_XYZW, a.ctypes.data
        return None

    def _dispatch_ls(self, points_b, mode, debug):
        hb = np.zeros((8, 8), dtype=np.float32)
        lay = self._source_layout(points_b)
        if lay is not None:
            check(self._lib.pcp_ls_staged(self._ctx, C.c_void_p(lay[1]), lay[0], len(points_b), mode, debug, f32p(hb)))
        else:
            b4 = self.pad_points(points_b)
            if mode == MODE_POINT:
                check(self._lib.pcp_ls_point(self._ctx, f32p(b4), len(b4), debug, f32p(hb)))
            else:
                check(self._lib.pcp_ls_plane(self._ctx, f32p(b4), len(b4), f32p(hb)))
        return hb[:6, :6], hb[:, 6][:6]

    def dispatchLS(self, points_b):
        """:517-529 -> (H (6,6) float32, b (6,) float32)."""
        debug = int(self.debug_info is not None and self.debug_info[0] >= 1.0)
        return self._dispatch_ls(points_b, MODE_POINT, debug)

    # ------------------------------------------------------------------ normals / point to plane
    def prepareDispatchNormals(self, points_a, scan_lines, origin, read=True):
        """:541-567 -- result in self.normals_out_a (N_a, 4)."""
        scan_lines = self.padScanlines(scan_lines)
        self._set_target(points_a, scan_lines, origin)
        self.lens_data = np.array([len(points_a), len(points_a), len(scan_lines), 0])
        if read:
            self.reallocate_out(self.lens_data)
            check(self._lib.pcp_normals(self._ctx, f32p(self.normals_out_a)))
        else:
            check(self._lib.pcp_normals(self._ctx, None))

    def preparePointPlane(self, points_a, scan_lines, points_b, origin, debug_mode=False):
        """:571-585"""
        self.prepareDispatchNormals(points_a, scan_lines, origin, read=False)
        scan_lines = self.padScanlines(scan_lines)
        self.saveUniformInfo(points_a, points_b, origin, scan_lines, debug_mode)
        self._active = "plane"

    def prepareTargetOnDevice(self, points_a, points_b, origin, point_to_plane):
        """prepareLS (:491-515) / preparePointPlane (:571-585) for a float64 cloud with the host passes of the
        reference done on the device: the ring segmentation of PointcloudIcpContainer.getScanLines (:362-392,
        pcp_frame_scan_lines), pad_points and the float32 cast (:956-961, :301-316); the cloud goes up once as
        float64.  Same target, scan lines and normals as the host path; returns the scan lines."""
        from .frame import DeviceFrame
        if self._frame is None:
            self._frame = DeviceFrame(max_points=self.maxPointsPerCloud, device=self.device, max_scans=self.maxScans)
        self._frame.upload(points_a)
        lines = self._frame.scan_lines(origin)
        self.target_serial = getattr(self, "target_serial", 0) + 1
        self._frame.as_target(self, lines, origin, point_to_plane)          # sets _active
        if point_to_plane:
            lines4 = self.padScanlines(lines)
        else:
            lines4 = np.ones((len(lines), 4), dtype=np.uint32)
        self.saveUniformInfo(points_a, points_b, origin, lines4, False)
        return lines

    def resetHints(self, n_points_b=None):
        """What every prepare* does to the ring hints (corr.w = -1, ComputeShader.py:488-489,
        :502-503) without touching the target: for callers that register several clouds against
        one prepared target."""
        check(self._lib.pcp_reset_hints(self._ctx))
        if n_points_b is not None:
            self.lens_data[1] = n_points_b

    def dispatchPointPlane(self, points_b):
        """:587-608 -> (H, b)."""
        return self._dispatch_ls(points_b, MODE_PLANE, 0)

    # ------------------------------------------------------------------ stand-alone NN
    def prepareNNS(self, points_a, scan_lines, points_b, origin):
        """:731-744"""
        scan_lines = self.padScanlines(scan_lines)
        self._set_target(points_a, scan_lines, origin)
        self.saveUniformInfo(points_a, points_b, origin, scan_lines, False)
        self._active = "nns"

    def dispatchNNS(self, points_b):
        """:747-759 -> (idx uint32[N_b], dist2 float32[N_b]); corr_out holds
        (idx as float, dist2, 0, ring) like the reference's read-back."""
        lay = self._source_layout(points_b)
        if lay is not None:
            # the cloud goes up as it lies (cast and padded through the context's page-locked buffer), the results come
            # down into page-locked arrays of the context; corr_out is put together from them when somebody reads it
            n = len(points_b)
            pi, pd, ph = C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(self._lib.pcp_nn_staged(self._ctx, C.c_void_p(lay[1]), lay[0], n, C.byref(pi), C.byref(pd), C.byref(ph)))
            views = (np.ctypeslib.as_array((C.c_uint32 * n).from_address(pi.value)),
                     np.ctypeslib.as_array((C.c_float * n).from_address(pd.value)),
                     np.ctypeslib.as_array((C.c_float * n).from_address(ph.value)))
            idx, dist = views[0].copy(), views[1].copy()          # the caller's own arrays, as the reference's astype gives
            self._corr_parts = (idx, dist, views[2])              # the hints stay where they are until corr_out is looked at
            self.lens_data = np.array([self.lens_data[0], n, self.lens_data[2], 0]).astype(np.uint32)
            return idx, dist
        b4 = self.pad_points(points_b)
        n = len(b4)
        idx = np.empty(n, dtype=np.uint32)
        dist = np.empty(n, dtype=np.float32)
        hint = np.empty(n, dtype=np.float32)
        check(self._lib.pcp_nn(self._ctx, f32p(b4), n, u32p(idx), f32p(dist), f32p(hint)))
        self.lens_data = np.array([self.lens_data[0], n, self.lens_data[2], 0]).astype(np.uint32)
        self._corr_parts = (idx, dist, hint)
        # the reference returns corr_out[:, 0].astype(uint32), corr_out[:, 1].astype(float32): indices below 2^24
        # survive the float32 column unchanged, so these are the same arrays without two strided passes
        return idx, dist

    @property
    def corr_out(self):
        """(N_b, 4) float32: index as float, squared distance, 0, ring -- the reference's read-back of its correspondence
        buffer (:756-759), a backend-owned array reused by the next call; assembled when it is looked at."""
        parts = getattr(self, "_corr_parts", None)
        if parts is not None:
            idx, dist, hint = parts
            self._corr_parts = None
            n = len(idx)
            if self._corr_out is None or len(self._corr_out) != n:
                self._corr_out = np.zeros((n, 4), dtype=np.float32)
                self.prev_out_b = n
            self._corr_out[:, 0] = idx
            self._corr_out[:, 1] = dist
            self._corr_out[:, 2] = 0
            self._corr_out[:, 3] = hint
        return self._corr_out

    @corr_out.setter
    def corr_out(self, value):
        self._corr_out = value
        self._corr_parts = None

    # ------------------------------------------------------------------ hierarchical sum
    def hbSum(self, tiles):
        """prepareDispatchConcurrentHbSum (:630-727) on caller-supplied tiles
        (n, 8, 8) float32 -> summed (8, 8) tile."""
        tiles = np.ascontiguousarray(tiles, dtype=np.float32).reshape(-1, 64)
        out = np.zeros((8, 8), dtype=np.float32)
        check(self._lib.pcp_hb_sum(self._ctx, f32p(tiles), len(tiles), f32p(out)))
        return out

    def lsTiles(self, points_b, mode=MODE_POINT, debug=False):
        """The per-point tiles Hs (N_b, 8, 8) of the LS kernels (what
        getBufferSubdata(ssbo_Hs) reads in the reference, :399-411)."""
        b4 = self.pad_points(points_b)
        self.hs_out = np.zeros((len(b4), 8, 8), dtype=np.float32)
        check(self._lib.pcp_ls_tiles(self._ctx, f32p(b4), len(b4), int(mode), int(debug), f32p(self.hs_out)))
        return self.hs_out

    # ------------------------------------------------------------------ device-resident loop
    def icpRun(self, points_b, mode, iterations, tol=1e-6, full_iter=False, want_points=False, want_hb=False):
        """One call for the whole iteration loop of full_pt2pt / full_pt2pl
        (PointcloudIcpContainer.py:84-94, :118-133) on the prepared target.
        Returns (dx_trace (iters, 6) f32, converged, hb_trace | None, moved points | None)."""
        pts = np.asarray(points_b)
        n = len(pts)
        dx = np.zeros((max(iterations, 1), 6), dtype=np.float32)
        hb = np.zeros((max(iterations, 1), 8, 8), dtype=np.float32) if want_hb else None
        out = np.empty((n, 3), dtype=np.float64) if want_points else None
        iters, conv = C.c_int(0), C.c_int(0)
        cols = pts[:, :3].T if pts.ndim == 2 else None
        if cols is not None and pts.dtype == np.float64 and n > 1 and cols.flags.c_contiguous:
            # a transposed result such as (R @ p.T).T (LidarDataReader.py:66-67): its three coordinate
            # rows go up as they lie, the (n, 3) order is made on the device
            row = cols.strides[0]
            base = cols.ctypes.data
            check(self._lib.pcp_upload_source_soa(self._ctx, base, base + row, base + 2 * row, n))
            check(self._lib.pcp_reset_hints(self._ctx))
            check(self._lib.pcp_icp_launch(self._ctx, int(mode), int(iterations), float(tol), int(full_iter)))
            check(self._lib.pcp_icp_fetch(self._ctx, f32p(dx), f32p(hb), C.byref(iters), C.byref(conv), f64p(out)))
        else:
            b = np.ascontiguousarray(pts[:, :3], dtype=np.float64)
            check(self._lib.pcp_icp_run(self._ctx, f64p(b), n, int(mode), int(iterations), float(tol), int(full_iter),
                                        f32p(dx), f32p(hb), C.byref(iters), C.byref(conv), f64p(out)))
        self._n_source = n
        k = iters.value
        return dx[:k], bool(conv.value), (hb[:k] if want_hb else None), out

    def fetchPose(self):
        """(t_opt (3,), R_opt (3, 3)) float64 of the loop icpRun ran last, as full_pt2pt / full_pt2pl return them
        (PointcloudIcpContainer.py:140, :333-347): tracking grid and Kabsch extraction on the device."""
        t_opt, R_opt = np.zeros(3), np.zeros((3, 3))
        check(self._lib.pcp_icp_fetch_pose(self._ctx, f64p(t_opt), f64p(R_opt)))
        return t_opt, R_opt

    def poseFromSteps(self, dx_steps):
        """(t_opt, R_opt) of a list of steps (rows of 6: t, angles) as PointcloudIcpContainer's loops extract it
        (:84-100, :333-347), on the device."""
        dx = np.ascontiguousarray(dx_steps, dtype=np.float32).reshape(-1, 6)
        t_opt, R_opt = np.zeros(3), np.zeros((3, 3))
        check(self._lib.pcp_pose_from_steps(self._ctx, f32p(dx) if len(dx) else None, len(dx), f64p(t_opt), f64p(R_opt)))
        return t_opt, R_opt

    # ------------------------------------------------------------------ voxel map
    def bufferVoxelIndex(self, voxel_index):
        """:837-840"""
        self._hand_over_voxel_map(None)
        self._feed_clean = False
        vi = np.ascontiguousarray(voxel_index, dtype=np.int32)
        check(self._lib.pcp_voxel_index_upload(self._ctx, i32p(vi), len(vi)))
        self._voxel_resident = None           # the host copy is the authority again
        self.resident_statistics = False

    def fullBufferVoxelData(self, voxel_data):
        """:842-845"""
        self._feed_clean = False
        vd = np.ascontiguousarray(voxel_data, dtype=np.int32)
        check(self._lib.pcp_voxel_data_upload(self._ctx, i32p(vd), vd.shape[0], vd.shape[1]))
        self._voxel_resident = None

    # -- voxel map kept on the device (no counterpart in the reference, whose host recreates and
    #    re-uploads table and rows every frame)
    def _hand_over_voxel_map(self, new_owner=None):
        """The device's voxel map is about to be replaced: the Voxelizer whose mirrors of it are fetched on demand
        fetches what it has not looked at yet while it still can."""
        import weakref
        ref = getattr(self, "_voxel_owner", None)
        old = ref() if ref is not None else None
        if old is not None and old is not new_owner and hasattr(old, "_detach_from_device"):
            old._detach_from_device()
        self._voxel_owner = weakref.ref(new_owner) if new_owner is not None else None

    def voxelReserve(self, rows, max_points, owner=None):
        """Fresh, empty device table / data of `rows` rows; the device copy becomes the authority."""
        self._hand_over_voxel_map(owner)
        self._feed_clean = False
        check(self._lib.pcp_voxel_reset(self._ctx))
        check(self._lib.pcp_voxel_reserve(self._ctx, int(rows), int(max_points)))
        self._voxel_resident = 0
        self.voxel_map_epoch = getattr(self, "voxel_map_epoch", 0) + 1      # a new map: mirrors of the old one are void
        self.resident_statistics = False        # a Voxelizer that reads its statistics lazily switches this on

    # -- a whole frame of Voxelizer.addPoints as one enqueue and one wait (pcp_voxel_add*)
    PTS_F32_XYZW, PTS_F32_XYZ, PTS_F32_ROWS, PTS_F64_XYZ, PTS_F64_ROWS = 4, 3, -3, 8, -8

    def voxelAddFrame(self, points, voxel_size, stored_voxel_num, begin_index, max_points, filter_outliers,
                      max_staging=1024, device_ptr=None, n_points=None, layout=None, after_stream=None):
        """Enqueues dispatch + stager + creation of the new voxels for one cloud on the device-resident map and returns
        a ticket for voxelAddWait.  points: a host array -- float32 (n, 4), (n, 3) or a transposed (3, n).T view,
        float64 (n, 3) -- that must stay alive until the wait; or device_ptr / n_points / layout for a cloud already
        in device memory (after_stream: the hipStream_t that produces it).  stored_voxel_num = -1: the device's own
        count (allowed after a frame fed before, with no step-by-step voxel call in between)."""
        serial = C.c_int(-1)
        size = float(np.asarray(voxel_size).reshape(-1)[0])
        stored = -1 if (stored_voxel_num is None or self._feed_clean) else int(stored_voxel_num)
        if device_ptr is not None:
            check(self._lib.pcp_voxel_add_dev(self._ctx, C.c_void_p(int(device_ptr)), int(layout), int(n_points), size, stored,
                                              int(begin_index), int(max_points), int(bool(filter_outliers)), int(max_staging),
                                              None if after_stream is None else C.c_void_p(int(after_stream)), C.byref(serial)))
        else:
            a = np.asarray(points)
            n = len(a)
            if a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 4 and a.flags.c_contiguous:
                lay, ptr = self.PTS_F32_XYZW, a.ctypes.data
            elif a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 3 and a.flags.c_contiguous:
                lay, ptr = self.PTS_F32_XYZ, a.ctypes.data
            elif a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 3 and n > 1 and a.T.flags.c_contiguous:
                lay, ptr = self.PTS_F32_ROWS, a.T.ctypes.data
            elif a.dtype == np.float64 and a.ndim == 2 and a.shape[1] == 3 and a.flags.c_contiguous:
                lay, ptr = self.PTS_F64_XYZ, a.ctypes.data
            else:
                raise PcpError("voxelAddFrame takes float32 (n, 4) / (n, 3) / transposed rows or float64 (n, 3), C-contiguous")
            self._feed_keep = a
            check(self._lib.pcp_voxel_add(self._ctx, C.c_void_p(ptr), lay, n, size, stored, int(begin_index), int(max_points),
                                          int(bool(filter_outliers)), int(max_staging), C.byref(serial)))
        self._feed_clean = True          # later frames may rely on the device's own count
        return serial.value

    def voxelAddWait(self, serial):
        """Waits for a fed frame -> its record (serial, stored_before, stored, unknown, created, staged, key_bits)."""
        rec = _lib.VoxelFrame()
        try:
            check(self._lib.pcp_voxel_add_wait(self._ctx, int(serial), C.byref(rec)))
        finally:
            self._feed_keep = None
        self._voxel_resident = rec.stored
        return rec

    def readStagedRows(self, rows, max_points):
        """The rows the last stager dispatch released (what prepareDispatchVoxelStager returns), fetched on request."""
        if rows <= 0:
            return []
        if self.voxel_stage_out is None or self.voxel_stage_out.shape[1] != max_points or len(self.voxel_stage_out) < rows:
            self.voxel_stage_out = np.zeros((max(rows, 1024), max_points), dtype=np.int32)
        check(self._lib.pcp_voxel_staged_read(self._ctx, int(rows), int(max_points), i32p(self.voxel_stage_out)))
        return self.voxel_stage_out[:rows]

    def readPointStore(self, first, n):
        """Points [first, first + n) of the device point store as (n, 4) float32 (keepPointStore must be on)."""
        out = np.zeros((int(n), 4), dtype=np.float32)
        check(self._lib.pcp_voxel_store_read(self._ctx, int(first), int(n), f32p(out)))
        return out

    def readUnknownOfLastFrame(self):
        """Global ids of the points of the last fed frame whose voxel did not exist yet (grouped by voxel)."""
        n = C.c_int(0)
        check(self._lib.pcp_voxel_add_unknown_read(self._ctx, None, 0, C.byref(n)))
        ids = np.zeros(max(n.value, 1), dtype=np.int32)
        if n.value:
            check(self._lib.pcp_voxel_add_unknown_read(self._ctx, i32p(ids), n.value, C.byref(n)))
        return ids[:n.value]

    def insertUnknownVoxels(self, max_points):
        """Voxelizer.storeUnknownPoints (Voxelizer.py:170-306) on the device for the unknown points of the
        last prepareDispatchVoxelizer call -> (stored voxels now, data rows the device has)."""
        self._feed_clean = False
        stored, cap = C.c_int(0), C.c_int(0)
        check(self._lib.pcp_voxel_insert_unknown(self._ctx, int(max_points), C.byref(stored), C.byref(cap)))
        self._voxel_resident = stored.value
        return stored.value, cap.value

    def readVoxelIndex(self, rows, voxel_index):
        """Fills voxel_index[:rows] (int32, C-contiguous) from the device table."""
        if voxel_index.dtype != np.int32 or not voxel_index.flags.c_contiguous:
            raise PcpError("voxel_index must be a C-contiguous int32 array")
        check(self._lib.pcp_voxel_index_read(self._ctx, int(rows), i32p(voxel_index)))

    def readVoxelCounts(self, first_row, n_rows):
        out = np.zeros(int(n_rows), dtype=np.int32)
        check(self._lib.pcp_voxel_counts_read(self._ctx, int(first_row), int(n_rows), i32p(out)))
        return out

    def prepareDispatchVoxelizer(self, np_points, voxel_index, voxel_data, voxel_size, stored_voxel_num,
                                 begin_index, max_points_to_store, realloc_needed, prev_stored_voxels,
                                 debug=False):
        """:847-944 -> (unknown int32[pad16(1 + N)], debug (N, 4) float32 | None).
        unknown[0] is the count, unknown[1:1+count] the global ids of the points
        whose voxel is not in the table (in no particular order)."""
        self._feed_clean = False
        n = len(np_points)
        unknown_len = 1 + n
        unknown_len += 16 - unknown_len % 16
        unknown_points = np.zeros(unknown_len, dtype=np.int32)
        if self._voxel_resident is not None and self._voxel_resident == int(stored_voxel_num):
            pass        # table and rows were created on the device (insertUnknownVoxels): nothing to upload
        else:
            self.bufferVoxelIndex(voxel_index)
            if realloc_needed:
                self.fullBufferVoxelData(voxel_data)
            else:
                new_rows = np.ascontiguousarray(voxel_data[prev_stored_voxels:stored_voxel_num], dtype=np.int32)
                check(self._lib.pcp_voxel_data_upload_rows(self._ctx, i32p(new_rows), int(prev_stored_voxels),
                                                           len(new_rows), int(max_points_to_store)))
        debug_data = np.zeros((n, 4), dtype=np.float32) if debug else None
        size = float(np.asarray(voxel_size).reshape(-1)[0])
        np_points = np.asarray(np_points)
        if (np_points.dtype == np.float32 and np_points.ndim == 2 and np_points.shape[1] == 3 and n > 1
                and np_points.T.flags.c_contiguous and not np_points.flags.c_contiguous):
            # float32 coordinate rows (a transposed cloud): interleaved and padded on the device
            check(self._lib.pcp_voxelize_rows(self._ctx, f32p(np_points.T), n, size, int(stored_voxel_num),
                                              int(begin_index), int(max_points_to_store), i32p(unknown_points),
                                              unknown_len, f32p(debug_data)))
        elif np_points.dtype == np.float32 and np_points.ndim == 2 and np_points.shape[1] == 3 and np_points.flags.c_contiguous:
            # already float32 (x, y, z): the zero fourth component of pad_points is added on the device
            check(self._lib.pcp_voxelize_xyz(self._ctx, f32p(np_points), n, size, int(stored_voxel_num),
                                             int(begin_index), int(max_points_to_store), i32p(unknown_points),
                                             unknown_len, f32p(debug_data)))
        else:
            pts = self.pad_points(np_points)
            check(self._lib.pcp_voxelize(self._ctx, f32p(pts), n, size, int(stored_voxel_num), int(begin_index),
                                         int(max_points_to_store), i32p(unknown_points), unknown_len, f32p(debug_data)))
        return unknown_points, debug_data

    def prepareDispatchVoxelStager(self, voxel_index, stored_voxel_num, max_points, voxel_statistics,
                                   filter_outliers, stage_everything=False, max_staging=1024, debug=False, fetch_rows=True):
        """:761-835.  voxel_index is not uploaded here (the reference's upload is
        commented out, :772-775): the table of the last bufferVoxelIndex /
        prepareDispatchVoxelizer call is used.  voxel_statistics is updated in place."""
        self._feed_clean = False
        if self.voxel_stage_out is None or self.voxel_stage_out.shape != (max_staging, max_points):
            self.voxel_stage_out = np.zeros((max_staging, max_points), dtype=np.int32)
        if voxel_statistics.dtype != np.float32 or not voxel_statistics.flags.c_contiguous:
            raise PcpError("voxel_statistics must be a C-contiguous float32 array")
        counter = np.zeros((1, 16), dtype=np.int32)
        if self.resident_statistics:
            # the statistics stay on the device (readVoxelStatistics fetches them); voxel_statistics only gives the row count.
            # fetch_rows=False (a caller that resolves the staged rows on the device): the rows stay there too, only their
            # number comes back -- readStagedRows fetches them if somebody asks after all
            check(self._lib.pcp_voxel_stage_resident(self._ctx, len(voxel_statistics), int(stored_voxel_num), int(max_points),
                                                     int(bool(filter_outliers)), int(bool(stage_everything)), int(max_staging),
                                                     i32p(counter), i32p(self.voxel_stage_out) if fetch_rows else None))
            if not fetch_rows and not debug:
                return voxel_statistics, int(counter[0][0])
        else:
            check(self._lib.pcp_voxel_stage(self._ctx, f32p(voxel_statistics), len(voxel_statistics),
                                            int(stored_voxel_num), int(max_points), int(bool(filter_outliers)),
                                            int(bool(stage_everything)), int(max_staging), i32p(counter),
                                            i32p(self.voxel_stage_out)))
        staged = int(counter[0][0])
        if debug:
            return voxel_statistics, self.voxel_stage_out, counter
        if staged > 0:
            return voxel_statistics, self.voxel_stage_out[:staged]
        return voxel_statistics, []

    def readVoxelStatistics(self, out):
        """The device's statistics rows into out ((rows, 4) float32, C-contiguous), for resident_statistics."""
        check(self._lib.pcp_voxel_stat_read(self._ctx, len(out), f32p(out)))

    def keepPointStore(self, on=True):
        """Keep a device copy of the Voxelizer's global point store (every dispatched cloud at its begin_index), so
        that staged voxels can be resolved to points without the host (stagedPointsDevice)."""
        check(self._lib.pcp_voxel_store_enable(self._ctx, int(bool(on))))

    def stagedPointsDevice(self):
        """(device address of n x 3 float64, n, rows): the points of the rows the last prepareDispatchVoxelStager call
        staged, row after row, gathered on the device; valid until the next call."""
        ptr, n, rows = C.c_void_p(), C.c_size_t(0), np.zeros(1, dtype=np.int32)
        check(self._lib.pcp_voxel_staged_points(self._ctx, C.byref(ptr), C.byref(n), i32p(rows)))
        return (ptr.value or 0), int(n.value), int(rows[0])

    def stagedLogBegin(self, slot, capacity_points, after_stream=None):
        """Opens log `slot` (0 or 1): every frame fed from now on (voxelAddFrame, a chain's feed) appends the points of the
        rows it staged (pcp_voxel_log_begin).  after_stream: HIP stream that read the slot's previous contents."""
        check(self._lib.pcp_voxel_log_begin(self._ctx, int(slot), int(capacity_points), C.c_void_p(after_stream or 0)))

    def stagedLogEnd(self, slot):
        check(self._lib.pcp_voxel_log_end(self._ctx, int(slot)))

    def stagedLogTake(self, slot):
        """Waits for the frames fed while the log was open -> (device address of n x 3 float64, n, rows staged); valid
        until the slot is opened again."""
        ptr, n, rows = C.c_void_p(), C.c_size_t(0), np.zeros(1, dtype=np.int32)
        check(self._lib.pcp_voxel_log_take(self._ctx, int(slot), C.byref(ptr), C.byref(n), i32p(rows)))
        return (ptr.value or 0), int(n.value), int(rows[0])

    def getFullVoxelData(self, prev_stored_voxels, max_points, voxelizer_voxel_data):
        """:946-954 -- fills voxelizer_voxel_data[:prev_stored_voxels] in place."""
        out = voxelizer_voxel_data
        if (isinstance(out, np.ndarray) and out.dtype == np.int32 and out.flags.c_contiguous and out.ndim == 2
                and out.shape[1] == int(max_points) and len(out) >= int(prev_stored_voxels)):
            check(self._lib.pcp_voxel_data_read(self._ctx, int(prev_stored_voxels), int(max_points), i32p(out)))
            return
        tmp = np.empty((int(prev_stored_voxels), int(max_points)), dtype=np.int32)
        check(self._lib.pcp_voxel_data_read(self._ctx, len(tmp), int(max_points), i32p(tmp)))
        voxelizer_voxel_data[:prev_stored_voxels] = tmp

    # ------------------------------------------------------------------ life cycle
    def synchronize(self):
        check(self._lib.pcp_synchronize(self._ctx))

    def cleanup(self):
        """:966-998"""
        if getattr(self, "_frame", None) is not None:
            self._frame.cleanup()
            self._frame = None
        if self._ctx:
            self._lib.pcp_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


class IcpBatch:
    """Several prepared ComputeShader contexts of one device whose iteration loops
    run together: one launch per iteration covers every pair.  Per-pair results
    are those of ComputeShader.icpRun."""

    def __init__(self, computes):
        self._lib = _lib.load()
        self.computes = list(computes)
        arr = (C.c_void_p * len(self.computes))(*[c._ctx for c in self.computes])
        self._batch = C.c_void_p()
        check(self._lib.pcp_batch_create(arr, len(self.computes), C.byref(self._batch)))

    def upload_sources(self, clouds):
        """float64 (N, 3) source cloud per pair (hints are reset by prepare())."""
        keep = []
        for c, pts in zip(self.computes, clouds):
            b = np.ascontiguousarray(np.asarray(pts)[:, :3], dtype=np.float64)
            check(self._lib.pcp_upload_source(c._ctx, f64p(b), len(b)))
            keep.append(b)                       # the copies are in flight: the arrays live until the one wait below
            c._n_source = len(b)
        self.synchronize()

    def prepare(self, with_normals):
        check(self._lib.pcp_batch_prepare(self._batch, int(with_normals)))

    def launch(self, mode, iterations, tol=1e-6, full_iter=False):
        check(self._lib.pcp_batch_icp_launch(self._batch, int(mode), int(iterations), float(tol), int(full_iter)))

    def synchronize(self):
        check(self._lib.pcp_batch_synchronize(self._batch))

    def results(self, iterations, want_hb=False, want_points=False):
        """[(dx_trace (iters, 6), converged, hb_trace | None, moved points | None)] per pair."""
        out = []
        for c in self.computes:
            dx = np.zeros((max(iterations, 1), 6), dtype=np.float32)
            hb = np.zeros((max(iterations, 1), 8, 8), dtype=np.float32) if want_hb else None
            it, cv = C.c_int(0), C.c_int(0)
            pts = None
            if want_points:
                if getattr(c, "_n_source", None) is None:
                    raise PcpError("moved points are only returned for sources uploaded by upload_sources")
                pts = np.empty((c._n_source, 3), dtype=np.float64)
            check(self._lib.pcp_icp_fetch(c._ctx, f32p(dx), f32p(hb), C.byref(it), C.byref(cv), f64p(pts)))
            out.append((dx[:it.value], bool(cv.value), hb[:it.value] if want_hb else None, pts))
        return out

    def profile(self, on):
        check(self._lib.pcp_batch_profile_enable(self._batch, int(on)))

    def profile_read(self):
        """(summed ms of the batched LS launches that did work, their count, source points they processed)."""
        ms, n, pts = C.c_float(0), C.c_int(0), C.c_longlong(0)
        check(self._lib.pcp_batch_profile_read(self._batch, C.byref(ms), C.byref(n), C.byref(pts)))
        return ms.value, n.value, pts.value

    def close(self):
        if self._batch:
            self._lib.pcp_batch_destroy(self._batch)
            self._batch = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
