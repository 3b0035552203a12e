"""Frame-to-frame odometry of a run of raw velodyne frames as one stream of device work (pcp_seq_*).

The per-frame loop of the reference -- read the file, filter and rotate it (LidarDataReader.py:59-67,
:144-150), scan lines, prepare, the iteration loop (EnvironmentConstructor.getRefinedTransition :116-132
-> PointcloudIcpContainer.full_pt2pl :103-144) -- for F consecutive frames at once: frame k is the target
of pair k and the source of pair k - 1.  submit() returns at once; the upload of the next run overlaps
the registration of the current one.  Per-pair steps are those of the one-pair path, bit for bit.
"""
import ctypes as C

import numpy as np

from . import _lib, lidar
from ._lib import check, f32p, f64p, i32p
from .compute_shader import MODE_PLANE, MODE_POINT
from .icp_container import PointcloudIcpContainer


class PinnedFrames:
    """Raw frames (N, 4) float32 copied once into page-locked host memory, so that their upload is a
    true asynchronous copy.  Index it like a list."""

    def __init__(self, raws):
        self._lib = _lib.load()
        self._ptrs, self.arrays = [], []
        for raw in raws:
            raw = np.asarray(raw, dtype=np.float32)
            if raw.ndim != 2 or raw.shape[1] not in (3, 4):
                raise ValueError("a raw frame is (N, 4) float32 rows x, y, z, intensity (or (N, 3): intensity 0)")
            p = C.c_void_p()
            check(self._lib.pcp_host_alloc(max(1, raw.shape[0]) * 16, C.byref(p)))
            self._ptrs.append(p)
            a = np.ctypeslib.as_array((C.c_float * (4 * max(1, raw.shape[0]))).from_address(p.value)).reshape(-1, 4)[:len(raw)]
            a[:, :raw.shape[1]] = raw
            if raw.shape[1] == 3:
                a[:, 3] = 0.0
            self.arrays.append(a)

    def __len__(self):
        return len(self.arrays)

    def __getitem__(self, k):
        return self.arrays[k]

    def close(self):
        self.arrays = []
        for p in self._ptrs:
            self._lib.pcp_host_free(p)
        self._ptrs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SequencePipeline:
    def __init__(self, max_frames, max_points=131072, max_scans=100, device=0):
        self._lib = _lib.load()
        self._seq = C.c_void_p()
        check(self._lib.pcp_seq_create(int(device), int(max_frames), int(max_points), int(max_scans), C.byref(self._seq)))
        self.max_frames, self.max_points, self.max_scans, self.device = int(max_frames), int(max_points), int(max_scans), int(device)
        self._runs = {}          # ticket -> (frames kept alive, n_frames, iterations, point_to_plane, full_iter)
        self._icp = PointcloudIcpContainer(None)

    def submit(self, raw_frames, origin=None, point_to_plane=True, iterations=10, tol=1e-6, full_iter=False,
               min_range=2.5, rotation=None, placements=None):
        """raw_frames: sequence of (N, 4) float32 C-contiguous arrays (rows of a velodyne .bin), e.g. a
        PinnedFrames or a list of its items; they must not change until collect().  Returns a ticket.
        placements: per frame a (rotation (3, 3), offset (3,)) as imu_placements() makes them, or an (F, 12)
        array: frame f is turned by its rotation after decoding, and as the source of pair f - 1 it is
        shifted by its offset before the first iteration (EnvironmentConstructor.py:169, :200-209)."""
        frames = [raw_frames[k] for k in range(len(raw_frames))]
        for a in frames:
            if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 4
                    and a.flags.c_contiguous):
                raise ValueError("raw frames must be C-contiguous (N, 4) float32 arrays")
        ptrs = (C.c_void_p * len(frames))(*[a.ctypes.data for a in frames])
        sizes = np.array([len(a) for a in frames], dtype=np.int32)
        R = np.ascontiguousarray(lidar.aligning_rotation() if rotation is None else rotation, dtype=np.float64)
        o = np.zeros(3) if origin is None else np.ascontiguousarray(origin, dtype=np.float64)
        ticket = C.c_int(-1)
        place = None
        if placements is not None:
            place = np.ascontiguousarray(
                [np.concatenate([np.asarray(q, dtype=np.float64).reshape(9), np.asarray(g, dtype=np.float64).reshape(3)])
                 for q, g in placements] if not isinstance(placements, np.ndarray) else placements, dtype=np.float64)
            if place.shape != (len(frames), 12):
                raise ValueError("placements: one (rotation, offset) = 12 numbers per frame")
        check(self._lib.pcp_seq_submit_placed(self._seq, ptrs, i32p(sizes), len(frames), float(min_range), f64p(R), f64p(o),
                                              None if place is None else f64p(place),
                                              MODE_PLANE if point_to_plane else MODE_POINT, int(iterations), float(tol),
                                              int(bool(full_iter)), C.byref(ticket)))
        self._runs[ticket.value] = (frames, len(frames), int(iterations), bool(point_to_plane), bool(full_iter))
        return ticket.value

    def collect(self, ticket):
        """-> (steps, n_points): steps[p] = (dx trace (iters, 6) float32, converged) of pair p = (frame p <- frame p + 1),
        n_points[f] = points frame f kept after the ego filter."""
        steps, n_pts, _ = self.collect_with_poses(ticket)
        return steps, n_pts

    def collect_with_poses(self, ticket):
        """collect() plus every pair's (t_opt, R_opt) as full_pt2pl / full_pt2pt return them (PointcloudIcpContainer.py:140,
        :333-347: p <- R_opt p - t_opt maps frame p + 1 into frame p), computed on the device behind the loops (tracking
        grid + Kabsch, pcp_seq_collect_poses) and fetched in the same copy as the steps."""
        frames, n_frames, iterations, plane, full_iter = self._runs.pop(ticket)
        P = n_frames - 1
        dx = np.zeros((P, iterations, 6), dtype=np.float32)
        iters = np.zeros(P, dtype=np.int32)
        conv = np.zeros(P, dtype=np.int32)
        n_pts = np.zeros(n_frames, dtype=np.int32)
        t_opt, R_opt = np.zeros((P, 3)), np.zeros((P, 3, 3))
        check(self._lib.pcp_seq_collect_poses(self._seq, int(ticket), f64p(t_opt), f64p(R_opt), f32p(dx), iterations,
                                              i32p(iters), i32p(conv), i32p(n_pts)))
        return ([(dx[p, :iters[p]].copy(), bool(conv[p])) for p in range(P)], n_pts,
                [(t_opt[p], R_opt[p]) for p in range(P)])

    def poses(self, steps, point_to_plane=True, full_iter=False):
        """The same poses recomputed on the host from the steps (NumPy replay of the tracking grid + SVD, as the
        reference does it); collect_with_poses returns the device's."""
        out = []
        for dx, conv in steps:
            applied = len(dx) - 1 if (conv and point_to_plane and not full_iter) else len(dx)
            out.append(self._icp._replay(dx, applied))
        return out

    def profile(self, on):
        check(self._lib.pcp_seq_profile_enable(self._seq, int(on)))

    def profile_read(self):
        """(summed ms of the fused NN + least-squares launches of the last collected run that did work, their
        count, source points they processed); HIP events on the stream the launches ran on."""
        ms, n, pts = C.c_float(0), C.c_int(0), C.c_longlong(0)
        check(self._lib.pcp_seq_profile_read(self._seq, C.byref(ms), C.byref(n), C.byref(pts)))
        return ms.value, n.value, pts.value

    def profile_read_launches(self):
        """The same launches one by one: list of ms, entry i the launch of iteration i (entry 0 searches all rings)."""
        ms, n = (C.c_float * 64)(), C.c_int(0)
        check(self._lib.pcp_seq_profile_read_launches(self._seq, ms, 64, C.byref(n)))
        return [ms[i] for i in range(min(n.value, 64))]

    def close(self):
        if self._seq:
            self._lib.pcp_seq_destroy(self._seq)
            self._seq = C.c_void_p()
        self._runs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _pairs_of_runs(runs, n_frames, lo, local, point_to_plane, iterations, device, max_points, placements=None):
    """Runs of frames (frame_files.MemoryRuns / FrameFileRuns: consecutive runs share one frame) through a
    SequencePipeline, two submissions in flight; local[lo + first pair of a run + j] = (t_opt, R_opt)."""
    pipe = SequencePipeline(min(runs.run, max(2, n_frames)), max_points=max_points, device=device)
    try:
        pending = []

        def take(c0, ticket, slot):
            for j, pose in enumerate(pipe.collect_with_poses(ticket)[2]):
                local[lo + c0 + j] = pose
            runs.release(slot)

        for c0, arrays, slot in runs:
            if len(arrays) < 2:
                runs.release(slot)
                continue
            place = None if placements is None else placements[lo + c0:lo + c0 + len(arrays)]
            pending.append((c0, pipe.submit(arrays, point_to_plane=point_to_plane, iterations=iterations, placements=place), slot))
            if len(pending) == 2:
                take(*pending.pop(0))
        for item in pending:
            take(*item)
    finally:
        pipe.close()


def sequence_odometry_raw(raw_frames, point_to_plane=True, iterations=10, rank=0, world=1, device=0, chunk=256,
                          gather_device="cpu", max_points=131072):
    """Poses of every frame in the frame of the first, [(R_w, t_w)] with p0 = R_w p + t_w, from raw velodyne
    frames.  Pairs are dealt to the ranks in contiguous blocks (a rank then decodes every frame of its
    block once), each block runs through a SequencePipeline in chunks of `chunk` pairs (larger launches use the GPU
    better: 64 / 256 pairs per launch measured 7,000 / 7,400-7,600 pairs/s on an MI355X); poses are gathered once
    (sharding.gather_poses) and chained on the host."""
    from . import sharding
    from .frame_files import MemoryRuns
    n_pairs = len(raw_frames) - 1
    per = -(-n_pairs // world)
    lo, hi = min(n_pairs, rank * per), min(n_pairs, (rank + 1) * per)
    local = {}
    if hi > lo:
        pinned = PinnedFrames(raw_frames[lo:hi + 1])
        try:
            _pairs_of_runs(MemoryRuns(pinned, min(chunk, hi - lo) + 1, min(chunk, hi - lo)), hi - lo + 1, lo, local,
                           point_to_plane, iterations, device, max_points)
        finally:
            pinned.close()
    poses = sharding.gather_poses(local, n_pairs, rank, world, device=gather_device)
    return sharding.compose_odometry(poses)


def sequence_odometry_files(paths, point_to_plane=True, iterations=10, rank=0, world=1, device=0, chunk=256,
                            gather_device="cpu", max_points=131072):
    """sequence_odometry_raw for frames that are still files (velodyne .bin, in the order given: frame_files.velodyne_files
    sorts a directory by name): a rank's files are read by a background thread straight into page-locked memory, one run
    ahead of the registration (LidarDataReader.getNextFrame reads a file when its frame is due, LidarDataReader.py:144-150)."""
    from . import sharding
    from .frame_files import FrameFileRuns
    paths = list(paths)
    n_pairs = len(paths) - 1
    per = -(-n_pairs // world)
    lo, hi = min(n_pairs, rank * per), min(n_pairs, (rank + 1) * per)
    local = {}
    if hi > lo:
        runs = FrameFileRuns(paths[lo:hi + 1], min(chunk, hi - lo) + 1, min(chunk, hi - lo), max_points=max_points)
        try:
            _pairs_of_runs(runs, hi - lo + 1, lo, local, point_to_plane, iterations, device, max_points)
        finally:
            runs.close()
    poses = sharding.gather_poses(local, n_pairs, rank, world, device=gather_device)
    return sharding.compose_odometry(poses)


def imu_placements(records):
    """Per frame (rotation, offset) from its OXTS record, as calculateTransition_imu forms them
    (EnvironmentConstructor.py:161-165, :200-205 with getCurrentOxtsData :104-105 and getDeltaVelocity :90-95):
    rotation = getYawRotation(I) @ inv(getTrueRotation(I)) -- the second matrix getCurrentOxtsData returns is inverted
    and multiplied from the right, whatever the driver's variable names suggest; offset = rotation @ (-velocity *
    seconds since the previous frame), the dead-reckoned motion since the previous frame (zero for the first)."""
    out, last = [], None
    for rec in records:
        rot = rec.getYawRotation(np.eye(3)) @ np.linalg.inv(rec.getTrueRotation(np.eye(3)))
        step = np.zeros(3)
        if last is not None:
            velocity = -rec.getVelocity()
            velocity *= (rec.getTime() - last).total_seconds()
            step = (rot @ velocity.T).T
        out.append((rot, step))
        last = rec.getTime()
    return out


def sequence_odometry_imu(raw_frames, records, point_to_plane=True, iterations=10, rank=0, world=1, device=0, chunk=256,
                          gather_device="cpu", max_points=131072):
    """Independent frame pairs whose initial guess comes from the IMU alone: frame k, turned by its IMU rotation, is
    the target of pair k at the origin; frame k + 1, turned by its own and shifted by the IMU's motion estimate, the
    source.  Pairs are dealt to the ranks in contiguous blocks and run through a SequencePipeline; the per-pair
    (t_opt, R_opt) are gathered once and chained as the driver chains them (EnvironmentConstructor.py:233-235):
    position += R_opt @ step - t_opt, rotation = R_opt @ imu_rotation.
    -> list of (imu_position, refined_position, imu_rotation, refined_rotation) per frame, like prev_poses."""
    from . import sharding
    place = imu_placements(records)
    n_pairs = len(raw_frames) - 1
    per = -(-n_pairs // world)
    lo, hi = min(n_pairs, rank * per), min(n_pairs, (rank + 1) * per)
    local = {}
    if hi > lo:
        from .frame_files import MemoryRuns
        pinned = PinnedFrames(raw_frames[lo:hi + 1])
        try:
            _pairs_of_runs(MemoryRuns(pinned, min(chunk, hi - lo) + 1, min(chunk, hi - lo)), hi - lo + 1, lo, local,
                           point_to_plane, iterations, device, max_points, placements=place)
        finally:
            pinned.close()
    pairs = sharding.gather_poses(local, n_pairs, rank, world, device=gather_device)
    imu_pos = refined = np.zeros(3)
    out = [(imu_pos, refined, place[0][0], place[0][0])]
    for k, (t_opt, R_opt) in enumerate(pairs):
        rot, step = place[k + 1]
        imu_pos = imu_pos + step
        refined = refined + ((R_opt @ step.T).T - t_opt)
        out.append((imu_pos, refined, rot, R_opt @ rot))
    return out
