"""The per-frame driver loop of a drive as one stream of device work (pcp_chain_*).

EnvironmentConstructor.calculateTransition_imu (EnvironmentConstructor.py:154-265) handles one frame per call and needs
the previous frame's refined position for the next one's guess; OdometryChain takes a run of raw velodyne frames with their
IMU rows and keeps that chain on the device: guess, placement, registration against the reference cloud of the block,
refined cloud and position, voxel map feed, replacement of the reference cloud every `reset_every` frames.  The host
enqueues and waits twice per run (for the frames' point counts, for the results).
"""
import ctypes as C

import numpy as np

from . import _lib, lidar
from ._lib import check, f32p, f64p, i32p
from .compute_shader import MODE_PLANE, MODE_POINT


def imu_rows(records, last_time=None):
    """Per OXTS record the 12 numbers a chain wants: the frame's IMU rotation, current_vert_rot @ inv(current_horz_rot) of
    EnvironmentConstructor.py:161-164 (getCurrentOxtsData :104-105 returns getTrueRotation first, getYawRotation second), and
    its dead-reckoned step rotation @ (-velocity * seconds since the previous frame) (getDeltaVelocity :90-95; zero for a
    frame without a predecessor).  last_time: time stamp of the frame before records[0], when the drive continues.
    -> (rows (n, 12) float64, rotations (n, 3, 3), steps (n, 3), times)."""
    rows, rots, steps, times = [], [], [], []
    for rec in records:
        rot = rec.getYawRotation(np.eye(3)) @ np.linalg.inv(rec.getTrueRotation(np.eye(3)))
        step = np.zeros(3)
        if last_time is not None:
            velocity = -rec.getVelocity()
            velocity *= (rec.getTime() - last_time).total_seconds()
            step = (rot @ velocity.T).T
        last_time = rec.getTime()
        rows.append(np.concatenate([rot.reshape(9), step]))
        rots.append(rot); steps.append(step); times.append(last_time)
    return np.ascontiguousarray(rows, dtype=np.float64), np.array(rots), np.array(steps), times


class OdometryChain:
    """One drive on one device.  submit() takes the next frames of the drive, collect() returns per frame
    (guess, refined position, t_opt, R_opt, steps (iters, 6) float32, converged)."""

    def __init__(self, max_frames=32, max_points=131072, max_scans=100, device=0, reset_every=10, stage_log_points=None):
        self._lib = _lib.load()
        self._chain = C.c_void_p()
        check(self._lib.pcp_chain_create(int(device), int(max_frames), int(max_points), int(max_scans), C.byref(self._chain)))
        self.max_frames, self.max_points, self.max_scans, self.device = int(max_frames), int(max_points), int(max_scans), int(device)
        self.reset_every = int(reset_every)          # EnvironmentConstructor.reset_origin_treshold (:35)
        self.stage_log_points = stage_log_points     # room for the points a run's frames stage (default: twice the run's points)
        self._runs = {}

    def reset(self):
        check(self._lib.pcp_chain_reset(self._chain))

    def submit(self, raw_frames, imu, point_to_plane=True, iterations=10, tol=1e-6, closing_iterations=15, voxelizer=None,
               min_range=2.5, rotation=None):
        """raw_frames: (N, 4) float32 C-contiguous arrays (rows of a velodyne .bin; page-locked ones, sequence.PinnedFrames,
        upload asynchronously) that must not change until collect(); imu: (F, 12) rows of imu_rows().  voxelizer: a
        Voxelizer of this package whose map lives on the device (init() called): every frame's final cloud is fed to it as
        Voxelizer.addPoints would (EnvironmentConstructor.py:189, :245); when its voxel manager is a map_sink.MapSink
        working on the device, the points of every voxel the run's frames stage reach it at collect().  Returns a ticket."""
        frames = [raw_frames[k] for k in range(len(raw_frames))]
        for a in frames:
            if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 4 and a.flags.c_contiguous):
                raise ValueError("raw frames must be C-contiguous (N, 4) float32 arrays")
        imu = np.ascontiguousarray(imu, dtype=np.float64)
        if imu.shape != (len(frames), 12):
            raise ValueError("imu: 12 numbers (rotation, step) per frame")
        ptrs = (C.c_void_p * len(frames))(*[a.ctypes.data for a in frames])
        sizes = np.array([len(a) for a in frames], dtype=np.int32)
        R = np.ascontiguousarray(lidar.aligning_rotation() if rotation is None else rotation, dtype=np.float64)
        ticket, serial0 = C.c_int(-1), C.c_int(-1)
        kept = np.zeros(len(frames), dtype=np.int32)
        vctx, size, stored, begin, width, filt, log_slot = None, 0.0, 0, 0, 1024, 0, None
        if voxelizer is not None:
            vctx, size, stored, begin, width, filt = voxelizer._feed_begin()
            room = self.stage_log_points or max(1 << 20, 2 * len(frames) * self.max_points)
            log_begin = getattr(voxelizer, "_log_begin", None)      # a feed without a hand-off of staged voxels has none
            log_slot = log_begin(room) if log_begin else None
        try:
            check(self._lib.pcp_chain_submit(self._chain, ptrs, i32p(sizes), len(frames), float(min_range), f64p(R), f64p(imu),
                                             MODE_PLANE if point_to_plane else MODE_POINT, int(iterations), float(tol),
                                             int(closing_iterations), self.reset_every, vctx, float(size), int(stored), int(begin),
                                             int(width), int(filt), 1024, C.byref(ticket), i32p(kept), C.byref(serial0)))
        except Exception:
            if log_slot is not None:
                voxelizer._log_end(log_slot, drop=True)
            raise
        if log_slot is not None:
            voxelizer._log_end(log_slot)
        if voxelizer is not None:
            voxelizer._feed_noted(kept, serial0.value)
        self._runs[ticket.value] = (frames, len(frames), int(max(iterations, closing_iterations)), voxelizer, kept, log_slot)
        return ticket.value

    def collect(self, ticket):
        """-> list per frame of (guess (3,), position (3,), t_opt (3,), R_opt (3, 3), steps (iters, 6) float32, converged);
        self.last_chain_ms = device time of the run's chain."""
        frames, F, cap, voxelizer, kept, log_slot = self._runs.pop(ticket)
        guess, pos, t_opt, R_opt = np.zeros((F, 3)), np.zeros((F, 3)), np.zeros((F, 3)), np.zeros((F, 3, 3))
        dx = np.zeros((F, cap, 6), dtype=np.float32)
        iters, conv = np.zeros(F, dtype=np.int32), np.zeros(F, dtype=np.int32)
        ms = C.c_float(0)
        try:
            check(self._lib.pcp_chain_collect(self._chain, int(ticket), f64p(guess), f64p(pos), f64p(t_opt), f64p(R_opt), f32p(dx),
                                              cap, i32p(iters), i32p(conv), C.byref(ms)))
        finally:
            if voxelizer is not None:
                voxelizer._feed_wait(log_slot)
        self.last_chain_ms = float(ms.value)
        self.last_points = kept
        return [(guess[f], pos[f], t_opt[f], R_opt[f], dx[f, :iters[f]].copy(), bool(conv[f])) for f in range(F)]

    def read_refined(self, ticket, frame, n):
        """The final cloud of a frame of the collected run `ticket` ((n, 3) float64), while its buffers are still there."""
        out = np.empty((int(n), 3), dtype=np.float64)
        check(self._lib.pcp_chain_read_refined(self._chain, int(ticket), int(frame), f64p(out), int(n)))
        return out

    def close(self):
        if self._chain:
            self._lib.pcp_chain_destroy(self._chain)
            self._chain = C.c_void_p()
        self._runs = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _drive(runs, n_frames, records, voxelizer, point_to_plane, iterations, run, device, max_points, reset_every):
    rows, rots, steps, times = imu_rows(records)
    chain = OdometryChain(max_frames=min(run, n_frames), max_points=max_points, device=device, reset_every=reset_every)
    out, imu_pos = [], np.zeros(3)
    try:
        pending = []

        def take(lo, ticket, slot):
            nonlocal imu_pos
            for j, (_, pos, _, R_opt, _, _) in enumerate(chain.collect(ticket)):
                k = lo + j
                imu_pos = imu_pos + steps[k] if k else np.zeros(3)
                out.append((imu_pos, pos, rots[k], R_opt @ rots[k], times[k]))
            runs.release(slot)

        for lo, arrays, slot in runs:
            pending.append((lo, chain.submit(arrays, rows[lo:lo + len(arrays)], point_to_plane=point_to_plane,
                                             iterations=iterations, voxelizer=voxelizer), slot))
            if len(pending) == 2:
                take(*pending.pop(0))
        for item in pending:
            take(*item)
    finally:
        chain.close()
    return out


def chain_odometry(raw_frames, records, voxelizer=None, point_to_plane=True, iterations=10, run=32, device=0, max_points=131072,
                   reset_every=10):
    """A whole drive: [(imu_position, refined_position, imu_rotation, refined_rotation, time)] per frame -- the tuples
    EnvironmentConstructor keeps in prev_poses (:181, :257) -- from raw frames and their OXTS records, in runs of `run`
    frames with two runs in flight."""
    from .frame_files import MemoryRuns
    from .sequence import PinnedFrames
    pinned = PinnedFrames(raw_frames)
    try:
        return _drive(MemoryRuns(pinned, min(run, len(raw_frames))), len(raw_frames), records, voxelizer, point_to_plane,
                      iterations, run, device, max_points, reset_every)
    finally:
        pinned.close()


def chain_odometry_files(paths, records, voxelizer=None, point_to_plane=True, iterations=10, run=32, device=0, max_points=131072,
                         reset_every=10):
    """chain_odometry for frames that are still files (velodyne .bin in the order given; frame_files.velodyne_files sorts a
    directory by name): a background thread reads the next run straight into page-locked memory while the current one is
    on the device (the reference reads each file when its frame is due, LidarDataReader.py:144-150)."""
    from .frame_files import FrameFileRuns
    paths = list(paths)
    runs = FrameFileRuns(paths, min(run, len(paths)), max_points=max_points)
    try:
        return _drive(runs, len(paths), records, voxelizer, point_to_plane, iterations, run, device, max_points, reset_every)
    finally:
        runs.close()
