"""Velodyne .bin files on their way to the device: read by a background thread straight into page-locked memory, run by run.

The reference reads one file per frame on the thread that then registers it (LidarDataReader.getNextFrame ->
eatBytes, LidarDataReader.py:144-150: np.fromfile, reshape(-1, 4), a float64 copy) and lists the directory in the
order the file system returns it (:125-127).  Here the files are taken in the order of their names, a run of them is read
with readinto() into buffers the GPU can copy from asynchronously (no intermediate array), and the next run is read while
the current one is registered.
"""
import ctypes as C
import os
import queue
import threading

import numpy as np

from . import _lib
from ._lib import check


def velodyne_files(directory):
    """The .bin files of a velodyne_points/data directory in the order of their names (= of the sweeps)."""
    return [os.path.join(directory, f) for f in sorted(os.listdir(directory)) if f.endswith(".bin")]


class FrameFileRuns:
    """Iterates over runs of consecutive frames of `paths`: run i holds frames [i * stride, i * stride + run) (the last
    run may be shorter; with stride < run consecutive runs share frames, as the pair pipeline needs: the last frame of
    one run is the first target of the next).  Yields (first_frame, arrays, slot): arrays are (N, 4) float32 views of
    page-locked buffers that stay valid until release(slot) hands the slot back to the reader.  `slots` runs exist at a
    time: two in flight on the device and one being read."""

    def __init__(self, paths, run, stride=None, slots=3, max_points=131072):
        self._lib = _lib.load()
        self.paths = list(paths)
        self.run, self.stride = int(run), int(stride if stride is not None else run)
        if self.run < 1 or self.stride < 1 or self.stride > self.run:
            raise ValueError("run >= stride >= 1")
        self.max_points = int(max_points)
        self._bufs = []                               # [slot][k] -> (pointer, ctypes array)
        for _ in range(int(slots)):
            row = []
            for _ in range(self.run):
                p = C.c_void_p()
                check(self._lib.pcp_host_alloc(self.max_points * 16, C.byref(p)))
                row.append((p, (C.c_char * (self.max_points * 16)).from_address(p.value)))
            self._bufs.append(row)
        self._free, self._ready = queue.Queue(), queue.Queue(maxsize=int(slots))
        for s in range(int(slots)):
            self._free.put(s)
        self._stop = False
        self._thread = threading.Thread(target=self._read_all, daemon=True)
        self._thread.start()

    def starts(self):
        n = len(self.paths)
        if n == 0:
            return []
        if self.stride == self.run:
            return list(range(0, n, self.run))
        # overlapping runs: a run that would only repeat the shared frame is not made
        return [lo for lo in range(0, max(1, n - (self.run - self.stride)), self.stride)]

    def _read_all(self):
        try:
            for lo in self.starts():
                slot = self._free.get()
                if self._stop or slot is None:
                    return
                arrays = []
                for k, path in enumerate(self.paths[lo:lo + self.run]):
                    size = os.path.getsize(path)
                    if size % 16 or size > self.max_points * 16:
                        raise ValueError("%s: %d bytes is not a whole number of (x, y, z, intensity) float32 rows within %d points"
                                         % (path, size, self.max_points))
                    _, raw = self._bufs[slot][k]
                    with open(path, "rb", buffering=0) as f:
                        got = f.readinto(memoryview(raw)[:size]) if size else 0
                    if got != size:
                        raise IOError("%s: read %d of %d bytes" % (path, got, size))
                    arrays.append(np.frombuffer(raw, dtype=np.float32, count=size // 4).reshape(-1, 4))
                self._ready.put((lo, arrays, slot))
            self._ready.put(None)
        except Exception as e:                        # handed to the consumer, which raises it
            self._ready.put(e)

    def __iter__(self):
        while True:
            item = self._ready.get()
            if item is None:
                return
            if isinstance(item, Exception):
                raise item
            yield item

    def release(self, slot):
        self._free.put(slot)

    def close(self):
        self._stop = True
        self._free.put(None)
        while self._thread.is_alive():                # a reader waiting to hand over a run nobody will take
            try:
                self._ready.get(timeout=0.05)
            except queue.Empty:
                pass
        self._thread.join()
        for row in self._bufs:
            for p, _ in row:
                self._lib.pcp_host_free(p)
        self._bufs = []

    def __del__(self):
        try:
            if self._bufs:
                self.close()
        except Exception:
            pass


class MemoryRuns:
    """The same protocol over frames already in (page-locked) memory: nothing to read, nothing to release."""

    def __init__(self, frames, run, stride=None):
        self.frames, self.run, self.stride = frames, int(run), int(stride if stride is not None else run)

    def __iter__(self):
        n = len(self.frames)
        if n == 0:
            return
        last = n if self.stride == self.run else max(1, n - (self.run - self.stride))
        for lo in range(0, last, self.stride):
            yield lo, [self.frames[k] for k in range(lo, min(n, lo + self.run))], None

    def release(self, slot):
        pass

    def close(self):
        pass
