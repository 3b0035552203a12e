"""Voxel map leg over voxel sizes: a synthetic 1 M-point LiDAR-like cloud fed as sweeps of about 100 k points (the
reference feeds one sweep per frame) for voxel sizes 0.1 ... 2 m, three ways:
  host     Voxelizer.addPoints(host float64 array): float32 cast on the host, upload, one enqueue + one wait per frame
  device   Voxelizer.addPointsDevice: the sweep already in device memory (float64, cast on the device), one wait per frame
  piped    pcp_voxel_add_dev for every sweep, then the waits: the device's own voxel count carries from frame to frame
    python tools/bench_voxel_sweep.py [sweeps] [--kitti]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, Voxelizer, _lib, lidar  # noqa: E402
from pointcloudprocessing_amd.frame import DeviceFrame  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
sweeps = int(args[0]) if args else 10
if "--kitti" in sys.argv:
    g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    raws = [np.load(os.path.join(g, "kitti_pair.npz"))[k] for k in ("raw0", "raw1")] + lidar.unpack_raw_frames(np.load(os.path.join(g, "kitti_seq.npz")))
    clouds = [lidar.frame_from_raw(r) + np.array([0.0, 0.0, -0.33 * k]) for k, r in enumerate(raws[:sweeps])]
else:
    clouds = [lidar.synthetic_scan(s, rings=64, steps=1953) + np.array([0.5 * s, 0.0, 0.0]) for s in range(sweeps)]
clouds = [np.ascontiguousarray(c) for c in clouds]
total = sum(len(c) for c in clouds)
cs = ComputeShader(maxPointsPerCloud=200000)
lib = cs._lib
frames = []
for c in clouds:
    f = DeviceFrame(max_points=200000)
    f.upload(c)
    frames.append(f)


def run(kind, size):
    v = Voxelizer(cs, None)
    v.init(size)
    per = []
    if kind == "piped":
        t0 = time.perf_counter()
        tickets, begin = [], 0
        rec = _lib.VoxelFrame()
        for k, f in enumerate(frames):
            ptr, n = f.points_dev()
            t = C.c_int(-1)
            _lib.check(lib.pcp_voxel_add_dev(cs._ctx, C.c_void_p(ptr), 8, n, float(size), 0 if not tickets else -1, begin, 1024, 0, 1024,
                                             None, C.byref(t)))
            tickets.append(t.value)
            begin += n
            if k >= 4:                       # four frames in flight (the ring of records holds eight)
                _lib.check(lib.pcp_voxel_add_wait(cs._ctx, tickets[k - 4], C.byref(rec)))
        for t in tickets[-4:]:
            _lib.check(lib.pcp_voxel_add_wait(cs._ctx, t, C.byref(rec)))
        per = [time.perf_counter() - t0]
        stored = rec.stored
    else:
        for c, f in zip(clouds, frames):
            t0 = time.perf_counter()
            if kind == "host":
                v.addPoints(c, None)
            else:
                ptr, n = f.points_dev()
                v.addPointsDevice(ptr, n, layout=8)
            per.append(time.perf_counter() - t0)
        stored = v.stored_voxels
    v.cleanup()
    return per, stored


for size in (2.0, 1.0, 0.5, 0.2, 0.1):
    for kind in ("host", "device", "piped"):
        best, stored = None, 0
        for rep in range(3):
            per, stored = run(kind, size)
            if best is None or sum(per) < sum(best):
                best = per
        extra = "" if kind == "piped" else f"; first sweep {best[0] * 1e3:.3f} ms, median later {np.median(best[1:]) * 1e3:.3f} ms"
        print(f"voxel {size:4.1f} m  {kind:6s}: {total} points in {sweeps} sweeps -> {stored:7d} voxels, {sum(best) * 1e3:8.3f} ms "
              f"({total / sum(best) / 1e6:7.1f} M points/s{extra})", flush=True)
for f in frames:
    f.cleanup()
cs.cleanup()
