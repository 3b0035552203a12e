"""Voxel map leg over voxel sizes: Voxelizer.addPoints of a synthetic 1 M-point LiDAR-like cloud, fed as
sweeps of 125 k points (the reference feeds one sweep per frame), for voxel sizes 0.1 ... 2 m:
    python tools/bench_voxel_sweep.py [sweeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, Voxelizer, lidar  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
clouds = [lidar.synthetic_scan(s, rings=64, steps=1953) + np.array([0.5 * s, 0.0, 0.0]) for s in range(sweeps)]
total = sum(len(c) for c in clouds)
cs = ComputeShader(maxPointsPerCloud=200000)
for size in (2.0, 1.0, 0.5, 0.2, 0.1):
    best = None
    for rep in range(2):
        v = Voxelizer(cs, None)
        v.init(size)
        per = []
        for c in clouds:
            t0 = time.perf_counter()
            v.addPoints(c, None)
            per.append(time.perf_counter() - t0)
        stored = v.stored_voxels
        v.cleanup()
        if best is None or sum(per) < sum(best):
            best = per
    print(f"voxel {size:4.1f} m: {total} points in {sweeps} sweeps -> {stored:7d} voxels, {sum(best) * 1e3:8.2f} ms "
          f"({total / sum(best) / 1e6:6.1f} M points/s; first sweep {best[0] * 1e3:.2f} ms, last {best[-1] * 1e3:.2f} ms)", flush=True)
cs.cleanup()
