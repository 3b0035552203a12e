"""Voxel map leg: Voxelizer.addPoints on the two KITTI frames (0.5 m), staging, and the split of the time."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, Voxelizer, lidar  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kitti_pair.npz"))
frames = [lidar.frame_from_raw(g["raw0"]), lidar.frame_from_raw(g["raw1"])]
cs = ComputeShader(maxPointsPerCloud=200000)
split = {}


def timed(obj, name):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        split[name] = split.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out
    setattr(obj, name, wrapper)


for m in ("prepareDispatchVoxelizer", "prepareDispatchVoxelStager", "insertUnknownVoxels", "readVoxelIndex",
          "readVoxelCounts", "pad_points"):
    timed(cs, m)
for rep in range(2):
    v = Voxelizer(cs, None)
    v.init(0.5)
    for m in ("handleAddingPoints", "storeUnknownPoints"):
        timed(v, m)
    for k, f in enumerate(frames * 2):
        t0 = time.perf_counter()
        v.addPoints(f, None)
        dt = time.perf_counter() - t0
        print(f"rep {rep} frame {k}: addPoints {dt * 1e3:7.2f} ms, {v.stored_voxels} voxels, {v.last_added_point} points  "
              + " ".join(f"{n}={t:.2f}" for n, t in split.items()))
        split.clear()
    t0 = time.perf_counter()
    v.stageAllRemaningVoxels() if hasattr(v, "stageAllRemaningVoxels") else None
    print(f"rep {rep}: stageAllRemaningVoxels {(time.perf_counter() - t0) * 1e3:7.2f} ms")
cs.cleanup()
