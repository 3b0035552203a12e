"""cProfile of Voxelizer.addPoints on the frames that create voxels (host side of the voxel feed)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, Voxelizer, lidar  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "kitti_pair.npz"))
frames = [lidar.frame_from_raw(g["raw0"]), lidar.frame_from_raw(g["raw1"])]
cs = ComputeShader(maxPointsPerCloud=200000)
for rep in range(2):
    v = Voxelizer(cs, None)
    v.init(0.5)
    pr = cProfile.Profile()
    pr.enable()
    v.addPoints(frames[0], None)
    v.addPoints(frames[1], None)
    pr.disable()
    if rep == 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
cs.cleanup()
