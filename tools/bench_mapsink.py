"""Voxelizer -> MapSink -> CellGrid per frame, staged voxels resolved on the host (the reference's hand-off) or on the
device (device point store):  python tools/bench_mapsink.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, MapSink, Voxelizer, lidar  # noqa: E402
from pointcloudprocessing_amd.cell_grid import CellGrid  # noqa: E402
from bench import real_frames  # noqa: E402

raws, _ = real_frames()
clouds = [lidar.frame_from_raw(r[:, :3]) + np.array([0.0, 0.0, -0.33 * k]) for k, r in enumerate(raws)]
for device_path in (False, True, False, True):
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
    grid = CellGrid(np.zeros((0, 3)), cell=0.5)
    sink = MapSink(grid, compute=cs if device_path else None)
    vox = Voxelizer(cs, None, voxman=sink)
    vox.init(0.5)
    vox.separate_colors, vox.filter_outliers = False, False
    per = []
    for c in clouds:
        t0 = time.perf_counter()
        vox.addPoints(c, None)
        cs.synchronize()
        per.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    vox.stageAllRemaningVoxels()
    flush = time.perf_counter() - t0
    print("%s: addPoints per frame ms %s | final staging of everything %.1f ms | %d points in the map, %d voxels handed over"
          % ("device" if device_path else "host  ", " ".join("%.1f" % (p * 1e3) for p in per), flush * 1e3,
             grid.info()["points"], sink.voxels_fed))
    vox.cleanup(); grid.cleanup(); cs.cleanup()
