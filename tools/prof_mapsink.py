import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pointcloudprocessing_amd import ComputeShader, MapSink, Voxelizer, lidar
from pointcloudprocessing_amd.cell_grid import CellGrid
from bench import real_frames
raws, _ = real_frames()
clouds = [lidar.frame_from_raw(r[:, :3]) + np.array([0.0, 0.0, -0.33 * k]) for k, r in enumerate(raws)]
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
grid = CellGrid(np.zeros((0, 3)), cell=0.5)
sink = MapSink(grid, compute=cs)
T = {"staged": 0.0, "append": 0.0, "n": 0, "pts": 0}
orig_s, orig_a = cs.stagedPointsDevice, grid.append_device
def s2():
    t = time.perf_counter(); r = orig_s(); T["staged"] += time.perf_counter() - t; return r
def a2(ptr, n):
    t = time.perf_counter(); r = orig_a(ptr, n); grid.synchronize() if hasattr(grid, "synchronize") else None; T["append"] += time.perf_counter() - t; T["n"] += 1; T["pts"] += n; return r
cs.stagedPointsDevice, grid.append_device = s2, a2
vox = Voxelizer(cs, None, voxman=sink)
vox.init(0.5); vox.separate_colors, vox.filter_outliers = False, False
for c in clouds:
    vox.addPoints(c, None)
print("frames:", {k: (v * 1e3 if isinstance(v, float) else v) for k, v in T.items()})
for k in T: T[k] = 0 if isinstance(T[k], int) else 0.0
t0 = time.perf_counter(); vox.stageAllRemaningVoxels(); tot = time.perf_counter() - t0
print("final: total %.1f ms" % (tot * 1e3), {k: (v * 1e3 if isinstance(v, float) else v) for k, v in T.items()})
