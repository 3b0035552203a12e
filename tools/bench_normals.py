"""Time of pcp_grid_estimate_normals (ring-less normals) on the device, results left resident:
    python tools/bench_normals.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import lidar
from pointcloudprocessing_amd._lib import check
from pointcloudprocessing_amd.cell_grid import CellGrid


def run(name, pts, cell, radius, reps=5):
    g = CellGrid(pts, cell=cell)
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        check(g._lib.pcp_grid_estimate_normals(g._g, float(radius), 3, None, None, None))
        best = min(best, time.perf_counter() - t)
    info = g.info()
    print(f"{name}: {len(pts)} points, cell {cell} m ({info['points'] / max(info['cells'], 1):.1f} per cell), radius {radius} m: "
          f"{best * 1e3:.2f} ms = {len(pts) / best / 1e6:.1f} M normals/s", flush=True)
    g.cleanup()


scan = lidar.synthetic_scan(1, rings=64, steps=1900)
for cell in (0.5, 0.25):
    run("one LiDAR-like sweep", scan, cell, 0.5)
rng = np.random.default_rng(1)
sweeps = [lidar.synthetic_scan(s, rings=64, steps=1900) + np.array([0.2 * s, 0.0, 0.0]) for s in range(8)]
world = np.vstack(sweeps)
for cell in (0.5, 0.25, 0.17):
    run("8 overlaid sweeps", world, cell, 0.5)
run("8 overlaid sweeps", world, 0.25, 0.25)
