"""Scan-to-map leg on one GPU: a map tiled from synthetic sweeps along a straight
trajectory, one 120k-point scan registered against it (point-to-point, exact neighbours within 0.4 m).
    python tools/bench_map.py [n_sweeps] [quick|ingest]      (quick: without the sweep over cell sizes;
    ingest: the same map grown from its first sweeps by appends of 20 sweeps at a time, then the same registration)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import lidar, scan_to_map as S  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
base = [lidar.synthetic_scan(s) for s in range(4)]
parts = [base[k % 4] + np.array([0.0, 0.0, 1.0 * k]) for k in range(sweeps)]
world_pts = np.vstack(parts)
k = sweeps // 2
R = lidar.rotation_xyz(0.004, -0.006, 0.005)
t = np.array([0.03, -0.01, 0.02])
rng = np.random.default_rng(0)
scan = (R @ parts[k].T).T + t + rng.normal(0, 0.003, parts[k].shape)
t0 = time.perf_counter()
reg = S.MapRegistration(world_pts, 0.5, nn_cell=0.15)
build_wall = time.perf_counter() - t0
info = reg.backend.grid.info()
print(f"map: {len(world_pts)} points, {info['cells']} cells, device build {info['build_ms']:.1f} ms "
      f"(sort {info['sort_ms']:.2f} ms x{info['sort_passes']}), wall incl. upload {build_wall * 1e3:.0f} ms")
for mode_name, check in (("loop with a done-check every iteration", 1), ("loop without host checks", 0)):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        t_opt, R_opt, dx, conv = reg.register(scan, S.MODE_POINT, 0.4, max_iter=10, check_every=check)
        best = min(best, time.perf_counter() - t0)
    err_R = np.abs(R_opt @ R - np.eye(3)).max()
    print(f"{mode_name}: {best * 1e3:.2f} ms per registration incl. 2.9 MB scan upload, {len(dx)} iterations, "
          f"rotation error {err_R:.1e}")
g = reg.backend.grid
g.set_source(scan)
ms = min(g.query(return_kernel_ms=True, distance_upper_bound=0.4)[2] for _ in range(3))
print(f"bounded NN of the scan: {ms * 1e3:.1f} us = {len(scan) / ms / 1e3:.0f} M queries/s")
reg.close()
if len(sys.argv) > 2 and sys.argv[2] == "ingest":
    step = 20
    t0 = time.perf_counter()
    grown = S.MapRegistration(np.vstack(parts[:step]), 0.5, nn_cell=0.15)
    first_wall = time.perf_counter() - t0
    walls, dev = [], []
    for k0 in range(step, sweeps, step):
        batch = np.vstack(parts[k0:k0 + step])
        t0 = time.perf_counter()
        grown.ingest(batch)
        walls.append(time.perf_counter() - t0)
        dev.append(grown.backend.grid.info()["build_ms"])
    print(f"grown by ingests: first {step} sweeps {first_wall * 1e3:.0f} ms, then {len(walls)} appends of {step} sweeps "
          f"({len(batch)} points): wall incl. upload median {np.median(walls) * 1e3:.1f} ms, last {walls[-1] * 1e3:.1f} ms; "
          f"table rebuild on the device last {dev[-1]:.1f} ms; total {first_wall + sum(walls):.2f} s, "
          f"{grown.backend.grid.info()['points']} points")
    t2, R2, dx2, conv2 = grown.register(scan, S.MODE_POINT, 0.4, max_iter=10, check_every=0)
    same = dx2.shape == dx.shape and np.array_equal(dx2.view(np.uint32), dx.view(np.uint32))
    print(f"registration against the grown map: steps bit-equal to the built map's: {same}")
    grown.close()
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "quick":
    sys.exit(0)
from pointcloudprocessing_amd.cell_grid import CellGrid  # noqa: E402
g = CellGrid(world_pts[:1000], cell=1.0)
for cell in (0.05, 0.1, 0.15, 0.25, 0.5, 1.0):
    g.rebuild(world_pts, cell=cell)
    info = g.info()
    g.set_source(scan)
    ms = min(g.query(return_kernel_ms=True, distance_upper_bound=0.4)[2] for _ in range(3))
    ms_u = min(g.query(return_kernel_ms=True)[2] for _ in range(2))
    print(f"cell {cell:5.2f}: {info['cells']:9d} cells, build {info['build_ms']:6.1f} ms, bounded(0.4) NN {ms * 1e3:8.1f} us "
          f"= {len(scan) / ms / 1e3:7.0f} M/s, unbounded {ms_u * 1e3:9.1f} us")
g.cleanup()
