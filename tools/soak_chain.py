"""The device chain over and over (default 150 drives of the thirty real sweeps, voxel map fed, two runs in flight): every
drive must end with the positions, steps and voxel count of the first.    python tools/soak_chain.py [drives [sink]]
With `sink` the voxels are 2 m wide and every staged voxel goes into a cell grid (MapSink through the staged-points log): the
number of voxels and of points handed over, and the grid's points per voxel, must not change from drive to drive either."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import CellGrid, ComputeShader, MapSink, PinnedFrames, Voxelizer  # noqa: E402
from pointcloudprocessing_amd.chain import OdometryChain, imu_rows  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

drives = int(sys.argv[1]) if len(sys.argv) > 1 else 150
with_sink = len(sys.argv) > 2 and sys.argv[2] == "sink"
raws, _ = bench.real_frames()
g = np.load(os.path.join(bench.GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
    rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
rd.finish()
rows = imu_rows([rd.getOx(k) for k in rd.getfilenames()[:len(raws)]])[0]
pinned = PinnedFrames(raws)
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
chain = OdometryChain(max_frames=10, max_points=1 << 17)
first, bad = None, 0
for d in range(drives):
    chain.reset()
    grid = CellGrid(np.zeros((0, 3)), cell=1.0) if with_sink else None
    sink = MapSink(grid, compute=cs) if with_sink else None
    vox = Voxelizer(cs, None, voxman=sink)
    vox.separate_colors, vox.filter_outliers = False, False
    vox.init(2.0 if with_sink else 0.5)
    track, pending = [], []
    for lo in range(0, len(raws), 10):
        pending.append(chain.submit([pinned[k] for k in range(lo, lo + 10)], rows[lo:lo + 10], voxelizer=vox))
        if len(pending) == 2:
            track += chain.collect(pending.pop(0))
    for t in pending:
        track += chain.collect(t)
    sig = (np.array([t[1] for t in track]).tobytes(), b"".join(t[4].tobytes() for t in track), int(vox.stored_voxels))
    if with_sink:
        coords, counts = np.unique(vox.vectorizedGetVoxelCoord(grid.sorted_points()[1]), axis=0, return_counts=True)
        sig += (sink.points_fed, sink.voxels_fed, coords.tobytes(), counts.tobytes())
        grid.cleanup()
    vox.cleanup()
    if first is None:
        first = sig
    elif sig != first:
        bad += 1
        print("drive", d, "differs", flush=True)
    if d % 25 == 0:
        print("drive", d, "done", flush=True)
print("drives that differ from the first:", bad)
chain.close()
pinned.close()
cs.cleanup()
