"""The thirty real sweeps as one drive through the device chain (pcp_chain_*), a few times over: wall and device
time per frame.  Run under rocprofv3 --kernel-trace --stats for the per-kernel split (tools/profile_chain_r03.sh)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import CellGrid, ComputeShader, MapSink, PinnedFrames, Voxelizer  # noqa: E402
from pointcloudprocessing_amd.chain import OdometryChain, imu_rows  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
with_map = not (len(sys.argv) > 2 and sys.argv[2] == "nomap")
with_sink = len(sys.argv) > 2 and sys.argv[2] == "sink"      # staged voxels go to a cell grid (2 m voxels so that rows fill up)
voxel = 2.0 if with_sink else 0.5
raws, _ = bench.real_frames()
g = np.load(os.path.join(bench.GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
    rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
rd.finish()
rows = imu_rows([rd.getOx(k) for k in rd.getfilenames()[:len(raws)]])[0]
pinned = PinnedFrames(raws)
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
chain = OdometryChain(max_frames=len(raws), max_points=1 << 17)
for rep in range(reps):
    chain.reset()
    vox = None
    grid = sink = None
    if with_map:
        if with_sink:
            grid = CellGrid(np.zeros((0, 3)), cell=1.0)
            sink = MapSink(grid, compute=cs)
        vox = Voxelizer(cs, None, voxman=sink)
        vox.separate_colors, vox.filter_outliers = False, False
        vox.init(voxel)
    cs.synchronize()
    c0 = time.perf_counter()
    ticket = chain.submit(pinned, rows, voxelizer=vox)
    enq = time.perf_counter() - c0
    track = chain.collect(ticket)
    wall = time.perf_counter() - c0
    print("rep %d: %.3f ms per frame wall (%.3f of it the enqueue), %.3f ms per frame on the chain's stream, %.1f iterations per frame%s"
          % (rep, wall / len(raws) * 1e3, enq / len(raws) * 1e3, chain.last_chain_ms / len(raws), np.mean([len(t[4]) for t in track[1:]]),
             (", %d voxels" % vox.stored_voxels if vox else "")
             + (", %d points of %d staged voxels in the map" % (sink.points_fed, sink.voxels_fed) if sink else "")), flush=True)
    if vox:
        vox.cleanup()
    if grid:
        grid.cleanup()
chain.close()
pinned.close()
cs.cleanup()
