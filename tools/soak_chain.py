"""The device chain over and over (default 150 drives of the thirty real sweeps, voxel map fed, two runs in flight): every
drive must end with the positions, steps and voxel count of the first.    python tools/soak_chain.py [drives]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import ComputeShader, PinnedFrames, Voxelizer  # noqa: E402
from pointcloudprocessing_amd.chain import OdometryChain, imu_rows  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

drives = int(sys.argv[1]) if len(sys.argv) > 1 else 150
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
    vox = Voxelizer(cs, None)
    vox.separate_colors, vox.filter_outliers = False, False
    vox.init(0.5)
    track, pending = [], []
    for lo in range(0, len(raws), 10):
        pending.append(chain.submit([pinned[k] for k in range(lo, lo + 10)], rows[lo:lo + 10], voxelizer=vox))
        if len(pending) == 2:
            track += chain.collect(pending.pop(0))
    for t in pending:
        track += chain.collect(t)
    sig = (np.array([t[1] for t in track]).tobytes(), b"".join(t[4].tobytes() for t in track), int(vox.stored_voxels))
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
