"""Where a frame of the host-driven driver goes (EnvironmentConstructor.calculateTransition_imu frame by frame over the thirty
real sweeps, the reference's own call surface): cProfile of one drive after a warm-up drive.
    python tools/profile_host_drive.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import ComputeShader, EnvironmentConstructor, PointcloudIcpContainer, Voxelizer, lidar  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

raws, _ = bench.real_frames()
g = np.load(os.path.join(bench.GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
    rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
rd.finish()
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
reader = lidar.LidarDataReader()
for k, raw in enumerate(raws):
    reader.add_frame(f"{k:010d}", raw)
env = EnvironmentConstructor(None, rd, reader, PointcloudIcpContainer(cs), cs, voxelizer=Voxelizer(cs, None))


def drive():
    env.init(0.5)
    for _ in range(len(raws)):
        frame, rec = env.getNextFrameData(0)
        env.calculateTransition_imu(frame, rec, point_to_plane=True, iterations=10)
    cs.synchronize()


drive()
c0 = time.perf_counter()
drive()
print("%.3f ms per frame" % ((time.perf_counter() - c0) / len(raws) * 1e3))
pr = cProfile.Profile()
pr.enable()
drive()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
