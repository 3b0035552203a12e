"""Rounds of the all-rings pre-pass (first_ring<8>) per wave, from the instrumented library
(make -C pointcloudprocessing_amd/csrc stats):
    PCP_LIB_PATH=pointcloudprocessing_amd/libpcp_stats.so python tools/first_ring_stats.py
on the device chain over the thirty real sweeps (sources placed by the IMU) and on the plain pair 0 -> 1 (unplaced)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import ComputeShader, PinnedFrames, PointcloudIcpContainer, _lib, lidar  # noqa: E402
from pointcloudprocessing_amd.chain import OdometryChain, imu_rows  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

lib = _lib.load()
lib.pcp_stats2_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()


def report(name):
    lib.pcp_stats2_read(out, 1)
    v = list(out)
    waves = max(v[0], 1)
    print(f"{name}: {v[0]} waves, {v[4]} queries, {v[1] / waves:.2f} loop rounds per wave, {v[2] / waves:.2f} of them searching, "
          f"{v[3] / max(v[4], 1):.2f} ring searches per query")
    print("   waves by searching rounds 0..6, 7+: " + " ".join(f"{100.0 * x / waves:.1f}%" for x in v[8:16]))


raws, _ = bench.real_frames()
g = np.load(os.path.join(bench.GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
    rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
rd.finish()
rows = imu_rows([rd.getOx(k) for k in rd.getfilenames()[:len(raws)]])[0]
pinned = PinnedFrames(raws)
chain = OdometryChain(max_frames=len(raws), max_points=1 << 17)
lib.pcp_stats2_read(out, 1)
chain.collect(chain.submit(pinned, rows))
report("chain, 29 IMU-placed frames")
chain.close()
pinned.close()
a, b = lidar.frame_from_raw(raws[0][:, :3]), lidar.frame_from_raw(raws[1][:, :3])
cs = ComputeShader(maxPointsPerCloud=200000)
icp = PointcloudIcpContainer(cs)
icp.full_pt2pl(a, b, np.zeros(3), iterations=2)
report("pair 0 -> 1, unplaced")
cs.cleanup()
