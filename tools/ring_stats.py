"""Lane utilisation of the ring search on the bench workload, from the instrumented library
(make -C pointcloudprocessing_amd/csrc stats):  PCP_LIB_PATH=pointcloudprocessing_amd/libpcp_stats.so python tools/ring_stats.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, lidar, _lib  # noqa: E402
from bench import base_cloud, moved_copy  # noqa: E402

lib = _lib.load()
lib.pcp_stats_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
a, _ = base_cloud()
b, R, t = moved_copy(a, 0)
cs = ComputeShader(maxPointsPerCloud=len(a))
icp = PointcloudIcpContainer(cs)
icp.preparePointToPlane(a, np.zeros(3), b)
out = (C.c_ulonglong * 8)()
lib.pcp_stats_read(out, 1)
dx, conv, _, _ = cs.icpRun(b, 1, 10)
lib.pcp_stats_read(out, 1)
v = list(out)
for name, o in (("hinted", 0), ("all rings", 4)):
    lanes, passes, waves = v[o], v[o + 1], v[o + 2]
    if waves:
        print(f"{name:9s}: {waves} wave searches, {passes / waves:6.2f} ring passes per wave, "
              f"{lanes / max(passes, 1):5.1f} of 64 lanes active per pass, {lanes / waves / 64:5.2f} ring visits per lane")
print("iterations", len(dx))
