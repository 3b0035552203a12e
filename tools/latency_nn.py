"""Where a dispatchNNS call spends its time (real pair 0 -> 1): for a transposed float64 cloud (what the reader hands out),
a C-contiguous float64 one, float32 (n, 3) and padded float32 (n, 4); all-rings call after a prepare and hinted calls."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, lidar  # noqa: E402
import ctypes as C  # noqa: E402
from pointcloudprocessing_amd._lib import check  # noqa: E402

raws, _ = bench.real_frames()
a, b = lidar.frame_from_raw(raws[0][:, :3]), lidar.frame_from_raw(raws[1][:, :3])
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
icp = PointcloudIcpContainer(cs)
ms = C.c_float(0)
variants = {"float64 transposed (reader's)": b, "float64 C-contiguous": np.ascontiguousarray(b),
            "float32 (n, 3)": np.ascontiguousarray(b, dtype=np.float32), "float32 (n, 4) padded": cs.pad_points(b)}
for name, pts in variants.items():
    first, later, kern = [], [], []
    for _ in range(7):
        icp.prepareNNS(a, np.zeros(3), pts)
        c0 = time.perf_counter(); cs.dispatchNNS(pts); first.append(time.perf_counter() - c0)
        for _ in range(3):
            c0 = time.perf_counter(); cs.dispatchNNS(pts); later.append(time.perf_counter() - c0)
            check(cs._lib.pcp_nn_kernel_ms(cs._ctx, C.byref(ms))); kern.append(ms.value)
    print("%-32s all-rings call %.3f ms, hinted call %.3f ms (kernel %.3f ms)" % (name, np.median(first) * 1e3, np.median(later) * 1e3, np.median(kern)))
for name, pts in variants.items():
    icp.preparePointToPlane(a, np.zeros(3), pts)
    t = []
    for _ in range(9):
        c0 = time.perf_counter(); cs.dispatchPointPlane(pts); t.append(time.perf_counter() - c0)
    print("%-32s dispatchPointPlane %.3f ms" % (name, np.median(t[2:]) * 1e3))
cs.cleanup()
