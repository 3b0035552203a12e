import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, lidar
from pointcloudprocessing_amd._lib import check, f32p
g = np.load("tests/golden/kitti_pair.npz")
a, b = lidar.frame_from_raw(g["raw0"]), lidar.frame_from_raw(g["raw1"])
cs = ComputeShader(); icp = PointcloudIcpContainer(cs)
icp.prepareNNS(a, np.zeros(3), b)
st = np.zeros((100, 4), np.float32)
check(cs._lib.pcp_ring_stats(cs._ctx, f32p(st)))
print(st[:5]); print(st[60:68]); print(np.isnan(st).any(1).sum())
