"""Stepwise dispatches of the reference's call surface on the fixture's KITTI pair (host buffers in, results out):
    python tools/latency_stepwise.py"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from pointcloudprocessing_amd import ComputeShader
from pointcloudprocessing_amd.icp_container import PointcloudIcpContainer
from tests._data import kitti_pair
a, b = kitti_pair()
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
icp = PointcloudIcpContainer(cs)
o = np.zeros(3)
lines = icp.getScanLines(a, o)
def T(f, n=10):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
def first_and_later(prepare, dispatch):
    first, later = [], []
    for _ in range(6):
        prepare()
        t0 = time.perf_counter(); dispatch(); first.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dispatch(); dispatch(); dispatch(); later.append((time.perf_counter() - t0) / 3)
    return np.median(first[1:]) * 1e3, np.median(later[1:]) * 1e3
print("dispatchNNS        first %.3f ms, later %.3f ms" % first_and_later(lambda: cs.prepareNNS(a, lines, b, o), lambda: cs.dispatchNNS(b)))
print("dispatchLS         first %.3f ms, later %.3f ms" % first_and_later(lambda: cs.prepareLS(a, lines, b, o), lambda: cs.dispatchLS(b)))
print("dispatchPointPlane first %.3f ms, later %.3f ms" % first_and_later(lambda: cs.preparePointPlane(a, lines, b, o), lambda: cs.dispatchPointPlane(b)))
cs.cleanup()
