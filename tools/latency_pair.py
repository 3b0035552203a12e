"""Where the time of ONE registration goes (KITTI fixture pair, point-to-plane, target kept prepared):
    python tools/latency_pair.py"""
import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.')
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, MODE_PLANE, lidar
from pointcloudprocessing_amd._lib import check, f64p, f32p
from tests._data import kitti_pair
a, b = kitti_pair()
cs = ComputeShader(maxPointsPerCloud=130000, maxScans=100)
icp = PointcloudIcpContainer(cs)
icp.keep_prepared_target = True
o = np.zeros(3)
for _ in range(3): icp.full_pt2pl(a, b, o, iterations=10)
def T(f, n=20):
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
print("full_pt2pl (target kept): %.3f ms" % T(lambda: icp.full_pt2pl(a, b, o, iterations=10))[0])
print("  ascontiguousarray f64:  %.3f ms" % T(lambda: np.ascontiguousarray(np.asarray(b)[:, :3], dtype=np.float64))[0])
ms, r = T(lambda: cs.icpRun(b, MODE_PLANE, 10)); dx = r[0]
print("  icpRun:                 %.3f ms (%d iterations)" % (ms, len(dx)))
b64 = np.ascontiguousarray(b, dtype=np.float64)
lib, ctx = cs._lib, cs._ctx
def up(): check(lib.pcp_upload_source(ctx, f64p(b64), len(b64))); check(lib.pcp_synchronize(ctx))
print("  upload 3 MB + sync:     %.3f ms" % T(up)[0])
def run():
    check(lib.pcp_reset_hints(ctx)); check(lib.pcp_icp_launch(ctx, MODE_PLANE, 10, 1e-6, 0)); check(lib.pcp_synchronize(ctx))
print("  launch loop + sync:     %.3f ms" % T(run)[0])
def fetch():
    d = np.zeros((10, 6), np.float32); it, cv = C.c_int(0), C.c_int(0)
    check(lib.pcp_icp_fetch(ctx, f32p(d), None, C.byref(it), C.byref(cv), None))
print("  fetch:                  %.3f ms" % T(fetch)[0])
print("  replay (host):          %.3f ms" % T(lambda: icp._replay(dx, len(dx) - 1))[0])
cs.cleanup()
