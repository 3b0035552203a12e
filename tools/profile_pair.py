"""Times one resident KITTI-size frame pair through the C ABI (development aid;
run under rocprofv3 --kernel-trace --stats for per-kernel durations)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pointcloudprocessing_amd import MODE_PLANE, MODE_POINT, ComputeShader, PointcloudIcpContainer, lidar  # noqa: E402
from pointcloudprocessing_amd._lib import check, f64p  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    gold = os.path.join(ROOT, "tests", "golden", "kitti_pair.npz")
    if os.path.exists(gold):
        g = np.load(gold)
        a, b = lidar.frame_from_raw(g["raw0"]), lidar.frame_from_raw(g["raw1"])
    else:
        a, b = lidar.synthetic_pair(0)
    cs = ComputeShader()
    icp = PointcloudIcpContainer(cs)
    lib, ctx = cs._lib, cs._ctx
    o = np.zeros(3)
    b64 = np.ascontiguousarray(b, dtype=np.float64)
    for mode, name, iters in ((MODE_POINT, "pt2pt x20", 20), (MODE_PLANE, "pt2pl <=10", 10)):
        if mode == MODE_POINT:
            icp.preparePointToPoint(a, o, b)
        else:
            icp.preparePointToPlane(a, o, b)
        best = 1e9
        for _ in range(reps):
            check(lib.pcp_upload_source(ctx, f64p(b64), len(b64)))
            check(lib.pcp_synchronize(ctx))
            t0 = time.perf_counter()
            check(lib.pcp_timer_start(ctx))
            check(lib.pcp_prepare_resident(ctx, int(mode == MODE_PLANE)))
            check(lib.pcp_icp_launch(ctx, mode, iters, 1e-6, 0))
            ms = C.c_float(0)
            check(lib.pcp_timer_stop(ctx, C.byref(ms)))
            wall = (time.perf_counter() - t0) * 1e3
            best = min(best, ms.value)
        it, cv = C.c_int(0), C.c_int(0)
        check(lib.pcp_icp_fetch(ctx, None, None, C.byref(it), C.byref(cv), None))
        print(f"{name}: {best:.3f} ms (device events), last wall {wall:.3f} ms, iterations {it.value}, converged {cv.value}")
    # stand-alone NN
    icp.prepareNNS(a, o, b)
    for k in range(3):
        t0 = time.perf_counter()
        cs.dispatchNNS(b)
        print(f"dispatchNNS call {k}: {(time.perf_counter() - t0) * 1e3:.3f} ms wall incl. copies")
    icp.preparePointToPoint(a, o, b)
    for k in range(3):
        t0 = time.perf_counter()
        cs.dispatchLS(b)
        print(f"dispatchLS call {k}: {(time.perf_counter() - t0) * 1e3:.3f} ms wall incl. copies")
    cs.cleanup()


if __name__ == "__main__":
    main()
