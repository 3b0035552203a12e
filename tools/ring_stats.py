"""Lane utilisation of the batched ring search (ls_body) from the instrumented library
(make -C pointcloudprocessing_amd/csrc stats):
    PCP_LIB_PATH=pointcloudprocessing_amd/libpcp_stats.so python tools/ring_stats.py [real|moved]
real: the consecutive KITTI sweeps 0 -> 1 of the fixtures; moved: sweep 0 against a moved, noisy copy of itself."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, IcpBatch, PointcloudIcpContainer, lidar, _lib  # noqa: E402
from bench import moved_copy, real_frames  # noqa: E402

lib = _lib.load()
lib.pcp_stats_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
raws, _ = real_frames()
a = lidar.frame_from_raw(raws[0][:, :3])
kind = sys.argv[1] if len(sys.argv) > 1 else "real"
b = lidar.frame_from_raw(raws[1][:, :3]) if kind == "real" else moved_copy(a, 0)[0]
computes = [ComputeShader(maxPointsPerCloud=max(len(a), len(b))) for _ in range(8)]
for cs in computes:
    PointcloudIcpContainer(cs).preparePointToPlane(a, np.zeros(3), b)
batch = IcpBatch(computes)
batch.upload_sources([b] * 8)
out = (C.c_ulonglong * 8)()
for iters in (1, 10):
    batch.prepare(True)
    batch.synchronize()
    lib.pcp_stats_read(out, 1)
    batch.launch(1, iters)
    batch.synchronize()
    lib.pcp_stats_read(out, 1)
    v = list(out)
    print(f"{kind} pair, {iters} iteration(s), 8 pairs in the batch")
    for name, o in (("hinted", 0), ("all rings", 4)):
        lanes, passes, waves = v[o], v[o + 1], v[o + 2]
        if waves:
            print(f"  {name:9s}: {waves} wave searches, {passes / waves:6.2f} ring passes per wave, "
                  f"{lanes / max(passes, 1):5.1f} of 64 lanes active per pass, {lanes / waves / 64:5.2f} ring visits per query")
batch.close()
for cs in computes:
    cs.cleanup()
