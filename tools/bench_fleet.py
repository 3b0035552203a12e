"""Several drives at once on one GPU: D device chains (pcp_chain_*), each with its own voxel map, fed by D host threads.
A single drive is a chain of dependent single-pair launches that fills a fraction of the GPU; this measures what the rest
of it is worth.    python tools/bench_fleet.py [drives ...]      (default: 1 2 4)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pointcloudprocessing_amd import ComputeShader, PinnedFrames, Voxelizer  # noqa: E402
from pointcloudprocessing_amd.chain import OdometryChain, imu_rows  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

raws, _ = bench.real_frames()
g = np.load(os.path.join(bench.GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
    rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
rd.finish()
rows = imu_rows([rd.getOx(k) for k in rd.getfilenames()[:len(raws)]])[0]
pinned = PinnedFrames(raws)
reps = 4


def drive(chain, cs, out, barrier):
    ends = []
    for rep in range(reps + 1):
        chain.reset()
        vox = Voxelizer(cs, None)
        vox.separate_colors, vox.filter_outliers = False, False
        vox.init(0.5)
        cs.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        track = chain.collect(chain.submit(pinned, rows, voxelizer=vox))
        ends.append((t0, time.perf_counter(), np.array([t[1] for t in track]), vox.stored_voxels))
        vox.cleanup()
    out.append(ends)


reference = None
for drives in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    css = [ComputeShader(maxPointsPerCloud=200000, maxScans=100) for _ in range(drives)]
    chains = [OdometryChain(max_frames=len(raws), max_points=1 << 17) for _ in range(drives)]
    out, barrier = [], threading.Barrier(drives)
    threads = [threading.Thread(target=drive, args=(chains[d], css[d], out, barrier)) for d in range(drives)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    rates = []
    for rep in range(1, reps + 1):                       # rep 0 warms up
        start = min(e[rep][0] for e in out)
        end = max(e[rep][1] for e in out)
        rates.append(drives * len(raws) / (end - start))
    positions = out[0][-1][2]
    if reference is None:
        reference = positions
    same = all(np.array_equal(e[-1][2], reference) for e in out)
    print("%d drive%s at once: %.0f frames/s in all (best %.0f), %.3f ms per frame and drive; every drive ends where a single one does: %s"
          % (drives, "" if drives == 1 else "s", float(np.median(rates)), max(rates), 1e3 * drives / float(np.median(rates)), same),
          flush=True)
    for c in chains:
        c.close()
    for c in css:
        c.cleanup()
pinned.close()
