"""Per-frame cost of the odometry driver (EnvironmentConstructor.calculateTransition_imu) on the two
KITTI frames of the fixture with their OXTS records: IMU prediction, point-to-plane refinement against
the reference cloud, voxel map feed.  Frame 1 is fed repeatedly (the map keeps growing).
    python tools/bench_env.py [repeats]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, Voxelizer, lidar  # noqa: E402
from pointcloudprocessing_amd.environment import EnvironmentConstructor  # noqa: E402
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pair = np.load(os.path.join(GOLD, "kitti_pair.npz"))
imu = np.load(os.path.join(GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k in range(2):
    rd.add_record(f"{k:010d}", imu["oxts_values"][k], parse_timestamp(str(imu["oxts_stamps"][k])))
rd.finish()
frames = [lidar.frame_from_raw(pair["raw0"]), lidar.frame_from_raw(pair["raw1"])]
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
icp = PointcloudIcpContainer(cs)
vox = Voxelizer(cs, None)
env = EnvironmentConstructor(None, rd, None, icp, cs, voxelizer=vox)
env.framekeys = rd.getfilenames()
env.frameindex = env.local_frame_counter = env.total_frame_counter = 0
env.prev_poses = []
vox.init(0.5)


class Clock:
    """Wraps a bound method and accumulates its wall time."""
    def __init__(self, obj, name):
        self.f, self.t, self.n = getattr(obj, name), 0.0, 0
        setattr(obj, name, self)

    def __call__(self, *a, **k):
        t0 = time.perf_counter()
        r = self.f(*a, **k)
        self.t += time.perf_counter() - t0
        self.n += 1
        return r


zeros = np.zeros((len(frames[0]), 3))
env.calculateTransition_imu((frames[0], None, zeros), rd.getOx(env.framekeys[0]), iterations=10)
first = env.prev_poses[0]
clocks = {"refine (full_pt2pl)": Clock(env, "getRefinedTransition"), "  scan lines": Clock(icp, "getScanLines"),
          "  prepare": Clock(icp, "preparePointToPlane"), "voxel feed": Clock(vox, "addPoints")}
zeros1 = np.zeros((len(frames[1]), 3))
for keep in (False, True):          # True is what EnvironmentConstructor.init() switches on
    icp.keep_prepared_target = keep
    t_all = 0.0
    for r in range(reps + 1):
        env.prev_poses = [first]
        env.total_frame_counter, env.local_frame_counter = 1, 0
        if r == 1:                      # the first repeat warms up
            for c in clocks.values():
                c.t, c.n = 0.0, 0
            t_all = 0.0
        t0 = time.perf_counter()
        env.calculateTransition_imu((frames[1], None, zeros1), rd.getOx(env.framekeys[1]), iterations=10)
        t_all += time.perf_counter() - t0
    print(f"frame, reference cloud {'kept prepared on the device' if keep else 'prepared every frame'}: "
          f"{t_all / reps * 1e3:.2f} ms  ({reps} repeats; {len(frames[1])} points)")
    for k, c in clocks.items():
        print(f"  {k:22s} {c.t / reps * 1e3:7.2f} ms")
pos = env.prev_poses[-1][1]
print("refined position of frame 1:", np.round(pos, 4), " imu-only:", np.round(env.prev_poses[-1][0], 4))
vox.cleanup()
cs.cleanup()
