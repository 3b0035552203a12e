"""cProfile of one odometry frame (reference cloud kept prepared): python tools/prof_env.py"""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader, PointcloudIcpContainer, Voxelizer, lidar
from pointcloudprocessing_amd.environment import EnvironmentConstructor
from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
pair = np.load(os.path.join(GOLD, "kitti_pair.npz")); imu = np.load(os.path.join(GOLD, "ref_imu.npz"))
rd = OxtsDataReader()
for k in range(2):
    rd.add_record(f"{k:010d}", imu["oxts_values"][k], parse_timestamp(str(imu["oxts_stamps"][k])))
rd.finish()
frames = [lidar.frame_from_raw(pair["raw0"]), lidar.frame_from_raw(pair["raw1"])]
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
icp = PointcloudIcpContainer(cs); vox = Voxelizer(cs, None)
env = EnvironmentConstructor(None, rd, None, icp, cs, voxelizer=vox)
env.framekeys = rd.getfilenames(); env.frameindex = env.local_frame_counter = env.total_frame_counter = 0
env.prev_poses = []; vox.init(0.5)
z0, z1 = np.zeros((len(frames[0]), 3)), np.zeros((len(frames[1]), 3))
env.calculateTransition_imu((frames[0], None, z0), rd.getOx(env.framekeys[0]), iterations=10)
first = env.prev_poses[0]
icp.keep_prepared_target = True
pr = cProfile.Profile()
for r in range(23):
    env.prev_poses = [first]; env.total_frame_counter, env.local_frame_counter = 1, 0
    if r == 3: pr.enable()
    env.calculateTransition_imu((frames[1], None, z1), rd.getOx(env.framekeys[1]), iterations=10)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
