"""Frame-by-frame odometry driver with the reference's call surface
(EnvironmentConstructor.py:8-270): IMU prediction from the OXTS record, refinement by
full_pt2pl / full_pt2pt against a reference cloud that is replaced every
`reset_origin_treshold` frames, pose bookkeeping in prev_poses, voxel map feed.
Rendering calls are skipped when renderer is None.
"""
import numpy as np

from .voxelizer import Voxelizer


class EnvironmentConstructor:
    """State: `prev_poses` holds one tuple per frame -- (IMU-only position, refined position, IMU rotation,
    refined rotation, time stamp); `reference_points` / `prev_origin` are the cloud and sensor position the
    next frames are registered against."""

    reset_origin_treshold = 10          # frames between two replacements of the reference cloud (:27)

    def __init__(self, renderer, oxtsdatareader, lidardatareader, icpcontainer, computeShader, voxelizer=None):
        self.renderer, self.oxts, self.lidar = renderer, oxtsdatareader, lidardatareader
        self.icp, self.compute = icpcontainer, computeShader
        self.voxelizer = Voxelizer(computeShader, renderer) if voxelizer is None else voxelizer
        self.prev_position = self.added_path_points = None
        self.inited = False
        self._reset(framekeys=None, counters=None, poses=None)

    def _reset(self, framekeys, counters, poses):
        self.framekeys = framekeys
        self.frameindex = self.local_frame_counter = self.total_frame_counter = counters
        self.prev_poses = poses
        self.reference_points = self.prev_origin = None

    def init(self, voxel_size):
        self.inited = False
        self._reset(framekeys=self.lidar.getfilenames(), counters=0, poses=[])
        self.added_path_points = None
        self.voxelizer.init(voxel_size)
        # the reference cloud is a new array every reset_origin_treshold frames and never changed in
        # place here, so the container may keep it prepared on the device between frames
        if hasattr(self.icp, "keep_prepared_target"):
            self.icp.keep_prepared_target = True
        self.inited = True

    def cleanup(self):
        self.inited = False
        self._reset(framekeys=[], counters=0, poses=[])
        self.voxelizer.cleanup()

    # ------------------------------------------------------------------ the reference's helpers (:77-132)
    def getNextFrameData(self, offset):
        """:77-87 -- (lidar tuple, OXTS record) of the next frame; the frame index wraps at the end."""
        key = self.framekeys[self.frameindex + offset]
        self.frameindex = (self.frameindex + 1) % len(self.framekeys)
        return self.lidar.getPointsWithIntensities(key), self.oxts.getOx(key)

    def getDeltaVelocity(self, current_velocity, current_rotation, current_time, prev_time):
        """:90-95 -- displacement since the previous frame; scales the velocity argument in place, as there."""
        current_velocity *= (current_time - prev_time).total_seconds()
        return (current_rotation @ current_velocity.T).T

    def getRefinedPosition(self, r_opt, t_opt, estimated_position):
        return (r_opt @ estimated_position.T).T - t_opt

    def applyOptimalTranslation(self, r_opt, t_opt, points):
        """:101-102 -- (r_opt @ points.T).T - t_opt; the offset is taken from the (3, n) product before the
        transposition (the same subtraction element by element, on contiguous rows)."""
        return (r_opt @ points.T - np.asarray(t_opt)[:, None]).T

    def getCurrentOxtsData(self, current_oxts):
        """:104-105"""
        return (-current_oxts.getVelocity(), current_oxts.getTrueRotation(np.eye(3)),
                current_oxts.getYawRotation(np.eye(3)), current_oxts.getTime())

    def getRefinedTransition(self, reference_points, reference_origin, points_to_refine, iterations, point_to_plane,
                             renderer=None, debug=False, full_iter=False):
        """:116-132"""
        if point_to_plane:
            return self.icp.full_pt2pl(reference_points, points_to_refine, reference_origin, iterations=iterations,
                                       full_iter=full_iter)
        return self.icp.full_pt2pt(reference_points, points_to_refine, reference_origin, iterations=iterations)

    # ------------------------------------------------------------------ one frame (:154-265)
    def _start(self, cloud, colours, imu_rotation, stamp, separate_colors, removeOutliers):
        """Frame 0 defines the map frame: its cloud, turned by the IMU attitude, is the first reference."""
        self.reference_points = (imu_rotation @ cloud.T).T
        self.prev_origin = np.array([0, 0, 0])
        self.prev_position = np.array([0, 0, 0])
        if self.renderer is not None:
            self.renderer.addPoints([self.prev_position], np.array([255, 0, 0]))
        self.prev_poses.append((np.zeros(3), np.zeros(3), imu_rotation, imu_rotation, stamp))
        self.voxelizer.separate_colors, self.voxelizer.filter_outliers = separate_colors, removeOutliers
        self.total_frame_counter += 1
        self.voxelizer.addPoints(self.reference_points, colours)

    def calculateTransition_imu(self, current_lidar, current_oxts, point_to_plane=True, debug=False, iterations=20,
                                separate_colors=False, removeOutliers=False, pure_imu=False):
        """:154-265 -- IMU prediction of the frame's pose, refinement against the reference cloud, map feed."""
        cloud, colours, intensities = current_lidar
        colours = colours if separate_colors else intensities
        velocity, yaw_rot, attitude_rot, stamp = self.getCurrentOxtsData(current_oxts)
        imu_rotation = attitude_rot @ np.linalg.inv(yaw_rot)
        if self.total_frame_counter == 0:
            self._start(cloud, colours, imu_rotation, stamp, separate_colors, removeOutliers)
            return

        last_imu_position, last_position, _, _, last_stamp = self.prev_poses[self.total_frame_counter - 1]
        step = self.getDeltaVelocity(current_velocity=velocity, current_rotation=imu_rotation, current_time=stamp,
                                     prev_time=last_stamp)
        guess = last_position + step
        placed = (imu_rotation @ cloud.T + guess[:, None]).T            # == (R @ p.T).T + position
        position, rotation = guess, imu_rotation
        if not pure_imu:
            # the last frame before the reference cloud is replaced runs 15 full iterations (:198-203)
            closing = self.local_frame_counter + 1 >= self.reset_origin_treshold
            t_opt, R_opt = self.getRefinedTransition(
                reference_points=self.reference_points, reference_origin=self.prev_origin, points_to_refine=placed,
                iterations=15 if closing else iterations, point_to_plane=point_to_plane, renderer=self.renderer,
                debug=debug, full_iter=closing)
            position = self.getRefinedPosition(r_opt=R_opt, t_opt=t_opt, estimated_position=guess)
            placed = self.applyOptimalTranslation(r_opt=R_opt, t_opt=t_opt, points=placed)
            rotation = R_opt @ imu_rotation

        self.voxelizer.addPoints(placed, colours)
        self.local_frame_counter += 1
        if self.local_frame_counter >= self.reset_origin_treshold:
            self.reference_points, self.prev_origin, self.local_frame_counter = placed, position, 0
        self.prev_poses.append((last_imu_position + step, position, imu_rotation, rotation, stamp))
        self.total_frame_counter += 1
