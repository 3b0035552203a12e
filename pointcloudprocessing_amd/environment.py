"""Frame-by-frame odometry driver with the reference's call surface
(EnvironmentConstructor.py:8-270): IMU prediction from the OXTS record, refinement by
full_pt2pl / full_pt2pt against a reference cloud that is replaced every
`reset_origin_treshold` frames, pose bookkeeping in prev_poses, voxel map feed.
Rendering calls are skipped when renderer is None.
"""
import numpy as np

from .voxelizer import Voxelizer


class EnvironmentConstructor:

    def __init__(self, renderer, oxtsdatareader, lidardatareader, icpcontainer, computeShader, voxelizer=None):
        self.prev_position = None
        self.renderer = renderer
        self.oxts = oxtsdatareader
        self.lidar = lidardatareader
        self.icp = icpcontainer
        self.compute = computeShader
        self.voxelizer = voxelizer if voxelizer is not None else Voxelizer(self.compute, self.renderer)
        self.framekeys = None
        self.frameindex = None
        self.reference_points = None
        self.prev_origin = None
        self.local_frame_counter = None
        self.total_frame_counter = None
        self.reset_origin_treshold = 10
        self.added_path_points = None
        self.prev_poses = None
        self.inited = False

    def init(self, voxel_size):
        self.inited = False
        self.framekeys = self.lidar.getfilenames()
        self.frameindex = 0
        self.local_frame_counter = 0
        self.total_frame_counter = 0
        self.prev_poses = []
        self.added_path_points = None
        self.voxelizer.init(voxel_size)
        self.reference_points = None
        self.prev_origin = None
        # the reference cloud is a new array every reset_origin_treshold frames and never changed in
        # place here, so the container may keep it prepared on the device between frames
        if hasattr(self.icp, "keep_prepared_target"):
            self.icp.keep_prepared_target = True
        self.inited = True

    def cleanup(self):
        self.inited = False
        self.frameindex = 0
        self.total_frame_counter = 0
        self.local_frame_counter = 0
        self.prev_poses = []
        self.framekeys = []
        self.reference_points = None
        self.prev_origin = None
        self.voxelizer.cleanup()

    def getNextFrameData(self, offset):
        """:77-87"""
        key = self.framekeys[self.frameindex + offset]
        lidar_points = self.lidar.getPointsWithIntensities(key)
        oxts = self.oxts.getOx(key)
        self.frameindex += 1
        if self.frameindex == len(self.framekeys):
            self.frameindex = 0
        return lidar_points, oxts

    def getDeltaVelocity(self, current_velocity, current_rotation, current_time, prev_time):
        """:90-95"""
        delta_time = (current_time - prev_time).total_seconds()
        current_velocity *= delta_time
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

    def calculateTransition_imu(self, current_lidar, current_oxts, point_to_plane=True, debug=False, iterations=20,
                                separate_colors=False, removeOutliers=False, pure_imu=False):
        """:154-265"""
        points, colors, intensities = current_lidar
        if not separate_colors:
            colors = intensities
        current_velocity, current_horz_rot, current_vert_rot, current_time = self.getCurrentOxtsData(current_oxts)
        current_horz_rot = np.linalg.inv(current_horz_rot)
        current_rotation = current_vert_rot @ current_horz_rot

        if self.total_frame_counter == 0:
            self.reference_points = (current_rotation @ points.T).T
            self.prev_origin = np.array([0, 0, 0])
            self.prev_position = np.array([0, 0, 0])
            if self.renderer is not None:
                self.renderer.addPoints([self.prev_position], np.array([255, 0, 0]))
            zero = np.array([0.0, 0.0, 0.0])
            self.prev_poses.append((zero, zero.copy(), current_rotation, current_rotation, current_time))
            self.voxelizer.separate_colors = separate_colors
            self.voxelizer.filter_outliers = removeOutliers
            self.total_frame_counter += 1
            self.voxelizer.addPoints(self.reference_points, colors)
            return

        p_imu_pos, p_refined_pos, _, _, p_time = self.prev_poses[self.total_frame_counter - 1]
        delta_velocity = self.getDeltaVelocity(current_velocity=current_velocity, current_time=current_time,
                                               current_rotation=current_rotation, prev_time=p_time)
        estimated_position = p_refined_pos + delta_velocity
        next_pure_imu_position = p_imu_pos + delta_velocity
        aligned_points = (current_rotation @ points.T + estimated_position[:, None]).T     # == (R @ p.T).T + position

        if not pure_imu:
            full_iteration = False
            iterations_to_do = iterations
            if self.local_frame_counter + 1 >= self.reset_origin_treshold:
                full_iteration = True
                iterations_to_do = 15
            t_opt, R_opt = self.getRefinedTransition(reference_points=self.reference_points,
                                                     reference_origin=self.prev_origin,
                                                     points_to_refine=aligned_points, iterations=iterations_to_do,
                                                     point_to_plane=point_to_plane, renderer=self.renderer,
                                                     debug=debug, full_iter=full_iteration)
            refined_points = self.applyOptimalTranslation(r_opt=R_opt, t_opt=t_opt, points=aligned_points)
            refined_position = self.getRefinedPosition(r_opt=R_opt, t_opt=t_opt, estimated_position=estimated_position)
            refined_rotation = R_opt @ current_rotation
        else:
            refined_points = aligned_points
            refined_position = estimated_position
            refined_rotation = current_rotation

        self.voxelizer.addPoints(refined_points, colors)
        self.local_frame_counter += 1
        if self.local_frame_counter >= self.reset_origin_treshold:
            self.reference_points = refined_points
            self.prev_origin = refined_position
            self.local_frame_counter = 0
        self.prev_poses.append((next_pure_imu_position, refined_position, current_rotation, refined_rotation,
                                current_time))
        self.total_frame_counter += 1
