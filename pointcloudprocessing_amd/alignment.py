"""The rotation helpers of the reference's PointcloudAlignment
(PointcloudAlignment.py:229-277).  The IMU dead-reckoning table of that class
is not part of the registration path and is not provided."""
import random

import numpy as np


class PointcloudAlignment:

    @staticmethod
    def rotation(x, y=None, z=None):
        """R = Rx @ Ry @ Rz (PointcloudAlignment.py:229-261); accepts three
        angles or one array of three."""
        if isinstance(x, np.ndarray):
            xr, yr, zr = x[0], x[1], x[2]
        else:
            xr, yr, zr = x, y, z
        cz, sz = np.cos(zr), np.sin(zr)
        cy, sy = np.cos(yr), np.sin(yr)
        cx, sx = np.cos(xr), np.sin(xr)
        R_z = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R_y = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        R_x = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        return R_x @ R_y @ R_z

    @staticmethod
    def randomRotation(scale):
        """:263-270"""
        x = random.uniform(-1, 1) * np.pi * 2 * scale
        y = random.uniform(-1, 1) * np.pi * 2 * scale
        z = random.uniform(-1, 1) * np.pi * 2 * scale
        return PointcloudAlignment.rotation(x, y, z)

    @staticmethod
    def randomTranslation1():
        """:272-277"""
        return np.array([random.uniform(-1, 1), random.uniform(-1, 1), random.uniform(-1, 1)])
