"""KITTI OXTS (GPS/IMU) records with the reference's call surface: class OXTS
(OXTS.py:4-121) and class OxtsDataReader (OxtsDataReader.py:10-203).  Files are
read in sorted name order (the reference iterates os.listdir, whose order is
file-system dependent)."""
import math
import os
from datetime import datetime

import numpy as np

FIELDS = ("lat", "lon", "alt", "roll", "pitch", "yaw", "velNorth", "velEast", "velForward", "velLeft", "velUp",
          "accx", "accy", "accz", "accForward", "accLeft", "accUp", "angx", "angy", "angz", "angForward",
          "angLeft", "angUp", "posAcc", "velAcc", "navStat", "numSats", "posMode", "velMode", "oriMode")


class OXTS:
    def __init__(self, values, time):
        """values: the 30 numbers of one oxts/data/*.txt line, in file order (OxtsDataReader.py:98-130)."""
        values = np.asarray(values, dtype=np.float64)
        if values.shape != (30,):
            raise ValueError("an OXTS record holds 30 values")
        for name, v in zip(FIELDS, values):
            setattr(self, name, v)
        self.time = time

    def getWorldPos(self):
        return np.array([self.lon, self.lat, self.alt])

    def getVelocityNE(self):
        return np.array([self.velNorth, 0, self.velEast])

    def getVelocity(self):
        """:76-77"""
        return np.array([self.velLeft, self.velUp, self.velForward])

    def getAcceleration(self):
        return np.array([self.accLeft, 0, self.accForward]) - np.array([0, 9.80665, 0])

    def getYawRotation(self, originMatrix):
        """:84-85"""
        return originMatrix @ OXTS.rotation(0, self.yaw, 0)

    def getTrueRotation(self, originMatrix):
        """:87-88"""
        return originMatrix @ OXTS.rotation(self.pitch, 0, self.roll)

    def getTime(self):
        return self.time

    @staticmethod
    def rotation(x, y, z):
        """:96-121: Rx @ Ry @ Rz."""
        cz, sz, cy, sy, cx, sx = np.cos(z), np.sin(z), np.cos(y), np.sin(y), np.cos(x), np.sin(x)
        R_z = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R_y = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        R_x = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        return R_x @ R_y @ R_z


def parse_timestamp(line):
    """OxtsDataReader.readTimestamps (:78-91): nanosecond stamps cut to microseconds."""
    head, frac = line.strip().split(".")
    return datetime.strptime(head + "." + frac[:6], "%Y-%m-%d %H:%M:%S.%f")


class OxtsDataReader:
    OXTS_DATA = os.path.join("oxts", "data")
    TIMESTAMPS = os.path.join("oxts", "timestamps.txt")

    def __init__(self):
        self.inited = False
        self.root = None
        self.currentName = 0
        self.datamap, self.offsetmap, self.rotationMap = {}, {}, {}

    def getRoot(self, path):
        files = os.listdir(path)
        return self.getRoot(os.path.join(path, files[0])) if len(files) == 1 else path

    def init(self, path):
        self.root = self.getRoot(path)
        with open(os.path.join(self.root, self.TIMESTAMPS)) as f:
            timestamps = [parse_timestamp(l) for l in f.read().splitlines() if l.strip()]
        names = sorted(n for n in os.listdir(os.path.join(self.root, self.OXTS_DATA)) if n.endswith(".txt"))
        for name, stamp in zip(names, timestamps):
            with open(os.path.join(self.root, self.OXTS_DATA, name)) as f:
                self.add_record(name[:-4], f.readline().split(), stamp)
        if len(self.datamap) != len(timestamps):
            raise RuntimeError(f"Timestamps and frames do not match: keys[{len(self.datamap)}] timestamps[{len(timestamps)}]")
        self.finish()

    def add_record(self, key, values, timestamp):
        self.datamap[key] = OXTS(np.asarray(values, dtype=np.float64), timestamp)

    def finish(self):
        """:56-62: offsets and rotations relative to the first record."""
        first = self.datamap[next(iter(self.datamap))]
        gps0, rot0 = self.getCartesian(first), self.readRotation(first)
        for k, ox in self.datamap.items():
            self.offsetmap[k] = self.getCartesian(ox) - gps0
            self.rotationMap[k] = self.readRotation(ox) - rot0
        self.inited = True

    def cleanup(self):
        self.inited = False
        self.datamap.clear()
        self.offsetmap.clear()
        self.rotationMap.clear()

    def getCartesian(self, oxts):
        """:139-154"""
        R = 6371000
        lon, lat, _ = oxts.getWorldPos()
        lat, lon = math.radians(lat), math.radians(lon)
        return np.mod(np.array([R * math.cos(lat) * math.cos(lon), R * math.cos(lat) * math.sin(lon), R * math.sin(lat)]),
                      10000)

    def readRotation(self, oxts):
        return np.array([oxts.roll, oxts.pitch, oxts.yaw])

    def getOxtsPosNp(self):
        return np.array([self.offsetmap[k] for k in self.datamap])

    def getRotation(self, key):
        return self.rotationMap[key]

    def getOx(self, filename):
        return self.datamap[filename]

    def getfilenames(self):
        return list(self.datamap.keys())

    def getOriginOx(self):
        return self.datamap[next(iter(self.datamap))]

    def getWorldOrigin(self):
        return self.getOriginOx().getWorldPos()

    def getCurrentName(self):
        return self.getfilenames()[self.currentName]

    def getCurrentOx(self):
        return self.getOx(self.getCurrentName())

    def peekNextName(self):
        names = self.getfilenames()
        return names[self.currentName + 1] if self.currentName < len(names) - 1 else "UNDEFINED"

    def getNextName(self):
        names = self.getfilenames()
        ret = names[self.currentName]
        self.currentName = (self.currentName + 1) % len(names)
        return ret
