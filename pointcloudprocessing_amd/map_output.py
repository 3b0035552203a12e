"""Map output: the point arena the renderer keeps its clouds in and the PLY files written from it.

GpuMemoryManager has the reference's call surface and behaviour (GpuMemoryManager.py:4-97:
append-only allocation at the end, compaction on every free, integer handles counting up from
124).  save_ply / read_ply_ascii follow PointcloudHandler.savePointcloudPly (:13-57) and
tryReadAsciPly (:60-113): ASCII PLY, `x y z` with six decimals, colours as 0-255 integers.
"""
import numpy as np


class GpuMemoryManager:

    def __init__(self, maxNumberOfPoints=1000000):
        self.maxNumPoints = maxNumberOfPoints
        self.points = np.zeros((maxNumberOfPoints, 6), dtype=np.float32)
        self.allocationIndexes = []
        self.last_buffered_point = 0
        self.pointers = {}
        self.currentPointer = 123

    def markLastBufferPoint(self):
        self.last_buffered_point = self.getMaxPointIndex()

    def getFreeSpace(self, required):
        """:21-39 -- space is only ever taken after the last allocation."""
        self.currentPointer += 1
        prev_end = self.allocationIndexes[-1][1] if self.allocationIndexes else 0
        if self.maxNumPoints - prev_end >= required:
            self.allocationIndexes.append((prev_end, prev_end + required, self.currentPointer))
            return prev_end, prev_end + required, self.currentPointer
        raise MemoryError("Out of available point memory")

    def addPoints(self, np_points):
        if isinstance(np_points, list):
            np_points = np.array(np_points).astype(np.float32)
        start, end, pointer = self.getFreeSpace(len(np_points))
        self.points[start:end] = np_points.astype(np.float32)
        self.pointers[pointer] = start, end
        return pointer

    def freeSpace(self, pointer):
        """:53-66 -- returns whether the compaction moved anything."""
        if pointer not in self.pointers:
            raise ValueError("Invalid range to free.")
        start, end = self.pointers[pointer]
        self.allocationIndexes.remove((start, end, pointer))
        del self.pointers[pointer]
        self.last_buffered_point = 0
        return self.fullDefragment()

    def dropPointers(self):
        self.allocationIndexes = []
        self.pointers.clear()
        self.last_buffered_point = 0

    def getMaxPointIndex(self):
        return self.allocationIndexes[-1][1] if self.allocationIndexes else 0

    def fullDefragment(self):
        """:82-97"""
        prev_end, moved = 0, False
        for i, (start, end, pointer) in enumerate(self.allocationIndexes):
            size = end - start
            if prev_end != start:
                self.points[prev_end:prev_end + size] = self.points[start:end]
                self.allocationIndexes[i] = (prev_end, prev_end + size, pointer)
                self.pointers[pointer] = (prev_end, prev_end + size)
                moved = True
            prev_end += size
        return moved


PLY_HEADER = ("ply", "format ascii 1.0", "element vertex {n}", "property float x", "property float y",
              "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header\n")


def save_ply(path_to_save_to, points6, chunk=1000000):
    """PointcloudHandler.savePointcloudPly (:13-57): rows (x, y, z, r, g, b) with colours in [0, 1]
    -> `<path>.ply`; colours are scaled by 255, clipped and truncated to uint8."""
    pts = np.asarray(points6)
    n = len(pts)
    with open(path_to_save_to + ".ply", "w") as f:
        f.write("\n".join(PLY_HEADER).format(n=n))
        for i in range(0, n, chunk):
            xyz = pts[i:i + chunk, 0:3]
            rgb = (pts[i:i + chunk, 3:6] * 255).clip(0, 255).astype(np.uint8)
            np.savetxt(f, np.hstack((xyz, rgb)), fmt="%.6f %.6f %.6f %d %d %d")
    return path_to_save_to + ".ply"


def save_arena_ply(path_to_save_to, memory_manager):
    """The reference's entry point: compact the arena, write its used part."""
    memory_manager.fullDefragment()
    return save_ply(path_to_save_to, memory_manager.points[:memory_manager.getMaxPointIndex()])


def save_grid_ply(path_to_save_to, grid, colour=(1.0, 1.0, 1.0)):
    """The accumulated map, as it lies on the device in a CellGrid (fed by map_sink.MapSink from the voxels the
    Voxelizer staged), written in the reference's PLY format (savePointcloudPly :13-57) in the order the points
    were fed.  The reference writes its render arena, which holds the same staged points; here the map itself is
    the source."""
    ids, xyz = grid.sorted_points()
    pts = np.empty((len(ids), 6), dtype=np.float64)
    pts[ids, :3] = xyz                                  # back into feeding order
    pts[:, 3:] = np.asarray(colour, dtype=np.float64)
    return save_ply(path_to_save_to, pts)


def read_ply_ascii(path):
    """PointcloudHandler.tryReadAsciPly (:60-113): (N, 6) float64, colours back in [0, 1]; None when the
    file is not text."""
    try:
        with open(path, "r") as f:
            num_points, header_lines = None, 0
            for line in f:
                header_lines += 1
                line = line.strip()
                if line.startswith("element vertex"):
                    num_points = int(line.split()[2])
                if line == "end_header":
                    break
            if num_points is None:
                raise ValueError("PLY header is missing 'element vertex' line")
            data = np.loadtxt(f, ndmin=2, max_rows=num_points) if num_points else np.zeros((0, 6))
    except UnicodeDecodeError:
        return None
    points = np.empty((num_points, 6))
    points[:] = data[:, :6]
    points[:, 3:] = points[:, 3:].clip(0, 255).astype(np.float32) / 255.0
    return points


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_binary(path):
    """PointcloudHandler.tryReadBinPly (:114-135) without the plyfile package: the first element of a
    binary_little_endian / binary_big_endian PLY, (N, 6) float32 rows (x, y, z, r, g, b) with colours
    scaled by 1/255 when the element has red/green/blue, zeros otherwise."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file: " + path)
        order, count, fields, element = None, None, [], 0
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            words = line.decode("ascii").split()
            if not words:
                continue
            if words[0] == "format":
                if words[1] not in ("binary_little_endian", "binary_big_endian"):
                    raise ValueError("not a binary PLY (format " + words[1] + ")")
                order = "<" if words[1] == "binary_little_endian" else ">"
            elif words[0] == "element":
                element += 1
                if element == 1:
                    count = int(words[2])
            elif words[0] == "property" and element == 1:
                if words[1] == "list":
                    raise ValueError("list properties in the first PLY element are not supported")
                if len(words) < 3 or words[1] not in _PLY_TYPES:
                    raise ValueError("malformed PLY property line: " + " ".join(words))
                fields.append((words[2], _PLY_TYPES[words[1]]))
            elif words[0] == "end_header":
                break
        if order is None or count is None:
            raise ValueError("PLY header is missing its format or element line")
        vertex = np.fromfile(f, dtype=np.dtype([(n, order + t) for n, t in fields]), count=count)
    if len(vertex) != count:
        raise ValueError("PLY file ends before its %d vertices" % count)
    xyz = np.stack([vertex["x"], vertex["y"], vertex["z"]], axis=-1)
    if "red" in vertex.dtype.names:
        rgb = np.stack([vertex["red"], vertex["green"], vertex["blue"]], axis=-1).astype(np.float32) / 255.0
    else:
        rgb = np.zeros_like(xyz)
    return np.hstack((xyz, rgb)).astype(np.float32)


def load_pointcloud(path):
    """PointcloudHandler.loadPointcloud (:152-166): a text PLY if the file reads as one, else a binary PLY."""
    points = read_ply_ascii(path)
    return points if points is not None else read_ply_binary(path)
