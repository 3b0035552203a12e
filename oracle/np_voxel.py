"""CPU restatement of the reference's voxel map path (TEST INFRASTRUCTURE ONLY).

Follows VoxelShader.glsl, VoxelStage.glsl and the host half in Voxelizer.py
(paths relative to /root/reference).  Two properties of the reference are
nondeterministic and are compared as sets by its own tests
(tests/VoxelizerTest.py:325-331): the order of point ids inside a voxel row
and the order of rows in the staging buffer.  One is a data race: a voxel's
neighbours' statistics are read while other threads update them
(VoxelStage.glsl:172-185); this restatement and the HIP kernel both take the
all-reads-before-all-writes outcome (neighbours are read from the statistics
as uploaded).  A neighbour that is not in the table leaves vstatic[i]
uninitialised in the shader (:186-188); it is false here.
"""
import numpy as np

F = np.float32
BIGINT = 9999999


def voxel_coord(points, voxel_size):
    """VoxelShader.glsl:33-36 / Voxelizer.vectorizedGetVoxelCoord (:167-168)."""
    p = np.asarray(points)[:, :3].astype(F)
    return np.ceil(p / F(voxel_size)).astype(np.int32)


def bin_search(vals, value, find_first=True):
    """Voxelizer.bin_search (:550-570) -> (occurrence or -1, insertion point l)."""
    l, h, occ = 0, len(vals) - 1, -1
    while l <= h:
        mid = (l + h) // 2
        if vals[mid] == value:
            occ = mid
            if find_first:
                h = mid - 1
            else:
                l = mid + 1
        elif vals[mid] < value:
            l = mid + 1
        else:
            h = mid - 1
    return occ, l


def bin_search_range(table, axis, value, find_first, lo, hi):
    """VoxelShader.glsl:38-62 (binSearch over voxel_index[lo..hi][axis])."""
    l, h, occ = lo, hi, -1
    while l <= h:
        mid = (l + h) // 2
        v = table[mid][axis]
        if v == value:
            occ = mid
            if find_first:
                h = mid - 1
            else:
                l = mid + 1
        elif v < value:
            l = mid + 1
        else:
            h = mid - 1
    return occ


def find_voxel_id(table, stored, coord):
    """VoxelShader.glsl:64-98 -> table position or -1."""
    if stored == 0:
        return -1
    fx = bin_search_range(table, 0, coord[0], True, 0, stored - 1)
    if fx == -1:
        return -1
    lx = bin_search_range(table, 0, coord[0], False, 0, stored - 1)
    fy = bin_search_range(table, 1, coord[1], True, fx, lx)
    if fy == -1:
        return -1
    ly = bin_search_range(table, 1, coord[1], False, fx, lx)
    return bin_search_range(table, 2, coord[2], True, fy, ly)


def find_voxel_id_host(table, stored, coord):
    """Voxelizer.findVoxelId (:469-512) -> (position or -1, insertion position)."""
    if stored == 0:
        return -1, 0
    x = table[:, 0]
    fx, ins_x = bin_search(x, coord[0], True)
    if fx == -1:
        return -1, ins_x
    lx, _ = bin_search(x, coord[0], False)
    y = table[:, 1][fx:lx + 1]
    fy, ins_y = bin_search(y, coord[1], True)
    if fy == -1:
        return -1, ins_x + ins_y
    ly, _ = bin_search(y, coord[1], False)
    fy, ly = fy + fx, ly + fx
    z = table[:, 2][fy:ly + 1]
    fz, ins_z = bin_search(z, coord[2], True)
    if fz == -1:
        return -1, ins_x + ins_y + ins_z
    lz, _ = bin_search(z, coord[2], False)
    if fz == lz:
        return fz + fy, None
    raise RuntimeError("In this impl. no same values should be present")


def _keys(coords):
    c = np.asarray(coords, dtype=np.int64)
    off = 1 << 20
    return ((c[:, 0] + off) << 42) | ((c[:, 1] + off) << 21) | (c[:, 2] + off)


def lookup(table, stored, coords):
    """find_voxel_id for many coordinates at once (positions in the sorted table, -1 = miss)."""
    if stored == 0:
        return np.full(len(coords), -1, dtype=np.int64)
    tk = _keys(table[:stored, :3])
    ck = _keys(coords)
    pos = np.searchsorted(tk, ck)
    pos_c = np.minimum(pos, stored - 1)
    return np.where(tk[pos_c] == ck, pos_c, -1)


def voxelize_dispatch(points, voxel_index, voxel_data, voxel_size, stored, begin_index, max_points=1024):
    """VoxelShader.glsl:100-146 in thread order (the order the atomics would be
    served in if threads ran one after the other).  Mutates voxel_data.
    Returns (unknown list [count, ids..., zero padding], coords, table positions)."""
    n = len(points)
    unknown_len = 1 + n
    unknown_len += 16 - unknown_len % 16                    # ComputeShader.py:851-853
    unknown = np.zeros(unknown_len, dtype=np.int32)
    coords = voxel_coord(points, voxel_size)
    entry = lookup(voxel_index, stored, coords)
    miss = np.nonzero(entry < 0)[0]
    unknown[0] = len(miss)
    unknown[1:1 + len(miss)] = begin_index + miss
    hit = np.nonzero(entry >= 0)[0]
    rows = voxel_index[entry[hit], 3]
    for i, r in zip(hit, rows):                            # atomicAdd order = thread order
        c = voxel_data[r, 0]
        if c < max_points - 1:
            voxel_data[r, 1 + c] = begin_index + i
        voxel_data[r, 0] = c + 1
    return unknown, coords, entry


class VoxelMap:
    """The state Voxelizer.py keeps (init :48-62, initVoxels :382-388) and its
    host-side growth (storeUnknownPoints :170-306, handleAddingPoints :398-425),
    restated with whole-array operations."""

    def __init__(self, voxel_size, max_points=1024):
        self.voxel_size = F(voxel_size)
        self.max_points = max_points
        self.voxel_index = np.full((4096, 4), BIGINT, dtype=np.int32)
        self.voxel_data = np.zeros((4096, max_points), dtype=np.int32)
        self.voxel_stats = np.zeros((4096, 4), dtype=F)
        self.voxel_stats[:, 0] = 0.5
        self.stored = 0
        self.points = np.zeros((0, 3), dtype=F)

    def add_points(self, pts):
        pts = np.asarray(pts).astype(F)
        begin = len(self.points)
        self.points = np.vstack([self.points, pts[:, :3]])
        unknown, _, _ = voxelize_dispatch(pts, self.voxel_index, self.voxel_data, self.voxel_size,
                                          self.stored, begin, self.max_points)
        self.store_unknown(unknown)
        return unknown

    def store_unknown(self, unknown):
        n = int(unknown[0])
        ids = np.sort(unknown[1:1 + n])
        if n == 0:
            return
        coords = voxel_coord(self.points[ids], self.voxel_size)
        uniq, inverse, counts = np.unique(coords, axis=0, return_inverse=True, return_counts=True)
        inverse = inverse.reshape(-1)
        order = np.argsort(inverse, kind="stable")
        sorted_ids = ids[order]
        new_stored = self.stored + len(uniq)
        while new_stored >= len(self.voxel_index):          # doubling, Voxelizer.py:238-252
            self.voxel_index = np.vstack([self.voxel_index, np.full_like(self.voxel_index, BIGINT)])
            self.voxel_data = np.vstack([self.voxel_data, np.zeros_like(self.voxel_data)])
            grown = np.zeros_like(self.voxel_stats)
            grown[:, 0] = 0.5
            self.voxel_stats = np.vstack([self.voxel_stats, grown])
        if lookup(self.voxel_index, self.stored, uniq).max(initial=-1) >= 0:
            raise RuntimeError("Here all unknown points should have a new voxel created")
        rows = np.arange(self.stored, new_stored, dtype=np.int32)
        new_entries = np.column_stack([uniq.astype(np.int32), rows])
        pos = np.searchsorted(_keys(self.voxel_index[:self.stored, :3]), _keys(uniq))
        merged = np.insert(self.voxel_index[:self.stored], pos, new_entries, axis=0)
        self.voxel_index[:new_stored] = merged
        start = 0
        for r, c in zip(rows, counts):
            self.voxel_data[r, 0] = c
            k = min(c, self.max_points - 1)
            self.voxel_data[r, 1:1 + k] = sorted_ids[start:start + k]
            start += c
        self.stored = new_stored


# ----------------------------------------------------------------------------- VoxelStage.glsl
def bayes(p, alpha, beta):
    """VoxelStage.glsl:37-39, float32."""
    p, alpha, beta = F(p), F(alpha), F(beta)
    with np.errstate(all="ignore"):
        return F(alpha * p) / F(F(alpha * p) + F(beta * F(F(1.0) - p)))


def voxel_stage(voxel_index, voxel_data, voxel_stat, stored, max_points, filter_outliers,
                stage_everything=False, max_staging=1024, stage_buffer=None):
    """VoxelStage.glsl:201-315 in thread (table) order.  voxel_stat is updated in
    place; returns (counter[16], stage_buffer)."""
    counter = np.zeros(16, dtype=np.int32)
    counter[1], counter[2], counter[3] = max_staging, max_points, stored
    counter[4] = 1 if stage_everything else 0
    counter[5] = 0 if filter_outliers else 1
    if stage_buffer is None:
        stage_buffer = np.zeros((max_staging, max_points), dtype=np.int32)
    snap = voxel_stat.copy()                                # neighbours are read as uploaded
    alpha, beta = F(0.55), F(0.45)
    sides = [(0, 1, 0), (0, -1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (0, 0, -1)]

    def recover(vid):
        found, static = [False] * 6, [False] * 6
        for k, s in enumerate(sides):
            e = find_voxel_id(voxel_index, stored, (vid[0] + s[0], vid[1] + s[1], vid[2] + s[2]))
            if e != -1:
                found[k] = True
                row = voxel_index[e][3]
                static[k] = bool(snap[row][3] > F(0.5) or snap[row][2] > F(0.1))
        horizontal = sum(static[2:6])
        return horizontal >= 3 or (horizontal >= 2 and not found[1])

    for pos in range(stored):
        vid = voxel_index[pos]
        idx = int(vid[3])
        if voxel_stat[idx][3] > F(0.5):
            continue
        cnt = int(voxel_data[idx][0])
        mpp = F(voxel_stat[idx][1])
        max_prob = F(mpp - F(int(mpp)))
        prev = int(mpp)
        status = F(voxel_stat[idx][2])                       # stays stale below, as in the shader
        if counter[4] == 1:
            if status > F(0.1):
                voxel_stat[idx][2] = F(0.8)
            elif counter[5] != 1 and recover(vid):
                voxel_stat[idx][2] = F(0.8)
        if status > F(0.5):
            slot = counter[0]
            counter[0] += 1
            if slot < max_staging:
                stage_buffer[slot][0] = cnt
                for i in range(1, min(cnt + 1, max_points)):
                    stage_buffer[slot][i] = voxel_data[idx][i]
                voxel_stat[idx][3] = F(1.0)
            else:
                counter[0] -= 1
            continue
        old = F(voxel_stat[idx][0])
        if cnt == prev:
            new = bayes(old, F(1.0) - alpha, F(1.0) - beta)
        else:
            new = bayes(old, alpha, beta)
        if new > max_prob:
            max_prob = new
        voxel_stat[idx][0] = new
        if max_prob > F(0.999):
            max_prob = F(0.999)
        if max_prob < F(0.001):
            max_prob = F(0.001)
        voxel_stat[idx][1] = F(F(cnt) + max_prob)
        if new > F(0.9) or status > F(0.1) or counter[5] == 1:
            if status < F(0.1):
                voxel_stat[idx][2] = F(0.2)
            if cnt + 1 >= max_points:
                voxel_stat[idx][2] = F(0.8)
    return counter, stage_buffer
