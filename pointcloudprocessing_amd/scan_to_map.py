"""Scan-to-map registration with the map split over the GPUs of a node.

The map is binned into cubic cells; the Morton key space of the cells is cut into
one contiguous range per rank (balanced by map points).  Rank r keeps the map
points whose cell, or any of its 26 neighbour cells, lies in its range (a one-cell
halo), so for every source point whose own cell key falls into the range the
exact neighbour within `radius <= cell` is present on that rank.  Per iteration
each rank sums the 8x8 least-squares tile over the source points it owns, the
tiles are all-reduced (the ONE exchange step of the path: 256 bytes), and every
rank solves the same 6x6 system and moves its copy of the scan, so no second
exchange is needed.

Results equal the single-rank registration up to float32 summation order (the
partial tiles are added by the all-reduce instead of the fixed tree).

The reference registers a scan against one frame only (EnvironmentConstructor.py:
116-132); the tile arithmetic, the solve and the pose update are the reference's
(LeastSquaresPoint/Plane.glsl, PointcloudIcpContainer.py:24-29, :47-69, :172-174).
"""
import numpy as np

from .compute_shader import MODE_PLANE, MODE_POINT  # noqa: F401  (re-exported for callers)

COORD_BIAS = 1 << 20
KEY_END = 1 << 63


def _spread21(v):
    v = v.astype(np.uint64) & np.uint64(0x1FFFFF)
    for shift, mask in ((32, 0x1F00000000FFFF), (16, 0x1F0000FF0000FF), (8, 0x100F00F00F00F00F),
                        (4, 0x10C30C30C30C30C3), (2, 0x1249249249249249)):
        v = (v | (v << np.uint64(shift))) & np.uint64(mask)
    return v


def cells_of(points, cell):
    """Integer cell coordinates floor(p / cell), (N, 3) int64."""
    return np.floor(np.asarray(points, dtype=np.float64)[:, :3] / float(cell)).astype(np.int64)


def keys_of_cells(cells):
    """Morton key of biased cell coordinates: the key space that is split over ranks
    (the same function as pcp_grid_cell_keys)."""
    c = cells + COORD_BIAS
    if c.size and (c.min() < 0 or c.max() >= (1 << 21)):
        raise ValueError("cell coordinates outside +-2^20")
    return _spread21(c[:, 0]) | (_spread21(c[:, 1]) << np.uint64(1)) | (_spread21(c[:, 2]) << np.uint64(2))


def split_key_space(map_points, cell, world):
    """[(lo, hi)] per rank: contiguous key ranges covering [0, 2^63) with about equal map points."""
    keys = np.sort(keys_of_cells(cells_of(map_points, cell)))
    cuts = [0]
    for r in range(1, world):
        k = int(keys[min(len(keys) - 1, (len(keys) * r) // world)]) if len(keys) else 0
        cuts.append(max(k, cuts[-1]))
    cuts.append(KEY_END)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def slab_mask(map_points, cell, lo, hi):
    """Map points a rank owning keys [lo, hi) must hold: own cells plus the one-cell halo."""
    cells = cells_of(map_points, cell)
    mask = np.zeros(len(cells), dtype=bool)
    lo, hi = np.uint64(lo), np.uint64(min(hi, KEY_END - 1))
    top = hi == np.uint64(KEY_END - 1)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                k = keys_of_cells(cells + np.array([dx, dy, dz], dtype=np.int64))
                mask |= (k >= lo) & ((k < hi) | top)
    return mask


class RcclComm:
    """The library's own RCCL communicator (pcp_comm_*): one rank per GPU.  The 128-byte id made by rank 0
    reaches the other ranks through `share`, a callable taking rank 0's bytes (None elsewhere) and returning
    them on every rank; by default a torch.distributed broadcast (any backend) when world > 1."""

    def __init__(self, rank=0, world=1, device=0, share=None):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        ident = None
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(self._lib.pcp_comm_unique_id(buf))
            ident = buf.raw
        if self.world > 1:
            if share is None:
                import torch.distributed as dist

                def share(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            ident = share(ident)
        self._comm = C.c_void_p()
        _lib.check(self._lib.pcp_comm_init(self.device, self.world, self.rank, ident, C.byref(self._comm)))

    def close(self):
        if self._comm:
            self._lib.pcp_comm_destroy(self._comm)
            self._comm = None

    def abort(self):
        """After a failed call on this rank: ends the communicator without waiting, so that peers blocked in an
        all-reduce are released with an error (pcp_comm_abort)."""
        if self._comm:
            self._lib.pcp_comm_abort(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipMapBackend:
    """The product backend: a CellGrid on this rank's device."""

    def __init__(self, device=0):
        import torch
        from . import _lib
        from .cell_grid import CellGrid
        self._torch = torch
        self._lib = _lib.load()
        self._check = _lib.check
        self._CellGrid = CellGrid
        self.device = int(device)
        self.grid = None
        self.tile = None
        self.stream = None

    def partition(self, points, cell, world):
        from .cell_grid import partition_keys
        return partition_keys(points, cell, world, self.device)

    def build(self, points, normals, cell, nn_cell=None, slab=None):
        """slab = (lo, hi): build from the whole map, keeping this rank's share (selected on the device)."""
        torch = self._torch
        self.own_cell = float(cell)
        nn_cell = float(nn_cell) if nn_cell else self.own_cell
        slab3 = None if slab is None else (self.own_cell, slab[0], slab[1])
        if self.grid is None:
            self.grid = self._CellGrid(points, cell=nn_cell, normals=normals, device=self.device, slab=slab3)
        else:
            self.grid.rebuild(points, cell=nn_cell, normals=normals, slab=slab3)
        self.held = self.grid.held
        dev = torch.device("cuda", self.device)
        self.tile = torch.zeros(64, dtype=torch.float32, device=dev)
        # kernels and collective are ordered by one stream: a torch side stream that the grid adopts
        # (the legacy default stream cannot be adopted: a null handle means "the grid's own stream")
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize(dev)
        self._check(self._lib.pcp_grid_set_stream(self.grid._g, self.stream.cuda_stream))

    def append(self, points, normals, first_id, slab=None):
        slab3 = None if slab is None else (self.own_cell, slab[0], slab[1])
        kept = self.grid.append(points, normals=normals, first_id=first_id, slab=slab3)
        self.held = self.grid.held
        return kept

    def append_device(self, xyz_dev, n, first_id, slab=None):
        slab3 = None if slab is None else (self.own_cell, slab[0], slab[1])
        kept = self.grid.append_device(xyz_dev, n, first_id=first_id, slab=slab3)
        self.held = self.grid.held
        return kept

    def stream_context(self):
        return self._torch.cuda.stream(self.stream)

    def set_owned(self, lo, hi):
        self.grid.set_owned_keys(lo, hi, self.own_cell)

    def set_source(self, scan):
        self.grid.set_source(scan)

    def reset(self):
        self._check(self._lib.pcp_grid_icp_reset(self.grid._g))

    def partial_tile(self, mode, radius):
        self._check(self._lib.pcp_grid_ls_dev(self.grid._g, int(mode), float(radius), self.tile.data_ptr()))
        return self.tile

    def pose_step(self, tile, mode, max_iter, tol, full_iter):
        if tile.data_ptr() != self.tile.data_ptr():
            self.tile.copy_(tile)
        self._check(self._lib.pcp_grid_pose_dev(self.grid._g, self.tile.data_ptr(), int(mode), int(max_iter),
                                                float(tol), 1 if full_iter else 0))

    def finish(self):
        self._check(self._lib.pcp_grid_icp_finish(self.grid._g))

    def all_reduce_rccl(self, comm, tile):
        """The exchange step through the library's communicator, on the grid's stream."""
        self._check(self._lib.pcp_grid_allreduce_tile(self.grid._g, comm._comm, tile.data_ptr()))
        return tile

    def run_comm(self, comm, mode, radius, max_iter, tol, full_iter):
        """The whole partitioned loop in one ABI call (pcp_grid_icp_run_comm)."""
        import ctypes as C
        from ._lib import f32p
        dx = np.zeros((max_iter, 6), dtype=np.float32)
        hb = np.zeros((max_iter, 8, 8), dtype=np.float32)
        it, cv = C.c_int(0), C.c_int(0)
        try:
            self._check(self._lib.pcp_grid_icp_run_comm(self.grid._g, comm._comm, int(mode), float(radius), int(max_iter),
                                                        float(tol), 1 if full_iter else 0, f32p(dx), f32p(hb),
                                                        C.byref(it), C.byref(cv)))
        except Exception:
            comm.abort()        # peers still inside an all-reduce are released with an error instead of waiting for this rank
            raise
        return dx[:it.value], bool(cv.value), hb[:it.value]

    def fetch(self, max_iter):
        import ctypes as C
        from ._lib import f32p
        dx = np.zeros((max_iter, 6), dtype=np.float32)
        hb = np.zeros((max_iter, 8, 8), dtype=np.float32)
        it, cv, dn = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.pcp_grid_icp_fetch(self.grid._g, f32p(dx), f32p(hb), C.byref(it), C.byref(cv), C.byref(dn)))
        return dx[:it.value], bool(cv.value), bool(dn.value), hb[:it.value]

    def read_source(self):
        return self.grid.read_source()

    def close(self):
        if self.grid is not None:
            self.grid.cleanup()
            self.grid = None


class MapRegistration:
    """register(scan) -> (t_opt, R_opt) against a map held jointly by `world` ranks.

    Every rank constructs it with the same map_points/cell (each keeps only its slab) and calls
    register() with the same scan.  backend: an object with HipMapBackend's methods (tests inject a
    CPU checker); group: a torch.distributed process group, None = the default one when world > 1."""

    def __init__(self, map_points, cell, normals=None, rank=0, world=1, device=0, backend=None, group=None,
                 nn_cell=None, comm=None, ranges=None):
        """cell: edge of the partition cells (ownership and halo; the search radius may not exceed it).
        nn_cell: edge of the search grid's cells on each rank (default: cell); dense maps want it
        smaller, a few point spacings.
        comm: an RcclComm -- the exchange step then runs inside the library (pcp_grid_icp_run_comm /
        pcp_grid_allreduce_tile) instead of through torch.distributed.
        ranges: the ranks' key ranges [(lo, hi)] when they have been fixed elsewhere (e.g. cut by one rank from a sample
        of the map and broadcast, partition_cuts()): map_points then only needs to hold what this rank is to start
        with, the rest of the map arrives through ingest()."""
        self.rank, self.world, self.cell = int(rank), int(world), float(cell)
        self.group = group
        self.comm = comm
        if comm is not None and (comm.world != self.world or comm.rank != self.rank):
            raise ValueError("the communicator's rank / world differ from the registration's")
        self.backend = backend if backend is not None else HipMapBackend(device)
        pts = np.asarray(map_points, dtype=np.float64)[:, :3]
        on_device = hasattr(self.backend, "partition")
        # the NumPy statement of the partition rule serves backends without a device (the CPU tests)
        if ranges is not None:
            self.ranges = [(int(lo), int(hi)) for lo, hi in ranges]
            if len(self.ranges) != self.world:
                raise ValueError("one key range per rank")
        else:
            self.ranges = self.backend.partition(pts, cell, self.world) if on_device else split_key_space(pts, cell, self.world)
        lo, hi = self.ranges[self.rank]
        if self.world > 1 and on_device:
            self.backend.build(pts, normals, self.cell, nn_cell, slab=(lo, hi))
            self.slab_points = self.backend.held
        else:
            if self.world > 1:
                keep = slab_mask(pts, cell, lo, hi)
                pts = pts[keep]
                normals = None if normals is None else np.asarray(normals)[keep]
            self.slab_points = len(pts)
            if nn_cell is None:
                self.backend.build(pts, normals, self.cell)
            else:
                self.backend.build(pts, normals, self.cell, nn_cell)
        self.backend.set_owned(lo, hi)
        self.total_points = len(np.asarray(map_points))

    def ingest(self, points, normals=None):
        """New map points (e.g. the voxels a Voxelizer staged this frame, through map_sink.MapSink) join the
        resident map without a rebuild.  Every rank is handed the same batch and keeps the points in or next
        to its key range (the routing of the partitioned map: no exchange, the ranges were fixed when the map
        was split); ids continue the numbering of the whole map.  Returns the points this rank kept."""
        pts = np.asarray(points, dtype=np.float64)[:, :3]
        lo, hi = self.ranges[self.rank]
        kept = self.backend.append(pts, normals, self.total_points, slab=(lo, hi) if self.world > 1 else None)
        self.total_points += len(pts)
        self.slab_points += kept
        return kept

    def ingest_device(self, xyz_dev, n):
        """ingest() for n points already in this rank's device memory (n x 3 float64 at address xyz_dev)."""
        lo, hi = self.ranges[self.rank]
        kept = self.backend.append_device(xyz_dev, n, self.total_points, slab=(lo, hi) if self.world > 1 else None)
        self.total_points += int(n)
        self.slab_points += kept
        return kept

    def _all_reduce(self, tile):
        if self.comm is not None:
            return self.backend.all_reduce_rccl(self.comm, tile)
        if self.world == 1:
            return tile
        import contextlib
        import torch.distributed as dist
        ctx = self.backend.stream_context() if hasattr(self.backend, "stream_context") else contextlib.nullcontext()
        with ctx:
            if tile.is_cuda and dist.get_backend(self.group) != "nccl":
                host = tile.cpu()                                 # rehearsal over gloo
                dist.all_reduce(host, group=self.group)
                tile.copy_(host)
            else:
                dist.all_reduce(tile, group=self.group)
        return tile

    def register(self, scan, mode=MODE_POINT, radius=None, max_iter=10, tol=1e-6, full_iter=False, check_every=1):
        """Returns (t_opt, R_opt, dx_trace, converged); p <- R_opt p - t_opt moves the scan onto the map
        (the reference's convention, EnvironmentConstructor.py:101-102)."""
        from .icp_container import PointcloudIcpContainer
        radius = self.cell if radius is None else float(radius)
        if radius > self.cell and self.world > 1:
            raise ValueError("the search radius must not exceed the cell size (one-cell halo)")
        b = self.backend
        b.set_source(scan)
        if self.comm is not None and check_every is None:
            # one ABI call: all max_iter exchanges enqueued by the library itself
            dx, converged, hb = b.run_comm(self.comm, mode, radius, max_iter, tol, full_iter)
            applied = len(dx) - 1 if (converged and mode == MODE_PLANE and not full_iter) else len(dx)
            t_opt, R_opt = PointcloudIcpContainer(None)._replay(dx, applied)
            self.last_tiles = hb
            return t_opt, R_opt, dx, converged
        b.reset()
        for it in range(max_iter):
            tile = self._all_reduce(b.partial_tile(mode, radius))
            b.pose_step(tile, mode, max_iter, tol, full_iter)
            if check_every and (it + 1) % check_every == 0 and it + 1 < max_iter:
                if b.fetch(max_iter)[2]:
                    break
        b.finish()
        dx, converged, _, hb = b.fetch(max_iter)
        applied = len(dx) - 1 if (converged and mode == MODE_PLANE and not full_iter) else len(dx)
        t_opt, R_opt = PointcloudIcpContainer(None)._replay(dx, applied)
        self.last_tiles = hb
        return t_opt, R_opt, dx, converged

    def moved_scan(self):
        return self.backend.read_source()

    def close(self):
        self.backend.close()
