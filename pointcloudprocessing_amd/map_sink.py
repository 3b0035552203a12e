"""The accumulated map as the destination of the Voxelizer's staged voxels.

The reference's Voxelizer hands the points of every voxel it stages (full, or static by its Bayes filter:
VoxelStage.glsl:41-60, :201-315) to a VoxelManager, whose frameVoxelized draws them (Voxelizer.py:130-133,
VoxelManager.py:31-45).  MapSink takes that manager's place: the staged points join a cell grid on the
device (CellGrid.append / MapRegistration.ingest, no rebuild), which is the map later scans are registered
against -- the voxel map the Voxelizer accumulates and the map scan-to-map searches become one structure.
"""
import numpy as np


class MapSink:
    needs_renderer = False            # Voxelizer._hand_off feeds it without a renderer

    def __init__(self, target, compute=None, defer=True):
        """target: a CellGrid (append) or a scan_to_map.MapRegistration (ingest: routed by key range).
        compute: the Voxelizer's ComputeShader, when it is on the target's device -- the staged voxels are then
        resolved to points and appended on the device (a device copy of the point store is switched on; do this
        before the map's first addPoints), and frameVoxelized is not used.
        defer: the per-frame batches wait on the device and join the map when it is next read (CellGrid.defer_appends)."""
        self.target = target
        grid = self._grid()
        if defer and grid is not None and hasattr(grid, "defer_appends"):
            grid.defer_appends(True)
        self._batch = []
        self.points_fed = 0
        self.voxels_fed = 0
        self.compute = compute
        if compute is not None:
            compute.keepPointStore(True)

    def framesStagedOnDevice(self):
        """Voxelizer._hand_off calls this instead of frameVoxelized per row when `compute` was given: the rows the
        last stager dispatch staged, gathered from the device point store, go into the map in one append."""
        ptr, n, rows = self.compute.stagedPointsDevice()
        if n:
            if hasattr(self.target, "ingest_device"):
                self.target.ingest_device(ptr, n)
            else:
                self.target.append_device(ptr, n)
        self.points_fed += n
        self.voxels_fed += rows
        return n

    def stagedLogTaken(self, ptr, n, rows):
        """A run of frames fed without waits in between (chain.OdometryChain) hands over what all of them staged: n points
        of `rows` voxels in device memory (ComputeShader.stagedLogTake)."""
        if n:
            if hasattr(self.target, "ingest_device"):
                self.target.ingest_device(ptr, n)
            else:
                self.target.append_device(ptr, n)
        self.points_fed += n
        self.voxels_fed += rows
        return n

    def beforeLogReuse(self):
        """The copies out of a log handed over earlier have completed when this returns."""
        grid = self._grid()
        if grid is not None and hasattr(grid, "synchronize"):
            grid.synchronize()

    def _grid(self):
        if hasattr(self.target, "collect"):
            return self.target
        return getattr(getattr(self.target, "backend", None), "grid", None)

    def collecting(self):
        """Context manager for a producer that hands over many batches in a row (Voxelizer.stageAllRemaningVoxels): the
        map takes them all in at the end, in one merge (CellGrid.collect)."""
        grid = self._grid()
        if self.compute is None or grid is None or not hasattr(grid, "collect"):
            import contextlib
            return contextlib.nullcontext()
        return grid.collect()

    def frameVoxelized(self, voxel_points, colors, separate_colors):
        """VoxelManager.frameVoxelized's signature; the colours are the renderer's business, not the map's."""
        if len(voxel_points):
            self._batch.append(np.asarray(voxel_points, dtype=np.float64)[:, :3])
            self.voxels_fed += 1

    def flush(self):
        """One append per dispatch: the staged voxels of a frame go in together."""
        if not self._batch:
            return 0
        pts = np.ascontiguousarray(np.concatenate(self._batch, axis=0))
        self._batch = []
        (self.target.ingest if hasattr(self.target, "ingest") else self.target.append)(pts)
        self.points_fed += len(pts)
        return len(pts)
