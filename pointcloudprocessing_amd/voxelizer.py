"""Host side of the voxel map with the reference's call surface (Voxelizer.py,
class Voxelizer; file:line citations point there).

The device appends the points of a new cloud to the voxels already in the table
(ComputeShader.prepareDispatchVoxelizer) and runs the static/dynamic filter
(prepareDispatchVoxelStager), as in the reference.  Voxels that do not exist yet
are created by the reference on the host, one at a time in a Python loop (an O(M)
shift of the sorted table per voxel, :199-255) followed by a re-upload of table
and rows; with this package's ComputeShader they are created on the device as well
(insertUnknownVoxels) and only the table comes back to the host per frame.  Table,
data-row numbering, counts and id sets come out the same.  The host copy of the rows
(voxel_data) is read back from the device when it is looked at (the attribute itself,
refreshVoxelData() or any getter of stored ids), never per frame.
When another compute object is injected (e.g. the reference's own), the batch
merge below runs on the host instead.
"""
import numpy as np


class _StagedRows:
    """The rows a fed frame's stager dispatch released: their number is known from the frame's record, the rows
    themselves are fetched from the device when somebody reads one (a device sink never does)."""

    def __init__(self, compute, count, max_points):
        self._compute, self._count, self._max_points, self._rows = compute, int(count), max_points, None

    def __len__(self):
        return self._count

    def _fetch(self):
        if self._rows is None:
            self._rows = self._compute.readStagedRows(self._count, self._max_points) if self._count else []
        return self._rows

    def __iter__(self):
        return iter(self._fetch())

    def __getitem__(self, k):
        return self._fetch()[k]


class Voxelizer:

    feed_frames = True        # addPoints as one enqueue + one wait when the map lives on the device; False: step by step

    def __init__(self, compute_shader, renderer=None, voxman=None):
        self.voxel_size = None
        self.max_points = 1024            # fixed, = row width of the device voxel data (:21)
        self.compute = compute_shader
        self.renderer = renderer
        self.voxman = voxman              # render hand-off object with frameVoxelized(points, colors, separate)

        self._index_stale = False
        self.voxel_index = None           # (M, 4) int32: vx, vy, vz, data row; sorted; grows x2
        self._rows_stale = False
        self.voxel_data = None            # (M, max_points) int32: [0] count, [1:] point ids
        self._stats_stale = False
        self._map_epoch = None
        self.voxel_stats = None           # (M, 4) float32, by data row
        self.stored_voxels = None

        self._pending = []                # clouds noted by handleAddingPoints, not yet in added_points
        self.added_points = None
        self.added_colors = None
        self.last_added_point = None
        self.bigint = 9999999

        self.realloc_needed = None
        self.prev_stored_voxels = None
        self.separate_colors = None
        self.filter_outliers = None
        self.last_run_realloc = None
        self.inited = False

    # The reference's host copy of the rows goes stale with every dispatch (the device appends to it) and
    # is brought up to date by refreshVoxelData.  With the map kept on the device the host copy is not
    # even written when voxels are created -- touching one count in every 4 KB row of a lazily zeroed
    # array costs a page fault per voxel, more than the whole device side of a frame -- and is instead
    # read back when somebody looks at it.
    # The table too, once whole frames are fed (addPoints -> ComputeShader.voxelAddFrame): the device merges the new
    # voxels into its table and the host copy is fetched when somebody looks at it.
    @property
    def voxel_index(self):
        if self._index_stale and self._voxel_index is not None:
            self.compute.readVoxelIndex(self.stored_voxels, self._voxel_index)
            self._index_stale = False
        return self._voxel_index

    @voxel_index.setter
    def voxel_index(self, value):
        self._voxel_index = value
        self._index_stale = False

    @property
    def voxel_data(self):
        if self._rows_stale and self._voxel_data is not None:
            self._pull_rows()
        return self._voxel_data

    @voxel_data.setter
    def voxel_data(self, value):
        self._voxel_data = value
        self._rows_stale = False

    # The statistics likewise: the reference uploads them to every stager dispatch and reads them back (16 bytes per
    # voxel each way per frame); kept on the device, the host copy is fetched when somebody looks at it.
    @property
    def voxel_stats(self):
        if self._stats_stale and self._voxel_stats is not None:
            if getattr(self.compute, "voxel_map_epoch", None) != self._map_epoch:
                raise RuntimeError("the device's voxel map has been replaced (another Voxelizer took this ComputeShader): "
                                   "its statistics can no longer be fetched")
            self.compute.readVoxelStatistics(self._voxel_stats)
            self._stats_stale = False
        if (self._voxel_stats is not None and getattr(self, "inited", False)
                and getattr(self.compute, "resident_statistics", False)):
            # the device copy is the authority: an in-place edit of the mirror would be lost silently, so the mirror is
            # handed out read-only (assigning a whole array to voxel_stats makes the host copy the authority again)
            view = self._voxel_stats.view()
            view.flags.writeable = False
            return view
        return self._voxel_stats

    @voxel_stats.setter
    def voxel_stats(self, value):
        self._voxel_stats = value
        self._stats_stale = False
        # statistics assigned from outside while a map is live: the host copy is the authority from here on
        if getattr(self, "inited", False) and getattr(self.compute, "resident_statistics", False):
            self.compute.resident_statistics = False

    def _detach_from_device(self):
        """Called by the ComputeShader before another map replaces this one on the device: every mirror that would have
        been fetched on demand is fetched now."""
        if not self.inited or self.stored_voxels is None:
            return
        if self._index_stale and self._voxel_index is not None:
            self.voxel_index
        if self._rows_stale and self._voxel_data is not None:
            self._pull_rows()
        if self._stats_stale and self._voxel_stats is not None:
            self.voxel_stats

    def _pull_rows(self):
        rows = len(self._voxel_index)
        if len(self._voxel_data) < rows:        # rows past stored_voxels are zero, on the device too
            self._voxel_data = np.zeros((rows, self.max_points), dtype=np.int32)
        self.compute.getFullVoxelData(self.stored_voxels, self.max_points, self._voxel_data)
        self._rows_stale = False

    # ------------------------------------------------------------------ life cycle (:48-79, :382-388)
    def init(self, voxel_size):
        self.inited = False
        self.last_added_point = 0
        self._pending = []
        self.stored_voxels = 0
        self.realloc_needed = True
        self.prev_stored_voxels = -1
        self.initVoxels()
        self.last_run_realloc = None
        self.voxel_size = np.array([voxel_size]).astype(np.float32)
        self._rows_stale = False
        if hasattr(self.compute, "voxelReserve"):
            try:
                self.compute.voxelReserve(len(self._voxel_index), self.max_points, owner=self)
            except TypeError:                       # a compute object with the older signature
                self.compute.voxelReserve(len(self._voxel_index), self.max_points)
            self.realloc_needed = False
            self.compute.resident_statistics = True     # fresh map: every voxel starts from the default statistics
            self._map_epoch = getattr(self.compute, "voxel_map_epoch", None)
        self.inited = True

    def initVoxels(self, init_size=100):
        n = 4096
        self.voxel_index = np.full((n, 4), self.bigint, dtype=np.int32)
        self.voxel_data = np.zeros((n, self.max_points), dtype=np.int32)
        self.voxel_stats = np.zeros((n, 4), dtype=np.float32)
        self.voxel_stats[:, 0] = 0.5

    def cleanup(self):
        self.inited = False
        self.last_added_point = 0
        self.stored_voxels = 0
        self.realloc_needed = True
        self.prev_stored_voxels = -1
        self.voxel_index = self.voxel_data = self.voxel_stats = None
        self.added_points = self.added_colors = None

    # ------------------------------------------------------------------ per-frame entry (:81-165)
    def _can_feed(self):
        """True when the whole frame can go to the device as one enqueue (ComputeShader.voxelAddFrame): the map's table,
        rows and statistics live there and nothing on the host has replaced them."""
        c = self.compute
        return (self.feed_frames and self.inited and hasattr(c, "voxelAddFrame") and getattr(c, "_voxel_resident", None) == self.stored_voxels
                and bool(getattr(c, "resident_statistics", False)) and getattr(c, "voxel_map_epoch", None) == self._map_epoch)

    def _took_frame(self, rec, hand_off=True):
        """Book-keeping after a fed frame: sizes follow the reference's growth rule, contents are fetched on demand."""
        prev = self.stored_voxels
        self._index_stale = self._rows_stale = self._stats_stale = True      # the device holds the contents
        self.stored_voxels = rec.stored
        while rec.stored >= len(self._voxel_index):
            self._grow()
        self.realloc_needed, self.prev_stored_voxels, self.last_run_realloc = False, prev, False
        if hand_off:
            self._hand_off(_StagedRows(self.compute, rec.staged, self.max_points))

    def addPoints(self, np_points, colors, printt=False):
        if self._can_feed():
            # dispatch, stager and creation of the new voxels as one stream of device work, one wait (pcp_voxel_add)
            self.handleAddingPoints(np_points, colors)
            pts32 = self._pending[-1][1]
            begin = self.last_added_point - len(np_points)
            ticket = self.compute.voxelAddFrame(pts32, self.voxel_size, self.stored_voxels, begin, self.max_points,
                                                self.filter_outliers)
            self._took_frame(self.compute.voxelAddWait(ticket))
            return
        self.handleAddingPoints(np_points, colors)
        pts32 = self._pending[-1][1]                   # the float32 copy the point store keeps serves the dispatch too
        begin = self.last_added_point - len(np_points)
        unknown_points, _ = self.compute.prepareDispatchVoxelizer(
            np_points=pts32, voxel_index=self.voxel_index if getattr(self.compute, "_voxel_resident", None) != self.stored_voxels
            else self._voxel_index, voxel_data=self._voxel_data,
            voxel_size=self.voxel_size, stored_voxel_num=self.stored_voxels, begin_index=begin,
            max_points_to_store=self.max_points, realloc_needed=self.realloc_needed,
            prev_stored_voxels=self.prev_stored_voxels, debug=False)
        if hasattr(self.compute, "insertUnknownVoxels"):
            self._rows_stale = True           # the device appended to its rows; the host copy is read back on demand
        voxels_to_render = self._stage(False)
        self.realloc_needed, self.prev_stored_voxels = self.storeUnknownPoints(unknown_points, printt)
        self.last_run_realloc = self.realloc_needed
        self._hand_off(voxels_to_render)

    def addPointsDevice(self, device_ptr, n_points, layout=8, after_stream=None):
        """addPoints for a cloud already in device memory (layout 8: n rows of 3 float64, cast to float32 on the device as
        handleAddingPoints does; 4: float32 rows of 4): nothing of the frame touches the host but its 32-byte record.
        The global point store is kept on the device for such frames (ComputeShader.keepPointStore must be on for
        added_points / getStoredPoints to be available; they are fetched on demand)."""
        if not self._can_feed():
            raise RuntimeError("addPointsDevice needs the voxel map resident on the device (a ComputeShader of this package, "
                               "init() called)")
        begin = self.last_added_point
        self._pending.append((begin, int(n_points), None))          # a segment the host store does not hold yet
        self.last_added_point += int(n_points)
        ticket = self.compute.voxelAddFrame(None, self.voxel_size, self.stored_voxels, begin, self.max_points, self.filter_outliers,
                                            device_ptr=device_ptr, n_points=n_points, layout=layout, after_stream=after_stream)
        self._took_frame(self.compute.voxelAddWait(ticket))

    # -- frames fed by a device-side producer (chain.OdometryChain): the producer enqueues pcp_voxel_add_dev itself
    def _feed_begin(self):
        """-> (context handle, voxel size, stored voxels or -1, begin index, row width, filter flag) for a producer that
        feeds this map from device memory; the same conditions as addPointsDevice."""
        if not self._can_feed():
            raise RuntimeError("the voxel map must be resident on the device (a ComputeShader of this package, init() called)")
        stored = -1 if self.compute._feed_clean else int(self.stored_voxels)
        return (self.compute._ctx, float(np.asarray(self.voxel_size).reshape(-1)[0]), stored, int(self.last_added_point),
                int(self.max_points), int(bool(self.filter_outliers)))

    def _feed_noted(self, counts, first_serial):
        """Book-keeping for frames a producer has enqueued: their points are segments of the store the host does not hold
        (they come back from the device store when somebody looks), their tickets run from first_serial on."""
        for n in counts:
            self._pending.append((self.last_added_point, int(n), None))
            self.last_added_point += int(n)
        self.compute._feed_clean = True
        self.__dict__.setdefault("_feed_tickets", []).append(int(first_serial) + len(counts) - 1)

    def _log_begin(self, capacity_points):
        """Before a producer enqueues a run of frames: when the voxel manager takes staged points on the device (a
        map_sink.MapSink made with this map's ComputeShader), the run's frames log the points of the rows they stage
        (ComputeShader.stagedLogBegin) -> the log's slot, else None (only the last frame's rows are handed on then)."""
        sink = self.voxman
        if sink is None or getattr(sink, "compute", None) is not self.compute or not hasattr(sink, "stagedLogTaken"):
            return None
        slot = self.__dict__.get("_log_next", 0)
        sink.beforeLogReuse()                          # whatever the sink still copies out of this slot's last contents
        try:
            self.compute.stagedLogBegin(slot, int(capacity_points))
        except RuntimeError as e:
            if "has not been taken" not in str(e):
                raise
            try:                                       # the log of a run its producer abandoned (never collected)
                self.compute.stagedLogTake(slot)
            except RuntimeError:
                pass
            self.compute.stagedLogBegin(slot, int(capacity_points))
        self._log_next = 1 - slot
        return slot

    def _log_end(self, slot, drop=False):
        self.compute.stagedLogEnd(slot)
        if drop:                                       # the run was never enqueued whole: nothing of it is handed on
            try:
                self.compute.stagedLogTake(slot)
            except RuntimeError:
                pass

    def _feed_wait(self, log_slot=None):
        """Waits for the oldest run of enqueued frames and takes its last record (voxels stored).  With a log
        (_log_begin) the points every frame of the run staged go to the voxel manager; without one only the rows staged by
        the last frame are handed on (the staged rows of the frames before it are not kept)."""
        tickets = self.__dict__.get("_feed_tickets") or []
        if not tickets:
            return
        if log_slot is None:
            self._took_frame(self.compute.voxelAddWait(tickets.pop(0)))
            return
        try:
            self._took_frame(self.compute.voxelAddWait(tickets.pop(0)), hand_off=False)
        finally:
            ptr, n, rows = self.compute.stagedLogTake(log_slot)
        self.voxman.stagedLogTaken(ptr, n, rows)

    def _stage(self, everything):
        """One stager dispatch (:116-119, :141-152) -> the staged rows.  With the statistics resident on the device the
        host array only tells the dispatch how many rows there are."""
        resident = bool(getattr(self.compute, "resident_statistics", False))
        if resident and getattr(self.compute, "voxel_map_epoch", None) != self._map_epoch:
            raise RuntimeError("the device's voxel map has been replaced (another Voxelizer took this ComputeShader): "
                               "this map can no longer be staged from its resident statistics")
        stats = self._voxel_stats if resident else self.voxel_stats
        # a sink that resolves the staged rows on the device never looks at them on the host: they are not fetched
        # (4 KB per row), only counted
        on_device = (resident and getattr(self.voxman, "compute", None) is not None and hasattr(self.voxman, "framesStagedOnDevice")
                     and not (self.renderer is None and getattr(self.voxman, "needs_renderer", True)))
        if on_device:
            out, count = self.compute.prepareDispatchVoxelStager(
                self._voxel_index, self.stored_voxels, self.max_points, stats, filter_outliers=self.filter_outliers,
                stage_everything=everything, fetch_rows=False)
            self._stats_stale = True
            return _StagedRows(self.compute, count, self.max_points)
        out, rows = self.compute.prepareDispatchVoxelStager(
            self._voxel_index, self.stored_voxels, self.max_points, stats,
            filter_outliers=self.filter_outliers, stage_everything=everything)
        if resident:
            self._stats_stale = True
        else:
            self.voxel_stats = out
        return rows

    def _hand_off(self, voxels_to_render):
        """:130-133 -- the points of every voxel staged by this dispatch go to the voxel manager.  The
        reference's manager draws them (VoxelManager.frameVoxelized :31-45, needs the renderer); a manager
        that says `needs_renderer = False` (map_sink.MapSink: the staged points accumulate in a cell grid
        that scans are registered against) is fed without one."""
        if self.voxman is None or (self.renderer is None and getattr(self.voxman, "needs_renderer", True)):
            return
        if getattr(self.voxman, "compute", None) is not None and hasattr(self.voxman, "framesStagedOnDevice"):
            if len(voxels_to_render):
                self.voxman.framesStagedOnDevice()         # the device resolves the staged rows itself
            return
        for v_slice in voxels_to_render:
            points, colors = self.getStoredPointsFromVoxelDataSlice(v_slice)
            self.voxman.frameVoxelized(points, colors, self.separate_colors)
        if hasattr(self.voxman, "flush"):
            self.voxman.flush()

    def refreshVoxelData(self):
        if not self.last_run_realloc:
            self._pull_rows()

    def _fresh_rows(self):
        if self._rows_stale:
            self.refreshVoxelData()

    def stageAllRemaningVoxels(self, ui=None):
        """:139-165 -- stage until nothing comes back, then ten flush passes (a voxel
        recovered by its neighbours is only staged by the NEXT dispatch)."""
        def one_pass():
            rows = self._stage(True)
            self._hand_off(rows)
            return rows
        import contextlib
        # a sink that can take many batches in one go (map_sink.MapSink on the device) merges them when the passes are over
        gather = self.voxman.collecting() if hasattr(self.voxman, "collecting") else contextlib.nullcontext()
        with gather:
            while True:
                rows = one_pass()
                if not (len(rows) if isinstance(rows, _StagedRows) else np.any(rows)):
                    for _ in range(10):
                        one_pass()
                    break

    # ------------------------------------------------------------------ host half of the map
    def vectorizedGetVoxelCoord(self, np_points):
        """:167-168"""
        return np.ceil(np_points.astype(np.float32) / self.voxel_size).astype(np.int32)

    def getVoxelCoord(self, point):
        """:391-396"""
        return np.ceil(point / self.voxel_size).astype(np.int32)

    @staticmethod
    def _lex_keys(coords):
        c = np.asarray(coords, dtype=np.int64)
        if c.size and (np.abs(c).max() >= (1 << 20)):
            raise RuntimeError("voxel coordinate beyond +-2^20")
        off = 1 << 20
        return ((c[:, 0] + off) << 42) | ((c[:, 1] + off) << 21) | (c[:, 2] + off)

    def _grow(self):
        """:238-252 -- double table, data and statistics."""
        if self._index_stale:                # the device holds the table: only the size of the host copy changes
            self._voxel_index = np.full((2 * len(self._voxel_index), 4), self.bigint, dtype=np.int32)
        else:
            self._voxel_index = np.vstack([self._voxel_index, np.full_like(self._voxel_index, self.bigint)])
        # Rows past stored_voxels are zero, so only the used rows are carried over into a freshly
        # zeroed (lazily mapped) array; when the device holds the authoritative rows the host copy is
        # stale anyway and will be read back before anybody looks at it, so only the counts move.
        if not self._rows_stale:             # a stale copy is re-made at full size when it is next read
            old, used = self._voxel_data, self.stored_voxels
            self._voxel_data = np.zeros((2 * len(old), old.shape[1]), dtype=old.dtype)
            self._voxel_data[:used] = old[:used]
        if self._stats_stale:                # the device holds the statistics: only the size of the host copy changes
            self._voxel_stats = np.zeros((2 * len(self._voxel_stats), 4), dtype=np.float32)
        else:
            fresh = np.zeros_like(self._voxel_stats)
            fresh[:, 0] = 0.5
            self._voxel_stats = np.vstack([self._voxel_stats, fresh])

    def storeUnknownPoints(self, unknown_points, printt=False):
        """:170-306 -> (was_realloc, prev_stored_voxels).  Creates one voxel per
        distinct coordinate of the unknown points, in lexicographic order of the
        coordinates (data rows are numbered in that order), each with the count of
        ALL its points and the ids of the first max_points-1 of them."""
        n = int(unknown_points[0])
        prev_stored = self.stored_voxels
        if n == 0:
            return False, prev_stored
        if hasattr(self.compute, "insertUnknownVoxels"):
            new_stored, _ = self.compute.insertUnknownVoxels(self.max_points)
            self._rows_stale = True
            while new_stored >= len(self._voxel_index):
                self._grow()                       # host arrays only: the device grew its own
            self.compute.readVoxelIndex(new_stored, self._voxel_index)
            self.stored_voxels = new_stored
            return False, prev_stored
        ids = np.sort(unknown_points[1:n + 1])     # the device appends in no particular order
        coords = self.vectorizedGetVoxelCoord(self.added_points[ids][:, :3])
        uniq, inverse, counts = np.unique(coords, axis=0, return_inverse=True, return_counts=True)
        order = np.argsort(inverse.reshape(-1), kind="stable")
        sorted_ids = ids[order]

        table_keys = self._lex_keys(self.voxel_index[:prev_stored, :3])
        new_keys = self._lex_keys(uniq)
        pos = np.searchsorted(table_keys, new_keys)
        if prev_stored and np.any(table_keys[np.minimum(pos, prev_stored - 1)] == new_keys):
            raise RuntimeError("Here all unknown points should have a new voxel created")

        new_stored = prev_stored + len(uniq)
        was_realloc = False
        while new_stored >= len(self.voxel_index):
            if not was_realloc and prev_stored > 0:
                # rows appended on the device since the last read-back would be lost by the
                # full re-upload that follows a reallocation (:241-244)
                self.compute.getFullVoxelData(prev_stored, self.max_points, self.voxel_data)
            self._grow()
            was_realloc = True

        rows = np.arange(prev_stored, new_stored, dtype=np.int32)
        entries = np.column_stack([uniq.astype(np.int32), rows])
        self.voxel_index[:new_stored] = np.insert(self.voxel_index[:prev_stored], pos, entries, axis=0)

        self.voxel_data[rows, 0] = counts
        keep = np.minimum(counts, self.max_points - 1)
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        # scatter the first `keep` ids of every group into its row
        within = np.arange(len(sorted_ids)) - np.repeat(starts, counts)
        sel = within < np.repeat(keep, counts)
        self.voxel_data[np.repeat(rows, counts)[sel], 1 + within[sel]] = sorted_ids[sel]
        self.stored_voxels = new_stored
        return was_realloc, prev_stored

    def handleAddingPoints(self, np_points, colors=None):
        """:398-425 -- the global point store the voxel rows index into.  A cloud is only noted here;
        it is written into added_points / added_colors (float32 (cap, 4) / (cap, 3), capacity doubling
        as in the reference) when one of the two is next looked at: the per-frame path never reads them."""
        n = len(np_points)
        src = np.asarray(np_points)[:, :3]
        if src.T.flags.c_contiguous and not src.flags.c_contiguous:
            # a transposed (3, n) cloud, e.g. (R @ p.T).T of LidarDataReader.py:66-67: the float32 copy keeps the
            # coordinate rows (a contiguous cast; into (n, 3) order it is a strided one, five times slower) and
            # the device interleaves them
            pts = np.array(src.T, dtype=np.float32, order="C").T
        else:
            pts = np.array(src, dtype=np.float32, order="C")                 # the caller may reuse its array
        cols = None if colors is None else np.array(colors, dtype=np.float32)
        self._pending.append((self.last_added_point, pts, cols))
        self.last_added_point += n

    def _flush_points(self):
        if not self._pending:
            return
        pending, self._pending = self._pending, []
        need = self.last_added_point
        if self._added_points is None:
            first = pending[0][1] if isinstance(pending[0][1], int) else len(pending[0][1])
            self._added_points = np.empty((max(first * 2, 1), 4), dtype=np.float32)
            self._added_colors = np.empty((max(first * 2, 1), 3), dtype=np.float32)
        size = len(self._added_points)
        if need >= size:
            while need >= size:                            # doubling, as the reference does
                size *= 2
            used = pending[0][0]
            for name in ("_added_points", "_added_colors"):  # only the rows in use are carried over
                old = getattr(self, name)
                new = np.empty((size, old.shape[1]), dtype=old.dtype)
                new[:used] = old[:used]
                setattr(self, name, new)
        for start, pts, cols in pending:
            if isinstance(pts, int):             # a frame fed from device memory: its points come back from the device store
                self._added_points[start:start + pts] = self.compute.readPointStore(start, pts)
                self._added_colors[start:start + pts] = 255.0
                continue
            self._added_points[start:start + len(pts), :3] = pts
            # colors=None means white (np.full((n, 3), 255) in the reference), written without the temporary
            self._added_colors[start:start + len(pts)] = 255.0 if cols is None else cols

    @property
    def added_points(self):
        self._flush_points()
        return self._added_points

    @added_points.setter
    def added_points(self, value):
        self._added_points = value
        if value is None:
            self._pending = []

    @property
    def added_colors(self):
        self._flush_points()
        return self._added_colors

    @added_colors.setter
    def added_colors(self, value):
        self._added_colors = value

    # ------------------------------------------------------------------ table search on the host (:429-570)
    def xyzSortedArray(self, np_array):
        return np_array[np.lexsort((np_array[:, 2], np_array[:, 1], np_array[:, 0]))]

    def bin_search(self, sorted_array_values, value, find_first=True):
        """:550-570 -> (first/last occurrence or -1, insertion point)."""
        lo, hi, occ = 0, len(sorted_array_values) - 1, -1
        while lo <= hi:
            mid = (lo + hi) // 2
            v = sorted_array_values[mid]
            if v == value:
                occ = mid
                if find_first:
                    hi = mid - 1
                else:
                    lo = mid + 1
            elif v < value:
                lo = mid + 1
            else:
                hi = mid - 1
        return occ, lo

    def bin_search_gpu(self, axis, value, sorted_array_values, find_first, sfrom, sto):
        """:515-548 -- the shader's variant: search rows [sfrom, sto] on one axis."""
        occ, _ = self.bin_search(sorted_array_values[sfrom:sto + 1, axis], value, find_first)
        return -1 if occ == -1 else occ + sfrom

    def findVoxelId(self, xyz_sorted_array, value):
        """:469-512 -> (position, None) or (-1, insertion position)."""
        if self.stored_voxels == 0:
            return -1, 0
        first_x, ins_x = self.bin_search(xyz_sorted_array[:, 0], value[0], True)
        if first_x == -1:
            return -1, ins_x
        last_x, _ = self.bin_search(xyz_sorted_array[:, 0], value[0], False)
        ys = xyz_sorted_array[first_x:last_x + 1, 1]
        first_y, ins_y = self.bin_search(ys, value[1], True)
        if first_y == -1:
            return -1, ins_x + ins_y
        last_y, _ = self.bin_search(ys, value[1], False)
        first_y, last_y = first_y + first_x, last_y + first_x
        zs = xyz_sorted_array[first_y:last_y + 1, 2]
        first_z, ins_z = self.bin_search(zs, value[2], True)
        if first_z == -1:
            return -1, ins_x + ins_y + ins_z
        last_z, _ = self.bin_search(zs, value[2], False)
        if first_z != last_z:
            raise RuntimeError("In this impl. no same values should be present")
        return first_z + first_y, None

    def findVoxelId_gpu(self, xyz_sorted_array, value, lens_data=None):
        """:432-466 -- the shader's search, bounded by the stored voxel count."""
        stored = self.stored_voxels if lens_data is None else int(lens_data[1])
        if stored == 0:
            return -1
        fx = self.bin_search_gpu(0, value[0], xyz_sorted_array, True, 0, stored - 1)
        if fx == -1:
            return -1
        lx = self.bin_search_gpu(0, value[0], xyz_sorted_array, False, 0, stored - 1)
        fy = self.bin_search_gpu(1, value[1], xyz_sorted_array, True, fx, lx)
        if fy == -1:
            return -1
        ly = self.bin_search_gpu(1, value[1], xyz_sorted_array, False, fx, lx)
        return self.bin_search_gpu(2, value[2], xyz_sorted_array, True, fy, ly)

    # ------------------------------------------------------------------ getters (:379-380, :573-630)
    def getVoxelDensities(self):
        self._fresh_rows()
        return self.voxel_data[:, 0]

    def getStoredCount(self, voxel_data_index):
        if voxel_data_index >= self.stored_voxels:
            return None
        self._fresh_rows()
        return self.voxel_data[voxel_data_index][0]

    def getStoredPointIndicies(self, voxel_data_index):
        if voxel_data_index >= self.stored_voxels:
            return None
        self._fresh_rows()
        stored = self.voxel_data[voxel_data_index][0]
        return self.voxel_data[voxel_data_index][1:min(stored, self.max_points)]

    def getStoredPoints(self, voxel_data_index):
        if voxel_data_index >= self.stored_voxels:
            return None
        self._fresh_rows()
        stored = self.voxel_data[voxel_data_index][0]
        ids = self.voxel_data[voxel_data_index][1:min(stored + 1, self.max_points)]
        return self.added_points[ids][:, :3]

    def getStoredPointsFromVoxelDataSlice(self, slice):
        stored = slice[0]
        ids = slice[1:min(stored + 1, self.max_points)]
        return self.added_points[ids][:, :3], self.added_colors[ids]

    def getStoredVoxelCount(self):
        return self.stored_voxels

    def getVoxelDataIndexPoint(self, np_point):
        pos = self.findVoxelId_gpu(self.voxel_index, self.getVoxelCoord(np_point))
        return None if pos == -1 else self.voxel_index[pos][3]

    def getVoxelDataIndexAt(self, index_into_voxel_index):
        if index_into_voxel_index >= self.stored_voxels:
            return None
        return self.voxel_index[index_into_voxel_index][3]

    def getVoxelPositions(self, normalize=True):
        coords = self.voxel_index[:, :3]
        return coords * self.voxel_size if normalize else coords
