"""Exact nearest neighbours on the GPU for ring-less clouds and maps.

`CellGrid` stands where the reference builds a scipy KD-tree
(PointcloudIcpContainer.getKDTree, PointcloudIcpContainer.py:471-472): it is
constructed from the target points and answers `query(points)` with the same
(distances, indices) pair (:468).  The target is binned into cubic cells on the
device (Morton sort + hash of occupied cells, pcp_grid_build); every query is
exact in float64, ties go to the lowest index.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f32p, f64p

MODE_POINT, MODE_PLANE = 0, 1


def _i64p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_longlong))


def auto_cell(points):
    """Cell edge that puts a handful of points into an occupied cell of a surface-like cloud."""
    p = np.asarray(points, dtype=np.float64)
    if len(p) < 2:
        return 1.0
    ext = np.maximum(p.max(axis=0) - p.min(axis=0), 1e-9)
    return float(max((np.prod(ext) / len(p)) ** (1.0 / 3.0), float(ext.max()) * 2.0 ** -19))


class CellGrid:
    def __init__(self, np_points, cell=None, normals=None, device=0, slab=None, refine=None):
        self._lib = _lib.load()
        self._g = C.c_void_p()
        check(self._lib.pcp_grid_create(int(device), C.byref(self._g)))
        self.device = int(device)
        self.rebuild(np_points, cell, normals, slab, refine)

    def rebuild(self, np_points, cell=None, normals=None, slab=None, refine=None):
        """slab = (own_cell, lo, hi): keep only the points whose partition cell (edge own_cell) or one of
        its 26 neighbours has its key in [lo, hi) -- the share of a map split over ranks, selected on the
        device; query results keep indexing np_points.
        refine = N: the answers of an exact search do not depend on the cell size, its speed does (a query scans its
        cell and the shells around it point by point): while an occupied cell holds more than N points on average the
        cell edge is halved and the grid built again (at most four times; a build is a few sorts)."""
        if refine:
            self.rebuild(np_points, cell, normals, slab)
            for _ in range(4):
                info = self.info()
                if info["cells"] == 0 or info["points"] <= refine * info["cells"]:
                    break
                self.rebuild(np_points, self.cell / 2.0, normals, slab)
            return
        pts = np.ascontiguousarray(np.asarray(np_points, dtype=np.float64)[:, :3])
        self.data = pts                                   # scipy's KDTree.data
        self.n = len(pts)
        self.cell = float(cell) if cell is not None else auto_cell(pts)
        nrm = None
        if normals is not None:
            nrm = np.zeros((len(pts), 4), dtype=np.float32)
            nn = np.asarray(normals, dtype=np.float32)
            nrm[:, :nn.shape[1]] = nn
        if slab is None:
            check(self._lib.pcp_grid_build(self._g, f64p(pts), len(pts), self.cell, f32p(nrm)))
            self.held = len(pts)
        else:
            kept = C.c_size_t(0)
            check(self._lib.pcp_grid_build_slab(self._g, f64p(pts), len(pts), self.cell, f32p(nrm), float(slab[0]),
                                                int(slab[1]), int(slab[2]), C.byref(kept)))
            self.held = int(kept.value)
        self.m = 0

    def append(self, np_points, normals=None, first_id=None, slab=None):
        """New points join the resident map without a rebuild (pcp_grid_append): sorted among themselves,
        merged into the map's order, cell table redone.  Ids continue where the points fed so far end
        (first_id overrides), so query results index the concatenation of everything fed.  With
        slab = (own_cell, lo, hi) only this rank's share of the batch is kept (pcp_grid_append_slab).
        Returns the number of points kept."""
        pts = np.ascontiguousarray(np.asarray(np_points, dtype=np.float64)[:, :3])
        first = self.n if first_id is None else int(first_id)
        nrm = None
        if normals is not None:
            nrm = np.zeros((len(pts), 4), dtype=np.float32)
            nn = np.asarray(normals, dtype=np.float32)
            nrm[:, :nn.shape[1]] = nn
        if slab is None:
            check(self._lib.pcp_grid_append(self._g, f64p(pts), len(pts), f32p(nrm), first))
            kept = len(pts)
        else:
            k = C.c_size_t(0)
            check(self._lib.pcp_grid_append_slab(self._g, f64p(pts), len(pts), f32p(nrm), float(slab[0]), int(slab[1]),
                                                 int(slab[2]), first, C.byref(k)))
            kept = int(k.value)
        self.n = max(self.n, first + len(pts))              # ids handed out so far: what a miss is reported as
        self.held += kept
        self.data = None                                    # the host copy of the first build no longer is the map
        return kept

    def collect(self):
        """Context manager: appends made inside it are collected on the device and join the map together when it ends
        (pcp_grid_collect_begin / _end: one sort, merge and cell table for all of them); nothing may read the map meanwhile."""
        import contextlib

        @contextlib.contextmanager
        def _collect():
            check(self._lib.pcp_grid_collect_begin(self._g))
            try:
                yield self
            finally:
                check(self._lib.pcp_grid_collect_end(self._g))
        return _collect()

    def synchronize(self):
        """Returns when everything enqueued on the grid's stream has completed."""
        check(self._lib.pcp_grid_synchronize(self._g))

    def defer_appends(self, on=True):
        """While on, appends wait on the device and join the map, all at once, when it is next read
        (pcp_grid_defer_appends): an append costs a copy, the merge is paid once per reader."""
        check(self._lib.pcp_grid_defer_appends(self._g, int(bool(on))))

    def append_device(self, xyz_dev, n, first_id=None, slab=None):
        """append() for n points that already lie in this device's memory as n x 3 float64 (xyz_dev: address, e.g.
        from ComputeShader.stagedPointsDevice); whatever produced them must have completed."""
        n = int(n)
        first = self.n if first_id is None else int(first_id)
        if n == 0:
            return 0
        if slab is None:
            check(self._lib.pcp_grid_append_dev(self._g, C.c_void_p(int(xyz_dev)), n, first))
            kept = n
        else:
            k = C.c_size_t(0)
            check(self._lib.pcp_grid_append_slab_dev(self._g, C.c_void_p(int(xyz_dev)), n, float(slab[0]), int(slab[1]),
                                                     int(slab[2]), first, C.byref(k)))
            kept = int(k.value)
        self.n = max(self.n, first + n)
        self.held += kept
        self.data = None
        return kept

    def info(self):
        n, cells = C.c_ulonglong(0), C.c_ulonglong(0)
        ext = np.zeros(3, dtype=np.int32)
        passes, bms, sms = C.c_int(0), C.c_float(0), C.c_float(0)
        check(self._lib.pcp_grid_info(self._g, C.byref(n), C.byref(cells), _lib.i32p(ext), C.byref(passes),
                                      C.byref(bms), C.byref(sms)))
        return {"points": n.value, "cells": cells.value, "extent": ext.tolist(), "cell": self.cell,
                "sort_passes": passes.value, "build_ms": bms.value, "sort_ms": sms.value}

    def sorted_points(self):
        ids = np.zeros(self.held, dtype=np.int64)
        xyz = np.zeros((self.held, 3), dtype=np.float64)
        check(self._lib.pcp_grid_sorted_ids(self._g, _i64p(ids), f64p(xyz)))
        return ids, xyz

    def estimate_normals(self, radius, min_points=3, viewpoint=None, return_counts=False):
        """Normals of the target itself for clouds without scan rings (the reference's NormalShader
        needs ring neighbours): direction of least spread of the target points strictly within
        `radius` of each point, turned towards `viewpoint` if given.  Returns (n, 4) float32 rows in
        the order of the points given to the constructor (zero rows for points with fewer than
        min_points neighbours, or not held by this slab); they stay on the device for MODE_PLANE."""
        nrm = np.zeros((self.held, 4), dtype=np.float32)
        cnt = np.zeros(self.held, dtype=np.uint32)
        vp = None if viewpoint is None else np.ascontiguousarray(np.asarray(viewpoint, dtype=np.float64)[:3])
        check(self._lib.pcp_grid_estimate_normals(self._g, float(radius), int(min_points), f64p(vp), f32p(nrm),
                                                  cnt.ctypes.data_as(C.POINTER(C.c_uint))))
        ids, _ = self.sorted_points()
        out = np.zeros((self.n, 4), dtype=np.float32)
        out[ids] = nrm
        if not return_counts:
            return out
        counts = np.zeros(self.n, dtype=np.uint32)
        counts[ids] = cnt
        return out, counts

    # -- source cloud ---------------------------------------------------------------------
    def set_source(self, np_points):
        src = np.ascontiguousarray(np.asarray(np_points, dtype=np.float64)[:, :3])
        check(self._lib.pcp_grid_source(self._g, f64p(src), len(src)))
        self.m = len(src)

    def rewind_source(self):
        check(self._lib.pcp_grid_source_rewind(self._g))

    def read_source(self):
        out = np.zeros((self.m, 3), dtype=np.float64)
        check(self._lib.pcp_grid_source_read(self._g, f64p(out)))
        return out

    def apply(self, R, t, minus_t=False):
        R = np.ascontiguousarray(R, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        check(self._lib.pcp_grid_source_apply(self._g, f64p(R), f64p(t), 1 if minus_t else 0))

    # -- KDTree.query ---------------------------------------------------------------------
    def query(self, np_points=None, distance_upper_bound=np.inf, return_kernel_ms=False):
        """(distances, indices) of the nearest target point of every row, as scipy's KDTree.query
        with k=1; misses under a finite bound give (inf, n) like scipy.  Without points the
        resident source cloud is queried."""
        if np_points is not None:
            self.set_source(np_points)
        idx = np.zeros(self.m, dtype=np.int64)
        dist = np.zeros(self.m, dtype=np.float64)
        ms = C.c_float(0)
        bound = float(distance_upper_bound)
        if np.isnan(bound) or (np.isfinite(bound) and bound <= 0.0) or bound == -np.inf:
            # nothing is strictly nearer than a bound <= 0: all misses, as scipy answers (0.0 is the library's "no bound")
            dist[:], idx[:] = np.inf, self.n
            return (dist, idx, 0.0) if return_kernel_ms else (dist, idx)
        bound = bound if np.isfinite(bound) else 0.0
        check(self._lib.pcp_grid_query_resident(self._g, bound, _i64p(idx), f64p(dist), C.byref(ms)))
        idx[idx < 0] = self.n
        return (dist, idx, ms.value) if return_kernel_ms else (dist, idx)

    def nn_results(self, selected=False):
        idx = np.zeros(self.m, dtype=np.int64)
        dist = np.zeros(self.m, dtype=np.float64)
        sel = np.zeros(self.m, dtype=np.uint8) if selected else None
        check(self._lib.pcp_grid_nn_read(self._g, _i64p(idx), f64p(dist),
                                         None if sel is None else sel.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return (idx, dist, sel.astype(bool)) if selected else (idx, dist)

    # -- icp_step_KD ----------------------------------------------------------------------
    def kd_step(self, keep_fraction=0.8):
        """icp_step_KD (PointcloudIcpContainer.py:474-498) on the resident source: returns (R, t)."""
        mr, mo, W = np.zeros(3), np.zeros(3), np.zeros((3, 3))
        taken = C.c_ulonglong(0)
        check(self._lib.pcp_grid_kd_step(self._g, float(keep_fraction), f64p(mr), f64p(mo), f64p(W), C.byref(taken)))
        U, S, Vh = np.linalg.svd(W)
        R = U @ Vh
        t = mr - R @ mo
        return R, t

    # -- scan-to-map least squares ----------------------------------------------------------
    def ls(self, mode, radius):
        hb = np.zeros((8, 8), dtype=np.float32)
        check(self._lib.pcp_grid_ls(self._g, int(mode), float(radius), f32p(hb)))
        return hb[:6, :6].copy(), hb[:6, 6].copy()

    def icp_run(self, mode, radius, max_iter=10, tol=1e-6, full_iter=False):
        dx = np.zeros((max_iter, 6), dtype=np.float32)
        hb = np.zeros((max_iter, 8, 8), dtype=np.float32)
        it, cv = C.c_int(0), C.c_int(0)
        check(self._lib.pcp_grid_icp_run(self._g, int(mode), float(radius), int(max_iter), float(tol),
                                         1 if full_iter else 0, f32p(dx), f32p(hb), C.byref(it), C.byref(cv)))
        return dx[:it.value], bool(cv.value), hb[:it.value]

    def set_owned_keys(self, lo, hi, own_cell=None):
        """Only source points whose partition-cell key (cells of edge own_cell, default the grid's
        cell) lies in [lo, hi) contribute to ls / icp_run."""
        check(self._lib.pcp_grid_set_owned_keys(self._g, float(own_cell if own_cell is not None else self.cell),
                                                int(lo), int(hi)))

    def cleanup(self):
        if self._g:
            check(self._lib.pcp_grid_destroy(self._g))
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass


def partition_keys(points, own_cell, world, device=0):
    """[(lo, hi)] per rank: the Morton key space of partition cells cut into `world` contiguous ranges
    with about equal numbers of points (sorted on the device)."""
    p = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :3])
    cuts = np.zeros(int(world) + 1, dtype=np.uint64)
    check(_lib.load().pcp_grid_partition(int(device), f64p(p), len(p), float(own_cell), int(world),
                                         cuts.ctypes.data_as(C.POINTER(C.c_ulonglong))))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(int(world))]


def cell_keys(points, cell):
    """Absolute Morton key of the cell of every point (the key space ranks split)."""
    p = np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :3])
    keys = np.zeros(len(p), dtype=np.uint64)
    check(_lib.load().pcp_grid_cell_keys(float(cell), f64p(p), len(p), keys.ctypes.data_as(C.POINTER(C.c_ulonglong))))
    return keys
