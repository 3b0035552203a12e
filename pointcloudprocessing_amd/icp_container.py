"""Host driver of the registration path with the reference's call surface
(PointcloudIcpContainer.py, class PointcloudIcpContainer; file:line citations
point there).  The per-iteration methods call the stepwise kernels exactly as
the reference does; full_pt2pt / full_pt2pl run the whole loop on the device
(ComputeShader.icpRun) and only replay the 6-vector steps on the 1000-point
tracking grid on the host.
"""
import numpy as np

from .alignment import PointcloudAlignment
from .compute_shader import MODE_PLANE, MODE_POINT


class PointcloudIcpContainer:

    def __init__(self, computeShader, pointcloudAlignment=None):
        self.compute = computeShader
        self.pointcloudAlignment = pointcloudAlignment
        # Opt-in: full_pt2pt / full_pt2pl skip scan lines, upload and normals when called again with
        # the SAME array object as `aligned` (and the same origin and mode) and nothing else has
        # uploaded a target to the compute object in between.  The caller promises not to change
        # that array in place.  The reference prepares every call (PointcloudIcpContainer.py:72,104).
        self.keep_prepared_target = False
        self._kept = None
        # preparePointToPoint / preparePointToPlane of a float64 cloud run their host passes (scan lines,
        # padding, cast) on the device; False keeps them in NumPy as the reference has them (same results)
        self.device_prepare = True

    # ------------------------------------------------------------------ scan lines (:349-392)
    def sphereProjectPont(self, point, origin):
        off = np.array([0, 0.254 / 2, 0])
        v = point - origin - off
        return v / np.linalg.norm(v)

    def sphereProjectPoints(self, np_points, origin):
        off = np.array([0, 0.254 / 2, 0])
        v = np_points - origin - off
        return v / np.linalg.norm(v, axis=1)[:, np.newaxis]

    def getScanLines(self, pts, origin):
        """:362-392 -- a ring ends where the azimuth wraps; the points after the
        last wrap belong to no ring."""
        proj = self.sphereProjectPoints(pts, origin)
        angs = np.arctan2(proj[:, 2], proj[:, 0])
        angs[angs < 0] += np.pi * 2
        wraps = np.abs(angs[1:] - angs[:-1]) > 6
        ends = np.nonzero(wraps)[0]
        scan_lines, begin = [], 0
        for e in ends:
            scan_lines.append((begin, e))
            begin = e + 1
        return scan_lines

    # ------------------------------------------------------------------ stepwise protocol
    def _on_device(self, points_a):
        """True when the reference cloud can be prepared by the device (ComputeShader.prepareTargetOnDevice:
        ring segmentation, padding and cast there instead of in NumPy): a float64 (n, 3+) array that fits."""
        return (self.device_prepare and hasattr(self.compute, "prepareTargetOnDevice") and isinstance(points_a, np.ndarray)
                and points_a.dtype == np.float64 and points_a.ndim == 2 and points_a.shape[1] >= 3
                and 0 < len(points_a) <= getattr(self.compute, "maxPointsPerCloud", 0))

    def preparePointToPoint(self, points_a, origin, points_b):
        """:20-22"""
        if self._on_device(points_a):
            self.compute.prepareTargetOnDevice(points_a, points_b, origin, False)
            return
        self.compute.prepareLS(points_a, self.getScanLines(points_a, origin), points_b, origin, False)

    def dispatchPointToPoint(self, points_b):
        """:24-29"""
        H, b = self.compute.dispatchLS(points_b)
        delta_x = np.linalg.solve(H, -b)
        t, r = delta_x[:3], delta_x[3:]
        return t, PointcloudAlignment.rotation(r[0], r[1], r[2])

    def preparePointToPlane(self, points_a, origin, points_b):
        """:31-39"""
        if self._on_device(points_a):
            self.compute.prepareTargetOnDevice(points_a, points_b, origin, True)
            return
        self.compute.preparePointPlane(points_a, self.getScanLines(points_a, origin), points_b, origin, False)

    def dispatchPointToPlane(self, points_b):
        """:47-69 -- a singular system raises LinAlgError (the reference prints
        and then fails on the unbound delta_x)."""
        H, b = self.compute.dispatchPointPlane(points_b)
        delta_x = np.linalg.solve(H, -b)
        t, r = delta_x[:3], delta_x[3:]
        return t, PointcloudAlignment.rotation(r[0], r[1], r[2]), np.mean(np.abs(delta_x))

    def initLS(self, aligned, to_align, origin):
        """:146-149"""
        self.compute.prepareLS(aligned, self.getScanLines(aligned, origin), to_align, origin)

    def prepareNNS(self, points_a, origin, points_b):
        """:151-153"""
        self.compute.prepareNNS(points_a, self.getScanLines(points_a, origin), points_b, origin)

    def dispatchNNS(self, points_b):
        """:155-156"""
        return self.compute.dispatchNNS(points_b)

    def getHsBsCompute(self, to_align):
        """:178-179"""
        return self.compute.dispatchLS(to_align)

    def apply_iteration_of_ls(self, to_align, t, R):
        """:172-174"""
        return (R @ to_align.T).T - t

    def applyIcpStep(self, np_points, R, t):
        """:320-322"""
        return (R @ np_points.T).T + t

    # ------------------------------------------------------------------ full loops
    def _replay(self, dx_trace, applied):
        grid = self.getUniformGrid(10)
        for dx in dx_trace[:applied]:
            R = PointcloudAlignment.rotation(dx[3], dx[4], dx[5])
            grid = self.applyIcpStep(grid, R, dx[:3])
        R_opt, t_opt = self.uniform_optimal_icp(grid, self.getUniformGrid(10))
        return t_opt, R_opt

    def _pose(self, dx_trace, applied):
        """(t_opt, R_opt) of the loop that just ran: the device's (ComputeShader.fetchPose: the tracking grid is
        replayed and the Kabsch extraction done behind the loop); the host replay for compute objects without it."""
        if hasattr(self.compute, "fetchPose"):
            return self.compute.fetchPose()
        return self._replay(dx_trace, applied)

    def _keep(self, aligned, origin, mode):
        self._kept = (aligned, np.array(origin, dtype=np.float64), mode, getattr(self.compute, "target_serial", None))

    def _target_kept(self, aligned, origin, mode, to_align):
        """True when the prepared target on the device is still `aligned` (keep_prepared_target);
        the ring hints are then reset as a prepare would."""
        k = self._kept
        if not (self.keep_prepared_target and k is not None and k[0] is aligned and k[2] == mode
                and k[3] is not None and k[3] == getattr(self.compute, "target_serial", None)
                and np.array_equal(k[1], np.asarray(origin, dtype=np.float64))):
            return False
        self.compute.resetHints(len(to_align))
        return True

    def full_pt2pt(self, aligned, to_align, origin, iterations=20, renderer=None):
        """:71-100 -> (t_opt, R_opt); exactly `iterations` steps."""
        if not self._target_kept(aligned, origin, "point", to_align):
            self.preparePointToPoint(aligned, origin, to_align)
            self._keep(aligned, origin, "point")
        if renderer is not None:
            return self._full_stepwise(to_align, iterations, renderer, plane=False, full_iter=True)
        dx, _, _, _ = self.compute.icpRun(to_align, MODE_POINT, iterations)
        return self._pose(dx, len(dx))

    def full_pt2pl(self, aligned, to_align, origin, iterations=20, renderer=None, full_iter=False):
        """:103-144 -> (t_opt, R_opt); stops when mean|dx| < 1e-6 unless full_iter."""
        if not self._target_kept(aligned, origin, "plane", to_align):
            self.preparePointToPlane(aligned, origin, to_align)
            self._keep(aligned, origin, "plane")
        if renderer is not None:
            return self._full_stepwise(to_align, iterations, renderer, plane=True, full_iter=full_iter)
        dx, converged, _, _ = self.compute.icpRun(to_align, MODE_PLANE, iterations, tol=1e-6, full_iter=full_iter)
        return self._pose(dx, len(dx) - 1 if converged else len(dx))

    def _full_stepwise(self, to_align, iterations, renderer, plane, full_iter):
        """The reference's host loop, one dispatch per iteration, for callers
        that want to draw every intermediate cloud."""
        grid = self.getUniformGrid(10)
        prev = renderer.addPoints(to_align, np.array([0, 0, 255]))
        for _ in range(iterations):
            if plane:
                t, R, delta = self.dispatchPointToPlane(to_align)
                if delta < 1e-6 and not full_iter:
                    break
            else:
                t, R = self.dispatchPointToPoint(to_align)
            to_align = self.apply_iteration_of_ls(to_align, t, R)
            if prev is not None:
                renderer.freePoints(prev)
            prev = renderer.addPoints(to_align)
            grid = self.applyIcpStep(grid, R, t)
        if plane and prev is not None:
            renderer.freePoints(prev)
        R_opt, t_opt = self.uniform_optimal_icp(grid, self.getUniformGrid(10))
        return t_opt, R_opt

    # ------------------------------------------------------------------ pose extraction (:333-347, :429-461)
    def uniform_optimal_icp(self, grid1, grid2):
        mean_ref = np.mean(grid1, axis=0)
        mean_other = np.mean(grid2, axis=0)
        W = (grid1 - mean_ref).T @ (grid2 - mean_other)
        U, S, Vh = np.linalg.svd(W)
        R = U @ Vh
        return R, mean_ref - R @ mean_other

    def getUniformGrid(self, pointsPerAxis):
        ax = np.linspace(-1, 1, pointsPerAxis)
        X, Y, Z = np.meshgrid(ax, ax, ax, indexing='ij')
        return np.vstack([X.ravel(), Y.ravel(), Z.ravel()]).T

    def getUniformShape(self):
        g = self.getUniformGrid(10)
        return np.vstack([g, g + np.array([1, 0, 0]) * 5, g + np.array([0, 1, 0]) * 5, g + np.array([1, 1, 1]) * 5])

    # ------------------------------------------------------------------ NumPy J / H / b (:181-205, :267-287)
    def getJ(self, points):
        n = points.shape[0]
        x, y, z = points[:, 0], points[:, 1], points[:, 2]
        J = np.zeros((n, 3, 6), dtype=np.float32)
        J[:, 0, 0] = 1; J[:, 0, 4] = -z; J[:, 0, 5] = y
        J[:, 1, 1] = 1; J[:, 1, 3] = z; J[:, 1, 5] = -x
        J[:, 2, 2] = 1; J[:, 2, 3] = -y; J[:, 2, 4] = x
        return J

    def getHsBsNumpy(self, aligned, to_align):
        J = self.getJ(to_align)
        J_T = J.swapaxes(1, 2)
        e = (aligned - to_align)[:, :, np.newaxis]
        return np.matmul(J_T, J), np.matmul(J_T, e).squeeze(-1)

    def sumHsBsNumpy(self, Hs, Bs):
        return np.sum(Hs, axis=0), np.sum(Bs, axis=0)

    def icp_step_LS_vector(self, aligned, to_align):
        H, b = self.sumHsBsNumpy(*self.getHsBsNumpy(aligned, to_align))
        delta_x = np.linalg.solve(H, -b)
        return delta_x[:3], delta_x[3:]

    # ------------------------------------------------------------------ one point at a time (:207-314)
    # The scalar statement of what LeastSquaresPoint.glsl computes per point (getH / getB, :20-43 there):
    # float64, explicit sums, kept for checking the vectorised and the device versions against.
    def getJ_i(self, point):
        """:289-299 -- rows (1,0,0,0,-z,y), (0,1,0,z,0,-x), (0,0,1,-y,x,0)."""
        x, y, z = float(point[0]), float(point[1]), float(point[2])
        return np.array([[1.0, 0.0, 0.0, 0.0, -z, y],
                         [0.0, 1.0, 0.0, z, 0.0, -x],
                         [0.0, 0.0, 1.0, -y, x, 0.0]])

    def getH_i(self, J_i):
        """:254-265 -- H[i][j] = sum over the three rows k of J[k][i] * J[k][j], in that order."""
        H_i = np.zeros((6, 6))
        for i in range(6):
            for j in range(6):
                acc = 0.0
                for k in range(3):
                    acc += J_i[k][i] * J_i[k][j]
                H_i[i][j] = acc
        return H_i

    def getH_i_vector(self, J_i):
        """:251-252"""
        return J_i.T @ J_i

    def getB(self, J_i, e_i):
        """:301-314 -- b[i] = sum over k of J[k][i] * e[k]."""
        b_i = np.zeros(6)
        for i in range(6):
            acc = 0.0
            for k in range(3):
                acc += J_i[k][i] * e_i[k]
            b_i[i] = acc
        return b_i

    def addH(self, H, H_i):
        """:236-242"""
        return np.array([[H[i][j] + H_i[i][j] for j in range(6)] for i in range(6)], dtype=np.float64)

    def addB(self, B, b_i):
        """:244-249"""
        return np.array([B[i] + b_i[i] for i in range(6)], dtype=np.float64)

    def getHsBsLoop(self, aligned, to_align):
        """:207-221 -- per point i: e = aligned[i] - to_align[i], J of to_align[i], H_i and b_i."""
        Hs, Bs = [], []
        for p_a, p_b in zip(aligned[:len(to_align)], to_align):
            J_i = self.getJ_i(p_b)
            Hs.append(self.getH_i(J_i))
            Bs.append(self.getB(J_i, p_a - p_b))
        return np.array(Hs), np.array(Bs)

    def sumHsBsLoop(self, Hs, Bs):
        """:223-231 -- running sums in point order."""
        H, b = np.zeros((6, 6)), np.zeros(6)
        for H_i, b_i in zip(Hs, Bs):
            H, b = self.addH(H, H_i), self.addB(b, b_i)
        return H, b

    # ------------------------------------------------------------------ geometry helpers of the reference's demos
    def dist(self, p1, p2):
        """:396-397"""
        return np.linalg.norm(p1 - p2)

    def getUniformGridSurface(self, pointsPerAxis):
        """:440-453 -- the points of getUniformGrid that lie on the cube's faces."""
        g = self.getUniformGrid(pointsPerAxis)
        return g[np.any(np.abs(g) == 1.0, axis=1)]

    def get_plane_grid(self, points_per_axis, y_value=0, gridsize=10.0):
        """:524-535 -- a square grid in the plane y = y_value."""
        axis = np.linspace(-1, 1, points_per_axis) * gridsize
        X, Z = np.meshgrid(axis, axis, indexing="ij")
        return np.vstack([X.ravel(), np.full(X.size, y_value, dtype=X.dtype), Z.ravel()]).T

    def connectPointsInOrder(self, np_points):
        """:517-522 -- line segments (p0, p1), (p1, p2), ... as consecutive row pairs."""
        out = np.empty((2 * (len(np_points) - 1), 3), dtype=np_points.dtype)
        out[0::2], out[1::2] = np_points[:-1], np_points[1:]
        return out

    def zipPointsToLines(self, points1, points2):
        """:325-330 -- segments (points1[i], points2[i]); like the reference, the last common pair is left out."""
        n = min(len(points1), len(points2)) - 1
        out = np.empty((2 * n, 3), dtype=points1.dtype)
        out[0::2], out[1::2] = points1[:n], points2[:n]
        return out

    # ------------------------------------------------------------------ legacy exact-NN ICP (:465-515)
    def getKDTree(self, np_points, cell=None):
        """:471-472.  The tree is a device cell grid answering the same query()."""
        from .cell_grid import CellGrid
        device = self.compute.device if self.compute is not None and hasattr(self.compute, "device") else 0
        return CellGrid(np_points, cell=cell, device=device)

    def getNearestNeighbours(self, kdTree_pc1, pc1, pc2):
        """:467-469."""
        distances, closestIndexes = kdTree_pc1.query(pc2)
        return np.asarray(pc1)[closestIndexes], pc2, distances

    def icp_step_KD(self, kdTree_pc1, pc1, pc2):
        """:474-498: exact neighbours, best 80 % by distance, Kabsch.  pc1 is the cloud the
        grid was built from; the sums run on the device over the resident copy of pc2."""
        kdTree_pc1.set_source(pc2)
        return kdTree_pc1.kd_step(0.8)

    def align_KD(self, reference_points, points_to_align, iterations=20):
        """:500-515, the moving cloud staying on the device between the steps."""
        tree = self.getKDTree(reference_points)
        try:
            tree.set_source(points_to_align)
            reference_grid = self.getUniformGrid(10)
            for _ in range(iterations):
                R, t = tree.kd_step(0.8)
                tree.apply(R, t)
                reference_grid = self.applyIcpStep(reference_grid, R, t)
            moved = tree.read_source()
        finally:
            tree.cleanup()
        R_opt, t_opt = self.uniform_optimal_icp(reference_grid, self.getUniformGrid(10))
        return moved, R_opt, t_opt

    def align_ringless_pt2pl(self, reference_points, points_to_align, radius, iterations=20, viewpoint=None,
                             normal_radius=None, cell=None, full_iter=False):
        """full_pt2pl (:103-144) for clouds without scan rings (PLY models, accumulated maps), where
        neither the ring search nor NormalShader.glsl applies: exact neighbours within `radius` on the cell
        grid of align_KD, normals from the reference cloud itself (least spread of the points within
        normal_radius, default = radius), the tiles and the sum of LeastSquaresPlane.glsl / ParalellHBSum.glsl,
        the whole loop on the device.  Returns (moved points, t_opt, R_opt) with full_pt2pl's pose convention
        (p <- R_opt p - t_opt).  No counterpart in the reference."""
        normal_radius = float(radius if normal_radius is None else normal_radius)
        tree = self.getKDTree(reference_points, cell=cell if cell is not None else max(float(radius), normal_radius))
        try:
            tree.estimate_normals(normal_radius, viewpoint=viewpoint)
            tree.set_source(points_to_align)
            dx, converged, _ = tree.icp_run(1, float(radius), max_iter=int(iterations), full_iter=full_iter)
            moved = tree.read_source()
        finally:
            tree.cleanup()
        t_opt, R_opt = self._replay(dx, len(dx) - (1 if converged else 0))
        return moved, t_opt, R_opt
