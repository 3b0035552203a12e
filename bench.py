"""Headline benchmark: ICP frame-pairs/s (and the NN queries/s inside them) on ~120k-point KITTI scans.

    python bench.py --gpus N --steps K --warmup W

Workload `sequence` (default): frame-to-frame odometry over real consecutive KITTI sweeps
(2011_09_26 drive, all thirty frames 0000000000 ... 0000000029 of the reference's sample_data, held as fixtures;
walked forwards and backwards so that every pair is a consecutive one).  One STEP = every rank registers
a run of --pairs + 1 raw frames = --pairs consecutive pairs by point-to-plane ICP with ring normals,
at most 10 iterations, stop at mean|dx| < 1e-6 (PointcloudIcpContainer.full_pt2pl as UI.py:400 calls
it), starting from the raw velodyne records in page-locked HOST memory: the timed region holds the
upload of every frame (copy stream, overlapped with the previous step's kernels), decode + ego filter
+ rotation, ring segmentation, unit vectors, ring bands, normals, the whole iteration loop of every
pair, the pose of every pair (tracking grid + Kabsch on the device) and the fetch of steps and poses.  Nothing of a
step is resident beforehand.
Pairs are independent, so ranks share nothing: weak scaling, no collective in the data path
(torch.distributed only supplies the barrier and the max).

Workload `moved-copy` is an earlier configs[1] substitute (frame 0 against rigidly moved +
noise copies, inputs resident), kept for comparison.

Extra objects on the line (N = 1 only): `odometry_e2e` (the driver's chained loop on the device beside the host-driven
one), `nn` (stand-alone NearestNeighbourScan), `pt2pt20` (configs[0]), `ringless_pt2pl` (configs[1] without scan rings),
`voxel_sweep` (configs[2]), `single_pair` (latency of one registration), `imu_guess`; for N > 1 `scan_to_map` (configs[4]).
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointcloudprocessing_amd import (MODE_PLANE, ComputeShader, IcpBatch, PinnedFrames,  # noqa: E402
                                      PointcloudIcpContainer, SequencePipeline, lidar)
from pointcloudprocessing_amd._lib import check, f64p  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
MAX_ITER = 10
TOL = 1e-6
GOLD = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------------- inputs
def real_frames():
    """The thirty raw sweeps (N, 4) float32; a synthetic drive of the same shape when the fixtures are absent."""
    pair, seq, seq2 = (os.path.join(GOLD, n) for n in ("kitti_pair.npz", "kitti_seq.npz", "kitti_seq2.npz"))
    if os.path.exists(pair) and os.path.exists(seq):
        g = np.load(pair)
        raws = [g["raw0"], g["raw1"]] + lidar.unpack_raw_frames(np.load(seq))
        if os.path.exists(seq2):
            raws += lidar.unpack_raw_frames(np.load(seq2))
        name = "kitti 2011_09_26 frames 0000000000-%010d (fixtures), consecutive pairs" % (len(raws) - 1)
    else:
        back = lidar.aligning_rotation().T
        raws = [(back @ lidar.synthetic_scan(k, pose_t=np.array([0.0, 0.0, -0.33 * k])).T).T.astype(np.float32)
                for k in range(12)]
        name = "synthetic 64-ring drive, 0.33 m per sweep"
    return [np.ascontiguousarray(np.column_stack([r, np.zeros(len(r), np.float32)]), dtype=np.float32) for r in raws], name


def ping_pong(n, distinct, start):
    """n frame indices walking 0 .. distinct-1 .. 0 .. from `start`: neighbours are always consecutive frames."""
    period = 2 * (distinct - 1)
    out = []
    for j in range(start, start + n):
        k = j % period
        out.append(k if k < distinct else period - k)
    return out


def moved_copy(a, seed):
    """frame vs. R*frame + t + N(0, 0.01^2): the expected registration is the inverse motion."""
    rng = np.random.default_rng(seed)
    s = 1.0 + 0.05 * (seed % 7)
    R = lidar.rotation_xyz(0.02 * s, -0.01 * s, 0.015 * s)
    t = np.array([0.05, -0.02, 0.03]) * s
    return (R @ a.T).T + t + rng.normal(0.0, 0.01, a.shape), R, t


def traffic_for(workload, pairs, n_b):
    """HBM bytes per launch of the dominant kernel from the counter passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, one pass each, which cannot run inside this process): their committed summary is quoted
    when it was taken on this very workload, else None."""
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[::-1]:
        tr = json.load(open(tf))
        if (tr.get("workload"), tr.get("pairs_per_launch")) == (workload, pairs) and abs(tr.get("n_b", 0) - n_b) < 2000:
            return tr["fetch_bytes_per_launch"] + tr["write_bytes_per_launch"]
    return None


def issue_pipes(workload, pairs):
    """How busy the two issue pipes the fused kernel leans on were, from the committed summary of the counter passes of
    this workload (rocprofv3 --pmc cannot run inside this process; tools/profile_r02.sh takes them)."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue_pipes.json")))[::-1]:
        d = json.load(open(f))
        if (d.get("workload"), d.get("pairs_per_launch")) == (workload, pairs):
            return {k: d[k] for k in ("valu_busy", "ta_busy_avg", "ta_busy_max", "waves_waiting", "l2_hit_rate", "how", "source")}
    return None


# ----------------------------------------------------------------------------------- workloads
def run_sequence(args, rank, world, device, barrier):
    raws, source = real_frames()
    pinned = PinnedFrames(raws)
    order = ping_pong(args.pairs + 1, len(raws), 3 * rank)
    frames = [pinned[k] for k in order]
    cap = 1 << 17
    while cap < max(len(r) for r in raws):
        cap <<= 1
    pipe = SequencePipeline(args.pairs + 1, max_points=cap, device=device)
    imu_file = os.path.join(GOLD, "ref_imu.npz")
    placed = None
    if os.path.exists(imu_file) and source.startswith("kitti"):
        # every frame's placement from its OXTS record, as the reference's driver places a frame before it refines it
        # (EnvironmentConstructor.py:169, :200-209); a pair walked backwards is guessed to undo the forward motion
        from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp
        from pointcloudprocessing_amd.sequence import imu_placements
        g = np.load(imu_file)
        rd = OxtsDataReader()
        for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
            rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
        rd.finish()
        place = imu_placements([rd.getOx(k) for k in rd.getfilenames()[:len(raws)]])
        placed = []
        for f, k in enumerate(order):
            back = f > 0 and order[f - 1] > k
            placed.append((place[k][0], np.zeros(3) if f == 0 else (-place[order[f - 1]][1] if back else place[k][1])))
    main_placements = placed if args.workload == "sequence-imu" else None
    if args.workload == "sequence-imu" and placed is None:
        raise SystemExit("--workload sequence-imu needs the KITTI fixtures and their OXTS records")

    def run(steps):
        # every collect fetches the pairs' steps AND their poses (t_opt, R_opt: tracking grid + Kabsch on the device,
        # one record array per run): the poses are on the host when the clock stops
        last, pending = None, None
        for _ in range(steps):
            t = pipe.submit(frames, point_to_plane=True, iterations=MAX_ITER, tol=TOL, placements=main_placements)
            if pending is not None:
                last = pipe.collect_with_poses(pending)
            pending = t
        if pending is not None:
            last = pipe.collect_with_poses(pending)
        return last

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps, n_pts, poses = run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0

    # a sanity check on the poses the timed steps delivered: a pair and the same pair walked backwards
    # must be inverse motions, and the car moves a few decimetres per sweep
    seen = {}
    for p, (t, R) in enumerate(poses):
        seen[(order[p], order[p + 1])] = (t, R)
        step_len = np.linalg.norm(t)
        # (with the IMU's guess applied first, what is left to find is centimetres)
        if not ((0.0 if main_placements else 0.05) < step_len < 2.0) and not os.environ.get("PCP_BENCH_NOCHECK"):
            raise SystemExit(f"INVALID: pair {order[p]}->{order[p + 1]} moved {step_len:.3f} m")
    for (i, j), (t, R) in seen.items():
        if (j, i) in seen and not os.environ.get("PCP_BENCH_NOCHECK"):
            tb, Rb = seen[(j, i)]
            if np.abs(R @ Rb - np.eye(3)).max() > 2e-3 or np.abs(R @ tb + t).max() > 3e-2:
                raise SystemExit(f"INVALID: pairs {i}->{j} and {j}->{i} are not inverse motions")
    iters = [len(dx) for dx, _ in steps]
    n_b = int(np.mean(n_pts[1:]))
    queries = float(sum(it * n for it, n in zip(iters, n_pts[1:]))) * world * args.steps

    # roofline leg: the same steps once more, one at a time, with a HIP event before/after every fused
    # NN + least-squares launch on the stream those launches run on
    # The first launch of a run searches all rings and is a kernel of its own (k_ls_batch_first); the dominant kernel is
    # the hinted one (k_ls_batch, every later launch): the two are timed and counted apart.
    pipe.profile(True)
    first_ms, first_n, first_points = 0.0, 0, 0
    ls_ms, ls_launches, ls_points, all_launches = 0.0, 0, 0, 0
    reps = max(1, min(args.steps, 5))
    for _ in range(reps):
        p_steps, p_pts = pipe.collect(pipe.submit(frames, point_to_plane=True, iterations=MAX_ITER, tol=TOL,
                                                  placements=main_placements))
        each = pipe.profile_read_launches()
        p_iters = [len(dx) for dx, _ in p_steps]
        first_ms, first_n, first_points = first_ms + each[0], first_n + 1, first_points + int(sum(p_pts[1:]))
        ls_ms, ls_launches = ls_ms + float(sum(each[1:])), ls_launches + len(each) - 1
        ls_points += int(sum((it - 1) * n for it, n in zip(p_iters, p_pts[1:])))
        all_launches += len(each)
    pipe.profile(False)
    # algorithmic bytes of one launch = 32 B per source point it processed (16 B point + 8 B hint r/w + 8 B
    # result) + the 32 B per target point (point + normal) each pair reads once, spread over all its launches
    target_share = 32.0 * float(sum(n_pts[:-1])) * reps / max(all_launches, 1)
    alg_bytes = 32.0 * ls_points / max(ls_launches, 1) + target_share
    avg_launch_s = ls_ms * 1e-3 / max(ls_launches, 1)
    first_launch = {"kernel": "k_ls_batch_first<plane> (iteration 0: every ring searched, dense pooled passes)",
                    "avg_launch_us": first_ms * 1e3 / max(first_n, 1), "launches_timed": first_n,
                    "algorithmic_bytes_per_launch": 32.0 * first_points / max(first_n, 1) + target_share}
    first_launch["achieved_gb_s"] = first_launch["algorithmic_bytes_per_launch"] / (first_launch["avg_launch_us"] * 1e-6) / 1e9

    # latency of one pair from its raw bytes in host memory to its steps on the host, nothing else in flight
    two = [pinned[0], pinned[1]]
    lat = [0.0] * 4
    for _ in range(0 if args.no_latency else 13):
        c0 = time.perf_counter()
        pipe.collect(pipe.submit(two, point_to_plane=True, iterations=MAX_ITER, tol=TOL))
        lat.append(time.perf_counter() - c0)
    raw_latency = float(np.median(lat[3:]) * 1e3)

    # the same run with every pair's initial guess taken from the IMU: an extra object, not part of `value`
    guessed = None
    if placed is not None and main_placements is None and not args.no_latency:
        reps_g = max(1, min(args.steps, 5))

        def run_guessed(n):
            last, pending = None, None
            for _ in range(n):
                t = pipe.submit(frames, point_to_plane=True, iterations=MAX_ITER, tol=TOL, placements=placed)
                if pending is not None:
                    last = pipe.collect(pending)
                pending = t
            return pipe.collect(pending)

        run_guessed(1)
        c0 = time.perf_counter()
        g_steps, _ = run_guessed(reps_g)
        g_elapsed = time.perf_counter() - c0
        guessed = {
            "what": "the same step with every source placed by its OXTS record first (IMU rotation of both frames, "
                    "dead-reckoned offset of the source), as calculateTransition_imu does before it refines",
            "pairs_per_s": args.pairs * reps_g / g_elapsed, "ms_per_step": g_elapsed / reps_g * 1e3,
            "iterations_per_pair": float(np.mean([len(dx) for dx, _ in g_steps])),
            "median_first_step_m": float(np.median([np.linalg.norm(dx[0][:3]) for dx, _ in g_steps])),
            "median_first_step_m_unguessed": float(np.median([np.linalg.norm(dx[0][:3]) for dx, _ in steps])),
        }
    pipe.close()
    out = {
        "value": args.pairs * world * args.steps / elapsed, "elapsed": elapsed, "queries": queries,
        "data": "real sweeps: " + source, "n_a": int(np.mean(n_pts[:-1])), "n_b": n_b, "iters": float(np.mean(iters)),
        "workload": "configs[3] on one GPU's share (the config the metric is quoted on): frame-to-frame point-to-plane ICP "
                    "with ring normals over consecutive real KITTI sweeps, <= 10 iterations, from raw bytes in host memory "
                    "to every pair's pose (t_opt, R_opt) on the host (upload, decode, ring segmentation, normals, loop, "
                    "tracking grid + Kabsch, fetch all inside the clock)"
                    + ("; every frame placed by its OXTS record first (IMU rotation, dead-reckoned offset of the source)"
                       if main_placements else ""),
        "kernel": "k_ls_batch<plane> (fused ring-NN + JtJ/Jte + level-1 tree sum, all pairs of the step in one launch; the "
                  "hinted launches, i.e. every iteration but the first)",
        "alg_bytes": alg_bytes, "avg_launch_s": avg_launch_s, "ls_launches": ls_launches, "first_launch": first_launch,
        "traffic": None if main_placements else traffic_for("sequence", args.pairs, n_b), "raw_latency_ms": raw_latency,
        "imu_guess": guessed,
        "frames": (raws, pinned),
    }
    return out


class Pair:
    def __init__(self, device, a, lines, b):
        self.cs = ComputeShader(maxPointsPerCloud=max(len(a), len(b)), maxScans=100, device=device)
        self.lib, self.ctx = self.cs._lib, self.cs._ctx
        self.icp = PointcloudIcpContainer(self.cs)
        self.n_a, self.n_b = len(a), len(b)
        self.cs.preparePointPlane(a, lines, b, np.zeros(3))          # uploads the target
        b64 = np.ascontiguousarray(b, dtype=np.float64)
        check(self.lib.pcp_upload_source(self.ctx, f64p(b64), len(b64)))
        check(self.lib.pcp_synchronize(self.ctx))

    def result(self):
        dx = np.zeros((MAX_ITER, 6), dtype=np.float32)
        it, cv = C.c_int(0), C.c_int(0)
        check(self.lib.pcp_icp_fetch(self.ctx, dx.ctypes.data_as(C.POINTER(C.c_float)), None,
                                     C.byref(it), C.byref(cv), None))
        return dx[:it.value], bool(cv.value)


def run_moved_copy(args, rank, world, device, barrier):
    """The moved-copy workload: KITTI frame 0 against R*frame + t + N(0, 1 cm) copies of itself; target, scan
    lines and float64 source resident before the timed region; one step = batched prepare + batched loop."""
    raws, source = real_frames()
    a = lidar.frame_from_raw(raws[0][:, :3])
    lines = PointcloudIcpContainer(None).getScanLines(a, np.zeros(3))
    pairs, truth = [], []
    for p in range(args.pairs):
        b, R, t = moved_copy(a, 1000 * rank + p)
        pairs.append(Pair(device, a, lines, b))
        truth.append((R, t))
    batch = IcpBatch([p.cs for p in pairs])

    def step():
        batch.prepare(True)
        batch.launch(MODE_PLANE, MAX_ITER, TOL, False)

    for _ in range(args.warmup):
        step()
    batch.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    batch.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    iters = []
    for p, (R, t) in zip(pairs, truth):
        dx, conv = p.result()
        t_opt, R_opt = p.icp._replay(dx, len(dx) - 1 if conv else len(dx))
        iters.append(len(dx))
        err_R, err_t = np.abs(R_opt @ R - np.eye(3)).max(), np.abs(R_opt @ t - t_opt).max()
        if not (err_R < 2e-3 and err_t < 2e-2) and not os.environ.get("PCP_BENCH_NOCHECK"):
            raise SystemExit(f"INVALID: pair did not register (rotation err {err_R:.2e}, translation err {err_t:.2e})")
    n_a, n_b = pairs[0].n_a, pairs[0].n_b
    batch.profile(True)
    ls_ms, ls_launches, ls_points = 0.0, 0, 0
    reps = max(1, min(args.steps, 5))
    for _ in range(reps):
        step()
        ms, n, pts = batch.profile_read()
        ls_ms, ls_launches, ls_points = ls_ms + ms, ls_launches + n, ls_points + pts
    batch.profile(False)
    batch.close()
    for p in pairs:
        p.cs.cleanup()
    return {
        "value": args.pairs * world * args.steps / elapsed, "elapsed": elapsed,
        "queries": float(sum(iters)) * n_b * world * args.steps,
        "data": "synthetic motion + noise on frame 0 of: " + source, "n_a": n_a, "n_b": n_b, "iters": float(np.mean(iters)),
        "workload": "configs[1] substitute: point-to-plane ICP with ring normals, <= 10 iterations, frame vs rigidly moved "
                    "+ N(0,1cm) copy, inputs resident before the clock",
        "kernel": "k_ls_batch<plane> (fused ring-NN + JtJ/Jte + level-1 tree sum, all pairs of the step in one launch)",
        "alg_bytes": (32.0 * ls_points + 32.0 * n_a * args.pairs * reps) / max(ls_launches, 1),
        "avg_launch_s": ls_ms * 1e-3 / max(ls_launches, 1), "ls_launches": ls_launches,
        "traffic": traffic_for("moved-copy", args.pairs, n_b), "raw_latency_ms": None, "frames": (raws, None),
    }


# ----------------------------------------------------------------------------------- extras (rank 0, one GPU)
def extras(raws, device):
    """Driver-visible numbers for the other halves of the metric: stand-alone NN scan, configs[0], configs[2]."""
    from pointcloudprocessing_amd import Voxelizer
    from pointcloudprocessing_amd.cell_grid import CellGrid
    out = {}
    a, b = lidar.frame_from_raw(raws[0][:, :3]), lidar.frame_from_raw(raws[1][:, :3])
    zero = np.zeros(3)
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100, device=device)
    icp = PointcloudIcpContainer(cs)
    ms = C.c_float(0)

    # --- stand-alone NearestNeighbourScan (ComputeShader.py:731-759): first call searches all rings, later ones the hinted four
    def nn_call():
        c0 = time.perf_counter()
        cs.dispatchNNS(b)
        wall = time.perf_counter() - c0
        check(cs._lib.pcp_nn_kernel_ms(cs._ctx, C.byref(ms)))
        return wall, ms.value * 1e-3

    first, hinted = [], []
    for _ in range(5):
        icp.prepareNNS(a, zero, b)
        first.append(nn_call())
        hinted.append(nn_call())
    fw, fk = np.median([w for w, _ in first]), np.median([k for _, k in first])
    hw, hk = np.median([w for w, _ in hinted]), np.median([k for _, k in hinted])
    nn_bytes = 32.0 * len(b) + 16.0 * len(a)          # 16 B query + 8 B result + 8 B hint r/w per query, the target once
    out["nn"] = {
        "what": "prepareNNS + dispatchNNS on real pair 0 -> 1 (%d queries against %d points, 100 padded rings)" % (len(b), len(a)),
        "all_rings": {"queries_per_s_kernel": len(b) / fk, "queries_per_s_call": len(b) / fw, "kernel_us": fk * 1e6,
                      "call_ms": fw * 1e3},
        "hinted": {"queries_per_s_kernel": len(b) / hk, "queries_per_s_call": len(b) / hw, "kernel_us": hk * 1e6,
                   "call_ms": hw * 1e3},
        "roofline": {"bound": "hbm", "kernel": "k_nn (hinted; one launch = one pair, so latency- not bandwidth-limited)",
                     "achieved": nn_bytes / hk / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": nn_bytes / hk / 1e9 / HBM_PEAK_GBS, "traffic": None},
        "note": "call = the reader's float64 cloud in host memory in, indices and distances in host arrays out (cast + pad "
                "through a page-locked buffer in chunks, 2 MB up, 1.5 MB down, PCIe inclusive); the PCIe transfers alone are "
                "about 65 us of a call",
    }

    # --- configs[0]: consecutive pair, voxelize + NN + point-to-point ICP x 20
    t_vox, voxels = None, 0
    for _ in range(3):                       # a fresh map each time; the first round also pays one-off allocations
        vox = Voxelizer(cs, None)
        vox.init(0.5)
        c0 = time.perf_counter()
        vox.addPoints(a, None)
        vox.addPoints(b, None)
        cs.synchronize()
        dt = time.perf_counter() - c0
        t_vox = dt if t_vox is None else min(t_vox, dt)
        voxels = vox.stored_voxels
        vox.cleanup()
    p2p = []
    for _ in range(5):
        c0 = time.perf_counter()
        t_opt, R_opt = icp.full_pt2pt(a, b, zero, iterations=20)
        p2p.append(time.perf_counter() - c0)
    out["pt2pt20"] = {
        "what": "configs[0]: real pair 0 -> 1, Voxelizer.addPoints of both sweeps at 0.5 m + prepareNNS/dispatchNNS + "
                "full_pt2pt(iterations=20), host arrays in, pose out",
        "voxelize_ms": t_vox * 1e3, "voxels": int(voxels), "nn_ms": (fw + hw) * 1e3 / 2,
        "full_pt2pt_ms": float(np.median(p2p) * 1e3), "pairs_per_s": 1.0 / (t_vox + fw + float(np.median(p2p))),
        "translation": [float(v) for v in t_opt],
    }
    cs.cleanup()

    # --- configs[2]: 1 M synthetic points, voxel sizes 0.1 ... 2 m: voxelize + Morton sort + NN scan
    sweeps = [lidar.synthetic_scan(s, rings=64, steps=1953) + np.array([0.5 * s, 0.0, 0.0]) for s in range(10)]
    cloud = np.concatenate(sweeps)
    rng = np.random.default_rng(1)
    q = cloud + rng.normal(0, 0.05, cloud.shape)
    cs = ComputeShader(maxPointsPerCloud=200000, device=device)
    from pointcloudprocessing_amd import _lib as L
    from pointcloudprocessing_amd.frame import DeviceFrame
    dev_sweeps = []
    for c in sweeps:                       # the same sweeps resident in device memory (float64 rows of 3), for the device-fed leg
        f = DeviceFrame(max_points=200000, device=device)
        f.upload(np.ascontiguousarray(c))
        dev_sweeps.append(f)

    def fed_from_device(size):
        """Every sweep enqueued from device memory (pcp_voxel_add_dev: float32 cast, dispatch, stager, creation of the new
        voxels), four frames in flight, the device's own voxel count carried from frame to frame."""
        v = Voxelizer(cs, None)
        v.init(size)
        rec, tickets, begin = L.VoxelFrame(), [], 0
        c0 = time.perf_counter()
        for k, f in enumerate(dev_sweeps):
            ptr, n = f.points_dev()
            t = C.c_int(-1)
            check(cs._lib.pcp_voxel_add_dev(cs._ctx, C.c_void_p(ptr), 8, n, float(size), 0 if not tickets else -1, begin, 1024, 0, 1024,
                                            None, C.byref(t)))
            tickets.append(t.value)
            begin += n
            if k >= 4:
                check(cs._lib.pcp_voxel_add_wait(cs._ctx, tickets[k - 4], C.byref(rec)))
        for t in tickets[-4:]:
            check(cs._lib.pcp_voxel_add_wait(cs._ctx, t, C.byref(rec)))
        dt = time.perf_counter() - c0
        v.cleanup()
        return dt, rec.stored

    grid, rows = None, []
    for size in (0.1, 0.2, 0.5, 1.0, 2.0):
        best = None
        for _ in range(2):
            v = Voxelizer(cs, None)
            v.init(size)
            c0 = time.perf_counter()
            for c in sweeps:
                v.addPoints(c, None)
            cs.synchronize()
            dt = time.perf_counter() - c0
            stored = v.stored_voxels
            v.cleanup()
            best = dt if best is None else min(best, dt)
        fed = min(fed_from_device(size) for _ in range(3))
        if fed[1] != stored and not os.environ.get("PCP_BENCH_NOCHECK"):
            raise SystemExit("INVALID: device-fed and host-fed voxel maps differ (%d / %d voxels)" % (fed[1], stored))
        if grid is None:
            grid = CellGrid(cloud, cell=size, device=device)
        else:
            grid.rebuild(cloud, cell=size)
        # the same search with the cell edge refined until an occupied cell holds at most 16 points (same answers)
        grid.rebuild(cloud, cell=size, refine=16)
        fine_cell = grid.cell
        grid.query(q)
        fine_ms = min(grid.query(return_kernel_ms=True)[2] for _ in range(3))
        grid.rebuild(cloud, cell=size)
        info = grid.info()
        grid.query(q)
        qms = min(grid.query(return_kernel_ms=True)[2] for _ in range(3))
        def roof(bytes_, seconds, kernel, why):
            gbs = bytes_ / seconds / 1e9 if seconds else 0.0
            return {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "traffic": None, "limited_by": why}

        sort_s, build_s = info["sort_ms"] * 1e-3, info["build_ms"] * 1e-3
        rows.append({"voxel_m": size, "voxels": int(stored), "voxelize_ms": best * 1e3, "voxelize_mpts_per_s": len(cloud) / best / 1e6,
                     "voxelize_device_fed_ms": fed[0] * 1e3, "voxelize_device_fed_mpts_per_s": len(cloud) / fed[0] / 1e6,
                     "cells": int(info["cells"]), "morton_sort_ms": float(info["sort_ms"]), "sort_passes": int(info["sort_passes"]),
                     "sort_gb_per_s": 24.0 * len(cloud) * info["sort_passes"] / sort_s / 1e9 if sort_s else None,
                     "grid_build_ms": float(info["build_ms"]), "nn_ms": float(qms), "nn_queries_per_s": len(q) / (qms * 1e-3),
                     "nn_refined": {"cell_m": fine_cell, "nn_ms": float(fine_ms), "nn_queries_per_s": len(q) / (fine_ms * 1e-3),
                                    "what": "CellGrid(refine=16): the search cell halved until an occupied cell holds <= 16 points"},
                     "roofline": {
                         "voxelize": roof(28.0 * len(cloud) + 16.0 * stored, fed[0], "the frame feed (k_feed_voxelize, k_voxel_stage, radix "
                                          "sort of the unknown points, k_feed_new_voxels, k_feed_shift_old: 14 launches per sweep)",
                                          "launch latency: 10 sweeps of 100 k points, each a chain of 14 small dependent kernels "
                                          "(profiles/r03_voxel_kernel_stats.csv); 28 B per point + 16 B per voxel is what the feed must move"),
                         "morton_sort": roof(24.0 * len(cloud) * info["sort_passes"], sort_s, "k_rsd_hist / k_rsd_scan / k_rsd_scatter per "
                                             "8-bit digit", "1 M keys: a pass is 3 launches of a few microseconds each; the scatter's 128-B runs "
                                             "hold a pass at 2.3 TB/s at 50 M keys (profiles/r01_grid_50m.txt)"),
                         "cell_table": roof(12.0 * len(cloud) + 16.0 * info["cells"], max(build_s - sort_s, 1e-9), "key / head / table kernels "
                                            "of the grid build", "launch latency at 1 M points (a dozen small kernels around the sort)"),
                         "grid_nn": roof(40.0 * len(q), qms * 1e-3, "k_grid_nn (exact neighbour over the 27 cells around a query)",
                                         "gathers from the cells' point runs (L2-served): %d points per occupied cell at this size"
                                         % max(1, len(cloud) // max(1, int(info["cells"])))),
                     }})
    grid.cleanup()
    for f in dev_sweeps:
        f.cleanup()
    cs.cleanup()
    out["ringless_pt2pl"] = ringless_leg(device)
    e2e = odometry_leg(raws, device)
    if e2e is not None:
        out["odometry_e2e"] = e2e
    out["voxel_sweep"] = {"what": "configs[2]: %d synthetic LiDAR-like points (10 sweeps): Voxelizer.addPoints per sweep, Morton radix "
                                  "sort + cell table of the whole cloud, exact NN of the cloud jittered by 5 cm" % len(cloud),
                          "rows": rows}
    return out


def odometry_leg(raws, device):
    """The reference's real unit of work, a driver frame (EnvironmentConstructor.calculateTransition_imu :154-265: place by
    the IMU, refine against the block's reference cloud, pose, apply, voxelize, replace the reference every 10th frame),
    over the thirty real sweeps with their OXTS records: as one stream of device work (pcp_chain_*) and, beside it, frame
    by frame through EnvironmentConstructor on the host with the same kernels underneath."""
    imu_file = os.path.join(GOLD, "ref_imu.npz")
    if not os.path.exists(imu_file) or len(raws) < 30:
        return None
    from pointcloudprocessing_amd import EnvironmentConstructor, Voxelizer
    from pointcloudprocessing_amd.chain import OdometryChain, imu_rows
    from pointcloudprocessing_amd.oxts import OxtsDataReader, parse_timestamp
    g = np.load(imu_file)
    rd = OxtsDataReader()
    for k, (vals, stamp) in enumerate(zip(g["oxts_values"], g["oxts_stamps"])):
        rd.add_record(f"{k:010d}", vals, parse_timestamp(str(stamp)))
    rd.finish()
    records = [rd.getOx(k) for k in rd.getfilenames()[:len(raws)]]
    rows, rots, steps, _ = imu_rows(records)
    F = len(raws)
    pinned = PinnedFrames(raws)
    cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100, device=device)
    chain = OdometryChain(max_frames=F, max_points=1 << 17, device=device)
    walls, devs, track, voxels = [], [], None, 0
    for rep in range(6):
        chain.reset()
        vox = Voxelizer(cs, None)
        vox.separate_colors, vox.filter_outliers = False, False
        vox.init(0.5)
        cs.synchronize()
        c0 = time.perf_counter()
        # runs of ten frames, two in flight: the upload and decode of a run overlap the chain of the run before it
        track, pending, dev_ms = [], [], 0.0
        for lo in range(0, F, 10):
            hi = min(F, lo + 10)
            pending.append(chain.submit([pinned[k] for k in range(lo, hi)], rows[lo:hi], point_to_plane=True, iterations=MAX_ITER,
                                        tol=TOL, voxelizer=vox))
            if len(pending) == 2:
                track += chain.collect(pending.pop(0))
                dev_ms += chain.last_chain_ms
        for t in pending:
            track += chain.collect(t)
            dev_ms += chain.last_chain_ms
        walls.append(time.perf_counter() - c0)
        devs.append(dev_ms)
        voxels = vox.stored_voxels
        vox.cleanup()
    # the same drive with the map output attached: 2 m voxels, every voxel a frame stages goes into a cell grid on the device
    from pointcloudprocessing_amd import CellGrid, MapSink
    sink_walls, sink_points, sink_voxels = [], 0, 0
    for rep in range(4):
        chain.reset()
        grid = CellGrid(np.zeros((0, 3)), cell=1.0, device=device)
        vox = Voxelizer(cs, None)
        vox.separate_colors, vox.filter_outliers = False, False
        vox.init(2.0)
        vox.voxman = MapSink(grid, compute=cs)      # switches the device point store on: after the reset, before the first frame
        sink = vox.voxman
        cs.synchronize()
        c0 = time.perf_counter()
        pending = []
        for lo in range(0, F, 10):
            hi = min(F, lo + 10)
            pending.append(chain.submit([pinned[k] for k in range(lo, hi)], rows[lo:hi], point_to_plane=True, iterations=MAX_ITER,
                                        tol=TOL, voxelizer=vox))
            if len(pending) == 2:
                chain.collect(pending.pop(0))
        for t in pending:
            chain.collect(t)
        grid.synchronize()
        sink_walls.append(time.perf_counter() - c0)
        sink_points, sink_voxels = sink.points_fed, sink.voxels_fed
        vox.cleanup()
        grid.cleanup()
    chain.close()
    pinned.close()
    # the same drive frame by frame on the host
    reader = lidar.LidarDataReader(device=device)
    for k, raw in enumerate(raws):
        reader.add_frame(f"{k:010d}", raw)
    icp = PointcloudIcpContainer(cs)
    env = EnvironmentConstructor(None, rd, reader, icp, cs, voxelizer=Voxelizer(cs, None))
    host = []
    for rep in range(3):
        env.init(0.5)
        c0 = time.perf_counter()
        for _ in range(F):
            frame, rec = env.getNextFrameData(0)
            env.calculateTransition_imu(frame, rec, point_to_plane=True, iterations=MAX_ITER)
        cs.synchronize()
        host.append(time.perf_counter() - c0)
    same = all(np.array_equal(track[k][1], env.prev_poses[k][1]) for k in range(F))
    path = float(np.linalg.norm(track[-1][1]))
    env.cleanup()
    reader.cleanup()
    cs.cleanup()
    if not same and not os.environ.get("PCP_BENCH_NOCHECK"):
        raise SystemExit("INVALID: the device chain and the frame-by-frame driver disagree on the refined positions")
    wall = float(np.median(walls[1:]))
    return {"what": "the %d real sweeps as one drive: IMU placement, point-to-plane refinement against the block's reference cloud "
                    "(<= 10 iterations, 15 full ones when a block closes), pose, refined cloud, Voxelizer feed at 0.5 m, reference "
                    "replaced every 10th frame; raw bytes in page-locked host memory in, every frame's pose and position out, map "
                    "on the device" % F,
            "frames_per_s": F / wall, "ms_per_frame": wall / F * 1e3, "device_ms_per_frame": float(np.median(devs[1:])) / F,
            "frames_per_s_best": F / min(walls[1:]), "voxels": int(voxels), "path_m": path,
            "iterations_per_frame": float(np.mean([len(t[4]) for t in track[1:]])),
            "with_map_output": {"what": "the same drive with 2 m voxels and every voxel a frame stages handed to a cell grid on the "
                                        "device (MapSink through the staged-points log)",
                                "frames_per_s": F / float(np.median(sink_walls[1:])), "points_in_map": int(sink_points),
                                "voxels_staged": int(sink_voxels)},
            "host_driven": {"what": "EnvironmentConstructor.calculateTransition_imu per frame (decoded clouds on the host, NumPy "
                                    "placement and pose algebra, device loop, Voxelizer.addPoints from host arrays)",
                            "frames_per_s": F / float(np.median(host[1:])), "ms_per_frame": float(np.median(host[1:])) / F * 1e3},
            "same_positions_as_host_driven": bool(same)}


def ringless_leg(device):
    """configs[1]: a model cloud without scan rings against itself under a rigid transform + Gaussian noise, point-to-plane
    ICP with normals estimated from the model.  golden-tiger.ply is not part of the reference checkout: PCP_BENCH_PLY names
    a file to use instead of the synthetic stand-in (a bumpy closed surface of the same order of size)."""
    from pointcloudprocessing_amd.map_output import read_ply_ascii, read_ply_binary
    path = os.environ.get("PCP_BENCH_PLY")
    if path and os.path.exists(path):
        cloud = read_ply_ascii(path)
        cloud = (read_ply_binary(path) if cloud is None else cloud)[:, :3].astype(np.float64)
        name = os.path.basename(path)
    else:
        rng = np.random.default_rng(7)
        u, v = rng.uniform(0, 2 * np.pi, 400000), np.arccos(rng.uniform(-1, 1, 400000))
        r = 1.0 + 0.15 * np.sin(3 * u) * np.sin(4 * v) + 0.08 * np.cos(5 * v + u)
        cloud = np.column_stack([1.3 * r * np.sin(v) * np.cos(u), r * np.cos(v), 0.8 * r * np.sin(v) * np.sin(u)])
        name = "synthetic closed surface, 400 k points (stand-in for golden-tiger.ply)"
    extent = float(np.linalg.norm(cloud.max(axis=0) - cloud.min(axis=0)))
    R = lidar.rotation_xyz(0.02, -0.01, 0.015)
    t = np.array([0.05, -0.02, 0.03]) * extent / 3.0
    rng = np.random.default_rng(0)
    moved = (R @ cloud.T).T + t + rng.normal(0.0, 0.002 * extent, cloud.shape)
    icp = PointcloudIcpContainer(None)
    radius = 0.03 * extent
    times = []
    for _ in range(3):
        c0 = time.perf_counter()
        _, t_opt, R_opt = icp.align_ringless_pt2pl(cloud, moved, radius, iterations=20, viewpoint=cloud.mean(axis=0) + 10 * extent)
        times.append(time.perf_counter() - c0)
    # p <- R_opt p - t_opt undoes p' = R p + t: R_opt = R^T, t_opt = R^T t
    return {"what": "configs[1]: ring-less model (%s, %d points) against R * model + t + N(0, (0.2 %% of its extent)^2): exact "
                    "neighbours on the cell grid, normals from the model (PCA of the ball), point-to-plane <= 20 iterations; "
                    "host arrays in, pose out" % (name, len(cloud)),
            "ms": float(min(times) * 1e3), "rotation_error": float(np.abs(R_opt - R.T).max()),
            "translation_error_over_extent": float(np.abs(t_opt - R.T @ t).max() / extent), "parity": "unpinned (no counterpart in the reference)"}


def scan_to_map_leg(raws, rank, world, device, dist, exchange):
    """configs[4] at N > 1: a scan registered against an accumulated map that is split over the ranks by key range
    (100 sweeps = 12.3 M points per rank), the 8x8 tile summed by an RCCL all-reduce per iteration through the
    library's own communicator (pcp_comm_*, pcp_grid_icp_run_comm).  With exchange "gloo" (rehearsal: ranks sharing
    a card) the tile goes through torch.distributed instead."""
    import torch
    from pointcloudprocessing_amd.scan_to_map import MODE_POINT, MapRegistration, RcclComm
    from pointcloudprocessing_amd.cell_grid import partition_keys
    a, b = lidar.frame_from_raw(raws[0][:, :3]), lidar.frame_from_raw(raws[1][:, :3])
    sweeps = 100 * world
    step = np.array([0.02, 0.0, -0.33])
    k = sweeps // 2
    scan = b + (k + 1) * step + np.array([0.04, 0.0, 0.05])
    comm = RcclComm(rank, world, device) if exchange == "rccl" else None
    c0 = time.perf_counter()
    # The key ranges are cut once, by rank 0, from a sample of the map (every 8th sweep: balance is all the cuts decide)
    # and handed to the others -- world + 1 keys instead of every rank sorting the whole map; each rank then takes its
    # share of the map chunk by chunk (MapRegistration.ingest keeps the points in or next to its range), so no rank
    # ever holds more than one chunk of the whole map in host memory.
    box = [None]
    if rank == 0:
        sample = np.concatenate([a + j * step for j in range(0, sweeps, 8)])
        box[0] = partition_keys(sample, 0.5, world, device)
    dist.broadcast_object_list(box, src=0)
    chunk = 25
    reg = MapRegistration(np.concatenate([a + j * step for j in range(chunk)]), 0.5, nn_cell=0.15, rank=rank, world=world,
                          device=device, comm=comm, ranges=box[0])
    for c in range(chunk, sweeps, chunk):
        reg.ingest(np.concatenate([a + j * step for j in range(c, min(sweeps, c + chunk))]))
    build_s = time.perf_counter() - c0
    map_points = sweeps * len(a)
    times, res = [], None
    for _ in range(6):
        torch.cuda.synchronize()
        dist.barrier()
        c0 = time.perf_counter()
        res = reg.register(scan, MODE_POINT, 0.4, max_iter=10, check_every=None if comm is not None else 0)
        times.append(time.perf_counter() - c0)
    steps = [None] * world
    dist.all_gather_object(steps, res[2].tobytes())
    same = all(x == steps[0] for x in steps)
    slabs = [None] * world
    dist.all_gather_object(slabs, int(reg.slab_points))
    reg.close()
    if comm is not None:
        comm.close()
    return {"what": "configs[4]: %d-point map (%d sweeps) split over %d ranks by partition-cell key range, %d-point scan, "
                    "point-to-point, 10 iterations, exact neighbours within 0.4 m; per iteration one all-reduce of the 8x8 tile"
                    % (map_points, sweeps, world, len(scan)),
            "exchange": "RCCL inside the library (pcp_grid_icp_run_comm)" if comm is not None else "torch.distributed over gloo (rehearsal)",
            "ms_per_registration": float(np.median(times[1:]) * 1e3), "build_and_partition_s": build_s,
            "map_points": int(map_points), "slab_points_per_rank": slabs, "iterations": int(len(res[2])), "all_ranks_hold_the_same_steps": bool(same),
            "translation": [float(v) for v in res[0]]}


def scan_to_map_child():
    """`bench.py --leg scan-to-map`: one rank of the leg above in a process of its own (control traffic over gloo on
    the port the parent chose).  The collective has never run on more than one GPU where this was written, so the
    parent keeps it at arm's length: a crash or a hang here costs the bench line one object, not the line."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    raws, _ = real_frames()
    out = scan_to_map_leg(raws, rank, world, device, dist, os.environ.get("PCP_LEG_EXCHANGE", "rccl"))
    dist.barrier()
    if rank == 0:
        print("PCP_LEG " + json.dumps(out), flush=True)
    dist.destroy_process_group()


def run_leg_in_children(rank, dist, exchange, seconds):
    """Every rank starts its child; rank 0 picks the rendezvous port and returns what its child printed."""
    import socket
    import subprocess
    box = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            box[0] = sk.getsockname()[1]
    dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(box[0]), PCP_LEG_EXCHANGE=exchange)
    for name in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "GROUP_RANK", "ROLE_RANK"):
        env.pop(name, None)       # the child makes its own store on the new port
    try:
        done = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "scan-to-map"], env=env, capture_output=True,
                              text=True, timeout=seconds)
    except subprocess.TimeoutExpired:
        return {"error": "did not finish within %d s (child ended)" % seconds}
    for line in done.stdout.splitlines()[::-1]:
        if line.startswith("PCP_LEG "):
            return json.loads(line[8:])
    if rank != 0 and done.returncode == 0:
        return None
    tail = (done.stderr or "").strip().splitlines()[-3:]
    return {"error": "child exit code %d: %s" % (done.returncode, " | ".join(tail)[-400:])}


def cpu_baseline_child(workload, pair):
    """`bench.py --leg cpu-baseline --pair k`: one frame pair of the headline workload through the NumPy restatement (the
    checker under oracle/), single-threaded, in a process of its own; prints its seconds."""
    from oracle import np_oracle as O
    raws, _ = real_frames()
    i = pair % (len(raws) - 1)
    c0 = time.perf_counter()
    fa, fb = lidar.frame_from_raw(raws[i][:, :3]), lidar.frame_from_raw(raws[i + 1][:, :3])
    if workload == "moved-copy":
        fb = moved_copy(fa, i + 1)[0]
    O.full_pt2pl(fa, fb, np.zeros(3), iterations=MAX_ITER)
    print("PCP_CPU %.6f" % (time.perf_counter() - c0), flush=True)


def cpu_baseline_all_cores(workload, n_frames):
    """The same algorithm on the host's cores: one frame pair per worker process (decode, scan lines, normals, loop, pose in
    NumPy, one thread each), as many workers as this box's share of cores (16 at most), all at once; value = pairs over the
    wall time of the whole batch."""
    import subprocess
    workers = max(1, min(16, os.cpu_count() or 1, n_frames - 1))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    c0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--leg", "cpu-baseline", "--pair", str(k), "--workload",
                               workload], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(workers)]
    own = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            continue
        for line in out.splitlines():
            if line.startswith("PCP_CPU "):
                own.append(float(line.split()[1]))
    wall = time.perf_counter() - c0
    if not own:
        return None
    return {"value": len(own) / wall, "unit": "frame-pairs/s", "cores": workers, "kind": "port",
            "one_core_value": 1.0 / float(np.mean(own)),
            "sample": "%d consecutive frame pairs of the same workload (decode, scan lines, normals, loop), NumPy restatement of "
                      "the same algorithm, one single-threaded process per pair, all running at once: %.1f s of wall time, %.1f s "
                      "per pair on its core; host reports %d logical cpus" % (len(own), wall, float(np.mean(own)), os.cpu_count() or 0)}


def headline_roofline(res, achieved, args):
    """The roofline object of the headline: achieved / peak / frac are the HBM figures (algorithmic bytes per launch over
    the launch time measured by HIP events in this run); what the kernel is actually bound by is named from the committed
    counter summaries of this workload (profiles/r*_issue_pipes.json, r*_traffic.json), not from a literal."""
    pipes = issue_pipes(args.workload, args.pairs)
    traffic = res["traffic"]
    bound, why = "hbm", "no counter summary of this workload is committed: only the HBM fraction is known"
    if pipes:
        busiest = max((("texture-address units (gather issue)", pipes["ta_busy_avg"]), ("vector ALU", pipes["valu_busy"]),
                       ("hbm", achieved / HBM_PEAK_GBS)), key=lambda kv: kv[1])
        bound = busiest[0]
        why = ("not HBM (%.1f %% of its peak): chains of dependent 16-byte gathers served by L1/L2. Counters of this workload (%s): "
               "texture-address units %.0f %% busy on average and %.0f %% on the busiest, vector ALU %.0f %% busy, waves waiting %.0f %% "
               "of their life, L2 hit rate %.0f %%"
               % (100.0 * achieved / HBM_PEAK_GBS, pipes["source"].split(" (")[0], 100 * pipes["ta_busy_avg"], 100 * pipes["ta_busy_max"],
                  100 * pipes["valu_busy"], 100 * pipes["waves_waiting"], 100 * pipes["l2_hit_rate"]))
        if traffic and res["alg_bytes"]:
            why += ("; `traffic` (L2 misses: every launch streams its float64 working copy, hints and each pair's target once) is "
                    "%.1fx the algorithmic bytes, which count a target once per pair, not once per launch" % (traffic / res["alg_bytes"]))
    return {"bound": bound, "kernel": res["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "hbm_frac": achieved / HBM_PEAK_GBS, "binding_pipe_busy": (pipes or {}).get("ta_busy_avg"),
            "avg_launch_us": res["avg_launch_s"] * 1e6, "launches_timed": res["ls_launches"],
            "algorithmic_bytes_per_launch": res["alg_bytes"], "first_launch": res.get("first_launch"),
            "issue_pipes": pipes, "limited_by": why}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=256,
                    help="consecutive frame pairs per GPU per step (256 = one GPU's share of a 2,048-pair sequence on 8 GPUs; the "
                         "end of every launch and the one-block-per-pair solve between launches leave the GPU part idle, which "
                         "larger launches amortise: 32 / 64 / 128 / 256 pairs measured 6,430 / 6,990 / 7,250 / 7,420 pairs/s)")
    ap.add_argument("--workload", choices=("sequence", "sequence-imu", "moved-copy"), default="sequence")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--leg", choices=("scan-to-map", "cpu-baseline"), default=None, help=argparse.SUPPRESS)
    ap.add_argument("--pair", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-latency", action="store_true",
                    help="skip the single-pair latency legs (profiling passes: every launch of the fused kernel then is a full step's)")
    args = ap.parse_args()
    if args.leg == "scan-to-map":
        return scan_to_map_child()
    if args.leg == "cpu-baseline":
        return cpu_baseline_child(args.workload, args.pair)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU; PCP_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with
    # fewer GPUs than ranks (ranks then share devices round robin)
    backend = os.environ.get("PCP_BENCH_BACKEND", "nccl")
    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    res = (run_moved_copy if args.workload == "moved-copy" else run_sequence)(args, rank, world, device, barrier)
    elapsed = res["elapsed"]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    raws, pinned = res.pop("frames")
    total_pairs = args.pairs * world * args.steps

    single, cpu, more = None, None, {}
    if rank == 0 and not args.no_latency:
        # one registration alone on the GPU, target kept prepared: upload of the 3 MB float64 source, loop, fetch
        a, b = lidar.frame_from_raw(raws[0][:, :3]), lidar.frame_from_raw(raws[1][:, :3])
        cs = ComputeShader(maxPointsPerCloud=max(len(a), len(b)), maxScans=100, device=device)
        PointcloudIcpContainer(cs).preparePointToPlane(a, np.zeros(3), b)
        times = []
        for _ in range(13):
            c0 = time.perf_counter()
            dx0, _, _, _ = cs.icpRun(b, MODE_PLANE, MAX_ITER, tol=TOL)
            times.append(time.perf_counter() - c0)
        cs.cleanup()
        single = {"ms": float(np.median(times[3:]) * 1e3), "iterations": int(len(dx0)),
                  "what": "real pair 0 -> 1 alone on the GPU, target prepared: upload of the 3 MB float64 source, the loop, fetch",
                  "from_raw_bytes_ms": res["raw_latency_ms"],
                  "from_raw_bytes_what": "the same pair from its two raw records in host memory: upload, decode, rings, normals, loop, fetch"}
    if rank == 0 and world == 1 and not args.no_extras:
        more = extras(raws, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_all_cores(args.workload, len(raws))
    if pinned is not None:
        pinned.close()

    if rank == 0:
        achieved = res["alg_bytes"] / res["avg_launch_s"] / 1e9 if res["avg_launch_s"] else 0.0
        out = {
            "metric": "ICP frame-pairs/s (120k-pt KITTI scans, NN queries/s alongside)",
            "value": total_pairs / elapsed,
            "unit": "frame-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": res["data"],
            "nn_queries_per_s": res["queries"] / elapsed,
            "config": {"workload": res["workload"], "pairs_per_gpu_per_step": args.pairs, "n_a": res["n_a"], "n_b": res["n_b"],
                       "iterations_per_pair": res["iters"], "parallelism": "independent pairs per GPU, batched launches"},
            "roofline": headline_roofline(res, achieved, args),
            "cpu_baseline": cpu,
            "single_pair": single,
        }
        if res.get("imu_guess"):
            out["imu_guess"] = res["imu_guess"]
        out.update(more)
    else:
        out = None
    if dist is not None and not args.no_extras:
        torch.cuda.synchronize()
        leg = run_leg_in_children(rank, dist, os.environ.get("PCP_BENCH_LEG_EXCHANGE", "rccl" if backend == "nccl" else "gloo"),
                                  int(os.environ.get("PCP_BENCH_LEG_SECONDS", "150")))
        if out is not None:
            out["scan_to_map"] = leg
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
