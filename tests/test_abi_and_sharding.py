"""CPU suite: the C-ABI library loads and exports exactly what include/pcp.h
declares (no compute without a GPU), the product refuses to run without its
native library, and the pair sharding used for multi-GPU runs is correct under
a 2-process gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import pointcloudprocessing_amd as pcp
from pointcloudprocessing_amd import _lib, sharding
from tests._data import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcp_[a-z0-9_]+)\s*\(", text)))


def test_header_library_and_binding_agree():
    names = declared_symbols()
    assert len(names) >= 25
    lib = pcp.load()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in pcp.h but not exported by libpcp.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding and pcp.h list different entry points"
    assert lib.pcp_version() >= 1


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pcp.PcpError) as e:
        pcp.ComputeShader(maxPointsPerCloud=1000, maxScans=100)
    assert "device" in str(e.value).lower() or "hip" in str(e.value).lower()
    ctx = C.c_void_p()
    assert pcp.load().pcp_create(0, 0, 100, C.byref(ctx)) != 0            # bad size is rejected before any HIP call
    assert b"max_points" in pcp.load().pcp_last_error()


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpcp.so"))
    with pytest.raises(pcp.PcpError) as e:
        _lib.load()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_the_checker():
    pkg = os.path.join(ROOT, "pointcloudprocessing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "oracle/" not in text, f


def test_packed_frames_round_trip_bit_for_bit():
    """lidar.pack_raw_frames / unpack_raw_frames (the container of the KITTI sequence fixture): millimetre-quantised
    float32 coordinates, negative zeros included, come back with their bit patterns."""
    from pointcloudprocessing_amd import lidar
    rng = np.random.default_rng(0)
    frames = []
    for n in (1, 7, 2000):
        mm = rng.integers(-80000, 80000, size=(n, 3))
        xyz = (mm.astype(np.float64) / 1000.0).astype(np.float32)
        xyz[rng.random((n, 3)) < 0.05] = -0.0
        frames.append(xyz)
    frames.append(np.zeros((0, 3), np.float32))
    packed = lidar.pack_raw_frames(frames)
    back = lidar.unpack_raw_frames(packed)
    assert len(back) == len(frames)
    for a, b in zip(frames, back):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_round_robin_sharding_and_odometry_chain():
    assert sharding.pairs_of_rank(29, 0, 8) == [0, 8, 16, 24]
    assert sharding.pairs_of_rank(29, 7, 8) == [7, 15, 23]
    assert sorted(sum((sharding.pairs_of_rank(29, r, 8) for r in range(8)), [])) == list(range(29))
    assert sharding.pairs_of_rank(3, 5, 8) == []
    # two steps of 1 m along -x each, no rotation: p_k = p_{k+1} - t  with t = (1, 0, 0)
    poses = sharding.compose_odometry([(np.array([1.0, 0, 0]), np.eye(3))] * 2)
    assert np.allclose(poses[2][1], [-2, 0, 0]) and np.allclose(poses[2][0], np.eye(3))
    t, R = sharding.unpack_pose(sharding.pack_pose(np.arange(3), np.arange(9).reshape(3, 3)))
    assert np.array_equal(t, np.arange(3)) and np.array_equal(R, np.arange(9).reshape(3, 3))


WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pointcloudprocessing_amd import sharding, lidar
from oracle import np_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
n_pairs = 5
a = lidar.synthetic_scan(0, rings=9, steps=150)
local = {}
for k in sharding.pairs_of_rank(n_pairs, rank, world):
    R = lidar.rotation_xyz(0.0, 0.002 * (k + 1), 0.0)
    b = (R @ a.T).T + np.array([0.01 * (k + 1), 0.0, 0.0])
    local[k] = O.full_pt2pt(a, b, np.zeros(3), iterations=4)
poses = sharding.gather_poses(local, n_pairs, rank, world)
assert len(poses) == n_pairs and all(p is not None for p in poses)
# the same gather when the pairs are dealt in contiguous blocks (what the raw-frame driver does)
per = -(-n_pairs // world)
mine = {k: (np.full(3, float(k)), np.eye(3) * (k + 1)) for k in range(rank * per, min(n_pairs, (rank + 1) * per))}
blocks = sharding.gather_poses(mine, n_pairs, rank, world)
assert [float(t[0]) for t, _ in blocks] == [0.0, 1.0, 2.0, 3.0, 4.0] and all(R[0, 0] == k + 1 for k, (_, R) in enumerate(blocks))
for k in local:                      # own results come back unchanged, in pair order
    assert np.array_equal(poses[k][0], local[k][0]) and np.array_equal(poses[k][1], local[k][1])
np.save(os.path.join(sys.argv[2], f"poses_{rank}.npy"), np.stack([sharding.pack_pose(*p) for p in poses]))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], env=dict(env, RANK=str(r)))
             for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    p0, p1 = np.load(tmp_path / "poses_0.npy"), np.load(tmp_path / "poses_1.npy")
    assert p0.shape == (5, 12) and np.array_equal(p0, p1)          # every rank holds every pose
    assert len({tuple(r) for r in p0}) == 5                           # and they are five different registrations
