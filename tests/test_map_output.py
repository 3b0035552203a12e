"""Point arena and PLY output against the reference's GpuMemoryManager (scripted run in
tests/golden/ref_arena.npz, scenarios of tests/GpuMemoryManagerTest.py) and the file format of
PointcloudHandler.savePointcloudPly."""
import numpy as np
import pytest

from pointcloudprocessing_amd.map_output import GpuMemoryManager, read_ply_ascii, save_arena_ply, save_ply
from tests._data import golden


def test_arena_follows_the_reference_through_a_scripted_sequence():
    g = golden("ref_arena.npz")
    m = GpuMemoryManager(maxNumberOfPoints=64)
    handles = {}
    for (op, k, state) in zip(g["script_ops"], g["script_args"], g["states"]):
        moved = -1
        if op == "add":
            handles[int(k)] = m.addPoints(g[f"cloud{int(k)}"])
        elif op == "free":
            moved = int(m.freeSpace(handles[int(k)]))
        else:
            m.markLastBufferPoint()
        flat = [v for a in m.allocationIndexes for v in a]
        mine = [moved, m.getMaxPointIndex(), m.last_buffered_point, len(m.pointers)] + flat + [-1] * (24 - len(flat))
        assert mine == state.tolist()
    assert np.array_equal(m.points, g["points"])


def test_arena_scenarios_of_the_reference_tests():
    """tests/GpuMemoryManagerTest.py:14-160."""
    rng = np.random.default_rng(0)
    m = GpuMemoryManager(maxNumberOfPoints=20)
    p1, p2, p3 = rng.random((2, 6)), rng.random((3, 6)), rng.random((2, 6))
    a, b, c = m.addPoints(p1), m.addPoints(p2), m.addPoints(p3)
    assert a == 124 and m.pointers[a] == (0, 2) and m.pointers[b] == (2, 5) and m.pointers[c] == (5, 7)
    assert np.allclose(m.points[2:5], p2, atol=1e-5)
    m.freeSpace(a)
    m.freeSpace(b)
    d = m.addPoints(rng.random((5, 6)))
    assert m.pointers[c] == (0, 2) and m.pointers[d] == (2, 7) and np.allclose(m.points[0:2], p3, atol=1e-5)
    with pytest.raises(ValueError):
        m.freeSpace(b)
    with pytest.raises(ValueError):
        m.freeSpace(5)
    with pytest.raises(MemoryError):
        m.addPoints(np.zeros((14, 6)))
    m2 = GpuMemoryManager(maxNumberOfPoints=20)
    m2.addPoints(np.zeros((5, 6)))
    m2.addPoints(np.zeros((1, 6)))
    m2.addPoints(np.zeros((3, 6)))
    assert m2.allocationIndexes == [(0, 5, 124), (5, 6, 125), (6, 9, 126)] and m2.fullDefragment() is False
    q = m2.addPoints(list(np.ones((2, 6))))
    assert m2.getMaxPointIndex() == 11
    m2.markLastBufferPoint()
    assert m2.last_buffered_point == 11
    assert m2.freeSpace(q) is False and m2.last_buffered_point == 0
    m2.dropPointers()
    assert m2.getMaxPointIndex() == 0 and m2.pointers == {}


def test_ply_text_format_and_round_trip(tmp_path):
    pts = np.array([[1.23456789, -2.5, 3.0, 0.0, 0.5, 1.0],
                    [-0.0000004, 100.0, 7.125, 1.2, -0.3, 0.999]], dtype=np.float32)
    path = save_ply(str(tmp_path / "cloud"), pts)
    text = open(path).read().split("\n")
    assert text[:10] == ["ply", "format ascii 1.0", "element vertex 2", "property float x", "property float y",
                         "property float z", "property uchar red", "property uchar green", "property uchar blue",
                         "end_header"]
    assert text[10] == "1.234568 -2.500000 3.000000 0 127 255"       # colour * 255, clipped, truncated
    assert text[11] == "-0.000000 100.000000 7.125000 255 0 254"
    back = read_ply_ascii(path)
    assert back.shape == (2, 6) and np.allclose(back[:, :3], pts[:, :3], atol=5e-7)
    assert np.allclose(back[:, 3:], np.array([[0, 127, 255], [255, 0, 254]]) / 255.0)
    m = GpuMemoryManager(maxNumberOfPoints=10)
    h = m.addPoints(np.zeros((3, 6)))
    m.addPoints(pts)
    m.freeSpace(h)
    assert open(save_arena_ply(str(tmp_path / "arena"), m)).read() == open(path).read()
    empty = save_ply(str(tmp_path / "empty"), np.zeros((0, 6)))
    assert read_ply_ascii(empty).shape == (0, 6)
    (tmp_path / "bin.ply").write_bytes(b"ply\nformat binary_little_endian 1.0\n\xff\xfe\x00\x80")
    assert read_ply_ascii(str(tmp_path / "bin.ply")) is None


@pytest.mark.parametrize("order,fmt", [("<", "binary_little_endian"), (">", "binary_big_endian")])
def test_binary_ply_is_read_like_the_reference_reads_it(tmp_path, order, fmt):
    """tryReadBinPly (PointcloudHandler.py:114-135) returns float32 (x, y, z, r/255, g/255, b/255), zeros
    for colours the file does not have; the plyfile package it uses is not installed, so the file is
    written here with numpy (parity of the reader unpinned beyond this statement of the format)."""
    from pointcloudprocessing_amd.map_output import load_pointcloud, read_ply_binary
    rng = np.random.default_rng(3)
    n = 1000
    v = np.zeros(n, dtype=[("x", order + "f4"), ("y", order + "f4"), ("z", order + "f4"), ("nx", order + "f8"),
                           ("red", "u1"), ("green", "u1"), ("blue", "u1"), ("alpha", "u1")])
    for k in ("x", "y", "z", "nx"):
        v[k] = rng.normal(size=n)
    for k in ("red", "green", "blue", "alpha"):
        v[k] = rng.integers(0, 256, n)
    header = ("ply\nformat %s 1.0\ncomment made by a test\nelement vertex %d\nproperty float x\nproperty float y\n"
              "property float z\nproperty double nx\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
              "property uchar alpha\nelement face 0\nproperty list uchar int vertex_indices\nend_header\n" % (fmt, n))
    path = str(tmp_path / "model.ply")
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        v.tofile(f)
    got = read_ply_binary(path)
    assert got.dtype == np.float32 and got.shape == (n, 6)
    assert np.array_equal(got[:, 0], v["x"].astype(np.float32)) and np.array_equal(got[:, 2], v["z"].astype(np.float32))
    assert np.array_equal(got[:, 3:], np.stack([v["red"], v["green"], v["blue"]], axis=-1).astype(np.float32) / 255.0)
    loaded = load_pointcloud(path)                      # the text reader gives up on the bytes, the binary one takes over
    assert loaded is not None and np.array_equal(loaded, got)
    plain = np.zeros(5, dtype=[("x", order + "f8"), ("y", order + "f8"), ("z", order + "f8")])
    plain["x"] = np.arange(5)
    p2 = str(tmp_path / "plain.ply")
    with open(p2, "wb") as f:
        f.write(("ply\nformat %s 1.0\nelement vertex 5\nproperty double x\nproperty double y\nproperty double z\nend_header\n"
                 % fmt).encode("ascii"))
        plain.tofile(f)
    got = read_ply_binary(p2)
    assert np.array_equal(got[:, 0], np.arange(5, dtype=np.float32)) and not np.any(got[:, 1:])
    import os
    for cut in (os.path.getsize(p2) - 30, 70):           # inside the vertex data / inside the header
        with open(p2, "r+b") as f:
            f.truncate(cut)
        with pytest.raises(ValueError):
            read_ply_binary(p2)
    text = save_ply(str(tmp_path / "text"), np.zeros((2, 6)))
    assert load_pointcloud(text).shape == (2, 6)
