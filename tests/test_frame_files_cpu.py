"""Host logic of the run sources (no device): which frames a run holds, and the order files are taken in."""
import os

from pointcloudprocessing_amd.frame_files import FrameFileRuns, MemoryRuns, velodyne_files


def test_memory_runs_cover_every_frame_and_share_one_frame_when_asked():
    frames = list(range(14))
    plain = [(lo, arr) for lo, arr, _ in MemoryRuns(frames, 5)]
    assert plain == [(0, [0, 1, 2, 3, 4]), (5, [5, 6, 7, 8, 9]), (10, [10, 11, 12, 13])]
    shared = [(lo, arr) for lo, arr, _ in MemoryRuns(frames, 5, 4)]            # pair pipeline: last frame = next run's first
    assert shared == [(0, [0, 1, 2, 3, 4]), (4, [4, 5, 6, 7, 8]), (8, [8, 9, 10, 11, 12]), (12, [12, 13])]
    pairs = sum(len(arr) - 1 for _, arr in shared)
    assert pairs == len(frames) - 1                                            # every consecutive pair exactly once
    assert [(lo, arr) for lo, arr, _ in MemoryRuns([7], 5, 4)] == [(0, [7])]   # a lone frame: one run, no pair
    assert list(MemoryRuns([], 5, 4)) == []
    # a run that would only repeat the shared frame is not made
    assert [lo for lo, _, _ in MemoryRuns(list(range(9)), 5, 4)] == [0, 4]


def test_file_runs_start_where_memory_runs_start():
    class Starts(FrameFileRuns):               # starts() alone: no buffers, no thread
        def __init__(self, n, run, stride):
            self.paths, self.run, self.stride = [None] * n, run, stride
            self._bufs = []
    for n in (0, 1, 2, 5, 9, 13, 14, 30):
        for run, stride in ((5, 5), (5, 4), (2, 1), (65, 64)):
            assert Starts(n, run, stride).starts() == [lo for lo, _, _ in MemoryRuns(list(range(n)), run, stride)], (n, run, stride)


def test_velodyne_files_are_taken_in_name_order(tmp_path):
    for k in (3, 0, 11, 2):
        (tmp_path / f"{k:010d}.bin").write_bytes(b"")
    (tmp_path / "notes.txt").write_text("x")
    assert [os.path.basename(p) for p in velodyne_files(str(tmp_path))] == [f"{k:010d}.bin" for k in (0, 2, 3, 11)]
