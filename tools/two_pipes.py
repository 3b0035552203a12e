"""One SequencePipeline of 64 pairs per step against two of 32 pairs each on streams of their own (do the idle
stretches of one -- the solve kernels, the launch gaps -- fill with the other's work?):  python tools/two_pipes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import PinnedFrames, SequencePipeline  # noqa: E402
from bench import ping_pong, real_frames  # noqa: E402

raws, _ = real_frames()
pinned = PinnedFrames(raws)


def run(n_pipes, pairs_each, steps=20, warm=5):
    pipes = [SequencePipeline(pairs_each + 1, max_points=131072) for _ in range(n_pipes)]
    frames = [[pinned[k] for k in ping_pong(pairs_each + 1, len(raws), 5 * j)] for j in range(n_pipes)]

    def go(n):
        pending = [None] * n_pipes
        for _ in range(n):
            for j, p in enumerate(pipes):
                t = p.submit(frames[j], point_to_plane=True, iterations=10)
                if pending[j] is not None:
                    p.collect(pending[j])
                pending[j] = t
        for j, p in enumerate(pipes):
            p.collect(pending[j])

    go(warm)
    t0 = time.perf_counter()
    go(steps)
    dt = time.perf_counter() - t0
    for p in pipes:
        p.close()
    return n_pipes * pairs_each * steps / dt


for n, each in ((1, 64), (2, 32), (1, 64), (2, 32), (4, 16), (2, 64)):
    print("%d pipeline(s) x %d pairs per step: %.0f pairs/s" % (n, each, run(n, each)))
