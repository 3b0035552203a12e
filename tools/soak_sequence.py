"""The batched pipeline over and over on the same run of frames (default 60 runs of 256 pairs, two in flight): every run must
return the steps, convergence flags and poses of the first, bit for bit.    python tools/soak_sequence.py [runs [pairs]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import PinnedFrames, SequencePipeline  # noqa: E402
from bench import ping_pong, real_frames  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
raws, _ = real_frames()
pinned = PinnedFrames(raws)
frames = [pinned[k] for k in ping_pong(pairs + 1, len(raws), 0)]
pipe = SequencePipeline(pairs + 1, max_points=131072)


def signature(ticket):
    steps, n_pts, poses = pipe.collect_with_poses(ticket)
    return (b"".join(dx.tobytes() + bytes([conv]) for dx, conv in steps), n_pts.tobytes(),
            b"".join(np.asarray(t).tobytes() + np.asarray(R).tobytes() for t, R in poses))


first, bad, pending = None, 0, []
for k in range(runs):
    pending.append(pipe.submit(frames, point_to_plane=True, iterations=10))
    if len(pending) == 2:
        sig = signature(pending.pop(0))
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
            print("run", k - 1, "differs", flush=True)
    if k % 20 == 0:
        print("run", k, "submitted", flush=True)
for t in pending:
    if signature(t) != first:
        bad += 1
print("runs that differ from the first:", bad)
pipe.close()
pinned.close()
