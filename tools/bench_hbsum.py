"""HBM-bound leg: the stand-alone 64-ary tile tree (ParalellHBSum.glsl semantics) on tiles
resident in HBM.  Prints achieved GB/s against the bytes the kernel must read
(6 rows x 8 floats = 192 B of every 256-B tile; 168 B carry data)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import np_oracle as O  # noqa: E402  (checker, small case only)
from pointcloudprocessing_amd import ComputeShader  # noqa: E402
from pointcloudprocessing_amd._lib import check  # noqa: E402


def main():
    cs = ComputeShader(maxPointsPerCloud=1024)
    lib, ctx = cs._lib, cs._ctx
    check(lib.pcp_set_stream(ctx, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for n in (123645, 1 << 20, 1 << 22, 1 << 24):
        tiles = torch.randn(n, 64, device="cuda", dtype=torch.float32)
        g1 = -(-n // 64)
        scratch = torch.empty((g1 + -(-g1 // 64)) * 64, device="cuda", dtype=torch.float32)
        out = torch.zeros(64, device="cuda", dtype=torch.float32)
        args = (ctx, C.c_void_p(tiles.data_ptr()), n, C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()))
        for _ in range(3):
            check(lib.pcp_hb_sum_dev(*args))
        torch.cuda.synchronize()
        reps = 20
        check(lib.pcp_timer_start(ctx))
        for _ in range(reps):
            check(lib.pcp_hb_sum_dev(*args))
        ms = C.c_float(0)
        check(lib.pcp_timer_stop(ctx, C.byref(ms)))
        t = ms.value * 1e-3 / reps
        if n <= (1 << 20):
            ref = O.hb_tree_sum(tiles.cpu().numpy().reshape(n, 8, 8))
            assert np.array_equal(out.cpu().numpy().reshape(8, 8).view(np.uint32), ref.view(np.uint32)), "sum differs"
        print(f"n = {n:9d} tiles ({n * 256 / 1e6:8.1f} MB): {t * 1e6:9.1f} us  "
              f"{n * 192 / t / 1e9:7.1f} GB/s read (192 B/tile), {n * 168 / t / 1e9:7.1f} GB/s algorithmic (168 B/tile)")
    cs.cleanup()


if __name__ == "__main__":
    main()
