"""The N-rank path of bench.py as the driver launches it (torch.distributed.run, one process per rank), rehearsed with
two ranks that share this box's GPU: control traffic over gloo (PCP_BENCH_BACKEND=gloo; RCCL refuses two ranks on one
device), the scan-to-map leg in child processes with the tile exchanged over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_on_one_card():
    with socket.socket() as sk:           # a rendezvous port nobody holds
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PCP_BENCH_BACKEND="gloo", PCP_BENCH_LEG_SECONDS="200")
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(name, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]              # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["pairs_per_gpu_per_step"] == 256
    # whole-job value: both ranks' pairs over the slower rank's time
    assert abs(out["value"] - 2 * 256 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    leg = out["scan_to_map"]
    assert "error" not in leg, leg
    assert leg["all_ranks_hold_the_same_steps"] is True and leg["iterations"] == 10
    assert len(leg["slab_points_per_rank"]) == 2 and sum(leg["slab_points_per_rank"]) >= leg["map_points"]     # shares + halos
    assert max(leg["slab_points_per_rank"]) < 0.75 * leg["map_points"]                                       # and really split
