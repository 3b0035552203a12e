"""Stress run of the ragged-ring parity test over many more seeds than the test suite holds:
    python tools/fuzz_ragged.py [first_seed] [count]"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudprocessing_amd import ComputeShader
import tests.test_gpu_registration as T
cs = ComputeShader(maxPointsPerCloud=200000, maxScans=100)
bad = 0
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for seed in range(first, first + count):
    try:
        T.test_ragged_ring_layouts_match_oracle_bitwise(cs, seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:100], flush=True)
    if seed % 50 == 0:
        print("seed", seed, "done", flush=True)
print("failures:", bad)
cs.cleanup()
