#!/usr/bin/env python3
"""Run the randomised parity battery of tests/test_hip_parity.py (shapes, heads, chunk sizes, both
window orders, forced knobs) over many more seeds than the test suite does:
    python tools/soak_fuzz.py [first_seed last_seed]      # default 40 340, needs an MI355X"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_hip_parity as T
dev = torch.device("cuda:0")
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (40, 340)
for seed in range(lo, hi):
    try:
        T.test_fuzz_shapes_and_paths(dev, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED", repr(e)[:300], flush=True)
    if seed % 50 == 0:
        print("seed", seed, "ok so far, failures:", bad, flush=True)
print("done, failures:", bad)
