#!/usr/bin/env python3
"""fp64 on the Reddit shape (the reference dispatches fp32 and fp64 through the same kernels, graphop_kernel.cu:291;
here fp64 takes the generic kernels + the plan's row-segment softmax): per-pass times of the 8-function step.
    python tools/time_fp64.py [--d 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, graphs, functions

ap = argparse.ArgumentParser(); ap.add_argument("--d", type=int, default=64); ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, E = graphs.SHAPES["reddit"]
g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
for dt in (torch.float32, torch.float64):
    Q, K, V, dO = (torch.rand(N, args.d, device=dev, generator=gen, dtype=dt) for _ in range(4))
    for t in (Q, K, V): t.requires_grad_(True)
    for _ in range(2): functions.attention_step(g, Q, K, V, dO)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps): functions.attention_step(g, Q, K, V, dO)
    t1.record(); torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print(str(dt), "step %.2f ms |" % (t0.elapsed_time(t1) / args.steps),
          " ".join("%s %.2f (%s)" % (k, v["mean_ms"], v["kernel"]) for k, v in prof.items() if k != "zero_fill"), flush=True)
