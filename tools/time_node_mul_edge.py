#!/usr/bin/env python3
"""Time node_mul_edge fwd/bwd (next-row N1) on a Reddit-shaped graph; prints ms and GB/s of the
E x d edge-feature stream (the op is pure HBM streaming over B (E, d))."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import graphs, graphop as ops

N, E = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else graphs.SHAPES["reddit"]
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda:0")
g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
A = torch.rand((N, d) if h == 1 else (N, h, d), device=dev)
B = torch.rand(E, d, device=dev)
dy = torch.rand((E,) if h == 1 else (E, h), device=dev)
a3 = (g.row, g.ptr_r, g.eid_r)
for name, fn, nbytes in (("forward", lambda: ops.node_mul_edge_forward(*a3, A, B), E * d * 4 + E * h * 4),
                         ("backward", lambda: ops.node_mul_edge_backward(*a3, A, B, dy), 2 * E * d * 4 + 2 * E * h * 4)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print("node_mul_edge %s h=%d d=%d: %.2f ms  %.0f GB/s (edge-feature stream %.1f GB)" % (name, h, d, ms, nbytes / ms / 1e6, nbytes / 1e9))
