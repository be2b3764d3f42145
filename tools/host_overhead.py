#!/usr/bin/env python3
"""Host-side cost of one op call through the two bindings (ctypes vs the compiled C++ extension) on a
Cora-shape graph, where the kernels take ~6 us and the call path dominates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import graphs, functions
from custom_op_benchmark_amd import graphop as ops

dev = torch.device("cuda:0")
g = graphs.chung_lu_graph(2708, 10556, alpha=0.5, seed=0, device=dev)
Q, K, V, dO = (torch.rand(2708, 64, device=dev) for _ in range(4))
a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
ext = ops.cpp_ext


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()          # host time to ENQUEUE n calls (the GPU runs behind)
    torch.cuda.synchronize()
    return 1e6 * (t1 - t0) / n, 1e6 * (time.perf_counter() - t0) / n


print("maskedmm_csr_forward  ctypes  : enqueue %.1f us, wall %.1f us" % timeit(lambda: ops.maskedmm_csr_forward(*a4, Q, K)))
if ext is not None:
    print("maskedmm_csr_forward  C++ ext : enqueue %.1f us, wall %.1f us" % timeit(lambda: ext.maskedmm_csr_forward(*a4, Q, K)))
    print("torch.ops.graphop (C++ reg.)  : enqueue %.1f us, wall %.1f us" % timeit(lambda: torch.ops.graphop.maskedmm_csr_forward(*a4, Q, K)))
s = ops.maskedmm_csr_forward(*a4, Q, K)
print("sparse_softmax_forward ctypes : enqueue %.1f us, wall %.1f us" % timeit(lambda: ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s)))
if ext is not None:
    print("sparse_softmax_forward C++ ext: enqueue %.1f us, wall %.1f us" % timeit(lambda: ext.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s)))
q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))


def step():
    q.grad = k.grad = v.grad = None
    functions.attention_step(g, q, k, v, dO)


print("attention_step (autograd, 8 ops): enqueue %.1f us, wall %.1f us" % timeit(step, 500))
