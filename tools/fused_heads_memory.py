#!/usr/bin/env python3
"""FusedAttention over several heads (round-4 verdict item 5): time per step and peak device memory ADDED by a step
(torch.cuda.max_memory_allocated minus what is allocated before it: graph arrays, plans and the leaf tensors are the
same for every form) of the 8-function step as a model would run it (s is not held), FusedAttention in head groups with
a_g kept, and with a_g recomputed.
    python tools/fused_heads_memory.py --graph reddit --heads 8 --d 32      | --graph products --heads 8 --d 16"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, graphs, functions

ap = argparse.ArgumentParser()
ap.add_argument("--graph", default="reddit"); ap.add_argument("--heads", type=int, default=8); ap.add_argument("--d", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, E = graphs.SHAPES[args.graph]
h, d = args.heads, args.d
g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
Q, K, V, dO = (torch.rand(N, h, d, device=dev, generator=gen) for _ in range(4))
for t in (Q, K, V): t.requires_grad_(True)
a8 = g.csr_args()


def unfused():
    Q.grad = K.grad = V.grad = None
    o = functions.VectorSPMM.apply(*a8, functions.SparseSoftmax.apply(g.row, g.ptr_r, g.eid_r, functions.MaskedMMCSR.apply(*a8, Q, K)), V)
    o.backward(dO)


def fused():
    Q.grad = K.grad = V.grad = None
    functions.FusedAttention.apply(*a8, Q, K, V).backward(dO)


def measure(name, step):
    for _ in range(2): step()
    Q.grad = K.grad = V.grad = None
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps): step()
    t1.record(); torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    print("%-44s %8.2f ms/step   step peak %7.2f GB  (= %.2f edge tensors of E x h floats; resident before the step %.2f GB, plans %.2f GB)"
          % (name, t0.elapsed_time(t1) / args.steps, peak / 2**30, peak / (E * h * 4), base / 2**30, _lib.plan_memory_bytes() / 2**30), flush=True)
    return peak


print("%s-shape N=%d E=%d h=%d d=%d (head groups of %d)" % (args.graph, N, E, h, d, functions._head_group(h, d)), flush=True)
p0 = measure("8-function step (s not held)", unfused)
for mode in ("keep", "recompute"):
    functions.FUSED_HEADS_MODE = mode
    p = measure("FusedAttention, head groups, a_g %s" % ("kept" if mode == "keep" else "recomputed"), fused)
    print("   -> %.2f x the 8-function step's peak" % (p / p0), flush=True)
