#!/usr/bin/env python3
"""Would an internal node reordering pay (SURVEY.md 7.3-1, round-3 verdict "missing" item 6)?  The same Reddit-shape
graph under three labelings of its nodes -- shuffled (the generator's default: hubs spread over all ids), sorted by
degree (hubs first: the first column windows hold most slots, hot rows share L2 lines) and sorted by degree in
alternating directions per block -- timed per pass.  A plan-level reordering would permute the gathered table per
call (60 MB at d = 64, ~0.03 ms) and un-permute the outputs; this measures what it could buy.
    python tools/reorder_experiment.py [--d 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, graphs, functions

ap = argparse.ArgumentParser(); ap.add_argument("--d", type=int, default=64); ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
N, E = graphs.SHAPES["reddit"]
g0 = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
deg = (g0.indptr_r[1:] - g0.indptr_r[:-1]) + (g0.indptr_c[1:] - g0.indptr_c[:-1])
order = torch.argsort(deg, descending=True)                    # new id -> old id
rank_sorted = torch.empty_like(order); rank_sorted[order] = torch.arange(N, device=dev)
# hubs dealt round-robin over 64 blocks of ids (every block of N / 64 ids gets its share of hubs, sorted inside)
blocks = 64
pos = torch.arange(N, device=dev)
rank_striped = torch.empty_like(order); rank_striped[order] = (pos % blocks) * (N // blocks + 1) + pos // blocks
rank_striped = torch.argsort(torch.argsort(rank_striped))      # compress to 0 .. N - 1
src, dst = g0.src.clone(), g0.dst.clone()
del g0
torch.cuda.empty_cache()
for name, rank in (("shuffled ids (default)", None), ("degree-sorted ids (hubs first)", rank_sorted), ("hubs striped over 64 id blocks", rank_striped)):
    s, d_ = (src, dst) if rank is None else (rank[src], rank[dst])
    g = graphs.graph_from_coo(s, d_, N, N, 32)
    gen = torch.Generator(device=dev).manual_seed(1)
    Q, K, V, dO = (torch.rand(N, args.d, device=dev, generator=gen) for _ in range(4))
    for t in (Q, K, V): t.requires_grad_(True)
    for _ in range(2): functions.attention_step(g, Q, K, V, dO)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps): functions.attention_step(g, Q, K, V, dO)
    t1.record(); torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print("%-34s d=%d step %.2f ms |" % (name, args.d, t0.elapsed_time(t1) / args.steps),
          " ".join("%s %.2f" % (k, v["mean_ms"]) for k, v in prof.items() if k != "zero_fill"), flush=True)
    del g, Q, K, V, dO
    _lib.clear_plan_cache(); torch.cuda.empty_cache()
