#!/usr/bin/env python3
"""Which side of a degree-correlated node labeling costs what (round-4 verdict item 2; profiles/r4_experiments.txt
section 4 only measured both sides together)?  The SAME Reddit-shape edge set with its ROW ids and its COLUMN ids
relabelled independently (the operators never assume the two label spaces are related: A / B are separate tensors):
shuffled | degree-sorted, per side, timed per pass at the default geometry.
    python tools/labeling_experiment.py [--d 64] [--cases ss,dd,ds,sd]      (first letter = rows, second = columns)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, graphs, functions

ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=64)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--cases", default="ss,dd,ds,sd")
ap.add_argument("--fused", action="store_true")
ap.add_argument("--tune", default="", help="knob=value,... applied before every case (graphop_tune)")
ap.add_argument("--graph", default="chung_lu", help="chung_lu | clustered (graphs.chung_lu_graph labeling)")
args = ap.parse_args()
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    _lib.tune(k, int(v))
dev = torch.device("cuda:0")
N, E = graphs.SHAPES["reddit"]
g0 = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev, labeling="clustered" if args.graph == "clustered" else "shuffled")
deg_r = g0.indptr_r[1:] - g0.indptr_r[:-1]
deg_c = g0.indptr_c[1:] - g0.indptr_c[:-1]


def rank_of(deg):
    order = torch.argsort(deg, descending=True)
    r = torch.empty_like(order); r[order] = torch.arange(N, device=dev)
    return r


rank = {"r": rank_of(deg_r), "c": rank_of(deg_c)}
src, dst = g0.src.clone(), g0.dst.clone()
del g0
torch.cuda.empty_cache()
for case in args.cases.split(","):
    s = rank["r"][src] if case[0] == "d" else src
    d_ = rank["c"][dst] if case[1] == "d" else dst
    g = graphs.graph_from_coo(s, d_, N, N, 32)
    del s, d_
    gen = torch.Generator(device=dev).manual_seed(1)
    Q, K, V, dO = (torch.rand(N, args.d, device=dev, generator=gen) for _ in range(4))
    for t in (Q, K, V): t.requires_grad_(True)
    step = (lambda: functions.fused_attention_step(g, Q, K, V, dO)) if args.fused else (lambda: functions.attention_step(g, Q, K, V, dO))
    for _ in range(2): step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps): step()
    t1.record(); torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print("[%s] rows %-8s cols %-8s d=%d step %.2f ms |" % (args.tune or "defaults", "degree" if case[0] == "d" else "shuffled", "degree" if case[1] == "d" else "shuffled",
                                                     args.d, t0.elapsed_time(t1) / args.steps),
          " ".join("%s %.2f" % (k, v["mean_ms"]) for k, v in prof.items() if k != "zero_fill"), flush=True)
    del g, Q, K, V, dO
    _lib.clear_plan_cache(); torch.cuda.empty_cache()
