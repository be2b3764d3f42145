#!/usr/bin/env python3
"""How much of a window-driver pass is lost to the max over the 4 lane groups of a wave and to
batch round-up?  Rebuilds the (vrow, window) granule lengths of the Reddit-shape graph the way
plan.hip does (T-slot pieces taking 1/P of every window) and compares, per wave task (4 groups x K
vrows in one window), max-over-groups batches with the mean.   python tools/divergence_model.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import graphs

dev = torch.device("cuda:0")
N, E = graphs.SHAPES["reddit"]
g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
for W, T, K in ((16, 512, 8), (8, 1024, 8), (16, 1024, 4)):
    win_cols = -(-N // W)
    cnt = torch.bincount(g.src * W + g.dst // win_cols, minlength=N * W).view(N, W)        # granule lengths per row
    deg = cnt.sum(1)
    P = torch.clamp((deg + T - 1) // T, min=1)
    rows = torch.repeat_interleave(torch.arange(N, device=dev), P)                              # vrow -> row
    first = torch.cumsum(P, 0) - P
    piece = torch.arange(rows.numel(), device=dev) - first[rows]
    q = (cnt[rows] + P[rows, None] - 1) // P[rows, None]
    lo = torch.minimum(piece[:, None] * q, cnt[rows]); hi = torch.minimum((piece[:, None] + 1) * q, cnt[rows])
    gl = (hi - lo)                                                                                # (V, W) granule lengths
    V = gl.shape[0]
    tile = 4 * K
    pad = (-V) % tile
    glp = torch.cat([gl, gl.new_zeros(pad, W)]).view(-1, 4, K, W)                                # task, group, k, window
    gsum = glp.sum(2)                                                                             # slots per (task, group, window)
    batches = (gsum + 15) // 16
    wave = batches.max(1).values.sum().item()                                                     # wave-steps actually spent
    ideal_groups = batches.sum().item() / 4.0                                                     # if the 4 groups were balanced
    ideal_slots = gl.sum().item() / 64.0                                                          # no round-up either
    print("W=%d T=%d K=%d: V=%d  wave batch-steps %.3g | balanced groups %.3g (%.1f%% less) | no round-up %.3g (%.1f%% less)"
          % (W, T, K, V, wave, ideal_groups, 100 * (1 - ideal_groups / wave), ideal_slots, 100 * (1 - ideal_slots / wave)))
