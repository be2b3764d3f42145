#!/usr/bin/env python3
"""Knob sweep for the gather passes: build one graph, then time the attention step per pass
(library hipEvent profiler) under a list of tuning-knob settings.

  python tools/tune_sweep.py [--graph reddit] [--d 64] [--heads 1] "vrow_t=1024" "walk=0" ...

Each positional argument is one setting (comma-separated key=value pairs on top of the defaults).
Prints one line per setting: step ms and the per-pass ms.  Speed only: results never depend on the
knobs (tests/test_hip_parity.py covers that)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, graphs, functions

DEFAULTS = dict(sweep=1, window_kb=4096, mall_window_kb=32768, max_windows=128,
                sweep_min_kb=4608, sweep_bpc=3, sweep_k=0, vrow_t=0, sweep_min_granule=4,
                sweep_w=0, spmm_window_scale=2,
                attn_fused=1, attn_window_scale=2, attn_k=0, attn_bpc=0, touch_sddmm=1, staged_ids=7,
                walk=6, walk_window_kb=4096, walk_window_kb_col=2048, walk_drift=3, walk_min_bin=1024, walk_blocks=0, walk_steps=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="reddit")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--fused", action="store_true", help="time the fused op (FusedAttention) instead of the 8-function step")
    ap.add_argument("settings", nargs="*", default=[""])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, E = graphs.SHAPES[args.graph]
    N, E = args.nodes or N, args.edges or E
    g = graphs.chung_lu_graph(N, E, alpha=args.alpha, seed=0, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    shp = (N, args.d) if args.heads == 1 else (N, args.heads, args.d)
    Q, K, V, dO = (torch.rand(shp, device=dev, generator=gen) for _ in range(4))
    for t in (Q, K, V):
        t.requires_grad_(True)
    order = ["sddmm_fwd", "softmax_fwd", "spmm_fwd", "spmm_bwd_dedata", "spmm_bwd_dx", "softmax_bwd",
             "sddmm_bwd_dA", "sddmm_bwd_dB"]
    step = functions.attention_step
    if args.fused:
        order = ["sddmm_fwd", "softmax_fwd", "spmm_fwd", "attn_pack", "attn_bwd_row", "attn_bwd_col", "attn_rows_row", "attn_rows_col"]
        step = functions.fused_attention_step
    print("# N=%d E=%d h=%d d=%d ; columns: step-wall (sum of pass times) " % (N, E, args.heads, args.d) + " ".join(order), flush=True)
    for setting in args.settings:
        knobs = dict(DEFAULTS)
        for kv in filter(None, setting.split(",")):
            k, v = kv.split("=")
            knobs[k] = int(v)
        for k, v in knobs.items():
            _lib.tune(k, v)
        _lib.clear_plan_cache()
        for _ in range(2):
            step(g, Q, K, V, dO)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(args.steps):
            step(g, Q, K, V, dO)
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.steps):
            step(g, Q, K, V, dO)
        t1.record(); torch.cuda.synchronize()
        wall = t0.elapsed_time(t1) / args.steps
        ms = [prof[n]["mean_ms"] if n in prof else float("nan") for n in order]
        print("%-44s %7.3f (sum of passes %7.3f) | %s" % (setting or "(defaults)", wall, sum(ms), " ".join("%6.3f" % m for m in ms)), flush=True)
    for k, v in DEFAULTS.items():
        _lib.tune(k, v)


if __name__ == "__main__":
    main()
