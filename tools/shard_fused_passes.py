"""Where the fused attention passes stand on a papers100M-shape 1/8 shard (n_own x (n_own + n_halo) local graph, d = 128):
the chunk-driver forms of the fused backward (k_attn_bwd_rows_f32: attn_rows_row gathers K|V, attn_rows_col gathers Q|dO and
the row statistics -- the column pass recomputes (a, ds) per slot, no E-sized scalar is read) against the unfused passes.
    python tools/shard_fused_passes.py        (one MI355X, ~60 GB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib, functions, graphs
from custom_op_benchmark_amd import dist as gdist

dev = torch.device("cuda:0")
N, E = graphs.SHAPES["papers100m"]
N, E = N // 8, E // 8
runner = gdist.ShardedAttention.synthetic(N, E, 8, 0, dev, alpha=0.5, seed=0, cut=0.1, timing_only=True)
g = runner.graph
d = 128
gen = torch.Generator(device=dev).manual_seed(1)
Q = torch.rand(g.n_src, d, device=dev, generator=gen).requires_grad_(True)
K = torch.rand(g.n_dst, d, device=dev, generator=gen).requires_grad_(True)
V = torch.rand(g.n_dst, d, device=dev, generator=gen).requires_grad_(True)
dO = torch.rand(g.n_src, d, device=dev, generator=gen)
print("local graph %d x %d, E = %d" % (g.n_src, g.n_dst, g.n_edges), flush=True)


def run(name, step, reps=3):
    for _ in range(2):
        Q.grad = K.grad = V.grad = None
        step()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        Q.grad = K.grad = V.grad = None
        step()
    t1.record(); torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    print("%-28s %.2f ms/step | %s" % (name, t0.elapsed_time(t1) / reps,
                                       " ".join("%s %.2f" % (k, v["total_ms"] / reps) for k, v in prof.items())), flush=True)


dO_ext = torch.cat([dO, torch.zeros(g.n_dst - g.n_src, d, device=dev)])     # the reference's SpMM output is zeros_like(x): n_dst rows
run("8-function step", lambda: functions.attention_step(g, Q, K, V, dO_ext))
del dO_ext
_lib.tune("attn_rows", 1); _lib.tune("attn_max_d", 128)
run("FusedAttention (rows forms)", lambda: functions.fused_attention_step(g, Q, K, V, dO))
