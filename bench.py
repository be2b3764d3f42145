#!/usr/bin/env python3
"""Headline benchmark: edges/sec for one fwd+bwd of SDDMM -> row-softmax -> SpMM on a
Reddit-shaped graph (N=232,965, E=114,615,892, d=64, 1 head, fp32), plus the HBM roofline of the
dominant kernel and the CPU PyTorch scatter/gather baseline (BASELINE.json metric).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = s = MaskedMMCSR(Q,K); a = SparseSoftmax(s); o = VectorSPMM(a,V); o.backward(dO)
through the product's autograd classes (the reference's usage, wrapper.py:201-206,231-239,291-299).
Inputs are synthetic (no datasets in the image): Chung-Lu power-law graph, U[0,1) features like
the reference harness (wrapper.py:151-153); they are resident in HBM before the timed region.
Graph preprocessing (CSR build, partition_csr, per-graph plans) is setup and reported separately.

At N > 1 the graph is node-range partitioned (custom_op_benchmark_amd.dist): every rank owns the
rows of one Reddit-shaped shard of an N-times larger graph (weak scaling) and exchanges halo
K/V rows and dK/dV partial rows by RCCL all-to-all each step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)
PASS_TAGS = ["sddmm_fwd", "softmax_fwd", "spmm_fwd", "spmm_bwd_dedata", "spmm_bwd_dx", "softmax_bwd",
             "sddmm_bwd_dA", "sddmm_bwd_dB"]


def pass_bytes(E, N_rows, N_cols, h, d, C, Cc):
    """ALGORITHMIC bytes per pass: API dtypes (int64 ids, fp32 values), every operand
    read once, every output written once, node rows once per pass, zero-fill/scratch not counted."""
    F4 = 4 * h * d
    gather = lambda n_out, n_in, chunks: E * (16 + 4 * h) + (n_out + n_in) * F4 + 16 * chunks
    return {
        "sddmm_fwd": gather(N_rows, N_cols, C),
        "softmax_fwd": E * (8 + 8 * h) + 16 * C,
        "spmm_fwd": gather(N_rows, N_cols, C),
        "spmm_bwd_dedata": gather(N_rows, N_cols, C),
        "spmm_bwd_dx": gather(N_cols, N_rows, Cc),
        "softmax_bwd": E * (8 + 12 * h) + 16 * C,
        "sddmm_bwd_dA": gather(N_rows, N_cols, C),
        "sddmm_bwd_dB": gather(N_cols, N_rows, Cc),
    }


KERNEL_OF = {  # pass tag -> device kernel family that executes it
    "sddmm_fwd": "k_sddmm_f32", "spmm_bwd_dedata": "k_sddmm_f32",
    "spmm_fwd": "k_spmm_f32", "spmm_bwd_dx": "k_spmm_f32", "sddmm_bwd_dA": "k_spmm_f32",
    "sddmm_bwd_dB": "k_spmm_f32", "softmax_fwd": "k_softmax_fwd_seg", "softmax_bwd": "k_softmax_bwd_seg",
}


def cpu_baseline(g, Q, K, V, dO, sample_edges, log):
    """The reference's CPU-runnable path (stock PyTorch gather/scatter, oracle/torch_path.py) timed
    on this box's host cores on a bounded row-block sample of the same graph."""
    from oracle import torch_path
    E = g.n_edges
    ip = g.indptr_r
    target = min(E, sample_edges)
    R = int(torch.searchsorted(ip, torch.tensor([target], device=ip.device))[0].item())
    R = max(1, min(R, g.n_src))
    ipc = ip[: R + 1].cpu()
    e1 = int(ipc[-1])
    dst = g.indices_r[:e1].cpu()
    src = torch.repeat_interleave(torch.arange(R), ipc[1:] - ipc[:-1])
    Qc, Kc, Vc, dOc = Q[:R].cpu(), K.cpu(), V.cpu(), dO[:R].cpu()
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        torch_path.attention_step_blocked(src, dst, ipc, Qc, Kc, Vc, dOc, R, rows_per_block=2048)
        times.append(time.perf_counter() - t0)
        log("cpu_baseline rep %d: %.2f s for %d edges" % (it, times[-1], e1))
    times = sorted(times[1:])
    med = times[len(times) // 2]
    return {"value": e1 / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "rows [0,%d) of the same graph = %d edges (%.2f%% of E), K/V/dK/dV full size; "
                      "stock-PyTorch gather/scatter step (oracle/torch_path.py), median of 3 after 1 warm-up, "
                      "%.2f s/step; host %s, os.cpu_count()=%d" % (R, e1, 100.0 * e1 / E, med, _cpu_model(), os.cpu_count())}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", default="reddit", help="reddit | products | cora | harness | custom")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--d", type=int, default=64, help="per-head feature dim")
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=0.5, help="Chung-Lu power-law exponent (0 = uniform)")
    ap.add_argument("--chunk-size", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample-edges", type=int, default=3_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single-GPU rehearsal: build rank 0's shard of a WORLD-way partition and time its "
                         "local compute (collectives replaced by local copies); not a headline number")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture the step once into a HIP graph and time its replays (single GPU; for the "
                         "launch-bound small shapes -- the headline line is measured with eager API calls)")
    ap.add_argument("--no-fused", action="store_true", help="skip the secondary measurement of the fused op")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d needs torch.distributed.run (one rank per GPU)" % args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)

    def log(msg):
        if args.verbose or os.environ.get("BENCH_VERBOSE"):
            print("[bench r%d] %s" % (rank, msg), file=sys.stderr, flush=True)

    from custom_op_benchmark_amd import _lib, functions, graphs
    _lib.lib()  # fail loudly now if the extension is missing

    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GRAPHOP_DIST_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing a GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.graph == "custom" or args.nodes or args.edges:
        N, E = args.nodes, args.edges
        name = "custom"
    else:
        N, E = graphs.SHAPES[args.graph]
        name = args.graph
    h, d = args.heads, args.d

    # ---- setup (not timed as part of a step; reported) -------------------------------------------
    t_setup = time.perf_counter()
    if world == 1 and args.emulate_world > 1:
        from custom_op_benchmark_amd import dist as gdist
        runner = gdist.ShardedAttention.synthetic(N, E, args.emulate_world, 0, dev, alpha=args.alpha,
                                                  seed=args.seed, chunk_size=args.chunk_size, emulate=True)
        g = runner.graph
        n_rows, n_cols = g.n_src, g.n_dst
    elif world == 1 and name == "harness":
        # the reference author's own fixture: 512 disjoint complete digraphs of 30 nodes (wrapper.py:79-112)
        g = graphs.block_diagonal_graph(512, 30, chunk_size=args.chunk_size, device=dev)
        runner = None
        n_rows, n_cols = g.n_src, g.n_dst
    elif world == 1:
        g = graphs.chung_lu_graph(N, E, alpha=args.alpha, seed=args.seed, chunk_size=args.chunk_size, device=dev)
        runner = None
        n_rows, n_cols = g.n_src, g.n_dst
    else:
        from custom_op_benchmark_amd import dist as gdist
        runner = gdist.ShardedAttention.synthetic(N, E, world, rank, dev, alpha=args.alpha, seed=args.seed,
                                                  chunk_size=args.chunk_size)
        g = runner.graph
        n_rows, n_cols = g.n_src, g.n_dst
    gen = torch.Generator(device=dev).manual_seed(args.seed + 1 + rank)
    shp = (lambda n: (n, d) if h == 1 else (n, h, d))
    n_own = N if runner is None else runner.n_own
    Q = torch.rand(shp(n_own), device=dev, generator=gen).requires_grad_(True)
    K = torch.rand(shp(n_own), device=dev, generator=gen).requires_grad_(True)
    V = torch.rand(shp(n_own), device=dev, generator=gen).requires_grad_(True)
    dO = torch.rand(shp(n_own), device=dev, generator=gen)
    torch.cuda.synchronize()
    t_graph = time.perf_counter() - t_setup

    def step():
        Q.grad = K.grad = V.grad = None
        if runner is None:
            functions.attention_step(g, Q, K, V, dO)
        else:
            runner.step(Q, K, V, dO)

    t0 = time.perf_counter()
    step()                                   # first call builds + caches the per-graph plans
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    log("graph %s N=%d E=%d C=%d C'=%d built in %.1f s; first step (plans) %.2f s" %
        (name, n_rows, g.n_edges, g.n_row_chunks, g.n_col_chunks, t_graph, t_first))

    for _ in range(args.warmup):
        step()

    timed_step = step
    if args.hip_graph:
        if runner is not None or world > 1:
            raise SystemExit("--hip-graph is a single-GPU option")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        Q.grad = K.grad = V.grad = None
        hip_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(hip_graph):
            functions.attention_step(g, Q, K, V, dO)
        timed_step = hip_graph.replay
        timed_step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ee = torch.tensor([g.n_edges], device=cdev, dtype=torch.int64)
        dist.all_reduce(ee)
        total_edges = int(ee.item())
    else:
        total_edges = g.n_edges
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_edges * args.steps / elapsed

    # ---- the same step through the fused op (extra op: one autograd node, no E-sized intermediates) ----
    fused = None
    if runner is None and world == 1 and not args.no_fused and not args.hip_graph:
        def fstep():
            Q.grad = K.grad = V.grad = None
            functions.fused_attention_step(g, Q, K, V, dO)
        for _ in range(max(2, args.warmup)):
            fstep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fstep()
        torch.cuda.synchronize()
        f_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        _lib.profile_enable(True)
        for _ in range(max(1, args.profile_steps)):
            fstep()
        torch.cuda.synchronize()
        fprof = _lib.profile_read()
        _lib.profile_enable(False)
        fused = {"ms_per_step": round(f_ms, 4), "value": g.n_edges / (f_ms * 1e-3), "unit": "edges/s",
                 "op": "FusedAttention (attention_forward / attention_backward)",
                 "passes_ms": {k: round(v["mean_ms"], 4) for k, v in fprof.items()}}

    # ---- per-kernel durations, live, hipEvents on the launch stream --------------------------------
    _lib.profile_enable(True)
    for _ in range(max(1, args.profile_steps)):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)

    pb = pass_bytes(g.n_edges, n_rows, n_cols, h, d, g.n_row_chunks, g.n_col_chunks)
    alg_step = sum(pb.values())
    passes = {}
    kern = {}
    for tag in PASS_TAGS:
        if tag not in prof:
            continue
        ms = prof[tag]["mean_ms"]
        passes[tag] = {"ms": round(ms, 4), "alg_GB": round(pb[tag] / 1e9, 4),
                       "alg_GBps": round(pb[tag] / 1e6 / ms, 1), "frac": round(pb[tag] / 1e6 / ms / HBM_PEAK_GBS, 4)}
        k = kern.setdefault(KERNEL_OF[tag], {"ms": 0.0, "bytes": 0.0, "launches": 0})
        k["ms"] += ms; k["bytes"] += pb[tag]; k["launches"] += 1
    dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"]) if kern else ("none", {"ms": 1.0, "bytes": 0.0, "launches": 1})
    achieved = dom["bytes"] / 1e6 / dom["ms"]          # GB/s = algorithmic bytes per launch / mean launch time
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic_note = None
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == "%s_h%d_d%d" % (name, h, d) and dom_name in tj.get("kernels", {}):
                traffic = tj["kernels"][dom_name]["hbm_bytes_per_launch"]
                traffic_note = tj.get("note")
        except (ValueError, KeyError):
            pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": dom_name,
                "launches_per_step": dom["launches"],
                "alg_bytes_per_launch": int(dom["bytes"] / max(1, dom["launches"])),
                "avg_launch_ms": round(dom["ms"] / max(1, dom["launches"]), 4),
                "step": {"alg_bytes": int(alg_step), "alg_GBps": round(alg_step / 1e6 / ms_per_step, 1),
                         "frac": round(alg_step / 1e6 / ms_per_step / HBM_PEAK_GBS, 4),
                         "kernel_ms_sum": round(sum(p["ms"] for p in passes.values()), 3)},
                "passes": passes}
    if traffic_note:
        roofline["traffic_note"] = traffic_note
    # block-dense workloads (harness fixture): the gather passes run as 32x32 fp32-MFMA tiles; report
    # tile flops against the dense fp32 MFMA peak next to the HBM figure (still the binding roofline)
    try:
        pinfo = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r).info
        if pinfo.n_dense_blocks > 0 and pinfo.dense_fill_pct >= 40 and d % 32 == 0:
            gather_ms = sum(passes[t]["ms"] for t in passes if not t.startswith("softmax"))
            n_gather = sum(1 for t in passes if not t.startswith("softmax"))
            tile_flops = 2.0 * 32 * 32 * d * h * pinfo.n_dense_blocks       # per gather pass, padded tiles
            tf = tile_flops * n_gather / (gather_ms * 1e-3) / 1e12
            roofline["mfma"] = {"instr": "v_mfma_f32_32x32x2_f32", "tile_TFLOPs": round(tf, 2),
                                "useful_TFLOPs": round(tf * pinfo.dense_fill_pct / 100.0, 2),
                                "peak_TFLOPs": 157.3, "frac": round(tf / 157.3, 4),
                                "blocks": int(pinfo.n_dense_blocks), "tile_fill_pct": int(pinfo.dense_fill_pct)}
    except Exception as exc:      # reporting only
        log("mfma report skipped: %r" % (exc,))

    out = {
        "metric": "edges/sec fwd+bwd (SDDMM+softmax+SpMM) on Reddit d=64; HBM GB/s vs roofline",
        "value": value, "unit": "edges/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s-shape Chung-Lu(alpha=%.2f) graph, N=%d E=%d per GPU, h=%d d=%d, chunk_size=%d, "
                               "int64 CSR both orientations" % (name, args.alpha, n_rows if runner is None else runner.n_own,
                                                                 g.n_edges, h, d, args.chunk_size),
                   "graph": name, "nodes": n_rows, "edges": total_edges, "heads": h, "d": d,
                   "chunk_size": args.chunk_size, "row_chunks": g.n_row_chunks, "col_chunks": g.n_col_chunks,
                   "parallelism": ("single GPU" if world == 1 else "node-range shards x%d, RCCL all-to-all halo" % world)
                                  + (" [EMULATED shard 0 of %d, no collectives]" % args.emulate_world if args.emulate_world > 1 else "")},
        "setup": {"graph_build_s": round(t_graph, 2), "first_step_with_plans_s": round(t_first, 3)},
        "launch": "hip graph replay" if args.hip_graph else "eager API calls",
        "roofline": roofline,
    }
    if fused is not None:
        out["fused"] = fused
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, Q.detach(), K.detach(), V.detach(), dO, args.cpu_sample_edges, log)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
