#!/usr/bin/env python3
"""Headline benchmark: edges/sec for one fwd+bwd of SDDMM -> row-softmax -> SpMM on a
Reddit-shaped graph (N=232,965, E=114,615,892, d=64, 1 head, fp32), plus the HBM roofline of the
dominant kernel and the CPU PyTorch scatter/gather baseline (BASELINE.json metric).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" = s = MaskedMMCSR(Q,K); a = SparseSoftmax(s); o = VectorSPMM(a,V); o.backward(dO)
through the product's autograd classes (the reference's usage, wrapper.py:201-206,231-239,291-299):
the reference's eight-function operator surface.  `value` is ALWAYS that step.  The same step
through the extra fused op (functions.FusedAttention) is measured next to it and reported under
"fused" -- a secondary figure, never `value`.
Inputs are synthetic (no datasets in the image): Chung-Lu power-law graph, U[0,1) features like
the reference harness (wrapper.py:151-153; --values normal gives N(0,1)/sqrt(d)); they are resident
in HBM before the timed region.  Graph preprocessing (CSR build, partition_csr, per-graph plans) is
setup and reported separately.

Workloads (--graph): N = 1 defaults to BASELINE.json config 2 (reddit).  N > 1 (and --emulate-world)
default to config 4: every rank owns one 1/8 shard of a papers100M-shaped graph (13.9 M nodes,
202 M edges, d = 128) -- weak scaling, node-range partition (custom_op_benchmark_amd.dist), halo K/V
rows and dK/dV partial rows by RCCL all-to-all each step.  `--graph rmat25` is config 5 (d = 256).
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The hosts of this pool only support dmabuf IPC: with the legacy IPC mode RCCL's (and torch's) cross-process buffer
# registration fails with `hipIpcGetMemHandle: invalid argument`.  The variable is read when the HSA runtime comes up,
# i.e. at the first GPU call of the process -- so it is set HERE, before torch is imported, not next to
# init_process_group (round 4 set it after torch.cuda.set_device had initialised the runtime: a no-op where it sat).
# The image exports it already; setdefault keeps an explicit choice of the caller.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, Chip-level parameters)
HBM_COPY_GBS = 6290.0   # measured device copy rate (MI355X_MICROARCH.md, chip level): informative only
# whole-row gathers of a table far beyond the Infinity Cache, each row fetched once: the guide's HBM row of
# "Indexed rows: gather into LDS" (MI355X_MICROARCH.md: 6.0-6.1 TB/s chip-wide; this repo's own random-row microbenchmark,
# profiles/r4_tlb_reach.txt, reads 5.9-6.0 from 8-64 GB tables).  The upper figure, so that a fraction of it stays <= 1
# (round 4 priced against 5.7 and printed 0.97-1.0 for the products shape)
HBM_RANDOM_ROW_GBS = 6100.0
# random 256-B row gathers by 16-lane groups from an L2-RESIDENT table, ids in registers, nothing else in the
# loop (tools/microbench/l2_gather.hip, profiles/r1_l2_resident_sweep.txt; 28.5 TB/s with all XCDs walking the
# windows in step, tools/microbench/cu_walk.hip, profiles/r2_cu_walk_microbench.txt): what a gather pass of a
# cache-resident table can reach on this chip however it is organised (every gathered edge moves a row from L2
# into a CU; the CU's 64 B/clk would allow 39 TB/s)
L2_GATHER_GBS = 30000.0
PASS_TAGS = ["sddmm_fwd", "softmax_fwd", "spmm_fwd", "spmm_bwd_dedata", "spmm_bwd_dx", "softmax_bwd",
             "sddmm_bwd_dA", "sddmm_bwd_dB"]
GATHER_TAGS = [t for t in PASS_TAGS if not t.startswith("softmax")]


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


def kernels_sha():
    """Hash of the kernel sources: a PMC traffic record is only quoted for the kernels it was
    collected on (profiles/pmc_traffic.json carries the hash of its build)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "custom_op_benchmark_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(g, Q, K, V, dO, sample_edges, log, incidence=False):
    """The reference's CPU-runnable path (stock PyTorch gather/scatter, oracle/torch_path.py) timed
    on this box's host cores on a bounded row-block sample of the same graph.  incidence=True adds
    the harness-verbatim copy-to-edge form through incidence matrices (wrapper.py:155-157, 57-75),
    which only fits small graphs (Cora-shape)."""
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
    kv_note = "K/V/dK/dV full size"
    if K.numel() * 4 > (1 << 30):
        # tables of several GB: keep only the key/value rows the sample touches (relabelled), so the CPU
        # step does not spend its time zero-filling and adding 10-GB gradient tables per row block
        uniq, dst = torch.unique(dst, return_inverse=True)
        Kc, Vc = K[uniq.to(K.device)].cpu(), V[uniq.to(V.device)].cpu()
        kv_note = "K/V/dK/dV restricted to the %d rows the sample touches" % uniq.numel()
    else:
        Kc, Vc = K.cpu(), V.cpu()
    Qc, dOc = Q[:R].cpu(), dO[:R].cpu()

    def timed(fn, reps):
        ts = []
        for it in range(reps + 1):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            log("cpu_baseline rep %d: %.3f s for %d edges" % (it, ts[-1], e1))
        return statistics.median(ts[1:])

    med = timed(lambda: torch_path.attention_step_blocked(src, dst, ipc, Qc, Kc, Vc, dOc, R, rows_per_block=2048), 3)
    out = {"value": e1 / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "rows [0,%d) of the same graph = %d edges (%.2f%% of E), %s; "
                     "stock-PyTorch gather/scatter step (oracle/torch_path.py), median of 3 after 1 warm-up, "
                     "%.3f s/step; host %s, os.cpu_count()=%d" % (R, e1, 100.0 * e1 / E, kv_note, med, _cpu_model(), os.cpu_count())}
    if incidence:
        med_i = timed(lambda: torch_path.attention_step_incidence(src, dst, Qc, Kc, Vc, dOc, R), 3)
        out["incidence_form"] = {"value": e1 / med_i, "unit": "edges/s", "s_per_step": round(med_i, 4),
                                 "what": "harness-verbatim copy-to-edge SDDMM through sparse incidence matrices "
                                         "(wrapper.py:155-157) + th.sparse.mm SpMM (wrapper.py:274), same sample"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", default="auto",
                    help="auto | reddit | products | cora | harness | papers100m | rmat25 | custom")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--edges", type=int, default=0)
    ap.add_argument("--d", "--dim", dest="d", type=int, default=0, help="per-head feature dim (0 = the workload's default)")
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=0.5, help="Chung-Lu power-law exponent (0 = uniform)")
    ap.add_argument("--labeling", default="shuffled", choices=["shuffled", "degree", "clustered"],
                    help="how node ids relate to the structure of the Chung-Lu graphs (graphs.chung_lu_graph): hubs spread "
                         "over the ids | ids sorted by degree | stochastic-block communities of 1024 consecutive ids "
                         "(p_in 0.9; generalises the reference fixture wrapper.py:84-112)")
    ap.add_argument("--values", default="uniform", help="uniform: U[0,1) like the harness | normal: N(0,1)/sqrt(d)")
    ap.add_argument("--cut", type=float, default=-1.0,
                    help="sharded graphs: fraction of a rank's edges whose destination is drawn from the GLOBAL "
                         "node distribution (the rest stay in the rank's range); default 0.1 for papers100m, 1 otherwise")
    ap.add_argument("--chunk-size", type=int, default=32)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-sample-edges", type=int, default=3_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single-GPU rehearsal: build rank 0's shard of a WORLD-way partition and time its "
                         "local compute (exchanges replaced by local copies of the right sizes); timing only")
    ap.add_argument("--rccl-self", action="store_true",
                    help="single GPU: run the sharded step through the REAL collective path -- a one-rank nccl (= RCCL) "
                         "process group, all_to_all_single with async handles and split-size views, no world == 1 short-cut -- "
                         "with the destinations drawn from the global distribution (--cut) fetched through the exchange as a "
                         "self-halo; use with --graph papers100m / rmat25")
    ap.add_argument("--no-autotune", action="store_true",
                    help="sharded steps: keep the default schedule (K and V exchanged separately, column-major backward passes "
                         "split) instead of measuring the schedules the shard supports and adopting the fastest (dist.autotune)")
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

    import datetime
    import torch.distributed as dist
    # a rank that dies during setup fails the run in two minutes (the other ranks' first collective times out), not at
    # the driver's limit; GRAPHOP_DIST_TIMEOUT_S for slower hosts
    pg_timeout = datetime.timedelta(seconds=int(os.environ.get("GRAPHOP_DIST_TIMEOUT_S", "120")))
    if args.rccl_self:
        if world > 1 or args.emulate_world > 1:
            raise SystemExit("--rccl-self is the one-rank rehearsal of the RCCL path (no torchrun, no --emulate-world)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev, timeout=pg_timeout)
    if world > 1:
        backend = os.environ.get("GRAPHOP_DIST_BACKEND", "nccl")   # "gloo": rehearsal with ranks sharing a GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))
        if dist.get_backend() != backend:
            raise SystemExit("asked for backend %s, got %s" % (backend, dist.get_backend()))

    sharded = world > 1 or args.emulate_world > 1 or args.rccl_self
    n_parts = world if world > 1 else max(1, args.emulate_world)
    name = args.graph
    if name == "auto":
        name = "papers100m" if sharded else "reddit"
    if name == "custom" or args.nodes or args.edges:
        N, E = args.nodes, args.edges
        name = "custom"
    else:
        N, E = graphs.SHAPES[name]
        if name in graphs.SHARDS_OF:          # one shard of the 8-way partition per GPU
            N, E = N // graphs.SHARDS_OF[name], E // graphs.SHARDS_OF[name]
    h = args.heads
    d = args.d or graphs.DEFAULT_D.get(name, 64)
    cut = args.cut if args.cut >= 0 else (0.1 if name == "papers100m" else 1.0)

    # ---- setup (not timed as part of a step; reported) -------------------------------------------
    t_setup = time.perf_counter()
    runner = None
    gdesc = "Chung-Lu(alpha=%.2f)" % args.alpha
    if sharded:
        from custom_op_benchmark_amd import dist as gdist
        kw = dict(chunk_size=args.chunk_size, timing_only=(world == 1 and not args.rccl_self))
        if args.rccl_self:
            kw["force_collectives"] = True
        if name == "rmat25" and args.rccl_self:
            raise SystemExit("--rccl-self builds its self-halo from the Chung-Lu generator's global draws: use --graph papers100m")
        if name == "rmat25":
            runner = gdist.ShardedAttention.synthetic_rmat(25 - 3 + (n_parts - 1).bit_length(), E, n_parts,
                                                           rank, dev, seed=args.seed, **kw)
            gdesc = "R-MAT(0.57,0.19,0.19,0.05) scale %d, equal node ranges" % (25 - 3 + (n_parts - 1).bit_length())
        else:
            if args.rccl_self:
                kw["self_halo"] = True
            runner = gdist.ShardedAttention.synthetic(N, E, n_parts, rank, dev, alpha=args.alpha, seed=args.seed,
                                                      cut=cut, **kw)
            gdesc += ", %.0f%% of a rank's edges drawn from the global node distribution" % (100 * cut)
        g = runner.graph
    elif name == "harness":
        # the reference author's own fixture: 512 disjoint complete digraphs of 30 nodes (wrapper.py:79-112)
        g = graphs.block_diagonal_graph(512, 30, chunk_size=args.chunk_size, device=dev)
        gdesc = "block-diagonal: 512 disjoint complete digraphs of 30 nodes (wrapper.py:79-112)"
    elif name == "rmat25":
        src, dst = graphs.rmat_edges(22, E, seed=args.seed, device=dev)
        g = graphs.graph_from_coo(src, dst, 1 << 22, 1 << 22, args.chunk_size)
        N = 1 << 22
        gdesc = "R-MAT(0.57,0.19,0.19,0.05) scale 22 (one eighth of scale 25)"
        del src, dst
    else:
        g = graphs.chung_lu_graph(N, E, alpha=args.alpha, seed=args.seed, chunk_size=args.chunk_size, device=dev,
                                  labeling=args.labeling)
        if args.labeling != "shuffled":
            gdesc += ", node labeling: %s" % args.labeling
    n_rows, n_cols = g.n_src, g.n_dst
    gen = torch.Generator(device=dev).manual_seed(args.seed + 1 + rank)
    shp = (lambda n: (n, d) if h == 1 else (n, h, d))
    n_own = n_rows if runner is None else runner.n_own

    def values(n):
        if args.values == "normal":
            return torch.randn(shp(n), device=dev, generator=gen) / (d ** 0.5)
        return torch.rand(shp(n), device=dev, generator=gen)

    Q = values(n_own).requires_grad_(True)
    if runner is not None:
        # the shard's K / V rows live in the own-row part of the runner's extended buffers (the halo rows land behind
        # them): no n_own-row copy in front of every exchange
        tail = (d,) if h == 1 else (h, d)
        K = runner.own_rows_view("K", tail).copy_(values(n_own)).requires_grad_(True)
        V = runner.own_rows_view("V", tail).copy_(values(n_own)).requires_grad_(True)
    else:
        K = values(n_own).requires_grad_(True)
        V = values(n_own).requires_grad_(True)
    dO = values(n_own)
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

    # sharded steps: measure the schedules the shard supports (K | V packed or not, column-major backward passes fused or
    # not) on THIS machine and adopt the fastest -- which one wins depends on the links (dist.ShardedAttention.autotune)
    schedule = None
    if runner is not None and not args.no_autotune:
        schedule = runner.autotune(Q, K, V, dO, steps=2)
        log("schedules (ms per step, max over ranks): %s -> kv_packed=%s columns_fused=%s forward_split=%s"
            % (schedule, runner.pack_kv, runner.fuse_columns, runner.use_forward_split))
    for _ in range(args.warmup):
        step()

    # N > 1: the same shard with every exchange replaced by a local copy of the same size (no traffic leaves
    # the GPU), timed in this process before the measured region -- the reference the weak-scaling
    # efficiency of THIS line is taken against
    single_ref_ms = None
    if runner is not None and world > 1:
        runner.emulate = True
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        single_ref_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        runner.emulate = False
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

    # the timed region: exactly K steps between two barrier + synchronize pairs (wall clock, max over
    # ranks); per-step hipEvents on the launch stream give the median / min the roofline is read at
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        timed_step()
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.check_errors()    # (device already synchronised) a walk kernel whose hand-over timed out: no number is reported
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
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
    plan_mb_step = _lib.plan_memory_bytes() / 2**20      # what the 8-function step alone keeps (before the fused op builds its layouts)

    # sharded lines: what the exchanges cost the step = this step - the same step with every exchange a no-op (nothing
    # is packed, sent, awaited or added; values involving halo rows are garbage: timing only), measured the same way
    # (barrier + synchronize on both sides, max over ranks) right behind the timed region
    exposed_ms = noop_ms = None
    if runner is not None:
        runner.noop_exchange = True
        for _ in range(2):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        noop_el = time.perf_counter() - t0
        runner.noop_exchange = False
        if world > 1:
            tt = torch.tensor([noop_el], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            noop_el = float(tt.item())
        noop_ms = 1e3 * noop_el / args.steps
        exposed_ms = ms_per_step - noop_ms
        step()      # (halo rows valid again for what follows)

    # ---- the same step through the fused op (extra op: one autograd node, no E-sized intermediates) ----
    fused = None
    if runner is None and world == 1 and not args.no_fused and not args.hip_graph:
        def fstep():
            Q.grad = K.grad = V.grad = None
            functions.fused_attention_step(g, Q, K, V, dO)
        for _ in range(max(2, args.warmup)):
            fstep()
        fev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fev[0].record()
        for i in range(args.steps):
            fstep()
            fev[i + 1].record()
        torch.cuda.synchronize()
        f_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        f_step = [fev[i].elapsed_time(fev[i + 1]) for i in range(args.steps)]
        _lib.profile_enable(True)
        for _ in range(max(1, args.profile_steps)):
            fstep()
        torch.cuda.synchronize()
        fprof = _lib.profile_read()
        _lib.profile_enable(False)
        fused = {"ms_per_step": round(f_ms, 4), "value": g.n_edges / (f_ms * 1e-3), "unit": "edges/s",
                 "step_ms_median": round(statistics.median(f_step), 4), "step_ms_min": round(min(f_step), 4),
                 "op": "FusedAttention = attention_forward + attention_backward (extra op; same o, dQ, dK, dV as "
                       "the 8-function step, s / a / da / ds never materialised in the backward)",
                 "passes_ms": {k: round(v["mean_ms"] * v["calls"] / max(1, args.profile_steps), 4) for k, v in fprof.items()},
                 "kernels": {k: v["kernel"] for k, v in fprof.items()}}

    # ---- per-kernel durations, live, hipEvents on the launch stream --------------------------------
    nprof = max(1, args.profile_steps)
    _lib.profile_enable(True)
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)

    pb = pass_bytes(g.n_edges, n_rows, n_cols, h, d, g.n_row_chunks, g.n_col_chunks)
    alg_step = sum(pb.values())
    # hardware-counter traffic per kernel family (profiles/pmc_traffic.json), quoted only for the kernel sources it
    # was collected on: next to every ALGORITHMIC fraction (API dtypes: int64 ids, every operand once) the line
    # carries what the kernel really moved beyond L2 -- the plans read 4-byte id mirrors and skip an identity
    # eid altogether, so an algorithmic fraction can exceed what the memory system delivered (softmax: 1.89 GB
    # algorithmic against 1.00 GB moved)
    sha = kernels_sha()
    moved_by_kernel, moved_by_tag, moved_note = {}, {}, "no profiles/pmc_traffic.json for this workload"
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") != "%s_h%d_d%d" % (name, h, d):
                moved_note = "profiles/pmc_traffic.json is for workload %s" % tj.get("workload")
            elif tj.get("kernels_sha") != sha:
                moved_note = ("profiles/pmc_traffic.json was collected on other kernel sources (sha %s, now %s): not quoted"
                              % (tj.get("kernels_sha"), sha))
            else:
                moved_by_kernel = {k: v["hbm_bytes_per_launch"] for k, v in tj.get("kernels", {}).items()}
                # per pass tag where the counter passes could tell the launches of one instantiation apart (ADVICE r4:
                # the row- and column-major launches of k_spmm_walk_f32 move different bytes)
                moved_by_tag = {k: v["hbm_bytes_per_launch"] for k, v in tj.get("passes", {}).items()}
                moved_note = ("NOT measured in this run: rocprofv3 --pmc passes of this command on the same kernel sources "
                              "(kernels_sha %s), profiles/pmc_traffic.json; per pass tag (launch order inside a step) where the "
                              "file has it, else not quoted per pass; %s" % (sha, tj.get("note", "")))
        except (ValueError, KeyError):
            moved_note = "profiles/pmc_traffic.json unreadable"
    passes = {}
    kern = {}
    if "sddmm_fwd_part" in prof and "sddmm_fwd" not in prof:
        # sharded step: the SDDMM forward runs as an own-column and a halo-column launch over ONE score array (dist.py)
        pp = prof.pop("sddmm_fwd_part")
        prof["sddmm_fwd"] = dict(pp, mean_ms=pp["total_ms"] / nprof, launches_per_step=pp["calls"] // nprof)
    # bytes of the survey's figure that the plan provably never moves: with eid == arange (row-major plans) the softmax
    # kernels read NO edge ids at all (the int64 eid stream of E x 8 B is elided, not mirrored).  A per-pass fraction is
    # quoted on the bytes that remain, so it cannot exceed what the memory system delivered; alg_GB stays the survey's.
    try:
        eid_identity = bool(_lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r).info.eid_identity)
    except Exception:
        eid_identity = False
    for tag in PASS_TAGS:
        if tag not in prof:
            continue
        ms = prof[tag]["mean_ms"]
        kname = prof[tag]["kernel"] or "?"
        moved = moved_by_tag.get(tag)
        elided = 8 * g.n_edges if (tag.startswith("softmax") and eid_identity) else 0
        frac_bytes = pb[tag] - elided
        passes[tag] = {"ms": round(ms, 4), "kernel": kname, "alg_GB": round(pb[tag] / 1e9, 4),
                       "alg_GBps": round(pb[tag] / 1e6 / ms, 1),
                       "elided_GB": round(elided / 1e9, 4),
                       "frac": round(frac_bytes / 1e6 / ms / HBM_PEAK_GBS, 4),
                       "moved_GB": round(moved / 1e9, 4) if moved else None,
                       "moved_frac": round(moved / 1e6 / ms / HBM_PEAK_GBS, 4) if moved else None}
        if "launches_per_step" in prof[tag]:
            passes[tag]["launches_per_step"] = prof[tag]["launches_per_step"]
        fam = "k_spmm_*" if kname.startswith("k_spmm") else ("k_sddmm_*" if kname.startswith("k_sddmm") else kname)
        k = kern.setdefault(fam, {"ms": 0.0, "bytes": 0.0, "launches": 0, "names": set()})
        k["ms"] += ms; k["bytes"] += pb[tag]; k["launches"] += 1; k["names"].add(kname)
    if "spmm_pair_cols" in prof:
        # sharded step with the column-major backward passes fused (graphop_spmm_pair): dV and dK in one launch
        pp = prof.pop("spmm_pair_cols")
        ms = pp["mean_ms"]
        b = pb["spmm_bwd_dx"] + pb["sddmm_bwd_dB"] - g.n_edges * 16 - 16 * g.n_col_chunks    # ids, edge ids and chunk metadata read once
        passes["spmm_bwd_dx+sddmm_bwd_dB"] = {"ms": round(ms, 4), "kernel": pp["kernel"], "alg_GB": round(b / 1e9, 4),
                                              "alg_GBps": round(b / 1e6 / ms, 1), "elided_GB": 0.0,
                                              "frac": round(b / 1e6 / ms / HBM_PEAK_GBS, 4), "moved_GB": None, "moved_frac": None}
        k = kern.setdefault("k_spmm_*", {"ms": 0.0, "bytes": 0.0, "launches": 0, "names": set()})
        k["ms"] += ms; k["bytes"] += b; k["launches"] += 1; k["names"].add(pp["kernel"])
    other = {t: {"ms_per_step": round(v["total_ms"] / nprof, 4), "launches_per_step": v["calls"] // nprof,
                 "kernel": v["kernel"]} for t, v in prof.items() if t not in PASS_TAGS}
    dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"]) if kern else ("none", {"ms": 1.0, "bytes": 0.0, "launches": 1, "names": set()})
    dom_kernel = "/".join(sorted(dom["names"])) or dom_name
    achieved = dom["bytes"] / 1e6 / dom["ms"]          # GB/s = algorithmic bytes per launch / mean launch time
    traffic, traffic_src = moved_by_kernel.get(dom_kernel), moved_note
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                # informative: against the 6.29 TB/s a device-wide copy reaches on this part
                "frac_of_copy_rate": round(achieved / HBM_COPY_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": dom_kernel, "launches_per_step": dom["launches"],
                "alg_bytes_per_launch": int(dom["bytes"] / max(1, dom["launches"])),
                "avg_launch_ms": round(dom["ms"] / max(1, dom["launches"]), 4),
                "step": {"alg_bytes": int(alg_step), "alg_GBps": round(alg_step / 1e6 / ms_per_step, 1),
                         "frac": round(alg_step / 1e6 / ms_per_step / HBM_PEAK_GBS, 4),
                         "frac_of_copy_rate": round(alg_step / 1e6 / ms_per_step / HBM_COPY_GBS, 4),
                         "frac_at_median_step": round(alg_step / 1e6 / statistics.median(step_ms) / HBM_PEAK_GBS, 4),
                         "kernel_ms_sum": round(sum(p["ms"] for p in passes.values()) +
                                                sum(o["ms_per_step"] for o in other.values()), 3)},
                "passes": passes, "passes_note": "frac = (algorithmic bytes (API dtypes) - elided_GB) / time / 8 TB/s, elided_GB = the "
                                                 "int64 eid stream a softmax pass over an identity-eid plan never reads; moved_GB / "
                                                 "moved_frac = fabric-side bytes of that pass from the hardware counters: " + moved_note,
                "other_launches": other}
    # tables far beyond the Infinity Cache with rows too short for column windows (products-shape): every
    # gathered edge is an HBM random-row read; what is physically reachable is that rate, not 8 TB/s of
    # algorithmic bytes -- a labelled secondary figure
    table_bytes = max(n_rows, n_cols) * h * d * 4
    # the secondary gather figures below price one neighbour row per edge and pass: they describe the GATHER drivers
    # (chunk / window / walk kernels).  The block-dense MFMA drivers read every node row of a block ONCE per pass, so a
    # "ceiling" built from per-edge gathers is not a bound for them (round 4 printed frac 1.78 there): not emitted.
    gather_drivers = bool(passes) and not any("block" in (passes[t]["kernel"] or "") for t in GATHER_TAGS if t in passes)
    # hard bound that needs no microbenchmark: every gathered edge moves its F x 4-byte row through a CU's vector L1,
    # 64 B per clock and CU (MI355X_MICROARCH.md, memory hierarchy), whatever the L2 does; + the softmax passes at 8 TB/s
    if gather_drivers:
        props = torch.cuda.get_device_properties(dev)
        clk_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
        cus = int(props.multi_processor_count)
        l1_rate = 64.0 * cus * clk_hz                       # B/s: 39.3 TB/s at 256 CUs x 2.4 GHz
        soft_b = sum(pb[t] for t in PASS_TAGS if t.startswith("softmax")) - (16 * g.n_edges if eid_identity else 0)
        l1_ms = 6.0 * g.n_edges * h * d * 4 / l1_rate * 1e3 + soft_b / 1e6 / HBM_PEAK_GBS
        roofline["l1_bound_ms"] = round(l1_ms, 3)
        roofline["l1_bound"] = {"what": "hard lower bound of the 8-function step on this device: 6 gather passes x E x F x 4 B through "
                                        "the CUs' vector L1s at 64 B/clk/CU + the two softmax passes at 8 TB/s",
                                "cus": cus, "clock_MHz": round(clk_hz / 1e6, 1), "l1_rate_TBps": round(l1_rate / 1e12, 2),
                                "frac_of_bound": round(l1_ms / ms_per_step, 4)}
    if table_bytes > (512 << 20) and gather_drivers:
        gbytes = 6.0 * g.n_edges * h * d * 4
        gms = sum(pp_["ms"] for t, pp_ in passes.items() if not t.startswith("softmax"))
        roofline["gather_roofline"] = {
            "what": "secondary: 6 gather passes x E x F x 4 B of neighbour rows at the measured HBM random-row rate "
                    "(MI355X_MICROARCH.md, Indexed rows, HBM: 6.0-6.1 TB/s); the tables (%.1f GB each) are beyond the Infinity Cache"
                    % (table_bytes / 1e9),
            "gather_bytes": int(gbytes), "peak_GBps": HBM_RANDOM_ROW_GBS, "achieved_GBps": round(gbytes / 1e6 / gms, 1),
            "frac": round(gbytes / 1e6 / gms / HBM_RANDOM_ROW_GBS, 4),
            "floor_ms_per_step": round(gbytes / 1e6 / HBM_RANDOM_ROW_GBS, 2)}
    # tables that fit the Infinity Cache (Reddit-shape: 59.6 MB): the gathers can be L2 hits when the passes
    # are organised in column windows; the reachable ceiling of the 8-function step is then six gather
    # passes at the L2-resident gather rate plus the two softmax passes at the HBM roofline
    if table_bytes <= (512 << 20) and gather_drivers:
        gbytes = 6.0 * g.n_edges * h * d * 4
        soft_bytes = sum(pb[t] for t in PASS_TAGS if t.startswith("softmax"))
        ceil_ms = gbytes / 1e6 / L2_GATHER_GBS + soft_bytes / 1e6 / HBM_PEAK_GBS
        gms = sum(pp_["ms"] for t, pp_ in passes.items() if not t.startswith("softmax"))
        roofline["l2_gather_ceiling"] = {
            "what": "secondary: the reachable ceiling of this operator surface on a cache-resident table: 6 gather passes x "
                    "E x F x 4 B of neighbour rows at the measured L2-resident lane-group gather rate (30 TB/s: "
                    "tools/microbench/l2_gather.hip, profiles/r1_l2_resident_sweep.txt, r2_cu_walk_microbench.txt) + the "
                    "two softmax passes at 8 TB/s; the algorithmic-byte roofline above prices a gathered row as read once "
                    "per pass, which no kernel of this 8-function API can do at mean degree %d" % (g.n_edges // max(1, n_rows)),
            "gather_bytes": int(gbytes), "gather_rate_GBps": L2_GATHER_GBS,
            "ceiling_ms_per_step": round(ceil_ms, 3),
            "ceiling_frac_of_hbm_roofline": round(alg_step / 1e6 / ceil_ms / HBM_PEAK_GBS, 4),
            "achieved_gather_GBps": round(gbytes / 1e6 / gms, 1) if gms > 0 else None,
            "frac_of_ceiling": round(ceil_ms / ms_per_step, 4)}
        roofline["ceiling_frac"] = round(ceil_ms / ms_per_step, 4)   # = l2_gather_ceiling.ceiling_ms_per_step / ms_per_step
        roofline["gather_roofline"] = {
            "what": "secondary: the six gather passes' neighbour-row bytes over their measured time against the L2-resident "
                    "gather rate", "gather_bytes": int(gbytes), "peak_GBps": L2_GATHER_GBS,
            "achieved_GBps": round(gbytes / 1e6 / gms, 1) if gms > 0 else None,
            "frac": round(gbytes / 1e6 / gms / L2_GATHER_GBS, 4) if gms > 0 else None,
            "floor_ms_per_step": round(gbytes / 1e6 / L2_GATHER_GBS, 2)}
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

    cfg = {"workload": "%s-shape %s graph, N=%d E=%d per GPU, h=%d d=%d, %s values, chunk_size=%d, int64 CSR both "
                       "orientations" % (name, gdesc, n_own, g.n_edges, h, d,
                                         "U[0,1)" if args.values != "normal" else "N(0,1)/sqrt(d)", args.chunk_size),
           "graph": name, "nodes": n_rows, "edges": total_edges, "heads": h, "d": d,
           "chunk_size": args.chunk_size, "row_chunks": g.n_row_chunks, "col_chunks": g.n_col_chunks,
           "parallelism": ("single GPU" if not sharded else "node-range shards x%d, RCCL all-to-all halo" % n_parts)
                          + (" [one-rank nccl process group, self-halo: the RCCL code path on one GPU]" if args.rccl_self else "")
                          + (" [TIMING-ONLY rehearsal of shard 0 of %d on one GPU, exchanges = local copies]" % args.emulate_world
                             if args.emulate_world > 1 and world == 1 else "")}
    if runner is not None:
        cfg["weak_scaling_note"] = ("N > 1 runs BASELINE.json config %s (one 1/8 shard of the %s shape per GPU, fixed work per GPU); "
                                    "the N = 1 default is the Reddit headline (config 2), a different workload: compare "
                                    "N > 1 values among themselves and with `bench.py --gpus 1 --graph %s` (the same shard "
                                    "without halo)" % ("4" if name == "papers100m" else "5" if name == "rmat25" else "?", name, name))
        cfg["halo"] = runner.halo_stats(h * d * 4)
        cfg["schedule"] = {"kv_packed": bool(cfg["halo"]["kv_packed"]), "columns_fused": bool(runner.fuse_columns),
                           "forward_split": bool(cfg["halo"]["forward_split"]),
                           "measured_ms_per_step": schedule,
                           "how": ("dist.ShardedAttention.autotune: every schedule timed for 2 steps after a warm-up, max over "
                                   "ranks, fastest adopted by all ranks" if schedule else "default (--no-autotune)")}
        # per-exchange wall times, measured in a separate pass (their syncs defeat the overlap)
        runner.timers = {}
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        cfg["halo"]["exchange_ms"] = {k: round(1e3 * statistics.median(v), 3) for k, v in runner.timers.items()}
        # bytes one rank receives + sends in that exchange over its wall time (halo_KV carries both tables)
        per_x = cfg["halo"]["bytes_per_exchange_in"] + cfg["halo"]["bytes_per_exchange_out"]
        cfg["halo"]["exchange_GBps_in_plus_out"] = {k: round((2 if k == "halo_KV" else 1) * per_x / 1e6 / max(1e-6, v), 1)
                                                    for k, v in cfg["halo"]["exchange_ms"].items()}
        # --emulate-world: the "exchanges" are local copies of the right sizes on this GPU, NOT link times
        cfg["halo"]["exchange_ms_is_local_copy"] = bool(runner.emulate)
        if args.rccl_self:
            cfg["halo"]["exchange_ms_note"] = ("one-rank RCCL process group: all_to_all_single from this GPU to itself (the real "
                                               "collective code path, no xGMI link involved)")
        runner.timers = None
    metric = "edges/sec fwd+bwd (SDDMM+softmax+SpMM) on Reddit d=64; HBM GB/s vs roofline"
    if not (name == "reddit" and h == 1 and d == 64 and not sharded):
        # the BASELINE.json metric is quoted on Reddit d=64; any other workload says what it ran
        metric = ("edges/sec fwd+bwd (SDDMM+softmax+SpMM) on %s%s h=%d d=%d; HBM GB/s vs roofline"
                  % (name, (" (one 1/%d node-range shard per GPU, BASELINE.json config %s)"
                            % (graphs.SHARDS_OF.get(name, n_parts), "4" if name == "papers100m" else "5" if name == "rmat25" else "-"))
                     if sharded else "-shape", h, d))
    if world > 1 or args.rccl_self:
        cfg["world_size"] = dist.get_world_size()
        cfg["backend"] = dist.get_backend()
    out = {
        "metric": metric,
        "value": value, "unit": "edges/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "step_ms": {"median": round(statistics.median(step_ms), 4), "min": round(min(step_ms), 4),
                    "max": round(max(step_ms), 4), "method": "hipEvents around every timed step on the launch stream"},
        "config": cfg,
        "setup": {"graph_build_s": round(t_graph, 2), "first_step_with_plans_s": round(t_first, 3),
                  "plan_memory_MB": round(_lib.plan_memory_bytes() / 2**20, 1),
                  "plan_memory_MB_8_function_step": round(plan_mb_step, 1),
                  "plan_memory_note": "device bytes of both orientations' plans: every layout of the 8-function step AND (plan_memory_MB) "
                                      "of the fused op measured behind it; a model uses one of the two forms"},
        "launch": "hip graph replay" if args.hip_graph else "eager API calls",
        "roofline": roofline,
    }
    if exposed_ms is not None:
        kern_ms = sum(other[t]["ms_per_step"] for t in ("halo_pack", "halo_unpack_add") if t in other)
        out["exposed_exchange_breakdown"] = {"pack_and_add_home_kernels_ms": round(kern_ms, 4),
                                             "not_hidden_by_the_overlap_ms": round(exposed_ms - kern_ms, 4),
                                             "note": "the pack and add-home kernels run on the compute stream (they cannot overlap "
                                                     "with the passes); the rest is what the collectives cost beyond their cover"}
        out["exposed_exchange_ms"] = round(exposed_ms, 4)
        out["step_without_exchanges_ms"] = round(noop_ms, 4)
        out["exposed_exchange_note"] = ("ms_per_step minus the same step with every halo exchange (pack, collective, wait, add-home) "
                                        "a no-op, timed the same way right behind the timed region; all four exchanges run "
                                        "under compute (dist.py), so this is what the overlap does NOT hide"
                                        + ("; NB exchanges here are local copies on this GPU, not link transfers"
                                           if (runner is not None and runner.emulate) else ""))
    if single_ref_ms is not None:
        out["single_gpu_reference_ms"] = round(single_ref_ms, 4)
        out["single_gpu_reference"] = ("rank %d's shard in this process with every halo exchange replaced by a local copy of "
                                       "the same size (no traffic leaves the GPU), %d steps before the timed region" % (rank, args.steps))
        out["weak_scaling_efficiency"] = round(single_ref_ms / ms_per_step, 4)
    if fused is not None:
        out["fused"] = fused
    if rank == 0 and world == 1 and runner is None and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, Q.detach(), K.detach(), V.detach(), dO, args.cpu_sample_edges, log,
                                           incidence=(g.n_edges <= 200_000))
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or args.rccl_self:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
