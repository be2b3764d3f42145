"""GPU tier: the node-range sharded step with the HIP operators, all shards on ONE GPU.

SURVEY.md 8e's validation mode: the W shards of a graph live in one process (dist.LocalGroup, one
thread per shard), every halo exchange is a true device-to-device copy between the shards' tensors
(HIP pack kernel -> all-to-all by copies -> HIP scatter-add kernel), and the re-assembled
o, dQ, dK, dV, s, a must equal the single-process oracle step on the global graph.  Exercises the
non-square n_own x (n_own + n_halo) local graphs on the HIP path, incl. the window drivers."""
import pytest
import torch

import oracle
from custom_op_benchmark_amd import _lib
from custom_op_benchmark_amd.dist import ShardedAttention, run_local_shards

from util import oracle_step, rand_inputs, random_graph

pytestmark = pytest.mark.gpu


def _run(dev, g, inp, world, chunk_size=8):
    def shard(rank, handle):
        sh = ShardedAttention.from_global_coo(g.src.to(dev), g.dst.to(dev), g.n_src, rank, world, dev,
                                              chunk_size=chunk_size, group=handle)
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        r = sh.step(*(inp[k][lo:hi].to(dev).contiguous() for k in ("Q", "K", "V", "dO")))
        torch.cuda.synchronize()
        ext_ids = torch.cat([torch.arange(lo, hi, device=dev), sh.halo_ids])
        key = (sh.graph.src + lo) * g.n_dst + ext_ids[sh.graph.dst]
        return dict(lo=lo, hi=hi, key=key.cpu(), n_halo=sh.n_halo, recv=sh.recv_counts,
                    **{k: v.detach().cpu() for k, v in r.items()})
    return run_local_shards(world, shard)


def _check(g, want, parts):
    got = {k: torch.zeros_like(want[k]) for k in ("o", "dQ", "dK", "dV", "s", "a")}
    for z in parts:
        lo, hi = z["lo"], z["hi"]
        for k in ("o", "dQ", "dK", "dV"):
            got[k][lo:hi] = z[k]
        m = (g.src >= lo) & (g.src < hi)
        order = torch.argsort(z["key"], stable=True)       # local slot order -> (src, global dst) order
        got["s"][m] = z["s"][order]
        got["a"][m] = z["a"][order]
    for k in got:
        torch.testing.assert_close(got[k], want[k], rtol=1e-4, atol=1e-5, msg=lambda msg: k + ": " + msg)


@pytest.mark.parametrize("world,h,d", [(2, 1, 64), (3, 2, 16), (4, 1, 16)])
def test_sharded_hip_step_matches_oracle(dev, world, h, d):
    g = random_graph(400, 400, 12000, seed=21, chunk_size=8, zero_rows=0.1, hub=700)
    inp = rand_inputs(g, h, d, seed=22, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    parts = _run(dev, g, inp, world)
    assert all(z["n_halo"] > 0 and z["recv"][r] == 0 for r, z in enumerate(parts))
    _check(g, want, parts)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_step_window_drivers(dev, world):
    """Same, with the column-window drivers forced on the shards' non-square local graphs."""
    _lib.tune("sweep_min_kb", 0); _lib.tune("window_kb", 4); _lib.tune("vrow_t", 64)
    _lib.tune("sweep_min_granule", 0); _lib.clear_plan_cache()
    try:
        g = random_graph(1500, 1500, 30000, seed=5, chunk_size=32, zero_rows=0.1, hub=900)
        inp = rand_inputs(g, 1, 64, seed=6, normal=True)
        want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
        _lib.profile_enable(True)
        parts = _run(dev, g, inp, world, chunk_size=32)
        tags = set(_lib.profile_read())
        _lib.profile_enable(False)
        assert {"halo_pack", "halo_unpack_add"} <= tags, tags
        _check(g, want, parts)
    finally:
        _lib.profile_enable(False)
        _lib.tune_reset(); _lib.clear_plan_cache()


def test_partial_sddmm_entry_composes_one_score_array(dev):
    """graphop_maskedmm_csr_forward_partial (ABI 7): two chunk lists over disjoint slot sets write ONE score array, no zero
    fill (entries no list names keep what they held), each named entry written once -- vs the oracle's full SDDMM."""
    import ctypes
    for h, d in ((1, 64), (2, 16), (1, 20)):
        g = random_graph(300, 400, 9000, seed=3 + h, chunk_size=8, zero_rows=0.1, hub=500)
        inp = rand_inputs(g, h, d, seed=4, normal=True)
        want = oracle.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, inp["Q"], inp["K"])
        gd = g.to(dev)
        Q, K = inp["Q"].to(dev), inp["K"].to(dev)
        from custom_op_benchmark_amd.part_csr import partition_csr
        pick = torch.rand(g.n_edges, generator=torch.Generator().manual_seed(9)) < 0.6
        s = torch.full(want.shape, float("nan"), device=dev)
        untouched = torch.zeros(g.n_edges, dtype=torch.bool)
        for k, mask in enumerate((pick, ~pick)):
            if k == 1:
                mask = mask & (torch.arange(g.n_edges) % 7 != 0)      # leave some entries to nobody
                untouched = ~(pick | mask)
            m = mask.to(dev)
            cum = torch.zeros(g.n_edges + 1, dtype=torch.int64, device=dev)
            torch.cumsum(m, 0, out=cum[1:])
            ip = cum[gd.indptr_r].contiguous()
            slots = torch.nonzero(m).flatten()
            row, ptr_ = partition_csr(ip, 8)
            idx = gd.indices_r[slots].contiguous()
            _lib.check(_lib.lib().graphop_maskedmm_csr_forward_partial(
                _lib.dtype_code(Q), _lib.ptr(row), _lib.ptr(ptr_), _lib.ptr(slots), _lib.ptr(idx), _lib.ptr(Q), _lib.ptr(K),
                _lib.ptr(s), row.size(0), slots.size(0), g.n_edges, Q.size(0), K.size(0), h, d, _lib.stream_of(Q)))
        torch.cuda.synchronize()
        got = s.cpu()
        assert bool(torch.isnan(got[untouched]).all()) and untouched.any()
        torch.testing.assert_close(got[~untouched], want[~untouched], rtol=1e-4, atol=1e-5)


def test_pack_and_scatter_add_kernels(dev):
    gen = torch.Generator(device=dev).manual_seed(0)
    for shape, dt in (((50, 64), torch.float32), ((50, 3, 5), torch.float32), ((40, 8, 16), torch.float64)):
        X = torch.rand(shape, device=dev, generator=gen, dtype=dt)
        idx = torch.randint(0, shape[0], (137,), device=dev, generator=gen)
        got = _lib.gather_rows(X, idx)
        assert torch.equal(got, X[idx])
        acc = torch.rand(shape, device=dev, generator=gen, dtype=dt)
        want = acc.clone().index_add_(0, idx, got)
        acc2 = acc.clone()
        _lib.scatter_add_rows(acc, idx, got)
        torch.testing.assert_close(acc, want, rtol=1e-5 if dt == torch.float32 else 1e-12, atol=1e-6)
        _lib.add_rows_grouped(acc2, _lib.group_rows(idx), got)      # one launch, rows grouped by destination
        torch.testing.assert_close(acc2, want, rtol=1e-5 if dt == torch.float32 else 1e-12, atol=1e-6)
    assert _lib.gather_rows(X, idx[:0]).shape[0] == 0


def test_two_shards_at_scale_vs_cpu_path(dev):
    """World 2 on one GPU (LocalGroup: real device-to-device halo exchanges) at 2.5e7 edges and the
    DEFAULT kernel geometry -- the shards' local graphs are large enough for the window and walk
    drivers -- against the stock-PyTorch CPU path on the global graph."""
    from custom_op_benchmark_amd import graphs
    from oracle import torch_path
    _lib.tune_reset(); _lib.clear_plan_cache()
    N, E, d = 60000, 25_000_000, 64
    g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=11, device=dev)
    gen = torch.Generator(device=dev).manual_seed(12)
    Q, K, V, dO = (torch.randn(N, d, device=dev, generator=gen) / 8 for _ in range(4))
    inp = dict(Q=Q, K=K, V=V, dO=dO)

    def shard(rank, handle):
        sh = ShardedAttention.from_global_coo(g.src, g.dst, N, rank, 2, dev, chunk_size=32, group=handle)
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        r = sh.step(*(inp[k][lo:hi].contiguous() for k in ("Q", "K", "V", "dO")))
        torch.cuda.synchronize()
        return dict(lo=lo, hi=hi, n_halo=sh.n_halo, **{k: r[k].detach() for k in ("o", "dQ", "dK", "dV")})
    _lib.profile_enable(True)
    parts = run_local_shards(2, shard)
    kernels = {r.get("kernel") for r in _lib.profile_read().values()}
    _lib.profile_enable(False)
    assert "k_spmm_walk_f32" in kernels or "k_spmm_wown_staged_f32" in kernels, kernels
    assert all(z["n_halo"] > 0 for z in parts)
    o0, dQ0, dK0, dV0 = torch_path.attention_step_blocked(g.src.cpu(), g.dst.cpu(), g.indptr_r.cpu(), Q.cpu(), K.cpu(),
                                                          V.cpu(), dO.cpu(), N, rows_per_block=2048)
    for name, want in (("o", o0), ("dQ", dQ0), ("dK", dK0), ("dV", dV0)):
        got = torch.cat([z[name] for z in parts]).cpu()
        torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5, msg=lambda m: name + ": " + m)


_RCCL_CHILD = r'''
import os, sys
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, os.path.join(sys.argv[2], "tests"))
import torch, torch.distributed as dist
import oracle
from custom_op_benchmark_amd.dist import ShardedAttention
from util import oracle_step, rand_inputs, random_graph
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
assert dist.get_backend() == "nccl"
calls = {"n": 0, "async": 0, "groups": 0, "p2p": 0}
real, real_b = dist.all_to_all_single, dist.batch_isend_irecv
def counted(*a, **k):
    calls["n"] += 1; calls["async"] += 1 if k.get("async_op") else 0
    return real(*a, **k)
def counted_b(ops):
    calls["groups"] += 1; calls["p2p"] += len(ops)
    return real_b(ops)
dist.all_to_all_single = counted
dist.batch_isend_irecv = counted_b
for h, d, pack in ((1, 64, False), (2, 16, True)):
    g = random_graph(600, 600, 20000, seed=31 + h, chunk_size=8, zero_rows=0.1, hub=700)
    inp = rand_inputs(g, h, d, seed=32, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    mask = torch.rand(g.src.numel(), generator=torch.Generator().manual_seed(5)) < 0.5   # half of the edges: fetched through RCCL
    n0 = calls["n"]
    sh = ShardedAttention.from_global_coo(g.src.to(dev), g.dst.to(dev), g.n_src, 0, 1, dev, chunk_size=8,
                                          force_collectives=True, halo_mask=mask, pack_kv=pack)
    assert sh.n_halo > 0 and sh.recv_counts == [sh.n_halo] and sh.send_counts == [sh.n_halo]
    assert sh.fwd_halves is not None                 # SDDMM forward = own-column half under the K exchange + halo-column half
    assert calls["n"] - n0 == 2                      # setup: counts, then the id lists -- through the process group
    n0, a0, g0, p0 = calls["n"], calls["async"], calls["groups"], calls["p2p"]
    r = sh.step(*(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    torch.cuda.synchronize()
    if pack:    # K | V as ONE grouped RCCL exchange (send + recv to self for each table), then dV, dK
        assert calls["n"] - n0 == 2 and calls["async"] - a0 == 2 and calls["groups"] - g0 == 1 and calls["p2p"] - p0 == 4, calls
        assert sh.collectives_last_step == 3
    else:       # K, V, dV, dK: asynchronous RCCL all-to-alls
        assert calls["n"] - n0 == 4 and calls["async"] - a0 == 4 and calls["groups"] == g0, calls
        assert sh.collectives_last_step == 4
    ext_ids = torch.cat([torch.arange(0, g.n_src, device=dev), sh.halo_ids])
    key = (sh.graph.src * g.n_dst + ext_ids[sh.graph.dst]).cpu()
    order = torch.argsort(key, stable=True)
    for k in ("o", "dQ", "dK", "dV"):
        torch.testing.assert_close(r[k].detach().cpu(), want[k], rtol=1e-4, atol=1e-5, msg=lambda m: k + ": " + m)
    for k in ("s", "a"):
        torch.testing.assert_close(r[k].detach().cpu()[order], want[k], rtol=1e-4, atol=1e-5, msg=lambda m: k + ": " + m)
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
'''


def test_rccl_path_at_world_size_one(dev):
    """The REAL collective path on a one-GPU box: a one-rank `nccl` (= RCCL) process group in a spawned child,
    ShardedAttention with force_collectives (no world == 1 short-cut) and a self-halo -- half of the edges fetch
    their K / V rows through all_to_all_single(async_op=True) + work.wait() with split-size views, and send their
    dK / dV rows home the same way -- result == the single-process oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(29500 + os.getpid() % 2000)
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD, port, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_line_through_the_rccl_path(dev):
    """`bench.py --rccl-self` end to end on a small graph: one-rank nccl process group, self-halo, the JSON line says so."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--graph", "custom", "--nodes", "30000",
                        "--edges", "600000", "--d", "64", "--cut", "0.2", "--rccl-self", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--profile-steps", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["backend"] == "nccl" and cfg["world_size"] == 1 and cfg["halo"]["n_halo"] > 0
    fwd = {"halo_KV"} if cfg["halo"]["kv_packed"] else {"halo_K", "halo_V"}
    assert cfg["halo"]["exchange_ms_is_local_copy"] is False and set(cfg["halo"]["exchange_ms"]) == fwd | {"grad_dV", "grad_dK"}
    assert isinstance(cfg["halo"]["forward_split"], bool) and "exposed_exchange_ms" in line and line["exposed_exchange_ms"] is not None
    assert len(cfg["schedule"]["measured_ms_per_step"]) == 8
    assert line["value"] > 0 and line["n_gpus"] == 1


def test_spmm_pair_entry_vs_oracle(dev):
    """graphop_spmm_pair (ABI 7): two SpMM-type passes over one column-major chunked CSR in one launch, weights as (E, 2)
    pairs -- against two oracle SpMMs; outputs handed over full of NaNs (the launch defines every row); short columns,
    a hub column cut between lane groups, a third of the columns without edges, an empty tail."""
    import oracle as orc
    _lib.tune_reset(); _lib.clear_plan_cache()
    for d, cs, cpg in ((64, 1, 3), (128, 32, 32), (256, 2, 70), (128, 1, 5)):
        _lib.tune("spmm_flat_cpg", cpg)
        g = random_graph(900, 1400, 2600 + cs, seed=d + cs, chunk_size=cs, zero_rows=0.0, hub=700)
        gen = torch.Generator().manual_seed(d)
        X0, X1 = torch.randn(g.n_src, d, generator=gen), torch.randn(g.n_src, d, generator=gen)
        w0, w1 = torch.rand(g.n_edges, generator=gen), torch.randn(g.n_edges, generator=gen)
        n_out = g.n_dst + 37                                            # rows behind the last column: zero
        pad = lambda X: torch.cat([X, torch.zeros(n_out - X.size(0), d)]) if X.size(0) < n_out else X
        want0 = orc.vector_spmm_forward(g.col, g.ptr_c, g.eid_c, g.indices_c, w0, pad(X0))[:n_out]
        want1 = orc.vector_spmm_forward(g.col, g.ptr_c, g.eid_c, g.indices_c, w1, pad(X1))[:n_out]
        gd = g.to(dev)
        w0d, w1d = w0.to(dev), w1.to(dev)
        w2 = torch.full((g.n_edges, 2), float("nan"), device=dev)
        _lib.check(_lib.lib().graphop_interleave_pairs(_lib.F32, _lib.ptr(w0d), _lib.ptr(w1d), _lib.ptr(w2), g.n_edges,
                                                       _lib.stream_of(w2)))
        assert torch.equal(w2, torch.stack((w0d, w1d), dim=1))          # (n not a multiple of 4: the tail lanes)
        X0d, X1d = X0.to(dev), X1.to(dev)
        out0 = torch.full((n_out, d), float("nan"), device=dev)
        out1 = torch.full((n_out, d), float("nan"), device=dev)
        with _lib.device_guard(dev):
            pc = _lib.get_plan(gd.col, gd.ptr_c, gd.eid_c, gd.indices_c, g.n_src)
            assert _lib.lib().graphop_spmm_pair_supported(_lib.F32, gd.col.size(0), g.n_edges, g.n_src, 1, d, pc.handle)
            _lib.check(_lib.lib().graphop_spmm_pair(_lib.F32, _lib.ptr(gd.col), _lib.ptr(gd.ptr_c), _lib.ptr(gd.eid_c),
                                                    _lib.ptr(gd.indices_c), _lib.ptr(w2), _lib.ptr(X0d), _lib.ptr(X1d),
                                                    _lib.ptr(out0), _lib.ptr(out1), gd.col.size(0), g.n_edges, g.n_src, n_out,
                                                    1, d, pc.handle, _lib.stream_of(out0)))
        torch.cuda.synchronize()
        assert not torch.isnan(out0).any() and not torch.isnan(out1).any()
        torch.testing.assert_close(out0.cpu(), want0, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out1.cpu(), want1, rtol=1e-4, atol=1e-5)
        _lib.clear_plan_cache()
    _lib.tune_reset()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_hip_step_fused_columns_and_autotune(dev, world):
    """The sharded step with its two column-major backward passes as ONE launch (fuse_columns) and K | V as one grouped
    exchange, all shards on one GPU against the oracle; autotune() adopts one of the schedules it measured."""
    g = random_graph(500, 500, 16000, seed=23, chunk_size=8, zero_rows=0.1, hub=700)
    inp = rand_inputs(g, 1, 128, seed=24, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])

    def shard(rank, handle):
        sh = ShardedAttention.from_global_coo(g.src.to(dev), g.dst.to(dev), g.n_src, rank, world, dev, chunk_size=8, group=handle,
                                              pack_kv=True)
        sh.fuse_columns = True
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        args = [inp[k][lo:hi].to(dev).contiguous() for k in ("Q", "K", "V", "dO")]
        _lib.profile_enable(True)
        r = sh.step(*args)
        torch.cuda.synchronize()
        ext_ids = torch.cat([torch.arange(lo, hi, device=dev), sh.halo_ids])
        key = (sh.graph.src + lo) * g.n_dst + ext_ids[sh.graph.dst]
        return dict(lo=lo, hi=hi, key=key.cpu(), n_halo=sh.n_halo, recv=sh.recv_counts, **{k: v.detach().cpu() for k, v in r.items()})

    parts = run_local_shards(world, shard)
    tags = set(_lib.profile_read())
    _lib.profile_enable(False)
    assert "spmm_pair_cols" in tags and "spmm_bwd_dx" not in tags and "sddmm_bwd_dB" not in tags, tags
    _check(g, want, parts)

    def tuned(rank, handle):
        sh = ShardedAttention.from_global_coo(g.src.to(dev), g.dst.to(dev), g.n_src, rank, world, dev, chunk_size=8, group=handle)
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        args = [inp[k][lo:hi].to(dev).contiguous() for k in ("Q", "K", "V", "dO")]
        times = sh.autotune(*args, steps=1)
        assert len(times) == 8 and isinstance(sh.pack_kv, bool) and isinstance(sh.fuse_columns, bool) and isinstance(sh.use_forward_split, bool)
        r = sh.step(*args)
        torch.cuda.synchronize()
        ext_ids = torch.cat([torch.arange(lo, hi, device=dev), sh.halo_ids])
        key = (sh.graph.src + lo) * g.n_dst + ext_ids[sh.graph.dst]
        return dict(lo=lo, hi=hi, key=key.cpu(), n_halo=sh.n_halo, recv=sh.recv_counts, **{k: v.detach().cpu() for k, v in r.items()})

    _check(g, want, run_local_shards(world, tuned))
