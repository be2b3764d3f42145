"""Multi-GPU path on CPU: world_size 2 (and 3) over gloo, local operators = the CPU oracle.
The sharded step, re-assembled, must reproduce the single-process step on the global graph; the
local<->global index maps must be exact."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from custom_op_benchmark_amd import graphs
from custom_op_benchmark_amd.dist import ShardedAttention, balanced_ranges, run_local_shards

from util import oracle_step, rand_inputs, random_graph


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, h, d, out_dir, pack_kv, split_forward):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = random_graph(97, 97, 2500, seed=21, chunk_size=8, zero_rows=0.1, hub=300)
        inp = rand_inputs(g, h, d, seed=22, normal=True)
        sh = ShardedAttention.from_global_coo(g.src, g.dst, g.n_src, rank, world, "cpu", chunk_size=8, ops=oracle,
                                              pack_kv=pack_kv, split_forward=split_forward)
        assert (sh.fwd_halves is not None) == (split_forward and sh.n_halo > 0)
        if sh.fwd_halves is not None:
            # the two halves cut the row-major slots into disjoint sets that cover all of them, own columns / halo columns
            own, halo = sh.fwd_halves
            both = torch.cat([own["slots"], halo["slots"]])
            assert torch.equal(torch.sort(both).values, torch.arange(sh.graph.n_edges))
            assert bool((own["indices"] < sh.n_own).all()) and bool((halo["indices"] >= sh.n_own).all())
            assert int(own["ptr"][-1]) == own["slots"].numel() and int(halo["ptr"][-1]) == halo["slots"].numel()
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        # index maps: local column ids map back to the global ids exactly
        ext_ids = torch.cat([torch.arange(lo, hi), sh.halo_ids])
        m = (g.src >= lo) & (g.src < hi)
        # (the local CSR orders a row's slots by LOCAL column id: own columns first, then halo)
        key_local = (sh.graph.src + lo) * g.n_dst + ext_ids[sh.graph.dst]
        key_global = g.src[m] * g.n_dst + g.dst[m]
        assert torch.equal(torch.sort(key_local).values, torch.sort(key_global).values)
        assert sum(sh.recv_counts) == sh.n_halo and sh.recv_counts[rank] == 0
        r = sh.step(inp["Q"][lo:hi], inp["K"][lo:hi], inp["V"][lo:hi], inp["dO"][lo:hi])
        assert sh.collectives_last_step == (3 if pack_kv else 4)     # K | V as one grouped exchange, or K and V; dV; dK
        if world == 3:
            # autotune: both exchange schedules timed, max over ranks, every rank adopts the same one
            times = sh.autotune(inp["Q"][lo:hi], inp["K"][lo:hi], inp["V"][lo:hi], inp["dO"][lo:hi], steps=1)
            assert set(times) == {"kv_%s+columns_split+forward_%s" % (k, f) for k in ("separate", "packed") for f in ("split", "whole")}
            picks = [None] * world
            dist.all_gather_object(picks, (sh.pack_kv, sh.fuse_columns, sh.use_forward_split, sorted(times.items())))
            assert all(p == picks[0] for p in picks), picks
            r2 = sh.step(inp["Q"][lo:hi], inp["K"][lo:hi], inp["V"][lo:hi], inp["dO"][lo:hi])
            for k in ("o", "dQ", "dK", "dV"):
                torch.testing.assert_close(r2[k], r[k], rtol=1e-5, atol=1e-6)
        torch.save({k: v for k, v in r.items()} | {"lo": lo, "hi": hi, "edge_mask": m, "key": key_local},
                   os.path.join(out_dir, "r%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,d,pack_kv,split_forward", [(2, 1, 16, False, True), (3, 2, 8, True, True), (2, 2, 8, True, False)])
def test_sharded_step_matches_single_process(tmp_path, world, h, d, pack_kv, split_forward):
    """(round 5) split_forward: the SDDMM forward as an own-column half under the K exchange + a halo-column half behind
    it, both writing the one score array; pack_kv: K | V halo rows as one grouped exchange (batch_isend_irecv)."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, h, d, str(tmp_path), pack_kv, split_forward), nprocs=world, join=True)
    g = random_graph(97, 97, 2500, seed=21, chunk_size=8, zero_rows=0.1, hub=300)
    inp = rand_inputs(g, h, d, seed=22, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = {k: torch.zeros_like(want[k]) for k in ("o", "dQ", "dK", "dV", "s", "a")}
    for r in range(world):
        z = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        lo, hi = z["lo"], z["hi"]
        for k in ("o", "dQ", "dK", "dV"):
            got[k][lo:hi] = z[k]
        # a rank's edges are the global edges of its rows; order them by (src, global dst).
        # (duplicate edges carry identical values, so ties are harmless)
        order = torch.argsort(z["key"], stable=True)
        got["s"][z["edge_mask"]] = z["s"][order]
        got["a"][z["edge_mask"]] = z["a"][order]
    for k in got:
        torch.testing.assert_close(got[k], want[k], rtol=1e-4, atol=1e-5, msg=lambda m: k + ": " + m)


def test_balanced_ranges():
    deg = torch.tensor([0, 10, 0, 0, 5, 5, 20, 0, 0, 0])
    b = balanced_ranges(deg, 4)
    assert b[0] == 0 and b[-1] == 10 and all(x <= y for x, y in zip(b, b[1:]))
    assert balanced_ranges(deg, 1) == [0, 10]
    assert balanced_ranges(torch.zeros(5, dtype=torch.int64), 3)[-1] == 5


def test_single_rank_has_no_halo():
    g = graphs.uniform_random_graph(40, 400, seed=3, chunk_size=8)
    inp = rand_inputs(g, 1, 8, seed=4)
    sh = ShardedAttention(0, 1, [0, 40], g.src, g.dst, "cpu", chunk_size=8, ops=oracle)
    assert sh.n_halo == 0
    r = sh.step(inp["Q"], inp["K"], inp["V"], inp["dO"])
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    for k in ("o", "dQ", "dK", "dV"):
        torch.testing.assert_close(r[k], want[k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("world", [2, 4])
def test_local_group_shards_match_single_process(world):
    """dist.LocalGroup (all shards in one process, exchanges = copies between the shards' tensors):
    the mode the GPU tier uses to run the sharded HIP step on one GPU (tests/test_dist_gpu.py)."""
    g = random_graph(97, 97, 2500, seed=21, chunk_size=8, zero_rows=0.1, hub=300)
    inp = rand_inputs(g, 1, 16, seed=22, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])

    def shard(rank, handle):
        sh = ShardedAttention.from_global_coo(g.src, g.dst, g.n_src, rank, world, "cpu", chunk_size=8, ops=oracle,
                                              group=handle, pack_kv=(world == 4))
        lo, hi = sh.bounds[rank], sh.bounds[rank + 1]
        K, V = inp["K"][lo:hi], inp["V"][lo:hi]
        if world == 4:      # the caller keeps its K / V rows inside the extended buffers: no own-row copy per exchange
            K = sh.own_rows_view("K", K.shape[1:]).copy_(K)
            V = sh.own_rows_view("V", V.shape[1:]).copy_(V)
            assert K.data_ptr() == sh._ext_buffer("K", K).data_ptr()
        return lo, hi, sh.step(inp["Q"][lo:hi], K, V, inp["dO"][lo:hi])

    for lo, hi, r in run_local_shards(world, shard):
        for k in ("o", "dQ", "dK", "dV"):
            torch.testing.assert_close(r[k], want[k][lo:hi], rtol=1e-4, atol=1e-5)


def test_local_group_propagates_errors():
    def shard(rank, handle):
        if rank == 1:
            raise ValueError("boom")
        handle.group.exchange_counts(rank, [0, 0])
    with pytest.raises(ValueError, match="boom"):
        run_local_shards(2, shard, timeout=20.0)


def test_rmat_shard_edges_stay_in_range():
    from custom_op_benchmark_amd.graphs import rmat_edges
    src, dst = rmat_edges(10, 5000, seed=1, src_prefix_bits=2, src_prefix=3)
    assert int(src.min()) >= 768 and int(src.max()) < 1024 and int(dst.min()) >= 0 and int(dst.max()) < 1024
    s2, d2 = rmat_edges(10, 20000, seed=2)
    # Graph500 marginals: P(src top bit = 0) = a + b = 0.76, P(dst top bit = 0) = a + c = 0.76
    assert abs(float((s2 < 512).float().mean()) - 0.76) < 0.02
    assert abs(float((d2 < 512).float().mean()) - 0.76) < 0.02


def _self_halo_worker(rank, world, port, out_dir, pack_kv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = random_graph(97, 97, 2500, seed=21, chunk_size=8, zero_rows=0.1, hub=300)
        inp = rand_inputs(g, 2, 8, seed=22, normal=True)
        mask = torch.rand(g.src.numel(), generator=torch.Generator().manual_seed(5)) < 0.5
        calls, groups = [], []
        real, real_b = dist.all_to_all_single, dist.batch_isend_irecv
        dist.all_to_all_single = lambda *a, **k: (calls.append(bool(k.get("async_op"))), real(*a, **k))[1]
        dist.batch_isend_irecv = lambda ops: (groups.append(len(ops)), real_b(ops))[1]
        sh = ShardedAttention.from_global_coo(g.src, g.dst, g.n_src, rank, world, "cpu", chunk_size=8, ops=oracle,
                                              force_collectives=True, halo_mask=mask, pack_kv=pack_kv)
        assert sh.n_halo > 0 and sh.recv_counts == [sh.n_halo] and len(calls) == 2
        r = sh.step(inp["Q"], inp["K"], inp["V"], inp["dO"])
        if pack_kv:     # K | V: one grouped exchange (gloo has no send-to-self: the own segment is a copy, the group is empty)
            assert len(calls) == 4 and all(calls[2:]) and sh.collectives_last_step == 3
        else:
            assert len(calls) == 6 and all(calls[2:]) and not groups    # K, V, dV, dK went through the process group, async
        want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
        for k in ("o", "dQ", "dK", "dV"):
            torch.testing.assert_close(r[k], want[k], rtol=1e-4, atol=1e-5)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pack_kv", [False, True])
def test_forced_collectives_at_world_size_one(tmp_path, pack_kv):
    """force_collectives: a one-rank process group still goes through all_to_all_single (no world == 1 short-cut),
    and a self-halo (halo_mask) makes the exchanges move real rows: the CPU twin of
    tests/test_dist_gpu.py::test_rccl_path_at_world_size_one."""
    mp.spawn(_self_halo_worker, args=(1, _free_port(), str(tmp_path), pack_kv), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


def test_group_rows_matches_index_add():
    """_lib.group_rows: the (ptr, rows, pos) grouping the HIP add-home kernel consumes (graphop_add_rows_grouped), checked
    on CPU against index_add_: rows distinct and ascending, every position listed once, peer order kept inside a group."""
    from custom_op_benchmark_amd import _lib
    gen = torch.Generator().manual_seed(0)
    idx = torch.randint(0, 50, (400,), generator=gen)
    ptr_, rows, pos = _lib.group_rows(idx)
    assert torch.equal(rows, torch.unique(idx)) and int(ptr_[-1]) == idx.numel() and torch.equal(torch.sort(pos).values, torch.arange(400))
    src = torch.rand(400, 3, generator=gen)
    want = torch.zeros(50, 3).index_add_(0, idx, src)
    got = torch.zeros(50, 3)
    for g in range(rows.numel()):
        p = pos[int(ptr_[g]):int(ptr_[g + 1])]
        assert torch.all(idx[p] == rows[g]) and torch.all(p[1:] > p[:-1])      # stable: positions ascend inside a group
        got[rows[g]] = src[p].sum(0)
    torch.testing.assert_close(got, want)
