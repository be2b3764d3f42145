"""GPU tier: plan lifecycle -- explicit prepare / release, HIP-graph capture safety, LRU cache,
torch-backed plan memory, and the graph container that carries plan state."""
import pytest
import torch

import oracle
from custom_op_benchmark_amd import _lib, functions, graphs
from custom_op_benchmark_amd import graphop as ops

from util import oracle_step, rand_inputs, random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_windows():
    _lib.tune("sweep_min_kb", 0); _lib.tune("window_kb", 8); _lib.tune("vrow_t", 64)
    _lib.tune("sweep_min_granule", 0); _lib.clear_plan_cache()
    yield
    _lib.tune_reset(); _lib.clear_plan_cache()


def _inputs(dev, n, d, seed=1):
    gen = torch.Generator(device=dev).manual_seed(seed)
    return [torch.randn(n, d, device=dev, generator=gen) / 8 for _ in range(4)]


def test_capture_without_prepare_fails_clearly_and_with_prepare_works(dev, small_windows):
    g = random_graph(1500, 1500, 15000, seed=3, chunk_size=32, hub=900).to(dev)
    Q, K, V, dO = _inputs(dev, 1500, 64)
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    _lib.get_plan(*a4, 1500)                      # the plan exists, its window structures do not
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.empty(1, device=dev)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="captured into a HIP graph"):
        with torch.cuda.graph(graph):
            ops.maskedmm_csr_forward(*a4, Q, K)
    torch.cuda.synchronize()
    # prepare builds plans + window structures; then the same capture succeeds and replays correctly
    ops.prepare(g, h=1, d=64)
    before = _lib.plan_memory_bytes()
    want = ops.maskedmm_csr_forward(*a4, Q, K)
    assert _lib.plan_memory_bytes() == before      # the eager call built nothing new
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.maskedmm_csr_forward(*a4, Q, K)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = ops.maskedmm_csr_forward(*a4, Q, K)
    graph.replay(); torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_prepare_covers_the_whole_step_and_release_frees(dev, small_windows):
    g = random_graph(1200, 1200, 14000, seed=4, chunk_size=32, hub=700).to(dev)
    Q, K, V, dO = _inputs(dev, 1200, 64, seed=2)
    base = _lib.plan_memory_bytes()
    ops.prepare(g, h=1, d=64, fused=True)
    held = _lib.plan_memory_bytes()
    assert held > base
    q, k, v = (x.clone().requires_grad_(True) for x in (Q, K, V))
    functions.attention_step(g, q, k, v, dO)
    functions.fused_attention_step(g, q, k, v, dO)
    torch.cuda.synchronize()
    assert _lib.plan_memory_bytes() == held        # neither step built anything after prepare
    ops.release(g)
    import gc; gc.collect()
    assert _lib.plan_memory_bytes() <= base


def test_plan_cache_is_capped_by_bytes(dev):
    """The cache also evicts least-recently-used graphs while plan memory exceeds its byte budget
    (GRAPHOP_PLAN_CACHE_GB): the evicted graph's plans are destroyed, their memory returns to torch."""
    import gc
    _lib.clear_plan_cache(); gc.collect()
    base = _lib.plan_memory_bytes()
    gs = [graphs.uniform_random_graph(3000, 60000, seed=s).to(dev) for s in range(3)]
    old = _lib._PLAN_CACHE_BYTES
    try:
        _lib.get_plan(gs[0].row, gs[0].ptr_r, gs[0].eid_r, gs[0].indices_r, 3000)
        one = _lib.plan_memory_bytes() - base
        assert one > 0
        _lib._PLAN_CACHE_BYTES = base + one // 2          # less than one graph's plans
        for g in gs[1:]:
            _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, 3000)
            gc.collect()
            assert len(_lib._plan_cache) == 1                                    # the previous graph went
            assert _lib.plan_memory_bytes() - base <= one + one // 4             # and so did its memory
        assert "_graphop_plans" not in gs[0].row.__dict__ and "_graphop_plans" in gs[2].row.__dict__
        # an evicted graph still works: its plan is rebuilt on the next use
        A = torch.rand(3000, 16, device=dev)
        y = ops.maskedmm_csr_forward(gs[0].row, gs[0].ptr_r, gs[0].eid_r, gs[0].indices_r, A, A)
        assert torch.isfinite(y).all()
    finally:
        _lib._PLAN_CACHE_BYTES = old
        _lib.clear_plan_cache()


def test_plan_cache_is_lru(dev):
    _lib.clear_plan_cache()
    gs = [graphs.uniform_random_graph(20, 100, seed=s).to(dev) for s in range(3)]
    old_max = _lib._PLAN_CACHE_MAX
    _lib._PLAN_CACHE_MAX = 2
    try:
        p0 = _lib.get_plan(gs[0].row, gs[0].ptr_r, gs[0].eid_r, gs[0].indices_r, 20)
        _lib.get_plan(gs[1].row, gs[1].ptr_r, gs[1].eid_r, gs[1].indices_r, 20)
        # touch graph 0 through the slow path (a fresh view of the same storage has no fast-path note)
        r0 = gs[0].row.view(-1)
        assert _lib.get_plan(r0, gs[0].ptr_r, gs[0].eid_r, gs[0].indices_r, 20) is p0
        _lib.get_plan(gs[2].row, gs[2].ptr_r, gs[2].eid_r, gs[2].indices_r, 20)     # evicts graph 1, not 0
        keys = list(_lib._plan_cache)
        assert len(keys) == 2 and _lib._key(gs[0].row, gs[0].ptr_r, gs[0].eid_r) in keys
        assert _lib._key(gs[1].row, gs[1].ptr_r, gs[1].eid_r) not in keys
    finally:
        _lib._PLAN_CACHE_MAX = old_max
        _lib.clear_plan_cache()


def test_in_place_edit_invalidates_the_fast_path(dev):
    g = graphs.uniform_random_graph(30, 300, seed=7).to(dev)
    A = torch.rand(30, 16, device=dev)
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    y0 = ops.maskedmm_csr_forward(*a4, A, A)
    p0 = _lib.get_plan(*a4, 30)
    g.indices_r[:] = g.indices_r.flip(0)                 # same storage, new contents (version bump)
    p1 = _lib.get_plan(*a4, 30)
    assert p1 is not p0
    y1 = ops.maskedmm_csr_forward(*a4, A, A)
    want = oracle.maskedmm_csr_forward(*(t.cpu() for t in a4), A.cpu(), A.cpu())
    torch.testing.assert_close(y1.cpu(), want, rtol=1e-5, atol=1e-6)
    assert not torch.equal(y0, y1)


def test_container_carries_plan_state(dev, tmp_path, small_windows):
    """save_graph on a prepared GPU graph stores info records, segment tables, 32-bit mirrors and the
    window structures; load_graph(device=gpu) imports them: the first step builds nothing and matches
    the oracle."""
    g = random_graph(1500, 1500, 15000, seed=9, chunk_size=32, zero_rows=0.1, hub=900)
    inp = rand_inputs(g, 1, 64, seed=8, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"][:g.n_src])
    gd = g.to(dev)
    ops.prepare(gd, h=1, d=64)
    path = str(tmp_path / "g2.pt")
    graphs.save_graph(gd, path)
    st = torch.load(path)
    assert st["format"] == 2 and set(st["plans"]) == {"r", "c"}
    assert st["plans"]["r"]["idx32"].dtype == torch.int32 and len(st["plans"]["r"]["sweeps"]) >= 1
    assert st["plans"]["c"]["eid32"] is not None and st["plans"]["r"]["eid32"] is None   # identity eid: no mirror
    ops.release(gd); del gd
    _lib.clear_plan_cache()
    g2 = graphs.load_graph(path, device=dev)
    pr = _lib.get_plan(g2.row, g2.ptr_r, g2.eid_r, g2.indices_r, g2.n_dst)
    assert pr.info.row_owned and pr.info.sorted_in_rows and pr.info.has_idx32 and pr.info.eid_identity
    held = _lib.plan_memory_bytes()
    q, k, v = (inp[x].to(dev).requires_grad_(True) for x in ("Q", "K", "V"))
    s, a, o = functions.attention_step(g2, q, k, v, inp["dO"][:g.n_src].to(dev))
    torch.cuda.synchronize()
    assert _lib.plan_memory_bytes() <= held           # nothing built (the plans may DROP imported builder inputs: plan_trim)
    for name, got, ref in (("s", s, want["s"]), ("a", a, want["a"]), ("o", o, want["o"]), ("dQ", q.grad, want["dQ"]),
                           ("dK", k.grad, want["dK"]), ("dV", v.grad, want["dV"])):
        torch.testing.assert_close(got.detach().cpu(), ref, rtol=1e-4, atol=1e-5, msg=lambda m: name + ": " + m)


def test_dlpack_arrays_feed_the_ops(dev):
    g = graphs.uniform_random_graph(64, 2000, seed=3).to(dev)
    g2 = graphs.from_dlpack(graphs.to_dlpack(g), 64)
    A = torch.rand(64, 32, device=dev)
    y = ops.maskedmm_csr_forward(g2.row, g2.ptr_r, g2.eid_r, g2.indices_r, A, A)
    assert torch.equal(y, ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, A, A))


def test_plan_trim_drops_builder_inputs_and_rebuilds_them_on_demand(dev, small_windows):
    """Round 5 (plan memory): the 32-bit mirrors and a window structure's wp tables are inputs of the layout builders;
    a plan drops them once a STAGED pass has its dealt layout (plan_trim) and rebuilds them when a builder or a per-batch
    (non-staged) kernel needs them again, which then PINS them.  One graph, one pair of plans: staged -> per-batch ->
    staged -> a new geometry -> export; the results never change, the memory goes down, up, and stays."""
    g = random_graph(1500, 1500, 30000, seed=12, chunk_size=32, zero_rows=0.1, hub=900)
    inp = rand_inputs(g, 1, 64, seed=13, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]

    def step():
        q, k, v = (x.clone().requires_grad_(True) for x in args[:3])
        s, a, o = functions.attention_step(gd, q, k, v, args[3])
        torch.cuda.synchronize()
        for name, got, ref in (("s", s, want["s"]), ("o", o, want["o"]), ("dQ", q.grad, want["dQ"]), ("dK", k.grad, want["dK"]),
                               ("dV", v.grad, want["dV"])):
            torch.testing.assert_close(got.detach().cpu(), ref, rtol=1e-4, atol=1e-5, msg=lambda m: name + ": " + m)

    _lib.tune("walk", 0)                        # window-owner kernels for every gather pass: staged strips read dealt copies only
    _lib.tune("plan_trim", 0); _lib.clear_plan_cache()
    step()
    untrimmed = _lib.plan_memory_bytes()
    _lib.tune("plan_trim", 1); _lib.clear_plan_cache()
    step()
    trimmed = _lib.plan_memory_bytes()
    assert trimmed < untrimmed - 4 * 3 * g.n_edges * 0.9, (trimmed, untrimmed)     # idx32 (x2) + eid32 are gone, and the wp tables
    _lib.tune("staged_ids", 0)                  # per-batch kernels: they READ the mirrors and the wp tables -> rebuilt and pinned
    step()
    pinned = _lib.plan_memory_bytes()
    assert pinned > trimmed + 4 * 3 * g.n_edges * 0.9
    _lib.tune("staged_ids", 7)
    step()
    assert _lib.plan_memory_bytes() == pinned   # pinned arrays stay (launches on other streams may be reading them)
    _lib.tune("window_kb", 16)                  # a new geometry on the same plans: its builders find their inputs
    step()
    # export after trimming (fresh plans, staged only): the container still holds the full derived state
    _lib.clear_plan_cache()
    _lib.tune("window_kb", 8)
    step()
    st = _lib.get_plan(gd.row, gd.ptr_r, gd.eid_r, gd.indices_r, g.n_dst).export_state()
    assert st["idx32"] is not None and torch.equal(st["idx32"].long(), g.indices_r)
    assert all(sw["wp_lo"] is not None and sw["wp_hi"] is not None for sw in st["sweeps"]) and st["sweeps"]
