"""Walk drivers (csrc/kernels_walk.h): lane groups own a bin of rows for a whole round, keep the rows'
partial sums in LDS and walk all column windows; the plan cuts the CSR into equal bins
(csrc/plan.hip: plan_get_walk).  Forced on small graphs through the tuning knobs, checked against
the oracle, against the window-owner drivers and against the chunk drivers."""
import pytest
import torch

from custom_op_benchmark_amd import _lib, graphs
import oracle
from util import random_graph, rand_inputs, oracle_step
from test_hip_parity import hip_step, close

pytestmark = pytest.mark.gpu

WALK_KERNELS = {"k_spmm_walk_f32"}


@pytest.fixture
def force_walk():
    _lib.tune_reset()
    _lib.tune("sweep_min_kb", 0); _lib.tune("walk_window_kb", 8); _lib.tune("walk_window_kb_col", 8); _lib.tune("walk_min_bin", 0)
    _lib.tune("sweep_min_granule", 0); _lib.tune("max_windows", 512); _lib.tune("walk", 6)
    _lib.clear_plan_cache()
    yield
    _lib.tune_reset()
    _lib.clear_plan_cache()


def kernels_of(step):
    _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    k = {r.get("kernel") for r in _lib.profile_read().values()}
    _lib.profile_enable(False)
    return k


@pytest.mark.parametrize("blocks", [8, 24, 0])
@pytest.mark.parametrize("d", [64, 128, 256])
def test_walk_vs_oracle(dev, force_walk, d, blocks):
    """Rows cut by bin boundaries (merged by atomics), empty rows, empty windows, a hub row spanning many
    bins, several rounds (small grids), bins of a handful of slots (the full grid on a small graph)."""
    _lib.tune("walk_blocks", blocks)
    n = 400 if d >= 256 else 1500
    g = random_graph(n, n + 41, 10 * n, seed=77 + d + blocks, chunk_size=32, zero_rows=0.15, hub=900)
    inp = rand_inputs(g, 1, d, seed=8, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]
    got = hip_step(gd, *args)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])
    assert WALK_KERNELS <= kernels_of(lambda: hip_step(gd, *args))


@pytest.mark.parametrize("blocks", [8, 0])
@pytest.mark.parametrize("h,d", [(8, 32), (4, 16), (2, 32), (4, 32), (8, 16), (2, 128)])
def test_walk_several_heads_vs_oracle(dev, force_walk, h, d, blocks):
    """SpMM-type walk kernel with h heads (the reference's (E, h) edge scalars against (N, h, d) rows,
    graphop_kernel.cu:100-130, 151-163): the feeder waves stage all h weights of a slot, a worker lane reads
    its head's; fewer rows per lane group where the weight rings take the LDS (256-B rows x 4 heads)."""
    _lib.tune("walk_blocks", blocks)
    n = 500 if h * d >= 256 else 1500
    g = random_graph(n, n + 41, 10 * n, seed=31 + h + d + blocks, chunk_size=32, zero_rows=0.15, hub=900)
    inp = rand_inputs(g, h, d, seed=9, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]
    got = hip_step(gd, *args)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])
    assert "k_spmm_walk_f32" in kernels_of(lambda: hip_step(gd, *args))


@pytest.mark.parametrize("drift", [0, 1, 2])
def test_walk_matches_other_drivers_medium(dev, force_walk, drift):
    """Same inputs through the walk, window-owner and chunk drivers: equal within fp32 re-association;
    pacing on and off (results never depend on it)."""
    _lib.tune("walk_drift", drift); _lib.tune("walk_window_kb", 256); _lib.tune("walk_window_kb_col", 256); _lib.tune("window_kb", 256)
    g = graphs.chung_lu_graph(20000, 1000000, alpha=0.6, seed=3).to(dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    Q, K, V, dO = (torch.rand(20000, 64, device=dev, generator=gen) for _ in range(4))
    wk = hip_step(g, Q, K, V, dO)
    assert WALK_KERNELS <= kernels_of(lambda: hip_step(g, Q, K, V, dO))
    _lib.tune("walk", 0)
    sw = hip_step(g, Q, K, V, dO)
    assert not (WALK_KERNELS & kernels_of(lambda: hip_step(g, Q, K, V, dO)))
    _lib.tune("sweep", 0)
    ch = hip_step(g, Q, K, V, dO)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        torch.testing.assert_close(wk[k], sw[k], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(wk[k], ch[k], rtol=2e-4, atol=2e-5)


def test_walk_rows_longer_than_many_bins(dev, force_walk):
    """One row holds a third of all slots: it is cut into pieces over dozens of bins (each piece takes its
    share of every window) and merged by atomics; the other rows are short."""
    n = 3000
    gen = torch.Generator().manual_seed(5)
    src = torch.cat([torch.randint(0, n, (40000,), generator=gen), torch.full((20000,), 17)])
    dst = torch.randint(0, n, (60000,), generator=gen)
    g = graphs.graph_from_coo(src, dst, n, n, chunk_size=32)
    _lib.tune("walk_blocks", 16); _lib.tune("walk_window_kb", 64); _lib.tune("walk_window_kb_col", 64)
    inp = rand_inputs(g, 1, 64, seed=3, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])


@pytest.mark.parametrize("h,d", [(1, 64), (4, 16)])
def test_prepare_builds_walk_layouts_and_the_step_captures(dev, force_walk, h, d):
    """graphop.prepare builds the walk layouts too: the step afterwards allocates nothing, so it captures
    into a HIP graph, and the replay (walk kernels with their pacer counters zeroed inside the graph)
    reproduces the eager result."""
    from custom_op_benchmark_amd import graphop as ops
    _lib.tune("walk_blocks", 16)
    g = random_graph(1500, 1500, 15000, seed=13, chunk_size=32, hub=900).to(dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    shape = (1500, 64) if h == 1 else (1500, h, d)
    Q, K, V, dO = (torch.randn(shape, device=dev, generator=gen) / 8 for _ in range(4))
    ops.prepare(g, h=h, d=d, fused=False)
    held = _lib.plan_memory_bytes()
    want = hip_step(g, Q, K, V, dO)
    assert _lib.plan_memory_bytes() == held            # the eager step built nothing after prepare
    assert WALK_KERNELS <= kernels_of(lambda: hip_step(g, Q, K, V, dO))
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    s = ops.maskedmm_csr_forward(*a4, Q, K)
    a = ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.vector_spmm_forward(*a4, a, V)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o = ops.vector_spmm_forward(*a4, a, V)
        dQ, dK = ops.maskedmm_csr_backward(*g.csr_args(), Q, K, a)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    torch.testing.assert_close(o, want["o"], rtol=1e-5, atol=1e-6)
    dQ2, dK2 = ops.maskedmm_csr_backward(*g.csr_args(), Q, K, a)
    torch.testing.assert_close(dQ, dQ2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dK, dK2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("fault,what", [(1, "ring chunk"), (2, "previous step")])
def test_hand_over_timeout_is_an_error_not_a_wrong_result(dev, force_walk, fault, what):
    """The worker <-> feeder and quad hand-overs of the walk kernel are bounded spins (a launch must not hang the
    device).  When a bound expires the wave must NOT carry on with whatever the ring holds: the workgroup aborts,
    a code lands in the host-visible error word and the library raises (include/graphop_hip.h:
    graphop_check_device_errors).  Forced with the fault-injection knob: 1 = a feeder never publishes one chunk,
    2 = a quad never reports its first step finished.  Run once; afterwards the library works again."""
    from custom_op_benchmark_amd import graphop as ops
    _lib.tune("walk_blocks", 8)
    g = random_graph(1500, 1500, 30000, seed=21, chunk_size=32, hub=900).to(dev)
    gen = torch.Generator(device=dev).manual_seed(6)
    V = torch.randn(1500, 64, device=dev, generator=gen)
    a = torch.rand(g.eid_r.numel(), device=dev, generator=gen)
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    good = ops.vector_spmm_forward(*a4, a, V)
    _lib.check_errors()                               # nothing reported by a healthy launch
    _lib.tune("walk_fault", fault)
    ops.vector_spmm_forward(*a4, a, V)                # asynchronous: the launch itself returns
    with pytest.raises(RuntimeError, match=what):
        _lib.check_errors()                           # synchronises, then looks at the error word
    _lib.tune("walk_fault", 0)
    ops.vector_spmm_forward(*a4, a, V)                # a failed launch is reported by the next entry point too ...
    _lib.tune("walk_fault", fault)
    ops.vector_spmm_forward(*a4, a, V)
    torch.cuda.synchronize()
    _lib.tune("walk_fault", 0)
    with pytest.raises(RuntimeError, match="walk kernel"):
        ops.vector_spmm_forward(*a4, a, V)            # ... here: the entry point checks the record before launching
    # ABI 7: the record is STICKY -- an entry point reports it but does not clear it (a thread that ignores a return
    # code must not swallow the failure for everybody else); the message names the pass and the device
    with pytest.raises(RuntimeError, match=r"pass 'spmm_fwd' on device \d+, walk launch #\d+"):
        ops.vector_spmm_forward(*a4, a, V)
    with pytest.raises(RuntimeError, match="sticky"):
        ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, a)     # any op, not only the walk's
    with pytest.raises(RuntimeError, match=what):
        _lib.check_errors(sync=False)                 # graphop_check_device_errors: reports AND clears
    again = ops.vector_spmm_forward(*a4, a, V)        # acknowledged: the library works again
    _lib.check_errors()
    torch.testing.assert_close(again, good, rtol=1e-5, atol=1e-6)


def test_step_exit_reports_an_abort_of_its_own_launches(dev, force_walk):
    """functions.attention_step looks at the device error record on its way out (round-4 advice: an abort in the LAST
    launch of a step was only reported by the next graphop call, after an optimizer could have consumed the gradients).
    The check does not synchronise: an abort that has already landed is raised by this step, one still in flight by
    the next call at the latest (sticky)."""
    from custom_op_benchmark_amd import functions
    _lib.tune("walk_blocks", 8)
    g = random_graph(1500, 1500, 30000, seed=21, chunk_size=32, hub=900).to(dev)
    gen = torch.Generator(device=dev).manual_seed(6)
    Q, K, V, dO = (torch.randn(1500, 64, device=dev, generator=gen) for _ in range(4))
    for t in (Q, K, V):
        t.requires_grad_(True)
    functions.attention_step(g, Q, K, V, dO)
    _lib.check_errors()
    _lib.tune("walk_fault", 1)
    try:
        with pytest.raises(RuntimeError, match="walk kernel"):
            functions.attention_step(g, Q, K, V, dO)          # spmm_fwd aborts early in the step: a later op or the exit check raises
            torch.cuda.synchronize()
            functions.attention_step(g, Q, K, V, dO)          # (if the first step's launches were all still in flight)
    finally:
        _lib.tune("walk_fault", 0)
        torch.cuda.synchronize()
        try:
            _lib.check_errors()
        except RuntimeError:
            pass
    functions.attention_step(g, Q, K, V, dO)
    _lib.check_errors()


def kernels_by_tag(step):
    _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    k = {tag: r.get("kernel") for tag, r in _lib.profile_read().items()}
    _lib.profile_enable(False)
    return k


@pytest.mark.parametrize("blocks", [8, 0])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_fp64_on_the_walk_and_window_drivers_vs_oracle(dev, force_walk, d, blocks):
    """fp64 takes the same plan-driven kernels as fp32 (the reference dispatches both types through the same
    kernels, graphop_kernel.cu:291): rows of 256 B / 512 B / 1 KB = the lane-group shapes of fp32 d = 64 / 128 / 256.
    Forced small geometry: SDDMM-type passes on the staged window-owner strip, SpMM-type passes (row- and
    column-major) on the walk kernel, 8-byte weights through the feeder rings, dense fp64 atomics for the rows
    a bin shares; against the oracle at the fp64 tolerance."""
    from test_hip_parity import TOL
    _lib.tune("walk_blocks", blocks); _lib.tune("window_kb", 8); _lib.tune("vrow_t", 64)
    n = 400 if d >= 128 else 1500
    g = random_graph(n, n + 41, 10 * n, seed=5 + d + blocks, chunk_size=32, zero_rows=0.15, hub=900)
    inp = rand_inputs(g, 1, d, seed=8, dtype=torch.float64, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]
    got = hip_step(gd, *args)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k], torch.float64)
    kern = kernels_by_tag(lambda: hip_step(gd, *args))
    assert kern["sddmm_fwd"] == kern["spmm_bwd_dedata"] == "k_sddmm_wown_staged_f64", kern
    assert all(kern[t] == "k_spmm_walk_f64" for t in ("spmm_fwd", "spmm_bwd_dx", "sddmm_bwd_dA", "sddmm_bwd_dB")), kern
    # the generic kernels (another family) agree
    _lib.tune("force_generic", 1)
    ref = hip_step(gd, *args)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        torch.testing.assert_close(got[k], ref[k], rtol=1e-10, atol=1e-12)


def test_tables_of_4gib_and_more_take_64bit_row_offsets(dev, force_walk):
    """Walk and staged window-owner kernels on a gathered table of >= 4 GiB (64-bit row offsets instead of the
    32-bit vector offset): a 4.3 GB table of 256-B rows with a few thousand edges whose neighbours lie at both
    ends of it, against the chunk drivers (which always used 64-bit indexing)."""
    n_cols = (1 << 24) + 1024                                  # x 256 B = 4.0 GiB + 256 KB
    n_rows = 1500
    gen = torch.Generator().manual_seed(3)
    src = torch.randint(0, n_rows, (40000,), generator=gen)
    dst = torch.cat([torch.randint(0, 3000, (20000,), generator=gen), n_cols - 1 - torch.randint(0, 3000, (20000,), generator=gen)])
    g = graphs.graph_from_coo(src, dst, n_rows, n_cols, chunk_size=32).to(dev)
    _lib.tune("walk_blocks", 8); _lib.tune("window_kb", 1 << 19); _lib.tune("walk_window_kb", 1 << 19)
    _lib.tune("walk_window_kb_col", 1 << 19); _lib.tune("vrow_t", 64)
    K = torch.zeros(n_cols, 64, device=dev)
    gk = torch.Generator(device=dev).manual_seed(4)
    K[:3000] = torch.randn(3000, 64, device=dev, generator=gk); K[-3000:] = torch.randn(3000, 64, device=dev, generator=gk)
    Q = torch.randn(n_rows, 64, device=dev, generator=gk)
    w = torch.rand(g.eid_r.numel(), device=dev, generator=gk)
    from custom_op_benchmark_amd import graphop as ops
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    kern = kernels_by_tag(lambda: (ops.maskedmm_csr_forward(*a4, Q, K), ops.vector_spmm_forward(*a4, w, K)))
    assert kern["sddmm_fwd"] == "k_sddmm_wown_staged_f32" and kern["spmm_fwd"] == "k_spmm_walk_f32", kern
    s1 = ops.maskedmm_csr_forward(*a4, Q, K)
    o1 = ops.vector_spmm_forward(*a4, w, K)[:n_rows]
    _lib.tune("sweep", 0); _lib.tune("walk", 0)
    s0 = ops.maskedmm_csr_forward(*a4, Q, K)
    o0 = ops.vector_spmm_forward(*a4, w, K)[:n_rows]
    torch.testing.assert_close(s1, s0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(o1, o0, rtol=1e-4, atol=1e-5)
    assert float(s0.abs().max()) > 0 and float(o0.abs().max()) > 0
