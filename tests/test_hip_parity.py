"""GPU tier: the HIP path (through the C ABI, via the Python binding) against the CPU oracle on
the same seeded inputs, against the committed golden fixtures, and -- at sizes the oracle cannot
reach -- through size-independent properties.

Tolerances: north_star allows 1e-3 fp32; the tests hold the HIP path to rtol 1e-4 / atol 1e-5
(fp32, summation order differs from the oracle's serial loops) and 1e-10 / 1e-12 (fp64).
Index outputs (partition_csr) are compared bit-exact in test_partition_csr.py.
"""
import numpy as np
import pytest
import torch

import oracle
from custom_op_benchmark_amd import _lib, functions, graphs
from custom_op_benchmark_amd import graphop as ops

from util import oracle_step, rand_inputs, random_graph, t

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(rtol=1e-4, atol=1e-5), torch.float64: dict(rtol=1e-10, atol=1e-12)}


def close(got, want, dtype=torch.float32, **kw):
    tol = dict(TOL[dtype]); tol.update(kw)
    torch.testing.assert_close(got.cpu(), want, **tol)


def hip_step(g, Q, K, V, dO):
    """Composed step through the product's autograd classes on the GPU."""
    Q = Q.clone().requires_grad_(True); K = K.clone().requires_grad_(True); V = V.clone().requires_grad_(True)
    a8 = g.csr_args()
    s = functions.MaskedMMCSR.apply(*a8, Q, K)
    a = functions.SparseSoftmax.apply(g.row, g.ptr_r, g.eid_r, s)
    o = functions.VectorSPMM.apply(*a8, a, V)
    o.backward(dO)
    torch.cuda.synchronize()
    return dict(s=s.detach(), a=a.detach(), o=o.detach(), dQ=Q.grad, dK=K.grad, dV=V.grad)


# ---- golden fixtures -----------------------------------------------------------------------------
@pytest.mark.parametrize("cs", [3, 32])
@pytest.mark.parametrize("h", [1, 8])
def test_k2_harness_fixture_on_gpu(golden, dev, cs, h):
    z = golden("k2_harness_small.npz")
    a8 = tuple(t(z["cs%d_%s" % (cs, k)], dev) for k in ("row", "ptr_r", "eid_r", "indices_r", "col",
                                                        "ptr_c", "eid_c", "indices_c"))
    p = "h%d_" % h
    A, B, ge, gn, x, w = (t(z[p + k], dev) for k in ("A", "B", "grad_e", "grad_n", "x", "w"))
    close(ops.maskedmm_csr_forward(*a8[:4], A, B), t(z[p + "sddmm_y"]))
    dA, dB = ops.maskedmm_csr_backward(*a8, A, B, ge)
    close(dA, t(z[p + "sddmm_dA"])); close(dB, t(z[p + "sddmm_dB"]))
    y = ops.sparse_softmax_forward(*a8[:3], x)
    close(y, t(z[p + "sm_scatter_y"]))
    close(ops.sparse_softmax_backward(*a8[:3], y, ge), t(z[p + "sm_scatter_dx"]), rtol=1e-3, atol=1e-6)
    y = ops.sparse_softmax_forward(*a8[4:7], x)
    close(y, t(z[p + "sm_gather_y"]))
    close(ops.sparse_softmax_backward(*a8[4:7], y, ge), t(z[p + "sm_gather_dx"]), rtol=1e-3, atol=1e-6)
    close(ops.vector_spmm_forward(*a8[:4], w, A), t(z[p + "spmm_y"]))
    dw, dx = ops.vector_spmm_backward(*a8, w, gn, A)
    close(dw, t(z[p + "spmm_dw"])); close(dx, t(z[p + "spmm_dx"]))


def test_k3_maskedmm_simple_on_gpu(golden, dev):
    z = golden("k3_maskedmm_simple.npz")
    a8 = tuple(t(z[k], dev) for k in ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c", "indices_c"))
    A = t(z["A"], dev).requires_grad_(True); B = t(z["B"], dev).requires_grad_(True)
    y = functions.MaskedMMCSR.apply(*a8, A, B)
    y.backward(t(z["grad"], dev))
    close(y.detach(), t(z["y"])); close(A.grad, t(z["dA"])); close(B.grad, t(z["dB"]))


@pytest.mark.parametrize("h", [1, 2])
def test_k4_function_classes_on_gpu(golden, dev, h):
    """Our autograd classes reproduce what the reference's classes returned (arg order, return
    order, grad routing -- wrapper.py:8-55)."""
    z = golden("k4_function_classes.npz")
    a8 = tuple(t(z[k], dev) for k in ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c", "indices_c"))
    p = "h%d_" % h
    A, B, x, w, Be = (t(z[p + k], dev).requires_grad_(True) for k in ("A", "B", "x", "w", "Be"))
    ge, gn = t(z[p + "ge"], dev), t(z[p + "gn"], dev)
    y = functions.MaskedMMCSR.apply(*a8, A, B); y.backward(ge)
    close(y.detach(), t(z[p + "mm_y"])); close(A.grad, t(z[p + "mm_dA"])); close(B.grad, t(z[p + "mm_dB"]))
    A.grad = None
    y = functions.SparseSoftmax.apply(*a8[:3], x); y.backward(ge)
    close(y.detach(), t(z[p + "sm_y"])); close(x.grad, t(z[p + "sm_dx"]))
    x.grad = None
    y = functions.SparseSoftmax.apply(*a8[4:7], x); y.backward(ge)
    close(y.detach(), t(z[p + "smg_y"])); close(x.grad, t(z[p + "smg_dx"]))
    y = functions.VectorSPMM.apply(*a8, w, A); y.backward(gn)
    close(y.detach(), t(z[p + "sp_y"])); close(w.grad, t(z[p + "sp_dw"])); close(A.grad, t(z[p + "sp_dx"]))
    A.grad = None
    y = functions.NodeMulEdge.apply(*a8[:3], A, Be); y.backward(ge)
    close(y.detach(), t(z[p + "ne_y"])); close(A.grad, t(z[p + "ne_dA"])); close(Be.grad, t(z[p + "ne_dB"]))


# ---- HIP vs oracle on irregular graphs -----------------------------------------------------------
CONFIGS = [(1, 64), (1, 16), (1, 32), (1, 128), (1, 256), (1, 512), (1, 1024),   # fast h = 1
           (8, 64), (8, 16), (2, 32), (4, 4), (8, 128), (16, 8),                 # fast multi-head
           (1, 7), (3, 5), (2, 6), (1, 20), (1, 2048)]                           # generic


@pytest.mark.parametrize("h,d", CONFIGS)
def test_step_vs_oracle_irregular(dev, h, d):
    n = 90 if h * d >= 512 else 300
    g = random_graph(n, n + 37, 12 * n, seed=h * 1000 + d, chunk_size=32, zero_rows=0.15, hub=700)
    inp = rand_inputs(g, h, d, seed=3, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    got = hip_step(gd, *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])


@pytest.mark.parametrize("h,d", [(1, 64), (4, 16), (3, 5)])
def test_step_vs_oracle_fp64(dev, h, d):
    g = random_graph(120, 150, 2000, seed=d, chunk_size=8, zero_rows=0.1, hub=300)
    inp = rand_inputs(g, h, d, seed=4, dtype=torch.float64, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k], torch.float64)


@pytest.mark.parametrize("chunk_size", [1, 5, 32, 1000])
def test_chunk_sizes(dev, chunk_size):
    g = random_graph(200, 200, 6000, seed=chunk_size, chunk_size=chunk_size, zero_rows=0.1, hub=400)
    inp = rand_inputs(g, 1, 64, seed=5)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])


@pytest.mark.parametrize("h,d", [(1, 64), (8, 16), (3, 5)])
def test_unordered_chunks_take_the_general_path(dev, h, d):
    """Chunks of a row need not be adjacent: shuffle the chunk list (row[] unsorted).  The plan
    reports row_owned = 0 and every op falls back to the atomics path; results still match."""
    g = random_graph(150, 150, 5000, seed=11, chunk_size=8, hub=300)
    inp = rand_inputs(g, h, d, seed=6, normal=True)
    gen = torch.Generator().manual_seed(0)

    def shuffled(row, ptr):
        C = row.numel()
        perm = torch.randperm(C, generator=gen)
        # a chunked CSR needs contiguous [ptr[c], ptr[c+1]) slots, so reorder the slots too
        lens = (ptr[1:] - ptr[:-1])[perm]
        new_ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
        slot = torch.cat([torch.arange(int(ptr[c]), int(ptr[c + 1])) for c in perm.tolist()])
        return row[perm].contiguous(), new_ptr, slot

    row, ptr_r, sr = shuffled(g.row, g.ptr_r)
    col, ptr_c, sc = shuffled(g.col, g.ptr_c)
    a8 = (row, ptr_r, g.eid_r[sr].contiguous(), g.indices_r[sr].contiguous(),
          col, ptr_c, g.eid_c[sc].contiguous(), g.indices_c[sc].contiguous())
    Q, K, V, dO, x, ge = (inp[k] for k in ("Q", "K", "V", "dO", "x", "ge"))
    a8d = tuple(v.to(dev) for v in a8)
    plan = _lib.get_plan(*a8d[:4], n_index_bound=g.n_dst)
    assert plan.info.row_owned == 0 and plan.info.rows_sorted == 0
    close(ops.maskedmm_csr_forward(*a8d[:4], Q.to(dev), K.to(dev)), oracle.maskedmm_csr_forward(*a8[:4], Q, K))
    y = ops.sparse_softmax_forward(*a8d[:3], x.to(dev))
    yo = oracle.sparse_softmax_forward(*a8[:3], x)
    close(y, yo)
    close(ops.sparse_softmax_backward(*a8d[:3], y, ge.to(dev)), oracle.sparse_softmax_backward(*a8[:3], yo, ge))
    close(ops.vector_spmm_forward(*a8d[:4], x.to(dev), V.to(dev)), oracle.vector_spmm_forward(*a8[:4], x, V))
    dw, dx = ops.vector_spmm_backward(*a8d, x.to(dev), dO.to(dev), V.to(dev))
    dwo, dxo = oracle.vector_spmm_backward(*a8, x, dO, V)
    close(dw, dwo); close(dx, dxo)
    dA, dB = ops.maskedmm_csr_backward(*a8d, Q.to(dev), K.to(dev), ge.to(dev))
    dAo, dBo = oracle.maskedmm_csr_backward(*a8, Q, K, ge)
    close(dA, dAo); close(dB, dBo)


def test_partial_coverage_leaves_zeros(dev):
    """Slots no chunk covers read 0 (outputs are at::zeros in the reference, :284,429)."""
    g = graphs.uniform_random_graph(50, 800, seed=9, chunk_size=4)
    inp = rand_inputs(g, 1, 64, seed=1)
    C = g.row.numel() // 2
    row, ptr = g.row[:C].contiguous(), g.ptr_r[: C + 1].contiguous()
    yo = oracle.maskedmm_csr_forward(row, ptr, g.eid_r, g.indices_r, inp["Q"], inp["K"])
    y = ops.maskedmm_csr_forward(row.to(dev), ptr.to(dev), g.eid_r.to(dev), g.indices_r.to(dev),
                                 inp["Q"].to(dev), inp["K"].to(dev))
    close(y, yo)
    assert (y[int(ptr[-1]):] == 0).all()
    so = oracle.sparse_softmax_forward(row, ptr, g.eid_r, inp["x"])
    s = ops.sparse_softmax_forward(row.to(dev), ptr.to(dev), g.eid_r.to(dev), inp["x"].to(dev))
    close(s, so)
    assert (s[int(ptr[-1]):] == 0).all()


def test_empty_graphs(dev):
    z = torch.zeros(0, dtype=torch.int64, device=dev)
    ptr = torch.zeros(1, dtype=torch.int64, device=dev)
    A = torch.rand(5, 64, device=dev)
    y = ops.maskedmm_csr_forward(z, ptr, z, z, A, A)
    assert y.shape == (0,)
    dA, dB = ops.maskedmm_csr_backward(z, ptr, z, z, z, ptr, z, z, A, A, y)
    assert (dA == 0).all() and (dB == 0).all() and dA.shape == A.shape
    assert ops.sparse_softmax_forward(z, ptr, z, y).shape == (0,)
    o = ops.vector_spmm_forward(z, ptr, z, z, y, A)
    assert (o == 0).all() and o.shape == A.shape
    dw, dx = ops.vector_spmm_backward(z, ptr, z, z, z, ptr, z, z, y, A, A)
    assert dw.shape == (0,) and (dx == 0).all()


def test_softmax_floor_and_large_values(dev):
    g = graphs.graph_from_coo(torch.tensor([0, 0, 1, 1]), torch.tensor([0, 1, 1, 0]), 2).to(dev)
    x = torch.tensor([-2e9, -3e9, 80.0, -80.0], device=dev)
    y = ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, x)
    yo = oracle.sparse_softmax_forward(g.row.cpu(), g.ptr_r.cpu(), g.eid_r.cpu(), x.cpu())
    assert torch.isnan(y[:2]).all() and torch.isnan(yo[:2]).all()      # -1e9 floor, :428
    close(y[2:], yo[2:])
    x64 = torch.tensor([700.0, 699.0, -5.0, 3.0], device=dev, dtype=torch.float64)
    close(ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, x64),
          oracle.sparse_softmax_forward(g.row.cpu(), g.ptr_r.cpu(), g.eid_r.cpu(), x64.cpu()), torch.float64)


def test_error_behaviour_on_gpu(dev):
    g = graphs.uniform_random_graph(20, 100, seed=2).to(dev)
    A = torch.rand(20, 64, device=dev)
    with pytest.raises(RuntimeError, match="A must be contiguous"):            # graphop.cpp:5
        ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, A.t().contiguous().t(), A)
    with pytest.raises(RuntimeError, match="Long"):
        ops.maskedmm_csr_forward(g.row.int(), g.ptr_r, g.eid_r, g.indices_r, A, A)
    with pytest.raises(RuntimeError, match="float32 / float64"):
        ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, A.half(), A.half())
    bad = g.indices_r.clone(); bad[3] = 20                                       # out of range
    with pytest.raises(RuntimeError, match="indices value"):
        ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, bad, A, A)
    with pytest.raises(RuntimeError, match="outside|only"):
        ops.vector_spmm_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, torch.rand(100, device=dev), A[:10].contiguous())


def test_out_of_range_row_ids_raise(dev):
    """Row ids index the row-side operand / output: an operand with too few rows is an error, not an
    out-of-bounds atomic on the device (every op; the reference reads / writes out of bounds)."""
    g = graphs.uniform_random_graph(20, 100, seed=2).to(dev)
    A = torch.rand(20, 16, device=dev); short = A[:12].contiguous()
    e = torch.rand(100, device=dev); Be = torch.rand(100, 16, device=dev)
    a4, a8 = (g.row, g.ptr_r, g.eid_r, g.indices_r), g.csr_args()
    with pytest.raises(RuntimeError, match="row id"):
        ops.maskedmm_csr_forward(*a4, short, A)
    with pytest.raises(RuntimeError, match="row id|neighbour id|indices"):
        ops.maskedmm_csr_backward(*a8, short, A, e)
    with pytest.raises(RuntimeError, match="row id|neighbour id|indices"):
        ops.maskedmm_csr_backward(*a8, A, short, e)
    with pytest.raises(RuntimeError, match="row id"):
        ops.node_mul_edge_forward(g.row, g.ptr_r, g.eid_r, short, Be)
    with pytest.raises(RuntimeError, match="row id"):
        ops.node_mul_edge_backward(g.row, g.ptr_r, g.eid_r, short, Be, e)
    with pytest.raises(RuntimeError, match="row id|neighbour id|indices"):
        ops.vector_spmm_backward(*a8, e, short, A)
    with pytest.raises(RuntimeError, match="row id|only|indices"):
        ops.vector_spmm_backward(*a8, e, A, short)


def test_interleaved_empty_chunks(dev):
    """Hand-built layouts with empty chunks between the chunks of a row: the sorted-ids check looks
    past them (unsorted rows must not reach the window drivers) and block detection skips them."""
    src = torch.tensor([0, 0, 0, 0, 1, 1, 2, 2, 2]); dst = torch.tensor([5, 1, 3, 2, 0, 4, 2, 2, 5])   # row 0 NOT sorted by id
    n = 6
    order = torch.argsort(src, stable=True)
    src, dst = src[order], dst[order]
    row = torch.tensor([0, 0, 0, 1, 1, 2, 2]); ptr = torch.tensor([0, 2, 2, 4, 6, 6, 6, 9])   # empty chunks inside rows 0, 1, 2
    eid = torch.arange(9)
    Q = torch.rand(3, 16); K = torch.rand(n, 16)
    want = oracle.maskedmm_csr_forward(row, ptr, eid, dst, Q, K)
    _lib.tune("sweep_min_kb", 0); _lib.tune("window_kb", 1); _lib.tune("sweep_min_granule", 0); _lib.clear_plan_cache()
    try:
        got = ops.maskedmm_csr_forward(row.to(dev), ptr.to(dev), eid.to(dev), dst.to(dev), Q.to(dev), K.to(dev))
        w = torch.rand(9)
        o = ops.vector_spmm_forward(row.to(dev), ptr.to(dev), eid.to(dev), dst.to(dev), w.to(dev), K.to(dev))
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()
    close(got, want)
    close(o[:3], oracle.vector_spmm_forward(row, ptr, eid, dst, w, K)[:3])


def test_torch_ops_namespace_on_gpu(dev):
    g = graphs.uniform_random_graph(64, 2000, seed=3).to(dev)
    A = torch.rand(64, 64, device=dev); B = torch.rand(64, 64, device=dev)
    y1 = torch.ops.graphop.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, A, B)
    y2 = ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, A, B)
    assert torch.equal(y1, y2)
    out = torch.ops.graphop.vector_spmm_backward(*g.csr_args(), y1, A, B)
    assert isinstance(out, (list, tuple)) and len(out) == 2


def test_node_mul_edge_vs_oracle(dev):
    # streaming fast paths (d in 16..256, h in 1,2,4,8) and generic fallbacks (odd d, h = 3, 16)
    for h, d in ((1, 64), (8, 64), (2, 5), (1, 16), (4, 32), (2, 128), (8, 256), (1, 256), (3, 64), (16, 16),
                 (1, 512)):
        g = random_graph(100, 100, 3000, seed=h + d, chunk_size=32, zero_rows=0.1, hub=200)
        gen = torch.Generator().manual_seed(1)
        A = torch.rand((100, d) if h == 1 else (100, h, d), generator=gen)
        Be = torch.rand(g.n_edges, d, generator=gen)
        ge = torch.rand((g.n_edges,) if h == 1 else (g.n_edges, h), generator=gen)
        a3 = (g.row, g.ptr_r, g.eid_r)
        a3d = tuple(v.to(dev) for v in a3)
        close(ops.node_mul_edge_forward(*a3d, A.to(dev), Be.to(dev)), oracle.node_mul_edge_forward(*a3, A, Be))
        dA, dB = ops.node_mul_edge_backward(*a3d, A.to(dev), Be.to(dev), ge.to(dev))
        dAo, dBo = oracle.node_mul_edge_backward(*a3, A, Be, ge)
        close(dA, dAo); close(dB, dBo)


# ---- medium size vs oracle, and full-size properties ----------------------------------------------
def test_medium_powerlaw_vs_oracle(dev):
    """~1M edges, power-law degrees (hubs of thousands of edges), d = 64: the oracle takes seconds."""
    g = graphs.chung_lu_graph(20000, 1000000, alpha=0.6, seed=0)
    inp = rand_inputs(g, 1, 64, seed=7, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k], rtol=2e-4, atol=2e-5)


def test_reddit_scale_properties(dev):
    """BASELINE config 2 size (N = 232,965, E = 114,615,892, d = 64): size-independent checks.
      * softmax rows sum to 1 and y > 0; backward of a constant upstream gradient is ~0
      * SDDMM of all-ones operands = d on every edge; SpMM with unit weights of all-ones = degree
      * linearity of SDDMM in A; a second run is bit-identical for the atomic-free ops
      * adjointness: <SpMM(w, X), G> = <w, dedata> = <X, dx> ties forward and both backward passes
    """
    N, E = graphs.SHAPES["reddit"]
    g = graphs.chung_lu_graph(N, E, alpha=0.5, seed=0, device=dev)
    d = 64
    gen = torch.Generator(device=dev).manual_seed(1)
    Q = torch.rand(N, d, device=dev, generator=gen); K = torch.rand(N, d, device=dev, generator=gen)
    a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
    ones = torch.ones(N, d, device=dev)
    s1 = ops.maskedmm_csr_forward(*a4, ones, ones)
    assert torch.equal(s1, torch.full_like(s1, float(d)))
    deg = (g.indptr_r[1:] - g.indptr_r[:-1]).float()
    o1 = ops.vector_spmm_forward(*a4, torch.ones(E, device=dev), ones)
    torch.testing.assert_close(o1[:, 0], deg, rtol=1e-5, atol=0)
    del s1, o1, ones
    s = ops.maskedmm_csr_forward(*a4, Q, K)
    assert torch.equal(s, ops.maskedmm_csr_forward(*a4, Q, K))               # deterministic
    s2 = ops.maskedmm_csr_forward(*a4, 2 * Q, K)
    torch.testing.assert_close(s2, 2 * s, rtol=1e-6, atol=0)                  # linear in A
    del s2
    a = ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s)
    assert torch.equal(a, ops.sparse_softmax_forward(g.row, g.ptr_r, g.eid_r, s))
    assert float(a.min()) > 0
    rowsum = torch.zeros(N, device=dev, dtype=torch.float64).index_add_(0, g.src, a.double())
    nz = deg > 0
    torch.testing.assert_close(rowsum[nz], torch.ones_like(rowsum[nz]), rtol=0, atol=1e-5)
    dsm = ops.sparse_softmax_backward(g.row, g.ptr_r, g.eid_r, a, torch.full_like(a, 3.0))
    assert float(dsm.abs().max()) < 1e-4
    del dsm, rowsum
    V = torch.rand(N, d, device=dev, generator=gen); G = torch.rand(N, d, device=dev, generator=gen)
    o = ops.vector_spmm_forward(*a4, a, V)
    da, dV = ops.vector_spmm_backward(*g.csr_args(), a, G, V)
    lhs = (o.double() * G.double()).sum()
    torch.testing.assert_close((a.double() * da.double()).sum(), lhs, rtol=1e-6, atol=0)
    torch.testing.assert_close((V.double() * dV.double()).sum(), lhs, rtol=1e-6, atol=0)
    dQ, dK = ops.maskedmm_csr_backward(*g.csr_args(), Q, K, da)
    ref = (s.double() * da.double()).sum()          # <SDDMM(Q,K), da> = <Q, dQ> = <K, dK>
    torch.testing.assert_close((Q.double() * dQ.double()).sum(), ref, rtol=1e-6, atol=0)
    torch.testing.assert_close((K.double() * dK.double()).sum(), ref, rtol=1e-6, atol=0)


# ---- window-sweep drivers (forced on small graphs through the tuning knobs) ----------------------
@pytest.fixture
def force_sweep():
    _lib.tune("sweep_min_kb", 0); _lib.tune("window_kb", 4); _lib.tune("vrow_t", 64)
    _lib.tune("max_windows", 128); _lib.tune("sweep_min_granule", 0)
    _lib.clear_plan_cache()
    yield
    _lib.tune_reset(); _lib.clear_plan_cache()


@pytest.mark.parametrize("h,d", [(1, 64), (1, 16), (1, 256), (1, 1024), (8, 16), (8, 64), (2, 32)])
def test_sweep_drivers_vs_oracle(dev, force_sweep, h, d):
    """Column-window drivers (plan path; XCDs own windows, waves pull tasks): rows longer than vrow_t are
    cut into pieces merged by atomics, empty rows and empty windows occur, several tasks per wave (tiny grid)."""
    _lib.tune("sweep_bpc", 1 if h * d <= 64 else 4)
    n = 120 if h * d >= 512 else 1500
    g = random_graph(n, n + 41, 10 * n, seed=h * 77 + d, chunk_size=32, zero_rows=0.15, hub=900)
    inp = rand_inputs(g, h, d, seed=8, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])


@pytest.mark.parametrize("staged", [0, 3])
@pytest.mark.parametrize("d", [16, 32, 64, 256, 1024])
def test_id_staging_on_and_off_vs_oracle(dev, force_sweep, d, staged):
    """Window-owner SDDMM / SpMM (row- and column-major) with the plan-time deal and ids staged through
    LDS (knob staged_ids, the default) and with the per-batch id loads (staged_ids = 0): same graph as
    above; strips longer than a segment, empty granules, the padded tail of a strip, two id streams
    in the column-major passes."""
    _lib.tune("staged_ids", staged); _lib.tune("sweep_bpc", 1 if d <= 64 else 4)
    _lib.tune("window_kb", 4 * max(1, d // 64)); _lib.clear_plan_cache()      # a few rows per window at every width
    try:
        n = 120 if d >= 512 else 1500
        g = random_graph(n, n + 41, 10 * n, seed=77 + d, chunk_size=32, zero_rows=0.15, hub=900)
        inp = rand_inputs(g, 1, d, seed=8, normal=True)
        want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
        gd = g.to(dev)
        args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]
        got = hip_step(gd, *args)
        for k in ("s", "a", "o", "dQ", "dK", "dV"):
            close(got[k], want[k])
        _lib.profile_enable(True)
        hip_step(gd, *args)
        torch.cuda.synchronize()
        kernels = {r.get("kernel") for r in _lib.profile_read().values()}
        _lib.profile_enable(False)
        if staged:
            assert "k_sddmm_wown_staged_f32" in kernels, kernels
            assert "k_spmm_wown_staged_f32" in kernels, kernels
        else:
            assert {"k_sddmm_wown_f32", "k_spmm_wown_f32"} <= kernels, kernels
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()


@pytest.mark.parametrize("h,d", [(8, 32), (4, 16), (2, 32), (8, 16), (4, 64), (2, 128), (2, 64), (4, 32)])
def test_staged_sddmm_several_heads_vs_oracle(dev, force_sweep, h, d):
    """Window-owner SDDMM-type passes with several heads on the dealt layout (kernels_fast.h:
    sddmm_strip_staged_heads): a head = 4 / 8 / 16 / 32 lanes of the row's lane group; the 16 x h results of a
    batch leave in 16 / min(lanes per head, 16) store instructions.  Row- and column-orientation consumers of
    the scores (softmax, SpMM) check every (edge, head) position."""
    _lib.tune("sweep_bpc", 1 if h * d <= 64 else 3)
    _lib.tune("window_kb", 4 * max(1, h * d // 64)); _lib.clear_plan_cache()
    n = 700 if h * d >= 256 else 1500
    g = random_graph(n, n + 41, 10 * n, seed=h * 7 + d, chunk_size=32, zero_rows=0.15, hub=900)
    inp = rand_inputs(g, h, d, seed=8, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    args = [inp[k].to(dev) for k in ("Q", "K", "V", "dO")]
    got = hip_step(gd, *args)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        close(got[k], want[k])
    _lib.profile_enable(True)
    hip_step(gd, *args)
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    assert prof["sddmm_fwd"]["kernel"] == "k_sddmm_wown_staged_f32", prof["sddmm_fwd"]
    assert prof["spmm_bwd_dedata"]["kernel"] == "k_sddmm_wown_staged_f32", prof["spmm_bwd_dedata"]


@pytest.mark.parametrize("k", [1, 4])
@pytest.mark.parametrize("deg", [15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200])
def test_staged_strips_at_segment_boundaries(dev, force_sweep, deg, k):
    """Strips whose slot count sits on, just below and just above the staging boundaries (half a
    segment = 64, a segment = 128, a batch = 16): every row has exactly `deg` neighbours in window 0
    and deg - 1 in window 1, rows kept whole, k rows per lane group -> strips of k*deg and k*(deg-1)
    slots; SDDMM, both SpMM orientations and the fused backward, vs the oracle."""
    n, n_cols = 96, 512
    gen = torch.Generator().manual_seed(deg * 7 + k)
    src, dst = [], []
    for i in range(n):
        for lo, cnt in ((0, deg), (256, deg - 1)):
            src.append(torch.full((cnt,), i, dtype=torch.int64))
            dst.append(lo + torch.randperm(256, generator=gen)[:cnt].sort().values)
    g = graphs.graph_from_coo(torch.cat(src), torch.cat(dst), n_cols, n_cols, chunk_size=32)
    _lib.tune("window_kb", 64); _lib.tune("vrow_t", 4096); _lib.tune("sweep_k", k); _lib.tune("attn_k", k)
    _lib.tune("attn_window_scale", 1); _lib.tune("staged_ids", 7); _lib.clear_plan_cache()     # 256 rows of 256 B per window
    try:
        inp = rand_inputs(g, 1, 64, seed=deg, normal=True)
        want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
        gd = g.to(dev)
        args = [inp[x].to(dev) for x in ("Q", "K", "V", "dO")]
        got = hip_step(gd, *args)
        for key in ("s", "a", "o", "dQ", "dK", "dV"):
            close(got[key], want[key])
        _lib.profile_enable(True)
        hip_step(gd, *args)
        torch.cuda.synchronize()
        kernels = {r.get("kernel") for r in _lib.profile_read().values()}
        _lib.profile_enable(False)
        assert {"k_sddmm_wown_staged_f32", "k_spmm_wown_staged_f32"} <= kernels, kernels
        q, kk, v = (t.clone().requires_grad_(True) for t in args[:3])
        functions.fused_attention_step(gd, q, kk, v, args[3])
        for key, grad in (("dQ", q.grad), ("dK", kk.grad), ("dV", v.grad)):
            close(grad, want[key])
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()


def test_sweep_matches_chunk_driver_medium(dev, force_sweep):
    """Same inputs through both drivers: equal within fp32 re-association."""
    g = graphs.chung_lu_graph(20000, 1000000, alpha=0.6, seed=3).to(dev)
    gen = torch.Generator(device=dev).manual_seed(2)
    Q, K, V, dO = (torch.rand(20000, 64, device=dev, generator=gen) for _ in range(4))
    _lib.tune("window_kb", 256)
    sw = hip_step(g, Q, K, V, dO)
    _lib.tune("sweep", 0)
    try:
        ch = hip_step(g, Q, K, V, dO)
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()
    # (the strips reduce a batch of 16 dots by a transpose-reduce, the chunk driver slot by slot:
    # same products, different summation tree)
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        torch.testing.assert_close(sw[k], ch[k], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("h", [1, 4])
def test_softmax_long_rows_and_cached_rows(dev, h):
    """Hub rows above the long-row threshold (8192 slots) take the workgroup-per-row kernel, rows
    up to G*8 items the register-cached single pass, the rest the two-sweep loop."""
    g = random_graph(400, 400, 30000, seed=5 + h, chunk_size=32, zero_rows=0.1, hub=20000)
    gen = torch.Generator().manual_seed(3)
    es = (g.n_edges,) if h == 1 else (g.n_edges, h)
    x = torch.randn(es, generator=gen) * 3
    ge = torch.randn(es, generator=gen)
    for a3 in ((g.row, g.ptr_r, g.eid_r), (g.col, g.ptr_c, g.eid_c)):
        a3d = tuple(v.to(dev) for v in a3)
        plan = _lib.get_plan(*a3d)
        yo = oracle.sparse_softmax_forward(*a3, x)
        y = ops.sparse_softmax_forward(*a3d, x.to(dev))
        close(y, yo)
        close(ops.sparse_softmax_backward(*a3d, y, ge.to(dev)), oracle.sparse_softmax_backward(*a3, yo, ge),
              rtol=1e-3, atol=1e-6)
    assert _lib.get_plan(g.row.to(dev), g.ptr_r.to(dev), g.eid_r.to(dev)).info.max_segment_len > 8192  # > 256*32 items: loop path


@pytest.mark.parametrize("h", [4, 8, 16, 64])
def test_softmax_several_heads_float4_items(dev, h):
    """h % 4 == 0 heads in storage order take the float4 kernels (kernels_fast.h: softmax_vec4_group): rows held in
    registers (up to G * 32 / G * 16 float4s forward / backward), rows looped over twice, hub rows one workgroup
    each, empty rows; the column-major orientation (gathered through eid) stays on the scalar kernels."""
    gen = torch.Generator().manual_seed(40 + h)
    lens = torch.cat([torch.randint(0, 40, (300,), generator=gen), torch.randint(250, 400, (20,), generator=gen),
                      torch.randint(700, 1024, (12,), generator=gen), torch.tensor([1025, 2000, 5000, 0, 1, 1024])])
    lens = lens[torch.randperm(len(lens), generator=gen)]
    n = len(lens)
    src = torch.repeat_interleave(torch.arange(n), lens)
    dst = torch.randint(0, n, (int(lens.sum()),), generator=gen)
    g = graphs.graph_from_coo(src, dst, n, n, chunk_size=32)
    x = torch.randn(g.n_edges, h, generator=gen) * 3
    ge = torch.randn(g.n_edges, h, generator=gen)
    for a3, want_kernel in (((g.row, g.ptr_r, g.eid_r), "vec4"), ((g.col, g.ptr_c, g.eid_c), "seg")):
        a3d = tuple(v.to(dev) for v in a3)
        yo = oracle.sparse_softmax_forward(*a3, x)
        _lib.profile_enable(True)
        y = ops.sparse_softmax_forward(*a3d, x.to(dev))
        dx = ops.sparse_softmax_backward(*a3d, y, ge.to(dev))
        torch.cuda.synchronize()
        ran = {r.get("kernel") for r in _lib.profile_read().values()}
        _lib.profile_enable(False)
        assert {k for k in ran if "softmax" in k} == {"k_softmax_fwd_" + want_kernel, "k_softmax_bwd_" + want_kernel}, ran
        close(y, yo)
        close(dx, oracle.sparse_softmax_backward(*a3, yo, ge), rtol=1e-3, atol=1e-6)


def test_misaligned_head_tensors_fall_back(dev):
    """The float4 softmax and the several-heads walk kernel read (E, h) arrays with 16-byte loads; a view that
    starts 4 bytes into its storage takes the scalar forms instead -- same results, no error."""
    h, d, n = 8, 32, 600
    g = random_graph(n, n, 12 * n, seed=91, chunk_size=32, hub=700).to(dev)
    gen = torch.Generator(device=dev).manual_seed(6)
    x = torch.randn(g.n_edges, h, device=dev, generator=gen)
    V = torch.randn(n, h, d, device=dev, generator=gen)
    def shifted(t):
        buf = torch.empty(t.numel() + 1, device=dev, dtype=t.dtype)
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 4 and v.is_contiguous()
        return v
    a3 = (g.row, g.ptr_r, g.eid_r)
    y = ops.sparse_softmax_forward(*a3, x)
    y_s = ops.sparse_softmax_forward(*a3, shifted(x))
    torch.testing.assert_close(y_s, y, rtol=1e-5, atol=1e-7)
    gy = torch.randn(g.n_edges, h, device=dev, generator=gen)
    torch.testing.assert_close(ops.sparse_softmax_backward(*a3, shifted(y), shifted(gy)),
                               ops.sparse_softmax_backward(*a3, y, gy), rtol=1e-5, atol=1e-7)
    _lib.tune("sweep_min_kb", 0); _lib.tune("walk_window_kb", 64); _lib.tune("walk_min_bin", 0)
    _lib.tune("sweep_min_granule", 0); _lib.tune("max_windows", 512); _lib.tune("walk_blocks", 8)
    _lib.clear_plan_cache()
    try:
        a4 = (g.row, g.ptr_r, g.eid_r, g.indices_r)
        _lib.profile_enable(True)
        o = ops.vector_spmm_forward(*a4, y, V)
        torch.cuda.synchronize()
        assert {r.get("kernel") for r in _lib.profile_read().values()} >= {"k_spmm_walk_f32"}
        o_s = ops.vector_spmm_forward(*a4, shifted(y), V)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        torch.testing.assert_close(o_s, o, rtol=1e-4, atol=1e-6)
    finally:
        _lib.profile_enable(False); _lib.tune_reset(); _lib.clear_plan_cache()


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_shapes_and_paths(dev, seed):
    """Randomised battery: shape, heads, chunk size, degree profile, square / non-square, with
    the sweep drivers forced on or off and random pacing / window knobs.  All vs the oracle."""
    rng = np.random.RandomState(1000 + seed)
    h = int(rng.choice([1, 1, 2, 4, 8]))
    d = int(rng.choice([4, 8, 16, 32, 64, 128])) if h > 1 else int(rng.choice([16, 32, 64, 128, 256, 512]))
    while h * d > 1024:
        d //= 2
    n_src = int(rng.randint(40, 700))
    n_dst = n_src if rng.rand() < 0.5 else int(rng.randint(40, 700))
    n_edges = int(rng.randint(1, 40) * n_src)
    cs = int(rng.choice([1, 7, 32, 64]))
    hub = int(rng.choice([0, 0, 300, 1500]))
    forced = bool(rng.rand() < 0.6)
    if forced:
        _lib.tune("sweep_min_kb", 0); _lib.tune("sweep_min_granule", 0); _lib.tune("max_windows", 128)
        _lib.tune("window_kb", int(rng.choice([1, 4, 16]))); _lib.tune("vrow_t", int(rng.choice([0, 64, 256])))
        rng.choice([0, 1, 2, 3]); _lib.tune("sweep_bpc", int(rng.choice([1, 2, 4])))   # (draws of the knobs removed in
        rng.choice([0, 1]); rng.choice([0, 1]); rng.choice([0, 1])                      # round 4 kept: same graphs per seed)
        _lib.tune("staged_ids", int(rng.choice([0, 7, 7]))); _lib.tune("sweep_k", int(rng.choice([0, 0, 1, 2, 4, 8])))
        if rng.rand() < 0.6:     # walk drivers too (one head and several), tiny windows and grids
            rng.choice([6, 7]); _lib.tune("walk", 6); _lib.tune("walk_min_bin", 0)
            _lib.tune("walk_window_kb", int(rng.choice([2, 8, 32]))); _lib.tune("walk_window_kb_col", int(rng.choice([2, 8, 32])))
            _lib.tune("walk_blocks", int(rng.choice([0, 8, 24]))); _lib.tune("walk_drift", int(rng.choice([0, 1, 2, 3])))
            _lib.tune("walk_steps", int(rng.choice([1, 2, 3]))); _lib.tune("max_windows", 512)
        else:
            _lib.tune("walk", 0)
    if np.random.RandomState(7000 + seed).rand() < 0.5:      # (own stream: the draws above keep their graphs per seed)
        _lib.tune("spmm_selfzero_min_mb", 0); _lib.tune("spmm_cpg", int(np.random.RandomState(7100 + seed).choice([1, 4, 16])))
    _lib.tune("spmm_flat_cpg", int(np.random.RandomState(7200 + seed).choice([2, 16, 128])))   # (slot-walking chunk driver)
    _lib.tune("spmm_flat_min_chunks", 0)
    _lib.clear_plan_cache()
    try:
        g = random_graph(n_src, n_dst, n_edges, seed=seed, chunk_size=cs, zero_rows=float(rng.choice([0, 0.2])),
                         hub=hub or None)
        # the reference's SpMM output is zeros_like(x): x (n_dst rows) must cover the row ids
        if g.n_dst < g.n_src:
            g = random_graph(n_src, n_src, n_edges, seed=seed, chunk_size=cs, hub=hub or None)
        inp = rand_inputs(g, h, d, seed=seed + 50, normal=True)
        want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
        got = hip_step(g.to(dev), *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
        for k in ("s", "a", "o", "dQ", "dK", "dV"):
            close(got[k], want[k], rtol=2e-4, atol=2e-5)
    finally:
        _lib.tune_reset()
        _lib.clear_plan_cache()


# ---- block-dense (fp32 MFMA) drivers -------------------------------------------------------------
def _plans(g):
    return (_lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r), _lib.get_plan(g.col, g.ptr_c, g.eid_c, g.indices_c))


@pytest.mark.parametrize("l,h,d", [(30, 1, 1024), (30, 8, 64), (32, 1, 128), (17, 2, 32), (5, 1, 8), (30, 4, 24),
                                   (9, 1, 16)])
def test_block_dense_drivers_vs_oracle(dev, l, h, d):
    """Disjoint complete digraphs (the harness fixture shape, wrapper.py:79-112): the plan finds one
    block per component in BOTH orientations and the MFMA drivers serve all six gather passes
    (d % 32 == 0) or the three SDDMM-shaped ones (d % 8 == 0); same results with them switched off."""
    bs = 7
    g = graphs.block_diagonal_graph(bs, l, chunk_size=32)
    inp = rand_inputs(g, h, d, seed=l + d, normal=True)
    want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
    gd = g.to(dev)
    _lib.clear_plan_cache()
    _lib.tune("dense_detect_min_fill", 1); _lib.tune("dense_min_fill", 1)   # small blocks too
    try:
        for p in _plans(gd):
            assert p.info.n_dense_blocks == bs and p.info.dense_fill_pct == max(1, round(100 * l * l / 1024))
        args = tuple(inp[k].to(dev) for k in ("Q", "K", "V", "dO"))
        got = hip_step(gd, *args)
        for k in ("s", "a", "o", "dQ", "dK", "dV"):
            close(got[k], want[k])
        _lib.tune("dense_blocks", 0)
        ref = hip_step(gd, *args)
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()
    for k in ("s", "a", "o", "dQ", "dK", "dV"):
        torch.testing.assert_close(got[k], ref[k], rtol=1e-4, atol=1e-5)


def test_block_dense_long_runs_and_ragged_blocks(dev):
    """Runs longer than 32 rows are cut into several blocks; blocks of different shapes, rows
    without edges between them, a non-square adjacency and a non-identity eid."""
    src, dst = [], []
    def biclique(rows, cols):
        for r in rows:
            for c in cols:
                src.append(r); dst.append(c)
    biclique(range(0, 50), range(10, 30))        # 50 rows x 20 ids -> blocks of 32 + 18 rows
    biclique(range(60, 61), range(0, 32))        # a single row with 32 ids
    biclique(range(70, 110), range(100, 103))    # 40 rows x 3 ids -> 32 + 8
    biclique(range(110, 120), range(40, 71))     # 10 rows x 31 ids
    n_src, n_dst = 120, 130
    g = graphs.graph_from_coo(torch.tensor(src), torch.tensor(dst), n_src, n_dst, chunk_size=7)
    gd = g.to(dev)
    _lib.clear_plan_cache()
    _lib.tune("dense_min_fill", 1); _lib.tune("dense_detect_min_fill", 1)
    try:
        pr, pc = _plans(gd)
        assert pr.info.n_dense_blocks == 6 and pr.info.max_segment_len == 32
        assert pc.info.max_segment_len > 32 and pc.info.n_dense_blocks == 0    # columns 10..29 have 50 sources
        for h, d in ((1, 64), (2, 32), (1, 8)):
            inp = rand_inputs(g, h, d, seed=3, normal=True)
            want = oracle_step(oracle, g, inp["Q"], inp["K"], inp["V"], inp["dO"])
            got = hip_step(gd, *(inp[k].to(dev) for k in ("Q", "K", "V", "dO")))
            for k in ("s", "a", "o", "dQ", "dK", "dV"):
                close(got[k], want[k])
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()


def test_block_dense_not_selected_on_irregular_graphs(dev):
    g = random_graph(300, 300, 3000, seed=1, chunk_size=32).to(dev)
    _lib.clear_plan_cache()
    for p in _plans(g):
        assert p.info.n_dense_blocks == 0 and p.info.dense_fill_pct == 0


# ---- HIP graph capture ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["irregular", "windowed", "block_dense"])
def test_step_replays_from_a_captured_hip_graph(dev, shape):
    """The whole fwd+bwd step (8 library launches + zero fills) captured once into a HIP graph and
    replayed on new inputs gives what eager execution gives: no op synchronises, allocates outside
    torch's graph pool or touches host state per call once the plans exist."""
    forced = shape == "windowed"
    if forced:
        _lib.tune("sweep_min_kb", 0); _lib.tune("window_kb", 4); _lib.tune("vrow_t", 64); _lib.tune("sweep_min_granule", 0)
    _lib.clear_plan_cache()
    try:
        if shape == "block_dense":
            g = graphs.block_diagonal_graph(6, 30, device=dev)
        else:
            g = random_graph(900, 900, 9000, seed=4, chunk_size=32, zero_rows=0.1, hub=400).to(dev)
        n, d = g.n_src, 64
        gen = torch.Generator(device=dev).manual_seed(0)
        Q, K, V = (torch.randn(n, d, device=dev, generator=gen).requires_grad_(True) for _ in range(3))
        dO = torch.randn(n, d, device=dev, generator=gen)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # warm-up: builds the plans
            for _ in range(2):
                functions.attention_step(g, Q, K, V, dO)
        torch.cuda.current_stream().wait_stream(side)
        for t_ in (Q, K, V):
            t_.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            s, a, o = functions.attention_step(g, Q, K, V, dO)
        outs = (s, a, o, Q.grad, K.grad, V.grad)
        for trial in range(2):
            with torch.no_grad():
                for t_ in (Q, K, V, dO):
                    t_.copy_(torch.randn(n, d, device=dev, generator=gen))
            graph.replay()
            torch.cuda.synchronize()
            got = [x.clone() for x in outs]
            ref = hip_step(g, Q.detach(), K.detach(), V.detach(), dO)
            for x, k in zip(got, ("s", "a", "o", "dQ", "dK", "dV")):
                torch.testing.assert_close(x, ref[k], rtol=1e-4, atol=1e-5)
    finally:
        _lib.tune_reset()
        _lib.clear_plan_cache()


@pytest.mark.parametrize("h,d,cs", [(1, 64, 8), (1, 128, 32), (4, 16, 1), (1, 16, 32), (8, 128, 8)])
def test_chunk_spmm_defines_every_output_row_without_a_zero_fill(dev, h, d, cs):
    """Self-zeroing chunk driver (k_spmm_f32<..., SELFZERO>; csrc/kernels_chunk.h): with sorted rows the SpMM-type
    passes leave EVERY output row defined -- owned rows stored, rows of nodes without edges zero-stored by the group
    that sees the gap, rows cut between two groups zeroed by k_zero_shared_rows and merged by atomics -- so the entry
    points skip the zero fill of large outputs (the extended tensors of the sharded step).  Outputs are handed over
    full of NaNs through the C ABI; forward, dx, dA and dB against the oracle."""
    _lib.tune_reset(); _lib.clear_plan_cache()
    _lib.tune("sweep", 0); _lib.tune("walk", 0); _lib.tune("spmm_selfzero_min_mb", 0); _lib.tune("spmm_cpg", 4)
    try:
        g = random_graph(700, 941, 9000, seed=3 + h + d, chunk_size=cs, zero_rows=0.3, hub=1500)
        inp = rand_inputs(g, h, d, seed=5, normal=True)
        gd = g.to(dev)
        a8 = g.csr_args()
        w = torch.rand(g.n_edges, h) if h > 1 else torch.rand(g.n_edges)
        X = inp["K"]                                              # (n_dst, [h,] d)
        dy = inp["dO"][:g.n_src]
        want_y = oracle.vector_spmm_forward(*a8[:4], w, X)
        want_dw, want_dx = oracle.vector_spmm_backward(*a8, w, torch.cat([dy, torch.zeros(g.n_dst - g.n_src, *dy.shape[1:])]), X)
        want_dA, want_dB = oracle.maskedmm_csr_backward(*a8, inp["Q"], inp["K"], w)
        L = _lib.lib()
        wd, Xd, dyd, Qd = w.to(dev), X.to(dev), dy.to(dev).contiguous(), inp["Q"].to(dev)
        nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
        tail = X.shape[1:]
        with _lib.device_guard(dev):
            pr = _lib.get_plan(gd.row, gd.ptr_r, gd.eid_r, gd.indices_r, g.n_dst)
            pc = _lib.get_plan(gd.col, gd.ptr_c, gd.eid_c, gd.indices_c, g.n_src)
            st = _lib.stream_of(Xd)
            _lib.profile_enable(True)
            y = nan(g.n_src, *tail)                               # n_y = n_src rows (the C ABI takes the row count)
            _lib.check(L.graphop_vector_spmm_forward(_lib.dtype_code(Xd), _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r),
                                                     _lib.ptr(gd.indices_r), _lib.ptr(wd), _lib.ptr(Xd), _lib.ptr(y), gd.row.size(0),
                                                     g.n_edges, g.n_dst, g.n_src, h, d, pr.handle, st))
            dw, dx = nan(*w.shape), nan(g.n_dst, *tail)
            _lib.check(L.graphop_vector_spmm_backward(_lib.dtype_code(Xd), _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r),
                                                      _lib.ptr(gd.indices_r), _lib.ptr(gd.col), _lib.ptr(gd.ptr_c), _lib.ptr(gd.eid_c),
                                                      _lib.ptr(gd.indices_c), _lib.ptr(wd), _lib.ptr(dyd), _lib.ptr(Xd), _lib.ptr(dw),
                                                      _lib.ptr(dx), gd.row.size(0), gd.col.size(0), g.n_edges, g.n_dst, g.n_src, h, d,
                                                      pr.handle, pc.handle, st))
            dA, dB = nan(g.n_src, *tail), nan(g.n_dst, *tail)
            _lib.check(L.graphop_maskedmm_csr_backward(_lib.dtype_code(Xd), _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r),
                                                       _lib.ptr(gd.indices_r), _lib.ptr(gd.col), _lib.ptr(gd.ptr_c), _lib.ptr(gd.eid_c),
                                                       _lib.ptr(gd.indices_c), _lib.ptr(Qd), _lib.ptr(Xd), _lib.ptr(wd), _lib.ptr(dA),
                                                       _lib.ptr(dB), gd.row.size(0), gd.col.size(0), g.n_edges, g.n_src, g.n_dst, h, d,
                                                       pr.handle, pc.handle, st))
            torch.cuda.synchronize()
            prof = _lib.profile_read()
            _lib.profile_enable(False)
        # none of the four node-sized outputs was zero-filled by the host (the E-sized dedata may be: its fill depends on
        # the plan's coverage flags, not on this path)
        assert prof.get("zero_fill", {}).get("calls", 0) <= 1, prof.get("zero_fill")
        for got, want in ((y, want_y[:g.n_src]), (dx, want_dx), (dA, want_dA), (dB, want_dB)):
            assert not torch.isnan(got).any()
            close(got, want)
        close(dw, want_dw)
    finally:
        _lib.profile_enable(False); _lib.tune_reset(); _lib.clear_plan_cache()


def _spmm_fwd_and_dx_through_the_abi(dev, g, gd, w, X, dy, h, d):
    """vector_spmm_forward + backward through the C ABI into NaN-filled outputs; -> (y, dw, dx, profile)."""
    L = _lib.lib()
    wd, Xd, dyd = w.to(dev), X.to(dev), dy.to(dev).contiguous()
    nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
    tail = X.shape[1:]
    with _lib.device_guard(dev):
        pr = _lib.get_plan(gd.row, gd.ptr_r, gd.eid_r, gd.indices_r, g.n_dst)
        pc = _lib.get_plan(gd.col, gd.ptr_c, gd.eid_c, gd.indices_c, g.n_src)
        st = _lib.stream_of(Xd)
        _lib.profile_enable(True)
        y = nan(g.n_src, *tail)
        _lib.check(L.graphop_vector_spmm_forward(_lib.dtype_code(Xd), _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r),
                                                 _lib.ptr(gd.indices_r), _lib.ptr(wd), _lib.ptr(Xd), _lib.ptr(y), gd.row.size(0),
                                                 g.n_edges, g.n_dst, g.n_src, h, d, pr.handle, st))
        dw, dx = nan(*w.shape), nan(g.n_dst, *tail)
        _lib.check(L.graphop_vector_spmm_backward(_lib.dtype_code(Xd), _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r),
                                                  _lib.ptr(gd.indices_r), _lib.ptr(gd.col), _lib.ptr(gd.ptr_c), _lib.ptr(gd.eid_c),
                                                  _lib.ptr(gd.indices_c), _lib.ptr(wd), _lib.ptr(dyd), _lib.ptr(Xd), _lib.ptr(dw),
                                                  _lib.ptr(dx), gd.row.size(0), gd.col.size(0), g.n_edges, g.n_dst, g.n_src, h, d,
                                                  pr.handle, pc.handle, st))
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    return y, dw, dx, prof


@pytest.mark.parametrize("selfzero", [1, 0])
@pytest.mark.parametrize("d,cs,fcpg", [(64, 1, 3), (64, 32, 40), (128, 2, 150), (128, 32, 3), (256, 1, 70), (256, 32, 128),
                                       (128, 1, 33)])
def test_flat_chunk_spmm_short_rows_vs_oracle(dev, d, cs, fcpg, selfzero):
    """Slot-walking form of the row-owning chunk driver (k_spmm_flat_f32, csrc/kernels_chunk.h; chosen below 10 slots per
    chunk: the extended column side of a node-range shard).  Rows of 0-3 slots, a third of the rows empty, one hub row
    cut into many chunks (and, at small chunks-per-group, between many lane groups: the atomic merge of shared rows),
    chunks-per-group below and above the 2 L chunks of metadata a group keeps in registers; with and without the
    self-zeroing rule (outputs arrive full of NaNs either way).  Semantics: graphop_kernel.cu:118-130 (forward),
    :135-163 (backward)."""
    _lib.tune_reset(); _lib.clear_plan_cache()
    _lib.tune("sweep", 0); _lib.tune("walk", 0); _lib.tune("dense_blocks", 0)
    _lib.tune("spmm_selfzero_min_mb", 0 if selfzero else 1 << 20); _lib.tune("spmm_flat_cpg", fcpg)
    _lib.tune("spmm_flat_min_chunks", 0)
    try:
        g = random_graph(2500, 3000, 6000, seed=11 + d + cs, chunk_size=cs, zero_rows=0.3, hub=300)
        assert g.n_edges < 10 * g.row.numel()                      # the predicate of the flat form
        inp = rand_inputs(g, 1, d, seed=5, normal=True)
        w, X, dy = torch.rand(g.n_edges), inp["K"], inp["dO"][:g.n_src]
        a8 = g.csr_args()
        want_y = oracle.vector_spmm_forward(*a8[:4], w, X)[:g.n_src]
        want_dw, want_dx = oracle.vector_spmm_backward(*a8, w, torch.cat([dy, torch.zeros(g.n_dst - g.n_src, d)]), X)
        y, dw, dx, prof = _spmm_fwd_and_dx_through_the_abi(dev, g, g.to(dev), w, X, dy, 1, d)
        assert prof["spmm_fwd"]["kernel"] == "k_spmm_flat_f32" and prof["spmm_bwd_dx"]["kernel"] == "k_spmm_flat_f32", prof
        for got, want in ((y, want_y), (dx, want_dx), (dw, want_dw)):
            assert not torch.isnan(got).any()
            close(got, want)
        _lib.tune("spmm_flat", 0)                                  # and the per-chunk loop it replaces, same inputs
        _lib.clear_plan_cache()
        y2, dw2, dx2, prof2 = _spmm_fwd_and_dx_through_the_abi(dev, g, g.to(dev), w, X, dy, 1, d)
        assert prof2["spmm_fwd"]["kernel"] == "k_spmm_f32", prof2
        torch.testing.assert_close(y, y2, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(dx, dx2, rtol=1e-4, atol=1e-5)
    finally:
        _lib.profile_enable(False); _lib.tune_reset(); _lib.clear_plan_cache()


@pytest.mark.parametrize("fcpg", [1, 2, 5, 64])
def test_flat_chunk_spmm_empty_chunks_and_gaps(dev, fcpg):
    """A hand-made chunk list for the slot-walking driver: chunks without slots at the start, in the middle and at the end
    of the list (the reference's kernel simply runs zero iterations for them, graphop_kernel.cu:121), rows without
    chunks between them, a row in three chunks.  Self-zeroing on: every output row must come back defined."""
    row = torch.tensor([0, 0, 2, 2, 2, 3, 6, 6, 9, 9])
    ptr = torch.tensor([0, 0, 2, 2, 5, 6, 9, 9, 9, 12, 12])
    n_rows, n_cols, E, d = 12, 7, 12, 64
    idx = torch.tensor([1, 3, 0, 6, 6, 2, 5, 4, 1, 0, 3, 3])
    eid = torch.randperm(E, generator=torch.Generator().manual_seed(3))
    w = torch.rand(E, generator=torch.Generator().manual_seed(4))
    X = torch.randn(n_cols, d, generator=torch.Generator().manual_seed(5))
    want = oracle.vector_spmm_forward(row, ptr, eid, idx, w, torch.cat([X, torch.zeros(n_rows - n_cols, d)]))[:n_rows]
    _lib.tune_reset(); _lib.clear_plan_cache()
    _lib.tune("spmm_selfzero_min_mb", 0); _lib.tune("spmm_flat_cpg", fcpg); _lib.tune("spmm_flat_max_mean", 100)
    _lib.tune("spmm_flat_min_chunks", 0)
    _lib.tune("dense_blocks", 0)
    try:
        L = _lib.lib()
        rd, pd, ed, idd, wd, Xd = (t.to(dev) for t in (row, ptr, eid, idx, w, X))
        y = torch.full((n_rows, d), float("nan"), device=dev)
        with _lib.device_guard(dev):
            plan = _lib.get_plan(rd, pd, ed, idd, n_cols)
            _lib.profile_enable(True)
            _lib.check(L.graphop_vector_spmm_forward(_lib.dtype_code(Xd), _lib.ptr(rd), _lib.ptr(pd), _lib.ptr(ed), _lib.ptr(idd),
                                                     _lib.ptr(wd), _lib.ptr(Xd), _lib.ptr(y), row.numel(), E, n_cols, n_rows, 1, d,
                                                     plan.handle, _lib.stream_of(Xd)))
            torch.cuda.synchronize()
            prof = _lib.profile_read()
        assert prof["spmm_fwd"]["kernel"] == "k_spmm_flat_f32", prof
        assert not torch.isnan(y).any()
        close(y, want)
    finally:
        _lib.profile_enable(False); _lib.tune_reset(); _lib.clear_plan_cache()


def test_selfzero_leaves_a_long_empty_tail_and_long_gaps_to_the_fill(dev):
    """Round-4 advice: the self-zeroing chunk driver zero-stores edge-less rows with ONE lane group per gap.  The rows
    behind the last chunk row are now zeroed from the host (one device-wide fill of the tail), and a plan whose longest
    interior gap exceeds 4 MB of rows does not take the self-zeroing form at all (graphop_plan_info_t.max_row_gap)."""
    _lib.tune_reset(); _lib.clear_plan_cache()
    _lib.tune("sweep", 0); _lib.tune("walk", 0); _lib.tune("spmm_selfzero_min_mb", 0)
    try:
        d = 64
        for n_rows, lo, hi, expect_selfzero in ((60000, 0, 300, True),          # 59,700 trailing rows without edges
                                                (60000, 59000, 60000, False)):  # a 59,000-row gap in FRONT (15 MB of rows)
            gen = torch.Generator().manual_seed(n_rows + lo)
            src = torch.randint(lo, hi, (4000,), generator=gen)
            dst = torch.randint(0, 500, (4000,), generator=gen)
            g = graphs.graph_from_coo(src, dst, n_rows, 500, 8)
            gd = g.to(dev)
            w, X = torch.rand(g.n_edges), torch.randn(500, d, generator=gen)
            # (the oracle keeps the reference's y = zeros_like(x): hand it a table padded to n_rows rows)
            want = oracle.vector_spmm_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, w, torch.cat([X, torch.zeros(n_rows - 500, d)]))
            y = torch.full((n_rows, d), float("nan"), device=dev)
            wd, Xd = w.to(dev), X.to(dev)
            with _lib.device_guard(dev):
                pr = _lib.get_plan(gd.row, gd.ptr_r, gd.eid_r, gd.indices_r, 500)
                u = torch.unique(src)
                assert pr.info.max_row_gap == max(int(u[0]), int((u[1:] - u[:-1]).max()) - 1)
                _lib.profile_enable(True)
                _lib.check(_lib.lib().graphop_vector_spmm_forward(
                    _lib.F32, _lib.ptr(gd.row), _lib.ptr(gd.ptr_r), _lib.ptr(gd.eid_r), _lib.ptr(gd.indices_r),
                    _lib.ptr(wd), _lib.ptr(Xd), _lib.ptr(y), gd.row.size(0), g.n_edges, 500, n_rows, 1, d,
                    pr.handle, _lib.stream_of(y)))
                torch.cuda.synchronize()
                prof = _lib.profile_read()
                _lib.profile_enable(False)
            assert not torch.isnan(y).any()
            close(y, want)
            fills = prof.get("zero_fill", {}).get("calls", 0)
            assert fills == 1, prof        # the tail fill (self-zeroing) or the whole-output fill (long gap): one launch either way
            _lib.clear_plan_cache()
    finally:
        _lib.profile_enable(False); _lib.tune_reset(); _lib.clear_plan_cache()


def test_seventeenth_window_geometry_is_counted_not_silent(dev, capfd):
    """A plan keeps at most 16 window geometries; the pass that asks for one more runs on the chunk drivers -- correct, and
    (round-4 verdict) no longer invisible: graphop_plan_info_t.n_geometry_fallbacks counts such passes and the first one
    warns on stderr."""
    _lib.tune_reset(); _lib.clear_plan_cache()
    _lib.tune("sweep_min_kb", 0); _lib.tune("sweep_min_granule", 0); _lib.tune("walk", 0); _lib.tune("vrow_t", 64)
    try:
        g = random_graph(1200, 1200, 40000, seed=8, chunk_size=32, hub=500).to(dev)
        gen = torch.Generator(device=dev).manual_seed(2)
        Q, K = (torch.randn(1200, 64, device=dev, generator=gen) for _ in range(2))
        want = None
        for w in range(2, 20):
            _lib.tune("sweep_w", w)
            s = ops.maskedmm_csr_forward(g.row, g.ptr_r, g.eid_r, g.indices_r, Q, K)
            want = s if want is None else want
            torch.testing.assert_close(s, want, rtol=1e-4, atol=1e-5)      # (window strips and chunk driver sum in different orders)
        plan = _lib.get_plan(g.row, g.ptr_r, g.eid_r, g.indices_r, 1200)
        assert plan.refresh_info().n_geometry_fallbacks >= 2        # W = 18 and W = 19 found the plan full
        assert "already holds 16 window geometries" in capfd.readouterr().err
    finally:
        _lib.tune_reset(); _lib.clear_plan_cache()
