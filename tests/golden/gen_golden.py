"""Generate the golden fixtures in tests/golden/*.npz.

RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).  It imports the reference's own
Python -- ``part_csr.partition_csr`` and ``wrapper`` (``MaskedMMSimple`` and the four
``autograd.Function`` classes) -- with ``PYTHONDONTWRITEBYTECODE=1`` (the reference tree is
read-only) and records inputs + expected outputs as small arrays.  Nothing from the reference
(source or bytecode) is copied: the fixtures are data.

  K1  partition_csr known answers, captured from the real ``part_csr.partition_csr``
      (part_csr.py:13-27; incl. the case in its __main__, :30-31).
  K2  the harness fixture shrunk to bs=3, l=4 (built like wrapper.py:93-112), chunk_size 3 and
      32, h in {1, 8}: expected values from the SAME stock-PyTorch formulations the reference
      asserts against: bmm (wrapper.py:185,364), th.softmax dims -1/-2 (:218,245,395,422),
      th.sparse.mm + autograd (:274-283,459).
  K3  ``wrapper.MaskedMMSimple.apply`` outputs and gradients on an irregular graph (zero-degree
      rows, degree > chunk_size).
  K4  the reference's own Function classes (wrapper.py:8-55) driven with the CPU oracle injected
      as the ``graphop`` module: pins .apply argument order, return order and gradient routing.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

import oracle  # noqa: E402  (CPU restatement; injected below as the reference's `graphop`)
from custom_op_benchmark_amd import graphs  # noqa: E402


def import_reference():
    mod = types.ModuleType("graphop")
    for name in oracle.EXPORTS:
        setattr(mod, name, getattr(oracle, name))
    mod.__all__ = list(oracle.EXPORTS)
    saved = sys.modules.get("graphop")
    sys.modules["graphop"] = mod            # `from graphop import *` in wrapper.py:2 binds these
    sys.path.insert(0, REF)
    try:
        import part_csr as ref_part
        import wrapper as ref_wrapper
    finally:
        sys.path.remove(REF)
        if saved is not None:
            sys.modules["graphop"] = saved
        else:
            del sys.modules["graphop"]
    return ref_part, ref_wrapper


def np_(t):
    return t.detach().cpu().numpy()


def k1_partition(ref_part):
    cases = [([0, 4, 8, 10], 4), ([0, 4, 8, 10], 32), ([0, 4, 8, 10], 3), ([0, 0, 5, 5, 7], 2),
             ([0], 4), ([0, 0, 0], 4), ([0, 1, 2, 3], 1), ([0, 70, 70, 135, 136], 32),
             ([3, 9, 9, 40], 8)]
    g = th.Generator().manual_seed(7)
    deg = th.randint(0, 100, (200,), generator=g)
    deg[th.randint(0, 200, (40,), generator=g)] = 0
    cases.append((th.cat([th.zeros(1, dtype=th.long), th.cumsum(deg, 0)]).tolist(), 32))
    cases.append((th.cat([th.zeros(1, dtype=th.long), th.cumsum(deg, 0)]).tolist(), 7))
    out = {"n_cases": np.int64(len(cases))}
    for i, (ip, cs) in enumerate(cases):
        row, ptr = ref_part.partition_csr(th.tensor(ip, dtype=th.long), chunk_size=cs)
        out["c%d_indptr" % i] = np.asarray(ip, dtype=np.int64)
        out["c%d_chunk" % i] = np.int64(cs)
        out["c%d_row" % i] = np_(row).astype(np.int64).reshape(-1)
        out["c%d_ptr" % i] = np_(ptr).astype(np.int64).reshape(-1)
    np.savez_compressed(os.path.join(HERE, "k1_partition_csr.npz"), **out)
    print("K1: %d partition_csr cases" % len(cases))


def k2_harness(ref_part):
    bs, l = 3, 4
    n, e = bs * l, bs * l * l
    out = {"bs": np.int64(bs), "l": np.int64(l)}
    for cs in (3, 32):
        g = graphs.block_diagonal_graph(bs, l, chunk_size=cs)
        # the chunk arrays must be what the reference's chunker gives for these indptrs
        r_ref, p_ref = ref_part.partition_csr(g.indptr_r, chunk_size=cs)
        c_ref, q_ref = ref_part.partition_csr(g.indptr_c, chunk_size=cs)
        assert th.equal(r_ref, g.row) and th.equal(p_ref, g.ptr_r)
        assert th.equal(c_ref, g.col) and th.equal(q_ref, g.ptr_c)
        # eid_c formula of wrapper.py:110
        b = th.arange(bs).view(-1, 1, 1)
        y = th.arange(l).view(1, -1, 1)
        x = th.arange(l).view(1, 1, -1)
        assert th.equal(g.eid_c, (b * l * l + x * l + y).expand(bs, l, l).reshape(-1))
        for k in ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c", "indices_c"):
            out["cs%d_%s" % (cs, k)] = np_(getattr(g, k))
    for h, d in ((1, 16), (8, 4)):
        gen = th.Generator().manual_seed(100 + h)
        shp = (n, d) if h == 1 else (n, h, d)
        eshp = (e,) if h == 1 else (e, h)
        A = th.rand(shp, generator=gen, requires_grad=True)
        B = th.rand(shp, generator=gen, requires_grad=True)
        grad_e = th.rand(eshp, generator=gen)
        grad_n = th.rand(shp, generator=gen)
        xs = th.rand(eshp, generator=gen, requires_grad=True)
        w = th.rand(eshp, generator=gen, requires_grad=True)
        p = "h%d_" % h
        # SDDMM via bmm (wrapper.py:185 / :364)
        if h == 1:
            yb = (A.view(bs, l, d) @ B.view(bs, l, d).transpose(-1, -2)).view(-1)
        else:
            yb = (A.view(bs, l, h, d).contiguous().transpose(1, 2) @
                  B.view(bs, l, h, d).contiguous().permute(0, 2, 3, 1)).permute(0, 2, 3, 1).contiguous().view(-1, h)
        yb.backward(grad_e)
        out.update({p + "A": np_(A), p + "B": np_(B), p + "grad_e": np_(grad_e), p + "grad_n": np_(grad_n),
                    p + "sddmm_y": np_(yb), p + "sddmm_dA": np_(A.grad), p + "sddmm_dB": np_(B.grad)})
        # softmax scatter / gather (wrapper.py:218,245 / :395,422)
        for name, dim in (("scatter", -1 if h == 1 else -2), ("gather", -2 if h == 1 else -3)):
            xs.grad = None
            ys = th.softmax(xs.view((bs, l, l) if h == 1 else (bs, l, l, h)), dim).view(eshp)
            ys.backward(grad_e)
            out.update({p + "sm_%s_y" % name: np_(ys), p + "sm_%s_dx" % name: np_(xs.grad)})
        out[p + "x"] = np_(xs)
        # SpMM via th.sparse.mm + autograd (wrapper.py:274-283 / :459)
        g = graphs.block_diagonal_graph(bs, l)
        ii = th.stack([g.src, g.dst])
        A.grad = None
        if h == 1:
            adj = th.sparse_coo_tensor(ii, w.detach(), (n, n)).coalesce().requires_grad_(True)
            ysp = th.sparse.mm(adj, A)
            ysp.backward(grad_n)
            dw = adj.grad.coalesce()._values()
        else:
            adjs = [th.sparse_coo_tensor(ii, w.detach()[:, k], (n, n)).coalesce().requires_grad_(True)
                    for k in range(h)]
            ysp = th.stack([th.sparse.mm(adjs[k], A[:, k, :]) for k in range(h)], 1)
            ysp.backward(grad_n)
            dw = th.stack([a.grad.coalesce()._values() for a in adjs], 1)
        out.update({p + "w": np_(w), p + "spmm_y": np_(ysp), p + "spmm_dx": np_(A.grad), p + "spmm_dw": np_(dw)})
    np.savez_compressed(os.path.join(HERE, "k2_harness_small.npz"), **out)
    print("K2: harness fixture bs=3 l=4, h in {1,8}")


def irregular_graph(seed=3, n=23, chunk_size=4):
    gen = th.Generator().manual_seed(seed)
    deg = th.tensor([0, 1, 9, 0, 4, 13, 2, 0, 0, 5, 1, 1, 7, 0, 3, 4, 4, 11, 0, 2, 6, 0, 1])
    assert deg.numel() == n
    src = th.repeat_interleave(th.arange(n), deg)
    dst = th.randint(0, n, (int(deg.sum()),), generator=gen)
    return graphs.graph_from_coo(src, dst, n, chunk_size=chunk_size)


def k3_masked_simple(ref_wrapper):
    g = irregular_graph()
    n, e, d = g.n_src, g.n_edges, 8
    gen = th.Generator().manual_seed(11)
    A = th.rand(n, d, generator=gen, requires_grad=True)
    B = th.rand(n, d, generator=gen, requires_grad=True)
    grad = th.rand(e, generator=gen)
    ar = th.arange(e)
    v = th.ones(e, dtype=th.uint8)
    inc_x = th.sparse_coo_tensor(th.stack([ar, g.src]), v, (e, n))   # wrapper.py:138-139
    inc_y = th.sparse_coo_tensor(th.stack([ar, g.dst]), v, (e, n))
    y = ref_wrapper.MaskedMMSimple.apply(inc_x, inc_y, A, B)           # wrapper.py:57-75
    y.backward(grad)
    out = {k: np_(getattr(g, k)) for k in ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c",
                                           "indices_c", "src", "dst")}
    out.update({"A": np_(A), "B": np_(B), "grad": np_(grad), "y": np_(y), "dA": np_(A.grad),
                "dB": np_(B.grad), "n": np.int64(n)})
    np.savez_compressed(os.path.join(HERE, "k3_maskedmm_simple.npz"), **out)
    print("K3: MaskedMMSimple on irregular graph (n=%d e=%d)" % (n, e))


def k4_function_classes(ref_wrapper):
    """Reference Function classes + oracle ops: records what each .apply / .backward returns."""
    g = irregular_graph(seed=5)
    n, e = g.n_src, g.n_edges
    out = {k: np_(getattr(g, k)) for k in ("row", "ptr_r", "eid_r", "indices_r", "col", "ptr_c", "eid_c",
                                           "indices_c", "src", "dst")}
    out["n"] = np.int64(n)
    for h, d in ((1, 8), (2, 4)):
        gen = th.Generator().manual_seed(21 + h)
        p = "h%d_" % h
        nshape = (n, d) if h == 1 else (n, h, d)
        eshape = (e,) if h == 1 else (e, h)
        A = th.rand(nshape, generator=gen, requires_grad=True)
        B = th.rand(nshape, generator=gen, requires_grad=True)
        x = th.rand(eshape, generator=gen, requires_grad=True)
        w = th.rand(eshape, generator=gen, requires_grad=True)
        Be = th.rand(e, d, generator=gen, requires_grad=True)
        ge = th.rand(eshape, generator=gen)
        gn = th.rand(nshape, generator=gen)
        args = g.csr_args()
        y = ref_wrapper.MaskedMMCSR.apply(*args, A, B)
        y.backward(ge)
        out.update({p + "A": np_(A), p + "B": np_(B), p + "x": np_(x), p + "w": np_(w), p + "Be": np_(Be),
                    p + "ge": np_(ge), p + "gn": np_(gn),
                    p + "mm_y": np_(y), p + "mm_dA": np_(A.grad), p + "mm_dB": np_(B.grad)})
        A.grad = None
        ys = ref_wrapper.SparseSoftmax.apply(g.row, g.ptr_r, g.eid_r, x)
        ys.backward(ge)
        out.update({p + "sm_y": np_(ys), p + "sm_dx": np_(x.grad)})
        x.grad = None
        yg = ref_wrapper.SparseSoftmax.apply(g.col, g.ptr_c, g.eid_c, x)   # "gather" orientation
        yg.backward(ge)
        out.update({p + "smg_y": np_(yg), p + "smg_dx": np_(x.grad)})
        yv = ref_wrapper.VectorSPMM.apply(*args, w, A)
        yv.backward(gn)
        out.update({p + "sp_y": np_(yv), p + "sp_dw": np_(w.grad), p + "sp_dx": np_(A.grad)})
        A.grad = None
        yn = ref_wrapper.NodeMulEdge.apply(g.row, g.ptr_r, g.eid_r, A, Be)
        yn.backward(ge)
        out.update({p + "ne_y": np_(yn), p + "ne_dA": np_(A.grad), p + "ne_dB": np_(Be.grad)})
    np.savez_compressed(os.path.join(HERE, "k4_function_classes.npz"), **out)
    print("K4: reference Function classes driven by the oracle ops")


if __name__ == "__main__":
    th.set_num_threads(1)
    ref_part, ref_wrapper = import_reference()
    k1_partition(ref_part)
    k2_harness(ref_part)
    k3_masked_simple(ref_wrapper)
    k4_function_classes(ref_wrapper)
