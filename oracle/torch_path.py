"""oracle.torch_path -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT.

The reference has no CPU kernels; its CPU-runnable path is the stock-PyTorch
formulation its harness compares the custom kernels against:

* SDDMM "copy to edge": ``(A[src] * B[dst]).sum(-1)``       (wrapper.py:155-157, 57-75)
* per-row softmax of edge scores                             (wrapper.py:218, 395)
* SpMM with edge weights: ``th.sparse.mm(adj, A)``           (wrapper.py:274, 459)
* backward through stock autograd                            (wrapper.py:161, 224, 280)

Restated here in gather / scatter form on COO (src, dst) edge lists so that it
runs on any graph (the harness' ``view(bs, l, l)`` tricks only work for its
block-diagonal fixture).  ``attention_step`` is the composed fwd+bwd step the
headline metric times; bench.py's ``cpu_baseline`` leg times exactly this.
"""
import torch


def sddmm(src, dst, A, B):
    """y[e,(k)] = <A[src[e],(k),:], B[dst[e],(k),:]>"""
    return (A[src] * B[dst]).sum(-1)


def segment_softmax(seg, x, n_seg):
    """softmax of x over edges sharing ``seg`` (per head if x is (E,h))."""
    shape = (n_seg,) + tuple(x.shape[1:])
    idx = seg if x.dim() == 1 else seg[:, None].expand_as(x)
    m = torch.full(shape, float("-inf"), dtype=x.dtype).scatter_reduce(0, idx, x.detach(), "amax", include_self=True)
    ex = torch.exp(x - m[seg])
    s = torch.zeros(shape, dtype=x.dtype).index_add_(0, seg, ex)
    return ex / s[seg]


def spmm(src, dst, w, X, n_out):
    """y[r,(k),:] = sum_{e: src[e]==r} w[e,(k)] * X[dst[e],(k),:]"""
    msg = w.unsqueeze(-1) * X[dst]
    return torch.zeros((n_out,) + tuple(X.shape[1:]), dtype=X.dtype).index_add_(0, src, msg)


def attention_step(src, dst, Q, K, V, dO, n_nodes):
    """One fwd+bwd of SDDMM -> row-softmax -> SpMM.  Returns (s, a, o, dQ, dK, dV)."""
    Q = Q.detach().clone().requires_grad_(True)
    K = K.detach().clone().requires_grad_(True)
    V = V.detach().clone().requires_grad_(True)
    s = sddmm(src, dst, Q, K)
    a = segment_softmax(src, s, n_nodes)
    o = spmm(src, dst, a, V, n_nodes)
    o.backward(dO)
    return s.detach(), a.detach(), o.detach(), Q.grad, K.grad, V.grad


def attention_step_blocked(src, dst, indptr, Q, K, V, dO, n_nodes, rows_per_block=4096):
    """Same step, processed in blocks of source rows so the E x d temporaries stay small.

    Valid because every stage is row-local on the source side; dK/dV are summed over
    blocks.  ``src`` must be sorted (row-major CSR order) and ``indptr`` its row pointer.
    Returns (o, dQ, dK, dV)."""
    o = torch.zeros_like(V[:n_nodes])
    dQ = torch.zeros_like(Q)
    dK = torch.zeros_like(K)
    dV = torch.zeros_like(V)
    for r0 in range(0, n_nodes, rows_per_block):
        r1 = min(n_nodes, r0 + rows_per_block)
        e0, e1 = int(indptr[r0]), int(indptr[r1])
        if e1 == e0:
            continue
        sl = slice(e0, e1)
        qb = Q[r0:r1].detach().clone().requires_grad_(True)
        Kr = K.detach().requires_grad_(True)
        Vr = V.detach().requires_grad_(True)
        lsrc = src[sl] - r0
        s = sddmm(lsrc, dst[sl], qb, Kr)
        a = segment_softmax(lsrc, s, r1 - r0)
        ob = spmm(lsrc, dst[sl], a, Vr, r1 - r0)
        ob.backward(dO[r0:r1])
        o[r0:r1] = ob.detach()
        dQ[r0:r1] = qb.grad
        dK += Kr.grad
        dV += Vr.grad
    return o, dQ, dK, dV


def attention_step_incidence(src, dst, Q, K, V, dO, n_nodes):
    """The harness-verbatim form (small graphs only): copy-to-edge SDDMM through sparse incidence
    matrices, ``(th.sparse.mm(inc_x, A) * th.sparse.mm(inc_y, B)).sum(-1)`` (wrapper.py:155-157; the
    same product MaskedMMSimple hand-differentiates, wrapper.py:57-75), the row softmax, and the
    SpMM as ``th.sparse.mm(adj, V)`` with autograd through the sparse values (wrapper.py:274-283).
    Returns (o, dQ, dK, dV)."""
    E = src.numel()
    ar = torch.arange(E)
    one = torch.ones(E, dtype=Q.dtype)
    inc_x = torch.sparse_coo_tensor(torch.stack([ar, src]), one, (E, Q.size(0)))
    inc_y = torch.sparse_coo_tensor(torch.stack([ar, dst]), one, (E, K.size(0)))
    Q = Q.detach().clone().requires_grad_(True)
    K = K.detach().clone().requires_grad_(True)
    V = V.detach().clone().requires_grad_(True)
    s = (torch.sparse.mm(inc_x, Q) * torch.sparse.mm(inc_y, K)).sum(-1)
    a = segment_softmax(src, s, n_nodes)
    adj = torch.sparse_coo_tensor(torch.stack([src, dst]), a, (n_nodes, V.size(0)))
    o = torch.sparse.mm(adj, V)
    o.backward(dO)
    return o.detach(), Q.grad, K.grad, V.grad
