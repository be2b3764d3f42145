"""autograd glue: the reference's four ``torch.autograd.Function`` classes.

Same class names, ``.apply`` argument orders and gradient routing as ``wrapper.py:8-55``: index
tensors get ``None`` gradients; ``MaskedMMCSR`` returns ``(dA, dB)`` last, ``VectorSPMM``
``(dedata, dx)`` last, ``SparseSoftmax`` a 4-tuple, ``NodeMulEdge`` a 5-tuple.  They call this
package's HIP ops instead of the CUDA extension.
"""
import torch
from torch.autograd import Function

from . import _lib
from . import graphop as _ops


class SparseSoftmax(Function):
    """y = softmax of edge values per row; apply(row, indptr, eid, x)   (wrapper.py:8-18)"""

    @staticmethod
    def forward(ctx, row, indptr, eid, x):
        y = _ops.sparse_softmax_forward(row, indptr, eid, x)
        ctx.save_for_backward(row, indptr, eid, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        row, indptr, eid, y = ctx.saved_tensors
        return None, None, None, _ops.sparse_softmax_backward(row, indptr, eid, y, dy)


class MaskedMMCSR(Function):
    """SDDMM; apply(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B)
    (wrapper.py:20-30)"""

    @staticmethod
    def forward(ctx, row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B):
        ctx.save_for_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B)
        return _ops.maskedmm_csr_forward(row, indptr_r, eid_r, indices_r, A, B)

    @staticmethod
    def backward(ctx, grad):
        row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B = ctx.saved_tensors
        dA, dB = _ops.maskedmm_csr_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c,
                                            indices_c, A, B, grad)
        return None, None, None, None, None, None, None, None, dA, dB


class NodeMulEdge(Function):
    """apply(row, indptr, eid, A, B)   (wrapper.py:32-42)"""

    @staticmethod
    def forward(ctx, row, indptr, eid, A, B):
        ctx.save_for_backward(row, indptr, eid, A, B)
        return _ops.node_mul_edge_forward(row, indptr, eid, A, B)

    @staticmethod
    def backward(ctx, grad):
        row, indptr, eid, A, B = ctx.saved_tensors
        dA, dB = _ops.node_mul_edge_backward(row, indptr, eid, A, B, grad)
        return None, None, None, dA, dB


class VectorSPMM(Function):
    """apply(row, indptr, eid, indices, col, ptr_t, eid_t, indices_t, edata, x)
    (wrapper.py:44-55)"""

    @staticmethod
    def forward(ctx, row, indptr, eid, indices, col, ptr_t, eid_t, indices_t, edata, x):
        y = _ops.vector_spmm_forward(row, indptr, eid, indices, edata, x)
        ctx.save_for_backward(row, indptr, eid, indices, col, ptr_t, eid_t, indices_t, edata, x)
        return y

    @staticmethod
    def backward(ctx, dy):
        row, indptr, eid, indices, col, ptr_t, eid_t, indices_t, edata, x = ctx.saved_tensors
        dedata, dx = _ops.vector_spmm_backward(row, indptr, eid, indices, col, ptr_t, eid_t,
                                               indices_t, edata, dy.contiguous(), x)
        return None, None, None, None, None, None, None, None, dedata, dx


# FusedAttention over several heads (round 5): "keep" = per head group only a_g (E x hg) survives the forward, the backward's
# da_g / ds_g are E x hg temporaries -- speed of the 8-function step, about half of its E-sized memory; "recompute" = nothing
# E-sized survives the forward, the backward recomputes s_g and a_g per group (two more passes per group: ~+17 % time,
# about a third of the 8-function step's E-sized memory).  Set before the forward; GRAPHOP_FUSED_HEADS overrides.
FUSED_HEADS_MODE = "keep"


def _head_group(h, d, n_edges=None, n_nodes=None):
    """Heads per group of the head-blocked FusedAttention: rows of hg x d floats = 256 B where d allows (the row width
    every driver is fastest at per byte: Reddit-shape 2 x 32 runs 13.6 ms against 55.4 / 4 at 8 x 32), one head per
    group from d = 64 on (d = 64: the one-head fused kernels then apply to every head).  hg divides h; hg == h: no blocking.
    With the graph's size given, blocking is only chosen where it SAVES memory: it trades (E, h)-sized temporaries
    (2 E (h - hg) floats) for about seven node-sized copies per group (7 n hg d floats) -- on graphs of few edges per
    node the node tensors are the big ones (products-shape 8 x 16, E / n = 25: measured 1.5 x the 8-function step's peak
    and +18 % time with groups of 4, tools/fused_heads_memory.py), so the margin is a factor two."""
    want = max(1, 64 // max(1, d))
    hg = 1
    for c in range(1, h + 1):
        if h % c == 0 and c <= want:
            hg = c
    if n_edges is not None and n_nodes and hg < h and 2 * n_edges * (h - hg) < 2 * 7 * n_nodes * hg * d:
        return h
    return hg


class FusedAttention(Function):
    """o = VectorSPMM(SparseSoftmax(MaskedMMCSR(Q, K)), V) as ONE autograd node (extra op, not in the
    reference): apply(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, Q, K, V).
    Saves (Q, K, V, o, row statistics) instead of the E-sized s / a; the backward recomputes them
    inside two fused passes.
    Several heads (the reference's second benchmarked layout is 8 x 64, wrapper.py:306-309; BASELINE config 3 is 8 x 128)
    are processed in HEAD GROUPS (round 5): a group's heads are copied to contiguous (n, hg, d) tensors (node-sized
    copies), run through this op's one-group form -- the fused kernels where they apply (one head of d <= 64), the
    unfused entry points otherwise -- and written back into the heads' slices of o / dQ / dK / dV.  No (E, h) tensor
    ever exists: the E-sized temporaries are (E, hg), one group at a time (FUSED_HEADS_MODE)."""

    @staticmethod
    def forward(ctx, row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, Q, K, V):
        a8 = (row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c)
        h = Q.size(1) if Q.dim() == 3 else 1
        hg = _head_group(h, Q.size(-1), eid_r.numel(), max(Q.size(0), K.size(0))) if h > 1 else 1
        ctx.groups = None
        if h > 1 and hg < h:
            import os
            mode = os.environ.get("GRAPHOP_FUSED_HEADS", FUSED_HEADS_MODE)
            ctx.groups = (h, hg, mode)
            o = Q.new_empty((Q.size(0),) + tuple(V.shape[1:]))
            kept = []
            for g0 in range(0, h, hg):
                Qg, Kg, Vg = (_head_slice(x, g0, hg) for x in (Q, K, V))
                og, keep = _group_forward(a8, Qg, Kg, Vg, mode)
                _head_store(o, og, g0, hg)
                kept.append(keep)
                del Qg, Kg, Vg, og
            ctx.kept = kept               # per group: ("fused", o_g, stats_g) | ("keep", a_g) | ("recompute",)
            ctx.save_for_backward(*a8, Q, K, V)
            return o
        ctx.fused = _ops.attention_backward_is_fused(*a8, Q, K)
        if ctx.fused:
            o, stats = _ops.attention_forward(row, indptr_r, eid_r, indices_r, Q, K, V)
            ctx.save_for_backward(*a8, Q, K, V, o, stats)
        else:
            # the fused passes do not apply (fp64, short rows, one group of several heads ...): keep a for the unfused
            # backward ops instead of letting attention_backward recompute s and a
            s = _ops.maskedmm_csr_forward(row, indptr_r, eid_r, indices_r, Q, K)
            a = _ops.sparse_softmax_forward(row, indptr_r, eid_r, s)
            del s
            o = _ops.vector_spmm_forward(row, indptr_r, eid_r, indices_r, a, V)
            if o.size(0) != Q.size(0):
                o = o[:Q.size(0)]
            ctx.save_for_backward(*a8, Q, K, V, a)
        return o

    @staticmethod
    def backward(ctx, dO):
        a8, rest = ctx.saved_tensors[:8], ctx.saved_tensors[8:]
        row, indptr_r, eid_r = a8[:3]
        if ctx.groups is not None:
            h, hg, mode = ctx.groups
            Q, K, V = rest
            dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
            for gi, g0 in enumerate(range(0, h, hg)):
                Qg, Kg, Vg, dOg = (_head_slice(x, g0, hg) for x in (Q, K, V, dO))
                dQg, dKg, dVg = _group_backward(a8, Qg, Kg, Vg, dOg, ctx.kept[gi])
                for full, part in ((dQ, dQg), (dK, dKg), (dV, dVg)):
                    _head_store(full, part, g0, hg)
                del Qg, Kg, Vg, dOg, dQg, dKg, dVg
            return None, None, None, None, None, None, None, None, dQ, dK, dV
        if ctx.fused:
            Q, K, V, o, stats = rest
            dQ, dK, dV = _ops.attention_backward(*a8, Q, K, V, o, stats, dO)
        else:
            Q, K, V, a = rest
            da, dV = _ops.vector_spmm_backward(*a8, a, dO.contiguous(), V)
            ds = _ops.sparse_softmax_backward(row, indptr_r, eid_r, a, da)
            del da
            dQ, dK = _ops.maskedmm_csr_backward(*a8, Q, K, ds)
        return None, None, None, None, None, None, None, None, dQ, dK, dV


def _head_slice(x, g0, hg):
    """Heads [g0, g0 + hg) of a (n, h, d) tensor as a contiguous (n, hg, d) tensor ((n, d) for one head: the one-head
    kernels and the fused passes take that form)."""
    part = x[:, g0:g0 + hg, :]
    return part.reshape(x.size(0), x.size(2)).contiguous() if hg == 1 else part.contiguous()


def _head_store(full, part, g0, hg):
    """full[:, g0 : g0 + hg, :] = part ((n, hg, d) or (n, d); the unfused SpMM returns zeros_like(x) rows: cut to full's)."""
    full[:, g0:g0 + hg, :].copy_(part[:full.size(0)].reshape(full.size(0), hg, full.size(2)))


def _group_forward(a8, Qg, Kg, Vg, mode):
    row, indptr_r, eid_r, indices_r = a8[:4]
    if _ops.attention_backward_is_fused(*a8, Qg, Kg):
        og, stats = _ops.attention_forward(row, indptr_r, eid_r, indices_r, Qg, Kg, Vg)
        return og, ("fused", og, stats)
    s = _ops.maskedmm_csr_forward(row, indptr_r, eid_r, indices_r, Qg, Kg)
    a = _ops.sparse_softmax_forward(row, indptr_r, eid_r, s)
    del s
    og = _ops.vector_spmm_forward(row, indptr_r, eid_r, indices_r, a, Vg)
    return og, (("keep", a) if mode != "recompute" else ("recompute",))


def _group_backward(a8, Qg, Kg, Vg, dOg, kept):
    row, indptr_r, eid_r, indices_r = a8[:4]
    if kept[0] == "fused":
        return _ops.attention_backward(*a8, Qg, Kg, Vg, kept[1], kept[2], dOg)
    if kept[0] == "keep":
        a = kept[1]
    else:
        s = _ops.maskedmm_csr_forward(row, indptr_r, eid_r, indices_r, Qg, Kg)
        a = _ops.sparse_softmax_forward(row, indptr_r, eid_r, s)
        del s
    da, dVg = _ops.vector_spmm_backward(*a8, a, dOg, Vg)
    ds = _ops.sparse_softmax_backward(row, indptr_r, eid_r, a, da)
    del da, a
    dQg, dKg = _ops.maskedmm_csr_backward(*a8, Qg, Kg, ds)
    return dQg, dKg, dVg


def fused_attention_step(g, Q, K, V, dO):
    """The same fwd+bwd as attention_step through the fused op; returns o."""
    o = FusedAttention.apply(*g.csr_args(), Q, K, V)
    o.backward(dO)
    _lib.check_errors(sync=False)     # as in attention_step
    return o


def attention_step(g, Q, K, V, dO):
    """One fwd+bwd of the composed hot path the headline metric times:
    s = SDDMM(Q, K); a = row-softmax(s); o = SpMM(a, V); o.backward(dO).
    Q, K, V must be leaf tensors with requires_grad; returns (s, a, o)."""
    args = g.csr_args()
    s = MaskedMMCSR.apply(*args, Q, K)
    a = SparseSoftmax.apply(g.row, g.ptr_r, g.eid_r, s)
    o = VectorSPMM.apply(*args, a, V)
    o.backward(dO)
    # a device-side abort (include/graphop_hip.h: graphop_check_device_errors) of a launch that has already finished is
    # raised HERE, before the gradients leave the step; one still in flight is sticky and fails the next op call
    _lib.check_errors(sync=False)
    return s, a, o
