"""autograd glue: the reference's four ``torch.autograd.Function`` classes.

Same class names, ``.apply`` argument orders and gradient routing as ``wrapper.py:8-55``: index
tensors get ``None`` gradients; ``MaskedMMCSR`` returns ``(dA, dB)`` last, ``VectorSPMM``
``(dedata, dx)`` last, ``SparseSoftmax`` a 4-tuple, ``NodeMulEdge`` a 5-tuple.  They call this
package's HIP ops instead of the CUDA extension.
"""
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


class FusedAttention(Function):
    """o = VectorSPMM(SparseSoftmax(MaskedMMCSR(Q, K)), V) as ONE autograd node (extra op, not in the
    reference): apply(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, Q, K, V).
    Saves (Q, K, V, o, row statistics) instead of the E-sized s / a; the backward recomputes them
    inside two fused passes."""

    @staticmethod
    def forward(ctx, row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, Q, K, V):
        a8 = (row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c)
        ctx.fused = _ops.attention_backward_is_fused(*a8, Q, K)
        if ctx.fused:
            o, stats = _ops.attention_forward(row, indptr_r, eid_r, indices_r, Q, K, V)
            ctx.save_for_backward(*a8, Q, K, V, o, stats)
        else:
            # the fused passes do not apply (several heads, fp64, short rows ...): keep a for the unfused
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
