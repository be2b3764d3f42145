"""The reference's operator surface, MI355X-native.

Mirror of the pybind11 module ``graphop`` (``graphop/graphop.cpp:216-225``): the same eight
names, positional signatures, return types (a Tensor, or a list of two Tensors), output shapes
(``(e)`` when h == 1 else ``(e, h)``, ``graphop_kernel.cu:284``) and error behaviour
(``RuntimeError("<arg> must be a CUDA tensor")`` / ``"<arg> must be contiguous"``,
``graphop.cpp:4-6``).  Each function flattens its tensors to pointers + sizes and calls the C ABI
of ``libgraphop_hip.so`` (``include/graphop_hip.h``) on the current stream, without syncing.

Additionally every op is registered as ``torch.ops.graphop.<name>`` (the reference has no
TORCH_LIBRARY registration; BASELINE.json's north_star asks for this surface).

Deliberate, documented deviations from the reference:
  * ``dy`` is made contiguous in the backward ops (the reference forgets to check it,
    ``graphop.cpp:120-129`` and reads garbage from a strided ``dy``).
  * graphs are validated once per (row, indptr, eid, indices) identity when their plan is built
    (index range, indptr bounds): the reference reads out of bounds instead.
  * ``vector_spmm_backward`` processes every column chunk (reference grid bug,
    ``graphop_kernel.cu:566,588``).
  * non-square operands are allowed: B / x may have a different row count than A / y.
"""
import torch

from . import _lib
from ._lib import check, dtype_code, get_plan, lib, ptr, stream_of

__all__ = ["maskedmm_csr_forward", "maskedmm_csr_backward", "node_mul_edge_forward",
           "node_mul_edge_backward", "sparse_softmax_forward", "sparse_softmax_backward",
           "vector_spmm_forward", "vector_spmm_backward"]
# extra ops (not in the reference's module): the fused attention step, SURVEY.md 8f N2
EXTRA_OPS = ["attention_forward", "attention_backward", "attention_backward_is_fused"]

_NULL = None


def _check_input(t, name):
    # CHECK_CUDA / CHECK_CONTIGUOUS, graphop.cpp:4-6
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _check_index(t, name):
    if t.dtype != torch.int64:
        # the reference throws from .data<int64_t>() (graphop_kernel.cu:293)
        raise RuntimeError("expected scalar type Long but found %s (%s)" % (t.dtype, name))


def _same_dtype(a, b, na, nb):
    if a.dtype != b.dtype:
        raise RuntimeError("expected %s and %s to have the same dtype, got %s and %s"
                           % (na, nb, a.dtype, b.dtype))


def _plan(row, indptr, eid, indices, bound):
    return get_plan(row, indptr, eid, indices, bound)


def maskedmm_csr_forward(row, indptr, eid, indices, A, B):
    """y[eid[j], k] = <A[row[c], k], B[indices[j], k]>   (graphop.cpp:16-30)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices"),
                 (A, "A"), (B, "B")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices")):
        _check_index(t, n)
    _same_dtype(A, B, "A", "B")
    e, d = eid.size(0), A.size(-1)
    h = 1 if A.dim() == 2 else A.size(1)                    # graphop_kernel.cu:283
    y = torch.empty((e,) if h == 1 else (e, h), dtype=A.dtype, device=A.device)
    with _lib.device_guard(A.device):
        plan = _plan(row, indptr, eid, indices, B.size(0))
        check(lib().graphop_maskedmm_csr_forward(
            dtype_code(A), ptr(row), ptr(indptr), ptr(eid), ptr(indices), ptr(A), ptr(B), ptr(y),
            row.size(0), e, A.size(0), B.size(0), h, d, plan.handle, stream_of(A)))
    return y


def maskedmm_csr_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c,
                          A, B, dy):
    """-> [dA, dB]   (graphop.cpp:108-131)"""
    names = ("row", "indptr_r", "eid_r", "indices_r", "col", "indptr_c", "eid_c", "indices_c")
    idx = (row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c)
    for t, n in zip(idx + (A, B), names + ("A", "B")):
        _check_input(t, n)
    for t, n in zip(idx, names):
        _check_index(t, n)
    if not isinstance(dy, torch.Tensor) or not dy.is_cuda:
        raise RuntimeError("dy must be a CUDA tensor")
    _same_dtype(A, B, "A", "B")
    _same_dtype(A, dy, "A", "dy")
    dy = dy.contiguous()
    d = A.size(-1)
    h = dy.size(1) if dy.dim() == 2 else 1                  # graphop_kernel.cu:373
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    with _lib.device_guard(A.device):
        plan_r = _plan(row, indptr_r, eid_r, indices_r, B.size(0))
        plan_c = _plan(col, indptr_c, eid_c, indices_c, A.size(0))
        check(lib().graphop_maskedmm_csr_backward(
            dtype_code(A), ptr(row), ptr(indptr_r), ptr(eid_r), ptr(indices_r), ptr(col),
            ptr(indptr_c), ptr(eid_c), ptr(indices_c), ptr(A), ptr(B), ptr(dy), ptr(dA), ptr(dB),
            row.size(0), col.size(0), eid_r.size(0), A.size(0), B.size(0), h, d,
            plan_r.handle, plan_c.handle, stream_of(A)))
    return [dA, dB]


def sparse_softmax_forward(row, indptr, eid, x):
    """Per-row (per-head) softmax of edge values   (graphop.cpp:59-69)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (x, "x")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid")):
        _check_index(t, n)
    h = x.size(1) if x.dim() == 2 else 1
    y = torch.empty_like(x)
    with _lib.device_guard(x.device):
        plan = _plan(row, indptr, eid, None, 0)
        ws, ws_rows = None, 0
        if not plan.info.row_owned:                          # general layout: atomics + scratch
            ws_rows = plan.info.max_row + 1
            ws = torch.empty(2 * ws_rows * h, dtype=x.dtype, device=x.device)
        check(lib().graphop_sparse_softmax_forward(
            dtype_code(x), ptr(row), ptr(indptr), ptr(eid), ptr(x), ptr(y), row.size(0),
            eid.size(0), h, ptr(ws), ws_rows, plan.handle, stream_of(x)))
    return y


def sparse_softmax_backward(row, indptr, eid, y, dy):
    """dx = dy*y - (sum_row dy*y)*y   (graphop.cpp:163-175)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (y, "y")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid")):
        _check_index(t, n)
    if not isinstance(dy, torch.Tensor) or not dy.is_cuda:
        raise RuntimeError("dy must be a CUDA tensor")
    _same_dtype(y, dy, "y", "dy")
    dy = dy.contiguous()
    h = dy.size(1) if dy.dim() == 2 else 1
    dx = torch.empty_like(dy)
    with _lib.device_guard(y.device):
        plan = _plan(row, indptr, eid, None, 0)
        ws, ws_rows = None, 0
        if not plan.info.row_owned:
            ws_rows = plan.info.max_row + 1
            ws = torch.empty(ws_rows * h, dtype=y.dtype, device=y.device)
        check(lib().graphop_sparse_softmax_backward(
            dtype_code(y), ptr(row), ptr(indptr), ptr(eid), ptr(y), ptr(dy), ptr(dx), row.size(0),
            eid.size(0), h, ptr(ws), ws_rows, plan.handle, stream_of(y)))
    return dx


def vector_spmm_forward(row, indptr, eid, indices, edata, x):
    """y[row[c], k] += sum_j edata[eid[j], k] * x[indices[j], k]   (graphop.cpp:79-93)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices"),
                 (edata, "edata"), (x, "x")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices")):
        _check_index(t, n)
    _same_dtype(edata, x, "edata", "x")
    h = edata.size(1) if edata.dim() == 2 else 1            # graphop_kernel.cu:520
    d = x.size(-1)
    y = torch.empty_like(x)                                  # zeros_like(x), :527
    with _lib.device_guard(x.device):
        plan = _plan(row, indptr, eid, indices, x.size(0))
        if plan.info.max_row >= x.size(0):
            raise RuntimeError("vector_spmm_forward: row id %d but y = zeros_like(x) has %d rows"
                               % (plan.info.max_row, x.size(0)))
        check(lib().graphop_vector_spmm_forward(
            dtype_code(x), ptr(row), ptr(indptr), ptr(eid), ptr(indices), ptr(edata), ptr(x),
            ptr(y), row.size(0), eid.size(0), x.size(0), x.size(0), h, d, plan.handle,
            stream_of(x)))
    return y


def vector_spmm_backward(row, indptr, eid, indices, col, indptr_t, eid_t, indices_t, edata, dy, x):
    """-> [dedata, dx]; NB ``dy`` comes before ``x``   (graphop.cpp:190-214)"""
    names = ("row", "indptr", "eid", "indices", "col", "indptr_t", "eid_t", "indices_t")
    idx = (row, indptr, eid, indices, col, indptr_t, eid_t, indices_t)
    for t, n in zip(idx + (edata, dy, x), names + ("edata", "dy", "x")):
        _check_input(t, n)
    for t, n in zip(idx, names):
        _check_index(t, n)
    _same_dtype(edata, x, "edata", "x")
    _same_dtype(dy, x, "dy", "x")
    h = edata.size(1) if edata.dim() == 2 else 1            # graphop_kernel.cu:560
    d = x.size(-1)
    dedata, dx = torch.empty_like(edata), torch.empty_like(x)
    with _lib.device_guard(x.device):
        plan_r = _plan(row, indptr, eid, indices, x.size(0))
        plan_c = _plan(col, indptr_t, eid_t, indices_t, dy.size(0))
        if plan_c.info.max_row >= x.size(0):
            raise RuntimeError("vector_spmm_backward: col id %d but dx has %d rows"
                               % (plan_c.info.max_row, x.size(0)))
        check(lib().graphop_vector_spmm_backward(
            dtype_code(x), ptr(row), ptr(indptr), ptr(eid), ptr(indices), ptr(col), ptr(indptr_t),
            ptr(eid_t), ptr(indices_t), ptr(edata), ptr(dy), ptr(x), ptr(dedata), ptr(dx),
            row.size(0), col.size(0), eid.size(0), x.size(0), dy.size(0), h, d, plan_r.handle,
            plan_c.handle, stream_of(x)))
    return [dedata, dx]


def node_mul_edge_forward(row, indptr, eid, A, B):
    """y[eid[j], k] = <A[row[c], k], B[eid[j]]>   (graphop.cpp:39-51)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (A, "A"), (B, "B")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid")):
        _check_index(t, n)
    _same_dtype(A, B, "A", "B")
    e, d = eid.size(0), A.size(-1)
    h = 1 if A.dim() == 2 else A.size(1)
    if B.size(0) < e or B.size(-1) != d:
        raise RuntimeError("node_mul_edge_forward: B must be (n_edges, d)")
    y = torch.empty((e,) if h == 1 else (e, h), dtype=A.dtype, device=A.device)
    with _lib.device_guard(A.device):
        plan = _plan(row, indptr, eid, None, 0)
        check(lib().graphop_node_mul_edge_forward(
            dtype_code(A), ptr(row), ptr(indptr), ptr(eid), ptr(A), ptr(B), ptr(y), row.size(0), e,
            A.size(0), h, d, plan.handle, stream_of(A)))
    return y


def node_mul_edge_backward(row, indptr, eid, A, B, dy):
    """-> [dA, dB]   (graphop.cpp:141-154)"""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (A, "A"), (B, "B")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid")):
        _check_index(t, n)
    if not isinstance(dy, torch.Tensor) or not dy.is_cuda:
        raise RuntimeError("dy must be a CUDA tensor")
    _same_dtype(A, B, "A", "B")
    _same_dtype(A, dy, "A", "dy")
    dy = dy.contiguous()
    d = A.size(-1)
    h = dy.size(1) if dy.dim() == 2 else 1
    e = eid.size(0)
    if B.size(0) != e or B.size(-1) != d:
        raise RuntimeError("node_mul_edge_backward: B must be (n_edges, d)")
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    with _lib.device_guard(A.device):
        plan = _plan(row, indptr, eid, None, 0)
        check(lib().graphop_node_mul_edge_backward(
            dtype_code(A), ptr(row), ptr(indptr), ptr(eid), ptr(A), ptr(B), ptr(dy), ptr(dA),
            ptr(dB), row.size(0), e, A.size(0), h, d, plan.handle, stream_of(A)))
    return [dA, dB]


# ---- fused attention step (extra op; the composition wrapper.py:20-30, 8-18, 44-55) ------------------
def _workspace(like, dtype, backward, e, n_q, n_k, h, d, plan_r, plan_c):
    import ctypes
    nbytes = ctypes.c_int64(0)
    check(lib().graphop_attention_workspace_bytes(
        dtype, 1 if backward else 0, e, n_q, n_k, h, d, plan_r.handle if plan_r is not None else _NULL,
        plan_c.handle if plan_c is not None else _NULL, stream_of(like), ctypes.byref(nbytes)))
    return torch.empty(max(1, nbytes.value), dtype=torch.uint8, device=like.device), nbytes.value


def attention_backward_is_fused(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, Q, K):
    """True when attention_backward will run its fused window passes for this graph and these
    shapes (fp32, one head, sweepable plans, tables beyond the L2); False when it would compose
    the unfused ops, recomputing s and a."""
    import ctypes
    d = Q.size(-1)
    h = 1 if Q.dim() == 2 else Q.size(1)
    out = ctypes.c_int(0)
    with _lib.device_guard(Q.device):
        plan_r = _plan(row, indptr_r, eid_r, indices_r, K.size(0))
        plan_c = _plan(col, indptr_c, eid_c, indices_c, Q.size(0))
        check(lib().graphop_attention_backward_is_fused(dtype_code(Q), eid_r.size(0), Q.size(0), K.size(0), h, d,
                                                        plan_r.handle, plan_c.handle, stream_of(Q), ctypes.byref(out)))
    return bool(out.value)


def attention_forward(row, indptr, eid, indices, Q, K, V):
    """-> [o, stats]: o = vector_spmm(sparse_softmax(maskedmm_csr(Q, K)), V) over the row-major CSR,
    without returning the E-sized s / a.  stats (n_q, h, 2) = (row max, 1 / sum exp) is what
    attention_backward needs to recompute them."""
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices"), (Q, "Q"),
                 (K, "K"), (V, "V")):
        _check_input(t, n)
    for t, n in ((row, "row"), (indptr, "indptr"), (eid, "eid"), (indices, "indices")):
        _check_index(t, n)
    _same_dtype(Q, K, "Q", "K")
    _same_dtype(Q, V, "Q", "V")
    if K.shape != V.shape or Q.shape[1:] != K.shape[1:]:
        raise RuntimeError("attention_forward: Q (n_q,[h,]d), K and V (n_k,[h,]d) expected")
    e, d = eid.size(0), Q.size(-1)
    h = 1 if Q.dim() == 2 else Q.size(1)
    n_q, n_k = Q.size(0), K.size(0)
    o = torch.empty_like(Q)
    stats = torch.empty((n_q, h, 2), dtype=Q.dtype, device=Q.device)
    with _lib.device_guard(Q.device):
        plan = _plan(row, indptr, eid, indices, n_k)
        ws, nbytes = _workspace(Q, dtype_code(Q), False, e, n_q, n_k, h, d, plan, None)
        check(lib().graphop_attention_forward(
            dtype_code(Q), ptr(row), ptr(indptr), ptr(eid), ptr(indices), ptr(Q), ptr(K), ptr(V),
            ptr(o), ptr(stats), row.size(0), e, n_q, n_k, h, d, ptr(ws), nbytes, plan.handle,
            stream_of(Q)))
    return [o, stats]


def attention_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c,
                       Q, K, V, o, stats, dO):
    """-> [dQ, dK, dV] of the fused step for the output gradient dO."""
    names = ("row", "indptr_r", "eid_r", "indices_r", "col", "indptr_c", "eid_c", "indices_c")
    idx = (row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c)
    for t, n in zip(idx + (Q, K, V, o, stats), names + ("Q", "K", "V", "o", "stats")):
        _check_input(t, n)
    for t, n in zip(idx, names):
        _check_index(t, n)
    if not isinstance(dO, torch.Tensor) or not dO.is_cuda:
        raise RuntimeError("dO must be a CUDA tensor")
    for t, n in ((K, "K"), (V, "V"), (o, "o"), (stats, "stats"), (dO, "dO")):
        _same_dtype(Q, t, "Q", n)
    dO = dO.contiguous()
    e, d = eid_r.size(0), Q.size(-1)
    h = 1 if Q.dim() == 2 else Q.size(1)
    n_q, n_k = Q.size(0), K.size(0)
    if o.shape != Q.shape or dO.shape != Q.shape or stats.numel() != n_q * h * 2:
        raise RuntimeError("attention_backward: o, dO must match Q and stats must be (n_q, h, 2)")
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
    with _lib.device_guard(Q.device):
        plan_r = _plan(row, indptr_r, eid_r, indices_r, n_k)
        plan_c = _plan(col, indptr_c, eid_c, indices_c, n_q)
        ws, nbytes = _workspace(Q, dtype_code(Q), True, e, n_q, n_k, h, d, plan_r, plan_c)
        check(lib().graphop_attention_backward(
            dtype_code(Q), ptr(row), ptr(indptr_r), ptr(eid_r), ptr(indices_r), ptr(col),
            ptr(indptr_c), ptr(eid_c), ptr(indices_c), ptr(Q), ptr(K), ptr(V), ptr(o), ptr(stats),
            ptr(dO), ptr(dQ), ptr(dK), ptr(dV), row.size(0), col.size(0), e, n_q, n_k, h, d,
            ptr(ws), nbytes, plan_r.handle, plan_c.handle, stream_of(Q)))
    return [dQ, dK, dV]


def prepare(graph, h=1, d=64, dtype=torch.float32, fused=True):
    """Build a graph's plans and window structures ahead of the first op call (see graphs.prepare)."""
    from . import graphs
    return graphs.prepare(graph, h, d, dtype, fused)


def release(graph):
    """Drop a graph's plans (see graphs.release)."""
    from . import graphs
    graphs.release(graph)


# ---- torch.ops.graphop.* ---------------------------------------------------------------------------
_SCHEMAS = {
    "maskedmm_csr_forward": "(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor A, Tensor B) -> Tensor",
    "maskedmm_csr_backward": "(Tensor row, Tensor indptr_r, Tensor eid_r, Tensor indices_r, Tensor col, Tensor indptr_c, Tensor eid_c, Tensor indices_c, Tensor A, Tensor B, Tensor dy) -> Tensor[]",
    "node_mul_edge_forward": "(Tensor row, Tensor indptr, Tensor eid, Tensor A, Tensor B) -> Tensor",
    "node_mul_edge_backward": "(Tensor row, Tensor indptr, Tensor eid, Tensor A, Tensor B, Tensor dy) -> Tensor[]",
    "sparse_softmax_forward": "(Tensor row, Tensor indptr, Tensor eid, Tensor x) -> Tensor",
    "sparse_softmax_backward": "(Tensor row, Tensor indptr, Tensor eid, Tensor y, Tensor dy) -> Tensor",
    "vector_spmm_forward": "(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor edata, Tensor x) -> Tensor",
    "vector_spmm_backward": "(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor col, Tensor indptr_t, Tensor eid_t, Tensor indices_t, Tensor edata, Tensor dy, Tensor x) -> Tensor[]",
    "attention_forward": "(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor Q, Tensor K, Tensor V) -> Tensor[]",
    "attention_backward": "(Tensor row, Tensor indptr_r, Tensor eid_r, Tensor indices_r, Tensor col, Tensor indptr_c, Tensor eid_c, Tensor indices_c, Tensor Q, Tensor K, Tensor V, Tensor o, Tensor stats, Tensor dO) -> Tensor[]",
}
_torch_lib = None


def _cpu_refusal(name):
    def _impl(*args):
        raise RuntimeError("graphop::%s has no CPU implementation: inputs must be CUDA (ROCm) "
                           "tensors" % name)
    return _impl


cpp_ext = None      # the compiled extension module (csrc/torch_ext.cpp) when it is built, else None


def register_torch_ops():
    """Define torch.ops.graphop.* once.  If the compiled C++ extension graphop_cpp is built, loading
    it registers the namespace from C++ (TORCH_LIBRARY(graphop), csrc/torch_ext.cpp: the reference-style
    boundary over the same C ABI); otherwise the ops are defined here (CUDA key -> the ctypes-bound
    functions above, CPU key -> error)."""
    global _torch_lib, cpp_ext
    if _torch_lib is not None:
        return
    from . import _ext
    cpp_ext = _ext.load()
    if cpp_ext is not None:
        _torch_lib = cpp_ext
        return
    l = torch.library.Library("graphop", "DEF")
    g = globals()
    for name, schema in _SCHEMAS.items():
        l.define(name + schema)
        l.impl(name, g[name], "CUDA")
        l.impl(name, _cpu_refusal(name), "CPU")
    _torch_lib = l


register_torch_ops()
