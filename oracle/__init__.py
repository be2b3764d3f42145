"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU checker for the graph-attention hot path.  Two independent restatements:

* ``oracle.graphop_oracle.c`` (loaded here through ctypes): a literal, serial C
  restatement of the reference's device kernels (``graphop/graphop_kernel.cu``),
  exposed below with the reference module's eight names and positional
  signatures (``graphop/graphop.cpp:216-225``) on CPU torch tensors.
* ``oracle.torch_path``: the stock-PyTorch gather/scatter formulation the
  reference harness asserts against (``wrapper.py:155-157,218,274``), also the
  ``cpu_baseline`` leg of ``bench.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``custom_op_benchmark_amd`` never does.

Parity status: PINNED -- against fixtures produced by importing the reference's
own Python (``tests/golden/gen_golden.py``; fixtures in ``tests/golden/*.npz``).
The reference's CUDA kernels themselves cannot be compiled or run anywhere in
this pipeline (no nvcc / NVIDIA GPU, THC headers gone from torch 2.10).
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgraphop_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "graphop_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgraphop_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_partition_csr.restype = ctypes.c_int64
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _i(v):
    return ctypes.c_int64(int(v))


def _suf(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise RuntimeError("oracle: value tensors must be float32 or float64, got %s" % t.dtype)


def _chk(*ts):
    for t in ts:
        assert t.device.type == "cpu" and t.is_contiguous(), "oracle works on contiguous CPU tensors"


def _idx(*ts):
    for t in ts:
        assert t.dtype == torch.int64, "index tensors are int64 (graphop_kernel.cu:293-296)"


# --- the reference module surface (graphop.cpp:216-225), on CPU tensors -----------------------

def maskedmm_csr_forward(row, indptr, eid, indices, A, B):
    _chk(row, indptr, eid, indices, A, B); _idx(row, indptr, eid, indices)
    e, d = eid.size(0), A.size(-1)
    h = 1 if A.dim() == 2 else A.size(1)                       # graphop_kernel.cu:283
    y = A.new_empty((e,) if h == 1 else (e, h))                # :284
    getattr(lib(), "oracle_maskedmm_csr_forward_" + _suf(A))(
        _p(row), _p(indptr), _p(eid), _p(indices), _p(A), _p(B), _p(y),
        _i(row.size(0)), _i(e), _i(d), _i(h))
    return y


def maskedmm_csr_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B, dy):
    _chk(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c, A, B, dy)
    d = A.size(-1)
    h = dy.size(1) if dy.dim() == 2 else 1                     # :373
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    getattr(lib(), "oracle_maskedmm_csr_backward_" + _suf(A))(
        _p(row), _p(indptr_r), _p(eid_r), _p(indices_r), _p(col), _p(indptr_c), _p(eid_c),
        _p(indices_c), _p(A), _p(B), _p(dy), _p(dA), _p(dB),
        _i(row.size(0)), _i(col.size(0)), _i(A.size(0)), _i(B.size(0)), _i(d), _i(h))
    return [dA, dB]


def _n_scratch(row, eid):
    # The reference sizes its scratch by eid.size(0) ("n <= e", :420); any size that covers
    # max(row)+1 gives the same observable result.
    return max(int(eid.size(0)), int(row.max()) + 1 if row.numel() else 0, 1)


def sparse_softmax_forward(row, indptr, eid, x):
    _chk(row, indptr, eid, x); _idx(row, indptr, eid)
    h = x.size(1) if x.dim() == 2 else 1
    n = _n_scratch(row, eid)
    y = torch.empty_like(x)
    smax, ssum = x.new_empty(n * h), x.new_empty(n * h)
    getattr(lib(), "oracle_sparse_softmax_forward_" + _suf(x))(
        _p(row), _p(indptr), _p(eid), _p(x), _p(y), _p(smax), _p(ssum),
        _i(row.size(0)), _i(eid.size(0)), _i(n), _i(h))
    return y


def sparse_softmax_backward(row, indptr, eid, y, dy):
    _chk(row, indptr, eid, y, dy); _idx(row, indptr, eid)
    h = dy.size(1) if dy.dim() == 2 else 1
    n = _n_scratch(row, eid)
    dx = torch.empty_like(dy)
    agg = dy.new_empty(n * h)
    getattr(lib(), "oracle_sparse_softmax_backward_" + _suf(y))(
        _p(row), _p(indptr), _p(eid), _p(y), _p(dy), _p(dx), _p(agg),
        _i(row.size(0)), _i(eid.size(0)), _i(n), _i(h))
    return dx


def vector_spmm_forward(row, indptr, eid, indices, edata, x):
    _chk(row, indptr, eid, indices, edata, x); _idx(row, indptr, eid, indices)
    h = edata.size(1) if edata.dim() == 2 else 1               # :520
    d = x.size(-1)
    y = torch.empty_like(x)                                    # zeros_like(x), :527
    # the reference writes y[row[c]] without a bound (it assumes #rows == #x rows): as a checker,
    # refuse the shapes for which that is an out-of-bounds write instead of corrupting the heap
    if row.numel() and int(row.max()) >= x.size(0):
        raise ValueError("oracle.vector_spmm_forward: row id %d but y = zeros_like(x) has %d rows"
                         % (int(row.max()), x.size(0)))
    getattr(lib(), "oracle_vector_spmm_forward_" + _suf(x))(
        _p(row), _p(indptr), _p(eid), _p(indices), _p(edata), _p(x), _p(y),
        _i(row.size(0)), _i(x.size(0)), _i(d), _i(h))
    return y


def vector_spmm_backward(row, indptr, eid, indices, col, indptr_t, eid_t, indices_t, edata, dy, x):
    # NB argument order: ..., edata, dy, x (graphop.cpp:190-201)
    _chk(row, indptr, eid, indices, col, indptr_t, eid_t, indices_t, edata, dy, x)
    h = edata.size(1) if edata.dim() == 2 else 1
    d = x.size(-1)
    dedata, dx = torch.empty_like(edata), torch.empty_like(x)
    getattr(lib(), "oracle_vector_spmm_backward_" + _suf(x))(
        _p(row), _p(indptr), _p(eid), _p(indices), _p(col), _p(indptr_t), _p(eid_t),
        _p(indices_t), _p(edata), _p(dy), _p(x), _p(dedata), _p(dx),
        _i(row.size(0)), _i(col.size(0)), _i(edata.size(0)), _i(x.size(0)), _i(d), _i(h))
    return [dedata, dx]                                        # :599


def node_mul_edge_forward(row, indptr, eid, A, B):
    _chk(row, indptr, eid, A, B); _idx(row, indptr, eid)
    e, d = eid.size(0), A.size(-1)
    h = 1 if A.dim() == 2 else A.size(1)
    y = A.new_empty((e,) if h == 1 else (e, h))
    getattr(lib(), "oracle_node_mul_edge_forward_" + _suf(A))(
        _p(row), _p(indptr), _p(eid), _p(A), _p(B), _p(y), _i(row.size(0)), _i(e), _i(d), _i(h))
    return y


def node_mul_edge_backward(row, indptr, eid, A, B, dy):
    _chk(row, indptr, eid, A, B, dy); _idx(row, indptr, eid)
    d = A.size(-1)
    h = dy.size(1) if dy.dim() == 2 else 1
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    getattr(lib(), "oracle_node_mul_edge_backward_" + _suf(A))(
        _p(row), _p(indptr), _p(eid), _p(A), _p(B), _p(dy), _p(dA), _p(dB),
        _i(row.size(0)), _i(A.size(0)), _i(B.size(0)), _i(d), _i(h))
    return [dA, dB]


def partition_csr(indptr, chunk_size=32):
    """part_csr.py:13-27 restated in C (two calls: count, then fill)."""
    ip = indptr.detach().cpu().contiguous().to(torch.int64)
    n = ip.numel() - 1
    c = lib().oracle_partition_csr(_p(ip), _i(n), _i(chunk_size), ctypes.c_void_p(0), ctypes.c_void_p(0))
    row = torch.empty(c, dtype=torch.int64)
    out = torch.empty(c + 1, dtype=torch.int64)
    lib().oracle_partition_csr(_p(ip), _i(n), _i(chunk_size), _p(row), _p(out))
    return row.to(indptr.device), out.to(indptr.device)


EXPORTS = ["maskedmm_csr_forward", "maskedmm_csr_backward", "node_mul_edge_forward",
           "node_mul_edge_backward", "sparse_softmax_forward", "sparse_softmax_backward",
           "vector_spmm_forward", "vector_spmm_backward"]
