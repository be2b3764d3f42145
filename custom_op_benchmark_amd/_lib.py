"""ctypes binding of libgraphop_hip.so (C ABI: include/graphop_hip.h).

There is NO fallback: if the library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C custom_op_benchmark_amd/csrc``), or a tensor is
not on a ROCm device, every op raises.  PyTorch is used only for device memory and the current
stream.
"""
import ctypes
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAPHOP_LIB") or os.path.join(_HERE, "libgraphop_hip.so")   # override: A/B builds
ABI_VERSION = 3

F32, F64 = 0, 1
_c64 = ctypes.c_int64
_vp = ctypes.c_void_p
_lib = None


class PlanInfo(ctypes.Structure):
    _fields_ = [("n_chunks", _c64), ("n_edges", _c64), ("n_segments", _c64), ("max_row", _c64),
                ("max_index", _c64), ("max_segment_len", _c64), ("rows_sorted", ctypes.c_int32),
                ("indptr_monotone", ctypes.c_int32), ("eid_identity", ctypes.c_int32),
                ("full_coverage", ctypes.c_int32), ("row_owned", ctypes.c_int32),
                ("has_idx32", ctypes.c_int32), ("dense_fill_pct", ctypes.c_int32),
                ("n_dense_blocks", _c64)]


class ProfileRec(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("kernel", ctypes.c_char * 48), ("calls", _c64), ("total_ms", ctypes.c_double),
                ("min_ms", ctypes.c_double), ("max_ms", ctypes.c_double)]


# name -> argtypes (all return int unless noted); mirrors include/graphop_hip.h one to one
_P = _vp
_SIGNATURES = {
    "graphop_tune": [ctypes.c_char_p, ctypes.c_int],
    "graphop_profile_enable": [ctypes.c_int],
    "graphop_profile_read": [ctypes.POINTER(ProfileRec), ctypes.c_int],
    "graphop_partition_csr_count": [_P, _c64, _c64, _P, _P],
    "graphop_partition_csr_fill": [_P, _P, _c64, _c64, _c64, _P, _P, _P],
    "graphop_plan_create": [_P, _P, _P, _P, _c64, _c64, _c64, _P, ctypes.POINTER(_vp)],
    "graphop_plan_info": [_P, ctypes.POINTER(PlanInfo)],
    "graphop_maskedmm_csr_forward": [ctypes.c_int] + [_P] * 7 + [_c64] * 6 + [_P, _P],
    "graphop_maskedmm_csr_backward": [ctypes.c_int] + [_P] * 13 + [_c64] * 7 + [_P, _P, _P],
    "graphop_sparse_softmax_forward": [ctypes.c_int] + [_P] * 5 + [_c64] * 3 + [_P, _c64, _P, _P],
    "graphop_sparse_softmax_backward": [ctypes.c_int] + [_P] * 6 + [_c64] * 3 + [_P, _c64, _P, _P],
    "graphop_vector_spmm_forward": [ctypes.c_int] + [_P] * 7 + [_c64] * 6 + [_P, _P],
    "graphop_vector_spmm_backward": [ctypes.c_int] + [_P] * 13 + [_c64] * 7 + [_P, _P, _P],
    "graphop_node_mul_edge_forward": [ctypes.c_int] + [_P] * 6 + [_c64] * 5 + [_P, _P],
    "graphop_node_mul_edge_backward": [ctypes.c_int] + [_P] * 8 + [_c64] * 5 + [_P, _P],
    "graphop_gather_rows": [ctypes.c_int, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_scatter_add_rows": [ctypes.c_int, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_attention_workspace_bytes": [ctypes.c_int, ctypes.c_int] + [_c64] * 5 + [_P, _P, _P,
                                                                                       ctypes.POINTER(_c64)],
    "graphop_attention_forward": [ctypes.c_int] + [_P] * 9 + [_c64] * 6 + [_P, _c64, _P, _P],
    "graphop_attention_backward": [ctypes.c_int] + [_P] * 17 + [_c64] * 7 + [_P, _c64, _P, _P, _P],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["graphop_abi_version", "graphop_last_error",
                                               "graphop_plan_destroy"])


def lib():
    """Load the HIP library (loudly)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "graphop: %s is missing -- the HIP extension has not been built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
                "custom_op_benchmark_amd/csrc`). There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        l.graphop_abi_version.restype = ctypes.c_int
        l.graphop_last_error.restype = ctypes.c_char_p
        l.graphop_plan_destroy.restype = None
        l.graphop_plan_destroy.argtypes = [_vp]
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
        if l.graphop_abi_version() != ABI_VERSION:
            raise RuntimeError("graphop: ABI version mismatch (library %d, binding %d)"
                               % (l.graphop_abi_version(), ABI_VERSION))
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().graphop_last_error()
        raise RuntimeError("graphop: " + (msg.decode() if msg else "error %d" % rc))


def ptr(t):
    return _vp(t.data_ptr()) if (t is not None and t.numel() > 0) else _vp(0)


def stream_of(t):
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    # the reference raises from AT_DISPATCH_FLOATING_TYPES (graphop_kernel.cu:291)
    raise RuntimeError('graphop: not implemented for \'%s\' (float32 / float64 only)' % t.dtype)


# ---- plans: cached per (row, indptr, eid, indices) tensor identity ------------------------------
class Plan:
    """Owner of a graphop_plan_t*.  Holds references to the tensors it was built from."""

    def __init__(self, row, indptr, eid, indices, n_index_bound):
        self.tensors = (row, indptr, eid, indices)
        handle = _vp(0)
        with torch.cuda.device(row.device):
            check(lib().graphop_plan_create(ptr(row), ptr(indptr), ptr(eid), ptr(indices),
                                            row.numel(), eid.numel(), int(n_index_bound),
                                            stream_of(row), ctypes.byref(handle)))
        self.handle = handle
        info = PlanInfo()
        check(lib().graphop_plan_info(handle, ctypes.byref(info)))
        self.info = info
        self._finalizer = weakref.finalize(self, lib().graphop_plan_destroy, handle)

    def __repr__(self):
        i = self.info
        return ("Plan(chunks=%d edges=%d segments=%d row_owned=%d eid_identity=%d full_coverage=%d "
                "max_seg=%d idx32=%d)" % (i.n_chunks, i.n_edges, i.n_segments, i.row_owned,
                                          i.eid_identity, i.full_coverage, i.max_segment_len,
                                          i.has_idx32))


_plan_cache = {}
_PLAN_CACHE_MAX = 64


def _key(*ts):
    return tuple((t.data_ptr(), t.numel(), t._version, t.device.index) if t is not None else None
                 for t in ts)


def get_plan(row, indptr, eid, indices=None, n_index_bound=0):
    """Plan for one CSR orientation, cached while the tensors are unchanged.  A request without
    ``indices`` (softmax, node_mul_edge) reuses any plan of the same (row, indptr, eid)."""
    k3 = _key(row, indptr, eid)
    entry = _plan_cache.get(k3)
    if entry is None:
        if len(_plan_cache) >= _PLAN_CACHE_MAX:
            _plan_cache.pop(next(iter(_plan_cache)))
        entry = _plan_cache[k3] = {}
    if indices is None:
        p = next(iter(entry.values()), None)
        ki = None
    else:
        ki = _key(indices)
        p = entry.get(ki)
    if p is None:
        p = Plan(row, indptr, eid, indices, n_index_bound)
        entry.pop(None, None)          # a plan with indices supersedes the index-less one
        entry[ki] = p
    elif indices is not None and n_index_bound > 0 and p.info.max_index >= n_index_bound:
        raise RuntimeError("graphop: indices holds %d but the gathered tensor has only %d rows"
                           % (p.info.max_index, n_index_bound))
    return p


def tune(key, value):
    """Set a tuning knob (include/graphop_hip.h: graphop_tune)."""
    check(lib().graphop_tune(key.encode(), int(value)))


def profile_enable(on=True):
    """Bracket every hot-path kernel launch with hipEvents (measurement aid, bench.py)."""
    check(lib().graphop_profile_enable(1 if on else 0))


def profile_read():
    """-> {tag: dict(calls, total_ms, mean_ms, min_ms, max_ms)}; synchronises and clears the log."""
    buf = (ProfileRec * 64)()
    n = lib().graphop_profile_read(buf, 64)
    if n < 0:
        check(1)
    out = {}
    for r in buf[:min(n, 64)]:
        out[r.name.decode()] = dict(kernel=r.kernel.decode(), calls=int(r.calls), total_ms=r.total_ms, mean_ms=r.total_ms / max(1, r.calls),
                                    min_ms=r.min_ms, max_ms=r.max_ms)
    return out


def clear_plan_cache():
    _plan_cache.clear()


# ---- partition_csr on the device ------------------------------------------------------------------
def partition_csr_device(indptr, chunk_size):
    ip = indptr.to(torch.int64).contiguous()
    n = ip.numel() - 1
    dev = ip.device
    with torch.cuda.device(dev):
        first = torch.empty(n + 1, dtype=torch.int64, device=dev)
        check(lib().graphop_partition_csr_count(ptr(ip), n, chunk_size, _vp(first.data_ptr()),
                                                stream_of(ip)))
        c = int(first[-1].item())
        row = torch.empty(c, dtype=torch.int64, device=dev)
        out = torch.empty(c + 1, dtype=torch.int64, device=dev)
        check(lib().graphop_partition_csr_fill(ptr(ip), _vp(first.data_ptr()), n, chunk_size, c,
                                               ptr(row), _vp(out.data_ptr()), stream_of(ip)))
    return row, out


# ---- halo pack / unpack (dist.py) ------------------------------------------------------------------
def gather_rows(src, idx, out=None):
    """out[i] = src[idx[i]] (HIP pack kernel; `out` = persistent send buffer)."""
    if out is None:
        out = src.new_empty((idx.numel(),) + tuple(src.shape[1:]))
    row = src[0].numel() if src.size(0) else 0
    with torch.cuda.device(src.device):
        check(lib().graphop_gather_rows(dtype_code(src), ptr(src), ptr(idx), ptr(out), idx.numel(),
                                        src.size(0), row, stream_of(src)))
    return out


def scatter_add_rows(dst, idx, src):
    """dst[idx[i]] += src[i] in place (idx may repeat)."""
    row = dst[0].numel() if dst.size(0) else 0
    with torch.cuda.device(dst.device):
        check(lib().graphop_scatter_add_rows(dtype_code(dst), ptr(src), ptr(idx), ptr(dst), idx.numel(),
                                             dst.size(0), row, stream_of(dst)))
    return dst
