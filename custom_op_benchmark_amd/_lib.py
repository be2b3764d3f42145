"""ctypes binding of libgraphop_hip.so (C ABI: include/graphop_hip.h).

There is NO fallback: if the library has not been built (``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C custom_op_benchmark_amd/csrc``), or a tensor is
not on a ROCm device, every op raises.  PyTorch is used only for device memory and the current
stream.
"""
import atexit
import collections
import ctypes
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRAPHOP_LIB") or os.path.join(_HERE, "libgraphop_hip.so")   # override: A/B builds
ABI_VERSION = 7

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
                ("sorted_in_rows", ctypes.c_int32), ("n_dense_blocks", _c64), ("max_row_gap", _c64),
                ("n_geometry_fallbacks", _c64)]


class SweepInfo(ctypes.Structure):
    _fields_ = [("win_cols", _c64), ("W", ctypes.c_int32), ("T", ctypes.c_int32), ("V", ctypes.c_int32),
                ("n_dealt", ctypes.c_int32)]


ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p)
FREE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class ProfileRec(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("kernel", ctypes.c_char * 48), ("calls", _c64), ("total_ms", ctypes.c_double),
                ("min_ms", ctypes.c_double), ("max_ms", ctypes.c_double)]


# name -> argtypes (all return int unless noted); mirrors include/graphop_hip.h one to one
_P = _vp
_SIGNATURES = {
    "graphop_tune": [ctypes.c_char_p, ctypes.c_int],
    "graphop_tune_reset": [],
    "graphop_check_device_errors": [],
    "graphop_tune_get": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "graphop_profile_enable": [ctypes.c_int],
    "graphop_profile_read": [ctypes.POINTER(ProfileRec), ctypes.c_int],
    "graphop_partition_csr_count": [_P, _c64, _c64, _P, _P],
    "graphop_partition_csr_fill": [_P, _P, _c64, _c64, _c64, _P, _P, _P],
    "graphop_plan_create": [_P, _P, _P, _P, _c64, _c64, _c64, _P, ctypes.POINTER(_vp)],
    "graphop_plan_info": [_P, ctypes.POINTER(PlanInfo)],
    "graphop_set_allocator": [ALLOC_FN, FREE_FN],
    "graphop_plan_prepare": [_P, ctypes.c_int, _c64, _c64, _c64, ctypes.c_int, _P],
    "graphop_plan_n_sweeps": [_P],
    "graphop_plan_sweep_info": [_P, ctypes.c_int, ctypes.POINTER(SweepInfo)],
    "graphop_plan_array": [_P, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_c64)],
    "graphop_plan_import": [_P] * 4 + [ctypes.POINTER(PlanInfo)] + [_P] * 4 + [_c64] + [_P] * 4 + [ctypes.POINTER(_vp)],
    "graphop_plan_import_sweep": [_P, ctypes.POINTER(SweepInfo), _P, _P, _P, _P],
    "graphop_plan_sweep_dealt": [_P, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)],
    "graphop_plan_sweep_build_dealt": [_P, ctypes.POINTER(SweepInfo), ctypes.c_int32, ctypes.c_int32, _P],
    "graphop_maskedmm_csr_forward": [ctypes.c_int] + [_P] * 7 + [_c64] * 6 + [_P, _P],
    "graphop_maskedmm_csr_forward_partial": [ctypes.c_int] + [_P] * 7 + [_c64] * 7 + [_P],
    "graphop_maskedmm_csr_backward": [ctypes.c_int] + [_P] * 13 + [_c64] * 7 + [_P, _P, _P],
    "graphop_sparse_softmax_forward": [ctypes.c_int] + [_P] * 5 + [_c64] * 3 + [_P, _c64, _P, _P],
    "graphop_sparse_softmax_backward": [ctypes.c_int] + [_P] * 6 + [_c64] * 3 + [_P, _c64, _P, _P],
    "graphop_vector_spmm_forward": [ctypes.c_int] + [_P] * 7 + [_c64] * 6 + [_P, _P],
    "graphop_vector_spmm_backward": [ctypes.c_int] + [_P] * 13 + [_c64] * 7 + [_P, _P, _P],
    "graphop_spmm_pair_supported": [ctypes.c_int] + [_c64] * 5 + [_P],
    "graphop_spmm_pair": [ctypes.c_int] + [_P] * 9 + [_c64] * 6 + [_P, _P],
    "graphop_interleave_pairs": [ctypes.c_int, _P, _P, _P, _c64, _P],
    "graphop_node_mul_edge_forward": [ctypes.c_int] + [_P] * 6 + [_c64] * 5 + [_P, _P],
    "graphop_node_mul_edge_backward": [ctypes.c_int] + [_P] * 8 + [_c64] * 5 + [_P, _P],
    "graphop_gather_rows": [ctypes.c_int, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_scatter_add_rows": [ctypes.c_int, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_add_rows_unique": [ctypes.c_int, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_add_rows_grouped": [ctypes.c_int, _P, _P, _P, _P, _P, _c64, _c64, _c64, _P],
    "graphop_attention_workspace_bytes": [ctypes.c_int, ctypes.c_int] + [_c64] * 5 + [_P, _P, _P,
                                                                                       ctypes.POINTER(_c64)],
    "graphop_attention_backward_is_fused": [ctypes.c_int] + [_c64] * 5 + [_P, _P, _P, ctypes.POINTER(ctypes.c_int)],
    "graphop_attention_forward": [ctypes.c_int] + [_P] * 9 + [_c64] * 6 + [_P, _c64, _P, _P],
    "graphop_attention_backward": [ctypes.c_int] + [_P] * 17 + [_c64] * 7 + [_P, _c64, _P, _P, _P],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["graphop_abi_version", "graphop_last_error",
                                               "graphop_plan_destroy", "graphop_memory_bytes", "graphop_tune_key"])


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
        l.graphop_memory_bytes.restype = ctypes.c_int64
        l.graphop_memory_bytes.argtypes = []
        l.graphop_tune_key.restype = ctypes.c_char_p
        l.graphop_tune_key.argtypes = [ctypes.c_int]
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = ctypes.c_int
            fn.argtypes = argtypes
        if l.graphop_abi_version() != ABI_VERSION:
            raise RuntimeError("graphop: ABI version mismatch (library %d, binding %d)"
                               % (l.graphop_abi_version(), ABI_VERSION))
        if os.environ.get("GRAPHOP_TORCH_ALLOC", "1") != "0":
            l.graphop_set_allocator(_ALLOC_CB, _FREE_CB)
            atexit.register(_drop_allocator, l)
        _lib = l
    return _lib


# Plan memory comes from torch's caching allocator (include/graphop_hip.h: graphop_set_allocator):
# visible to torch.cuda's accounting, reclaimable, and freed stream-ordered (no device-wide stall).
_live_blocks = {}


def _alloc_cb(nbytes, device, stream):
    # The block is allocated ON the stream the library names (torch's caching allocator recycles memory in
    # allocation-stream order), and a block is only handed back once the device is idle (below): a plan may have
    # been used from any stream since, and the hipFree this hook replaces synchronised the device as well.
    try:
        dev = torch.device("cuda", device)
        if stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                t = torch.empty(max(1, int(nbytes)), dtype=torch.uint8, device=dev)
        else:
            t = torch.empty(max(1, int(nbytes)), dtype=torch.uint8, device=dev)
    except RuntimeError:        # out of memory: the library reports it
        return None
    _live_blocks[t.data_ptr()] = t
    return t.data_ptr()


def _free_cb(p):
    t = _live_blocks.pop(p, None)
    if t is not None:
        try:
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize(t.device)
        except Exception:       # (interpreter shutdown, device lost: drop the block regardless)
            pass


_ALLOC_CB, _FREE_CB = ALLOC_FN(_alloc_cb), FREE_FN(_free_cb)


def _drop_allocator(l):
    # interpreter shutdown: plans destroyed later must not call back into Python
    l.graphop_set_allocator(ctypes.cast(None, ALLOC_FN), ctypes.cast(None, FREE_FN))


def plan_memory_bytes():
    """Device bytes currently held by plans, their window structures and id layouts (whichever binding
    created them: the library keeps the count, include/graphop_hip.h: graphop_memory_bytes)."""
    return int(lib().graphop_memory_bytes())


def check(rc):
    if rc != 0:
        msg = lib().graphop_last_error()
        raise RuntimeError("graphop: " + (msg.decode() if msg else "error %d" % rc))


def check_errors(sync=True):
    """Raise if a kernel of an earlier launch reported a failure through the device error record (a walk kernel whose
    hand-over spin expired: include/graphop_hip.h, graphop_check_device_errors) and CLEAR the record -- it is sticky
    (ABI 7): until this call acknowledges it, every op entry point fails with the same message.  sync=True waits for
    the device first, so every launch made so far is covered; sync=False only looks (one host-memory load: the steps
    call it on their way out, so an abort in the LAST launch of a step is reported at the latest by the next step)."""
    if sync and torch.cuda.is_available():
        torch.cuda.synchronize()
    check(lib().graphop_check_device_errors())


def ptr(t):
    # (a plain int / None: ctypes converts it for a c_void_p parameter without building an object)
    return t.data_ptr() if (t is not None and t.numel() > 0) else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """hipStream_t of torch's current stream on t's device (raw handle, no Stream object)."""
    idx = t.device.index
    if _raw_stream is not None and idx is not None:
        return _raw_stream(idx)
    return torch.cuda.current_stream(t.device).cuda_stream


class device_guard:
    """`with torch.cuda.device(d)` without its cost when d already is the current device (the
    per-op hot path: eight calls per step)."""
    __slots__ = ("idx", "prev")

    def __init__(self, device):
        self.idx = device.index

    def __enter__(self):
        self.prev = torch.cuda.current_device()
        if self.idx is not None and self.idx != self.prev:
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.idx is not None and self.idx != self.prev:
            torch.cuda.set_device(self.prev)
        return False


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    # the reference raises from AT_DISPATCH_FLOATING_TYPES (graphop_kernel.cu:291)
    raise RuntimeError('graphop: not implemented for \'%s\' (float32 / float64 only)' % t.dtype)


# ---- plans: cached per (row, indptr, eid, indices) tensor identity ------------------------------
_PLAN_ARRAYS = (("seg_chunk", torch.int64), ("idx32", torch.int32), ("eid32", torch.int32),
                ("long_segs", torch.int32), ("blk_seg", torch.int32), ("seg_e0", torch.int32),
                ("seg_row", torch.int32))
_SWEEP_ARRAYS = ("vr_row", "wp_lo", "wp_hi")


class Plan:
    """Owner of a graphop_plan_t*.  Holds references to the tensors it was built from."""

    def __init__(self, row, indptr, eid, indices, n_index_bound, state=None):
        self.tensors = (row, indptr, eid, indices)
        handle = _vp(0)
        with torch.cuda.device(row.device):
            if state is None:
                check(lib().graphop_plan_create(ptr(row), ptr(indptr), ptr(eid), ptr(indices),
                                                row.numel(), eid.numel(), int(n_index_bound),
                                                stream_of(row), ctypes.byref(handle)))
            else:
                handle = self._import(row, indptr, eid, indices, state)
        self.handle = handle
        info = PlanInfo()
        check(lib().graphop_plan_info(handle, ctypes.byref(info)))
        self.info = info
        self._finalizer = weakref.finalize(self, lib().graphop_plan_destroy, handle)

    # ---- persistence (graphs.save_graph / load_graph) -------------------------------------------
    def _array(self, name, dtype, sweep=-1):
        p, n = _vp(0), _c64(0)
        check(lib().graphop_plan_array(self.handle, name.encode(), sweep, ctypes.byref(p), ctypes.byref(n)))
        if not p.value or n.value == 0:
            return None
        dev = self.tensors[1].device
        out = torch.empty(n.value // torch.empty((), dtype=dtype).element_size(), dtype=dtype, device=dev)
        with torch.cuda.device(dev):
            torch.cuda.current_stream().synchronize()
            _memcpy_d2d(out.data_ptr(), p.value, n.value)
        return out.cpu()

    def export_state(self):
        """Everything needed to re-create this plan without analysing the graph: the info record,
        the derived arrays and every window structure built so far (CPU tensors)."""
        st = {"info": {f: getattr(self.info, f) for f, _ in PlanInfo._fields_}, "sweeps": []}
        for name, dt in _PLAN_ARRAYS:
            st[name] = self._array(name, dt)
        for i in range(lib().graphop_plan_n_sweeps(self.handle)):
            si = SweepInfo()
            check(lib().graphop_plan_sweep_info(self.handle, i, ctypes.byref(si)))
            sw = {"W": si.W, "T": si.T, "V": si.V, "win_cols": si.win_cols, "dealt": []}
            for name in _SWEEP_ARRAYS:
                sw[name] = self._array(name, torch.int32, i)
            for j in range(si.n_dealt):      # window-major id layouts: only their geometry travels
                L, K = ctypes.c_int32(0), ctypes.c_int32(0)
                check(lib().graphop_plan_sweep_dealt(self.handle, i, j, ctypes.byref(L), ctypes.byref(K)))
                sw["dealt"].append((int(L.value), int(K.value)))
            st["sweeps"].append(sw)
        return st

    def _import(self, row, indptr, eid, indices, state):
        dev = indptr.device
        info = PlanInfo(**state["info"])
        arrs = {name: (state.get(name).to(dev) if state.get(name) is not None else None) for name, _ in _PLAN_ARRAYS}
        handle = _vp(0)
        n_long = arrs["long_segs"].numel() if arrs["long_segs"] is not None else 0
        check(lib().graphop_plan_import(ptr(row), ptr(indptr), ptr(eid), ptr(indices), ctypes.byref(info),
                                        ptr(arrs["seg_chunk"]), ptr(arrs["idx32"]), ptr(arrs["eid32"]),
                                        ptr(arrs["long_segs"]), n_long, ptr(arrs["blk_seg"]), ptr(arrs["seg_e0"]),
                                        ptr(arrs["seg_row"]), stream_of(indptr), ctypes.byref(handle)))
        for sw in state.get("sweeps", []):
            si = SweepInfo(win_cols=sw["win_cols"], W=sw["W"], T=sw["T"], V=sw["V"], n_dealt=0)
            a = [sw[name].to(dev) for name in _SWEEP_ARRAYS]
            check(lib().graphop_plan_import_sweep(handle, ctypes.byref(si), ptr(a[0]), ptr(a[1]), ptr(a[2]),
                                                  stream_of(indptr)))
            for L, K in sw.get("dealt", []):
                check(lib().graphop_plan_sweep_build_dealt(handle, ctypes.byref(si), L, K, stream_of(indptr)))
        torch.cuda.current_stream(dev).synchronize()     # the staging tensors die here
        return handle

    def prepare(self, dtype, n_table_rows, h, d, fused=True):
        """Build every cached window structure ops on (n_table_rows, h, d) tensors will use.
        fused: False / True (the plan may serve either side of the fused passes) / 2 (row-major side
        only) / 3 (column-major side only)."""
        with torch.cuda.device(self.tensors[1].device):
            check(lib().graphop_plan_prepare(self.handle, dtype, int(n_table_rows), int(h), int(d),
                                             int(fused), stream_of(self.tensors[1])))

    def refresh_info(self):
        """Re-read the info record (n_geometry_fallbacks is a live count: include/graphop_hip.h)."""
        check(lib().graphop_plan_info(self.handle, ctypes.byref(self.info)))
        return self.info

    def __repr__(self):
        i = self.info
        return ("Plan(chunks=%d edges=%d segments=%d row_owned=%d eid_identity=%d full_coverage=%d "
                "max_seg=%d idx32=%d)" % (i.n_chunks, i.n_edges, i.n_segments, i.row_owned,
                                          i.eid_identity, i.full_coverage, i.max_segment_len,
                                          i.has_idx32))


def _memcpy_d2d(dst, src, nbytes):
    hip = _hip_runtime()
    rc = hip.hipMemcpy(_vp(dst), _vp(src), ctypes.c_size_t(nbytes), 3)   # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError("graphop: hipMemcpy failed (%d)" % rc)


_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
    return _hip


# ---- plan cache: least-recently-used, keyed by tensor identity ----------------------------------
# Fast path: the row tensor of an orientation remembers its plans (compared by object identity and
# version counters: no data_ptr() calls per op).  Slow path: an LRU dict keyed by storage address,
# so equal views of the same arrays share a plan.
_plan_cache = collections.OrderedDict()
_PLAN_CACHE_MAX = 64
# ... and by bytes: plans with their window structures and window-major id copies hold 40-45 B per
# edge (Reddit-shape: 4.9 GB for both orientations), so the cache also evicts least-recently-used
# graphs while plan memory exceeds this many bytes (GRAPHOP_PLAN_CACHE_GB, default 96 of the 288 GB).
_PLAN_CACHE_BYTES = int(float(os.environ.get("GRAPHOP_PLAN_CACHE_GB", "96")) * 2**30)


def _evict_lru():
    """Drop the least recently used graph from both cache levels (its plans are destroyed once no
    op holds them; the next op on that graph rebuilds them)."""
    _, ent = _plan_cache.popitem(last=False)
    for p in ent.values():
        p.tensors[0].__dict__.pop("_graphop_plans", None)


def _key(*ts):
    return tuple((t.data_ptr(), t.numel(), t._version, t.device.index) if t is not None else None
                 for t in ts)


def _remember(row, indptr, eid, indices, plan):
    lst = row.__dict__.setdefault("_graphop_plans", [])
    lst.append((indptr, eid, indices, (row._version, indptr._version, eid._version,
                                       indices._version if indices is not None else -1), plan))
    if len(lst) > 4:
        del lst[0]


def get_plan(row, indptr, eid, indices=None, n_index_bound=0, state=None):
    """Plan for one CSR orientation, cached while the tensors are unchanged.  A request without
    ``indices`` (softmax, node_mul_edge) reuses any plan of the same (row, indptr, eid).
    ``state``: an export_state() record to re-create the plan from instead of analysing the graph."""
    lst = row.__dict__.get("_graphop_plans")
    if lst is not None and state is None:
        for ip, ei, ix, ver, p in lst:
            if ip is indptr and ei is eid and (indices is None or ix is indices) and \
                    ver[0] == row._version and ver[1] == indptr._version and ver[2] == eid._version and \
                    (indices is None or ver[3] == indices._version):
                if indices is not None and n_index_bound > 0 and p.info.max_index >= n_index_bound:
                    raise RuntimeError("graphop: indices holds %d but the gathered tensor has only %d rows"
                                       % (p.info.max_index, n_index_bound))
                return p
    k3 = _key(row, indptr, eid)
    entry = _plan_cache.get(k3)
    if entry is None:
        while _plan_cache and (len(_plan_cache) >= _PLAN_CACHE_MAX or plan_memory_bytes() > _PLAN_CACHE_BYTES):
            _evict_lru()
        entry = _plan_cache[k3] = {}
    else:
        _plan_cache.move_to_end(k3)
    if indices is None:
        p = next(iter(entry.values()), None)
        ki = None
    else:
        ki = _key(indices)
        p = entry.get(ki)
    if p is None or state is not None:
        p = Plan(row, indptr, eid, indices, n_index_bound, state)
        entry.pop(None, None)          # a plan with indices supersedes the index-less one
        entry[ki] = p
    elif indices is not None and n_index_bound > 0 and p.info.max_index >= n_index_bound:
        raise RuntimeError("graphop: indices holds %d but the gathered tensor has only %d rows"
                           % (p.info.max_index, n_index_bound))
    _remember(row, indptr, eid, indices, p)
    return p


def release_plans(*rows):
    """Drop the plans attached to these row tensors (both cache levels); their device memory goes
    back to torch's allocator as soon as no op holds them."""
    for row in rows:
        lst = row.__dict__.pop("_graphop_plans", None) or []
        plans = {id(e[4]) for e in lst}
        for k in [k for k, ent in _plan_cache.items() if any(id(p) in plans for p in ent.values())]:
            del _plan_cache[k]
        ext = _cpp_ext()
        if ext is not None and row.is_cuda:
            ext.release_plans(row)      # the compiled extension keeps its own entries (csrc/torch_ext.cpp)


def _cpp_ext():
    from . import _ext
    return _ext._mod        # only if it has been loaded: nothing to release otherwise


def tune(key, value):
    """Set a tuning knob (include/graphop_hip.h: graphop_tune)."""
    check(lib().graphop_tune(key.encode(), int(value)))


def tune_get(key):
    """Current value of a tuning knob."""
    v = ctypes.c_int(0)
    check(lib().graphop_tune_get(key.encode(), ctypes.byref(v)))
    return int(v.value)


def tune_snapshot():
    """{knob: value} of every tuning knob (include/graphop_hip.h: graphop_tune_key / graphop_tune_get)."""
    out, i = {}, 0
    while True:
        k = lib().graphop_tune_key(i)
        if not k:
            return out
        out[k.decode()] = tune_get(k.decode())
        i += 1


def tune_reset():
    """Every knob back to its default (include/graphop_hip.h: graphop_tune_reset)."""
    check(lib().graphop_tune_reset())


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
    """Drop every cached plan of both bindings (this one and the compiled extension's)."""
    for ent in _plan_cache.values():
        for p in ent.values():
            p.tensors[0].__dict__.pop("_graphop_plans", None)
    _plan_cache.clear()
    ext = _cpp_ext()
    if ext is not None:
        ext.clear_plan_cache()


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


def group_rows(idx):
    """Grouping of a row-index list for add_rows_grouped: (ptr, rows, pos) with rows = the distinct values of idx in
    ascending order, pos[ptr[g]:ptr[g + 1]] = the positions of rows[g] in idx, in their original order (stable)."""
    order = torch.argsort(idx, stable=True)
    rows, counts = torch.unique_consecutive(idx[order], return_counts=True)
    ptr_ = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=idx.device)
    torch.cumsum(counts, 0, out=ptr_[1:])
    return ptr_, rows.contiguous(), order.contiguous()


def add_rows_grouped(dst, groups, src):
    """dst[rows[g]] += sum of src[pos[p]] over the group's positions, one launch (include/graphop_hip.h:
    graphop_add_rows_grouped); groups = group_rows(idx)."""
    ptr_, rows, pos = groups
    row = dst[0].numel() if dst.size(0) else 0
    with torch.cuda.device(dst.device):
        check(lib().graphop_add_rows_grouped(dtype_code(dst), ptr(src), ptr(ptr_), ptr(rows), ptr(pos), ptr(dst),
                                             rows.numel(), dst.size(0), row, stream_of(dst)))
    return dst


def scatter_add_rows(dst, idx, src, unique_runs=None):
    """dst[idx[i]] += src[i] in place.  idx may repeat -> float atomics; with unique_runs = the
    lengths of consecutive idx runs that hold no repeats (the rows served to each peer) the runs are
    added one after the other with plain read-add-write kernels."""
    row = dst[0].numel() if dst.size(0) else 0
    with torch.cuda.device(dst.device):
        if unique_runs is None:
            check(lib().graphop_scatter_add_rows(dtype_code(dst), ptr(src), ptr(idx), ptr(dst), idx.numel(),
                                                 dst.size(0), row, stream_of(dst)))
        else:
            o = 0
            for n in unique_runs:
                if n:
                    check(lib().graphop_add_rows_unique(dtype_code(dst), ptr(src[o:o + n]), ptr(idx[o:o + n]),
                                                        ptr(dst), n, dst.size(0), row, stream_of(dst)))
                o += n
    return dst
