"""Where the column-major SpMM-type passes of a node-range shard lose their time (papers100M shape, one 1/8 shard:
28.9 M output rows -- 13.9 M own columns of ~13 slots, 15 M halo columns of 1-2 slots -- gathering 512-B rows of a
13.9 M-row table): the per-chunk loop (k_spmm_f32) against the slot-walking form (k_spmm_flat_f32), each with the
per-slot scalars in storage order (identity eid) and behind a random permutation (what the column orientation sees).
    python tools/short_rows_experiment.py            (one MI355X; ~35 GB of HBM)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
n_tab, n_own, n_halo, d = 13_882_494, 13_882_494, 14_988_579, 128
deg = torch.cat([torch.randint(6, 21, (n_own,), device=dev), 1 + (torch.rand(n_halo, device=dev) < 0.35).long()])
indptr = torch.zeros(n_own + n_halo + 1, dtype=torch.int64, device=dev)
indptr[1:] = torch.cumsum(deg, 0)
E = int(indptr[-1])
indices = torch.randint(0, n_tab, (E,), device=dev)
row, ptr = _lib.partition_csr_device(indptr, 32)
n_out = n_own + n_halo
print("E = %d, chunks = %d (%.2f slots per chunk), output rows = %d" % (E, row.numel(), E / row.numel(), n_out), flush=True)
w = torch.rand(E, device=dev)
X = torch.rand(n_tab, d, device=dev)
out = torch.empty(n_out, d, device=dev)
L = _lib.lib()


def run(eid, tag):
    with _lib.device_guard(dev):
        plan = _lib.get_plan(row, ptr, eid, indices, n_tab)
        st = _lib.stream_of(X)
        for flat, cpg in ((0, 128), (1, 128), (1, 32), (0, 128), (1, 128)):
            _lib.tune("spmm_flat", flat); _lib.tune("spmm_flat_cpg", cpg)
            ts = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                _lib.check(L.graphop_vector_spmm_forward(_lib.dtype_code(X), _lib.ptr(row), _lib.ptr(ptr), _lib.ptr(eid), _lib.ptr(indices),
                                                         _lib.ptr(w), _lib.ptr(X), _lib.ptr(out), row.numel(), E, n_tab, n_out, 1, d,
                                                         plan.handle, st))
                b.record(); b.synchronize()
                ts.append(a.elapsed_time(b))
            print("%-28s %s chunks/group %3d: %.2f ms (min of 3 after 1 warm-up; %s)" %
                  (tag, "k_spmm_flat_f32" if flat else "k_spmm_f32     ", cpg, min(ts[1:]), " ".join("%.2f" % t for t in ts)), flush=True)
    _lib.tune_reset(); _lib.clear_plan_cache()


run(torch.arange(E, device=dev), "scalars in storage order")
run(torch.randperm(E, device=dev), "scalars behind a permutation")
