"""The sharded step's two column-major backward passes as two launches (k_spmm_flat_f32 twice) against ONE launch
(k_spmm_flat2_f32, graphop_spmm_pair) on the column side of a papers100M-shape 1/8 shard rebuilt in isolation (as
tools/short_rows_experiment.py): 28.9 M output rows, ~200 M slots, 512-B rows of two 13.9 M-row tables, per-slot scalars
behind a random permutation.      python tools/pair_columns_experiment.py     (one MI355X; ~45 GB of HBM)"""
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
eid = torch.randperm(E, device=dev)
print("E = %d, chunks = %d (%.2f slots per chunk), output rows = %d" % (E, row.numel(), E / row.numel(), n_out), flush=True)
w0, w1 = torch.rand(E, device=dev), torch.rand(E, device=dev)
X0, X1 = torch.rand(n_tab, d, device=dev), torch.rand(n_tab, d, device=dev)
out0, out1 = torch.empty(n_out, d, device=dev), torch.empty(n_out, d, device=dev)
L = _lib.lib()
with _lib.device_guard(dev):
    plan = _lib.get_plan(row, ptr, eid, indices, n_tab)
    st = _lib.stream_of(X0)

    def timed(fn, name):
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        print("%-64s %.2f ms (min of 3 after 1 warm-up; %s)" % (name, min(ts[1:]), " ".join("%.2f" % t for t in ts)), flush=True)

    def single(w, X, out):
        _lib.check(L.graphop_vector_spmm_forward(_lib.F32, _lib.ptr(row), _lib.ptr(ptr), _lib.ptr(eid), _lib.ptr(indices), _lib.ptr(w),
                                                 _lib.ptr(X), _lib.ptr(out), row.numel(), E, n_tab, n_out, 1, d, plan.handle, st))
    timed(lambda: (single(w0, X0, out0), single(w1, X1, out1)), "two launches (k_spmm_flat_f32 x 2)")
    w2 = torch.stack((w0, w1), dim=1)
    ref0, ref1 = out0.clone(), out1.clone()
    for cpg in (32, 64, 16):
        _lib.tune("spmm_flat_cpg", cpg)
        timed(lambda: _lib.check(L.graphop_spmm_pair(_lib.F32, _lib.ptr(row), _lib.ptr(ptr), _lib.ptr(eid), _lib.ptr(indices), _lib.ptr(w2),
                                                     _lib.ptr(X0), _lib.ptr(X1), _lib.ptr(out0), _lib.ptr(out1), row.numel(), E, n_tab,
                                                     n_out, 1, d, plan.handle, st)),
              "one launch (k_spmm_flat2_f32), %d chunks per lane group" % cpg)
    torch.testing.assert_close(out0, ref0, rtol=1e-4, atol=1e-5); torch.testing.assert_close(out1, ref1, rtol=1e-4, atol=1e-5)
    timed(lambda: torch.stack((w0, w1), dim=1), "packing the weights (torch.stack)")
    # (round 5 also measured the pairs SCATTERED into slot order first -- k_scatter_pairs 8.3 ms, the pass then 47.4 ms with
    # streamed weights -- and removed that form: profiles/r5_pair_columns_experiment.txt)
