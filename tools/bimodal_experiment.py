"""Why do passes of the papers100M-shape shard step differ by ~10 % from process to process (same kernels, same inputs)?
One process, one graph and one set of plans; the value tensors (Q, K_ext, V_ext, dO and everything a step allocates) are
released to the driver and re-created in every trial, behind a dummy allocation of a different size, and every pass is
timed again.  If the modes move with the trials, they belong to where the big buffers land, not to the kernels.
    python tools/bimodal_experiment.py        (one MI355X, ~120 GB of HBM)"""
import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custom_op_benchmark_amd import _lib
from custom_op_benchmark_amd.dist import ShardedAttention

dev = torch.device("cuda:0")
d = 128
sh = ShardedAttention.synthetic(111_059_956 // 8, 1_615_685_872 // 8, 8, 0, dev, alpha=0.5, seed=0, cut=0.1, chunk_size=32,
                                timing_only=True)
n_own = sh.n_own
PASSES = ("sddmm_fwd", "spmm_fwd", "spmm_bwd_dedata", "spmm_bwd_dx", "sddmm_bwd_dA", "sddmm_bwd_dB", "softmax_fwd", "softmax_bwd")
for trial in range(6):
    keep = [k for k in sh._buffers if k != "empty_chunks"]
    for k in keep:
        del sh._buffers[k]
    gc.collect(); torch.cuda.empty_cache()
    dummy = torch.empty(int(trial * 2.7e9) + 16, dtype=torch.uint8, device=dev)     # shifts what the driver hands out next
    gen = torch.Generator(device=dev).manual_seed(1)
    Q = torch.rand(n_own, d, device=dev, generator=gen).requires_grad_(True)
    K = sh.own_rows_view("K", (d,)).copy_(torch.rand(n_own, d, device=dev, generator=gen)).requires_grad_(True)
    V = sh.own_rows_view("V", (d,)).copy_(torch.rand(n_own, d, device=dev, generator=gen)).requires_grad_(True)
    dO = torch.rand(n_own, d, device=dev, generator=gen)
    sh.step(Q, K, V, dO); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(3):
        Q.grad = K.grad = V.grad = None
        sh.step(Q, K, V, dO)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); _lib.profile_enable(False)
    ptrs = " ".join("%s@%x" % (n, t.data_ptr()) for n, t in (("Q", Q), ("K_ext", K), ("V_ext", V), ("dO", dO)))
    print("trial %d (dummy %.1f GB): " % (trial, dummy.numel() / 1e9) + " ".join("%s=%.2f" % (p, prof[p]["mean_ms"]) for p in PASSES if p in prof), flush=True)
    print("         " + ptrs, flush=True)
    del Q, K, V, dO, dummy
