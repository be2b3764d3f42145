"""partition_csr -- the CSR row chunker feeding every graphop kernel.

Mirror of the reference's ``partition_csr(indptr, chunk_size=32)`` (``part_csr.py:13-27``):
each CSR row is split into chunks of at most ``chunk_size`` slots; returns ``row[C]`` (owning
row of each chunk) and ``indptr_[C+1]`` (first slot of each chunk, then ``indptr[-1]``), both
int64 on the input's device.  Rows with no slots emit no chunk.  Results are bit-exact
with the reference (tests/test_partition_csr.py checks the fixtures captured from it).

The reference walks rows in a Python loop on the CPU (``part_csr.py:15-21``: one D2H copy,
then O(N + C) interpreter iterations -- minutes at 1e8+ edges).  Here:

* CPU tensors: O(N + C) vectorised torch (host logic; runs in the no-GPU test tier).
* GPU tensors: two hand-written HIP kernels behind the C ABI
  (``graphop_partition_csr_count`` / ``graphop_partition_csr_fill``, include/graphop_hip.h);
  no host copy of the graph, only the chunk count C comes back to size the outputs.
"""
import torch


def _check(indptr, chunk_size):
    if indptr.dim() != 1 or indptr.numel() < 1:
        raise RuntimeError("partition_csr: indptr must be a non-empty 1-D tensor")
    if int(chunk_size) < 1:
        raise ValueError("partition_csr: chunk_size must be >= 1")  # range() step 0 -> ValueError


def partition_csr_host(indptr, chunk_size=32):
    """Vectorised CPU/torch form (any device torch supports, no custom kernels)."""
    _check(indptr, chunk_size)
    dev = indptr.device
    ip = indptr.to(torch.int64)
    n = ip.numel() - 1
    if n == 0:
        return (torch.empty(0, dtype=torch.int64, device=dev), ip[-1:].clone())
    deg = (ip[1:] - ip[:-1]).clamp_(min=0)          # range(a, b, c) is empty when b <= a
    cnt = (deg + (chunk_size - 1)) // chunk_size
    first = torch.cumsum(cnt, 0) - cnt              # index of each row's first chunk
    row = torch.repeat_interleave(torch.arange(n, dtype=torch.int64, device=dev), cnt)
    k = torch.arange(row.numel(), dtype=torch.int64, device=dev) - first[row]
    ptr = torch.empty(row.numel() + 1, dtype=torch.int64, device=dev)
    ptr[:-1] = ip[row] + k * chunk_size
    ptr[-1] = ip[-1]
    return row, ptr


def partition_csr(indptr, chunk_size=32):
    _check(indptr, chunk_size)
    if indptr.device.type != "cuda":
        return partition_csr_host(indptr, chunk_size)
    from . import _lib                                # HIP path: fails loudly if not built
    return _lib.partition_csr_device(indptr, int(chunk_size))
