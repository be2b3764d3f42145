"""custom_op_benchmark_amd -- MI355X-native graph-attention operators.

The hot path of yzh119/custom_op_benchmark (SDDMM ``maskedmm_csr``, per-row ``sparse_softmax``,
``vector_spmm``; forward and backward) as hand-written gfx950 HIP kernels behind a C ABI
(``include/graphop_hip.h``), with the reference's Python surface on top:

* ``custom_op_benchmark_amd.graphop`` -- the eight functions of the reference's ``graphop``
  module (also importable as top-level ``graphop`` and as ``torch.ops.graphop.*``)
* ``custom_op_benchmark_amd.functions`` -- the four autograd.Function classes of ``wrapper.py``
* ``custom_op_benchmark_amd.part_csr.partition_csr`` -- the CSR row chunker
* ``custom_op_benchmark_amd.graphs`` -- graph containers / synthetic graph builders

Importing the package does not load the HIP library; the first op call does, and raises if it
has not been built.  There is no CPU fallback.
"""
from .part_csr import partition_csr, partition_csr_host  # noqa: F401

__version__ = "0.1.0"
