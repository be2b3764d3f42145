/*
 * oracle/graphop_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, single thread) of the reference's graph-attention
 * kernels.  Every function walks the chunked CSR exactly the way the reference
 * device kernel does (one "block" per chunk, serial loops where the reference
 * has thread/stride loops) and cites the reference lines it restates.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (custom_op_benchmark_amd/) never does.
 *
 * Parity pinning: see oracle/README.md and tests/golden/gen_golden.py -- the
 * oracle is checked against fixtures produced by importing the reference's own
 * Python (part_csr.partition_csr, wrapper.MaskedMMSimple, the wrapper's
 * autograd.Function classes) and the stock-PyTorch formulations the reference
 * harness asserts against (wrapper.py:155-157,185,218,245,274-283,395,422,459).
 *
 * Layout conventions (all row-major, contiguous, int64 indices):
 *   row[C], indptr[C+1]      chunked CSR from partition_csr (part_csr.py:13-27)
 *   eid[E], indices[E]       edge id / neighbour id per CSR slot
 *   node tensors             (N, h, d)  ->  element (v,k,i) at (v*h + k)*d + i
 *   edge tensors             (E, h)     ->  element (e,k)   at e*h + k
 *
 * The reference transposes B / x to (h, d, N) before its dot-product kernels
 * (graphop_kernel.cu:289,569) and indexes Bt[(ko*d+ki)*n + indices[j]]; that is
 * the same element as B[indices[j], ko, ki], which is what is read here.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

#define DEFINE_ORACLE(T, SUF, EXPF)                                                          \
                                                                                             \
/* graphop_kernel.cu:40-55 (kernel), :269-304 (launcher: y zero-init :284) */                \
ORACLE_API void oracle_maskedmm_csr_forward_##SUF(                                           \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const int64_t* indices,   \
    const T* A, const T* B, T* y, int64_t n_chunks, int64_t n_edges, int64_t d, int64_t h) { \
  memset(y, 0, sizeof(T) * (size_t)(n_edges * h));                                           \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j)                                      \
      for (int64_t ko = 0; ko < h; ++ko) {                                                   \
        T sum = 0;                                                                           \
        for (int64_t ki = 0; ki < d; ++ki)                                                   \
          sum += A[(row[i] * h + ko) * d + ki] * B[(indices[j] * h + ko) * d + ki];          \
        y[eid[j] * h + ko] = sum;                                                            \
      }                                                                                      \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:100-112; one launch of the backward kernel (the launcher  */            \
/* :355-409 calls it twice: row-CSR with operand B -> dA, col-CSR with A -> dB) */           \
static void maskedmm_bwd_pass_##SUF(                                                         \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const int64_t* indices,   \
    const T* Bop, const T* dy, T* dA, int64_t n_chunks, int64_t d, int64_t h) {              \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = 0; j < d * h; ++j) {                                                    \
      T sum = 0;                                                                             \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        sum += dy[eid[k] * h + j / d] * Bop[indices[k] * d * h + j];                         \
      dA[row[i] * d * h + j] += sum; /* dgl::AtomicAdd, :109 */                              \
    }                                                                                        \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:355-409; dA,dB = zeros_like (:379-380) */                               \
ORACLE_API void oracle_maskedmm_csr_backward_##SUF(                                          \
    const int64_t* row, const int64_t* indptr_r, const int64_t* eid_r,                       \
    const int64_t* indices_r, const int64_t* col, const int64_t* indptr_c,                   \
    const int64_t* eid_c, const int64_t* indices_c, const T* A, const T* B, const T* dy,     \
    T* dA, T* dB, int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_a, int64_t n_b,      \
    int64_t d, int64_t h) {                                                                  \
  memset(dA, 0, sizeof(T) * (size_t)(n_a * h * d));                                          \
  memset(dB, 0, sizeof(T) * (size_t)(n_b * h * d));                                          \
  maskedmm_bwd_pass_##SUF(row, indptr_r, eid_r, indices_r, B, dy, dA, n_row_chunks, d, h);   \
  maskedmm_bwd_pass_##SUF(col, indptr_c, eid_c, indices_c, A, dy, dB, n_col_chunks, d, h);   \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:170-202 (three kernels), launcher :411-463.                */           \
/* max_val is pre-filled with -1e9, not -inf (:428); scratch indexed by row id. */           \
ORACLE_API void oracle_sparse_softmax_forward_##SUF(                                         \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const T* x, T* y,         \
    T* scratch_max, T* scratch_sum, int64_t n_chunks, int64_t n_edges, int64_t n_scratch,    \
    int64_t h) {                                                                             \
  memset(y, 0, sizeof(T) * (size_t)(n_edges * h));                                           \
  for (int64_t i = 0; i < n_scratch * h; ++i) { scratch_max[i] = (T)-1e9; scratch_sum[i] = 0; } \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t tx = 0; tx < h; ++tx)                                                       \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {                                  \
        T v = x[eid[k] * h + tx];                                                            \
        if (v > scratch_max[row[i] * h + tx]) scratch_max[row[i] * h + tx] = v; /* :176 */   \
      }                                                                                      \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t tx = 0; tx < h; ++tx) {                                                     \
      T max_v = scratch_max[row[i] * h + tx];                                                \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {                                  \
        T now = EXPF(x[eid[k] * h + tx] - max_v);                                            \
        y[eid[k] * h + tx] = now;                                                            \
        scratch_sum[row[i] * h + tx] += now; /* :189 */                                      \
      }                                                                                      \
    }                                                                                        \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t tx = 0; tx < h; ++tx)                                                       \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        y[eid[k] * h + tx] /= scratch_sum[row[i] * h + tx]; /* :200 */                       \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:208-230 (two kernels), launcher :465-507 */                             \
ORACLE_API void oracle_sparse_softmax_backward_##SUF(                                        \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const T* y, const T* dy,  \
    T* dx, T* scratch_aggre, int64_t n_chunks, int64_t n_edges, int64_t n_scratch,           \
    int64_t h) {                                                                             \
  memset(dx, 0, sizeof(T) * (size_t)(n_edges * h));                                          \
  memset(scratch_aggre, 0, sizeof(T) * (size_t)(n_scratch * h));                             \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t tx = 0; tx < h; ++tx) {                                                     \
      T sum = 0;                                                                             \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        sum += dy[eid[k] * h + tx] * y[eid[k] * h + tx];                                     \
      scratch_aggre[row[i] * h + tx] += sum; /* :217 */                                      \
    }                                                                                        \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t tx = 0; tx < h; ++tx)                                                       \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        dx[eid[k] * h + tx] = dy[eid[k] * h + tx] * y[eid[k] * h + tx] -                     \
                              scratch_aggre[row[i] * h + tx] * y[eid[k] * h + tx]; /* :227 */\
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:118-130 (= :151-163 with the transposed CSR) */                         \
static void spmm_pass_##SUF(                                                                 \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const int64_t* indices,   \
    const T* edata, const T* x, T* y, int64_t n_chunks, int64_t d, int64_t h) {              \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = 0; j < d * h; ++j) {                                                    \
      T sum = 0;                                                                             \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        sum += edata[eid[k] * h + j / d] * x[indices[k] * d * h + j];                        \
      y[row[i] * d * h + j] += sum; /* dgl::AtomicAdd, :127 / :160 */                        \
    }                                                                                        \
}                                                                                            \
                                                                                             \
/* launcher graphop_kernel.cu:509-542; y = zeros_like(x) (:527) */                           \
ORACLE_API void oracle_vector_spmm_forward_##SUF(                                            \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const int64_t* indices,   \
    const T* edata, const T* x, T* y, int64_t n_chunks, int64_t n_x, int64_t d, int64_t h) { \
  memset(y, 0, sizeof(T) * (size_t)(n_x * h * d));                                           \
  spmm_pass_##SUF(row, indptr, eid, indices, edata, x, y, n_chunks, d, h);                   \
}                                                                                            \
                                                                                             \
/* launcher graphop_kernel.cu:544-600: kernel_0 :135-149 (dedata = SDDMM(dy,x)  */           \
/* over the row-CSR), kernel_1 :151-163 (dx = SpMM over the transposed CSR).    */           \
/* The reference launches kernel_1 with blocks(n_row) but bound n_col (:566,    */           \
/* :588-596) -- a latent bug when the two chunk counts differ; the restatement  */           \
/* covers every col-chunk, which is what the reference does whenever C' <= C.   */           \
ORACLE_API void oracle_vector_spmm_backward_##SUF(                                           \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const int64_t* indices,   \
    const int64_t* col, const int64_t* indptr_t, const int64_t* eid_t,                       \
    const int64_t* indices_t, const T* edata, const T* dy, const T* x, T* dedata, T* dx,     \
    int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_edges, int64_t n_x, int64_t d,     \
    int64_t h) {                                                                             \
  memset(dedata, 0, sizeof(T) * (size_t)(n_edges * h));                                      \
  memset(dx, 0, sizeof(T) * (size_t)(n_x * h * d));                                          \
  for (int64_t i = 0; i < n_row_chunks; ++i)                                                 \
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j)                                      \
      for (int64_t ko = 0; ko < h; ++ko) {                                                   \
        T sum = 0;                                                                           \
        for (int64_t ki = 0; ki < d; ++ki)                                                   \
          sum += dy[(row[i] * h + ko) * d + ki] * x[(indices[j] * h + ko) * d + ki];         \
        dedata[eid[j] * h + ko] = sum;                                                       \
      }                                                                                      \
  spmm_pass_##SUF(col, indptr_t, eid_t, indices_t, edata, dy, dx, n_col_chunks, d, h);       \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:19-34, launcher :235-266 (B is (E, d), shared by heads) */              \
ORACLE_API void oracle_node_mul_edge_forward_##SUF(                                          \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const T* A, const T* B,   \
    T* y, int64_t n_chunks, int64_t n_edges, int64_t d, int64_t h) {                         \
  memset(y, 0, sizeof(T) * (size_t)(n_edges * h));                                           \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j)                                      \
      for (int64_t ko = 0; ko < h; ++ko) {                                                   \
        T sum = 0;                                                                           \
        for (int64_t ki = 0; ki < d; ++ki)                                                   \
          sum += A[(row[i] * h + ko) * d + ki] * B[eid[j] * d + ki];                         \
        y[eid[j] * h + ko] = sum;                                                            \
      }                                                                                      \
}                                                                                            \
                                                                                             \
/* graphop_kernel.cu:61-73 (dA) and :79-94 (dB), launcher :306-351 */                        \
ORACLE_API void oracle_node_mul_edge_backward_##SUF(                                         \
    const int64_t* row, const int64_t* indptr, const int64_t* eid, const T* A, const T* B,   \
    const T* dy, T* dA, T* dB, int64_t n_chunks, int64_t n_a, int64_t n_edges_b, int64_t d,  \
    int64_t h) {                                                                             \
  memset(dA, 0, sizeof(T) * (size_t)(n_a * h * d));                                          \
  memset(dB, 0, sizeof(T) * (size_t)(n_edges_b * d));                                        \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = 0; j < d * h; ++j) {                                                    \
      T sum = 0;                                                                             \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k)                                    \
        sum += dy[eid[k] * h + j / d] * B[eid[k] * d + j % d];                               \
      dA[row[i] * d * h + j] += sum; /* :70 */                                               \
    }                                                                                        \
  for (int64_t i = 0; i < n_chunks; ++i)                                                     \
    for (int64_t j = 0; j < d; ++j)                                                          \
      for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) {                                  \
        T sum = 0;                                                                           \
        for (int64_t ki = 0; ki < h; ++ki)                                                   \
          sum += dy[eid[k] * h + ki] * A[(row[i] * h + ki) * d + j];                         \
        dB[eid[k] * d + j] = sum; /* :91 */                                                  \
      }                                                                                      \
}

DEFINE_ORACLE(float, f32, expf)
DEFINE_ORACLE(double, f64, exp)

/* part_csr.py:13-27.  Two-call protocol: call with indptr_out == NULL to get C. */
ORACLE_API int64_t oracle_partition_csr(const int64_t* indptr, int64_t n_rows,
                                        int64_t chunk_size, int64_t* row, int64_t* indptr_out) {
  int64_t c = 0;
  for (int64_t i = 0; i < n_rows; ++i)
    for (int64_t j = indptr[i]; j < indptr[i + 1]; j += chunk_size) { /* range(a, b, chunk) */
      if (indptr_out) { row[c] = i; indptr_out[c] = j; }
      ++c;
    }
  if (indptr_out) indptr_out[c] = indptr[n_rows]; /* indptr_.append(indptr[-1]), :23 */
  return c;
}
