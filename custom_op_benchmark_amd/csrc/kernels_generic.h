// Generic kernels: any d, h, float or double, any chunk layout.  One wave per chunk.
// They are the correctness floor (odd feature sizes, fp64, >2^31 ids) under the fast
// fp32 paths in kernels_fast.h; semantics follow the reference kernels cited per function.
#pragma once
#include "common.h"

namespace graphop {

constexpr int kGenericBlock = 256;                       // 4 waves
constexpr int kGenericWavesPerBlock = kGenericBlock / kWave;

__device__ __forceinline__ i64 generic_chunk_id() {
  return (i64)blockIdx.x * kGenericWavesPerBlock + (threadIdx.x >> 6);
}

// y[eid[j]*h + ko] = <A[row[c], ko, :], B[src(j), ko, :]>     (graphop_kernel.cu:40-55, :135-149)
// EDGE_B = false: src(j) = indices[j], B is (n_b, h, d)       (maskedmm / spmm-backward-0)
// EDGE_B = true : src(j) = eid[j],     B is (n_edges, d)      (node_mul_edge, :19-34)
template <typename T, bool EDGE_B>
__global__ __launch_bounds__(kGenericBlock) void k_sddmm_generic(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const T* __restrict__ A, const T* __restrict__ B,
    T* __restrict__ y, i64 n_chunks, i64 h, i64 d) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c];
  const i64 j1 = indptr[c + 1];
  for (i64 j = indptr[c]; j < j1; ++j) {
    const i64 e = eid[j];
    const i64 s = EDGE_B ? e : indices[j];
    for (i64 ko = 0; ko < h; ++ko) {
      const T* a = A + (r * h + ko) * d;
      const T* b = EDGE_B ? (B + s * d) : (B + (s * h + ko) * d);
      T sum = 0;
      for (i64 ki = lane; ki < d; ki += kWave) sum += a[ki] * b[ki];
      sum = wave_sum(sum);
      if (lane == 0) y[e * h + ko] = sum;
    }
  }
}

// out[row[c]*F + f] += sum_k w[eid[k]*h + f/d] * X[src(k)*xs + xo(f)]
//   (graphop_kernel.cu:100-112, :118-130, :151-163; EDGE_X: :61-73 with X = B (n_edges, d))
template <typename T, bool EDGE_X>
__global__ __launch_bounds__(kGenericBlock) void k_spmm_generic(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const T* __restrict__ w, const T* __restrict__ X,
    T* __restrict__ out, i64 n_chunks, i64 h, i64 d) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 F = h * d;
  const i64 r = row[c];
  const i64 k0 = indptr[c], k1 = indptr[c + 1];
  for (i64 f = lane; f < F; f += kWave) {
    const i64 ko = f / d;
    T sum = 0;
    for (i64 k = k0; k < k1; ++k) {
      const i64 e = eid[k];
      const T xv = EDGE_X ? X[e * d + (f - ko * d)] : X[indices[k] * F + f];
      sum += w[e * h + ko] * xv;
    }
    if (k1 > k0) atomicAdd(out + r * F + f, sum);
  }
}

// dB[eid[k], j] = sum_ki dy[eid[k], ki] * A[row[c], ki, j]      (graphop_kernel.cu:79-94)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_node_mul_edge_bwd_b(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ A, const T* __restrict__ dy, T* __restrict__ dB, i64 n_chunks, i64 h,
    i64 d) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c];
  const i64 k0 = indptr[c], k1 = indptr[c + 1];
  for (i64 j = lane; j < d; j += kWave)
    for (i64 k = k0; k < k1; ++k) {
      const i64 e = eid[k];
      T sum = 0;
      for (i64 ki = 0; ki < h; ++ki) sum += dy[e * h + ki] * A[(r * h + ki) * d + j];
      dB[e * d + j] = sum;
    }
}

// Zero fill as a kernel.  hipMemsetAsync is not used on the data path: as a memset node of a
// captured HIP graph (ROCm 7.2) it left half of the dwords of the byte range untouched on replay.
__global__ void k_zero16(uint4* __restrict__ p, i64 n16, unsigned char* __restrict__ tail, int n_tail) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  if (i < n_tail) tail[i] = 0;
  for (; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ void k_zero1(unsigned char* __restrict__ p, i64 n) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0;
}

// ---- sparse softmax, general layout: the reference's three passes with native atomics ----------
template <typename T>
__global__ void k_fill(T* p, i64 n, T v) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// pass 0: max_val[row[c]*h + t] = max(., x[eid[k]*h + t])                  (:170-178)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_softmax_max(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ x, T* __restrict__ max_val, i64 n_chunks, i64 h) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c], k0 = indptr[c];
  const i64 items = (indptr[c + 1] - k0) * h;
  for (i64 q = lane; q < items; q += kWave) {
    const i64 k = k0 + q / h, t = q % h;
    atomic_max_float(max_val + r * h + t, x[eid[k] * h + t]);
  }
}

// pass 1: y = exp(x - max); sum[row] += y                                   (:180-192)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_softmax_exp_sum(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ x, const T* __restrict__ max_val, T* __restrict__ sum,
    T* __restrict__ y, i64 n_chunks, i64 h) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c], k0 = indptr[c];
  const i64 items = (indptr[c + 1] - k0) * h;
  for (i64 q = lane; q < items; q += kWave) {
    const i64 k = k0 + q / h, t = q % h;
    const i64 o = eid[k] * h + t;
    const T now = exp_t(x[o] - max_val[r * h + t]);
    y[o] = now;
    atomicAdd(sum + r * h + t, now);
  }
}

// pass 2: y /= sum[row]                                                     (:194-202)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_softmax_norm(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ sum, T* __restrict__ y, i64 n_chunks, i64 h) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c], k0 = indptr[c];
  const i64 items = (indptr[c + 1] - k0) * h;
  for (i64 q = lane; q < items; q += kWave) {
    const i64 k = k0 + q / h, t = q % h;
    y[eid[k] * h + t] /= sum[r * h + t];
  }
}

// backward pass 0: aggre[row] += sum_k dy*y                                 (:208-219)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_softmax_bwd_aggre(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ aggre, i64 n_chunks,
    i64 h) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c], k0 = indptr[c];
  const i64 items = (indptr[c + 1] - k0) * h;
  for (i64 q = lane; q < items; q += kWave) {
    const i64 k = k0 + q / h, t = q % h;
    const i64 o = eid[k] * h + t;
    atomicAdd(aggre + r * h + t, dy[o] * y[o]);
  }
}

// backward pass 1: dx = dy*y - aggre[row]*y                                 (:221-230)
template <typename T>
__global__ __launch_bounds__(kGenericBlock) void k_softmax_bwd_dx(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ aggre,
    T* __restrict__ dx, i64 n_chunks, i64 h) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c], k0 = indptr[c];
  const i64 items = (indptr[c + 1] - k0) * h;
  for (i64 q = lane; q < items; q += kWave) {
    const i64 k = k0 + q / h, t = q % h;
    const i64 o = eid[k] * h + t;
    dx[o] = dy[o] * y[o] - aggre[r * h + t] * y[o];
  }
}

}  // namespace graphop
