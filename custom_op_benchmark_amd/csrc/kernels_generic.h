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
// Edges are taken kGenericBatch at a time: their ids are loaded by the first lanes and handed round by shuffles, and
// the batch's rows are requested together (one row in flight per wave made fp64 on the Reddit shape 17.6 ms per pass).
// (SpMM: the order of every sum is unchanged; SDDMM: the cross-lane reduction is a transpose-reduce, another tree.)
constexpr int kGenericBatch = 8;

template <typename T, bool EDGE_B>
__global__ __launch_bounds__(kGenericBlock) void k_sddmm_generic(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const T* __restrict__ A, const T* __restrict__ B,
    T* __restrict__ y, i64 n_chunks, i64 h, i64 d) {
  const i64 c = generic_chunk_id();
  if (c >= n_chunks) return;
  constexpr int U = kGenericBatch;
  const int lane = threadIdx.x & 63;
  const i64 r = row[c];
  const i64 j1 = indptr[c + 1];
  for (i64 jb = indptr[c]; jb < j1; jb += U) {
    const int nb = (j1 - jb) < U ? (int)(j1 - jb) : U;
    i64 my_e = 0, my_s = 0;
    if (lane < nb) {
      my_e = eid[jb + lane];
      my_s = EDGE_B ? my_e : indices[jb + lane];
    }
    for (i64 ko = 0; ko < h; ++ko) {
      const T* a = A + (r * h + ko) * d;
      T part[U];
#pragma unroll
      for (int u = 0; u < U; ++u) part[u] = 0;
      for (i64 k0 = 0; k0 < d; k0 += kWave) {
        const i64 ki = k0 + lane;
        const T av = ki < d ? a[ki] : (T)0;
        T bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const i64 s = __shfl(my_s, u < nb ? u : nb - 1);
          const T* b = EDGE_B ? (B + s * d) : (B + (s * h + ko) * d);
          bv[u] = ki < d ? b[ki] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) part[u] += av * bv[u];
      }
      // transpose-reduce: lanes whose bits 5..3 spell u end up with edge u's dot product (10 shuffles for the 8
      // edges instead of 8 x 6), and the batch's results leave in one store instruction
      static_assert(U == 8, "three halving steps");
      const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
      T t4[4], t2[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T keep = b5 ? part[i + 4] : part[i], send = b5 ? part[i] : part[i + 4];
        t4[i] = keep + __shfl_xor(send, 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const T keep = b4 ? t4[i + 2] : t4[i], send = b4 ? t4[i] : t4[i + 2];
        t2[i] = keep + __shfl_xor(send, 16);
      }
      T sum = (b3 ? t2[1] : t2[0]) + __shfl_xor(b3 ? t2[0] : t2[1], 8);
      sum += __shfl_xor(sum, 4);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 1);
      const int u_mine = lane >> 3;
      const i64 e = __shfl(my_e, u_mine < nb ? u_mine : nb - 1);
      if ((lane & 7) == 0 && u_mine < nb) y[e * h + ko] = sum;
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
  constexpr int U = kGenericBatch;
  const int lane = threadIdx.x & 63;
  const i64 F = h * d;
  const i64 r = row[c];
  const i64 k0 = indptr[c], k1 = indptr[c + 1];
  for (i64 f0 = 0; f0 < F; f0 += kWave) {          // (wave-uniform trip count: every lane takes part in the shuffles)
    const i64 f = f0 + lane;
    const bool ok = f < F;
    const i64 ko = ok ? f / d : 0;
    T sum = 0;
    for (i64 kb = k0; kb < k1; kb += U) {
      const int nb = (k1 - kb) < U ? (int)(k1 - kb) : U;
      i64 my_e = 0, my_i = 0;
      if (lane < nb) {
        my_e = eid[kb + lane];
        if constexpr (!EDGE_X) my_i = indices[kb + lane];
      }
      T xv[U], wv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int uu = u < nb ? u : nb - 1;
        const i64 e = __shfl(my_e, uu);
        const i64 src = EDGE_X ? e : __shfl(my_i, uu);
        const bool live = ok && u < nb;
        xv[u] = live ? (EDGE_X ? X[e * d + (f - ko * d)] : X[src * F + f]) : (T)0;
        wv[u] = live ? w[e * h + ko] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (u < nb) sum += wv[u] * xv[u];
    }
    if (ok && k1 > k0) atomicAdd(out + r * F + f, sum);
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
