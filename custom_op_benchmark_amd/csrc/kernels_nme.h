#pragma once
#include "kernels_base.h"

namespace graphop {

// -------------------------------------------------------------------------------------------------
// node_mul_edge (graphop_kernel.cu:19-34, :61-94): per-edge features B (n_edges, d) shared by all
// heads.  Pure streaming over B: a group of LD = d/4 lanes owns one edge row at a time, the H head
// rows of A[row] sit in registers, a group walks a run of chunks.
//   forward : y[e, k] = <A[row, k, :], B[e, :]>
//   backward: dB[e, :] = sum_k dy[e, k] * A[row, k, :]   (one full-row store per edge)
//             dA[row, k, :] += sum_e dy[e, k] * B[e, :]  (registers; atomics when the row changes)
template <int LD, int H>
__global__ __launch_bounds__(kFastBlock) void k_nme_fwd_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y, i64 n_chunks,
    int chunks_per_group) {
  constexpr int EB = LD < 16 ? LD : 16;
  constexpr int U = H >= 8 ? 2 : 4;
  const int l = threadIdx.x % LD;
  const i64 gid = (i64)blockIdx.x * (kFastBlock / LD) + threadIdx.x / LD;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  float4 a[H];
  i64 cur_row = -1;
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
#pragma unroll
      for (int k = 0; k < H; ++k) a[k] = ld4(A, (r * H + k) * LD + l);
      cur_row = r;
    }
    const i64 j1 = indptr[c + 1];
    for (i64 jb = indptr[c]; jb < j1; jb += EB) {
      const int nb = (j1 - jb) < EB ? (int)(j1 - jb) : EB;
      int my_e = -1;
      if (l < nb) my_e = (int)eid[jb + l];
      float res[H];
#pragma unroll
      for (int k = 0; k < H; ++k) res[k] = 0.f;
      for (int t = 0; t < nb; t += U) {
        float4 b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const i64 e = __shfl(my_e, (t + u) < nb ? (t + u) : (nb - 1), LD);
          b[u] = ld4(B, e * LD + l);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int k = 0; k < H; ++k) {
            const float p = group_sum<LD>(dot4(a[k], b[u]));
            if (l == t + u) res[k] = p;
          }
      }
      if (my_e >= 0) {
#pragma unroll
        for (int k = 0; k < H; ++k) y[(i64)my_e * H + k] = res[k];
      }
    }
  }
}

template <int LD, int H>
__global__ __launch_bounds__(kFastBlock) void k_nme_bwd_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ dy,
    float* __restrict__ dA, float* __restrict__ dB, i64 n_chunks, int chunks_per_group) {
  constexpr int EB = LD < 16 ? LD : 16;
  constexpr int U = 4;
  const int l = threadIdx.x % LD;
  const i64 gid = (i64)blockIdx.x * (kFastBlock / LD) + threadIdx.x / LD;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  float4 a[H], acc[H];
#pragma unroll
  for (int k = 0; k < H; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  i64 cur_row = -1;
  bool dirty = false;
  auto flush = [&]() {
    if (dirty) {
#pragma unroll
      for (int k = 0; k < H; ++k) {
        float* p = dA + ((cur_row * H + k) * LD + l) * 4;
        atomicAdd(p + 0, acc[k].x); atomicAdd(p + 1, acc[k].y);
        atomicAdd(p + 2, acc[k].z); atomicAdd(p + 3, acc[k].w);
        acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    dirty = false;
  };
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
      flush();
#pragma unroll
      for (int k = 0; k < H; ++k) a[k] = ld4(A, (r * H + k) * LD + l);
      cur_row = r;
    }
    const i64 j1 = indptr[c + 1];
    for (i64 jb = indptr[c]; jb < j1; jb += EB) {
      const int nb = (j1 - jb) < EB ? (int)(j1 - jb) : EB;
      dirty = true;
      int my_e = 0;
      if (l < nb) my_e = (int)eid[jb + l];
      for (int t = 0; t < nb; t += U) {
        float4 b[U];
        float g[U][H];
        i64 es[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (t + u) < nb;
          es[u] = __shfl(my_e, live ? (t + u) : (nb - 1), LD);
          b[u] = ld4(B, es[u] * LD + l);
#pragma unroll
          for (int k = 0; k < H; ++k) g[u][k] = live ? dy[es[u] * H + k] : 0.f;   // same address in the group
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < H; ++k) {
            const float w = g[u][k];
            o.x = fmaf(w, a[k].x, o.x); o.y = fmaf(w, a[k].y, o.y);
            o.z = fmaf(w, a[k].z, o.z); o.w = fmaf(w, a[k].w, o.w);
            acc[k].x = fmaf(w, b[u].x, acc[k].x); acc[k].y = fmaf(w, b[u].y, acc[k].y);
            acc[k].z = fmaf(w, b[u].z, acc[k].z); acc[k].w = fmaf(w, b[u].w, acc[k].w);
          }
          if ((t + u) < nb) reinterpret_cast<float4*>(dB)[es[u] * LD + l] = o;
        }
      }
    }
  }
  flush();
}

}  // namespace graphop
