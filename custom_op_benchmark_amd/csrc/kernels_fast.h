// Fast fp32 kernels for gfx950 (wave64).  Feature rows are walked as float4 (16 B / lane);
// a node row of F = h*d floats is covered by a GROUP of L lanes x NV float4 slots
// (F = 4*L*NV).  For the headline shape (d = 64, h = 1) L = 16: one 256-B row per 16-lane
// DPP row, four edges in flight per wave instruction, reductions by DPP inside the row.
//
//  * k_sddmm_f32   per-edge dot products (maskedmm fwd, spmm-backward dedata, node_mul_edge fwd)
//                  one group per chunk; indices of up to 16 slots loaded coalesced and broadcast
//                  with ds_bpermute; U neighbour rows in flight per group.
//  * k_spmm_f32    weighted row accumulation (spmm fwd, dx, dA, dB): one group per run of
//                  consecutive chunks, partial sums kept in registers while the row id does not
//                  change, merged with native global_atomic_add_f32 (256 contiguous bytes per
//                  group) only when it does.
//  * k_softmax_*_seg  per-row softmax / its backward over row segments from the plan
//                  (online max+sum, shuffle reductions, no atomics, no scratch).
#pragma once
#include "common.h"

namespace graphop {

constexpr int kFastBlock = 256;

template <int L>
struct GroupCfg {
  static constexpr int kGroupsPerBlock = kFastBlock / L;
  static constexpr int kEdgeBatch = L < 16 ? L : 16;  // slots whose ids one index load covers
};

template <int NV>
struct Unroll {  // neighbour rows in flight per group (16*NV*U bytes per lane)
  static constexpr int value = NV == 1 ? 8 : (NV == 2 ? 4 : 2);
};

__device__ __forceinline__ float4 ld4(const float* base, i64 f4_index) {
  return reinterpret_cast<const float4*>(base)[f4_index];
}

// -------------------------------------------------------------------------------------------------
// SDDMM-type: y[e*h + k] = <A[row[c], k, :], B[src, k, :]>
//   EDGE_B = false: src = indices[j], B (n_b, h, d)     (graphop_kernel.cu:40-55, :135-149)
//   EDGE_B = true : src = eid[j],     B (n_edges, d) shared by heads, requires d4 == L*NV / h ...
//                   handled as: B row has d floats = L*NV/h float4 -> only H1 instantiation uses
//                   EDGE_B with the same row width as A (h == 1); multi-head node_mul_edge takes
//                   the generic kernel.
// H1: h == 1 (one scalar per edge; results are collected across lanes and stored coalesced).
// d4 = d/4 (float4 per head); h*d4 == L*NV.
template <int L, int NV, bool H1, bool EDGE_B>
__global__ __launch_bounds__(kFastBlock) void k_sddmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ A, const float* __restrict__ B,
    float* __restrict__ y, i64 n_chunks, int h, int d4, int chunks_per_group) {
  using Cfg = GroupCfg<L>;
  constexpr int EB = Cfg::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * Cfg::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;

  float4 a[NV];
  i64 cur_row = -1;
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v] = ld4(A, r * F4 + v * L + l);
      cur_row = r;
    }
    const i64 j1 = indptr[c + 1];
    for (i64 jb = indptr[c]; jb < j1; jb += EB) {
      const int nb = (j1 - jb) < EB ? (int)(j1 - jb) : EB;
      int my_e = -1, my_src = 0;
      if (l < nb) {
        my_e = (int)eid[jb + l];
        my_src = EDGE_B ? my_e : (int)indices[jb + l];
      }
      float res = 0.f;
      for (int t = 0; t < nb; t += U) {
        float4 b[U][NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int tt = (t + u) < nb ? (t + u) : (nb - 1);
          const i64 src = __shfl(my_src, tt, L);
#pragma unroll
          for (int v = 0; v < NV; ++v) b[u][v] = ld4(B, src * F4 + v * L + l);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (H1) {
            float p = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) p += dot4(a[v], b[u][v]);
            p = group_sum<L>(p);
            if (l == t + u) res = p;
          } else {
            const bool live = (t + u) < nb;
            const int tt = live ? (t + u) : (nb - 1);
            const i64 e = __shfl(my_e, tt, L);
            if (d4 >= L) {  // a head spans d4/L whole slots: add them, then reduce the group
              const int sph = d4 / L;
              float acc = 0.f;
#pragma unroll
              for (int v = 0; v < NV; ++v) {
                acc += dot4(a[v], b[u][v]);
                if ((v + 1) % sph == 0) {
                  const float s = group_sum<L>(acc);
                  if (live && l == 0) y[e * h + v / sph] = s;
                  acc = 0.f;
                }
              }
            } else {  // a slot holds L/d4 heads: reduce sub-groups of d4 lanes
              const int hps = L / d4;
#pragma unroll
              for (int v = 0; v < NV; ++v) {
                const float s = group_sum_rt(dot4(a[v], b[u][v]), d4);
                if (live && (l % d4) == 0) y[e * h + v * hps + l / d4] = s;
              }
            }
          }
        }
      }
      if constexpr (H1) {
        if (my_e >= 0) y[my_e] = res;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// SpMM-type: out[row[c], f] += sum_k w[eid[k]*h + f/d] * X[indices[k], f]
//   (graphop_kernel.cu:100-112 dA/dB, :118-130 forward, :151-163 dx)
// One group walks `chunks_per_group` consecutive chunks and keeps the running row sum in
// registers; it is flushed with float atomics when the row id changes (and at the end), so any
// chunk layout is correct and adjacent chunks of one row cost one atomic burst per group.
template <int L, int NV, bool H1>
__global__ __launch_bounds__(kFastBlock) void k_spmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ w, const float* __restrict__ X,
    float* __restrict__ out, i64 n_chunks, int h, int d4, int chunks_per_group) {
  using Cfg = GroupCfg<L>;
  constexpr int EB = Cfg::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * Cfg::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;

  int hv[NV];  // head owning each of this lane's slots
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = H1 ? 0 : (v * L + l) / d4;

  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  i64 cur_row = -1;
  bool dirty = false;

  auto flush = [&]() {
    if (dirty) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float* p = out + (cur_row * F4 + v * L + l) * 4;
        atomicAdd(p + 0, acc[v].x);
        atomicAdd(p + 1, acc[v].y);
        atomicAdd(p + 2, acc[v].z);
        atomicAdd(p + 3, acc[v].w);
        acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    dirty = false;
  };

  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
      flush();
      cur_row = r;
    }
    const i64 j1 = indptr[c + 1];
    for (i64 jb = indptr[c]; jb < j1; jb += EB) {
      const int nb = (j1 - jb) < EB ? (int)(j1 - jb) : EB;
      dirty = true;
      int my_e = 0, my_src = 0;
      float my_w = 0.f;
      if (l < nb) {
        my_e = (int)eid[jb + l];
        my_src = (int)indices[jb + l];
        if constexpr (H1) my_w = w[my_e];
      }
      for (int t = 0; t < nb; t += U) {
        float4 x[U][NV];
        float wt[U][H1 ? 1 : NV];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool live = (t + u) < nb;
          const int tt = live ? (t + u) : (nb - 1);
          const i64 src = __shfl(my_src, tt, L);
          if constexpr (H1) {
            const float ww = __shfl(my_w, tt, L);
            wt[u][0] = live ? ww : 0.f;
          } else {
            const i64 e = __shfl(my_e, tt, L);
#pragma unroll
            for (int v = 0; v < NV; ++v) wt[u][v] = live ? w[e * h + hv[v]] : 0.f;
          }
#pragma unroll
          for (int v = 0; v < NV; ++v) x[u][v] = ld4(X, src * F4 + v * L + l);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const float ww = wt[u][H1 ? 0 : v];
            acc[v].x = fmaf(ww, x[u][v].x, acc[v].x);
            acc[v].y = fmaf(ww, x[u][v].y, acc[v].y);
            acc[v].z = fmaf(ww, x[u][v].z, acc[v].z);
            acc[v].w = fmaf(ww, x[u][v].w, acc[v].w);
          }
      }
    }
  }
  flush();
}

// -------------------------------------------------------------------------------------------------
// Row-segment softmax (plan.row_owned).  Segment s = chunks [seg_chunk[s], seg_chunk[s+1]) =
// slots [indptr[c0], indptr[c1]); all of one row.  A group of G lanes owns a segment; items are
// the flattened (slot, head) pairs so that for eid == identity the reads are fully coalesced.
// Requires G % h == 0 (then a lane always sees the same head t = lane % h).
// Semantics: graphop_kernel.cu:170-202 (m starts at -1e9, :428).
template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_fwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr,
    const i64* __restrict__ eid, const T* __restrict__ x, T* __restrict__ y, i64 n_seg, int h) {
  const int l = threadIdx.x % G;
  const i64 s = (i64)blockIdx.x * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = indptr[seg_chunk[s]];
  const i64 items = (indptr[seg_chunk[s + 1]] - e0) * h;
  const int t = l % h;

  T m = (T)-1e9, sum = 0;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const T v = x[(EID_ID ? k : eid[k]) * h + t];
    if (v > m) {
      sum = sum * exp_t(m - v) + (T)1;
      m = v;
    } else {
      sum += exp_t(v - m);
    }
  }
#pragma unroll
  for (int mask = G / 2; mask >= 1; mask >>= 1) {
    if (mask >= h) {  // wave-uniform
      const T m2 = __shfl_xor(m, mask, G);
      const T s2 = __shfl_xor(sum, mask, G);
      const T mn = m > m2 ? m : m2;
      sum = sum * exp_t(m - mn) + s2 * exp_t(m2 - mn);
      m = mn;
    }
  }
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    y[o] = exp_t(x[o] - m) / sum;
  }
}

// Backward: g = sum dy*y over the row; dx = dy*y - g*y   (graphop_kernel.cu:208-230)
template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_bwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr,
    const i64* __restrict__ eid, const T* __restrict__ y, const T* __restrict__ dy,
    T* __restrict__ dx, i64 n_seg, int h) {
  const int l = threadIdx.x % G;
  const i64 s = (i64)blockIdx.x * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = indptr[seg_chunk[s]];
  const i64 items = (indptr[seg_chunk[s + 1]] - e0) * h;
  const int t = l % h;

  T g = 0;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    g += dy[o] * y[o];
  }
#pragma unroll
  for (int mask = G / 2; mask >= 1; mask >>= 1)
    if (mask >= h) g += __shfl_xor(g, mask, G);
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    const T yy = y[o];
    dx[o] = dy[o] * yy - g * yy;
  }
}

// Any h (G need not be a multiple of h): heads in an outer loop, strided reads.
template <typename T, bool BWD>
__global__ __launch_bounds__(kFastBlock) void k_softmax_seg_anyh(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr,
    const i64* __restrict__ eid, const T* __restrict__ in0 /* x | y */,
    const T* __restrict__ in1 /* - | dy */, T* __restrict__ out, i64 n_seg, i64 h) {
  const int lane = threadIdx.x & 63;
  const i64 s = (i64)blockIdx.x * (kFastBlock / kWave) + (threadIdx.x >> 6);
  if (s >= n_seg) return;
  const i64 e0 = indptr[seg_chunk[s]], e1 = indptr[seg_chunk[s + 1]];
  for (i64 t = 0; t < h; ++t) {
    if constexpr (!BWD) {
      T m = (T)-1e9;
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const T v = in0[eid[k] * h + t];
        m = v > m ? v : m;
      }
#pragma unroll
      for (int mask = 32; mask >= 1; mask >>= 1) {
        const T m2 = __shfl_xor(m, mask);
        m = m > m2 ? m : m2;
      }
      T sum = 0;
      for (i64 k = e0 + lane; k < e1; k += kWave) sum += exp_t(in0[eid[k] * h + t] - m);
      sum = wave_sum(sum);
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        out[o] = exp_t(in0[o] - m) / sum;
      }
    } else {
      T g = 0;
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        g += in1[o] * in0[o];
      }
      g = wave_sum(g);
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        out[o] = in1[o] * in0[o] - g * in0[o];
      }
    }
  }
}

}  // namespace graphop
