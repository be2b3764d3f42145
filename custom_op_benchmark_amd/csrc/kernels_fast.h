// Fast fp32 kernels for gfx950 (wave64).  Feature rows are walked as float4 (16 B / lane);
// a node row of F = h*d floats is covered by a GROUP of L lanes x NV float4 slots
// (F = 4*L*NV).  For the headline shape (d = 64, h = 1) L = 16: one 256-B row per 16-lane
// DPP row, four edges in flight per wave instruction, reductions by DPP inside the row.
//
// Inner loops (shared):
//  * sddmm_range  per-edge dot products over a slot range [lo, hi) of one row
//  * spmm_range   weighted accumulation of gathered rows over a slot range
//    ids of up to 16 slots are loaded coalesced by the group and broadcast with ds_bpermute;
//    U neighbour rows (16 B/lane each) are in flight per group.
//
// Drivers:
//  * k_sddmm_f32 / k_spmm_f32   CHUNK drivers: work unit = the caller's chunk list; correct for
//    any chunk layout (SpMM keeps the running row sum in registers while the row id does not
//    change and merges with native global_atomic_add_f32, 256 contiguous bytes per group).
//  * k_sddmm_sweep_f32 / k_spmm_sweep_f32   WINDOW-SWEEP drivers (need a plan): the gathered
//    table is cut into W column windows of ~2 MB so that the window being gathered from stays in
//    every XCD's 4 MiB L2; every lane group owns K (pieces of) rows, keeps their A rows /
//    partial sums in LDS and walks windows in the OUTER loop, so all resident groups move through
//    the windows together.  The gather then runs at L2 rate instead of Infinity-Cache rate.
//  * k_softmax_*_seg   per-row softmax / its backward over row segments from the plan
//    (online max+sum, shuffle reductions, no atomics, no scratch).
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

template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// -------------------------------------------------------------------------------------------------
// y[e*h + k] = <a[k], B[src, k, :]> for slots [lo, hi) (all of one row whose features are in a[]).
//   EDGE_B = false: src = idx[j], B (n_b, h, d)        (graphop_kernel.cu:40-55, :135-149)
//   EDGE_B = true : src = eid[j], B (n_edges, d), h==1  (node_mul_edge, :19-34)
//   EID_ID: eid[j] == j (skip the load).  IT: int64 API arrays or the plan's int32 mirrors.
//   H1: h == 1 -> results are collected across lanes and stored coalesced.
template <int L, int NV, bool H1, bool EDGE_B, bool EID_ID, bool NT, typename IT>
__device__ __forceinline__ void sddmm_range(const float4 (&a)[NV], i64 lo, i64 hi,
                                            const IT* __restrict__ eid,
                                            const IT* __restrict__ idx,
                                            const float* __restrict__ B, float* __restrict__ y,
                                            int h, int d4, int l) {
  constexpr int EB = GroupCfg<L>::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  for (i64 jb = lo; jb < hi; jb += EB) {
    const int nb = (hi - jb) < EB ? (int)(hi - jb) : EB;
    int my_e = -1, my_src = 0;
    if (l < nb) {
      my_e = EID_ID ? (int)(jb + l) : (int)ld_stream<NT>(eid + jb + l);
      my_src = EDGE_B ? my_e : (int)ld_stream<NT>(idx + jb + l);
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
      if (my_e >= 0) st_stream<NT>(y + my_e, res);
    }
  }
}

// acc[f] += sum_{k in [lo,hi)} w[eid[k]*h + head(f)] * X[idx[k], f]
//   (graphop_kernel.cu:100-112 dA/dB, :118-130 forward, :151-163 dx)
template <int L, int NV, bool H1, bool EID_ID, bool NT, typename IT>
__device__ __forceinline__ void spmm_range(float4 (&acc)[NV], i64 lo, i64 hi,
                                           const IT* __restrict__ eid, const IT* __restrict__ idx,
                                           const float* __restrict__ w,
                                           const float* __restrict__ X, int h,
                                           const int (&hv)[NV], int l) {
  constexpr int EB = GroupCfg<L>::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  for (i64 jb = lo; jb < hi; jb += EB) {
    const int nb = (hi - jb) < EB ? (int)(hi - jb) : EB;
    int my_e = 0, my_src = 0;
    float my_w = 0.f;
    if (l < nb) {
      my_e = EID_ID ? (int)(jb + l) : (int)ld_stream<NT>(eid + jb + l);
      my_src = (int)ld_stream<NT>(idx + jb + l);
      if constexpr (H1) my_w = EID_ID ? ld_stream<NT>(w + my_e) : w[my_e];
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

template <int L, int NV>
__device__ __forceinline__ void atomic_flush(float* __restrict__ out, i64 row, float4 (&acc)[NV],
                                             int l) {
  constexpr i64 F4 = (i64)L * NV;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float* p = out + (row * F4 + v * L + l) * 4;
    atomicAdd(p + 0, acc[v].x);
    atomicAdd(p + 1, acc[v].y);
    atomicAdd(p + 2, acc[v].z);
    atomicAdd(p + 3, acc[v].w);
  }
}

// ---- CHUNK drivers (any chunk layout, no plan) ---------------------------------------------------
template <int L, int NV, bool H1, bool EDGE_B>
__global__ __launch_bounds__(kFastBlock) void k_sddmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ A, const float* __restrict__ B,
    float* __restrict__ y, i64 n_chunks, int h, int d4, int chunks_per_group) {
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
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
    sddmm_range<L, NV, H1, EDGE_B, false, false, i64>(a, indptr[c], indptr[c + 1], eid, indices, B,
                                                      y, h, d4, l);
  }
}

template <int L, int NV, bool H1>
__global__ __launch_bounds__(kFastBlock) void k_spmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ w, const float* __restrict__ X,
    float* __restrict__ out, i64 n_chunks, int h, int d4, int chunks_per_group) {
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
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
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
      if (dirty) {
        atomic_flush<L, NV>(out, cur_row, acc, l);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        dirty = false;
      }
      cur_row = r;
    }
    const i64 j0 = indptr[c], j1 = indptr[c + 1];
    if (j1 > j0) dirty = true;
    spmm_range<L, NV, H1, false, false, i64>(acc, j0, j1, eid, indices, w, X, h, hv, l);
  }
  if (dirty) atomic_flush<L, NV>(out, cur_row, acc, l);
}

// ---- WINDOW-SWEEP drivers (plan) -----------------------------------------------------------------
// vrow v = slots [wp[v], wp[W*V + v]) of row vr_row[v]; wp[w*V + v] = first slot of vrow v whose
// neighbour id is >= w * win_cols (ids ascend inside a row).  Group g of round r owns vrows
// [(r*n_groups + g)*K, +K).  LDS holds the K rows (A rows / partial sums) of every group.
struct SweepView {
  const int* wp;      // [(W+1) * V]
  const int* vr_row;  // [V]
  const int* idx32;   // [E]
  const int* eid32;   // [E] or nullptr when eid is the identity
  int V, W, K, rounds;
};

template <int L, int NV, bool H1, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_sddmm_sweep_f32(
    SweepView s, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y,
    int h, int d4) {
  extern __shared__ float4 lds[];
  constexpr int GPB = GroupCfg<L>::kGroupsPerBlock;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  const i64 n_groups = (i64)gridDim.x * GPB;
  const i64 gid = (i64)blockIdx.x * GPB + g_in_blk;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;  // [K][NV][L]
  for (int r = 0; r < s.rounds; ++r) {
    const i64 v0 = ((i64)r * n_groups + gid) * s.K;
    if (v0 >= s.V) break;
    const int nv = (s.V - v0) < s.K ? (int)(s.V - v0) : s.K;
    for (int k = 0; k < nv; ++k) {
      const i64 row = s.vr_row[v0 + k];
#pragma unroll
      for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = ld4(A, row * F4 + v * L + l);
    }
    for (int w = 0; w < s.W; ++w) {
      int lo_l = 0, hi_l = 0;
      if (l < nv) {
        lo_l = s.wp[(i64)w * s.V + v0 + l];
        hi_l = s.wp[(i64)(w + 1) * s.V + v0 + l];
      }
      for (int k = 0; k < nv; ++k) {
        const int lo = __shfl(lo_l, k, L), hi = __shfl(hi_l, k, L);
        if (hi <= lo) continue;
        float4 a[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = mine[(k * NV + v) * L + l];
        sddmm_range<L, NV, H1, false, EID_ID, true, int>(a, lo, hi, s.eid32, s.idx32, B, y, h, d4, l);
      }
    }
  }
}

template <int L, int NV, bool H1, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_spmm_sweep_f32(
    SweepView s, const float* __restrict__ wgt, const float* __restrict__ X,
    float* __restrict__ out, int h, int d4) {
  extern __shared__ float4 lds[];
  constexpr int GPB = GroupCfg<L>::kGroupsPerBlock;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  const i64 n_groups = (i64)gridDim.x * GPB;
  const i64 gid = (i64)blockIdx.x * GPB + g_in_blk;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;
  int hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = H1 ? 0 : (v * L + l) / d4;
  for (int r = 0; r < s.rounds; ++r) {
    const i64 v0 = ((i64)r * n_groups + gid) * s.K;
    if (v0 >= s.V) break;
    const int nv = (s.V - v0) < s.K ? (int)(s.V - v0) : s.K;
    for (int k = 0; k < nv; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = 0; w < s.W; ++w) {
      int lo_l = 0, hi_l = 0;
      if (l < nv) {
        lo_l = s.wp[(i64)w * s.V + v0 + l];
        hi_l = s.wp[(i64)(w + 1) * s.V + v0 + l];
      }
      for (int k = 0; k < nv; ++k) {
        const int lo = __shfl(lo_l, k, L), hi = __shfl(hi_l, k, L);
        if (hi <= lo) continue;
        float4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = mine[(k * NV + v) * L + l];
        spmm_range<L, NV, H1, EID_ID, true, int>(acc, lo, hi, s.eid32, s.idx32, wgt, X, h, hv, l);
#pragma unroll
        for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = acc[v];
      }
    }
    // pieces of one (long) row may live in several groups: merge with float atomics
    for (int k = 0; k < nv; ++k) {
      if (s.wp[v0 + k] == s.wp[(i64)s.W * s.V + v0 + k]) continue;
      float4 acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = mine[(k * NV + v) * L + l];
      atomic_flush<L, NV>(out, s.vr_row[v0 + k], acc, l);
    }
  }
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
