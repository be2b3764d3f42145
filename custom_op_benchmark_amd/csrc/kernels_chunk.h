// CHUNK drivers: work unit = the caller's chunk list; correct for ANY chunk layout, no plan needed.
// k_spmm_f32 keeps the running row sum in registers while the row id does not change and merges with native
// global_atomic_add_f32 (256 contiguous bytes per group); with a plan whose rows are sorted, rows owned by one
// lane group are stored (OWNED).
#pragma once
#include "kernels_base.h"

namespace graphop {

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

// OWNED: rows[] is non-decreasing (plan.row_owned), so a row whose first AND last chunk lie inside
// this group's chunk range is written by nobody else: its sum is stored, not added with atomics
// (the output is zero-filled beforehand either way).  Graphs of short rows -- one or two chunks per
// row, tens of millions of rows: the sharded papers100M-shape columns -- otherwise pay one 4*F-byte
// atomic flush per row at the memory-side atomic rate instead of a plain store.
template <int L, int NV, bool H1, bool OWNED>
__global__ __launch_bounds__(kFastBlock) void k_spmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ w, const float* __restrict__ X,
    float* __restrict__ out, i64 n_chunks, int h, int d4, int chunks_per_group) {
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  if (c0 >= c1) return;
  int hv[NV];  // head owning each of this lane's slots
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = H1 ? 0 : (v * L + l) / d4;
  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  // rows shared with the neighbouring groups (only these need atomics when OWNED)
  i64 row_before = -1, row_after = -1;
  if constexpr (OWNED) {
    if (c0 > 0) row_before = row[c0 - 1];
    if (c1 < n_chunks) row_after = row[c1];
  }
  auto flush = [&](i64 r) {
    if (OWNED && r != row_before && r != row_after) {
#pragma unroll
      for (int v = 0; v < NV; ++v) reinterpret_cast<float4*>(out)[r * F4 + v * L + l] = acc[v];
    } else {
      atomic_flush<L, NV>(out, r, acc, l);
    }
  };
  i64 cur_row = -1;
  bool dirty = false;
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
      if (dirty) {
        flush(cur_row);
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
  if (dirty) flush(cur_row);
}

}  // namespace graphop
