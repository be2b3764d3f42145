// CHUNK drivers: work unit = the caller's chunk list; correct for ANY chunk layout, no plan needed.
// k_spmm_f32 keeps the running row sum in registers while the row id does not change and merges with native
// global_atomic_add_f32 (256 contiguous bytes per group); with a plan whose rows are sorted, rows owned by one
// lane group are stored (OWNED).
#pragma once
#include <type_traits>
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
// SELFZERO (with OWNED; round 4): the output arrives UNINITIALISED and this launch leaves every one of its n_out_rows
// rows defined -- rows a group owns are stored whether or not they hold slots, the rows between two chunk rows (nodes
// without edges) are zero-stored by the group that sees the gap (the one whose chunk follows it; the last group also
// takes the rows behind the last chunk), and only the rows cut between two groups' chunk ranges (merged by atomics)
// need zeros beforehand: k_zero_shared_rows, a few rows per group boundary.  The extended outputs of the sharded step
// (29 M rows x 512 B on the papers100M-shape shard) otherwise pay a 14.8 GB zero fill per pass in front of stores that
// overwrite nearly all of it.
template <int L, int NV>
__global__ __launch_bounds__(kFastBlock) void k_zero_shared_rows(const i64* __restrict__ row, float* __restrict__ out,
                                                                 i64 n_chunks, int chunks_per_group) {
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 g = (i64)blockIdx.x * (kFastBlock / L) + threadIdx.x / L;     // boundary in front of group g + 1
  const i64 c = (g + 1) * chunks_per_group;
  if (c >= n_chunks) return;
  const i64 r = row[c];
  if (r != row[c - 1]) return;                       // the row is not cut by this boundary
#pragma unroll
  for (int v = 0; v < NV; ++v) reinterpret_cast<float4*>(out)[r * F4 + v * L + l] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// (compiled for five workgroups per CU at one float4 per lane: the chunk loops are chains of dependent loads, their rate
// is resident waves x rows in flight; the self-zeroing form otherwise lands at 102 VGPRs = four)
template <int L, int NV, bool H1, bool OWNED, bool SELFZERO = false>
__global__ __launch_bounds__(kFastBlock, NV == 1 ? 5 : 1) void k_spmm_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ w, const float* __restrict__ X,
    float* __restrict__ out, i64 n_chunks, int h, int d4, int chunks_per_group, i64 n_out_rows = 0) {
  static_assert(!SELFZERO || OWNED, "self-zeroing needs row ownership");
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
  // rows shared with the neighbouring groups (only these need atomics when OWNED).  (Row ids as 32-bit values in the
  // self-zeroing form -- its host predicate admits outputs below 2^31 rows only -- : the extra bookkeeping then fits
  // the register budget of five workgroups per CU.)
  using RT = typename std::conditional<SELFZERO, int, i64>::type;
  RT row_before = -1, row_after = -1;
  if constexpr (OWNED) {
    if (c0 > 0) row_before = (RT)row[c0 - 1];
    if (c1 < n_chunks) row_after = (RT)row[c1];
  }
  auto flush = [&](RT r) {
    if (OWNED && r != row_before && r != row_after) {
#pragma unroll
      for (int v = 0; v < NV; ++v) reinterpret_cast<float4*>(out)[(i64)r * F4 + v * L + l] = acc[v];
    } else {
      atomic_flush<L, NV>(out, (i64)r, acc, l);
    }
  };
  auto zero_rows = [&](RT a, RT b) {   // rows [a, b): nodes without edges
    for (RT g = a; g < b; ++g)
#pragma unroll
      for (int v = 0; v < NV; ++v) reinterpret_cast<float4*>(out)[(i64)g * F4 + v * L + l] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // SELFZERO: `cur_row` starts at the row in front of this group's chunks (-1 for group 0), so the first row change also
  // zero-stores the gap in front of the group's first row; it never flushes that foreign row (`mine` is false until then)
  RT cur_row = SELFZERO ? row_before : (RT)-1;
  bool dirty = false, mine = false;
  for (i64 c = c0; c < c1; ++c) {
    const RT r = (RT)row[c];
    if (r != cur_row || (SELFZERO && !mine)) {
      // SELFZERO: an owned row is stored even when its chunks hold no slots (a shared one only adds, and only when dirty)
      if (dirty || (SELFZERO && mine && cur_row != row_before && cur_row != row_after)) {
        flush(cur_row);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        dirty = false;
      }
      if constexpr (SELFZERO) zero_rows(cur_row + 1, r);
      cur_row = r;
      mine = true;
    }
    const i64 j0 = indptr[c], j1 = indptr[c + 1];
    if (j1 > j0) dirty = true;
    spmm_range<L, NV, H1, false, false, i64>(acc, j0, j1, eid, indices, w, X, h, hv, l);
  }
  if (dirty || (SELFZERO && cur_row != row_before && cur_row != row_after)) flush(cur_row);
  if constexpr (SELFZERO) { if (c1 == n_chunks) zero_rows(cur_row + 1, (RT)n_out_rows); }
}

}  // namespace graphop
