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

// FLAT form of the row-owning chunk driver (round 4; one head, one float4 per lane): the same chunk ranges per lane group,
// the same ownership / self-zeroing rules as k_spmm_f32<.., OWNED, SELFZERO>, but the group walks its SLOTS, not its chunks:
// ids and weights are fetched a batch (L slots) and two batches ahead, neighbour rows are requested U at a time -- two
// bursts in flight, across chunk AND batch boundaries -- and a chunk boundary is an event inside the slot loop (row / end
// slot of the next 2 L chunks sit in registers, one chunk per lane).  k_spmm_f32 finishes a chunk -- metadata, ids, weight,
// rows: four dependent round trips -- before it touches the next one, which is fine at 13+ slots per chunk and ruinous at
// one or two: the halo columns of a node-range shard (papers100M shape: 15 M columns of 1.35 slots each) had a lane group
// wait out the whole chain for a single row.
template <int L, bool SELFZERO>
__global__ __launch_bounds__(kFastBlock, 5) void k_spmm_flat_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float* __restrict__ w, const float* __restrict__ X,
    float* __restrict__ out, i64 n_chunks, int chunks_per_group, i64 n_out_rows) {
  constexpr i64 F4 = L;
  constexpr int U = 4;                               // rows per request burst; two bursts in flight
  static_assert(L % (2 * U) == 0, "the id window moves by whole double bursts");
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  if (c0 >= c1) return;
  using RT = typename std::conditional<SELFZERO, int, i64>::type;
  RT row_before = -1, row_after = -1;
  if (c0 > 0) row_before = (RT)row[c0 - 1];
  if (c1 < n_chunks) row_after = (RT)row[c1];
  const i64 jlo = indptr[c0];
  const int n = (int)(indptr[c1] - jlo);             // slots of this group (the host admits < 2^31 edges)
  float4 accv[1];
  float4& acc = accv[0];
  acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto flush = [&](RT r) {
    if (r != row_before && r != row_after) reinterpret_cast<float4*>(out)[(i64)r * F4 + l] = acc;
    else atomic_flush<L, 1>(out, (i64)r, accv, l);
  };
  auto zero_rows = [&](RT a, RT b) {                 // rows [a, b): nodes without edges
    for (RT g = a; g < b; ++g) reinterpret_cast<float4*>(out)[(i64)g * F4 + l] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  // chunk metadata: lane l holds the row and the END slot (relative to jlo) of chunk cb + l; the next L chunks are on their way
  i64 cb = c0;
  auto ld_meta = [&](i64 base, RT& r, int& e) {
    i64 c = base + l;
    if (c > c1 - 1) c = c1 - 1;
    r = (RT)row[c];
    e = (int)(indptr[c + 1] - jlo);
  };
  RT mr, mr2;
  int me, me2;
  ld_meta(cb, mr, me);
  ld_meta(cb + L, mr2, me2);
  int ci = 0;                                        // current chunk = cb + ci
  RT cur_row = SELFZERO ? row_before : (RT)-1;
  bool dirty = false, mine = false;
  int next_end = 0;
  auto enter_chunk = [&]() {                         // what k_spmm_f32 does at the top of a chunk
    const RT r = (RT)__shfl(mr, ci, L);
    if (r != cur_row || (SELFZERO && !mine)) {
      if (dirty || (SELFZERO && mine && cur_row != row_before && cur_row != row_after)) {
        flush(cur_row);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        dirty = false;
      }
      if constexpr (SELFZERO) zero_rows(cur_row + 1, r);
      cur_row = r;
      mine = true;
    }
    next_end = __shfl(me, ci, L);
  };
  auto advance = [&]() {
    if (++ci == L) {
      ci = 0;
      cb += L;
      mr = mr2;
      me = me2;
      ld_meta(cb + L, mr2, me2);
    }
    enter_chunk();
  };
  enter_chunk();
  // id window: lane l holds slot jb + l (s0, w0) and slot jb + L + l (s1, w1, e1); the (edge id, neighbour id) pairs of
  // the batch behind them and the weights of batch 1 are requested when the window moves, a batch's time ahead of their use
  auto ld_ids = [&](int jb_, int& e, int& s) {
    const int j = jb_ + l;
    e = -1;
    s = 0;
    if (j < n) { e = (int)eid[jlo + j]; s = (int)indices[jlo + j]; }
  };
  int jb = 0, s0, s1, s2, e0, e1, e2;
  ld_ids(0, e0, s0);
  ld_ids(L, e1, s1);
  ld_ids(2 * L, e2, s2);
  float w0 = e0 >= 0 ? w[e0] : 0.f;
  float w1 = e1 >= 0 ? w[e1] : 0.f;
  float4 xa[U], xb[U];
  auto request = [&](float4 (&x)[U], int j) {        // rows of slots j .. j + U - 1 (behind the last slot: the last one again)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = (j + u) < n ? (j + u) : (n - 1);
      const int dd = jj - jb;
      const int a = __shfl(s0, dd, L), b = __shfl(s1, dd - L, L);
      x[u] = ld4(X, (i64)(dd < L ? a : b) * F4 + l);
    }
  };
  auto consume = [&](const float4 (&x)[U], int j) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j + u < n) {
        while (j + u == next_end) advance();         // chunk boundary (empty chunks: several at one slot)
        const float ww = __shfl(w0, j + u - jb, L);
        acc.x = fmaf(ww, x[u].x, acc.x);
        acc.y = fmaf(ww, x[u].y, acc.y);
        acc.z = fmaf(ww, x[u].z, acc.z);
        acc.w = fmaf(ww, x[u].w, acc.w);
        dirty = true;
      }
    }
  };
  if (n > 0) request(xa, 0);
  for (int j = 0; j < n; j += 2 * U) {
    request(xb, j + U);
    consume(xa, j);
    request(xa, j + 2 * U);
    consume(xb, j + U);
    if (j + 2 * U == jb + L) {                       // every slot of batch 0 is consumed: move the window
      jb += L;
      s0 = s1; w0 = w1;
      s1 = s2; e1 = e2;
      ld_ids(jb + 2 * L, e2, s2);
      w1 = e1 >= 0 ? w[e1] : 0.f;
    }
  }
  while (cb + ci + 1 < c1) advance();                // chunks without slots behind the last slot
  if (dirty || (SELFZERO && cur_row != row_before && cur_row != row_after)) flush(cur_row);
  if constexpr (SELFZERO) { if (c1 == n_chunks) zero_rows(cur_row + 1, (RT)n_out_rows); }
}

// DUAL form (round 5; the sharded step's two column-major passes as one): two tables gathered through the SAME slot list,
// each with its own per-slot weight --
//   out0[row[c]] += sum_j w2[eid[j]].x * X0[indices[j]]      out1[row[c]] += sum_j w2[eid[j]].y * X1[indices[j]]
// (dV = col-SpMM(a, dO) and dK = col-SpMM(ds, Q): graphop_kernel.cu:151-163 and :100-112 over one column-major CSR).
// The weights come as ONE (E, 2) array: a slot's pair is one 8-byte random read instead of two 4-byte ones in two passes
// -- those reads are what separates the column-major passes of a node-range shard from the row-major ones (3.7-4 ms per
// pass at the papers100M shape, profiles/r4_experiments.txt section 8) -- and the ids, edge ids and chunk metadata are
// streamed once.  Same chunk ranges, ownership and self-zeroing rules as the single form; twice the rows in flight.
template <int L, bool SELFZERO, int U = 4, int BPC = 3>
__global__ __launch_bounds__(kFastBlock, BPC) void k_spmm_flat2_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ eid,
    const i64* __restrict__ indices, const float2* __restrict__ w, const float* __restrict__ X, const float* __restrict__ X1,
    float* __restrict__ out, float* __restrict__ out1, i64 n_chunks, int chunks_per_group, i64 n_out_rows) {
  constexpr i64 F4 = L;
  // U rows per request burst and table, two bursts in flight (U = 2 at four workgroups per CU measured the same: 51.1 / 55.4
  // of the two-launch time against 52.4 / 56.9)
  static_assert(L % (2 * U) == 0, "the id window moves by whole double bursts");
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  if (c0 >= c1) return;
  using RT = typename std::conditional<SELFZERO, int, i64>::type;
  RT row_before = -1, row_after = -1;
  if (c0 > 0) row_before = (RT)row[c0 - 1];
  if (c1 < n_chunks) row_after = (RT)row[c1];
  const i64 jlo = indptr[c0];
  const int n = (int)(indptr[c1] - jlo);             // slots of this group (the host admits < 2^31 edges)
  float4 accv[1], accv1[1];
  float4& acc = accv[0];
  float4& acc1 = accv1[0];
  acc = acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto flush = [&](RT r) {
    if (r != row_before && r != row_after) {
      reinterpret_cast<float4*>(out)[(i64)r * F4 + l] = acc;
      reinterpret_cast<float4*>(out1)[(i64)r * F4 + l] = acc1;
    } else {
      atomic_flush<L, 1>(out, (i64)r, accv, l);
      atomic_flush<L, 1>(out1, (i64)r, accv1, l);
    }
  };
  auto zero_rows = [&](RT a, RT b) {                 // rows [a, b): nodes without edges
    for (RT g = a; g < b; ++g) {
      reinterpret_cast<float4*>(out)[(i64)g * F4 + l] = make_float4(0.f, 0.f, 0.f, 0.f);
      reinterpret_cast<float4*>(out1)[(i64)g * F4 + l] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // chunk metadata: lane l holds the row and the END slot (relative to jlo) of chunk cb + l; the next L chunks are on their way
  i64 cb = c0;
  auto ld_meta = [&](i64 base, RT& r, int& e) {
    i64 c = base + l;
    if (c > c1 - 1) c = c1 - 1;
    r = (RT)row[c];
    e = (int)(indptr[c + 1] - jlo);
  };
  RT mr, mr2;
  int me, me2;
  ld_meta(cb, mr, me);
  ld_meta(cb + L, mr2, me2);
  int ci = 0;                                        // current chunk = cb + ci
  RT cur_row = SELFZERO ? row_before : (RT)-1;
  bool dirty = false, mine = false;
  int next_end = 0;
  auto enter_chunk = [&]() {                         // what k_spmm_f32 does at the top of a chunk
    const RT r = (RT)__shfl(mr, ci, L);
    if (r != cur_row || (SELFZERO && !mine)) {
      if (dirty || (SELFZERO && mine && cur_row != row_before && cur_row != row_after)) {
        flush(cur_row);
        acc = acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
        dirty = false;
      }
      if constexpr (SELFZERO) zero_rows(cur_row + 1, r);
      cur_row = r;
      mine = true;
    }
    next_end = __shfl(me, ci, L);
  };
  auto advance = [&]() {
    if (++ci == L) {
      ci = 0;
      cb += L;
      mr = mr2;
      me = me2;
      ld_meta(cb + L, mr2, me2);
    }
    enter_chunk();
  };
  enter_chunk();
  // id window: lane l holds slot jb + l (s0, w0) and slot jb + L + l (s1, w1, e1); the (edge id, neighbour id) pairs of
  // the batch behind them and the weights of batch 1 are requested when the window moves, a batch's time ahead of their use
  auto ld_ids = [&](int jb_, int& e, int& s) {
    const int j = jb_ + l;
    e = -1;
    s = 0;
    if (j < n) { e = (int)eid[jlo + j]; s = (int)indices[jlo + j]; }
  };
  int jb = 0, s0, s1, s2, e0, e1, e2;
  ld_ids(0, e0, s0);
  ld_ids(L, e1, s1);
  ld_ids(2 * L, e2, s2);
  // (an explicit branch: `e >= 0 ? w[e] : zero` becomes a load through a SELECTED pointer -- a flat load, and a stack slot for the zero)
  auto ld_w = [&](int e) { float2 v = make_float2(0.f, 0.f); if (e >= 0) v = w[e]; return v; };
  float2 w0 = ld_w(e0);
  float2 w1 = ld_w(e1);
  float4 xa[U], xb[U], ya[U], yb[U];
  auto request = [&](float4 (&x)[U], float4 (&y)[U], int j) {   // rows of slots j .. j + U - 1 of both tables
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = (j + u) < n ? (j + u) : (n - 1);
      const int dd = jj - jb;
      const int a = __shfl(s0, dd, L), b = __shfl(s1, dd - L, L);
      const i64 o = (i64)(dd < L ? a : b) * F4 + l;
      x[u] = ld4(X, o);
      y[u] = ld4(X1, o);
    }
  };
  auto consume = [&](const float4 (&x)[U], const float4 (&y)[U], int j) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j + u < n) {
        while (j + u == next_end) advance();         // chunk boundary (empty chunks: several at one slot)
        const float wx = __shfl(w0.x, j + u - jb, L), wy = __shfl(w0.y, j + u - jb, L);
        acc.x = fmaf(wx, x[u].x, acc.x);
        acc.y = fmaf(wx, x[u].y, acc.y);
        acc.z = fmaf(wx, x[u].z, acc.z);
        acc.w = fmaf(wx, x[u].w, acc.w);
        acc1.x = fmaf(wy, y[u].x, acc1.x);
        acc1.y = fmaf(wy, y[u].y, acc1.y);
        acc1.z = fmaf(wy, y[u].z, acc1.z);
        acc1.w = fmaf(wy, y[u].w, acc1.w);
        dirty = true;
      }
    }
  };
  if (n > 0) request(xa, ya, 0);
  for (int j = 0; j < n; j += 2 * U) {
    request(xb, yb, j + U);
    consume(xa, ya, j);
    request(xa, ya, j + 2 * U);
    consume(xb, yb, j + U);
    if (j + 2 * U == jb + L) {                       // every slot of batch 0 is consumed: move the window
      jb += L;
      s0 = s1; w0 = w1;
      s1 = s2; e1 = e2;
      ld_ids(jb + 2 * L, e2, s2);
      w1 = ld_w(e1);
    }
  }
  while (cb + ci + 1 < c1) advance();                // chunks without slots behind the last slot
  if (dirty || (SELFZERO && cur_row != row_before && cur_row != row_after)) flush(cur_row);
  if constexpr (SELFZERO) { if (c1 == n_chunks) zero_rows(cur_row + 1, (RT)n_out_rows); }
}

}  // namespace graphop
