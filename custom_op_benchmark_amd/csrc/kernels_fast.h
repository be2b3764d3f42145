// Fast fp32 kernels for gfx950 (wave64).  Feature rows are walked as float4 (16 B / lane);
// a node row of F = h*d floats is covered by a GROUP of L lanes x NV float4 slots
// (F = 4*L*NV).  For the headline shape (d = 64, h = 1) L = 16: one 256-B row per 16-lane
// DPP row, four edges in flight per wave instruction, reductions by DPP inside the row.
//
// Inner loops:
//  * sddmm_range / spmm_range   (chunk drivers) per-edge dots / weighted accumulation over a slot
//    range [lo, hi) of one row: ids of up to 16 slots are loaded coalesced by the group and
//    broadcast with ds_bpermute; U neighbour rows (16 B/lane each) are in flight per group.
//  * sddmm_strip / spmm_strip   (window drivers) the K granules a group owns inside one window,
//    walked as ONE flat slot list in full 16-slot batches with the ids of the next batch(es) in
//    flight behind the current batch's row requests; rows come in through scalar-base +
//    32-bit-offset loads; the SDDMM batch is a single basic block.
//
// Drivers:
//  * k_sddmm_f32 / k_spmm_f32   CHUNK drivers: work unit = the caller's chunk list; correct for
//    any chunk layout (SpMM keeps the running row sum in registers while the row id does not
//    change and merges with native global_atomic_add_f32, 256 contiguous bytes per group).
//  * COLUMN-WINDOW drivers (need a plan): the gathered table is cut into W column windows (<= 4 MB,
//    the size of an XCD's L2; 32 MB Infinity-Cache windows for tables beyond 128 MB), a row's slots
//    inside a window are one contiguous range.  Two loop orders over (window, vrow):
//      k_sddmm_wown_f32 / k_spmm_wown_f32    (default) every XCD owns the windows x, x+8, ... and
//        its waves pull (window, tile of vrows) tasks from a per-XCD queue; a window lives in one
//        L2 only; SpMM hands each granule sum to the output with a dense atomic flush;
//      k_sddmm_sweep_f32 / k_spmm_sweep_f32  every lane group owns K vrows (A rows / partial sums
//        in LDS) and all walk the windows together, kept within `drift` windows by a per-XCD
//        soft barrier (SweepPacer).
//  * k_softmax_*_seg   per-row softmax / its backward over row segments from the plan: rows up
//    to G*16 (forward) / G*32 (backward) items in registers, rows above 1024 / 2048 slots one
//    workgroup each (same launch), shuffle / LDS reductions, no atomics, no scratch.
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
typedef float vfloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* base, i64 f4_index) {   // streamed once: keep it out of the caches
  const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(base) + f4_index);
  return make_float4(v.x, v.y, v.z, v.w);
}

// Row slice of a gathered table.  OFF32: the table is < 4 GiB, so the byte offset fits 32 bits and
// the load uses the scalar-base + 32-bit vector-offset form (no 64-bit VALU address math).
template <bool OFF32>
__device__ __forceinline__ float4 ld_row(const float* base, int src, int f4_in_row, int row_f4) {
  if constexpr (OFF32) {
    const unsigned off = ((unsigned)src * (unsigned)row_f4 + (unsigned)f4_in_row) * 16u;
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
  } else {
    return reinterpret_cast<const float4*>(base)[(i64)src * row_f4 + f4_in_row];
  }
}

// 16 bytes at base + off, off < 4 GiB: scalar base + 32-bit vector offset (no 64-bit VALU add)
__device__ __forceinline__ float4 ld4_off(const float* base, unsigned off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)off);
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
__device__ __forceinline__ void atomic_flush(float* __restrict__ out, i64 row,
                                             const float4 (&acc)[NV], int l) {
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

// Same sum, but every atomic wave-instruction covers CONSECUTIVE dwords of the row (a group's L
// lanes add L consecutive floats = whole 64-B memory-side atomic requests) instead of one dword
// out of every 16 B: for flushes that are frequent enough to load the memory-side atomic units.
// Group-uniform call (all L lanes of the group active).
template <int L, int NV>
__device__ __forceinline__ void atomic_flush_dense(float* __restrict__ out, i64 row,
                                                   const float4 (&acc)[NV], int l) {
  constexpr i64 F = 4LL * L * NV;
  float* base = out + row * F;
  if constexpr (L >= 4) {
    const int comp = l & 3;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int src = i * (L / 4) + (l >> 2);
        const float x = __shfl(acc[v].x, src, L), y = __shfl(acc[v].y, src, L);
        const float z = __shfl(acc[v].z, src, L), w = __shfl(acc[v].w, src, L);
        const float val = comp == 0 ? x : (comp == 1 ? y : (comp == 2 ? z : w));
        atomicAdd(base + (v * 4 + i) * L + l, val);
      }
    }
  } else {
    atomic_flush<L, NV>(out, row, acc, l);
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


// Touch the 128-B lines of this lane's granule [lo, lo + n) of a 4-byte stream (ids, edge ids,
// row-major weights) so that the per-batch loads of the strip find them in L2.  Vector-memory loads
// return in issue order: a per-batch id load that misses L2 holds back the 16 row loads issued
// behind it for an HBM latency (tools/microbench/l2_gather_ids.hip: 23 TB/s of row gathers with the
// ids streamed from HBM, 30.5 TB/s with an L2-resident id stream).  Issued once per task, in front
// of the first id load the strip has to wait for anyway, the misses of a whole granule overlap.
// Measured on Reddit-shape: SDDMM-type passes 1.79-1.84 -> 1.73-1.75 ms; the SpMM-type and fused
// passes (windows of twice the L2 size: the touched lines evict rows) got 3-7 % SLOWER, so only the
// SDDMM strip uses it (tuning knob touch_sddmm).
// Covers granules of up to ~65 slots (three lines); longer ones keep some cold lines (speed only).
// The values must be `retire`d at the end of the strip (keeps the landing registers reserved).
struct LineTouch {
  int a, b, c;
  template <typename T>
  __device__ __forceinline__ void issue(const T* __restrict__ base, int lo, int n) {
    static_assert(sizeof(T) == 4, "4-byte streams");
    a = b = c = 0;
    if (n > 0) {
      const int* p = reinterpret_cast<const int*>(base);
      a = p[lo];
      b = p[lo + (n >> 1)];
      c = p[lo + n - 1];
    }
  }
  __device__ __forceinline__ void retire() const { asm volatile("; touched %0 %1 %2" ::"v"(a), "v"(b), "v"(c)); }
};

// ---- STRIP inner loops (window-sweep drivers) -----------------------------------------------------
// A strip is what one lane group does in one window: lane k < nv owns granule k = slots
// [lo_l, lo_l + n_l) of its vrow k.  The K granules are walked as ONE flat slot list in full
// batches of SB slots (no per-granule round-up), the ids of the next batch are fetched while the
// current batch's rows are in flight, and all SB rows of a batch are requested before any is used.
template <int L, int NV>
struct StripCfg {
  static constexpr int kMaxBatch = NV == 1 ? 16 : (NV == 2 ? 8 : 4);   // 64 VGPRs of rows in flight
  static constexpr int SB = L < kMaxBatch ? L : kMaxBatch;
};

struct StripMap {   // flat slot j of the strip -> (granule k, slot e); all group-local
  int P;            // inclusive prefix of granule lengths (lane k)
  int Pex;          // exclusive prefix
  int lo;           // granule start (lane k)
  int total;
  template <int L>
  __device__ __forceinline__ void init(int lo_l, int n_l, int l) {
    lo = lo_l;
    P = n_l;
#pragma unroll
    for (int off = 1; off < L; off <<= 1) {
      const int t = __shfl_up(P, off, L);
      if (l >= off) P += t;
    }
    Pex = P - n_l;
    total = __shfl(P, L - 1, L);
  }
  // granule of flat slot j (j < total): number of granules whose inclusive prefix is <= j
  template <int L>
  __device__ __forceinline__ void locate(int j, int& k, int& e) const {
    k = 0;
#pragma unroll
    for (int step = L / 2; step >= 1; step >>= 1) {
      const int pv = __shfl(P, k + step - 1, L);
      if (pv <= j) k += step;
    }
    e = __shfl(lo, k, L) + (j - __shfl(Pex, k, L));
  }
};

// SDDMM strip: y[eid[e]*h + head] = <A_k, B[idx[e]]> ; A rows of the group's K vrows are in LDS.
// `stage_rows()` is called once the ids of the first batch have been requested: the caller puts
// the A rows into LDS there, so their fetch overlaps the id fetch instead of preceding it.
struct NoStage { __device__ __forceinline__ void operator()() const {} };
template <int L, int NV, bool H1, bool EID_ID, bool OFF32, typename Stage = NoStage>
__device__ __forceinline__ void sddmm_strip(const float4* __restrict__ rowsA, int lo_l, int n_l,
                                            const int* __restrict__ eid32,
                                            const int* __restrict__ idx32,
                                            const float* __restrict__ B, float* __restrict__ y,
                                            int h, int d4, int l, Stage&& stage_rows = Stage(),
                                            int touch = 0) {
  constexpr int SB = StripCfg<L, NV>::SB;
  constexpr i64 F4 = (i64)L * NV;
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  float4 a[NV];
  LineTouch t_idx, t_eid;
  t_idx.issue(idx32, lo_l, (touch & 1) ? n_l : 0);
  if constexpr (!EID_ID) t_eid.issue(eid32, lo_l, (touch & 2) ? n_l : 0);
  // prefetch batch 0
  int nk = 0, ne = -1, nsrc = 0;
  {
    const int j = l;
    int e;
    m.locate<L>(j < m.total ? j : m.total - 1, nk, e);   // every lane takes part in the shuffles
    if (l < SB && j < m.total) {
      ne = EID_ID ? e : (*(eid32 + e));
      nsrc = (*(idx32 + e));
    }
  }
  stage_rows();
  // h == 1: the batch's 16 results are stored AFTER the next batch's rows have been requested.
  // vmcnt retires in issue order, so a store issued ahead of those loads would have to be
  // acknowledged (a write to HBM) before their data could be used.
  float prev_res = 0.f;
  int prev_e = -1;
  const char* lds_l = reinterpret_cast<const char*>(rowsA) + l * 16;
  for (int jb = 0; jb < m.total; jb += SB) {
    const int nb = (m.total - jb) < SB ? (m.total - jb) : SB;
    // Owner lanes turn (vrow k, neighbour id) into byte offsets once; slots beyond nb keep valid
    // (stale or zero) ids, so the batch needs no per-slot clamping: their rows are fetched and
    // dotted like the others and only the final store is masked.
    const int my_e = ne;
    const unsigned my_koff = (unsigned)nk * (unsigned)(F4 * 16);
    const unsigned my_off = OFF32 ? (unsigned)nsrc * (unsigned)(F4 * 16) : (unsigned)nsrc;
    float4 b[SB][NV];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (OFF32)
          b[u][v] = ld4_off(B, o + (unsigned)((v * L + l) * 16));
        else
          b[u][v] = reinterpret_cast<const float4*>(B)[(i64)o * F4 + v * L + l];
      }
    });
    if constexpr (H1) {
      if (prev_e >= 0) y[prev_e] = prev_res;
    }
    // ids of the next batch (issued after the row requests so they stay in flight behind them)
    ne = -1;
    {
      const int j = jb + SB + l;
      int e;
      m.locate<L>(j < m.total ? j : m.total - 1, nk, e);
      if (l < SB && j < m.total) {
        ne = EID_ID ? e : (*(eid32 + e));
        nsrc = (*(idx32 + e));
      }
    }
    float res = 0.f;
    float part[H1 ? SB : 1];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const bool live = u < nb;
      // A row of this slot's vrow straight from LDS every time: no branch, so the batch stays one
      // basic block and the 16 dot products / reductions interleave
      const unsigned ko = group_bcast<L, u>(my_koff);
#pragma unroll
      for (int v = 0; v < NV; ++v) a[v] = *reinterpret_cast<const float4*>(lds_l + ko + v * L * 16);
      if constexpr (H1) {
        float p = dot4(a[0], b[u][0]);
#pragma unroll
        for (int v = 1; v < NV; ++v) p += dot4(a[v], b[u][v]);
        part[u] = p;
      } else {
        const i64 e = group_bcast<L, u>(my_e);
        if (d4 >= L) {
          const int sph = d4 / L;
          float acc = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc += dot4(a[v], b[u][v]);
            if ((v + 1) % sph == 0) {
              const float sum = group_sum<L>(acc);
              if (live && l == 0) y[e * h + v / sph] = sum;
              acc = 0.f;
            }
          }
        } else {
          const int hps = L / d4;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const float sum = group_sum_rt(dot4(a[v], b[u][v]), d4);
            if (live && (l % d4) == 0) y[e * h + v * hps + l / d4] = sum;
          }
        }
      }
    });
    if constexpr (H1) res = group_dots_to_owner<L, SB>(part, l);
    if constexpr (H1) {
      prev_res = res;
      prev_e = l < nb ? my_e : -1;
    }
  }
  if constexpr (H1) {
    if (prev_e >= 0) y[prev_e] = prev_res;
  }
  t_idx.retire();
  if constexpr (!EID_ID) t_eid.retire();
}

// ---- staged id streams (dealt layouts) ---------------------------------------------------------------
// With a dealt layout (plan.hip, Sweep::Dealt) the neighbour ids -- and edge ids -- of a lane group's
// strip are ONE contiguous 16-byte-aligned run starting at pos0.  IdStage fetches them a segment
// (SEG slots) at a time with dwordx4 loads, parks the segment in the group's LDS ring (two segments
// per stream) and hands them out by flat slot: between two batches of row requests the vector memory
// pipeline then sees no small load of ids (tools/microbench/sweep_model.hip: 1.66 -> 1.18 ms for the
// Reddit-shape edge count when every gather hits L2; the shipped SDDMM passes gain 4-5 %).
// Protocol: init() once; advance(jb) at every batch start (it acts when jb reaches the middle of a
// segment: the next segment becomes readable, the one after is requested); id(j) / eid(j) for any
// flat slot j in [jb, jb + SEG / 2].
template <int L, int NS = 1>
struct StageCfg {
  static constexpr int kMin = NS == 1 ? 128 : 64;         // two streams: half the segment, same registers
  static constexpr int SEG = 4 * L > kMin ? 4 * L : kMin; // slots per segment (power of two)
  static constexpr int NQ = SEG / (4 * L);                // dwordx4 per lane, segment and stream
  static constexpr int kLdsIntsPerGroup = NS * 2 * SEG;
};
template <int L, int NS>
struct IdStage {
  static constexpr int SEG = StageCfg<L, NS>::SEG, NQ = StageCfg<L, NS>::NQ;
  typedef int vint4 __attribute__((ext_vector_type(4)));   // (HIP's int4 struct keeps the array in scratch)
  vint4 nx[NS][NQ];
  const int* base[NS];   // wave-uniform
  int at;                // this lane's first id of segment 0 (element index: pos0 + 4 * lane)
  int* buf;              // [NS][2][SEG]
  int l, total;
  __device__ __forceinline__ void load(int seg) {
    if (seg >= total) return;   // group-uniform
    static_for<NS>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      static_for<NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        nx[st][q] = *reinterpret_cast<const vint4*>(base[st] + ((i64)at + seg + q * 4 * L));
      });
    });
  }
  __device__ __forceinline__ void park(int seg) {
    if (seg >= total) return;
    static_for<NS>([&](auto sc) {
      constexpr int st = decltype(sc)::value;
      static_for<NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        *reinterpret_cast<vint4*>(buf + (st * 2 + ((seg / SEG) & 1)) * SEG + q * 4 * L + l * 4) = nx[st][q];
      });
    });
  }
  __device__ __forceinline__ void init(const int* __restrict__ ids_w, const int* __restrict__ eids_w, int pos0,
                                       int* group_buf, int lane, int n_total) {
    buf = group_buf; l = lane; total = n_total;
    base[0] = ids_w;
    if constexpr (NS > 1) base[1] = eids_w;
    at = pos0 + lane * 4;
    load(0);
    park(0);
    load(SEG);
  }
  __device__ __forceinline__ void advance(int jb) {
    if ((jb & (SEG - 1)) == SEG / 2) {
      const int seg = jb & ~(SEG - 1);
      park(seg + SEG);
      load(seg + 2 * SEG);
    }
  }
  __device__ __forceinline__ int id(int j) const { return buf[((j / SEG) & 1) * SEG + (j & (SEG - 1))]; }
  __device__ __forceinline__ int eid(int j) const { return buf[(2 + ((j / SEG) & 1)) * SEG + (j & (SEG - 1))]; }
};

// Staged SDDMM strip (h == 1, identity eid, 32-bit offsets; dealt layout).
template <int L, int NV, typename Stage>
__device__ __forceinline__ void sddmm_strip_staged(const float4* __restrict__ rowsA, int lo_l, int n_l,
                                                   int pos0, const int* __restrict__ ids_w,
                                                   int* __restrict__ idbuf, const float* __restrict__ B,
                                                   float* __restrict__ y, int l, Stage&& stage_rows) {
  constexpr int SB = StripCfg<L, NV>::SB;
  constexpr i64 F4 = (i64)L * NV;
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  IdStage<L, 1> ids;
  ids.init(ids_w, nullptr, pos0, idbuf, l, m.total);
  stage_rows();
  // The results of kStoreBatch batches are stored together, behind the row requests of the next batch:
  // vmcnt retires in issue order and a store is acknowledged later than an L2-hit load returns, so
  // every store instruction between two batches of row requests delays the rows behind it once;
  // kStoreBatch stores issued back to back share that delay.  Nontemporal: 1.50 -> 1.46 ms per pass on the Reddit
  // shape once the stores are batched (plain stores were the faster form while there was one per batch; write-through
  // agent-scope stores measure 1.73).
  // (measured at 256-B rows; 1-KB rows got slower with it, 8.1 -> 10.0 ms per pass at d = 256, and keep one plain store per batch)
  constexpr int kStoreBatch = (NV == 1 && L == 16) ? 4 : 1;
  float held_res[kStoreBatch];
  int held_e[kStoreBatch];
#pragma unroll
  for (int q = 0; q < kStoreBatch; ++q) { held_res[q] = 0.f; held_e[q] = -1; }
  int n_held = 0;   // group-uniform
  auto flush_results = [&]() {
#pragma unroll
    for (int q = 0; q < kStoreBatch; ++q) {
      if (held_e[q] >= 0) {
        if constexpr (kStoreBatch > 1) __builtin_nontemporal_store(held_res[q], y + held_e[q]);
        else y[held_e[q]] = held_res[q];
      }
      held_e[q] = -1;
    }
    n_held = 0;
  };
  const char* lds_l = reinterpret_cast<const char*>(rowsA) + l * 16;
  for (int jb = 0; jb < m.total; jb += SB) {
    const int nb = (m.total - jb) < SB ? (m.total - jb) : SB;
    ids.advance(jb);
    const int j = (jb + l) < m.total ? jb + l : m.total - 1;   // lanes past the end re-read the last slot
    const int nsrc = ids.id(j);
    int nk, e;
    m.locate<L>(j, nk, e);
    const int my_e = (l < nb) ? e : -1;
    const unsigned my_koff = (unsigned)nk * (unsigned)(F4 * 16);
    const unsigned my_off = (unsigned)nsrc * (unsigned)(F4 * 16);
    float4 b[SB][NV];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) b[u][v] = ld4_off(B, o + (unsigned)((v * L + l) * 16));
    });
    if (n_held == kStoreBatch) flush_results();
    float part[SB];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned ko = group_bcast<L, u>(my_koff);
      float4 av[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) av[v] = *reinterpret_cast<const float4*>(lds_l + ko + v * L * 16);
      float p = dot4(av[0], b[u][0]);
#pragma unroll
      for (int v = 1; v < NV; ++v) p += dot4(av[v], b[u][v]);
      part[u] = p;
    });
    const float res = group_dots_to_owner<L, SB>(part, l);
#pragma unroll
    for (int q = 0; q < kStoreBatch; ++q)
      if (q == n_held) { held_res[q] = res; held_e[q] = my_e; }
    ++n_held;
  }
  flush_results();
}

// Several heads: a head's d floats lie in D4 = d / 4 consecutive lanes.  p[u] = this lane's partial of slot u's
// dot products (16 slots); on return r[i] = the total of slot R * (l % min(D4, 16)) + i for the lane's head,
// R = 16 / min(D4, 16) -- the transpose-reduce of group_dots_to_owner, stopped at the head's width.
template <int D4>
__device__ __forceinline__ void heads_dots_to_owners(float (&p)[16], int l, float (&r)[16 / (D4 < 16 ? D4 : 16)]) {
  static_assert(D4 == 4 || D4 == 8 || D4 == 16 || D4 == 32, "lanes per head");
  if constexpr (D4 >= 16) {
    float v = group_dots_to_owner<16, 16>(p, l);
    if constexpr (D4 == 32) v += __shfl_xor(v, 16);
    r[0] = v;
  } else if constexpr (D4 == 8) {
    float t8[8], t4[4];
    const bool b2 = l & 4, b1 = l & 2, b0 = l & 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float keep = b2 ? p[u + 8] : p[u], send = b2 ? p[u] : p[u + 8];
      t8[u] = keep + dpp_f32<0x141>(send);      // row_half_mirror = lane ^ 7
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float keep = b1 ? t8[u + 4] : t8[u], send = b1 ? t8[u] : t8[u + 4];
      t4[u] = keep + dpp_f32<0x4E>(send);       // lane ^ 2
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float keep = b0 ? t4[u + 2] : t4[u], send = b0 ? t4[u] : t4[u + 2];
      r[u] = keep + dpp_f32<0xB1>(send);        // lane ^ 1
    }
  } else {
    float t8[8];
    const bool b1 = l & 2, b0 = l & 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float keep = b1 ? p[u + 8] : p[u], send = b1 ? p[u] : p[u + 8];
      t8[u] = keep + dpp_f32<0x4E>(send);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float keep = b0 ? t8[u + 4] : t8[u], send = b0 ? t8[u] : t8[u + 4];
      r[u] = keep + dpp_f32<0xB1>(send);
    }
  }
}

// Staged SDDMM strip for H = L / D4 heads (identity eid, 32-bit offsets, dealt layout, one float4 per lane):
// y[e * H + head] = <A_k[head], B[idx[e]][head]>  (graphop_kernel.cu:40-55, :135-149).  After the reduce a
// batch's 16 x H results sit R per lane, one head per lane.  Heads of up to 8 lanes pass them through `scr`
// (16 x H floats of the lane group's LDS) so that a lane holds 16 / D4 CONSECUTIVE floats of y -- the heads of
// one edge -- and the batch leaves in ONE 8- or 16-byte store instruction behind the next batch's row
// requests (the unstaged strip stores per slot: 16 store instructions between two batches of row requests;
// R scalar stores per batch measured 2.11 ms per pass at h = 4, d = 16 against 1.46 at h = 1, d = 64).
template <int L, int D4, typename Stage>
__device__ __forceinline__ void sddmm_strip_staged_heads(const float4* __restrict__ rowsA, int lo_l, int n_l,
                                                         int pos0, const int* __restrict__ ids_w,
                                                         int* __restrict__ idbuf, float* __restrict__ scr,
                                                         const float* __restrict__ B, float* __restrict__ y,
                                                         int l, Stage&& stage_rows) {
  constexpr int SB = 16, H = L / D4, R = 16 / (D4 < 16 ? D4 : 16);
  constexpr bool VIA_LDS = D4 <= 8;                          // R = 4 or 2 results per lane -> one float4 / float2
  static_assert(StripCfg<L, 1>::SB == SB && L % D4 == 0 && H >= 2, "16-slot batches, whole heads");
  constexpr i64 F4 = L;
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  IdStage<L, 1> ids;
  ids.init(ids_w, nullptr, pos0, idbuf, l, m.total);
  stage_rows();
  const int head = l / D4;
  const int slot0 = R * (l % (D4 < 16 ? D4 : 16));          // first of the R slots whose totals this lane receives
  const bool owner = D4 <= 16 || (l & 16) == 0;             // 32 lanes per head: both 16-lane rows hold the total
  const int out_slot = VIA_LDS ? (l * R) / H : 0;            // VIA_LDS: this lane stores floats [l * R, l * R + R) of the batch
  float held[R];
  i64 held_at[R];                                            // VIA_LDS: only [0] (first float of the vector)
#pragma unroll
  for (int i = 0; i < R; ++i) { held[i] = 0.f; held_at[i] = -1; }
  const char* lds_l = reinterpret_cast<const char*>(rowsA) + l * 16;
  auto store_held = [&]() {
    if constexpr (VIA_LDS) {
      if (held_at[0] >= 0) {
        if constexpr (R == 4) *reinterpret_cast<float4*>(y + held_at[0]) = make_float4(held[0], held[1], held[2], held[3]);
        else *reinterpret_cast<float2*>(y + held_at[0]) = make_float2(held[0], held[1]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < R; ++i)
        if (held_at[i] >= 0) y[held_at[i]] = held[i];
    }
  };
  for (int jb = 0; jb < m.total; jb += SB) {
    const int nb = (m.total - jb) < SB ? (m.total - jb) : SB;
    ids.advance(jb);
    const int j = (jb + l) < m.total ? jb + l : m.total - 1;   // lanes past the end re-read the last slot
    const int nsrc = ids.id(j);
    int nk, e;
    m.locate<L>(j, nk, e);
    const int my_e = (l < nb) ? e : -1;
    const unsigned my_koff = (unsigned)nk * (unsigned)(F4 * 16);
    const unsigned my_off = (unsigned)nsrc * (unsigned)(F4 * 16);
    float4 b[SB];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      b[u] = ld4_off(B, group_bcast<L, u>(my_off) + (unsigned)(l * 16));
    });
    store_held();                                            // the previous batch's results, behind the row requests
    float part[SB];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned ko = group_bcast<L, u>(my_koff);
      part[u] = dot4(*reinterpret_cast<const float4*>(lds_l + ko), b[u]);
    });
    heads_dots_to_owners<D4>(part, l, held);
    if constexpr (VIA_LDS) {
      // [slot][head] through LDS (operations of a wave execute in order; the scratch is this lane group's own)
#pragma unroll
      for (int i = 0; i < R; ++i) scr[(slot0 + i) * H + head] = held[i];
      if constexpr (R == 4) {
        const float4 t = *reinterpret_cast<const float4*>(scr + l * 4);
        held[0] = t.x; held[1] = t.y; held[2] = t.z; held[3] = t.w;
      } else {
        const float2 t = *reinterpret_cast<const float2*>(scr + l * 2);
        held[0] = t.x; held[1] = t.y;
      }
      const int es = __shfl(my_e, out_slot, L);
      held_at[0] = es >= 0 ? (i64)es * H + (l * R) % H : -1;
    } else {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int es = __shfl(my_e, slot0 + i, L);
        held_at[i] = (es >= 0 && owner) ? (i64)es * H + head : -1;
      }
    }
  }
  store_held();
}

// `sink(k, acc)` receives the finished partial sum of granule k (group-uniform call).
template <int L, int NV, bool H1, bool EID_ID, bool OFF32, typename Sink>
__device__ __forceinline__ void spmm_strip(Sink&& sink, int lo_l, int n_l,
                                           const int* __restrict__ eid32,
                                           const int* __restrict__ idx32,
                                           const float* __restrict__ w,
                                           const float* __restrict__ X, int h,
                                           const int (&hv)[NV], int l) {
  constexpr int SB = StripCfg<L, NV>::SB;
  constexpr i64 F4 = (i64)L * NV;
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k_cur = -1;
  auto spill = [&]() {
    if (k_cur >= 0) {
      sink(k_cur, acc);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // Id pipeline.  Stage A (flat slot -> slot index, eid / idx loads) runs one batch ahead; when eid
  // is not the identity the weight w[eid] is a second dependent long-latency load, so stage A runs
  // two batches ahead and stage B (the weight) one batch ahead.
  struct Pre { int k, e, src; float w; };
  auto stage_a = [&](int jbase, Pre& p) {
    const int j = jbase + l;
    int e;
    m.locate<L>(j < m.total ? j : m.total - 1, p.k, e);   // every lane takes part in the shuffles
    p.e = -1; p.src = 0; p.w = 0.f;
    if (l < SB) {
      // slots past the end re-read the strip's last neighbour id with weight 0 (a row that is in
      // the sum anyway), so the batch loop needs no per-slot clamping
      p.src = (*(idx32 + e));
      if (j < m.total) {
        p.e = EID_ID ? e : (*(eid32 + e));
        if constexpr (H1 && EID_ID) p.w = (*(w + p.e));
      }
    }
  };
  auto stage_b = [&](Pre& p) {
    if constexpr (H1 && !EID_ID) p.w = p.e >= 0 ? w[p.e] : 0.f;
  };
  Pre p1, p2;
  stage_a(0, p1);
  stage_b(p1);
  if constexpr (!EID_ID) stage_a(SB, p2);
  for (int jb = 0; jb < m.total; jb += SB) {
    const int nb = (m.total - jb) < SB ? (m.total - jb) : SB;
    const int my_k = p1.k, my_e = p1.e;
    const unsigned my_off = OFF32 ? (unsigned)p1.src * (unsigned)(F4 * 16) : (unsigned)p1.src;
    const float my_w = p1.w;
    float4 x[SB][NV];
    float wt[H1 ? 1 : SB][H1 ? 1 : NV];   // per-head weights are loads and must be issued early
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (OFF32)
          x[u][v] = ld4_off(X, o + (unsigned)((v * L + l) * 16));
        else
          x[u][v] = reinterpret_cast<const float4*>(X)[(i64)o * F4 + v * L + l];
      }
      if constexpr (!H1) {
        const i64 e = group_bcast<L, u>(my_e);
#pragma unroll
        for (int v = 0; v < NV; ++v) wt[u][v] = u < nb ? w[e * h + hv[v]] : 0.f;
      }
    });
    // ids of the following batches (issued after the row requests so they stay in flight behind them)
    if constexpr (EID_ID) {
      stage_a(jb + SB, p1);
    } else {
      p1 = p2;
      stage_b(p1);
      stage_a(jb + 2 * SB, p2);
    }
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int kt = group_bcast<L, u>(my_k);
      if (kt != k_cur) {   // group-uniform
        spill();
        k_cur = kt;
      }
      float w1 = 0.f;
      if constexpr (H1) w1 = group_bcast<L, u>(my_w);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float ww = H1 ? w1 : wt[H1 ? 0 : u][H1 ? 0 : v];
        acc[v].x = fmaf(ww, x[u][v].x, acc[v].x);
        acc[v].y = fmaf(ww, x[u][v].y, acc[v].y);
        acc[v].z = fmaf(ww, x[u][v].z, acc[v].z);
        acc[v].w = fmaf(ww, x[u][v].w, acc[v].w);
      }
    });
  }
  spill();
}

// Staged SpMM strip (h == 1, 32-bit offsets; dealt layout): spmm_strip with the neighbour ids (and,
// when eid is not the identity, the edge ids) taken from IdStage instead of per-batch loads.  The
// weights are still loads: w[e] of the granule's slot run (identity eid) or the gather w[eid].
template <int L, int NV, bool EID_ID, typename Sink>
__device__ __forceinline__ void spmm_strip_staged(Sink&& sink, int lo_l, int n_l, int pos0,
                                                  const int* __restrict__ ids_w,
                                                  const int* __restrict__ eids_w, int* __restrict__ idbuf,
                                                  const float* __restrict__ w,
                                                  const float* __restrict__ X, int l) {
  constexpr int SB = StripCfg<L, NV>::SB;
  constexpr i64 F4 = (i64)L * NV;
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  IdStage<L, EID_ID ? 1 : 2> ids;
  ids.init(ids_w, eids_w, pos0, idbuf, l, m.total);
  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k_cur = -1;
  auto spill = [&]() {
    if (k_cur >= 0) {
      sink(k_cur, acc);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // weight pipeline: identity eid -> w[e] one batch ahead; otherwise the gather w[eid] one batch ahead
  struct Pre { int k, src; float w; };
  auto stage = [&](int jbase, Pre& p) {
    const int jj = jbase + l;
    const bool live = l < SB && jj < m.total;
    const int j = jj < m.total ? jj : m.total - 1;
    int e;
    m.locate<L>(j, p.k, e);
    p.src = ids.id(j);          // slots past the end re-read the last neighbour with weight 0
    p.w = 0.f;
    if (live) p.w = w[EID_ID ? e : ids.eid(j)];
  };
  Pre p1;
  stage(0, p1);
  for (int jb = 0; jb < m.total; jb += SB) {
    const int my_k = p1.k;
    const unsigned my_off = (unsigned)p1.src * (unsigned)(F4 * 16);
    const float my_w = p1.w;
    float4 x[SB][NV];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[u][v] = ld4_off(X, o + (unsigned)((v * L + l) * 16));
    });
    if (jb + SB < m.total) {   // next batch: its weight load stays in flight behind this batch's row requests
      ids.advance(jb + SB);
      stage(jb + SB, p1);
    }
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int kt = group_bcast<L, u>(my_k);
      if (kt != k_cur) {   // group-uniform
        spill();
        k_cur = kt;
      }
      const float w1 = group_bcast<L, u>(my_w);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc[v].x = fmaf(w1, x[u][v].x, acc[v].x);
        acc[v].y = fmaf(w1, x[u][v].y, acc[v].y);
        acc[v].z = fmaf(w1, x[u][v].z, acc[v].z);
        acc[v].w = fmaf(w1, x[u][v].w, acc[v].w);
      }
    });
  }
  spill();
}

// ---- WINDOW-SWEEP drivers (plan) -----------------------------------------------------------------
// vrow v = (a piece of) row vr_row[v]; inside window w it owns slots [wp_lo[w*V+v], wp_hi[w*V+v])
// whose neighbour ids lie in [w*win_cols, (w+1)*win_cols).  A row longer than T slots is cut into
// P pieces that each take 1/P of the row's slots in EVERY window.  Group g of round r owns vrows
// [(r*n_groups + g)*K, +K).  LDS holds the K rows (A rows / partial sums) of every group.
constexpr int kSweepBlocksPerCu = 4;   // most co-resident 256-thread workgroups per CU any sweep kernel
                                       // is compiled for (<= 128 VGPRs, 32 KB LDS each)
// Resident workgroups per CU a given instantiation is compiled for (its __launch_bounds__ and the
// grid the host launches): the one-head 16..512-float rows fit 128 VGPRs without spilling; rows of
// 1024 floats (NV = 4), per-head weights (H1 = false: a weight register per slot and float4) and
// the vrow-owner order (LDS partial sums + pacer state) get 168.
__host__ __device__ constexpr int sweep_bpc(int NV, bool H1, bool owner) {
  return (owner && H1 && NV < 4) ? 4 : 3;
}
struct SweepView {
  const int* wp_lo;   // [W * V]
  const int* wp_hi;   // [W * V]
  const int* vr_row;  // [V]
  const int* idx32;   // [E]
  const int* eid32;   // [E] or nullptr when eid is the identity
  int* sync;          // [rounds * W] arrival counters, zeroed before the launch (nullptr = free-running)
  int V, W, K, rounds;
  int drift;          // a workgroup may run at most `drift` windows ahead of the slowest one
  int xcd_slots;      // grid is a multiple of this; workgroup b serves XCD slot b % xcd_slots
  int vx;             // vrows per XCD slot (multiple of K)
  i64 win_bytes;      // bytes of gathered table per window
  i64 table_bytes;    // bytes of the gathered table
  int prefetch;       // 1: every workgroup touches a slice of the NEXT window at the start of a step
  int touch;          // bit 0: touch the granule's id lines at task start, bit 1: also its edge-id / weight lines
  // dealt (window-major) layout of the window-owner tasks, or nullptr (plan.hip, Sweep::Dealt)
  const int4* rec;    // [W * tiles * tile] (first slot, length, row id, position in ids_w) per granule
  const int* ids_w;   // neighbour ids in dealt order: a lane group's strip is one contiguous aligned run
  const int* eids_w;  // edge ids in the same order (nullptr when eid is the identity)
};

// Soft pacing between the workgroups of one sweep launch, per XCD (each XCD has its own L2, so
// only workgroups sharing an XCD need to walk the windows together).  Counters are sharded by the
// hardware XCC id: sync[0..7] = workgroups registered per XCD, then one 256-B line per
// (XCD, step).  One thread per workgroup signals / polls; the other waves wait at a barrier.
// It only keeps the gather window L2-resident (speed); results never depend on it: the spin is
// bounded and a workgroup that times out stops waiting for good, so a grid that is not fully
// co-resident cannot hang.
constexpr int kSyncStride = 64;   // ints between counters: one 256-B line each
constexpr int kSyncXcds = 8;
// Layout of SweepView::sync (ints, all zero at launch):
//   [xcc * kSyncStride]                                   workgroups registered on XCD xcc
//   [(kSyncXcds + (xcc*steps + s)*2 + 0) * kSyncStride]    arrivals at the end of step s
//   [(kSyncXcds + (xcc*steps + s)*2 + 1) * kSyncStride]    1 once every registered workgroup arrived
// Arrivals are returning agent-scope atomics (memory side, ~12 ns each per line); the workgroup
// whose add completes the count publishes the release word, which the others poll with relaxed
// agent loads served by their own XCD's L2 (writer and readers share that L2).
struct SweepPacer {
  int* ctr;        // this XCD's (arrivals, released) pairs
  int* reg;        // this XCD's registration counter
  int drift;
  bool active;     // per wave
  int* lds;        // [0..3] waves of this workgroup done with step (s & 3); [4] highest released step + 1
  static constexpr int kWaves = kFastBlock / kWave;
  __device__ __forceinline__ SweepPacer(const SweepView& s, int* lds_words)
      : ctr(nullptr), reg(nullptr), drift(s.drift), active(s.sync != nullptr && s.drift > 0),
        lds(lds_words) {
    if (!active) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= kSyncXcds - 1;
    const int steps = s.rounds * s.W;
    reg = s.sync + (i64)xcc * kSyncStride;
    ctr = s.sync + (i64)kSyncStride * (kSyncXcds + 2 * (i64)xcc * steps);
    if (threadIdx.x < 5) lds[threadIdx.x] = 0;
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  // Called by every lane of a wave after the wave finished `done_step` (-1 before step 0); returns
  // once the wave may start step done_step + 1.  No workgroup barrier: each wave signals through an
  // LDS counter (the last wave of the workgroup forwards the arrival to the XCD counter) and polls
  // the XCD's release word only when the LDS copy of "released up to" is not enough.
  __device__ __forceinline__ void step_done_and_wait(int done_step) {
    if (!active) return;
    int gave_up = 0;
    if ((threadIdx.x & 63) == 0) {
      if (done_step >= 0) {
        int* slot = lds + (done_step & 3);
        const int old = __hip_atomic_fetch_add(slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (old == kWaves - 1) {            // last wave of this workgroup for done_step
          __hip_atomic_store(slot, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          int* c = ctr + (i64)done_step * 2 * kSyncStride;
          const int prev = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int n = __hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (prev + 1 >= n)
            __hip_atomic_store(c + kSyncStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const int need = done_step + 1 - drift;   // step that every workgroup of the XCD must have finished
      if (need >= 0 &&
          __hip_atomic_load(lds + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= need) {
        const int* rel = ctr + ((i64)need * 2 + 1) * kSyncStride;
        int it = 0;
        while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          __builtin_amdgcn_s_sleep(16);
          if (++it > 1500) { gave_up = 1; break; }   // ~2 ms without progress: give up pacing for good
        }
        if (!gave_up)
          __hip_atomic_fetch_max(lds + 4, need + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    if (__shfl(gave_up, 0)) active = false;
  }
};

// Which vrows a lane group owns.  Workgroups b, b+8, ... share an XCD under the observed
// round-robin placement (speed only): XCD slot x = b % 8 owns the CONTIGUOUS vrow range
// [x*vx, (x+1)*vx), so neighbouring rows -- whose per-slot scalars share cache lines in the
// transposed passes -- are walked by workgroups behind the same L2 at the same time.
struct SweepOwner {
  i64 base, end, stride_groups, group;
  __device__ __forceinline__ SweepOwner(const SweepView& s, int gpb, int g_in_blk) {
    const int slots = s.xcd_slots;                       // 8, or 1 for tiny grids
    const i64 x = blockIdx.x % slots, lb = blockIdx.x / slots;
    const i64 blocks_per_slot = gridDim.x / slots;
    base = x * (i64)s.vx;
    end = base + s.vx < s.V ? base + s.vx : s.V;
    stride_groups = blocks_per_slot * gpb;
    group = lb * gpb + g_in_blk;
  }
  __device__ __forceinline__ i64 first_vrow(int r, int K) const {
    return base + ((i64)r * stride_groups + group) * K;
  }
  __device__ __forceinline__ int count(i64 v0, int K) const {
    return v0 >= end ? 0 : ((end - v0) < K ? (int)(end - v0) : K);
  }
};

// Pull the next window into this XCD's L2 while the current one is being gathered from: the
// workgroups of an XCD slot (b % 8) together touch one 128-B line per thread.  The value is only
// "used" by an empty asm so the load is kept but nothing depends on it (speed only).
__device__ __forceinline__ int sweep_prefetch(const SweepView& s, const float* table, int next_w) {
  int v = 0;
  if (s.prefetch && next_w < s.W) {
    const i64 lb = blockIdx.x / s.xcd_slots;
    const i64 off = (lb * kFastBlock + threadIdx.x) * 128;
    const i64 base = (i64)next_w * s.win_bytes;
    if (off < s.win_bytes && base + off < s.table_bytes)
      v = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(table) + base + off);
  }
  return v;
}
__device__ __forceinline__ void sweep_prefetch_retire(int v) { asm volatile("; prefetched %0" ::"v"(v)); }

template <int L, int NV, bool H1, bool EID_ID, bool OFF32>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1, false)) void k_sddmm_sweep_f32(
    SweepView s, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y,
    int h, int d4) {
  extern __shared__ float4 lds[];
  constexpr int GPB = GroupCfg<L>::kGroupsPerBlock;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;  // [K][NV][L]
  __shared__ int pace_words[8];
  SweepPacer pacer(s, pace_words);
  const SweepOwner own(s, GPB, g_in_blk);
  for (int r = 0; r < s.rounds; ++r) {
    const i64 v0 = own.first_vrow(r, s.K);
    const int nv = own.count(v0, s.K);
    for (int k = 0; k < nv; ++k) {
      const i64 row = s.vr_row[v0 + k];
#pragma unroll
      for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = ld4(A, row * F4 + v * L + l);
    }
    int lo_n = 0, hi_n = 0;   // bounds of the NEXT window's granules, fetched one step ahead
    if (l < nv) { lo_n = s.wp_lo[v0 + l]; hi_n = s.wp_hi[v0 + l]; }
    for (int w = 0; w < s.W; ++w) {
      pacer.step_done_and_wait(r * s.W + w - 1);
      const int lo_l = lo_n, hi_l = hi_n;
      if (l < nv && w + 1 < s.W) {
        lo_n = s.wp_lo[(i64)(w + 1) * s.V + v0 + l];
        hi_n = s.wp_hi[(i64)(w + 1) * s.V + v0 + l];
      }
      const int pf = sweep_prefetch(s, B, w + 1);
      sddmm_strip<L, NV, H1, EID_ID, OFF32>(mine, lo_l, hi_l - lo_l, s.eid32, s.idx32, B, y, h, d4, l);
      sweep_prefetch_retire(pf);
    }
  }
}

template <int L, int NV, bool H1, bool EID_ID, bool OFF32>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1, false)) void k_spmm_sweep_f32(
    SweepView s, const float* __restrict__ wgt, const float* __restrict__ X,
    float* __restrict__ out, int h, int d4) {
  extern __shared__ float4 lds[];
  constexpr int GPB = GroupCfg<L>::kGroupsPerBlock;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;
  int hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = H1 ? 0 : (v * L + l) / d4;
  __shared__ int pace_words[8];
  SweepPacer pacer(s, pace_words);
  const SweepOwner own(s, GPB, g_in_blk);
  for (int r = 0; r < s.rounds; ++r) {
    const i64 v0 = own.first_vrow(r, s.K);
    const int nv = own.count(v0, s.K);
    for (int k = 0; k < nv; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt_l = 0;        // lane k: slots vrow k received over all windows
    int lo_n = 0, hi_n = 0;
    if (l < nv) { lo_n = s.wp_lo[v0 + l]; hi_n = s.wp_hi[v0 + l]; }
    for (int w = 0; w < s.W; ++w) {
      pacer.step_done_and_wait(r * s.W + w - 1);
      const int lo_l = lo_n, hi_l = hi_n;
      if (l < nv && w + 1 < s.W) {
        lo_n = s.wp_lo[(i64)(w + 1) * s.V + v0 + l];
        hi_n = s.wp_hi[(i64)(w + 1) * s.V + v0 + l];
      }
      cnt_l += hi_l - lo_l;
      const int pf = sweep_prefetch(s, X, w + 1);
      auto to_lds = [&](int k, const float4 (&acc)[NV]) {   // partial sums of the K vrows live in LDS
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          float4 o = mine[(k * NV + v) * L + l];
          o.x += acc[v].x; o.y += acc[v].y; o.z += acc[v].z; o.w += acc[v].w;
          mine[(k * NV + v) * L + l] = o;
        }
      };
      spmm_strip<L, NV, H1, EID_ID, OFF32>(to_lds, lo_l, hi_l - lo_l, s.eid32, s.idx32, wgt, X, h, hv, l);
      sweep_prefetch_retire(pf);
    }
    // pieces of one (long) row may live in several groups: merge with float atomics
    for (int k = 0; k < nv; ++k) {
      if (__shfl(cnt_l, k, L) == 0) continue;
      float4 acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = mine[(k * NV + v) * L + l];
      atomic_flush<L, NV>(out, s.vr_row[v0 + k], acc, l);
    }
  }
}

// ---- WINDOW-OWNER drivers (plan) -----------------------------------------------------------------
// The loop interchange of the sweep above: instead of workgroups owning vrows and all of them
// walking the windows together (every XCD's L2 sees every window once per round, kept in step by
// the pacer), every XCD owns the windows w = x, x+8, ... and its waves pull (window, vrow-tile)
// tasks from that XCD's queue, window-major.  A window is then brought into exactly one L2, once,
// and stays there for as long as that XCD works on it; nothing has to be paced.  A task is one
// wave = 64/L lane groups x K consecutive vrows in one window.  The price: the rows' own operand
// (SDDMM: A rows) is re-read and the partial sums (SpMM) are flushed once per (vrow, window)
// instead of once per vrow.  An XCD whose queue is empty steals from the other queues (those tasks
// gather through the Infinity Cache; it only matters for the tail).
// Queue heads: SweepView::sync[y * kSyncStride], y < 8, zero at launch.
struct WownQueue {
  int* q;
  int ntasks, W, x, s;
  __device__ __forceinline__ WownQueue(const SweepView& sv, int tiles) : q(sv.sync), ntasks(tiles), W(sv.W), s(0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    x = (int)(xcc & (kSyncXcds - 1));
  }
  // Split form of pull(): issue() starts the dequeue on the current queue (the returned word is
  // not waited for), resolve() -- a whole task later -- decodes it, falling back to the blocking
  // pull() when that queue turned out to be drained.
  __device__ __forceinline__ int issue() {
    int raw = -1;
    if (s < kSyncXcds && (threadIdx.x & (kWave - 1)) == 0)
      raw = __hip_atomic_fetch_add(q + (i64)((x + s) & (kSyncXcds - 1)) * kSyncStride, 1, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    return raw;
  }
  __device__ __forceinline__ bool resolve(int raw, int& w, int& t) {
    if (s >= kSyncXcds) return false;
    const int y = (x + s) & (kSyncXcds - 1);
    const int nwin = (W - y + kSyncXcds - 1) / kSyncXcds;
    const int idx = __shfl(raw, 0);
    if (idx >= 0 && idx < nwin * ntasks) {
      w = y + kSyncXcds * (idx / ntasks);
      t = idx % ntasks;
      return true;
    }
    ++s;
    return pull(w, t);
  }
  // wave-uniform; every lane calls.  Returns false when all eight queues are drained.
  __device__ __forceinline__ bool pull(int& w, int& t) {
    while (s < kSyncXcds) {
      const int y = (x + s) & (kSyncXcds - 1);
      const int nwin = (W - y + kSyncXcds - 1) / kSyncXcds;   // windows y, y+8, ... < W
      if (nwin > 0) {
        int idx = 0;
        if ((threadIdx.x & (kWave - 1)) == 0)
          idx = __hip_atomic_fetch_add(q + (i64)y * kSyncStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __shfl(idx, 0);
        if (idx < nwin * ntasks) {
          w = y + kSyncXcds * (idx / ntasks);
          t = idx % ntasks;
          return true;
        }
      }
      ++s;
    }
    return false;
  }
};

// Bounds of one (window, vrow tile) task.  The tile's GW*K vrows are DEALT to the wave's GW lane
// groups by granule length: the wave ranks the granules of this window (longest first) and hands
// them out in snake order, so the groups -- which run in lock step -- get nearly equal slot
// counts.  Without it a wave spends 12-22 % more batch steps than its groups need on average
// (tools/divergence_model.py).  Afterwards lane k < K of a group holds the slot range and row id
// of that group's k-th vrow (empty granules have hi == lo).
template <int L>
struct WownTask {
  int lo, hi, row, nv;
  int pos;   // dealt layouts only: position of this lane's granule in ids_w
  // Plan-time deal: lane (g, k) reads its granule's record; nothing to rank at run time.
  __device__ __forceinline__ void load_dealt(const SweepView& s, int w, int t, int tile, int tiles) {
    const int lane = threadIdx.x & (kWave - 1);
    const int g = lane / L, k = lane % L;
    int4 r = make_int4(0, 0, 0, 0);
    if (k < s.K) r = s.rec[((i64)w * tiles + t) * tile + g * s.K + k];
    lo = r.x; hi = r.x + r.y; row = r.z; pos = r.w;
    nv = s.K;
  }
  __device__ __forceinline__ void load(const SweepView& s, int w, int t, int tile) {
    constexpr int GW = kWave / L;
    const int lane = threadIdx.x & (kWave - 1);
    const i64 v = (i64)t * tile + lane;
    int lo_s = 0, hi_s = 0, row_s = 0;
    if (lane < tile && v < s.V) {
      lo_s = s.wp_lo[(i64)w * s.V + v];
      hi_s = s.wp_hi[(i64)w * s.V + v];
      row_s = s.vr_row[v];
    }
    nv = s.K;
    if constexpr (GW == 1) {          // one group per wave: nothing to balance
      lo = lo_s; hi = hi_s; row = row_s;
      return;
    }
    const int len = hi_s - lo_s;
    int rank = 0;                     // position of this lane's granule, longest first (ties by lane)
    for (int j = 0; j < tile; ++j) {
      const int lj = __shfl(len, j);
      rank += (lj > len || (lj == len && j < lane)) ? 1 : 0;
    }
    if (lane >= tile) rank = lane;    // bystanders map to themselves: the scatter stays a bijection
    const int inv = __builtin_amdgcn_ds_permute(rank << 2, lane);   // inv[r] = lane holding rank r
    const int g = lane / L, k = lane % L;
    const int r = k * GW + ((k & 1) ? GW - 1 - g : g);               // snake deal
    const int src = __shfl(inv, r < tile ? r : 0);
    const int lo_d = __shfl(lo_s, src), hi_d = __shfl(hi_s, src), row_d = __shfl(row_s, src);
    const bool mine = k < s.K;
    lo = mine ? lo_d : 0; hi = mine ? hi_d : 0; row = mine ? row_d : 0;
  }
};

// Task pipeline of both kernels: the id of task i+2 is being dequeued and the bounds of task i+1
// are being fetched while task i runs, so a task starts with its bounds in registers.
template <int L, int NV, bool H1, bool EID_ID, bool OFF32>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1, true)) void k_sddmm_wown_f32(
    SweepView s, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y,
    int h, int d4) {
  extern __shared__ float4 lds[];
  constexpr i64 F4 = (i64)L * NV;
  constexpr int GW = kWave / L;                // lane groups per wave
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;  // [K][NV][L]
  const int tile = GW * s.K;
  WownQueue queue(s, (s.V + tile - 1) / tile);
  int w, t;
  bool more = queue.pull(w, t);
  int raw = more ? queue.issue() : -1;
  WownTask<L> cur, nxt;
  if (more) cur.load(s, w, t, tile);
  while (more) {
    int wn = 0, tn = 0;
    const bool more_n = queue.resolve(raw, wn, tn);
    raw = more_n ? queue.issue() : -1;
    nxt.nv = 0; nxt.lo = nxt.hi = nxt.row = 0;
    if (more_n) nxt.load(s, wn, tn, tile);
    auto stage_rows = [&]() {   // A rows of this task's non-empty granules -> LDS
      for (int k = 0; k < cur.nv; ++k) {
        const i64 row = __shfl(cur.row, k, L);
        if (__shfl(cur.hi - cur.lo, k, L) == 0) continue;   // group-uniform
#pragma unroll
        for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = ld4(A, row * F4 + v * L + l);
      }
    };
    sddmm_strip<L, NV, H1, EID_ID, OFF32>(mine, cur.lo, cur.hi - cur.lo, s.eid32, s.idx32, B, y, h, d4, l,
                                          stage_rows, s.touch);
    cur = nxt;
    more = more_n;
  }
}

// Staged form (h == 1, identity eid, table < 4 GiB, dealt layout in the view): tasks come with their
// granules already dealt, ids through the group's LDS buffer (behind the A rows in dynamic LDS).
// D4 > 0: L / D4 heads of D4 float4s each (NV == 1).
template <int L, int NV, int D4 = 0>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, D4 == 0, true)) void k_sddmm_wown_staged_f32(
    SweepView s, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y) {
  extern __shared__ float4 lds[];
  constexpr i64 F4 = (i64)L * NV;
  constexpr int GW = kWave / L;
  constexpr int GPB = kFastBlock / L;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  float4* mine = lds + (i64)g_in_blk * s.K * F4;  // [K][NV][L]
  int* idbuf = reinterpret_cast<int*>(lds + (i64)GPB * s.K * F4) + g_in_blk * StageCfg<L, 1>::kLdsIntsPerGroup;
  const int tile = GW * s.K;
  const int tiles = (s.V + tile - 1) / tile;
  WownQueue queue(s, tiles);
  int w, t;
  bool more = queue.pull(w, t);
  int raw = more ? queue.issue() : -1;
  WownTask<L> cur, nxt;
  if (more) cur.load_dealt(s, w, t, tile, tiles);
  while (more) {
    int wn = 0, tn = 0;
    const bool more_n = queue.resolve(raw, wn, tn);
    raw = more_n ? queue.issue() : -1;
    nxt.nv = 0; nxt.lo = nxt.hi = nxt.row = nxt.pos = 0;
    if (more_n) nxt.load_dealt(s, wn, tn, tile, tiles);
    auto stage_rows = [&]() {   // A rows of this task's non-empty granules -> LDS
      for (int k = 0; k < cur.nv; ++k) {
        const i64 row = __shfl(cur.row, k, L);
        if (__shfl(cur.hi - cur.lo, k, L) == 0) continue;   // group-uniform
#pragma unroll
        for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = ld4(A, row * F4 + v * L + l);
      }
    };
    if constexpr (D4 == 0)
      sddmm_strip_staged<L, NV>(mine, cur.lo, cur.hi - cur.lo, __shfl(cur.pos, 0, L), s.ids_w, idbuf, B, y, l,
                                stage_rows);
    else
      sddmm_strip_staged_heads<L, D4>(mine, cur.lo, cur.hi - cur.lo, __shfl(cur.pos, 0, L), s.ids_w, idbuf,
                                      reinterpret_cast<float*>(reinterpret_cast<int*>(lds + (i64)GPB * s.K * F4) +
                                                               GPB * StageCfg<L, 1>::kLdsIntsPerGroup) + g_in_blk * 16 * (L / D4),
                                      B, y, l, stage_rows);
    cur = nxt;
    more = more_n;
  }
}

template <int L, int NV, bool H1, bool EID_ID, bool OFF32>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1, true)) void k_spmm_wown_f32(
    SweepView s, const float* __restrict__ wgt, const float* __restrict__ X,
    float* __restrict__ out, int h, int d4) {
  constexpr int GW = kWave / L;
  const int l = threadIdx.x % L;
  int hv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) hv[v] = H1 ? 0 : (v * L + l) / d4;
  const int tile = GW * s.K;
  WownQueue queue(s, (s.V + tile - 1) / tile);
  int w, t;
  bool more = queue.pull(w, t);
  int raw = more ? queue.issue() : -1;
  WownTask<L> cur, nxt;
  if (more) cur.load(s, w, t, tile);
  while (more) {
    int wn = 0, tn = 0;
    const bool more_n = queue.resolve(raw, wn, tn);
    raw = more_n ? queue.issue() : -1;
    nxt.nv = 0; nxt.lo = nxt.hi = nxt.row = 0;
    if (more_n) nxt.load(s, wn, tn, tile);
    // a granule's sum goes straight to the output row: one dense atomic flush per (vrow, window)
    const int row_l = cur.row;
    auto to_out = [&](int k, const float4 (&acc)[NV]) {
      atomic_flush_dense<L, NV>(out, __shfl(row_l, k, L), acc, l);
    };
    spmm_strip<L, NV, H1, EID_ID, OFF32>(to_out, cur.lo, cur.hi - cur.lo, s.eid32, s.idx32, wgt, X, h, hv, l);
    cur = nxt;
    more = more_n;
  }
}

// Staged form (h == 1, table < 4 GiB, dealt layout in the view).
// 64-lane groups (d >= 256) are compiled for 3 resident workgroups per CU (the launch default): the
// staging registers do not fit the 128 VGPRs that 4 per CU leave.
template <int L, int NV, bool EID_ID>
__global__ __launch_bounds__(kFastBlock, L == 64 ? 3 : sweep_bpc(NV, true, true)) void k_spmm_wown_staged_f32(
    SweepView s, const float* __restrict__ wgt, const float* __restrict__ X, float* __restrict__ out) {
  extern __shared__ float4 lds[];
  constexpr int GW = kWave / L;
  const int l = threadIdx.x % L;
  int* idbuf = reinterpret_cast<int*>(lds) + (threadIdx.x / L) * StageCfg<L, EID_ID ? 1 : 2>::kLdsIntsPerGroup;
  const int tile = GW * s.K;
  const int tiles = (s.V + tile - 1) / tile;
  WownQueue queue(s, tiles);
  int w, t;
  bool more = queue.pull(w, t);
  int raw = more ? queue.issue() : -1;
  WownTask<L> cur, nxt;
  if (more) cur.load_dealt(s, w, t, tile, tiles);
  while (more) {
    int wn = 0, tn = 0;
    const bool more_n = queue.resolve(raw, wn, tn);
    raw = more_n ? queue.issue() : -1;
    nxt.nv = 0; nxt.lo = nxt.hi = nxt.row = nxt.pos = 0;
    if (more_n) nxt.load_dealt(s, wn, tn, tile, tiles);
    const int row_l = cur.row;
    auto to_out = [&](int k, const float4 (&acc)[NV]) {
      atomic_flush_dense<L, NV>(out, __shfl(row_l, k, L), acc, l);
    };
    spmm_strip_staged<L, NV, EID_ID>(to_out, cur.lo, cur.hi - cur.lo, __shfl(cur.pos, 0, L), s.ids_w, s.eids_w,
                                     idbuf, wgt, X, l);
    cur = nxt;
    more = more_n;
  }
}

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

// Transpose per-slot scalars between the two CSR orientations: out[inv[e]] = in[e] for every
// slot e of the ROW-major sweep (in is read in slot order = sequentially inside a granule; the
// writes of one (XCD vrow range, column window) step land in a few MB of the column-major array
// and are combined in that XCD's L2 before they leave).  The column-major pass then reads its
// weights sequentially instead of gathering 4 bytes per slot.
template <int L>   // (a template so that the header can be included by several translation units)
__global__ __launch_bounds__(kFastBlock) void k_scatter_scalars_sweep(
    SweepView s, const int* __restrict__ inv, const float* __restrict__ in, float* __restrict__ out) {
  constexpr int GPB = kFastBlock / L;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  __shared__ int pace_words[8];
  SweepPacer pacer(s, pace_words);
  const SweepOwner own(s, GPB, g_in_blk);
  for (int r = 0; r < s.rounds; ++r) {
    const i64 v0 = own.first_vrow(r, s.K);
    const int nv = own.count(v0, s.K);
    for (int w = 0; w < s.W; ++w) {
      pacer.step_done_and_wait(r * s.W + w - 1);
      int lo_l = 0, n_l = 0;
      if (l < nv) {
        lo_l = s.wp_lo[(i64)w * s.V + v0 + l];
        n_l = s.wp_hi[(i64)w * s.V + v0 + l] - lo_l;
      }
      StripMap m;
      m.init<L>(lo_l, n_l, l);
      constexpr int UB = 8;   // batches in flight per group: the loads are HBM-latency bound
      for (int jb = 0; jb < m.total; jb += UB * L) {
        float v[UB];
        int pos[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int j = jb + u * L + l;
          int k, e;
          m.locate<L>(j < m.total ? j : m.total - 1, k, e);
          pos[u] = -1;
          v[u] = 0.f;
          if (j < m.total) {
            v[u] = __builtin_nontemporal_load(in + e);
            pos[u] = __builtin_nontemporal_load(inv + e);
          }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (pos[u] >= 0) out[pos[u]] = v[u];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Row-segment softmax (plan.row_owned).  Segment s = chunks [seg_chunk[s], seg_chunk[s+1]) =
// slots [indptr[c0], indptr[c1]); all of one row.  A group of G lanes owns a segment; items are
// the flattened (slot, head) pairs so that for eid == identity the reads are fully coalesced.
// Requires G % h == 0 (then a lane always sees the same head t = lane % h).
// Semantics: graphop_kernel.cu:170-202 (m starts at -1e9, :428).
// items per lane kept in registers (rows up to G*R items are read once).  Measured on Reddit-shape
// (mean row 492, 23 % of the rows above 512): forward best at 16, backward at 32.
// First slot of segment s: from the plan's per-segment array when it has one (one dependent load less in front of
// every row: short rows are bound by that chain), else through the chunk arrays.
__device__ __forceinline__ i64 seg_first(const i64* __restrict__ seg_eptr, const i64* __restrict__ seg_chunk,
                                         const i64* __restrict__ indptr, i64 s) {
  return seg_eptr ? seg_eptr[s] : indptr[seg_chunk[s]];
}

constexpr int kSoftmaxCacheFwd = 16;
constexpr int kSoftmaxCacheBwd = 32;

template <typename T>
__device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

// Segments longer than `long_len` slots are left to k_softmax_*_long (one workgroup per row).
template <typename T, int G, bool EID_ID>
__device__ __forceinline__ void softmax_fwd_seg_body(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ x, T* __restrict__ y, i64 n_seg, int h,
    i64 long_len, i64 block, const i64* __restrict__ row, T* __restrict__ stats) {
  constexpr int R = kSoftmaxCacheFwd;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const i64 items = len * h;
  const int t = l % h;

  auto offs = [&](i64 q) -> i64 {   // identity eid: (e0 + q/h)*h + q%h == e0*h + q
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  // whole row in registers: x read once, one exp per item.  Tiers by row length: the unrolled loops run all RR
  // iterations whatever the row holds
  auto in_regs = [&](auto rc) {
    constexpr int RR = decltype(rc)::value;
    T v[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const i64 q = l + (i64)r * G;
      v[r] = q < items ? x[offs(q)] : neg_inf<T>();
    }
    T m = (T)-1e9;
#pragma unroll
    for (int r = 0; r < RR; ++r) m = v[r] > m ? v[r] : m;
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) {
        const T m2 = __shfl_xor(m, mask, G);
        m = m > m2 ? m : m2;
      }
    T sum = 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      v[r] = (l + (i64)r * G) < items ? exp_le0(v[r] - m) : (T)0;
      sum += v[r];
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) sum += __shfl_xor(sum, mask, G);
    const T inv = (T)1 / sum;                     // one division per row; the items are scaled
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const i64 q = l + (i64)r * G;
      if (q < items) y[offs(q)] = v[r] * inv;
    }
    if (stats && l < h) {   // row statistics for the fused attention backward: (max, 1 / sum)
      const i64 o = (row[seg_chunk[s]] * h + l) * 2;
      stats[o] = m; stats[o + 1] = inv;
    }
  };
  if (items <= (i64)G * (R / 4)) { in_regs(std::integral_constant<int, R / 4>{}); return; }
  if (items <= (i64)G * (R / 2)) { in_regs(std::integral_constant<int, R / 2>{}); return; }
  if (items <= (i64)G * R) { in_regs(std::integral_constant<int, R>{}); return; }

  T m = (T)-1e9, sum = 0;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const T v = x[(EID_ID ? k : eid[k]) * h + t];
    if (v > m) {
      sum = sum * exp_le0(m - v) + (T)1;
      m = v;
    } else {
      sum += exp_le0(v - m);
    }
  }
#pragma unroll
  for (int mask = G / 2; mask >= 1; mask >>= 1) {
    if (mask >= h) {  // wave-uniform
      const T m2 = __shfl_xor(m, mask, G);
      const T s2 = __shfl_xor(sum, mask, G);
      const T mn = m > m2 ? m : m2;
      sum = sum * exp_le0(m - mn) + s2 * exp_le0(m2 - mn);
      m = mn;
    }
  }
  const T inv = (T)1 / sum;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    y[o] = exp_le0(x[o] - m) * inv;
  }
  if (stats && l < h) {
    const i64 o = (row[seg_chunk[s]] * h + l) * 2;
    stats[o] = m; stats[o + 1] = inv;
  }
}

// Backward: g = sum dy*y over the row; dx = dy*y - g*y   (graphop_kernel.cu:208-230)
template <typename T, int G, bool EID_ID>
__device__ __forceinline__ void softmax_bwd_seg_body(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ y, const T* __restrict__ dy,
    T* __restrict__ dx, i64 n_seg, int h, i64 long_len, i64 block) {
  // gathered through eid every cached item carries its own 64-bit offset: 8 per lane fit the
  // register file, 32 spilled (the identity form walks one base pointer with immediate offsets)
  constexpr int R = EID_ID ? kSoftmaxCacheBwd : 8;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const i64 items = len * h;
  const int t = l % h;

  auto offs = [&](i64 q) -> i64 {
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  auto in_regs = [&](auto rc) {
    constexpr int RR = decltype(rc)::value;
    T yy[RR], dd[RR];
    T g = 0;
    const int n_it = (int)items;
    if constexpr (EID_ID) {   // one base address + immediate offsets r*G
      const T* yp = y + e0 * h + l;
      const T* dp = dy + e0 * h + l;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const bool ok = l + r * G < n_it;
        yy[r] = ok ? yp[r * G] : (T)0;
        dd[r] = ok ? dp[r * G] : (T)0;
      }
    } else {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const i64 q = l + (i64)r * G;
        yy[r] = 0; dd[r] = 0;
        if (q < items) {
          const i64 o = offs(q);
          yy[r] = y[o];
          dd[r] = dy[o];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) g += dd[r] * yy[r];
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) g += __shfl_xor(g, mask, G);
    if constexpr (EID_ID) {
      T* xp = dx + e0 * h + l;
#pragma unroll
      for (int r = 0; r < RR; ++r)
        if (l + r * G < n_it) xp[r * G] = dd[r] * yy[r] - g * yy[r];
    } else {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const i64 q = l + (i64)r * G;
        if (q < items) dx[offs(q)] = dd[r] * yy[r] - g * yy[r];
      }
    }
  };
  if (items <= (i64)G * (R / 4)) { in_regs(std::integral_constant<int, R / 4>{}); return; }
  if (items <= (i64)G * (R / 2)) { in_regs(std::integral_constant<int, R / 2>{}); return; }
  if (items <= (i64)G * R) { in_regs(std::integral_constant<int, R>{}); return; }

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

// Long rows: one 256-thread workgroup per row segment listed in long_segs[] (rows above
// kLongSegment slots).  Up to 256*kBlockCache items are held in registers (one read of the inputs,
// one exp per item); longer rows loop twice.  Per-head partials are merged through LDS.
// Requires 256 % h == 0.
constexpr int kBlockCache = 8;

template <typename T, bool BWD>
__device__ __forceinline__ void block_merge(T& m, T& sum, T* sh_m, T* sh_s, int h) {
  const int tid = threadIdx.x;
  __syncthreads();                       // previous users of sh_m / sh_s are done reading
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= h; stride >>= 1) {   // tid and tid+stride share a head
    if (tid < stride) {
      if constexpr (!BWD) {
        const T m1 = sh_m[tid], m2 = sh_m[tid + stride];
        const T mn = m1 > m2 ? m1 : m2;
        sh_s[tid] = sh_s[tid] * exp_le0(m1 - mn) + sh_s[tid + stride] * exp_le0(m2 - mn);
        sh_m[tid] = mn;
      } else {
        sh_s[tid] += sh_s[tid + stride];
      }
    }
    __syncthreads();
  }
  m = sh_m[tid % h]; sum = sh_s[tid % h];
}

template <typename T, bool BWD, bool EID_ID>
__device__ __forceinline__ void softmax_long_body(
    const int* __restrict__ long_segs, const i64* __restrict__ seg_chunk,
    const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const i64* __restrict__ eid, const T* __restrict__ in0,
    const T* __restrict__ in1, T* __restrict__ out, int h, T* sh_m, T* sh_s, i64 long_len,
    const i64* __restrict__ row = nullptr, T* __restrict__ stats = nullptr) {
  constexpr int RB = kBlockCache;
  const i64 s = long_segs[blockIdx.x];
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  if (seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0 <= long_len) return;   // block-uniform: the per-row groups take it
  const i64 items = (seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0) * h;
  const int tid = threadIdx.x, t = tid % h;
  auto offs = [&](i64 q) -> i64 {   // 256 % h == 0, so q % h == t for every q of this thread
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  if (items <= (i64)kFastBlock * RB) {
    T v[RB], u[BWD ? RB : 1];
    const int n_it = (int)items;
    // identity eid: one base address per array + immediate offsets r*256 (few address registers)
    const T* p0 = in0 + e0 * h + tid;
    const T* p1 = BWD ? in1 + e0 * h + tid : nullptr;
    T* po = out + e0 * h + tid;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int q = tid + r * kFastBlock;
      v[r] = BWD ? (T)0 : neg_inf<T>();
      if constexpr (BWD) u[r] = 0;
      if (q < n_it) {
        if constexpr (EID_ID) {
          v[r] = p0[r * kFastBlock];
          if constexpr (BWD) u[r] = p1[r * kFastBlock];
        } else {
          const i64 o = offs(q);
          v[r] = in0[o];
          if constexpr (BWD) u[r] = in1[o];
        }
      }
    }
    T m = (T)-1e9, sum = 0;
    if constexpr (!BWD) {
#pragma unroll
      for (int r = 0; r < RB; ++r) m = v[r] > m ? v[r] : m;
      // block max per head first, so every thread exponentiates against the final maximum
      __syncthreads();
      sh_m[tid] = m;
      __syncthreads();
      for (int stride = kFastBlock / 2; stride >= h; stride >>= 1) {
        if (tid < stride) { const T a = sh_m[tid], b2 = sh_m[tid + stride]; sh_m[tid] = a > b2 ? a : b2; }
        __syncthreads();
      }
      m = sh_m[t];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        v[r] = (tid + r * kFastBlock) < n_it ? exp_le0(v[r] - m) : (T)0;
        sum += v[r];
      }
      T mm = 0;
      block_merge<T, true>(mm, sum, sh_m, sh_s, h);
      const T inv = (T)1 / sum;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int q = tid + r * kFastBlock;
        if (q < n_it) {
          if constexpr (EID_ID) po[r * kFastBlock] = v[r] * inv;
          else out[offs(q)] = v[r] * inv;
        }
      }
      if (stats && tid < h) {
        const i64 o = (row[seg_chunk[s]] * h + tid) * 2;
        stats[o] = m; stats[o + 1] = inv;
      }
    } else {
#pragma unroll
      for (int r = 0; r < RB; ++r) sum += u[r] * v[r];
      block_merge<T, true>(m, sum, sh_m, sh_s, h);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int q = tid + r * kFastBlock;
        if (q < n_it) {
          if constexpr (EID_ID) po[r * kFastBlock] = u[r] * v[r] - sum * v[r];
          else out[offs(q)] = u[r] * v[r] - sum * v[r];
        }
      }
    }
    return;
  }
  T m = (T)-1e9, sum = 0;
  for (i64 q = tid; q < items; q += kFastBlock) {
    const i64 o = offs(q);
    if constexpr (!BWD) {
      const T v = in0[o];
      if (v > m) { sum = sum * exp_le0(m - v) + (T)1; m = v; }
      else sum += exp_le0(v - m);
    } else {
      sum += in1[o] * in0[o];
    }
  }
  block_merge<T, BWD>(m, sum, sh_m, sh_s, h);
  const T inv = BWD ? sum : (T)1 / sum;
  for (i64 q = tid; q < items; q += kFastBlock) {
    const i64 o = offs(q);
    if constexpr (!BWD) out[o] = exp_le0(in0[o] - m) * inv;
    else { const T yy = in0[o]; out[o] = in1[o] * yy - sum * yy; }
  }
  if constexpr (!BWD) {
    if (stats && tid < h) {
      const i64 o = (row[seg_chunk[s]] * h + tid) * 2;
      stats[o] = m; stats[o + 1] = inv;
    }
  }
}

// One launch: workgroups [0, n_long) take the hub rows (dispatched first, so their long serial
// loops overlap the bulk), the rest take kFastBlock/G ordinary row segments each.
template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_fwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ x, T* __restrict__ y, i64 n_seg, int h,
    i64 long_len, const int* __restrict__ long_segs, int n_long, const i64* __restrict__ row,
    T* __restrict__ stats) {
  __shared__ T sh_m[kFastBlock];
  __shared__ T sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_long_body<T, false, EID_ID>(long_segs, seg_chunk, indptr, seg_eptr, eid, x, (const T*)nullptr, y, h,
                                        sh_m, sh_s, long_len, row, stats);
  else
    softmax_fwd_seg_body<T, G, EID_ID>(seg_chunk, indptr, seg_eptr, eid, x, y, n_seg, h, long_len,
                                       (i64)blockIdx.x - n_long, row, stats);
}

template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_bwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ y, const T* __restrict__ dy,
    T* __restrict__ dx, i64 n_seg, int h, i64 long_len, const int* __restrict__ long_segs,
    int n_long) {
  __shared__ T sh_m[kFastBlock];
  __shared__ T sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_long_body<T, true, EID_ID>(long_segs, seg_chunk, indptr, seg_eptr, eid, y, dy, dx, h, sh_m, sh_s, long_len);
  else
    softmax_bwd_seg_body<T, G, EID_ID>(seg_chunk, indptr, seg_eptr, eid, y, dy, dx, n_seg, h, long_len,
                                       (i64)blockIdx.x - n_long);
}

// -------------------------------------------------------------------------------------------------
// Several heads, identity eid, h % 4 == 0, fp32: the (slot, head) items of a row are len * h contiguous
// floats, read and written as float4s.  Component j of a lane's float4 belongs to head (4 * lane + j) % h
// for every float4 the lane touches (4 * G % h == 0), so a lane keeps four running statistics and lanes
// h / 4 apart are merged.  (The scalar form above reads a row of 492 slots x 8 heads twice with 4-byte
// loads in a latency-bound loop: 2.7 TB/s; this one holds rows up to G * 32 float4s in registers.)
__device__ __forceinline__ float4 f4_splat(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 f4_max(float4 a, float4 b) {
  return make_float4(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z, a.w > b.w ? a.w : b.w);
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_exp_sub(float4 a, float4 b) {
  return make_float4(exp_nonpos(a.x - b.x), exp_nonpos(a.y - b.y), exp_nonpos(a.z - b.z), exp_nonpos(a.w - b.w));
}
__device__ __forceinline__ float4 f4_rcp(float4 a) { return make_float4(1.f / a.x, 1.f / a.y, 1.f / a.z, 1.f / a.w); }
template <int G>
__device__ __forceinline__ float4 f4_shfl_xor(float4 a, int mask) {
  return make_float4(__shfl_xor(a.x, mask, G), __shfl_xor(a.y, mask, G), __shfl_xor(a.z, mask, G), __shfl_xor(a.w, mask, G));
}
// dx = dy * y - g * y
__device__ __forceinline__ float4 f4_bwd(float4 dy, float4 y, float4 g) {
  return make_float4(dy.x * y.x - g.x * y.x, dy.y * y.y - g.y * y.y, dy.z * y.z - g.z * y.z, dy.w * y.w - g.w * y.w);
}
// online softmax statistics of two partial rows merged: (m, sum) <- (m, sum) + (m2, s2)
__device__ __forceinline__ void f4_merge(float4& m, float4& sum, float4 m2, float4 s2) {
  const float4 mn = f4_max(m, m2);
  sum = f4_add(f4_mul(sum, f4_exp_sub(m, mn)), f4_mul(s2, f4_exp_sub(m2, mn)));
  m = mn;
}

constexpr int kVec4CacheFwd = 32;   // float4s per lane held in registers
constexpr int kVec4CacheBwd = 16;

template <int G, int R4, bool BWD>
__device__ __forceinline__ void softmax_vec4_regs(const float4* __restrict__ p0, const float4* __restrict__ p1,
                                                  float4* __restrict__ po, int n4, int l, int hq, float* st_row) {
    if constexpr (!BWD) {
      float4 v[R4];
#pragma unroll
      for (int r = 0; r < R4; ++r) v[r] = (l + r * G) < n4 ? p0[l + r * G] : f4_splat(-INFINITY);
      float4 m = f4_splat(-1e9f);
#pragma unroll
      for (int r = 0; r < R4; ++r) m = f4_max(m, v[r]);
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) m = f4_max(m, f4_shfl_xor<G>(m, mask));
      float4 sum = f4_splat(0.f);
#pragma unroll
      for (int r = 0; r < R4; ++r) {
        v[r] = (l + r * G) < n4 ? f4_exp_sub(v[r], m) : f4_splat(0.f);
        sum = f4_add(sum, v[r]);
      }
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) sum = f4_add(sum, f4_shfl_xor<G>(sum, mask));
      const float4 inv = f4_rcp(sum);              // one division per row and head; items are scaled
#pragma unroll
      for (int r = 0; r < R4; ++r)
        if ((l + r * G) < n4) po[l + r * G] = f4_mul(v[r], inv);
      if (st_row) {
        st_row[0] = m.x; st_row[1] = inv.x; st_row[2] = m.y; st_row[3] = inv.y;
        st_row[4] = m.z; st_row[5] = inv.z; st_row[6] = m.w; st_row[7] = inv.w;
      }
    } else {
      float4 yy[R4], dd[R4];
      float4 g = f4_splat(0.f);
#pragma unroll
      for (int r = 0; r < R4; ++r) {
        const bool ok = (l + r * G) < n4;
        yy[r] = ok ? p0[l + r * G] : f4_splat(0.f);
        dd[r] = ok ? p1[l + r * G] : f4_splat(0.f);
      }
#pragma unroll
      for (int r = 0; r < R4; ++r) g = f4_add(g, f4_mul(dd[r], yy[r]));
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) g = f4_add(g, f4_shfl_xor<G>(g, mask));
#pragma unroll
      for (int r = 0; r < R4; ++r)
        if ((l + r * G) < n4) po[l + r * G] = f4_bwd(dd[r], yy[r], g);
    }
}

// in0 = x (forward) | y (backward), in1 = dy.  Semantics as softmax_*_seg_body (graphop_kernel.cu:170-230).
// RMAX = most float4s per lane the register tiers may hold: the kernel's register count -- and with it how many
// waves a SIMD holds -- follows the largest tier compiled in.  Graphs of short rows (products-shape: 25 slots
// x 8 heads = 50 float4s per row) are bound by the chain of dependent loads per row (segment bounds, slot bounds,
// items), i.e. by resident waves: they take the RMAX = 8 instantiation (longer rows loop).
template <int G, bool BWD, int RMAX>
__device__ __forceinline__ void softmax_vec4_group(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ in0,
    const float* __restrict__ in1, float* __restrict__ out, i64 n_seg, int h, i64 long_len, i64 block,
    const i64* __restrict__ row, float* __restrict__ stats) {
  constexpr int R4 = RMAX;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;                         // group-uniform
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const int n4 = (int)(len * h / 4);
  const int hq = h / 4;                           // lanes hq apart hold the same heads
  const float4* p0 = reinterpret_cast<const float4*>(in0 + e0 * h);
  const float4* p1 = BWD ? reinterpret_cast<const float4*>(in1 + e0 * h) : nullptr;
  float4* po = reinterpret_cast<float4*>(out + e0 * h);
  // whole row in registers: inputs read once, one exp per item.  Tiers by row length: the unrolled loops run all
  // R4 iterations whatever the row holds (a fixed 32 made the pass VALU-bound: 1.8 ms against 1.0 of traffic)
  float* st_row = (stats && l < hq) ? stats + (row[seg_chunk[s]] * h + 4 * l) * 2 : nullptr;
  if (n4 <= G * 4) { softmax_vec4_regs<G, 4, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  if (n4 <= G * 8) { softmax_vec4_regs<G, 8, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  if constexpr (R4 >= 16) {
    if (n4 <= G * 16) { softmax_vec4_regs<G, 16, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  }
  if constexpr (R4 > 16) {
    if (n4 <= G * R4) { softmax_vec4_regs<G, R4, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  }
  constexpr int U = 4;
  if constexpr (!BWD) {
    float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * G) < n4 ? p0[q0 + u * G] : f4_splat(-INFINITY);
      float4 mb = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mb = f4_max(mb, t[u]);
      sum = f4_mul(sum, f4_exp_sub(m, mb));
#pragma unroll
      for (int u = 0; u < U; ++u) sum = f4_add(sum, f4_exp_sub(t[u], mb));   // exp(-inf) = 0 past the end
      m = mb;
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= hq) f4_merge(m, sum, f4_shfl_xor<G>(m, mask), f4_shfl_xor<G>(sum, mask));
    const float4 inv = f4_rcp(sum);
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * G) < n4 ? p0[q0 + u * G] : f4_splat(0.f);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) po[q0 + u * G] = f4_mul(f4_exp_sub(t[u], m), inv);
    }
    if (stats && l < hq) {
      float* o = stats + (row[seg_chunk[s]] * h + 4 * l) * 2;
      o[0] = m.x; o[1] = inv.x; o[2] = m.y; o[3] = inv.y;
      o[4] = m.z; o[5] = inv.z; o[6] = m.w; o[7] = inv.w;
    }
  } else {
    float4 g = f4_splat(0.f);
    for (int q0 = l; q0 < n4; q0 += U * G) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) g = f4_add(g, f4_mul(p1[q0 + u * G], p0[q0 + u * G]));
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= hq) g = f4_add(g, f4_shfl_xor<G>(g, mask));
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 ty[U], td[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = (q0 + u * G) < n4;
        ty[u] = ok ? p0[q0 + u * G] : f4_splat(0.f);
        td[u] = ok ? p1[q0 + u * G] : f4_splat(0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) po[q0 + u * G] = f4_bwd(td[u], ty[u], g);
    }
  }
}

// A workgroup's row held in registers (up to 256 * R4 float4s): inputs read once, one exp per item -- every thread
// exponentiates against its OWN maximum, the (max, sum) pairs are merged through LDS, and the items are rescaled
// by exp(own max - row max) / row sum.
template <int R4, bool BWD>
__device__ __forceinline__ void softmax_vec4_long_regs(const float4* __restrict__ p0, const float4* __restrict__ p1,
                                                       float4* __restrict__ po, int n4, int hq, float4* sh_m,
                                                       float4* sh_s, float* st_row) {
  const int tid = threadIdx.x;
  float4 a[R4], b[BWD ? R4 : 1];
#pragma unroll
  for (int r = 0; r < R4; ++r) {
    const bool ok = (tid + r * kFastBlock) < n4;
    a[r] = ok ? p0[tid + r * kFastBlock] : f4_splat(BWD ? 0.f : -INFINITY);
    if constexpr (BWD) b[r] = ok ? p1[tid + r * kFastBlock] : f4_splat(0.f);
  }
  float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
  if constexpr (!BWD) {
#pragma unroll
    for (int r = 0; r < R4; ++r) m = f4_max(m, a[r]);
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      a[r] = f4_exp_sub(a[r], m);                 // 0 for the -inf of a padding slot
      sum = f4_add(sum, a[r]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R4; ++r) sum = f4_add(sum, f4_mul(b[r], a[r]));
  }
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= hq; stride >>= 1) {
    if (tid < stride) {
      if constexpr (!BWD) {
        float4 x = sh_m[tid], y = sh_s[tid];
        f4_merge(x, y, sh_m[tid + stride], sh_s[tid + stride]);
        sh_m[tid] = x; sh_s[tid] = y;
      } else {
        sh_s[tid] = f4_add(sh_s[tid], sh_s[tid + stride]);
      }
    }
    __syncthreads();
  }
  const float4 M = sh_m[tid % hq], S = sh_s[tid % hq];
  if constexpr (!BWD) {
    const float4 inv = f4_rcp(S);
    const float4 c = f4_mul(f4_exp_sub(m, M), inv);
#pragma unroll
    for (int r = 0; r < R4; ++r)
      if ((tid + r * kFastBlock) < n4) po[tid + r * kFastBlock] = f4_mul(a[r], c);
    if (st_row) {
      st_row[0] = M.x; st_row[1] = inv.x; st_row[2] = M.y; st_row[3] = inv.y;
      st_row[4] = M.z; st_row[5] = inv.z; st_row[6] = M.w; st_row[7] = inv.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R4; ++r)
      if ((tid + r * kFastBlock) < n4) po[tid + r * kFastBlock] = f4_bwd(b[r], a[r], S);
  }
}

// Rows above long_len slots: one workgroup per row, float4 items, statistics merged through LDS.
template <bool BWD, int RMAX>
__device__ __forceinline__ void softmax_vec4_long(
    const int* __restrict__ long_segs, const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ out, int h,
    float4* sh_m, float4* sh_s, i64 long_len, const i64* __restrict__ row, float* __restrict__ stats) {
  const i64 s = long_segs[blockIdx.x];
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len <= long_len) return;                    // block-uniform: the per-row groups take it
  const i64 n4 = len * h / 4;
  const int tid = threadIdx.x, hq = h / 4;
  const float4* p0 = reinterpret_cast<const float4*>(in0 + e0 * h);
  const float4* p1 = BWD ? reinterpret_cast<const float4*>(in1 + e0 * h) : nullptr;
  float4* po = reinterpret_cast<float4*>(out + e0 * h);
  if (n4 <= (RMAX >= 16 ? 16 : 8) * kFastBlock) {  // block-uniform
    float* st_row = (!BWD && stats && tid < hq) ? stats + (row[seg_chunk[s]] * h + 4 * tid) * 2 : nullptr;
    if (RMAX < 16 || n4 <= 8 * kFastBlock) softmax_vec4_long_regs<8, BWD>(p0, p1, po, (int)n4, hq, sh_m, sh_s, st_row);
    else softmax_vec4_long_regs<(RMAX >= 16 ? 16 : 8), BWD>(p0, p1, po, (int)n4, hq, sh_m, sh_s, st_row);
    return;
  }
  constexpr int U = 4;
  float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
  for (i64 q0 = tid; q0 < n4; q0 += U * kFastBlock) {
    if constexpr (!BWD) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * kFastBlock) < n4 ? p0[q0 + u * kFastBlock] : f4_splat(-INFINITY);
      float4 mb = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mb = f4_max(mb, t[u]);
      sum = f4_mul(sum, f4_exp_sub(m, mb));
#pragma unroll
      for (int u = 0; u < U; ++u) sum = f4_add(sum, f4_exp_sub(t[u], mb));
      m = mb;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * kFastBlock) < n4) sum = f4_add(sum, f4_mul(p1[q0 + u * kFastBlock], p0[q0 + u * kFastBlock]));
    }
  }
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= hq; stride >>= 1) {   // tid and tid + stride hold the same heads
    if (tid < stride) {
      if constexpr (!BWD) {
        float4 a = sh_m[tid], b = sh_s[tid];
        f4_merge(a, b, sh_m[tid + stride], sh_s[tid + stride]);
        sh_m[tid] = a; sh_s[tid] = b;
      } else {
        sh_s[tid] = f4_add(sh_s[tid], sh_s[tid + stride]);
      }
    }
    __syncthreads();
  }
  m = sh_m[tid % hq]; sum = sh_s[tid % hq];
  const float4 inv = BWD ? sum : f4_rcp(sum);
  for (i64 q0 = tid; q0 < n4; q0 += U * kFastBlock) {
    float4 t0[U], t1[BWD ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = (q0 + u * kFastBlock) < n4;
      t0[u] = ok ? p0[q0 + u * kFastBlock] : f4_splat(0.f);
      if constexpr (BWD) t1[u] = ok ? p1[q0 + u * kFastBlock] : f4_splat(0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((q0 + u * kFastBlock) < n4) {
        if constexpr (!BWD) po[q0 + u * kFastBlock] = f4_mul(f4_exp_sub(t0[u], m), inv);
        else po[q0 + u * kFastBlock] = f4_bwd(t1[u], t0[u], sum);
      }
  }
  if constexpr (!BWD) {
    if (stats && tid < hq) {
      float* o = stats + (row[seg_chunk[s]] * h + 4 * tid) * 2;
      o[0] = m.x; o[1] = inv.x; o[2] = m.y; o[3] = inv.y;
      o[4] = m.z; o[5] = inv.z; o[6] = m.w; o[7] = inv.w;
    }
  }
}

template <int G, int RMAX>
__global__ __launch_bounds__(kFastBlock) void k_softmax_fwd_vec4(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ x,
    float* __restrict__ y, i64 n_seg, int h, i64 long_len, const int* __restrict__ long_segs, int n_long,
    const i64* __restrict__ row, float* __restrict__ stats) {
  __shared__ float4 sh_m[kFastBlock];
  __shared__ float4 sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_vec4_long<false, RMAX>(long_segs, seg_chunk, indptr, seg_eptr, x, nullptr, y, h, sh_m, sh_s, long_len, row, stats);
  else
    softmax_vec4_group<G, false, RMAX>(seg_chunk, indptr, seg_eptr, x, nullptr, y, n_seg, h, long_len, (i64)blockIdx.x - n_long, row, stats);
}

template <int G, int RMAX>
__global__ __launch_bounds__(kFastBlock) void k_softmax_bwd_vec4(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ y,
    const float* __restrict__ dy, float* __restrict__ dx, i64 n_seg, int h, i64 long_len,
    const int* __restrict__ long_segs, int n_long) {
  __shared__ float4 sh_m[kFastBlock];
  __shared__ float4 sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_vec4_long<true, RMAX>(long_segs, seg_chunk, indptr, seg_eptr, y, dy, dx, h, sh_m, sh_s, long_len, nullptr, nullptr);
  else
    softmax_vec4_group<G, true, RMAX>(seg_chunk, indptr, seg_eptr, y, dy, dx, n_seg, h, long_len, (i64)blockIdx.x - n_long, nullptr, nullptr);
}

// Any h (G need not be a multiple of h): heads in an outer loop, strided reads.
template <typename T, bool BWD>
__global__ __launch_bounds__(kFastBlock) void k_softmax_seg_anyh(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ in0 /* x | y */,
    const T* __restrict__ in1 /* - | dy */, T* __restrict__ out, i64 n_seg, i64 h,
    const i64* __restrict__ row = nullptr, T* __restrict__ stats = nullptr) {
  const int lane = threadIdx.x & 63;
  const i64 s = (i64)blockIdx.x * (kFastBlock / kWave) + (threadIdx.x >> 6);
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s), e1 = seg_first(seg_eptr, seg_chunk, indptr, s + 1);
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
      if (stats && lane == 0) {
        const i64 o = (row[seg_chunk[s]] * h + t) * 2;
        stats[o] = m; stats[o + 1] = (T)1 / sum;
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
