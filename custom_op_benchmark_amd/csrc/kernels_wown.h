// WINDOW-OWNER drivers (plan): the gathered table is cut into W column windows (<= 4 MB, the size of an XCD's L2;
// 32 MB Infinity-Cache windows for tables beyond 128 MB); because ids ascend inside a row, a row's slots inside a
// window are one contiguous range.  vrow v = (a piece of) row vr_row[v]; inside window w it owns slots
// [wp_lo[w*V+v], wp_hi[w*V+v]) whose neighbour ids lie in [w*win_cols, (w+1)*win_cols).  A row longer than T slots
// is cut into P pieces that each take 1/P of the row's slots in EVERY window.
#pragma once
#include "kernels_strip.h"

namespace graphop {

constexpr int kSweepBlocksPerCu = 4;   // most co-resident 256-thread workgroups per CU any window-owner kernel
                                       // is compiled for (<= 128 VGPRs, 32 KB LDS each)
// Resident workgroups per CU a given instantiation is compiled for (its __launch_bounds__ and the
// grid the host launches): the one-head 16..512-float rows fit 128 VGPRs without spilling; rows of
// 1024 floats (NV = 4) and per-head weights (H1 = false: a weight register per slot and float4) get 168.
__host__ __device__ constexpr int sweep_bpc(int NV, bool H1) {
  return (H1 && NV < 4) ? 4 : 3;
}
struct SweepView {
  const int* wp_lo;   // [W * V]
  const int* wp_hi;   // [W * V]
  const int* vr_row;  // [V]
  const int* idx32;   // [E]
  const int* eid32;   // [E] or nullptr when eid is the identity
  int* sync;          // the eight per-XCD task-queue heads (sync[y * kSyncStride]), zeroed before the launch
  int V, W, K;
  i64 win_bytes;      // bytes of gathered table per window
  i64 table_bytes;    // bytes of the gathered table
  int touch;          // bit 0: touch the granule's id lines at task start, bit 1: also its edge-id / weight lines
  // dealt (window-major) layout of the window-owner tasks, or nullptr (plan.hip, Sweep::Dealt)
  const int4* rec;    // [W * tiles * tile] (first slot, length, row id, position in ids_w) per granule
  const int* ids_w;   // neighbour ids in dealt order: a lane group's strip is one contiguous aligned run
  const int* eids_w;  // edge ids in the same order (nullptr when eid is the identity)
};

constexpr int kSyncStride = 64;   // ints between counters (queue heads here, pacer counters in kernels_walk.h): one 256-B line each
constexpr int kSyncXcds = 8;

// Every XCD (read from HW_REG_XCC_ID) owns the windows w = x, x+8, ... and its waves pull (window, vrow-tile)
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
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1)) void k_sddmm_wown_f32(
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

// Staged form (h == 1 or heads of D4 float4s, identity eid, dealt layout in the view): tasks come with their
// granules already dealt, ids through the group's LDS buffer (behind the A rows in dynamic LDS).
// D4 > 0: L / D4 heads of D4 float4s each (NV == 1, fp32, table < 4 GiB).  T / OFF32 as in sddmm_strip_staged.
template <int L, int NV, int D4, typename T, bool OFF32>
__device__ __forceinline__ void sddmm_wown_staged_body(const SweepView& s, const T* __restrict__ A,
                                                       const T* __restrict__ B, T* __restrict__ y) {
  using vec = typename RowT<T>::vec;
  extern __shared__ float4 lds_raw[];
  vec* lds = reinterpret_cast<vec*>(lds_raw);
  constexpr i64 F4 = (i64)L * NV;
  constexpr int GW = kWave / L;
  constexpr int GPB = kFastBlock / L;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  vec* mine = lds + (i64)g_in_blk * s.K * F4;  // [K][NV][L]
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
        for (int v = 0; v < NV; ++v) mine[(k * NV + v) * L + l] = ld16<T>(A, row * F4 + v * L + l);
      }
    };
    if constexpr (D4 == 0)
      sddmm_strip_staged<L, NV, T, OFF32>(mine, cur.lo, cur.hi - cur.lo, __shfl(cur.pos, 0, L), s.ids_w, idbuf, B, y, l,
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
// (the 64-bit-offset instantiations are compiled for 3 workgroups per CU -- the grid the host launches by default --:
// the address arithmetic does not fit the 128 VGPRs of 4)
template <int L, int NV, int D4 = 0, bool OFF32 = true>
__global__ __launch_bounds__(kFastBlock, OFF32 ? sweep_bpc(NV, D4 == 0) : 3) void k_sddmm_wown_staged_f32(
    SweepView s, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ y) {
  sddmm_wown_staged_body<L, NV, D4, float, OFF32>(s, A, B, y);
}
// fp64, one head (graphop_kernel.cu:291 dispatches both types through the same kernels)
template <int L, int NV, bool OFF32 = true>
__global__ __launch_bounds__(kFastBlock, 3) void k_sddmm_wown_staged_f64(
    SweepView s, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ y) {
  sddmm_wown_staged_body<L, NV, 0, double, OFF32>(s, A, B, y);
}

template <int L, int NV, bool H1, bool EID_ID, bool OFF32>
__global__ __launch_bounds__(kFastBlock, sweep_bpc(NV, H1)) void k_spmm_wown_f32(
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
__global__ __launch_bounds__(kFastBlock, L == 64 ? 3 : sweep_bpc(NV, true)) void k_spmm_wown_staged_f32(
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

}  // namespace graphop
