// STRIP inner loops of the column-window drivers: what one lane group does inside one window -- the K granules it
// owns walked as ONE flat slot list in full 16-slot batches, ids of the next batch(es) in flight behind the current
// batch's row requests, rows through scalar-base + 32-bit-offset loads; staged forms read their ids from the
// plan's dealt (window-major) layouts through an LDS ring (IdStage).
#pragma once
#include "kernels_base.h"

namespace graphop {

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

// Staged SDDMM strip (h == 1, identity eid; dealt layout).  T = float or double (rows of 16 * L * NV bytes either way);
// OFF32 = false: tables of 4 GiB and more (64-bit row offsets).
template <int L, int NV, typename T, bool OFF32, typename Stage>
__device__ __forceinline__ void sddmm_strip_staged(const typename RowT<T>::vec* __restrict__ rowsA, int lo_l, int n_l,
                                                   int pos0, const int* __restrict__ ids_w,
                                                   int* __restrict__ idbuf, const T* __restrict__ B,
                                                   T* __restrict__ y, int l, Stage&& stage_rows) {
  using TR = RowT<T>;
  using vec = typename TR::vec;
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
  constexpr int kStoreBatch = (NV == 1 && L == 16 && sizeof(T) == 4) ? 4 : 1;
  T held_res[kStoreBatch];
  int held_e[kStoreBatch];
#pragma unroll
  for (int q = 0; q < kStoreBatch; ++q) { held_res[q] = 0; held_e[q] = -1; }
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
    const unsigned my_off = OFF32 ? (unsigned)nsrc * (unsigned)(F4 * 16) : (unsigned)nsrc;
    vec b[SB][NV];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (OFF32) b[u][v] = ld16_off<T>(B, o + (unsigned)((v * L + l) * 16));
        else b[u][v] = ld16_row64<T>(B, o, (unsigned)(F4 * 16), (unsigned)((v * L + l) * 16));
      }
    });
    if (n_held == kStoreBatch) flush_results();
    T part[SB];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned ko = group_bcast<L, u>(my_koff);
      vec av[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) av[v] = *reinterpret_cast<const vec*>(lds_l + ko + v * L * 16);
      T p = TR::dot(av[0], b[u][0]);
#pragma unroll
      for (int v = 1; v < NV; ++v) p += TR::dot(av[v], b[u][v]);
      part[u] = p;
    });
    const T res = group_dots_to_owner<L, SB>(part, l);
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

}  // namespace graphop
