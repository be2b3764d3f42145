// Fused attention kernels (extra op next to the reference's eight): the backward of
//   s = maskedmm_csr(Q, K);  a = sparse_softmax(s);  o = vector_spmm(a, V)
// (the composition of wrapper.py:20-30, 8-18, 44-55) WITHOUT any E-sized intermediate.  Flash-style
// recompute from the row statistics the forward leaves behind:
//   s_e  = <Q_i, K_j>                       (bitwise the forward's value: same lane layout, dot4 and
//                                            DPP tree are symmetric in their operands)
//   a_e  = exp(s_e - m_i) * linv_i          m_i = row max, linv_i = 1 / sum_e exp(s_e - m_i)
//   da_e = <dO_i, V_j>
//   ds_e = a_e * (da_e - D_i)               D_i = sum_e a_e da_e = <dO_i, o_i>   (N-sized table)
//   dQ_i += ds_e K_j      dK_j += ds_e Q_i      dV_j += a_e dO_i
// Two window-owner passes (kernels_fast.h: XCDs own L2-resident column windows, waves pull
// (window, vrow tile) tasks):
//   ROW pass (row-major CSR):  own rows (Q_i | dO_i) + (m, linv, D)_i, gathers (K_j | V_j)      -> dQ
//   COL pass (col-major CSR):  own rows (K_j | V_j), gathers (Q_i | dO_i) and (m, linv, D)_i    -> dK, dV
// Both read 4 B of neighbour id per slot and nothing else that is E-sized: no eid, no s / a / da / ds
// streams, no transposed scalar gather.  The two operands of a pass are PACKED side by side
// ([n][2F] floats, built per call by k_attn_pack: 2 x N x F x 4 B), so one slot is one contiguous
// 2F*4-byte fetch and both halves share the id -> offset arithmetic.
#pragma once
#include "kernels_fast.h"

namespace graphop {

template <int L, int NV>
struct AttnCfg {
  static constexpr int kMaxBatch = NV == 1 ? 8 : (NV == 2 ? 4 : 2);   // 2 packed rows per slot: 64 VGPRs in flight
  static constexpr int SB = L < kMaxBatch ? L : kMaxBatch;
};

// (A_r | B_r) -> out[r][2F]; with STATS also st4[r] = (m_r, linv_r, <B_r, O_r>, 0)   (h == 1)
template <int L, int NV, bool STATS>
__global__ __launch_bounds__(kFastBlock) void k_attn_pack(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, i64 n,
    const float* __restrict__ O, const float* __restrict__ stats2, float4* __restrict__ st4) {
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 r = (i64)blockIdx.x * (kFastBlock / L) + threadIdx.x / L;
  if (r >= n) return;
  float dsum = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float4 a = ld4(A, r * F4 + v * L + l);
    const float4 b = ld4(B, r * F4 + v * L + l);
    reinterpret_cast<float4*>(out)[r * 2 * F4 + v * L + l] = a;
    reinterpret_cast<float4*>(out)[r * 2 * F4 + F4 + v * L + l] = b;
    if constexpr (STATS) dsum += dot4(b, ld4(O, r * F4 + v * L + l));
  }
  if constexpr (STATS) {
    dsum = group_sum<L>(dsum);
    if (l == 0) st4[r] = make_float4(stats2[r * 2], stats2[r * 2 + 1], dsum, 0.f);
  }
}

// One lane group's strip of a (window, vrow tile) task.  `own` = the group's K packed own rows in
// LDS ([K][2*NV][L] float4, a straight copy of the packed table rows).  `sink(k, acc0, acc1)`
// receives the finished sums of granule k (group-uniform call): acc0 = sum ds * X0, acc1 = sum a * X1.
// STAGED: the neighbour ids come from the dealt layout through IdStage (kernels_fast.h) -- idx32 is
// then ids_w, pos0 the strip's start in it and idbuf the group's LDS ring.
template <int L, int NV, bool COL, bool OFF32, bool STAGED, typename Sink, typename Stage>
__device__ __forceinline__ void attn_bwd_strip(Sink&& sink, Stage&& stage_rows,
                                               const float4* __restrict__ own, int lo_l, int n_l,
                                               const int* __restrict__ idx32,
                                               const float* __restrict__ XT,
                                               const float4* __restrict__ stats4, float4 own_st,
                                               int l, int pos0 = 0, int* __restrict__ idbuf = nullptr) {
  constexpr int SB = AttnCfg<L, NV>::SB;
  constexpr int F4 = L * NV;
  constexpr unsigned ROWB = 2u * F4 * 16u;   // bytes of a packed row
  StripMap m;
  m.init<L>(lo_l, n_l, l);
  if (m.total == 0) return;
  float4 acc0[NV], acc1[COL ? NV : 1];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc0[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int v = 0; v < (COL ? NV : 1); ++v) acc1[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k_cur = -1;
  auto spill = [&]() {
    if (k_cur >= 0) {
      sink(k_cur, acc0, acc1);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc0[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int v = 0; v < (COL ? NV : 1); ++v) acc1[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // Id pipeline.  Stage A (flat slot -> granule, neighbour id) runs one batch ahead; in the COL pass
  // the row statistics of the gathered row are a second dependent load, so stage A runs two batches
  // ahead and stage B (the 16-B statistics) one batch ahead.
  struct Pre { int k, src, live; float4 st; };
  IdStage<L, 1> ids;
  if constexpr (STAGED) ids.init(idx32, nullptr, pos0, idbuf, l, m.total);
  auto stage_a = [&](int jbase, Pre& p) {
    const int j = jbase + l;
    const int jc = j < m.total ? j : m.total - 1;
    int e;
    m.locate<L>(jc, p.k, e);   // every lane takes part in the shuffles
    p.live = (l < SB && j < m.total) ? 1 : 0;
    p.src = 0;
    // slots past the end re-read the strip's last neighbour id with weights 0, so the batch loop
    // needs no per-slot clamping
    if constexpr (STAGED) {
      if (jbase < m.total) {   // group-uniform
        ids.advance(jbase);
        p.src = ids.id(jc);
      }
    } else {
      if (l < SB) p.src = idx32[e];
    }
  };
  auto stage_b = [&](Pre& p) {
    if constexpr (COL) {
      if (l < SB) p.st = stats4[p.src];
    }
  };
  Pre p1, p2;
  p1.st = p2.st = make_float4(0.f, 0.f, 0.f, 0.f);
  stage_a(0, p1);
  if constexpr (COL) {
    stage_b(p1);
    stage_a(SB, p2);
  }
  stage_rows();   // own rows -> LDS once the first id requests are in flight
  const char* lds_l = reinterpret_cast<const char*>(own) + l * 16;
  for (int jb = 0; jb < m.total; jb += SB) {
    const Pre cur = p1;
    const unsigned my_off = OFF32 ? (unsigned)cur.src * ROWB : (unsigned)cur.src;
    const unsigned my_koff = (unsigned)cur.k * ROWB;
    float4 x0[SB][NV], x1[SB][NV];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned o = group_bcast<L, u>(my_off);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (OFF32) {
          x0[u][v] = ld4_off(XT, o + (unsigned)((v * L + l) * 16));
          x1[u][v] = ld4_off(XT, o + (unsigned)((F4 + v * L + l) * 16));
        } else {
          const float4* rowp = reinterpret_cast<const float4*>(XT) + (i64)o * (2 * F4) + v * L + l;
          x0[u][v] = rowp[0];
          x1[u][v] = rowp[F4];
        }
      }
    });
    // ids / statistics of the following batches (issued behind the row requests)
    if constexpr (COL) {
      p1 = p2;
      stage_b(p1);
      stage_a(jb + 2 * SB, p2);
    } else {
      stage_a(jb + SB, p1);
    }
    float ps[SB], pd[SB];
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const unsigned ko = group_bcast<L, u>(my_koff);
      float p = 0.f, q = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 y0 = *reinterpret_cast<const float4*>(lds_l + ko + v * L * 16);
        const float4 y1 = *reinterpret_cast<const float4*>(lds_l + ko + (F4 + v * L) * 16);
        if (v == 0) { p = dot4(y0, x0[u][0]); q = dot4(y1, x1[u][0]); }
        else { p += dot4(y0, x0[u][v]); q += dot4(y1, x1[u][v]); }
      }
      ps[u] = p; pd[u] = q;
    });
    const float my_s = group_dots_to_owner<L, SB>(ps, l);
    const float my_da = group_dots_to_owner<L, SB>(pd, l);
    // lane u < SB owns slot u: one exp per slot, then broadcast
    float st_m, st_linv, st_D;
    if constexpr (COL) {
      st_m = cur.st.x; st_linv = cur.st.y; st_D = cur.st.z;
    } else {
      st_m = __shfl(own_st.x, cur.k, L); st_linv = __shfl(own_st.y, cur.k, L); st_D = __shfl(own_st.z, cur.k, L);
    }
    float a_l = 0.f, ds_l = 0.f;
    if (cur.live) {
      a_l = expf(my_s - st_m) * st_linv;
      ds_l = a_l * (my_da - st_D);
    }
    static_for<SB>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      const int kt = group_bcast<L, u>(cur.k);
      if (kt != k_cur) {   // group-uniform
        spill();
        k_cur = kt;
      }
      const float dsu = group_bcast<L, u>(ds_l);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        acc0[v].x = fmaf(dsu, x0[u][v].x, acc0[v].x);
        acc0[v].y = fmaf(dsu, x0[u][v].y, acc0[v].y);
        acc0[v].z = fmaf(dsu, x0[u][v].z, acc0[v].z);
        acc0[v].w = fmaf(dsu, x0[u][v].w, acc0[v].w);
      }
      if constexpr (COL) {
        const float au = group_bcast<L, u>(a_l);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          acc1[v].x = fmaf(au, x1[u][v].x, acc1[v].x);
          acc1[v].y = fmaf(au, x1[u][v].y, acc1[v].y);
          acc1[v].z = fmaf(au, x1[u][v].z, acc1[v].z);
          acc1[v].w = fmaf(au, x1[u][v].w, acc1[v].w);
        }
      }
    });
  }
  spill();
}

// Window-owner driver of both passes (task pipeline as in k_sddmm_wown_f32).
//   COL = false: OWN = (Q | dO) packed [n_rows][2F], XT = (K | V) packed, out0 = dQ
//   COL = true : OWN = (K | V) packed [n_cols][2F], XT = (Q | dO) packed, out0 = dK, out1 = dV
// stats4[i] = (m_i, linv_i, D_i, 0) per ROW i of the attention matrix in both passes.
template <int L, int NV, bool COL, bool OFF32, int BPC, bool STAGED = false>
__global__ __launch_bounds__(kFastBlock, BPC) void k_attn_bwd_wown_f32(
    SweepView s, const float* __restrict__ OWN, const float* __restrict__ XT,
    const float4* __restrict__ stats4, float* __restrict__ out0, float* __restrict__ out1) {
  extern __shared__ float4 lds[];
  constexpr i64 F4 = (i64)L * NV;
  constexpr int GW = kWave / L;
  constexpr int GPB = kFastBlock / L;
  const int l = threadIdx.x % L;
  const int g_in_blk = threadIdx.x / L;
  float4* mine = lds + (i64)g_in_blk * s.K * 2 * F4;   // [K][2*NV][L]
  int* idbuf = reinterpret_cast<int*>(lds + (i64)GPB * s.K * 2 * F4) + g_in_blk * StageCfg<L, 1>::kLdsIntsPerGroup;
  const int tile = GW * s.K;
  const int tiles = (s.V + tile - 1) / tile;
  WownQueue queue(s, tiles);
  int w, t;
  bool more = queue.pull(w, t);
  int raw = more ? queue.issue() : -1;
  WownTask<L> cur, nxt;
  cur.pos = 0;
  if (more) { if constexpr (STAGED) cur.load_dealt(s, w, t, tile, tiles); else cur.load(s, w, t, tile); }
  while (more) {
    int wn = 0, tn = 0;
    const bool more_n = queue.resolve(raw, wn, tn);
    raw = more_n ? queue.issue() : -1;
    nxt.nv = 0; nxt.lo = nxt.hi = nxt.row = nxt.pos = 0;
    if (more_n) { if constexpr (STAGED) nxt.load_dealt(s, wn, tn, tile, tiles); else nxt.load(s, wn, tn, tile); }
    float4 own_st = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (!COL) own_st = stats4[cur.row];   // lane k: statistics of the group's k-th vrow
    auto stage_rows = [&]() {   // packed own rows of this task's non-empty granules -> LDS
      for (int k = 0; k < cur.nv; ++k) {
        const i64 row = __shfl(cur.row, k, L);
        if (__shfl(cur.hi - cur.lo, k, L) == 0) continue;   // group-uniform
#pragma unroll
        for (int v = 0; v < 2 * NV; ++v) mine[(k * 2 * NV + v) * L + l] = ld4(OWN, row * 2 * F4 + v * L + l);
      }
    };
    const int row_l = cur.row;
    auto to_out = [&](int k, const float4 (&acc0)[NV], const float4 (&acc1)[COL ? NV : 1]) {
      const i64 r = __shfl(row_l, k, L);
#ifdef GRAPHOP_DEBUG_NOFLUSH   // measurement builds only (wrong results): what the partial-row flushes cost
      if (acc0[0].x == 1234.5f) atomic_flush_dense<L, NV>(out0, r, acc0, l);
      if constexpr (COL) { if (acc1[0].x == 1234.5f) atomic_flush_dense<L, NV>(out1, r, acc1, l); }
#else
      atomic_flush_dense<L, NV>(out0, r, acc0, l);
      if constexpr (COL) atomic_flush_dense<L, NV>(out1, r, acc1, l);
#endif
    };
    if constexpr (STAGED)
      attn_bwd_strip<L, NV, COL, OFF32, true>(to_out, stage_rows, mine, cur.lo, cur.hi - cur.lo, s.ids_w, XT,
                                              stats4, own_st, l, __shfl(cur.pos, 0, L), idbuf);
    else
      attn_bwd_strip<L, NV, COL, OFF32, false>(to_out, stage_rows, mine, cur.lo, cur.hi - cur.lo, s.idx32, XT,
                                               stats4, own_st, l);
    cur = nxt;
    more = more_n;
  }
}

// ---- chunk-driver form of the two passes (no window structure) -------------------------------------
// For graphs the window drivers do not take (rows too short for column windows, tables that fit the
// L2, chunk layouts without sorted rows): one lane group walks a run of consecutive chunks like
// k_spmm_f32, keeps the packed own row and the accumulators in registers while the row id does not
// change, gathers the packed neighbour rows straight from the table (HBM / Infinity Cache / L2 as
// the table size has it) and stores (OWNED: the row's chunks all lie inside this group's run) or
// atomically adds the sums.  Same per-slot arithmetic as the strips; the dot products are reduced
// slot by slot in the chunk drivers' order (group_sum), which is the order the forward's SDDMM
// used on these graphs.
template <int L, int NV, bool COL, bool OWNED>
__global__ __launch_bounds__(kFastBlock) void k_attn_bwd_rows_f32(
    const i64* __restrict__ row, const i64* __restrict__ indptr, const i64* __restrict__ indices,
    const float* __restrict__ OWN, const float* __restrict__ XT, const float4* __restrict__ stats4,
    float* __restrict__ out0, float* __restrict__ out1, i64 n_chunks, int chunks_per_group) {
  constexpr int SB = AttnCfg<L, NV>::SB;
  constexpr i64 F4 = (i64)L * NV;
  const int l = threadIdx.x % L;
  const i64 gid = (i64)blockIdx.x * GroupCfg<L>::kGroupsPerBlock + threadIdx.x / L;
  const i64 c0 = gid * chunks_per_group;
  i64 c1 = c0 + chunks_per_group;
  if (c1 > n_chunks) c1 = n_chunks;
  if (c0 >= c1) return;
  i64 row_before = -1, row_after = -1;
  if constexpr (OWNED) {
    if (c0 > 0) row_before = row[c0 - 1];
    if (c1 < n_chunks) row_after = row[c1];
  }
  float4 y0[NV], y1[NV], acc0[NV], acc1[COL ? NV : 1];
  float4 own_st = make_float4(0.f, 0.f, 0.f, 0.f);
  auto zero_acc = [&]() {
#pragma unroll
    for (int v = 0; v < NV; ++v) acc0[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < (COL ? NV : 1); ++v) acc1[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto flush = [&](i64 r) {
    if (OWNED && r != row_before && r != row_after) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        reinterpret_cast<float4*>(out0)[r * F4 + v * L + l] = acc0[v];
        if constexpr (COL) reinterpret_cast<float4*>(out1)[r * F4 + v * L + l] = acc1[v];
      }
    } else {
      atomic_flush<L, NV>(out0, r, acc0, l);
      if constexpr (COL) atomic_flush<L, NV>(out1, r, acc1, l);
    }
  };
  zero_acc();
  i64 cur_row = -1;
  bool dirty = false;
  for (i64 c = c0; c < c1; ++c) {
    const i64 r = row[c];
    if (r != cur_row) {
      if (dirty) { flush(cur_row); zero_acc(); dirty = false; }
      cur_row = r;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        y0[v] = ld4(OWN, r * 2 * F4 + v * L + l);
        y1[v] = ld4(OWN, r * 2 * F4 + F4 + v * L + l);
      }
      if constexpr (!COL) own_st = stats4[r];
    }
    const i64 j0 = indptr[c], j1 = indptr[c + 1];
    if (j1 > j0) dirty = true;
    for (i64 jb = j0; jb < j1; jb += SB) {
      const int nb = (j1 - jb) < SB ? (int)(j1 - jb) : SB;
      // slots past the end re-read the batch's last neighbour with weights 0
      i64 my_src = 0;
      float4 st = own_st;
      if (l < SB) {
        my_src = indices[jb + (l < nb ? l : nb - 1)];
        if constexpr (COL) st = stats4[my_src];
      }
      float4 x0[SB][NV], x1[SB][NV];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const i64 src = __shfl(my_src, u, L);
        const float4* rowp = reinterpret_cast<const float4*>(XT) + src * (2 * F4) + l;
#pragma unroll
        for (int v = 0; v < NV; ++v) { x0[u][v] = rowp[v * L]; x1[u][v] = rowp[F4 + v * L]; }
      }
      float my_s = 0.f, my_da = 0.f;
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        float p = dot4(y0[0], x0[u][0]), q = dot4(y1[0], x1[u][0]);
#pragma unroll
        for (int v = 1; v < NV; ++v) { p += dot4(y0[v], x0[u][v]); q += dot4(y1[v], x1[u][v]); }
        p = group_sum<L>(p);
        q = group_sum<L>(q);
        if (l == u) { my_s = p; my_da = q; }
      }
      float a_l = 0.f, ds_l = 0.f;
      if (l < nb) {
        a_l = expf(my_s - st.x) * st.y;
        ds_l = a_l * (my_da - st.z);
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const float dsu = __shfl(ds_l, u, L);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          acc0[v].x = fmaf(dsu, x0[u][v].x, acc0[v].x); acc0[v].y = fmaf(dsu, x0[u][v].y, acc0[v].y);
          acc0[v].z = fmaf(dsu, x0[u][v].z, acc0[v].z); acc0[v].w = fmaf(dsu, x0[u][v].w, acc0[v].w);
        }
        if constexpr (COL) {
          const float au = __shfl(a_l, u, L);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            acc1[v].x = fmaf(au, x1[u][v].x, acc1[v].x); acc1[v].y = fmaf(au, x1[u][v].y, acc1[v].y);
            acc1[v].z = fmaf(au, x1[u][v].z, acc1[v].z); acc1[v].w = fmaf(au, x1[u][v].w, acc1[v].w);
          }
        }
      }
    }
  }
  if (dirty) flush(cur_row);
}

// Row statistics of the general (plan-less) softmax path: its scratch holds max / sum per row.
template <typename T>
__global__ void k_attn_stats_from_ws(const T* __restrict__ max_val, const T* __restrict__ sum,
                                     T* __restrict__ stats, i64 n) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const T sm = sum[i];
    stats[i * 2] = max_val[i];
    stats[i * 2 + 1] = sm > (T)0 ? (T)1 / sm : (T)0;
  }
}

}  // namespace graphop
