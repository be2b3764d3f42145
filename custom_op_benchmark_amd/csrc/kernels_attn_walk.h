// Fused attention FORWARD as ONE walk-style pass (extra op):
//   o_i = sum_j softmax_j(<Q_i, K_j>) V_j        = wrapper.py:20-30 (MaskedMMCSR) + :8-18 (SparseSoftmax) + :44-55 (VectorSPMM)
// without s or a ever leaving the chip.  Ownership and schedule are the walk drivers' (kernels_walk.h): a lane group
// owns a BIN of rows for a whole round -- here Q_i, the running output row and the running (max, sum) of every row
// live in its LDS -- and walks all column windows; eight worker waves issue row requests and nothing else, feeder
// waves stage the bin's (row-in-bin | neighbour id) run through LDS rings, a per-XCD soft pacer keeps the waves of
// an XCD inside the same window of BOTH gathered tables (K and V: windows of half the size).
//
// Online softmax per (row, window) granule.  A lane group keeps a LOCAL partial (m, l, o) of the row it is in: per
// slot  mn = max(m, s); o = o * exp(m - mn) + exp(s - mn) * V_j; l likewise.  At a row change the local partial is
// MERGED into the row's LDS state with the deferred read-modify-write of the SpMM walk (the state is requested at
// this row change, merged and written back at the next one: no LDS round trip between two slots):
//   mn = max(m_lds, m_loc);  o_lds = o_lds * exp(m_lds - mn) + o_loc * exp(m_loc - mn);  l likewise.
// At the end of the round a row that is wholly inside the bin is normalised and stored (o, and the statistics
// (m, 1 / l) the fused backward recomputes from); the at most two rows a bin shares with its neighbours (hub rows
// cut by bin boundaries) leave as PIECE records (m, l, o) that three tiny kernels merge afterwards
// (k_attn_piece_max / _add / _fin): a softmax cannot be merged by float atomics alone.
// Padding slots of a run's last chunk carry the row-in-bin 63: they gather a valid row and contribute nothing.
#pragma once
#include "kernels_walk.h"

namespace graphop {

constexpr int kAttnPadK = 63;          // row-in-bin of a padding slot (kWalkK <= 15 < 63: the 6-bit field has room)
constexpr float kAttnNegBig = -3.0e38f;   // "no score yet" (finite: differences of two of them are 0, not NaN)

struct AttnWalkArgs {
  const float* Q;       // [n_q][F]   own rows
  const float* K;       // [n_k][F]   gathered
  const float* V;       // [n_k][F]   gathered
  float* o;             // [n_q][F]
  float* stats;         // [n_q][2]   (m, 1 / l); rows without slots keep the (0, 0) of the caller's zero fill
  int* p_row;           // piece records of shared rows, [2 * bins]: row | first-piece flag, -1 = none (pre-filled)
  float* p_ml;          // [2 * bins][2]
  float* p_o;           // [2 * bins][F]
};

// LDS per lane group: K x (Q row | o row) + K x (m, l) + the id ring
template <int L>
__host__ __device__ constexpr size_t attn_walk_group_bytes(int K) {
  return (size_t)K * 2 * L * 16 + (((size_t)K * 8 + 15) & ~(size_t)15) + (size_t)kFeedChunk * kFeedRing * 4;
}
template <int L>
__host__ __device__ constexpr int attn_walk_rows() {
  int k = kWalkK;
  while (k > 0 && (size_t)(kWalkWorkers / L) * attn_walk_group_bytes<L>(k) > 160 * 1024 - 2048) --k;
  return k;
}

__device__ __forceinline__ float attn_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }   // x <= 0

template <int L>
__device__ __forceinline__ void attn_fwd_walk_body(const WalkView& s, const AttnWalkArgs& a) {
  static_assert(L == 16, "256-B rows (d = 64): 8-slot batches reduced by the 16-lane transpose tree");
  extern __shared__ float4 lds_raw[];
  constexpr int GPB = kWalkWorkers / L;
  constexpr int SB = 8;                       // slots per batch: two gathered rows per slot, 64 VGPRs in flight
  constexpr int RING = kFeedChunk * kFeedRing;
  constexpr unsigned ROWB = 16u * L;
  const int K = s.K;
  const size_t GB = attn_walk_group_bytes<L>(K);
  __shared__ int pace_words[16];
  __shared__ int feed_ready[GPB], feed_done[GPB];
  constexpr int GW = kWave / L, NQ = GPB / GW;
  static_assert(NQ <= 8, "pacer progress words");
  __shared__ int tk_next, quad_done[NQ], quad_len[NQ], bin_total[GPB], bin_seg[GPB];
  __shared__ int wg_abort;
  if (threadIdx.x == 0) wg_abort = 0;
  if (threadIdx.x < GPB) { feed_ready[threadIdx.x] = 0; feed_done[threadIdx.x] = 0; bin_total[threadIdx.x] = 0; bin_seg[threadIdx.x] = 0; }
  if (threadIdx.x < NQ) { quad_done[threadIdx.x] = 0; quad_len[threadIdx.x] = SB; }
  if (threadIdx.x == 0) tk_next = 0;
  __syncthreads();
  const long long t_start = __builtin_amdgcn_s_memtime();
  WalkPacer pacer(s, pace_words, NQ);
  char* lds_b = reinterpret_cast<char*>(lds_raw);
  auto ring_of = [&](int bin) { return reinterpret_cast<int*>(lds_b + (size_t)bin * GB + (GB - (size_t)RING * 4)); };

  if (threadIdx.x >= kWalkWorkers) {
    // ---------------- feeder wave: the lane groups' (row-in-bin | id) runs -> LDS rings ----------------
    const int h = (threadIdx.x - kWalkWorkers) % kWave;
    constexpr int NG = GPB / kFeeders > 0 ? GPB / kFeeders : 1;
    const int g0 = __builtin_amdgcn_readfirstlane(((threadIdx.x - kWalkWorkers) / kWave) * NG);
    if (g0 >= GPB) return;
    int chunk_base[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) chunk_base[g] = 0;
    for (int r = 0; r < s.rounds; ++r) {
      int pos0[NG], total[NG], nchunk[NG], c[NG], idW[NG], idA[NG];
      auto load_id = [&](int g, int ck) {
        int idw = kAttnPadK << kWalkKShift;
        if (ck * kFeedChunk < total[g]) {
          const int j = ck * kFeedChunk + h;
          const int jc = j < total[g] ? j : total[g] - 1;
          idw = __builtin_nontemporal_load(s.ids + pos0[g] + jc);
          if (j >= total[g]) idw = (idw & kWalkIdMask) | (kAttnPadK << kWalkKShift);   // padding: a valid neighbour, no row
        }
        return idw;
      };
      int left = 0;
      {
        int p_l = 0, t_l = 0;
        if (h < NG) {
          const i64 tb = walk_bin_index_of<GPB>(s, r, g0 + h);
          p_l = s.bin_pos[tb];
          t_l = s.bin_cum[tb];
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          pos0[g] = __builtin_amdgcn_readlane(p_l, g);
          total[g] = __builtin_amdgcn_readlane(t_l, g);
          nchunk[g] = (total[g] + kFeedChunk - 1) / kFeedChunk;
          c[g] = 0;
          left += nchunk[g];
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) idW[g] = load_id(g, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) idA[g] = load_id(g, 1);
      while (left > 0) {
        const int done_l = h < NG ? lds_ld(feed_done + g0 + h) : 0;
        unsigned adv = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (c[g] >= nchunk[g]) continue;
          const int gc = chunk_base[g] + c[g];
          if (gc - __builtin_amdgcn_readlane(done_l, g) >= kFeedRing) continue;
          adv |= 1u << g;
          ring_of(g0 + g)[(gc % kFeedRing) * kFeedChunk + h] = idW[g];
          lds_st(feed_ready + g0 + g, gc + 1);
        }
        if (adv == 0) {
          __builtin_amdgcn_s_sleep(4);
          if (walk_aborted(&wg_abort)) return;
          continue;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (!((adv >> g) & 1)) continue;
          idW[g] = idA[g];
          idA[g] = load_id(g, c[g] + 2);
          ++c[g];
          --left;
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) chunk_base[g] += nchunk[g];
    }
    return;
  }

  // ---------------- worker waves ----------------
  const int l = threadIdx.x % L;
  const int gq = (threadIdx.x / L) % GW;
  const int n_steps = s.rounds * s.steps;
  for (;;) {
    int t = 0;
    if ((threadIdx.x & (kWave - 1)) == 0)
      t = __hip_atomic_fetch_add(&tk_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    const int gstep = t / NQ, q = t % NQ;
    if (gstep >= n_steps || walk_aborted(&wg_abort)) break;
    const int r = gstep / s.steps, sidx = gstep - r * s.steps;
    {
      int it = 0, dead = 0;
      while (__builtin_amdgcn_readfirstlane(lds_ld(quad_done + q)) < gstep) {
        __builtin_amdgcn_s_sleep(1);
        if (walk_aborted(&wg_abort)) { dead = 1; break; }
        if (++it > kWalkSpinQuad) { walk_fail(&wg_abort, kWalkErrQuad); dead = 1; break; }
      }
      if (dead) break;
    }
    pacer.wait_enter(gstep, gstep);
    const int bin = q * GW + gq;
    char* gbase = lds_b + (size_t)bin * GB;
    float4* rowsQ = reinterpret_cast<float4*>(gbase);                 // [K][L]
    float4* accO = rowsQ + K * L;                                     // [K][L]
    float2* ml = reinterpret_cast<float2*>(accO + K * L);             // [K] (running max, running sum)
    const int* ring = ring_of(bin);
    const i64 tb = walk_bin_index_of<GPB>(s, r, bin);
    int total, step_len;
    if (sidx == 0) {                                  // the bin's round starts: state to zero, Q rows into LDS
      total = s.bin_cum[tb];
      const int my_row = l < K ? s.bin_rows[tb * K + l] : -1;
      for (int k = 0; k < K; ++k) {
        accO[k * L + l] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int rec = __shfl(my_row, k, L);
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rec != -1) qv = ld4(a.Q, (i64)(rec & kWalkRowMask) * L + l);
        rowsQ[k * L + l] = qv;
      }
      if (l < K) ml[l] = make_float2(kAttnNegBig, 0.f);
      const int quad_total = wave_max_int<L>(total);
      step_len = (((quad_total + s.steps - 1) / s.steps + SB - 1) / SB) * SB;
      step_len = step_len > 0 ? step_len : SB;
      if (l == 0) bin_total[bin] = total;
      if ((threadIdx.x & (kWave - 1)) == 0) quad_len[q] = step_len;
    } else {
      total = bin_total[bin];
      step_len = quad_len[q];
    }
    const int seg_base = bin_seg[bin];
    const int j0 = sidx * step_len;
    int j1 = j0 + step_len;
    {
      const int quad_total = wave_max_int<L>(total);
      j1 = j1 < quad_total ? j1 : quad_total;
    }
    // local partial of the row the group is in; pending partial of the row it left (merged at the next row change)
    float m_c = kAttnNegBig, l_c = 0.f;
    float4 o_c = make_float4(0.f, 0.f, 0.f, 0.f);
    int k_cur = -1, pend_k = -1;
    float pend_m = kAttnNegBig, pend_l = 0.f;
    float4 pend_o = make_float4(0.f, 0.f, 0.f, 0.f), rd_o = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 rd_ml = make_float2(kAttnNegBig, 0.f);
    auto finish_pending = [&]() {
      if (pend_k >= 0) {
        const float mn = fmaxf(rd_ml.x, pend_m);
        const float ea = attn_exp(rd_ml.x - mn), eb = attn_exp(pend_m - mn);
        float4 o;
        o.x = fmaf(rd_o.x, ea, pend_o.x * eb); o.y = fmaf(rd_o.y, ea, pend_o.y * eb);
        o.z = fmaf(rd_o.z, ea, pend_o.z * eb); o.w = fmaf(rd_o.w, ea, pend_o.w * eb);
        accO[pend_k * L + l] = o;
        if (l == 0) ml[pend_k] = make_float2(mn, fmaf(rd_ml.y, ea, pend_l * eb));
      }
    };
    auto row_change = [&](int kt) {
      finish_pending();
      pend_k = k_cur;
      if (k_cur >= 0) {
        pend_m = m_c; pend_l = l_c; pend_o = o_c;
        rd_o = accO[k_cur * L + l];
        rd_ml = ml[k_cur];
      }
      m_c = kAttnNegBig; l_c = 0.f; o_c = make_float4(0.f, 0.f, 0.f, 0.f);
      k_cur = kt;
    };
    struct Meta { int k; };
    int ring_failed = 0;
    auto stage = [&](int jb, Meta& m, unsigned& off) {
      if ((jb % kFeedChunk) == 0) {
        const int gs = seg_base + jb / kFeedChunk;
        lds_st(feed_done + bin, gs);
        int it = 0;
        while (lds_ld(feed_ready + bin) <= gs) {
          __builtin_amdgcn_s_sleep(1);
          if (lds_ld(&wg_abort)) { ring_failed = 1; break; }
          if (++it > kWalkSpinRing) { walk_fail(&wg_abort, kWalkErrRing); ring_failed = 1; break; }
        }
      }
      const int at = (seg_base * kFeedChunk + jb + (l & (SB - 1))) % RING;     // lanes u and u + 8 hold slot u
      const int idw = ring_failed ? (kAttnPadK << kWalkKShift) : ring[at];
      m.k = (int)((unsigned)idw >> kWalkKShift);
      off = (unsigned)(idw & kWalkIdMask) * ROWB;
    };
    float4 x0[SB], x1[SB];
    Meta mc, mn;
    mc.k = mn.k = kAttnPadK;
    unsigned off_c = 0, off_n = 0;
    if (j0 < total && j0 < j1) stage(j0, mc, off_c);
    const char* q_l = reinterpret_cast<const char*>(rowsQ) + l * 16;
    for (int jb = j0; jb < j1; jb += SB) {
      if (jb < total) {
        static_for<SB>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          const unsigned o = group_bcast<L, u>(off_c) + (unsigned)(l * 16);
          x0[u] = ld4_off(a.K, o);
          x1[u] = ld4_off(a.V, o);
        });
        if (jb + SB < total && jb + SB < j1) stage(jb + SB, mn, off_n);
        // scores of the batch: <Q_k(u), K_j(u)>, reduced so that lanes u and u + 8 hold slot u's
        float part[SB];
        const unsigned my_qoff = (unsigned)(mc.k == kAttnPadK ? 0 : mc.k) * ROWB;
        static_for<SB>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          const unsigned qo = group_bcast<L, u>(my_qoff);
          const float4 qv = *reinterpret_cast<const float4*>(q_l + qo);
          part[u] = dot4(qv, x0[u]);
        });
        const float s_mine = group_dots_to_owner<L, SB>(part, l);
        static_for<SB>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          const int kt = group_bcast<L, u>(mc.k);
          const bool pad = kt == kAttnPadK;                    // group-uniform
          if (__builtin_expect(!pad && kt != k_cur, 0)) row_change(kt);
          const float su = group_bcast<L, u>(s_mine);
          const float sv = pad ? kAttnNegBig : su;
          const float mx = fmaxf(m_c, sv);
          const float sc = attn_exp(m_c - mx);
          const float p = pad ? 0.f : attn_exp(sv - mx);
          l_c = fmaf(l_c, sc, p);
          // (a select, not 0 * V: a padding slot gathers SOME valid row, and a non-finite value there must not reach this row)
          const float4 xv = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : x1[u];
          o_c.x = fmaf(o_c.x, sc, p * xv.x); o_c.y = fmaf(o_c.y, sc, p * xv.y);
          o_c.z = fmaf(o_c.z, sc, p * xv.z); o_c.w = fmaf(o_c.w, sc, p * xv.w);
          m_c = mx;
        });
        mc = mn; off_c = off_n;
      }
    }
    if (__any(ring_failed)) break;
    row_change(-1);
    finish_pending();
    if (sidx == s.steps - 1) {
      // the bin's round is over: rows wholly inside the bin are normalised and stored; the (at most two) rows it
      // shares with its neighbours leave as piece records for the merge kernels
      const int n_ch = (total + kFeedChunk - 1) / kFeedChunk;
      if (l == 0) bin_seg[bin] = seg_base + n_ch;
      lds_st(feed_done + bin, seg_base + n_ch);
      const int my_row = l < K ? s.bin_rows[tb * K + l] : -1;
      for (int k = 0; k < K; ++k) {
        const int rec = __shfl(my_row, k, L);
        if (rec == -1) continue;
        const i64 row = rec & kWalkRowMask;
        float4 o = accO[k * L + l];
        const float2 st = ml[k];
        if (rec < 0) {
          const i64 pi = tb * 2 + (k == 0 ? 0 : 1);
          reinterpret_cast<float4*>(a.p_o)[pi * L + l] = o;
          if (l == 0) {
            a.p_row[pi] = (int)(rec & 0x7fffffff);             // row | first-piece flag
            reinterpret_cast<float2*>(a.p_ml)[pi] = st;
          }
        } else {
          const float mf = fmaxf(st.x, -1e9f);                  // the reference's floor of the row maximum (graphop_kernel.cu:428)
          const float e = attn_exp(st.x - mf);
          const float lsum = st.y * e;
          const float linv = lsum > 0.f ? 1.f / lsum : 0.f;
          const float f = e * linv;
          o.x *= f; o.y *= f; o.z *= f; o.w *= f;
          reinterpret_cast<float4*>(a.o)[row * L + l] = o;
          if (l == 0 && lsum > 0.f) reinterpret_cast<float2*>(a.stats)[row] = make_float2(mf, linv);
        }
      }
    }
    if ((threadIdx.x & (kWave - 1)) == 0) lds_st(quad_done + q, gstep + 1);
    pacer.signal_slot(q, gstep + 1);
  }
  walk_report(s.err, s.launch_id, &wg_abort);
  pacer.report(s.dbg, t_start);
}

template <int L>
__global__ __launch_bounds__(kWalkThreads, 3) void k_attn_fwd_walk_f32(WalkView s, AttnWalkArgs a) {
  attn_fwd_walk_body<L>(s, a);
}

// ---- merge of the shared rows' pieces (three tiny launches over 2 * bins records) ------------------------------
// Mtmp[row] starts at kAttnNegBig, Ltmp[row] at 0, o is zero where pieces will be added (the caller's zero fill).
__global__ void k_attn_piece_max(const int* __restrict__ p_row, const float* __restrict__ p_ml, float* __restrict__ Mtmp, i64 n) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int rec = p_row[i];
  if (rec < 0) return;
  atomic_max_float(Mtmp + (rec & kWalkRowMask), p_ml[i * 2]);
}
template <int L>
__global__ __launch_bounds__(kFastBlock) void k_attn_piece_add(const int* __restrict__ p_row, const float* __restrict__ p_ml,
                                                               const float* __restrict__ p_o, const float* __restrict__ Mtmp,
                                                               float* __restrict__ Ltmp, float* __restrict__ o, i64 n) {
  const int l = threadIdx.x % L;
  const i64 i = (i64)blockIdx.x * (kFastBlock / L) + threadIdx.x / L;
  if (i >= n) return;
  const int rec = p_row[i];
  if (rec < 0) return;
  const i64 row = rec & kWalkRowMask;
  const float sc = attn_exp(p_ml[i * 2] - Mtmp[row]);
  float4 v[1];
  v[0] = reinterpret_cast<const float4*>(p_o)[i * L + l];
  v[0].x *= sc; v[0].y *= sc; v[0].z *= sc; v[0].w *= sc;
  atomic_flush_dense<L, 1>(o, row, v, l);
  if (l == 0) atomicAdd(Ltmp + row, p_ml[i * 2 + 1] * sc);
}
template <int L>
__global__ __launch_bounds__(kFastBlock) void k_attn_piece_fin(const int* __restrict__ p_row, const float* __restrict__ Mtmp,
                                                               const float* __restrict__ Ltmp, float* __restrict__ o,
                                                               float* __restrict__ stats, i64 n) {
  const int l = threadIdx.x % L;
  const i64 i = (i64)blockIdx.x * (kFastBlock / L) + threadIdx.x / L;
  if (i >= n) return;
  const int rec = p_row[i];
  if (rec < 0 || !(rec & kWalkFirstPiece)) return;       // one piece per shared row finishes it
  const i64 row = rec & kWalkRowMask;
  const float m = Mtmp[row];
  const float mf = fmaxf(m, -1e9f);
  const float e = attn_exp(m - mf);
  const float lsum = Ltmp[row] * e;
  const float linv = lsum > 0.f ? 1.f / lsum : 0.f;
  const float f = e * linv;
  float4 v = reinterpret_cast<float4*>(o)[row * L + l];
  v.x *= f; v.y *= f; v.z *= f; v.w *= f;
  reinterpret_cast<float4*>(o)[row * L + l] = v;
  if (l == 0 && lsum > 0.f) reinterpret_cast<float2*>(stats)[row] = make_float2(mf, linv);
}

}  // namespace graphop
