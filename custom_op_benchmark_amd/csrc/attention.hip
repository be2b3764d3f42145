// Fused attention step (extra op next to the reference's eight):
//   forward : o = vector_spmm(sparse_softmax(maskedmm_csr(Q, K)), V), leaving only o and the row
//             statistics (max, 1 / sum) behind -- s and a live in the caller's workspace;
//   backward: dQ, dK, dV from (Q, K, V, o, stats, dO) by two fused passes (kernels_attn.h) that
//             recompute a and ds per slot -- window-owner drivers where a sweepable plan and a table
//             beyond the L2 make column windows pay, chunk drivers otherwise; when neither applies
//             (no plan, fp64, several heads) the same result comes from the composition of the
//             unfused entry points, with the intermediates in the workspace.
// Reference composition: wrapper.py:20-30 (MaskedMMCSR), :8-18 (SparseSoftmax), :44-55 (VectorSPMM).
#include "common.h"
#include "host.h"
#include "kernels_attn.h"

namespace graphop {
namespace {

// resident workgroups per CU the fused kernels are compiled for: the column-major pass carries two
// accumulators and the statistics pipeline and needs more than 128 VGPRs; so do rows of >= 512 floats
constexpr int attn_bpc(int NV, bool col) { return col ? (NV >= 4 ? 2 : 3) : (NV >= 2 ? 3 : 4); }
// with the ids staged through LDS (4 more VGPRs, 1 KB more LDS per group) the one-row pass runs 3 per CU
constexpr int attn_bpc_staged(int NV, bool col) { return col ? (NV >= 4 ? 2 : 3) : 3; }
inline bool attn_staged(int L, int NV) { return (tuning().staged_ids & 4) && L == 16 && NV == 1; }   // d = 64: the other widths would spill in the column pass

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct AttnFast {
  SweepLaunch r, c;   // row-major pass (gathers K|V), column-major pass (gathers Q|dO)
};

SweepOpts attn_opts(int L, int NV, bool col, bool dry_run) {
  const Tuning& t = tuning();
  SweepOpts o;
  o.row_bytes = 2 * 16LL * L * NV;
  o.K = t.attn_k > 0 ? t.attn_k : (4 / NV > 0 ? 4 / NV : 1);
  o.window_scale = t.attn_window_scale > 0 ? t.attn_window_scale : 1;
  o.dry_run = dry_run ? 1 : 0;
  o.staged = attn_staged(L, NV) ? 1 : 0;
  o.no_eids = 1;          // (a, ds are recomputed per slot: neither fused pass reads eid)
  o.stage_lds_per_group = (4 * L > 128 ? 4 * L : 128) * 2 * (int)sizeof(int);   // StageCfg<L, 1>::kLdsIntsPerGroup ints
  const int cap = o.staged ? attn_bpc_staged(NV, col) : attn_bpc(NV, col);
  o.bpc = (t.attn_bpc > 0 && t.attn_bpc < cap) ? t.attn_bpc : cap;
  return o;
}

// Do the fused window-owner passes apply?  1 = yes (launch geometry in *out), 0 = no, < 0 = error.
// dry_run: only decide (and build the cached window structures), take no task queue.
int attn_fast_plan(int dtype, i64 h, i64 d, i64 n_edges, i64 n_q, i64 n_k, const graphop_plan* plan_r,
                   const graphop_plan* plan_c, hipStream_t st, bool dry_run, AttnFast* out) {
  const Tuning& t = tuning();
  if (!t.attn_fused || t.force_generic || dtype != GRAPHOP_F32 || h != 1) return 0;
  if (!plan_r || !plan_c || d % 4 != 0 || !pow2(d) || d < 16 || d > 1024 || d > t.attn_max_d) return 0;
  if (n_edges >= 0x7fffffffLL || n_q >= 0x7fffffffLL || n_k >= 0x7fffffffLL || n_edges == 0) return 0;
  if (plan_r->info.max_row >= n_q || plan_r->info.max_index >= n_k || plan_c->info.max_row >= n_k ||
      plan_c->info.max_index >= n_q)
    return 0;
  int use_r = 0, use_c = 0;
  GO_DISPATCH_LNV((int)d, {
    SweepOpts o = attn_opts(L, NV, false, dry_run);
    use_r = choose_sweep(plan_r, n_k, L, NV, st, &out->r, true, &o);
    o = attn_opts(L, NV, true, dry_run);
    if (use_r == 1) use_c = choose_sweep(plan_c, n_q, L, NV, st, &out->c, true, &o);
  });
  if (use_r < 0) return use_r;
  if (use_c < 0) return use_c;
  return (use_r == 1 && use_c == 1) ? 1 : 0;
}

// Workspace layouts: byte offsets first (the size queries pass no buffer), pointers only from a real base.
inline char* at_offset(char* base, size_t off) { return base ? base + off : nullptr; }

// The chunk-driver form of the fused passes (k_attn_bwd_rows_f32): any fp32 one-head graph with plans
// (they carry the id ranges that make the gathers safe); no window structure needed.
bool attn_rows_ok(int dtype, i64 h, i64 d, i64 n_edges, i64 n_q, i64 n_k, const graphop_plan* plan_r,
                  const graphop_plan* plan_c, hipStream_t st) {
  const Tuning& t = tuning();
  if (!t.attn_fused || !t.attn_rows || t.force_generic || dtype != GRAPHOP_F32 || h != 1) return false;
  if (!plan_r || !plan_c || d % 4 != 0 || !pow2(d) || d < 16 || d > 1024 || n_edges == 0) return false;
  if (!plan_r->indices || !plan_c->indices) return false;
  // Worth it only where the E-sized streams the fused passes avoid (~180 B per slot: two 64-B scalar
  // gather sectors, four 16-B id pairs, da / ds) outweigh packing the two operand tables
  // (16*F bytes of traffic per node, x1.5 margin).  Measured: products-shape d=16 (E/N = 25)
  // 16.0 -> 9.5 ms per step; papers100M-shape shard d=128 (E/N = 14.5: the 512-B rows dominate, the
  // step is HBM-gather-bound either way) 139 -> 152 ms, so that shape keeps the unfused passes.
  if (t.attn_rows < 0 && 180.0 * (double)n_edges <= 24.0 * (double)d * (double)(n_q + n_k)) return false;
  // The chunk-driver passes gather at the rate of wherever the tables live; where the unfused passes
  // would run on the column-window drivers (tables beyond the L2 with enough slots per window) those
  // are faster than any chunk-driver form, and this shape was excluded from the fused window passes
  // on purpose (attn_max_d): keep the unfused passes.
  if (t.attn_rows < 0) {
    int windows = 0;
    GO_DISPATCH_LNV((int)d, {
      SweepLaunch sl;
      SweepOpts o;
      o.dry_run = 1;
      o.bpc = sweep_bpc(NV, true);
      windows = choose_sweep(plan_r, n_k, L, NV, st, &sl, false, &o);
    });
    if (windows != 0) return false;
  }
  return plan_r->info.max_row < n_q && plan_r->info.max_index < n_k && plan_c->info.max_row < n_k &&
         plan_c->info.max_index < n_q;
}

template <bool COL>
int launch_attn_rows(const char* tag, const graphop_plan* plan, i64 n_chunks, int F, const float* own,
                     const float* xt, const float4* st4, float* out0, float* out1, hipStream_t st) {
  if (n_chunks == 0) return GRAPHOP_OK;
  ProfScope prof(tag, st, "k_attn_bwd_rows_f32");
  const bool owned = plan->info.rows_sorted != 0;
  GO_DISPATCH_LNV(F, {
    constexpr int GPB = kFastBlock / L;
    const i64 groups_wanted = (i64)tuning().n_cu * GPB * 8;
    i64 cpg = n_chunks / (groups_wanted > 0 ? groups_wanted : 1);
    cpg = cpg < 1 ? 1 : (cpg > 16 ? 16 : cpg);
    const unsigned nb = (unsigned)ceil_div(ceil_div(n_chunks, cpg), GPB);
    if (owned)
      hipLaunchKernelGGL((k_attn_bwd_rows_f32<L, NV, COL, true>), dim3(nb), dim3(kFastBlock), 0, st,
                         (const i64*)plan->row, (const i64*)plan->indptr, (const i64*)plan->indices, own, xt,
                         st4, out0, out1, n_chunks, (int)cpg);
    else
      hipLaunchKernelGGL((k_attn_bwd_rows_f32<L, NV, COL, false>), dim3(nb), dim3(kFastBlock), 0, st,
                         (const i64*)plan->row, (const i64*)plan->indptr, (const i64*)plan->indices, own, xt,
                         st4, out0, out1, n_chunks, (int)cpg);
  });
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

struct FastWs {
  float* kv; float* qdo; float4* st4; size_t total;
  FastWs(char* base, i64 n_q, i64 n_k, i64 F) {
    size_t off = 0;
    kv = (float*)at_offset(base, off); off += align_up(sizeof(float) * (size_t)(n_k * 2 * F));
    qdo = (float*)at_offset(base, off); off += align_up(sizeof(float) * (size_t)(n_q * 2 * F));
    st4 = (float4*)at_offset(base, off); off += align_up(sizeof(float4) * (size_t)n_q);
    total = off;
  }
};

// E-sized intermediates of the composed path: n_arrays of n_edges*h values + the softmax scratch
struct SlowWs {
  char* arr[4]; char* soft; size_t total;
  SlowWs(char* base, int n_arrays, size_t es, i64 E, i64 h, i64 soft_rows) {
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
      arr[i] = at_offset(base, off);
      if (i < n_arrays) off += align_up(es * (size_t)(E * h));
    }
    soft = at_offset(base, off); off += align_up(es * (size_t)(2 * soft_rows * h));
    total = off;
  }
};

inline i64 soft_rows_of(const graphop_plan* plan_r, i64 n_q) {
  return (plan_r && plan_r->info.row_owned) ? 0 : n_q;   // general softmax path: max / sum per row
}

template <bool COL>
int launch_attn_pass(const char* tag, const SweepLaunch& sl, int F, i64 n_gathered, const float* own,
                     const float* xt, const float4* st4, float* out0, float* out1, hipStream_t st) {
  ProfScope prof(tag, st, "k_attn_bwd_wown_f32");
  const dim3 grid(sl.blocks), block(kFastBlock);
  GO_DISPATCH_LNV(F, {
    const bool off32 = n_gathered * 2 * 16LL * L * NV < (1LL << 32);
    constexpr int BPC = attn_bpc(NV, COL);
    bool staged = false;
    if constexpr (L == 16 && NV == 1) {
      if (sl.view.rec != nullptr) {
        constexpr int BPS = attn_bpc_staged(NV, COL);
        staged = true;
        if (off32)
          hipLaunchKernelGGL((k_attn_bwd_wown_f32<L, NV, COL, true, BPS, true>), grid, block, sl.lds_bytes, st,
                             sl.view, own, xt, st4, out0, out1);
        else
          hipLaunchKernelGGL((k_attn_bwd_wown_f32<L, NV, COL, false, BPS, true>), grid, block, sl.lds_bytes, st,
                             sl.view, own, xt, st4, out0, out1);
      }
    }
    if (staged) {
    } else if (off32)
      hipLaunchKernelGGL((k_attn_bwd_wown_f32<L, NV, COL, true, BPC>), grid, block, sl.lds_bytes, st,
                         sl.view, own, xt, st4, out0, out1);
    else
      hipLaunchKernelGGL((k_attn_bwd_wown_f32<L, NV, COL, false, BPC>), grid, block, sl.lds_bytes, st,
                         sl.view, own, xt, st4, out0, out1);
  });
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

}  // namespace

int attn_prepare_plan(const graphop_plan* plan, i64 n_table_rows, i64 d, bool col, hipStream_t st) {
  const Tuning& t = tuning();
  if (!t.attn_fused || t.force_generic || !plan || d % 4 != 0 || !pow2(d) || d < 16 || d > 1024 || d > t.attn_max_d) return 0;
  int use = 0;
  GO_DISPATCH_LNV((int)d, {
    SweepLaunch sl;
    SweepOpts o = attn_opts(L, NV, col, true);
    use = choose_sweep(plan, n_table_rows, L, NV, st, &sl, true, &o);
  });
  return use;
}
}  // namespace graphop

using namespace graphop;

extern "C" {

int graphop_attention_workspace_bytes(int dtype, int backward, int64_t n_edges, int64_t n_q,
                                      int64_t n_k, int64_t h, int64_t d,
                                      const graphop_plan_t* plan_r, const graphop_plan_t* plan_c,
                                      void* stream, int64_t* bytes_out) {
  const char* fn = "attention_workspace_bytes";
  GO_CHECK_ARG(bytes_out != nullptr, "%s: bytes_out is NULL", fn);
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_edges >= 0 && n_q >= 0 && n_k >= 0 && h >= 1 && d >= 0, "%s: negative size", fn);
  const size_t es = esize(dtype);
  if (!backward) {
    size_t need = 0;
    const int fw = (plan_r && n_edges * h > 0)
                       ? attn_fwd_walk(plan_r, n_q, n_k, h, d, dtype, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                       (hipStream_t)stream, /*dry_run=*/true, &need)
                       : 0;
    if (fw < 0) return -fw;
    // the one-pass forward only needs the piece records of shared rows; the composed forward keeps s and a here
    *bytes_out = fw == 1 ? (int64_t)need : (int64_t)SlowWs(nullptr, 2, es, n_edges, h, soft_rows_of(plan_r, n_q)).total;
    return GRAPHOP_OK;
  }
  AttnFast af;
  const int fast = attn_fast_plan(dtype, h, d, n_edges, n_q, n_k, plan_r, plan_c, (hipStream_t)stream,
                                  /*dry_run=*/true, &af);
  if (fast < 0) return -fast;
  if (fast == 1 || attn_rows_ok(dtype, h, d, n_edges, n_q, n_k, plan_r, plan_c, (hipStream_t)stream))
    *bytes_out = (int64_t)FastWs(nullptr, n_q, n_k, h * d).total;
  else *bytes_out = (int64_t)SlowWs(nullptr, 4, es, n_edges, h, soft_rows_of(plan_r, n_q)).total;
  return GRAPHOP_OK;
}

int graphop_attention_backward_is_fused(int dtype, int64_t n_edges, int64_t n_q, int64_t n_k, int64_t h,
                                        int64_t d, const graphop_plan_t* plan_r,
                                        const graphop_plan_t* plan_c, void* stream, int* fused_out) {
  GO_CHECK_ARG(fused_out != nullptr, "attention_backward_is_fused: fused_out is NULL");
  AttnFast af;
  const int fast = attn_fast_plan(dtype, h, d, n_edges, n_q, n_k, plan_r, plan_c, (hipStream_t)stream,
                                  /*dry_run=*/true, &af);
  if (fast < 0) return -fast;
  *fused_out = (fast == 1 || attn_rows_ok(dtype, h, d, n_edges, n_q, n_k, plan_r, plan_c, (hipStream_t)stream)) ? 1 : 0;
  return GRAPHOP_OK;
}

int graphop_attention_forward(int dtype, const int64_t* row, const int64_t* indptr, const int64_t* eid,
                              const int64_t* indices, const void* Q, const void* K, const void* V,
                              void* o, void* stats, int64_t n_chunks, int64_t n_edges, int64_t n_q,
                              int64_t n_k, int64_t h, int64_t d, void* workspace,
                              int64_t workspace_bytes, const graphop_plan_t* plan, void* stream) {
  const char* fn = "attention_forward";
  GO_TRY(check_async_error(false));
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_chunks >= 0 && n_edges >= 0 && n_q >= 0 && n_k >= 0 && h >= 1 && d >= 0,
               "%s: negative size", fn);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = esize(dtype);
  if (n_q * h > 0) {
    GO_PTR(fn, stats);
    GO_HIP(zero_async(stats, es * (size_t)(n_q * h * 2), st));   // rows without edges: (0, 0)
  }
  const graphop_plan* pm = plan_matches_full(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                                             (const i64*)indices, n_chunks, n_edges) ? plan : nullptr;
  // ONE walk-style pass where it applies (kernels_attn_walk.h): s and a never leave the chip; its workspace (piece
  // records of the rows that bins share) is what graphop_attention_workspace_bytes reported for this call
  if (pm && n_edges * h > 0) {
    GO_PTR(fn, Q); GO_PTR(fn, K); GO_PTR(fn, V); GO_PTR(fn, o);
    size_t need = 0;
    const int ok = attn_fwd_walk(pm, n_q, n_k, h, d, dtype, Q, K, V, o, stats, nullptr, 0, st, /*dry_run=*/true, &need);
    if (ok < 0) return -ok;
    if (ok == 1) {
      GO_CHECK_ARG(workspace != nullptr && (size_t)workspace_bytes >= need,
                   "%s: workspace of %lld bytes needed (graphop_attention_workspace_bytes), got %lld", fn,
                   (long long)need, (long long)workspace_bytes);
      const int fw = attn_fwd_walk(pm, n_q, n_k, h, d, dtype, Q, K, V, o, stats, workspace, workspace_bytes, st,
                                   /*dry_run=*/false, nullptr);
      if (fw < 0) return -fw;
      if (fw == 1) return GRAPHOP_OK;
    }
  }
  const i64 soft_rows = soft_rows_of(pm, n_q);
  SlowWs ws((char*)workspace, 2, es, n_edges, h, soft_rows);
  GO_CHECK_ARG(n_edges * h == 0 || (workspace != nullptr && (size_t)workspace_bytes >= ws.total),
               "%s: workspace of %lld bytes needed (graphop_attention_workspace_bytes), got %lld", fn,
               (long long)ws.total, (long long)workspace_bytes);
  void* s = ws.arr[0];
  void* a = ws.arr[1];
  GO_TRY(graphop_maskedmm_csr_forward(dtype, row, indptr, eid, indices, Q, K, s, n_chunks, n_edges, n_q,
                                      n_k, h, d, pm, stream));
  if (n_edges * h > 0)
    GO_TRY(softmax_forward_stats(dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid, s, a,
                                 n_chunks, n_edges, h, soft_rows ? ws.soft : nullptr, soft_rows, pm, st,
                                 stats));
  return graphop_vector_spmm_forward(dtype, row, indptr, eid, indices, a, V, o, n_chunks, n_edges, n_k,
                                     n_q, h, d, pm, stream);
}

int graphop_attention_backward(int dtype, const int64_t* row, const int64_t* indptr_r,
                               const int64_t* eid_r, const int64_t* indices_r, const int64_t* col,
                               const int64_t* indptr_c, const int64_t* eid_c, const int64_t* indices_c,
                               const void* Q, const void* K, const void* V, const void* o,
                               const void* stats, const void* dO, void* dQ, void* dK, void* dV,
                               int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_edges,
                               int64_t n_q, int64_t n_k, int64_t h, int64_t d, void* workspace,
                               int64_t workspace_bytes, const graphop_plan_t* plan_r,
                               const graphop_plan_t* plan_c, void* stream) {
  const char* fn = "attention_backward";
  GO_TRY(check_async_error(false));
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_row_chunks >= 0 && n_col_chunks >= 0 && n_edges >= 0 && n_q >= 0 && n_k >= 0 &&
               h >= 1 && d >= 0, "%s: negative size", fn);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = esize(dtype);
  const graphop_plan* pr = plan_matches_full(plan_r, (const i64*)row, (const i64*)indptr_r, (const i64*)eid_r,
                                             (const i64*)indices_r, n_row_chunks, n_edges) ? plan_r : nullptr;
  const graphop_plan* pc = plan_matches_full(plan_c, (const i64*)col, (const i64*)indptr_c, (const i64*)eid_c,
                                             (const i64*)indices_c, n_col_chunks, n_edges) ? plan_c : nullptr;
  AttnFast af;
  const int fast = attn_fast_plan(dtype, h, d, n_edges, n_q, n_k, pr, pc, st, /*dry_run=*/false, &af);
  if (fast < 0) return -fast;
  const bool rows_path = fast != 1 && attn_rows_ok(dtype, h, d, n_edges, n_q, n_k, pr, pc, st);
  if (fast == 1 || rows_path) {
    const i64 F = d;   // h == 1
    FastWs ws((char*)workspace, n_q, n_k, F);
    GO_CHECK_ARG(workspace != nullptr && (size_t)workspace_bytes >= ws.total,
                 "%s: workspace of %lld bytes needed (graphop_attention_workspace_bytes), got %lld", fn,
                 (long long)ws.total, (long long)workspace_bytes);
    GO_PTR(fn, Q); GO_PTR(fn, K); GO_PTR(fn, V); GO_PTR(fn, o); GO_PTR(fn, stats); GO_PTR(fn, dO);
    GO_PTR(fn, dQ); GO_PTR(fn, dK); GO_PTR(fn, dV);
    GO_HIP(zero_async(dQ, sizeof(float) * (size_t)(n_q * F), st));
    GO_HIP(zero_async(dK, sizeof(float) * (size_t)(n_k * F), st));
    GO_HIP(zero_async(dV, sizeof(float) * (size_t)(n_k * F), st));
    {
      ProfScope prof("attn_pack", st, "k_attn_pack");
      GO_DISPATCH_LNV((int)F, {
        constexpr int RPB = kFastBlock / L;
        hipLaunchKernelGGL((k_attn_pack<L, NV, false>), dim3((unsigned)ceil_div(n_k, RPB)), dim3(kFastBlock),
                           0, st, (const float*)K, (const float*)V, ws.kv, n_k, (const float*)nullptr,
                           (const float*)nullptr, (float4*)nullptr);
        hipLaunchKernelGGL((k_attn_pack<L, NV, true>), dim3((unsigned)ceil_div(n_q, RPB)), dim3(kFastBlock),
                           0, st, (const float*)Q, (const float*)dO, ws.qdo, n_q, (const float*)o,
                           (const float*)stats, ws.st4);
      });
      GO_LAUNCH_CHECK();
    }
    if (rows_path) {
      GO_TRY(launch_attn_rows<false>("attn_rows_row", pr, n_row_chunks, (int)F, ws.qdo, ws.kv, ws.st4, (float*)dQ,
                                     nullptr, st));
      GO_TRY(launch_attn_rows<true>("attn_rows_col", pc, n_col_chunks, (int)F, ws.kv, ws.qdo, ws.st4, (float*)dK,
                                    (float*)dV, st));
      return GRAPHOP_OK;
    }
    GO_TRY(launch_attn_pass<false>("attn_bwd_row", af.r, (int)F, n_k, ws.qdo, ws.kv, ws.st4, (float*)dQ,
                                   nullptr, st));
    GO_TRY(launch_attn_pass<true>("attn_bwd_col", af.c, (int)F, n_q, ws.kv, ws.qdo, ws.st4, (float*)dK,
                                  (float*)dV, st));
    return GRAPHOP_OK;
  }
  // composition of the unfused entry points: recompute s and a, then the three backward ops
  const i64 soft_rows = soft_rows_of(pr, n_q);
  SlowWs ws((char*)workspace, 4, es, n_edges, h, soft_rows);
  GO_CHECK_ARG(n_edges * h == 0 || (workspace != nullptr && (size_t)workspace_bytes >= ws.total),
               "%s: workspace of %lld bytes needed (graphop_attention_workspace_bytes), got %lld", fn,
               (long long)ws.total, (long long)workspace_bytes);
  void* s = ws.arr[0];
  void* a = ws.arr[1];
  void* da = ws.arr[2];
  void* ds = ws.arr[3];
  GO_TRY(graphop_maskedmm_csr_forward(dtype, row, indptr_r, eid_r, indices_r, Q, K, s, n_row_chunks,
                                      n_edges, n_q, n_k, h, d, pr, stream));
  GO_TRY(graphop_sparse_softmax_forward(dtype, row, indptr_r, eid_r, s, a, n_row_chunks, n_edges, h,
                                        soft_rows ? ws.soft : nullptr, soft_rows, pr, stream));
  GO_TRY(graphop_vector_spmm_backward(dtype, row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c,
                                      indices_c, a, dO, V, da, dV, n_row_chunks, n_col_chunks, n_edges,
                                      n_k, n_q, h, d, pr, pc, stream));
  GO_TRY(graphop_sparse_softmax_backward(dtype, row, indptr_r, eid_r, a, da, ds, n_row_chunks, n_edges, h,
                                         soft_rows ? ws.soft : nullptr, soft_rows, pr, stream));
  return graphop_maskedmm_csr_backward(dtype, row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c,
                                       indices_c, Q, K, ds, dQ, dK, n_row_chunks, n_col_chunks, n_edges,
                                       n_q, n_k, h, d, pr, pc, stream);
}

}  // extern "C"
