// libgraphop_hip: the OPERATOR entry points of the C ABI (include/graphop_hip.h) over the gfx950 kernels.
// Host-side dispatch only: validates sizes, zero-fills outputs (the reference returns at::zeros
// tensors, graphop_kernel.cu:284,379-380,429,482,527,571-572) and picks the driver of every pass.
// (Runtime services -- errors, allocator hook, timing, knobs, zero fill, the device error record -- are in runtime.hip,
// the plan entry points in plan_api.hip: split in round 5.)
#include <stdarg.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "host.h"
#include "kernels_fast.h"
#include "kernels_walk.h"
#include "kernels_attn_walk.h"
#include "kernels_block.h"
#include "kernels_attn.h"
#include "kernels_generic.h"

namespace graphop {

namespace {

// Chunks per lane group of the chunk drivers: the tuned value amortises row switches on big graphs,
// but a small graph (Cora-shape: 2.8 K chunks) must still put a few lane groups on every CU --
// 16 chunks per group there left 13 workgroups walking chunks serially (33 us per pass).
inline int cpg_for(i64 n_chunks, int cpg_max, int F) {
  const int lanes = F >= 256 ? 64 : (F / 4 > 0 ? F / 4 : 1);
  const i64 groups_wanted = (i64)tuning().n_cu * (kFastBlock / lanes) * 8;
  i64 c = n_chunks / (groups_wanted > 0 ? groups_wanted : 1);
  if (c < 1) c = 1;
  return (int)(c < cpg_max ? c : cpg_max);
}

inline unsigned blocks_for(i64 work, i64 per_block) {
  i64 b = ceil_div(work > 0 ? work : 1, per_block);
  return (unsigned)b;
}

// fast fp32 path applies when a node row is 16..1024 floats (power of two), ids fit 32 bits
inline bool fast_ok(int dtype, i64 h, i64 d, i64 n_edges, i64 n_src_rows) {
  if (tuning().force_generic || dtype != GRAPHOP_F32) return false;
  const i64 F = h * d;
  if (d % 4 != 0 || !pow2(F) || F < 16 || F > 1024 || !pow2(h)) return false;
  if (n_edges >= 0x7fffffffLL || n_src_rows >= 0x7fffffffLL) return false;
  return true;
}

// fp64 on the plan-driven kernels (kernels_walk.h / kernels_strip.h are written against 16-byte row pieces): one head,
// rows of 256 B, 512 B or 1 KB -- the lane-group shapes of fp32 d = 64 / 128 / 256
inline bool fast64_ok(int dtype, i64 h, i64 d, i64 n_edges, i64 n_src_rows) {
  if (tuning().force_generic || dtype != GRAPHOP_F64 || h != 1) return false;
  if (d != 32 && d != 64 && d != 128) return false;
  return n_edges < 0x7fffffffLL && n_src_rows < 0x7fffffffLL;
}
}  // namespace

int choose_sweep(const graphop_plan* plan, i64 n_table_rows, int L, int NV, hipStream_t st,
                 SweepLaunch* out, bool accumulating, const SweepOpts* opts) {
  const Tuning& t = tuning();
  if (!t.sweep || !plan) return 0;
  const graphop_plan_info_t& pi = plan->info;
  if (!pi.row_owned || !plan->sorted_in_rows || !pi.has_idx32 || !plan->indices) return 0;   // (has_idx32: mirrors exist or can be rebuilt)
  if (pi.n_segments == 0 || pi.n_edges == 0) return 0;
  const i64 row_bytes = (opts && opts->row_bytes > 0) ? opts->row_bytes : 16LL * L * NV;
  const i64 table_bytes = n_table_rows * row_bytes;
  if (table_bytes < (i64)t.sweep_min_kb * 1024) return 0;
  // Two tiers.  L2-sized windows (<= 4 MB) while a (row, window) granule still holds a few slots;
  // otherwise, for tables that do not fit the 256 MiB Infinity Cache next to the streams, 32 MB
  // windows that stay Infinity-Cache resident (HBM-rate random row gathers become Infinity-Cache-
  // rate ones).  Anything else stays on the chunk drivers.
  const i64 mean_row = pi.n_edges / pi.n_segments;
  auto windows_ok = [&](i64 w) {
    return w >= 2 && w <= t.max_windows && mean_row >= (i64)t.sweep_min_granule * w;
  };
  // A window-owner SpMM over identity-eid slots flushes one partial row per (vrow, window): it is
  // faster with windows of twice the L2 size (half as many flushes, misses served by the Infinity
  // Cache) and vrows twice as long -- measured on Reddit-shape: 2.25 ms at W=8 vs 2.46 at W=16.
  int coarse = (accumulating && pi.eid_identity && t.spmm_window_scale > 1) ? t.spmm_window_scale : 1;
  // the fused kernels state their window size directly (their packed rows are twice as wide)
  const i64 win_kb = (opts && opts->window_scale > 0) ? (i64)t.window_kb * opts->window_scale : (i64)t.window_kb;
  if (opts && opts->window_scale > 0) coarse = 1;
  i64 W = pow2ceil(ceil_div(table_bytes, win_kb * 1024));
  if (t.sweep_w > 0) W = t.sweep_w;   // experiments / tests: window count given directly
  if (!windows_ok(W)) {
    coarse = 1;   // Infinity-Cache tier: windows are sized for that cache, not for the flush count
    W = table_bytes > (128LL << 20) ? pow2ceil(ceil_div(table_bytes, (i64)t.mall_window_kb * 1024)) : 0;
    if (!windows_ok(W)) return 0;
  } else if (t.sweep_w <= 0 && coarse > 1) {
    const i64 Wc = pow2ceil(ceil_div(table_bytes, (i64)t.window_kb * 1024 * coarse));
    if (Wc >= 2) W = Wc; else coarse = 1;
  } else if (t.sweep_w <= 0 && !accumulating && !(opts && opts->window_scale > 0) && row_bytes >= 1024) {
    // 1-KB rows: 4 MB windows leave a (row, window) granule with half a batch of slots and one A-row fetch per
    // 8 gathered rows; windows of twice the L2 size measure 7 % faster (Reddit-shape d = 256: 8.08 -> 7.49 ms)
    while (W > 2 && mean_row < 12 * W) W >>= 1;
  }
  if (W < 2 || W > t.max_windows) return 0;
  const i64 win_cols = ceil_div(n_table_rows, W);
  int K = t.sweep_k > 0 ? t.sweep_k : (8 / NV > 0 ? 8 / NV : 1);
  if (opts && opts->K > 0) K = opts->K;
  // passes that gather their per-slot scalars through eid (the column-major ones) run with half
  // the vrows per lane group: the columns in flight on an XCD then span half as many ids, and more
  // of the scalar lines they share are still in L2 (Reddit-shape: 3.27 -> 3.06 ms per pass)
  if (t.sweep_k <= 0 && !(opts && opts->K > 0) && !pi.eid_identity && K > 1) K /= 2;
  if (K > L) K = L;
  const int gpb = kFastBlock / L;
  const int bpc_req = (opts && opts->bpc > 0) ? (opts->bpc < t.sweep_bpc || opts->window_scale > 0 ? opts->bpc : t.sweep_bpc)
                                              : t.sweep_bpc;
  const int bpc = bpc_req < 1 ? 1 : (bpc_req > kSweepBlocksPerCu ? kSweepBlocksPerCu : bpc_req);
  int T = t.vrow_t;
  if (T <= 0) {
    // vrow length cap: long rows are cut so that one round of the resident grid gets a vrow per
    // (group, K) slot -- about E / (resident groups * K) slots each -- and never below the mean row
    const i64 resident_vrows = (i64)t.n_cu * bpc * gpb * K;
    i64 target = pi.n_edges / (resident_vrows > 0 ? resident_vrows : 1);
    i64 p2 = 64;
    while (p2 * 2 <= target && p2 < 4096) p2 <<= 1;
    T = (int)(p2 * coarse * ((opts && opts->window_scale > 0) ? opts->window_scale : 1));
  }
  const Sweep* sw = nullptr;
  const int rc = plan_get_sweep(const_cast<graphop_plan*>(plan), (int)W, win_cols, T, st, &sw);
  if (rc != GRAPHOP_OK) return -rc;
  if (!sw) return 0;   // the plan already holds its share of window geometries: chunk drivers for this one
  // a resident grid of waves pulling (window, vrow tile) tasks from the eight per-XCD queue heads
  // (sync[y * 64], zeroed here)
  const int tile = (kWave / L) * K;
  const i64 tiles = ceil_div(sw->V, tile);
  if (tiles * ceil_div((i64)sw->W, 8) >= 0x7fffffffLL) return 0;
  out->view = SweepView{};
  const bool dry = opts && opts->dry_run;
  out->view.sync = dry ? nullptr : plan_take_queue(const_cast<graphop_plan*>(plan), sw);
  out->view.V = sw->V;
  out->view.W = sw->W;
  out->view.K = K;
  out->view.win_bytes = win_cols * row_bytes;
  out->view.table_bytes = table_bytes;
  out->view.touch = opts ? opts->touch : 0;
  if (!dry && zero_async(out->view.sync, sizeof(int) * kQueueInts, st) != hipSuccess) return -GRAPHOP_ERR_HIP;
  i64 nb = (i64)t.n_cu * bpc;
  const i64 need = ceil_div(tiles * sw->W, (i64)(kFastBlock / kWave));
  if (nb > need) nb = need;
  out->blocks = (unsigned)(nb < 1 ? 1 : nb);
  out->lds_bytes = (size_t)gpb * K * row_bytes;
  bool staged_ok = false;
  if (opts && opts->staged) {
    const Sweep::Dealt* dl = nullptr;
    const int rcd = plan_get_dealt(const_cast<graphop_plan*>(plan), sw, L, K, st, &dl, /*want_eids=*/!opts->no_eids);
    if (rcd != GRAPHOP_OK) return -rcd;
    if (dl) {
      out->view.rec = (const int4*)dl->rec;
      out->view.ids_w = dl->ids;
      out->view.eids_w = dl->eids;
      out->lds_bytes += (size_t)gpb * opts->stage_lds_per_group;
      staged_ok = pi.eid_identity || dl->eids != nullptr || opts->no_eids;   // the staged kernels read the dealt copies and nothing else
    }
  }
  if (!staged_ok) {
    // a per-batch kernel will read the window tables and the 32-bit mirrors at run time: resident, and kept (plan.hip)
    const int rcp = plan_pin_sweep_tables(const_cast<graphop_plan*>(plan), sw, st);
    if (rcp != GRAPHOP_OK) return -rcp;
  }
  out->view.wp_lo = sw->wp_lo;
  out->view.wp_hi = sw->wp_hi;
  out->view.vr_row = sw->vr_row;
  out->view.idx32 = plan->idx32;
  out->view.eid32 = plan->eid32;
  plan_trim(const_cast<graphop_plan*>(plan));   // what only the builders read goes (nothing a pinned reader needs)
  out->view.wp_lo = sw->wp_lo;                  // (staged launches: possibly NULL now -- they read rec / ids_w / eids_w only)
  out->view.wp_hi = sw->wp_hi;
  out->view.idx32 = plan->idx32;
  out->view.eid32 = plan->eid32;
  return 1;
}

namespace {

// Which window-owner passes take the dealt layout + staged ids (also what graphop_plan_prepare builds).
inline bool table_off32(i64 n_table_rows, int L, int NV) { return n_table_rows * 16LL * L * NV < (1LL << 32); }
// several heads: one float4 per lane, a head = 4 / 8 / 16 / 32 lanes (kernels_fast.h: sddmm_strip_staged_heads)
inline bool sddmm_staged_heads(int L, int NV, i64 h, i64 n_edges) {
  if (h < 2 || NV != 1 || L < 16 || L % h != 0 || n_edges * h >= 0x7fffffffLL) return false;
  const i64 d4 = L / h;
  return d4 == 4 || d4 == 8 || d4 == 16 || d4 == 32;
}
inline bool sddmm_staged(const graphop_plan* plan, int L, int NV, i64 h, i64 n_table_rows) {
  // one head: tables of 4 GiB and more too (64-bit row offsets, kernels_strip.h); several heads: 32-bit offsets only
  return (tuning().staged_ids & 1) && plan->info.eid_identity &&
         (h == 1 || (sddmm_staged_heads(L, NV, h, plan->info.n_edges) && table_off32(n_table_rows, L, NV)));
}
inline bool spmm_staged(int L, int NV, i64 h, i64 n_table_rows) {
  return (tuning().staged_ids & 2) && h == 1 && table_off32(n_table_rows, L, NV);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ---- walk drivers (kernels_walk.h) -----------------------------------------------------------------
// The walk kernels take nearly all of a CU's LDS as dynamic shared memory: the limit is raised once per kernel function.
inline void allow_full_lds(const void* fn) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;   // (per device: function attributes are)
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert({dev, fn}).second)
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}
struct WalkLaunch {
  WalkView view;
  unsigned blocks;
  size_t lds_bytes;
};

// Diagnostics (knob walk_debug): per-wave cycle counts of one walk launch, summarised on stderr.
struct WalkDebug {
  long long* buf = nullptr;
  unsigned blocks = 0;
  const char* tag = "";
  hipStream_t st = nullptr;
  int rounds = 0, W = 0, waves_per_wg = 4;
  void arm(WalkLaunch* wl, const char* t, hipStream_t s, int wpw = 4) {
    waves_per_wg = wpw;
    if (!tuning().walk_debug) return;
    blocks = wl->blocks; tag = t; st = s; rounds = wl->view.rounds; W = wl->view.steps;
    if (hipMalloc((void**)&buf, sizeof(long long) * 34 * blocks) != hipSuccess) { buf = nullptr; return; }
    (void)hipMemsetAsync(buf, 0, sizeof(long long) * 34 * blocks, s);
    wl->view.dbg = buf;
  }
  ~WalkDebug() {
    if (!buf) return;
    std::vector<long long> h((size_t)34 * blocks);
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(h.data(), buf, sizeof(long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      double tot = 0, wait = 0, nw = 0, feed = 0, ftot = 0, fspace = 0; long long tmax = 0, tmin = 1LL << 62; int gave = 0;
      for (size_t b = 0; b < blocks; ++b) { ftot += h[(size_t)32 * blocks + 2 * b]; fspace += h[(size_t)32 * blocks + 2 * b + 1]; }
      double xt[8] = {0}, xw[8] = {0}; int xn[8] = {0};
      const size_t n = (size_t)waves_per_wg * blocks;
      for (size_t i = 0; i < n; ++i) {
        const long long* d = &h[i * 4];
        tot += d[0]; wait += d[1]; nw += d[2] & 0xffff; feed += (double)(d[2] >> 16);
        tmax = d[0] > tmax ? d[0] : tmax; tmin = d[0] < tmin ? d[0] : tmin;
        gave += (d[3] & 16) ? 1 : 0;
        const int x = (int)(d[3] & 7); xt[x] += d[0]; xw[x] += d[1]; xn[x]++;
      }
      fprintf(stderr, "[walk] %s rounds=%d steps=%d waves=%zu cycles mean %.0f min %lld max %lld | pacer wait %.1f %% of wave time, %.1f waits/wave, %d gave up | per XCD wait%%:",
              tag, rounds, W, n, tot / n, tmin, tmax, 100.0 * wait / (tot > 0 ? tot : 1), nw / n, gave);
      for (int x = 0; x < 8; ++x) fprintf(stderr, " %.0f(%d)", xn[x] ? 100.0 * xw[x] / (xt[x] > 0 ? xt[x] : 1) : 0.0, xn[x]);
      fprintf(stderr, " | workers wait for the feeder %.1f %%; feeder: %.0f cycles, %.1f %% waiting for ring space", 100.0 * feed / (tot > 0 ? tot : 1),
              ftot / (blocks ? blocks : 1), 100.0 * fspace / (ftot > 0 ? ftot : 1));
      fprintf(stderr, "\n");
    }
    (void)hipFree(buf);
  }
};

// Decide whether the walk drivers apply to a pass over `plan` that gathers rows of 16*L*NV bytes from a
// table of n_table_rows rows, and fetch / build the layout.  1 = use it, 0 = no, < 0 = error (negated).
// `K` = rows per lane group the kernel's LDS holds.
template <int L, int NV>
int choose_walk(const graphop_plan* plan, i64 n_table_rows, int K, hipStream_t st, WalkLaunch* out, bool dry_run = false,
                int tables = 1) {   // tables: gathered tables that must share an L2 window (the fused attention pass: K and V)
  if constexpr (NV != 1 || L < 16) {
    return 0;   // wider rows: kWalkK of them per lane group do not fit the LDS
  } else {
    const Tuning& t = tuning();
    if (!plan) return 0;
    const graphop_plan_info_t& pi = plan->info;
    if (!pi.row_owned || !plan->sorted_in_rows || !pi.has_idx32 || !plan->indices) return 0;
    if (pi.n_segments == 0 || pi.n_edges == 0) return 0;
    if (pi.max_index >= (1LL << kWalkKShift) || n_table_rows > (1LL << kWalkKShift)) return 0;   // ids share a word with the row-in-bin
    const i64 table_bytes = n_table_rows * 16LL * L * NV;
    if (table_bytes < (i64)t.sweep_min_kb * 1024) return 0;
    // column-major passes: 2 MB windows at 256-B rows (the scalar lines share the L2), 4 MB from 1-KB rows on
    // (Reddit-shape d = 256: 9.42 -> 8.83 ms; d = 128 is best at 2 MB)
    const i64 win_kb = pi.eid_identity ? t.walk_window_kb : (16LL * L * NV >= 1024 ? 2 * (i64)t.walk_window_kb_col : t.walk_window_kb_col);
    i64 W = t.sweep_w > 0 ? t.sweep_w : pow2ceil(ceil_div(table_bytes * tables, (win_kb > 0 ? win_kb : 1) * 1024));
    if (W < 2) W = 2;
    if (W > t.max_windows || W > 512) return 0;
    const i64 mean_row = pi.n_edges / pi.n_segments;
    if (mean_row < (i64)t.sweep_min_granule * W / 2) return 0;
    const int GPB = kWalkWorkers / L;
    i64 blocks = t.walk_blocks > 0 ? ceil_div((i64)t.walk_blocks * kFastBlock, kWalkWorkers) : (i64)t.n_cu;   // one workgroup per CU
    int slots = 8;
    if (blocks >= 8) blocks -= blocks % 8; else slots = 1;
    const i64 groups = blocks * GPB;
    if (pi.n_edges < groups * (i64)t.walk_min_bin) return 0;
    if (K < 1 || K > kWalkK) return 0;
    const Walk* wk = nullptr;
    const int rc = plan_get_walk(const_cast<graphop_plan*>(plan), (int)W, ceil_div(n_table_rows, W), (int)groups, /*lane groups per bin=*/1, K,
                                 slots, st, &wk, /*want_widx=*/tables == 1);
    if (rc != GRAPHOP_OK) return -rc;
    plan_trim(const_cast<graphop_plan*>(plan));   // the walk kernels read their own layout: the mirrors were builder input
    if (!wk) return 0;
    out->view.ids = wk->ids; out->view.widx = wk->widx; out->view.bin_pos = wk->bin_pos;
    out->view.bin_rows = wk->bin_rows; out->view.bin_cum = wk->bin_cum;
    out->view.W = wk->W; out->view.groups = wk->groups; out->view.rounds = wk->rounds;
    out->view.K = wk->K;
    out->view.xcd_slots = slots;
    {
      i64 steps = (i64)wk->W * (t.walk_steps > 0 ? t.walk_steps : 1);
      if (steps > wk->max_steps) steps = wk->max_steps;
      if (steps > wk->longest_run / 16) steps = wk->longest_run / 16;   // at least a batch per step
      out->view.steps = (int)(steps < 1 ? 1 : steps);
    }
    out->view.drift = t.walk_drift;
    out->view.sync = (t.walk_drift > 0 && !dry_run) ? plan_take_walk_sync(const_cast<graphop_plan*>(plan), wk) : nullptr;
    out->view.dbg = nullptr;
    out->view.err = device_error_word(false);   // (created with the plan's first walk layout, never during a capture)
    out->view.launch_id = 0;                     // (set by the launcher: walk_launch_id(tag))
    out->view.fault = t.walk_fault;
    out->view.stream_weights = pi.eid_identity ? 1 : 0;
    out->blocks = (unsigned)blocks;
    out->lds_bytes = 0;   // (set by the launcher: depends on the heads)
    if (!dry_run && out->view.sync &&
        zero_async(out->view.sync, sizeof(int) * 64 * (size_t)(8 + 16 * (i64)wk->rounds * out->view.steps), st) != hipSuccess)
      return -GRAPHOP_ERR_HIP;
    return 1;
  }
}

// Rows per lane group of the SpMM-type walk kernel for `h` heads (0 = the pass does not take the walk): what the
// LDS holds next to the per-head weight rings, at least 8 (fewer rows = more rounds = more table streams).
template <int L, int NV>
inline int spmm_walk_k(i64 h, int dtype = GRAPHOP_F32) {
  if (h != 1 && h != 2 && h != 4 && h != 8) return 0;
  if (dtype == GRAPHOP_F64 && h != 1) return 0;                                    // fp64: one head
  if (h > 1 && (NV != 1 || (4 * L) % h != 0 || (4 * L / h) % 4 != 0)) return 0;   // a lane's float4 lies inside one head
  const int k = spmm_walk_rows<L, NV>((int)h, dtype == GRAPHOP_F64 ? 2 : 1);
  return k >= 8 ? k : 0;
}

template <int L, int NV>
int try_spmm_walk(const char* tag, int dtype, const graphop_plan* plan, i64 n_table_rows, const void* w, const void* X,
                  void* out, i64 h, int d4, hipStream_t st) {
  if constexpr (NV != 1 || L < 16) {
    return 0;
  } else {
    if (!plan) return 0;
    if (!(tuning().walk & (plan->info.eid_identity ? 2 : 4))) return 0;
    const int K = spmm_walk_k<L, NV>(h, dtype);
    if (K == 0 || (h > 1 && !aligned16(w))) return 0;
    const bool off32 = table_off32(n_table_rows, L, NV);
    if (h > 1 && !off32) return 0;   // several heads: 32-bit row offsets only
    WalkLaunch wl;
    const int use = choose_walk<L, NV>(plan, n_table_rows, K, st, &wl);
    if (use != 1) return use;
    WalkDebug dbg;
    dbg.arm(&wl, tag, st, kWalkWorkers / kWave);
    wl.view.launch_id = walk_launch_id(tag);
    ProfScope prof(tag, st, dtype == GRAPHOP_F64 ? "k_spmm_walk_f64" : "k_spmm_walk_f32");
    auto launch = [&](auto kfn, size_t lds_bytes, auto... args) {
      allow_full_lds((const void*)kfn);
      if (tuning().walk_debug) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kfn, kWalkThreads, lds_bytes);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void*)kfn);
        fprintf(stderr, "[walk] occupancy API: %d workgroups of %d threads per CU (dynamic LDS %zu, static %zu, regs %d, K %d)\n", nb,
                kWalkThreads, lds_bytes, (size_t)fa.sharedSizeBytes, fa.numRegs, K);
      }
      hipLaunchKernelGGL(kfn, dim3(wl.blocks), dim3(kWalkThreads), lds_bytes, st, wl.view, args...);
    };
    if (dtype == GRAPHOP_F64) {
      if constexpr (spmm_walk_rows<L, NV>(1, 2) >= 8) {
        const size_t lds_bytes = spmm_walk_lds_bytes<L, NV>(K, 1, 2);
        if (off32) launch(k_spmm_walk_f64<L, NV, true>, lds_bytes, (const double*)w, (const double*)X, (double*)out);
        else launch(k_spmm_walk_f64<L, NV, false>, lds_bytes, (const double*)w, (const double*)X, (double*)out);
      }
      return 1;
    }
    auto go = [&](auto hc) {
      constexpr int HV = decltype(hc)::value;
      if constexpr (spmm_walk_rows<L, NV>(HV) < 8) return;    // (spmm_walk_k said no: never instantiated)
      else {
        const size_t lds_bytes = spmm_walk_lds_bytes<L, NV>(K, HV);
        if constexpr (HV == 1) {
          if (!off32) { launch(k_spmm_walk_f32<L, NV, 1, false>, lds_bytes, (const float*)w, (const float*)X, (float*)out, d4); return; }
        }
        launch(k_spmm_walk_f32<L, NV, HV, true>, lds_bytes, (const float*)w, (const float*)X, (float*)out, d4);
      }
    };
    switch ((int)h) {
      case 1: go(std::integral_constant<int, 1>{}); break;
      case 2: go(std::integral_constant<int, 2>{}); break;
      case 4: go(std::integral_constant<int, 4>{}); break;
      default: go(std::integral_constant<int, 8>{}); break;
    }
    return 1;
  }
}

}  // namespace

// Fused attention forward as one walk-style pass (kernels_attn_walk.h): fp32, one head, d = 64, row-major plan with
// identity eid, tables < 4 GiB.  Workspace: piece records of the rows that bins share + two n_q-sized merge arrays.
int attn_fwd_walk(const graphop_plan* plan, i64 n_q, i64 n_k, i64 h, i64 d, int dtype, const void* Q, const void* K,
                  const void* V, void* o, void* stats, void* ws, i64 ws_bytes, hipStream_t st, bool dry_run,
                  size_t* ws_needed) {
  constexpr int L = 16;
  const Tuning& t = tuning();
  if (ws_needed) *ws_needed = 0;
  if (!t.attn_fwd_walk || !t.attn_fused || t.force_generic || dtype != GRAPHOP_F32 || h != 1 || d != 4 * L || !plan) return 0;
  if (!plan->info.eid_identity || !(t.walk & 2) || !table_off32(n_k, L, 1)) return 0;
  if (plan->info.max_row >= n_q || plan->info.max_index >= n_k) return 0;
  constexpr int KR = attn_walk_rows<L>();
  static_assert(KR >= 4, "rows per lane group of the fused forward");
  WalkLaunch wl;
  const int use = choose_walk<L, 1>(plan, n_k, KR, st, &wl, dry_run, /*tables=*/2);
  if (use != 1) return use;
  // one pacing step per window (the SpMM walk's two per window measured 1 % slower here: 11.93 vs 11.78 ms per fused step)
  if (t.walk_steps > 1 && wl.view.steps >= 2 * t.walk_steps) wl.view.steps /= t.walk_steps;
  const i64 bins = (i64)wl.view.groups * wl.view.rounds, np = 2 * bins, F = 4 * L;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_row = 0, o_ml = o_row + up(sizeof(int) * (size_t)np), o_po = o_ml + up(sizeof(float) * 2 * (size_t)np);
  const size_t o_m = o_po + up(sizeof(float) * (size_t)(np * F)), o_l = o_m + up(sizeof(float) * (size_t)n_q);
  const size_t need = o_l + up(sizeof(float) * (size_t)n_q);
  if (ws_needed) *ws_needed = need;
  if (dry_run) return 1;
  if (!ws || (size_t)ws_bytes < need) return 0;    // (the caller sized the workspace for the composed path: always larger)
  char* base = (char*)ws;
  AttnWalkArgs a;
  a.Q = (const float*)Q; a.K = (const float*)K; a.V = (const float*)V; a.o = (float*)o; a.stats = (float*)stats;
  a.p_row = (int*)(base + o_row); a.p_ml = (float*)(base + o_ml); a.p_o = (float*)(base + o_po);
  float* Mtmp = (float*)(base + o_m);
  float* Ltmp = (float*)(base + o_l);
  if (zero_async(o, sizeof(float) * (size_t)(n_q * F), st) != hipSuccess) return -GRAPHOP_ERR_HIP;
  {
    ProfScope prof("attn_fwd_setup", st, "k_fill");
    const unsigned fb = (unsigned)(ceil_div(np > n_q ? np : n_q, 256) > 4096 ? 4096 : ceil_div(np > n_q ? np : n_q, 256));
    hipLaunchKernelGGL((k_fill<int>), dim3(fb), dim3(256), 0, st, a.p_row, np, -1);
    hipLaunchKernelGGL((k_fill<float>), dim3(fb), dim3(256), 0, st, Mtmp, n_q, kAttnNegBig);
    hipLaunchKernelGGL((k_fill<float>), dim3(fb), dim3(256), 0, st, Ltmp, n_q, 0.f);
  }
  {
    WalkDebug dbg;
    dbg.arm(&wl, "attn_fwd", st, kWalkWorkers / kWave);
    wl.view.launch_id = walk_launch_id("attn_fwd");
    ProfScope prof("attn_fwd", st, "k_attn_fwd_walk_f32");
    allow_full_lds((const void*)k_attn_fwd_walk_f32<L>);
    const size_t lds_bytes = (size_t)(kWalkWorkers / L) * attn_walk_group_bytes<L>(KR);
    hipLaunchKernelGGL((k_attn_fwd_walk_f32<L>), dim3(wl.blocks), dim3(kWalkThreads), lds_bytes, st, wl.view, a);
  }
  {
    ProfScope prof("attn_fwd_merge", st, "k_attn_piece_add");
    hipLaunchKernelGGL(k_attn_piece_max, dim3((unsigned)ceil_div(np, 256)), dim3(256), 0, st, a.p_row, a.p_ml, Mtmp, np);
    const unsigned gb = (unsigned)ceil_div(np, kFastBlock / L);
    hipLaunchKernelGGL((k_attn_piece_add<L>), dim3(gb), dim3(kFastBlock), 0, st, a.p_row, a.p_ml, a.p_o, Mtmp, Ltmp, a.o, np);
    hipLaunchKernelGGL((k_attn_piece_fin<L>), dim3(gb), dim3(kFastBlock), 0, st, a.p_row, Mtmp, Ltmp, a.o, a.stats, np);
  }
  if (hipGetLastError() != hipSuccess) return -GRAPHOP_ERR_HIP;
  return 1;
}

namespace {

template <int L, int NV>
int try_sddmm_sweep(const char* tag, int dtype, const graphop_plan* plan, i64 n_table_rows, const void* A,
                    const void* B, void* y, i64 h, int d4, hipStream_t st) {
  if (!plan) return 0;   // (plan-less calls -- the C ABI allows them, the partial SDDMM entry always makes them -- take the chunk drivers)
  SweepLaunch sl;
  SweepOpts so;
  const bool f64 = dtype == GRAPHOP_F64;
  if (f64 && (h != 1 || NV != 1 || L < 16 || !sddmm_staged(plan, L, NV, h, n_table_rows))) return 0;   // fp64: the staged strip only
  const bool off32 = table_off32(n_table_rows, L, NV);   // table < 4 GiB: 32-bit byte offsets
  so.bpc = (f64 || !off32) ? 3 : sweep_bpc(NV, h == 1);
  so.touch = tuning().touch_sddmm;
  const bool id = plan->info.eid_identity != 0;
  so.staged = sddmm_staged(plan, L, NV, h, n_table_rows);
  so.stage_lds_per_group = StageCfg<L, 1>::kLdsIntsPerGroup * (int)sizeof(int) + (h > 1 ? 64 * (int)h : 0);   // + [16 slots][h] results
  if (h > 1 && !aligned16(y)) so.staged = false;
  const int use = choose_sweep(plan, n_table_rows, L, NV, st, &sl, false, &so);
  if (use != 1) return use;
  const bool staged = sl.view.rec != nullptr;
  if (f64 && !staged) return 0;     // (no dealt layout: the caller falls back to the generic kernels; nothing launched yet)
  ProfScope prof(tag, st, f64 ? "k_sddmm_wown_staged_f64" : staged ? "k_sddmm_wown_staged_f32" : "k_sddmm_wown_f32");
  const dim3 grid(sl.blocks), block(kFastBlock);
  if (f64) {
    if constexpr (NV == 1 && L >= 16) {
      if (off32) hipLaunchKernelGGL((k_sddmm_wown_staged_f64<L, NV, true>), grid, block, sl.lds_bytes, st, sl.view,
                                    (const double*)A, (const double*)B, (double*)y);
      else hipLaunchKernelGGL((k_sddmm_wown_staged_f64<L, NV, false>), grid, block, sl.lds_bytes, st, sl.view,
                              (const double*)A, (const double*)B, (double*)y);
    }
    return 1;
  }
  const float* a = (const float*)A;
  const float* b = (const float*)B;
  float* yy = (float*)y;
  if (staged) {
    if (h == 1) {
      if (off32) hipLaunchKernelGGL((k_sddmm_wown_staged_f32<L, NV, 0, true>), grid, block, sl.lds_bytes, st, sl.view, a, b, yy);
      else hipLaunchKernelGGL((k_sddmm_wown_staged_f32<L, NV, 0, false>), grid, block, sl.lds_bytes, st, sl.view, a, b, yy);
    } else if constexpr (NV == 1 && L >= 16) {
      auto go = [&](auto dc) {
        constexpr int D4 = decltype(dc)::value;
        if constexpr (D4 < L)
          hipLaunchKernelGGL((k_sddmm_wown_staged_f32<L, NV, D4>), grid, block, sl.lds_bytes, st, sl.view, a, b, yy);
      };
      switch (d4) {
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        case 16: go(std::integral_constant<int, 16>{}); break;
        default: go(std::integral_constant<int, 32>{}); break;
      }
    }
    return 1;
  }
#define GO_K(H1, ID, O32)                                                                      \
  hipLaunchKernelGGL((k_sddmm_wown_f32<L, NV, H1, ID, O32>), grid, block, sl.lds_bytes, st,    \
                     sl.view, a, b, yy, (int)h, d4)
  if (h == 1) {
    if (id) { if (off32) GO_K(true, true, true); else GO_K(true, true, false); }
    else { if (off32) GO_K(true, false, true); else GO_K(true, false, false); }
  } else {
    if (id) GO_K(false, true, false); else GO_K(false, false, false);
  }
#undef GO_K
  return 1;
}

template <int L, int NV>
int try_spmm_sweep(const char* tag, const graphop_plan* plan, i64 n_table_rows, const void* w,
                   const void* X, void* out, i64 h, int d4, hipStream_t st) {
  SweepLaunch sl;
  SweepOpts so;
  so.bpc = sweep_bpc(NV, h == 1);
  const bool off32 = table_off32(n_table_rows, L, NV);
  so.staged = spmm_staged(L, NV, h, n_table_rows);
  const int use = choose_sweep(plan, n_table_rows, L, NV, st, &sl, /*accumulating=*/true, &so);
  if (use != 1) return use;
  const bool id = plan->info.eid_identity != 0;
  const float* ww = (const float*)w;
  const bool staged = sl.view.rec != nullptr && (id || sl.view.eids_w != nullptr);
  ProfScope prof(tag, st, staged ? "k_spmm_wown_staged_f32" : "k_spmm_wown_f32");
  const dim3 grid(sl.blocks), block(kFastBlock);
  const float* x = (const float*)X;
  float* o = (float*)out;
  if (staged) {
    const size_t lds = (size_t)(kFastBlock / L) * (id ? StageCfg<L, 1>::kLdsIntsPerGroup : StageCfg<L, 2>::kLdsIntsPerGroup) * sizeof(int);
    if (id) hipLaunchKernelGGL((k_spmm_wown_staged_f32<L, NV, true>), grid, block, lds, st, sl.view, ww, x, o);
    else hipLaunchKernelGGL((k_spmm_wown_staged_f32<L, NV, false>), grid, block, lds, st, sl.view, ww, x, o);
    return 1;
  }
#define GO_K(H1, ID, O32)                                                                      \
  hipLaunchKernelGGL((k_spmm_wown_f32<L, NV, H1, ID, O32>), grid, block, 0, st, sl.view, ww,   \
                     x, o, (int)h, d4)
  if (h == 1) {
    if (id) { if (off32) GO_K(true, true, true); else GO_K(true, true, false); }
    else { if (off32) GO_K(true, false, true); else GO_K(true, false, false); }
  } else {
    if (id) GO_K(false, true, false); else GO_K(false, false, false);
  }
#undef GO_K
  return 1;
}

// ---- block-dense drivers (kernels_block.h) ---------------------------------------------------------
inline bool block_ok(const graphop_plan* plan, int dtype, i64 h, i64 d, i64 n_table_rows) {
  const Tuning& t = tuning();
  return t.dense_blocks && !t.force_generic && dtype == GRAPHOP_F32 && plan && plan->blk_seg &&
         plan->info.n_dense_blocks > 0 && plan->info.dense_fill_pct >= t.dense_min_fill &&
         plan->idx32 && (plan->info.eid_identity || plan->eid32) && n_table_rows < 0x7fffffffLL &&
         plan->info.n_dense_blocks * h < 0x7fffffffLL && h * d < 0x7fffffffLL;
}
inline BlockView block_view(const graphop_plan* plan) {
  BlockView bv;
  bv.blk_seg = plan->blk_seg; bv.seg_e0 = plan->seg_e0; bv.seg_row = plan->seg_row;
  bv.idx32 = plan->idx32; bv.eid32 = plan->eid32; bv.nb = (int)plan->info.n_dense_blocks;
  return bv;
}

// Returns 1 when the block-dense SDDMM ran, 0 when it does not apply.
inline int try_sddmm_block(const char* tag, int dtype, const graphop_plan* plan, i64 n_table_rows,
                           const void* A, const void* B, void* y, i64 h, i64 d, hipStream_t st) {
  if (!block_ok(plan, dtype, h, d, n_table_rows) || d % 8 != 0 || !aligned16(A) || !aligned16(B)) return 0;
  const int nw = d % 32 == 0 ? 4 : (d % 16 == 0 ? 2 : 1);
  ProfScope prof(tag, st, "k_sddmm_block_f32");
  const dim3 grid((unsigned)(plan->info.n_dense_blocks * h)), block(kWave * nw);
  const BlockView bv = block_view(plan);
  if (plan->info.eid_identity)
    hipLaunchKernelGGL((k_sddmm_block_f32<true>), grid, block, 0, st, bv, (const float*)A,
                       (const float*)B, (float*)y, (int)h, (int)d);
  else
    hipLaunchKernelGGL((k_sddmm_block_f32<false>), grid, block, 0, st, bv, (const float*)A,
                       (const float*)B, (float*)y, (int)h, (int)d);
  return 1;
}

inline bool spmm_block_applies(int dtype, const graphop_plan* plan, i64 n_table_rows, const void* X,
                               const void* out, i64 h, i64 d) {
  return block_ok(plan, dtype, h, d, n_table_rows) && d % 32 == 0 && aligned16(X) && aligned16(out);
}
// The block-dense SpMM writes every feature of every row that has a segment, with plain stores:
// when the plan's segments are exactly the rows [0, n_out_rows) the output needs no zero fill.
inline bool spmm_block_writes_all(int dtype, const i64* row, const i64* indptr, const i64* eid,
                                  const i64* indices, i64 C, i64 E, const graphop_plan* plan,
                                  i64 n_table_rows, const void* X, const void* out, i64 h, i64 d,
                                  i64 n_out_rows) {
  return plan_matches_full(plan, row, indptr, eid, indices, C, E) &&
         spmm_block_applies(dtype, plan, n_table_rows, X, out, h, d) && plan->info.rows_sorted &&
         plan->info.n_segments == n_out_rows && plan->info.max_row == n_out_rows - 1;
}

inline int try_spmm_block(const char* tag, int dtype, const graphop_plan* plan, i64 n_table_rows,
                          const void* w, const void* X, void* out, i64 h, i64 d, hipStream_t st) {
  if (!spmm_block_applies(dtype, plan, n_table_rows, X, out, h, d)) return 0;
  const int vw = d % 128 == 0 ? 4 : (d % 64 == 0 ? 2 : 1);
  const i64 groups = d / (32 * vw);
  const int nw = (int)(groups < 4 ? groups : 4);
  ProfScope prof(tag, st, "k_spmm_block_f32");
  const dim3 grid((unsigned)(plan->info.n_dense_blocks * h)), block(kWave * nw);
  const BlockView bv = block_view(plan);
  const bool id = plan->info.eid_identity != 0;
#define GO_K(VW, ID) hipLaunchKernelGGL((k_spmm_block_f32<VW, ID>), grid, block, 0, st, bv, \
                                        (const float*)w, (const float*)X, (float*)out, (int)h, (int)d)
  if (vw == 4) { if (id) GO_K(4, true); else GO_K(4, false); }
  else if (vw == 2) { if (id) GO_K(2, true); else GO_K(2, false); }
  else { if (id) GO_K(1, true); else GO_K(1, false); }
#undef GO_K
  return 1;
}

// Will an SpMM-type pass over `plan` run on the chunk driver with row ownership, so that it can leave its output fully
// defined itself (k_spmm_f32<..., SELFZERO>) and the caller may skip the zero fill?  Only for outputs large enough for
// the fill to matter, fp32 lane-group shapes, sorted rows, and when neither the block-dense, the walk nor the
// window-owner drivers take the pass (asked with dry runs: what they build is what the pass would have built).
inline bool spmm_selfzero(int dtype, const i64* row, const i64* indptr, const i64* eid, const i64* indices, i64 C, i64 E,
                          const graphop_plan* plan, i64 n_table_rows, const void* w, const void* X, const void* out,
                          i64 h, i64 d, i64 n_out_rows, hipStream_t st) {
  const Tuning& t = tuning();
  if (!t.spmm_selfzero || C == 0 || !plan_matches_full(plan, row, indptr, eid, indices, C, E)) return false;
  if (!plan->info.rows_sorted || !fast_ok(dtype, h, d, E, n_table_rows) || !aligned16(out)) return false;
  if (plan->info.max_row >= n_out_rows || n_out_rows >= 0x7fffffffLL) return false;
  if ((double)n_out_rows * (double)(h * d) * 4.0 < (double)t.spmm_selfzero_min_mb * 1048576.0) return false;
  if (spmm_block_applies(dtype, plan, n_table_rows, X, out, h, d)) return false;
  // rows without edges between two chunk rows are zero-stored by ONE lane group per gap, serially (the tail behind the
  // last row is the host's: launch_spmm): a gap of more than 4 MB of rows is the device-wide fill's job after all
  if ((double)plan->info.max_row_gap * (double)(h * d) * 4.0 > 4.0 * 1048576.0) return false;
  int use = 0;
  GO_DISPATCH_LNV((int)(h * d), {
    if constexpr (NV == 1 && L >= 16) {
      const int bit = plan->info.eid_identity ? 2 : 4;
      const int K = spmm_walk_k<L, NV>(h, dtype);
      if ((t.walk & bit) && K > 0 && (h == 1 || (table_off32(n_table_rows, L, NV) && aligned16(w)))) {
        WalkLaunch wl;
        use = choose_walk<L, NV>(plan, n_table_rows, K, st, &wl, /*dry_run=*/true);
      }
    }
    if (use == 0) {
      SweepLaunch sl;
      SweepOpts so;
      so.dry_run = 1;
      so.bpc = sweep_bpc(NV, h == 1);
      so.staged = spmm_staged(L, NV, h, n_table_rows);
      use = choose_sweep(plan, n_table_rows, L, NV, st, &sl, /*accumulating=*/true, &so);
    }
  });
  return use == 0;
}

// The slot-walking form of the row-owning chunk driver (k_spmm_flat_f32): one head, one float4 per lane (d = 64 / 128 / 256),
// chosen where the chunks are short on average (the extended column side of a node-range shard: 6.8 slots per chunk, half
// of the chunks with one or two) -- at 13+ slots per chunk the per-chunk loop is within 6 % of the random-row rate already.
inline bool spmm_flat(i64 C, i64 E, i64 h, i64 d) {
  const Tuning& t = tuning();
  if (!t.spmm_flat || h != 1 || (d != 64 && d != 128 && d != 256) || C == 0) return false;
  // (small graphs are launch-bound and its longer prologue shows: Cora-shape passes 6.5 -> 9.1 us)
  return E < (i64)t.spmm_flat_max_mean * C && C >= (i64)t.spmm_flat_min_chunks;
}

// ---- launch helpers -------------------------------------------------------------------------------
template <bool EDGE_B>
int launch_sddmm(const char* tag, int dtype, const i64* row, const i64* indptr, const i64* eid,
                 const i64* indices, const void* A, const void* B, void* y, i64 C, i64 E,
                 i64 n_src_rows, i64 h, i64 d, const graphop_plan* plan, hipStream_t st) {
  if (C == 0) return GRAPHOP_OK;
  if (!plan_matches_full(plan, row, indptr, eid, indices, C, E)) plan = nullptr;
  if constexpr (!EDGE_B) {
    if (try_sddmm_block(tag, dtype, plan, n_src_rows, A, B, y, h, d, st)) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
  }
  // fp64 with a plan: the staged window-owner strip at rows of 256 B - 1 KB (d = 32 / 64 / 128), else the generic kernels
  if constexpr (!EDGE_B) {
    if (plan && fast64_ok(dtype, h, d, E, n_src_rows)) {
      int use = 0;
      GO_DISPATCH_LNV((int)(2 * d), { use = try_sddmm_sweep<L, NV>(tag, dtype, plan, n_src_rows, A, B, y, h, 0, st); });
      if (use < 0) return -use;
      if (use == 1) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
    }
  }
  // EDGE_B (node_mul_edge): B rows are d wide, A rows h*d wide -> fast path only for h == 1
  if (fast_ok(dtype, h, d, E, n_src_rows) && (!EDGE_B || h == 1)) {
    const int cpg = cpg_for(C, tuning().sddmm_cpg, (int)(h * d));
    const int F = (int)(h * d), d4 = (int)(d / 4);
    if constexpr (!EDGE_B) {
      int use = 0;
      GO_DISPATCH_LNV(F, { use = try_sddmm_sweep<L, NV>(tag, dtype, plan, n_src_rows, A, B, y, h, d4, st); });
      if (use < 0) return -use;
      if (use == 1) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
    }
    ProfScope prof(tag, st, "k_sddmm_f32");
    GO_DISPATCH_LNV(F, {
      const i64 groups = ceil_div(C, cpg);
      const unsigned nb = blocks_for(groups, GroupCfg<L>::kGroupsPerBlock);
      if (h == 1)
        hipLaunchKernelGGL((k_sddmm_f32<L, NV, true, EDGE_B>), dim3(nb), dim3(kFastBlock), 0, st,
                           row, indptr, eid, indices, (const float*)A, (const float*)B, (float*)y,
                           C, (int)h, d4, cpg);
      else if constexpr (!EDGE_B)
        hipLaunchKernelGGL((k_sddmm_f32<L, NV, false, false>), dim3(nb), dim3(kFastBlock), 0, st,
                           row, indptr, eid, indices, (const float*)A, (const float*)B, (float*)y,
                           C, (int)h, d4, cpg);
    });
  } else {
    ProfScope prof(tag, st, "k_sddmm_generic");
    const unsigned nb = blocks_for(C, kGenericWavesPerBlock);
    if (dtype == GRAPHOP_F32)
      hipLaunchKernelGGL((k_sddmm_generic<float, EDGE_B>), dim3(nb), dim3(kGenericBlock), 0, st,
                         row, indptr, eid, indices, (const float*)A, (const float*)B, (float*)y, C,
                         h, d);
    else
      hipLaunchKernelGGL((k_sddmm_generic<double, EDGE_B>), dim3(nb), dim3(kGenericBlock), 0, st,
                         row, indptr, eid, indices, (const double*)A, (const double*)B, (double*)y,
                         C, h, d);
  }
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

template <bool EDGE_X>
int launch_spmm(const char* tag, int dtype, const i64* row, const i64* indptr, const i64* eid,
                const i64* indices, const void* w, const void* X, void* out, i64 C, i64 E,
                i64 n_src_rows, i64 h, i64 d, const graphop_plan* plan, hipStream_t st, i64 selfzero_rows = -1) {
  // selfzero_rows >= 0 (spmm_selfzero: the caller did NOT zero-fill `out`): straight to the self-zeroing chunk driver
  if (C == 0) return GRAPHOP_OK;
  if (!plan_matches_full(plan, row, indptr, eid, indices, C, E)) plan = nullptr;
  if constexpr (!EDGE_X) {
    if (selfzero_rows >= 0) {
      const bool flat = spmm_flat(C, E, h, d);
      const int cpg = cpg_for(C, flat ? tuning().spmm_flat_cpg : tuning().spmm_cpg, (int)(h * d));
      const int F = (int)(h * d), d4 = (int)(d / 4);
      // the rows behind the last chunk row (trailing nodes without edges) are zeroed from here, device-wide: left to the
      // kernel they are one lane group's serial stores (an output with a long empty tail: GBs through 16-64 lanes)
      if (plan && plan->info.max_row + 1 < selfzero_rows) {
        const i64 covered = plan->info.max_row + 1;
        GO_HIP(zero_async((char*)out + sizeof(float) * (size_t)(covered * F), sizeof(float) * (size_t)((selfzero_rows - covered) * F), st));
        selfzero_rows = covered;
      }
      ProfScope prof(tag, st, flat ? "k_spmm_flat_f32" : "k_spmm_f32");
      GO_DISPATCH_LNV(F, {
        const i64 groups = ceil_div(C, cpg);
        if (groups > 1)
          hipLaunchKernelGGL((k_zero_shared_rows<L, NV>), dim3(blocks_for(groups - 1, kFastBlock / L)), dim3(kFastBlock), 0, st,
                             row, (float*)out, C, cpg);
        const unsigned nb = blocks_for(groups, GroupCfg<L>::kGroupsPerBlock);
        if constexpr (NV == 1 && L >= 16) {
          if (flat) {
            hipLaunchKernelGGL((k_spmm_flat_f32<L, true>), dim3(nb), dim3(kFastBlock), 0, st, row, indptr, eid, indices,
                               (const float*)w, (const float*)X, (float*)out, C, cpg, selfzero_rows);
            GO_LAUNCH_CHECK();
            return GRAPHOP_OK;
          }
        }
        if (h == 1)
          hipLaunchKernelGGL((k_spmm_f32<L, NV, true, true, true>), dim3(nb), dim3(kFastBlock), 0, st, row, indptr, eid, indices,
                             (const float*)w, (const float*)X, (float*)out, C, (int)h, d4, cpg, selfzero_rows);
        else
          hipLaunchKernelGGL((k_spmm_f32<L, NV, false, true, true>), dim3(nb), dim3(kFastBlock), 0, st, row, indptr, eid, indices,
                             (const float*)w, (const float*)X, (float*)out, C, (int)h, d4, cpg, selfzero_rows);
      });
      GO_LAUNCH_CHECK();
      return GRAPHOP_OK;
    }
    if (try_spmm_block(tag, dtype, plan, n_src_rows, w, X, out, h, d, st)) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
  }
  // fp64 with a plan: the walk kernel at rows of 256 B - 1 KB (d = 32 / 64 / 128), else the generic kernels
  if constexpr (!EDGE_X) {
    if (plan && fast64_ok(dtype, h, d, E, n_src_rows)) {
      int use = 0;
      GO_DISPATCH_LNV((int)(2 * d), { use = try_spmm_walk<L, NV>(tag, dtype, plan, n_src_rows, w, X, out, h, 0, st); });
      if (use < 0) return -use;
      if (use == 1) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
    }
  }
  if (!EDGE_X && fast_ok(dtype, h, d, E, n_src_rows)) {
    const int cpg = cpg_for(C, tuning().spmm_cpg, (int)(h * d));
    const int F = (int)(h * d), d4 = (int)(d / 4);
    {
      int use = 0;
      GO_DISPATCH_LNV(F, { use = try_spmm_walk<L, NV>(tag, dtype, plan, n_src_rows, w, X, out, h, d4, st); });
      if (use < 0) return -use;
      if (use == 1) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
      GO_DISPATCH_LNV(F, { use = try_spmm_sweep<L, NV>(tag, plan, n_src_rows, w, X, out, h, d4, st); });
      if (use < 0) return -use;
      if (use == 1) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
    }
    // plan.rows_sorted: a row's chunks are adjacent, rows inside one group's range need no atomics
    const bool owned = plan && plan->info.rows_sorted && ((uintptr_t)out & 15) == 0;
    if (owned && spmm_flat(C, E, h, d)) {      // short chunks: the slot-walking form of the same driver
      int done = 0;
      const int fcpg = cpg_for(C, tuning().spmm_flat_cpg, (int)(h * d));
      ProfScope prof(tag, st, "k_spmm_flat_f32");
      GO_DISPATCH_LNV(F, {
        if constexpr (NV == 1 && L >= 16) {
          const unsigned nb = blocks_for(ceil_div(C, fcpg), GroupCfg<L>::kGroupsPerBlock);
          hipLaunchKernelGGL((k_spmm_flat_f32<L, false>), dim3(nb), dim3(kFastBlock), 0, st, row, indptr, eid, indices,
                             (const float*)w, (const float*)X, (float*)out, C, fcpg, (i64)0);
          done = 1;
        }
      });
      if (done) { GO_LAUNCH_CHECK(); return GRAPHOP_OK; }
    }
    ProfScope prof(tag, st, "k_spmm_f32");
    GO_DISPATCH_LNV(F, {
      const i64 groups = ceil_div(C, cpg);
      const unsigned nb = blocks_for(groups, GroupCfg<L>::kGroupsPerBlock);
      auto go = [&](auto h1, auto ow) {
        hipLaunchKernelGGL((k_spmm_f32<L, NV, decltype(h1)::value, decltype(ow)::value>), dim3(nb),
                           dim3(kFastBlock), 0, st, row, indptr, eid, indices, (const float*)w,
                           (const float*)X, (float*)out, C, (int)h, d4, cpg);
      };
      if (h == 1) { if (owned) go(std::true_type{}, std::true_type{}); else go(std::true_type{}, std::false_type{}); }
      else { if (owned) go(std::false_type{}, std::true_type{}); else go(std::false_type{}, std::false_type{}); }
    });
  } else {
    ProfScope prof(tag, st, "k_spmm_generic");
    const unsigned nb = blocks_for(C, kGenericWavesPerBlock);
    if (dtype == GRAPHOP_F32)
      hipLaunchKernelGGL((k_spmm_generic<float, EDGE_X>), dim3(nb), dim3(kGenericBlock), 0, st,
                         row, indptr, eid, indices, (const float*)w, (const float*)X, (float*)out,
                         C, h, d);
    else
      hipLaunchKernelGGL((k_spmm_generic<double, EDGE_X>), dim3(nb), dim3(kGenericBlock), 0, st,
                         row, indptr, eid, indices, (const double*)w, (const double*)X,
                         (double*)out, C, h, d);
  }
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

// node_mul_edge fast paths: d/4 lanes per edge row, H head rows of A in registers
inline bool nme_fast_ok(int dtype, i64 h, i64 d, i64 n_edges) {
  if (tuning().force_generic || dtype != GRAPHOP_F32 || n_edges >= 0x7fffffffLL) return false;
  return (d == 16 || d == 32 || d == 64 || d == 128 || d == 256) && (h == 1 || h == 2 || h == 4 || h == 8);
}
#define GO_NME_CASE(D, LDV, HV, ...) case D * 16 + HV: { constexpr int LD = LDV, H = HV; __VA_ARGS__; } break;
#define GO_NME_ROW(D, LDV, ...) GO_NME_CASE(D, LDV, 1, __VA_ARGS__) GO_NME_CASE(D, LDV, 2, __VA_ARGS__) \
                                GO_NME_CASE(D, LDV, 4, __VA_ARGS__) GO_NME_CASE(D, LDV, 8, __VA_ARGS__)
#define GO_DISPATCH_NME(d, h, ...)                                          \
  switch ((int)(d) * 16 + (int)(h)) {                                        \
    GO_NME_ROW(16, 4, __VA_ARGS__) GO_NME_ROW(32, 8, __VA_ARGS__)            \
    GO_NME_ROW(64, 16, __VA_ARGS__) GO_NME_ROW(128, 32, __VA_ARGS__)         \
    GO_NME_ROW(256, 64, __VA_ARGS__)                                         \
    default: break;                                                          \
  }

inline bool plan_matches(const graphop_plan* p, const i64* row, const i64* indptr, const i64* eid,
                         i64 C, i64 E) {
  return p && p->row == (const int64_t*)row && p->indptr == (const int64_t*)indptr &&
         p->eid == (const int64_t*)eid && p->info.n_chunks == C && p->info.n_edges == E;
}

inline int seg_group_width(i64 items_per_seg, i64 h) {
  int g = items_per_seg >= 48 ? 64 : items_per_seg >= 24 ? 32 : items_per_seg >= 12 ? 16 : 8;
  while (g < h) g <<= 1;
  return g;
}

template <typename T, bool BWD>
int launch_softmax_seg(const graphop_plan* p, const i64* indptr, const i64* eid, const T* in0,
                       const T* in1, T* out, i64 h, hipStream_t st, T* stats = nullptr) {
  const i64 S = p->info.n_segments;
  if (S == 0) return GRAPHOP_OK;
  ProfScope prof(BWD ? "softmax_bwd" : "softmax_fwd", st, BWD ? "k_softmax_bwd_seg" : "k_softmax_fwd_seg");
  const bool id = p->info.eid_identity != 0;
  if constexpr (std::is_same<T, float>::value) {
    // several heads in storage order: float4 items (kernels_fast.h: softmax_vec4_group)
    if (id && pow2(h) && h >= 4 && h <= 64 && aligned16(in0) && aligned16(out) && (!BWD || aligned16(in1)) &&
        kLongSegment * h < 0x7fffffffLL) {
      const i64 mean_n4 = p->info.n_edges * h / 4 / S;
      const int G = seg_group_width(mean_n4, h / 4);
      const bool small_rows = mean_n4 <= 2 * G;     // most rows within 8 float4s per lane: the low-register instantiation
      const int n_long = (int)p->n_long;
      const unsigned nb = blocks_for(S, kFastBlock / G) + (unsigned)n_long;
      const i64 long_len = n_long > 0 ? kLongSegment : (i64)1 << 30;
      prof.kernel = BWD ? "k_softmax_bwd_vec4" : "k_softmax_fwd_vec4";
#define GO_V4R(GW, RM)                                                                              \
  if constexpr (!BWD)                                                                               \
    hipLaunchKernelGGL((k_softmax_fwd_vec4<GW, (RM ? 8 : kVec4CacheFwd)>), dim3(nb), dim3(kFastBlock), 0, st, \
                       (const i64*)p->seg_chunk, indptr, (const i64*)p->seg_eptr, in0, out, S, (int)h, long_len,             \
                       (const int*)p->long_segs, n_long, (const i64*)p->row, stats);                \
  else                                                                                              \
    hipLaunchKernelGGL((k_softmax_bwd_vec4<GW, (RM ? 8 : kVec4CacheBwd)>), dim3(nb), dim3(kFastBlock), 0, st, \
                       (const i64*)p->seg_chunk, indptr, (const i64*)p->seg_eptr, in0, in1, out, S, (int)h, long_len,        \
                       (const int*)p->long_segs, n_long);
#define GO_V4(GW) if (small_rows) { GO_V4R(GW, 1) } else { GO_V4R(GW, 0) }
      switch (G) {
        case 8: GO_V4(8) break;
        case 16: GO_V4(16) break;
        case 32: GO_V4(32) break;
        default: GO_V4(64) break;
      }
#undef GO_V4
#undef GO_V4R
      GO_LAUNCH_CHECK();
      return GRAPHOP_OK;
    }
  }
  if (pow2(h) && h <= 64) {
    const int G = seg_group_width(p->info.n_edges * h / S, h);
    const int n_long = (int)p->n_long;
    const unsigned nb = blocks_for(S, kFastBlock / G) + (unsigned)n_long;
    const i64 long_len = n_long > 0 ? (BWD ? kLongSegmentBwd : kLongSegment) : (i64)1 << 62;
#define GO_SEG(GW, ID)                                                                           \
  if constexpr (!BWD)                                                                            \
    hipLaunchKernelGGL((k_softmax_fwd_seg<T, GW, ID>), dim3(nb), dim3(kFastBlock), 0, st,        \
                       (const i64*)p->seg_chunk, indptr, (const i64*)p->seg_eptr, eid, in0, out, S, (int)h, long_len,     \
                       (const int*)p->long_segs, n_long, (const i64*)p->row, stats);             \
  else                                                                                           \
    hipLaunchKernelGGL((k_softmax_bwd_seg<T, GW, ID>), dim3(nb), dim3(kFastBlock), 0, st,        \
                       (const i64*)p->seg_chunk, indptr, (const i64*)p->seg_eptr, eid, in0, in1, out, S, (int)h, long_len, \
                       (const int*)p->long_segs, n_long);
#define GO_SEG_ID(GW) if (id) { GO_SEG(GW, true) } else { GO_SEG(GW, false) }
    switch (G) {
      case 8: GO_SEG_ID(8) break;
      case 16: GO_SEG_ID(16) break;
      case 32: GO_SEG_ID(32) break;
      default: GO_SEG_ID(64) break;
    }
#undef GO_SEG_ID
#undef GO_SEG
  } else {
    const unsigned nb = blocks_for(S, kFastBlock / kWave);
    hipLaunchKernelGGL((k_softmax_seg_anyh<T, BWD>), dim3(nb), dim3(kFastBlock), 0, st,
                       (const i64*)p->seg_chunk, indptr, (const i64*)p->seg_eptr, eid, in0, in1, out, S, h, (const i64*)p->row,
                       BWD ? (T*)nullptr : stats);
  }
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

template <typename T>
int softmax_forward_t(const i64* row, const i64* indptr, const i64* eid, const T* x, T* y, i64 C,
                      i64 E, i64 h, T* ws, i64 ws_rows, const graphop_plan* plan, hipStream_t st,
                      T* stats = nullptr) {
  const bool owned = plan_matches(plan, row, indptr, eid, C, E) && plan->info.row_owned;
  const bool covered = owned && plan->info.full_coverage && plan->info.eid_identity;
  if (!covered && E * h > 0) GO_HIP(zero_async(y, sizeof(T) * (size_t)(E * h), st));
  if (C == 0) return GRAPHOP_OK;
  if (owned) return launch_softmax_seg<T, false>(plan, indptr, eid, x, (const T*)nullptr, y, h, st, stats);
  GO_CHECK_ARG(ws != nullptr && ws_rows > 0,
               "sparse_softmax_forward: the general (plan-less) path needs a workspace of "
               "2*workspace_rows*h values with workspace_rows > max(row)");
  T* max_val = ws;
  T* sum = ws + ws_rows * h;
  const unsigned fb = (unsigned)ceil_div(ws_rows * h, 256) > 4096u ? 4096u
                                                                   : (unsigned)ceil_div(ws_rows * h, 256);
  hipLaunchKernelGGL((k_fill<T>), dim3(fb), dim3(256), 0, st, max_val, ws_rows * h, (T)-1e9);
  GO_HIP(zero_async(sum, sizeof(T) * (size_t)(ws_rows * h), st));
  const unsigned nb = blocks_for(C, kGenericWavesPerBlock);
  hipLaunchKernelGGL((k_softmax_max<T>), dim3(nb), dim3(kGenericBlock), 0, st, row, indptr, eid, x,
                     max_val, C, h);
  hipLaunchKernelGGL((k_softmax_exp_sum<T>), dim3(nb), dim3(kGenericBlock), 0, st, row, indptr,
                     eid, x, (const T*)max_val, sum, y, C, h);
  hipLaunchKernelGGL((k_softmax_norm<T>), dim3(nb), dim3(kGenericBlock), 0, st, row, indptr, eid,
                     (const T*)sum, y, C, h);
  if (stats)   // row statistics for the fused attention backward, rows [0, ws_rows)
    hipLaunchKernelGGL((k_attn_stats_from_ws<T>), dim3(fb), dim3(256), 0, st, (const T*)max_val,
                       (const T*)sum, stats, ws_rows * h);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

template <typename T>
int softmax_backward_t(const i64* row, const i64* indptr, const i64* eid, const T* y, const T* dy,
                       T* dx, i64 C, i64 E, i64 h, T* ws, i64 ws_rows, const graphop_plan* plan,
                       hipStream_t st) {
  const bool owned = plan_matches(plan, row, indptr, eid, C, E) && plan->info.row_owned;
  const bool covered = owned && plan->info.full_coverage && plan->info.eid_identity;
  if (!covered && E * h > 0) GO_HIP(zero_async(dx, sizeof(T) * (size_t)(E * h), st));
  if (C == 0) return GRAPHOP_OK;
  if (owned) return launch_softmax_seg<T, true>(plan, indptr, eid, y, dy, dx, h, st);
  GO_CHECK_ARG(ws != nullptr && ws_rows > 0,
               "sparse_softmax_backward: the general (plan-less) path needs a workspace of "
               "workspace_rows*h values with workspace_rows > max(row)");
  GO_HIP(zero_async(ws, sizeof(T) * (size_t)(ws_rows * h), st));
  const unsigned nb = blocks_for(C, kGenericWavesPerBlock);
  hipLaunchKernelGGL((k_softmax_bwd_aggre<T>), dim3(nb), dim3(kGenericBlock), 0, st, row, indptr,
                     eid, dy, y, ws, C, h);
  hipLaunchKernelGGL((k_softmax_bwd_dx<T>), dim3(nb), dim3(kGenericBlock), 0, st, row, indptr, eid,
                     dy, y, (const T*)ws, dx, C, h);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

inline int check_common(const char* fn, int dtype, i64 C, i64 E, i64 h, i64 d) {
  GO_TRY(check_async_error(false));   // a kernel of an earlier launch reported a failure: sticky until acknowledged
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: dtype must be GRAPHOP_F32 or "
               "GRAPHOP_F64", fn);
  GO_CHECK_ARG(C >= 0 && E >= 0 && h >= 1 && d >= 0, "%s: negative size (n_chunks=%lld n_edges=%lld "
               "h=%lld d=%lld)", fn, (long long)C, (long long)E, (long long)h, (long long)d);
  return GRAPHOP_OK;
}

// Row ids address the row-side operand / output: with a plan the largest one is known, so an
// operand with too few rows is an error here instead of an out-of-bounds access on the device.
inline int check_rows(const char* fn, const char* what, const graphop_plan* plan, i64 n_rows) {
  GO_CHECK_ARG(!plan || plan->info.max_row < n_rows,
               "%s: row id %lld but %s has only %lld rows", fn, plan ? (long long)plan->info.max_row : 0LL,
               what, (long long)n_rows);
  return GRAPHOP_OK;
}
inline int check_cols(const char* fn, const char* what, const graphop_plan* plan, i64 n_rows) {
  GO_CHECK_ARG(!plan || plan->info.max_index < n_rows,
               "%s: neighbour id %lld but %s has only %lld rows", fn,
               plan ? (long long)plan->info.max_index : 0LL, what, (long long)n_rows);
  return GRAPHOP_OK;
}


}  // namespace

// sparse_softmax_forward that also leaves the row statistics (max, 1 / sum) behind (attention.hip)
int softmax_forward_stats(int dtype, const i64* row, const i64* indptr, const i64* eid, const void* x,
                          void* y, i64 C, i64 E, i64 h, void* ws, i64 ws_rows,
                          const graphop_plan* plan, hipStream_t st, void* stats) {
  if (dtype == GRAPHOP_F32)
    return softmax_forward_t<float>(row, indptr, eid, (const float*)x, (float*)y, C, E, h, (float*)ws,
                                    ws_rows, plan, st, (float*)stats);
  return softmax_forward_t<double>(row, indptr, eid, (const double*)x, (double*)y, C, E, h, (double*)ws,
                                   ws_rows, plan, st, (double*)stats);
}
}  // namespace graphop

using namespace graphop;

extern "C" {

int graphop_plan_prepare(graphop_plan_t* plan, int dtype, int64_t n_table_rows, int64_t h, int64_t d,
                         int fused, void* stream) {
  GO_CHECK_ARG(plan != nullptr && n_table_rows >= 0 && h >= 1 && d >= 0, "plan_prepare: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const bool f64 = fast64_ok(dtype, h, d, plan->info.n_edges, n_table_rows);   // fp64: staged SDDMM strip + walk only
  if (!f64 && !fast_ok(dtype, h, d, plan->info.n_edges, n_table_rows)) return GRAPHOP_OK;   // generic kernels: nothing cached
  int rc = 0;
  GO_DISPATCH_LNV((int)(f64 ? 2 * d : h * d), {
    SweepLaunch sl;
    SweepOpts o;
    o.dry_run = 1;
    o.bpc = f64 ? 3 : sweep_bpc(NV, h == 1);
    o.staged = sddmm_staged(plan, L, NV, h, n_table_rows);
    o.stage_lds_per_group = StageCfg<L, 1>::kLdsIntsPerGroup * (int)sizeof(int) + (h > 1 ? 64 * (int)h : 0);
    if (!f64 || (o.staged && NV == 1 && L >= 16))
      rc = choose_sweep(plan, n_table_rows, L, NV, st, &sl, /*accumulating=*/false, &o);
    o.staged = spmm_staged(L, NV, h, n_table_rows);
    if (rc >= 0 && !f64) rc = choose_sweep(plan, n_table_rows, L, NV, st, &sl, /*accumulating=*/true, &o);
    // walk layouts of the passes that would take them (kernels_walk.h)
    if constexpr (NV == 1 && L >= 16) {
      WalkLaunch wl;
      const int bit = plan->info.eid_identity ? 2 : 4;
      const int K = spmm_walk_k<L, NV>(h, dtype);
      if (rc >= 0 && K > 0 && (tuning().walk & bit) && (h == 1 || table_off32(n_table_rows, L, NV))) {
        const int r2 = choose_walk<L, NV>(plan, n_table_rows, K, st, &wl, /*dry_run=*/true);
        if (r2 < 0) rc = r2;
      }
    }
  });
  if (rc < 0) return -rc;
  if (fused && h == 1) {
    // the plan serves as the row-major side (fused = 2), as the column-major side (3) or as either
    // (1) of the fused passes: same window geometry, but the piece length and the dealt id layout
    // differ with the resident grid of the pass
    rc = 0;
    if (fused != 3) {   // the row-major side also serves the one-pass forward (kernels_attn_walk.h)
      const int fw = attn_fwd_walk(plan, plan->info.max_row + 1, n_table_rows, h, d, dtype, nullptr, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, 0, st, /*dry_run=*/true, nullptr);
      if (fw < 0) return -fw;
    }
    if (fused != 3) rc = attn_prepare_plan(plan, n_table_rows, d, false, st);
    if (rc >= 0 && fused != 2) rc = attn_prepare_plan(plan, n_table_rows, d, true, st);
    if (rc < 0) return -rc;
  }
  return GRAPHOP_OK;
}

int graphop_maskedmm_csr_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                 const int64_t* eid, const int64_t* indices, const void* A,
                                 const void* B, void* y, int64_t n_chunks, int64_t n_edges,
                                 int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                 const graphop_plan_t* plan, void* stream) {
  const char* fn = "maskedmm_csr_forward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, d));
  hipStream_t st = (hipStream_t)stream;
  if (n_edges * h == 0) return GRAPHOP_OK;
  GO_PTR(fn, y);
  const bool covered = plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                                    n_chunks, n_edges) &&
                       plan->info.full_coverage && plan->info.eid_identity &&
                       plan->info.indptr_monotone;
  if (!covered) GO_HIP(zero_async(y, esize(dtype) * (size_t)(n_edges * h), st));
  if (n_chunks == 0) return GRAPHOP_OK;
  GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, indices); GO_PTR(fn, A); GO_PTR(fn, B);
  {
    const graphop_plan* pm = plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid, n_chunks, n_edges) ? plan : nullptr;
    GO_TRY(check_rows(fn, "A", pm, n_a));
    GO_TRY(check_cols(fn, "B", pm && pm->indices == indices ? pm : nullptr, n_b));
  }
  return launch_sddmm<false>("sddmm_fwd", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                             (const i64*)indices, A, B, y, n_chunks, n_edges, n_b, h, d, plan, st);
}

// SDDMM over a subset of the slots into a shared result array: no zero fill, eid indexes y (n_y entries).
int graphop_maskedmm_csr_forward_partial(int dtype, const int64_t* row, const int64_t* indptr,
                                         const int64_t* eid, const int64_t* indices, const void* A,
                                         const void* B, void* y, int64_t n_chunks, int64_t n_slots,
                                         int64_t n_y, int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                         void* stream) {
  const char* fn = "maskedmm_csr_forward_partial";
  GO_TRY(check_common(fn, dtype, n_chunks, n_slots, h, d));
  GO_CHECK_ARG(n_y >= n_slots && n_a >= 0 && n_b >= 0, "%s: n_y (%lld) must be at least n_slots (%lld): eid[] names distinct "
               "entries of y", fn, (long long)n_y, (long long)n_slots);
  hipStream_t st = (hipStream_t)stream;
  if (n_slots * h == 0 || n_chunks == 0) return GRAPHOP_OK;
  GO_PTR(fn, y); GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, indices); GO_PTR(fn, A); GO_PTR(fn, B);
  // E bounds the 32-bit forms of the kernels' edge offsets: the result array's size, not the sub-graph's
  return launch_sddmm<false>("sddmm_fwd_part", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                             (const i64*)indices, A, B, y, n_chunks, n_y, n_b, h, d, nullptr, st);
}

int graphop_maskedmm_csr_backward(int dtype, const int64_t* row, const int64_t* indptr_r,
                                  const int64_t* eid_r, const int64_t* indices_r,
                                  const int64_t* col, const int64_t* indptr_c,
                                  const int64_t* eid_c, const int64_t* indices_c, const void* A,
                                  const void* B, const void* dy, void* dA, void* dB,
                                  int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_edges,
                                  int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                  const graphop_plan_t* plan_r, const graphop_plan_t* plan_c,
                                  void* stream) {
  const char* fn = "maskedmm_csr_backward";
  GO_TRY(check_common(fn, dtype, n_row_chunks, n_edges, h, d));
  GO_CHECK_ARG(n_col_chunks >= 0 && n_a >= 0 && n_b >= 0, "%s: negative size", fn);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = esize(dtype);
  {
    const graphop_plan* pr = plan_matches_full(plan_r, (const i64*)row, (const i64*)indptr_r, (const i64*)eid_r, (const i64*)indices_r, n_row_chunks, n_edges) ? plan_r : nullptr;
    const graphop_plan* pc = plan_matches_full(plan_c, (const i64*)col, (const i64*)indptr_c, (const i64*)eid_c, (const i64*)indices_c, n_col_chunks, n_edges) ? plan_c : nullptr;
    GO_TRY(check_rows(fn, "A / dA", pr, n_a)); GO_TRY(check_cols(fn, "B", pr, n_b));
    GO_TRY(check_rows(fn, "B / dB", pc, n_b)); GO_TRY(check_cols(fn, "A", pc, n_a));
  }
  // an output whose orientation has no chunks may be NULL: that half of the op is skipped entirely
  // (the sharded step calls the op once per orientation to overlap the dK exchange, dist.py)
  // (sz_*: the pass will run on the self-zeroing chunk driver, which leaves every output row defined: no fill)
  bool sz_a = false, sz_b = false;
  if (n_a * h * d > 0 && !(dA == nullptr && n_row_chunks == 0)) {
    GO_PTR(fn, dA);
    sz_a = dy != nullptr && B != nullptr &&
           spmm_selfzero(dtype, (const i64*)row, (const i64*)indptr_r, (const i64*)eid_r, (const i64*)indices_r, n_row_chunks,
                         n_edges, plan_r, n_b, dy, B, dA, h, d, n_a, st);
    if (!sz_a && !spmm_block_writes_all(dtype, (const i64*)row, (const i64*)indptr_r, (const i64*)eid_r, (const i64*)indices_r,
                                        n_row_chunks, n_edges, plan_r, n_b, B, dA, h, d, n_a))
      GO_HIP(zero_async(dA, es * (size_t)(n_a * h * d), st));
  }
  if (n_b * h * d > 0 && !(dB == nullptr && n_col_chunks == 0)) {
    GO_PTR(fn, dB);
    sz_b = dy != nullptr && A != nullptr &&
           spmm_selfzero(dtype, (const i64*)col, (const i64*)indptr_c, (const i64*)eid_c, (const i64*)indices_c, n_col_chunks,
                         n_edges, plan_c, n_a, dy, A, dB, h, d, n_b, st);
    if (!sz_b && !spmm_block_writes_all(dtype, (const i64*)col, (const i64*)indptr_c, (const i64*)eid_c, (const i64*)indices_c,
                                        n_col_chunks, n_edges, plan_c, n_a, A, dB, h, d, n_b))
      GO_HIP(zero_async(dB, es * (size_t)(n_b * h * d), st));
  }
  if (h * d == 0) return GRAPHOP_OK;
  if (n_row_chunks > 0) {
    GO_PTR(fn, row); GO_PTR(fn, indptr_r); GO_PTR(fn, eid_r); GO_PTR(fn, indices_r); GO_PTR(fn, B); GO_PTR(fn, dy);
    GO_TRY(launch_spmm<false>("sddmm_bwd_dA", dtype, (const i64*)row, (const i64*)indptr_r, (const i64*)eid_r,
                              (const i64*)indices_r, dy, B, dA, n_row_chunks, n_edges, n_b, h, d, plan_r, st, sz_a ? n_a : -1));
  }
  if (n_col_chunks > 0) {
    GO_PTR(fn, col); GO_PTR(fn, indptr_c); GO_PTR(fn, eid_c); GO_PTR(fn, indices_c); GO_PTR(fn, A); GO_PTR(fn, dy);
    GO_TRY(launch_spmm<false>("sddmm_bwd_dB", dtype, (const i64*)col, (const i64*)indptr_c, (const i64*)eid_c,
                              (const i64*)indices_c, dy, A, dB, n_col_chunks, n_edges, n_a, h, d, plan_c, st, sz_b ? n_b : -1));
  }
  return GRAPHOP_OK;
}

int graphop_sparse_softmax_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                   const int64_t* eid, const void* x, void* y, int64_t n_chunks,
                                   int64_t n_edges, int64_t h, void* workspace,
                                   int64_t workspace_rows, const graphop_plan_t* plan,
                                   void* stream) {
  const char* fn = "sparse_softmax_forward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, 0));
  if (n_edges * h == 0) return GRAPHOP_OK;
  GO_PTR(fn, y);
  if (n_chunks > 0) { GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, x); }
  if (dtype == GRAPHOP_F32)
    return softmax_forward_t<float>((const i64*)row, (const i64*)indptr, (const i64*)eid,
                                    (const float*)x, (float*)y, n_chunks, n_edges, h,
                                    (float*)workspace, workspace_rows, plan, (hipStream_t)stream);
  return softmax_forward_t<double>((const i64*)row, (const i64*)indptr, (const i64*)eid,
                                   (const double*)x, (double*)y, n_chunks, n_edges, h,
                                   (double*)workspace, workspace_rows, plan, (hipStream_t)stream);
}

int graphop_sparse_softmax_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                    const int64_t* eid, const void* y, const void* dy, void* dx,
                                    int64_t n_chunks, int64_t n_edges, int64_t h,
                                    void* workspace, int64_t workspace_rows,
                                    const graphop_plan_t* plan, void* stream) {
  const char* fn = "sparse_softmax_backward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, 0));
  if (n_edges * h == 0) return GRAPHOP_OK;
  GO_PTR(fn, dx);
  if (n_chunks > 0) { GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, y); GO_PTR(fn, dy); }
  if (dtype == GRAPHOP_F32)
    return softmax_backward_t<float>((const i64*)row, (const i64*)indptr, (const i64*)eid,
                                     (const float*)y, (const float*)dy, (float*)dx, n_chunks,
                                     n_edges, h, (float*)workspace, workspace_rows, plan,
                                     (hipStream_t)stream);
  return softmax_backward_t<double>((const i64*)row, (const i64*)indptr, (const i64*)eid,
                                    (const double*)y, (const double*)dy, (double*)dx, n_chunks,
                                    n_edges, h, (double*)workspace, workspace_rows, plan,
                                    (hipStream_t)stream);
}

int graphop_vector_spmm_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                const int64_t* eid, const int64_t* indices, const void* edata,
                                const void* x, void* y, int64_t n_chunks, int64_t n_edges,
                                int64_t n_x, int64_t n_y, int64_t h, int64_t d,
                                const graphop_plan_t* plan, void* stream) {
  const char* fn = "vector_spmm_forward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, d));
  GO_CHECK_ARG(n_x >= 0 && n_y >= 0, "%s: negative size", fn);
  hipStream_t st = (hipStream_t)stream;
  {
    const graphop_plan* pm = plan_matches_full(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid, (const i64*)indices, n_chunks, n_edges) ? plan : nullptr;
    GO_TRY(check_rows(fn, "y", pm, n_y)); GO_TRY(check_cols(fn, "x", pm, n_x));
  }
  if (n_y * h * d == 0) return GRAPHOP_OK;
  GO_PTR(fn, y);
  const bool sz = edata != nullptr && x != nullptr &&
                  spmm_selfzero(dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid, (const i64*)indices, n_chunks, n_edges,
                                plan, n_x, edata, x, y, h, d, n_y, st);
  if (!sz && !spmm_block_writes_all(dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid, (const i64*)indices,
                                    n_chunks, n_edges, plan, n_x, x, y, h, d, n_y))
    GO_HIP(zero_async(y, esize(dtype) * (size_t)(n_y * h * d), st));
  if (n_chunks == 0) return GRAPHOP_OK;
  GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, indices); GO_PTR(fn, edata); GO_PTR(fn, x);
  return launch_spmm<false>("spmm_fwd", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                            (const i64*)indices, edata, x, y, n_chunks, n_edges, n_x, h, d, plan, st, sz ? n_y : -1);
}

int graphop_vector_spmm_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                 const int64_t* eid, const int64_t* indices, const int64_t* col,
                                 const int64_t* indptr_t, const int64_t* eid_t,
                                 const int64_t* indices_t, const void* edata, const void* dy,
                                 const void* x, void* dedata, void* dx, int64_t n_row_chunks,
                                 int64_t n_col_chunks, int64_t n_edges, int64_t n_x, int64_t n_dy,
                                 int64_t h, int64_t d, const graphop_plan_t* plan_r,
                                 const graphop_plan_t* plan_c, void* stream) {
  const char* fn = "vector_spmm_backward";
  GO_TRY(check_common(fn, dtype, n_row_chunks, n_edges, h, d));
  GO_CHECK_ARG(n_col_chunks >= 0 && n_x >= 0 && n_dy >= 0, "%s: negative size", fn);
  hipStream_t st = (hipStream_t)stream;
  const size_t es = esize(dtype);
  {
    const graphop_plan* pr = plan_matches_full(plan_r, (const i64*)row, (const i64*)indptr, (const i64*)eid, (const i64*)indices, n_row_chunks, n_edges) ? plan_r : nullptr;
    const graphop_plan* pc = plan_matches_full(plan_c, (const i64*)col, (const i64*)indptr_t, (const i64*)eid_t, (const i64*)indices_t, n_col_chunks, n_edges) ? plan_c : nullptr;
    GO_TRY(check_rows(fn, "dy", pr, n_dy)); GO_TRY(check_cols(fn, "x", pr, n_x));
    GO_TRY(check_rows(fn, "x / dx", pc, n_x)); GO_TRY(check_cols(fn, "dy", pc, n_dy));
  }
  if (n_edges * h > 0) {
    GO_PTR(fn, dedata);
    const bool covered = plan_matches(plan_r, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                                      n_row_chunks, n_edges) &&
                         plan_r->info.full_coverage && plan_r->info.eid_identity &&
                         plan_r->info.indptr_monotone;
    if (!covered) GO_HIP(zero_async(dedata, es * (size_t)(n_edges * h), st));
  }
  bool sz_x = false;
  // (dx may be NULL when n_col_chunks == 0: that half of the op is skipped, as in maskedmm_csr_backward -- the sharded
  // step computes dx together with dK in one column-major launch, graphop_spmm_pair)
  if (n_x * h * d > 0 && !(dx == nullptr && n_col_chunks == 0)) {
    GO_PTR(fn, dx);
    sz_x = edata != nullptr && dy != nullptr &&
           spmm_selfzero(dtype, (const i64*)col, (const i64*)indptr_t, (const i64*)eid_t, (const i64*)indices_t, n_col_chunks,
                         n_edges, plan_c, n_dy, edata, dy, dx, h, d, n_x, st);
    if (!sz_x && !spmm_block_writes_all(dtype, (const i64*)col, (const i64*)indptr_t, (const i64*)eid_t, (const i64*)indices_t,
                                        n_col_chunks, n_edges, plan_c, n_dy, dy, dx, h, d, n_x))
      GO_HIP(zero_async(dx, es * (size_t)(n_x * h * d), st));
  }
  if (h * d == 0) return GRAPHOP_OK;
  if (n_row_chunks > 0 && n_edges > 0) {
    GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, indices); GO_PTR(fn, dy); GO_PTR(fn, x);
    // kernel_0: dedata = SDDMM(dy, x) over the row-major CSR (graphop_kernel.cu:135-149)
    GO_TRY(launch_sddmm<false>("spmm_bwd_dedata", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                               (const i64*)indices, dy, x, dedata, n_row_chunks, n_edges, n_x, h, d, plan_r, st));
  }
  if (n_col_chunks > 0) {
    GO_PTR(fn, col); GO_PTR(fn, indptr_t); GO_PTR(fn, eid_t); GO_PTR(fn, indices_t); GO_PTR(fn, edata); GO_PTR(fn, dy);
    // kernel_1: dx = SpMM(edata, dy) over the column-major CSR, all C' chunks (:151-163)
    GO_TRY(launch_spmm<false>("spmm_bwd_dx", dtype, (const i64*)col, (const i64*)indptr_t, (const i64*)eid_t,
                              (const i64*)indices_t, edata, dy, dx, n_col_chunks, n_edges, n_dy, h, d, plan_c, st, sz_x ? n_x : -1));
  }
  return GRAPHOP_OK;
}

int graphop_spmm_pair_supported(int dtype, int64_t n_chunks, int64_t n_edges, int64_t n_x, int64_t h, int64_t d,
                                const graphop_plan_t* plan) {
  if (tuning().force_generic || dtype != GRAPHOP_F32 || h != 1 || (d != 64 && d != 128 && d != 256)) return 0;
  if (!plan || !plan->info.rows_sorted || plan->info.n_chunks != n_chunks || plan->info.n_edges != n_edges) return 0;
  return n_edges < 0x7fffffffLL && n_x < 0x7fffffffLL && plan->info.max_index < n_x;
}

int graphop_spmm_pair(int dtype, const int64_t* row, const int64_t* indptr, const int64_t* eid,
                      const int64_t* indices, const void* w2, const void* X0, const void* X1, void* out0,
                      void* out1, int64_t n_chunks, int64_t n_edges, int64_t n_x, int64_t n_out, int64_t h,
                      int64_t d, const graphop_plan_t* plan, void* stream) {
  const char* fn = "spmm_pair";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, d));
  hipStream_t st = (hipStream_t)stream;
  GO_CHECK_ARG(plan_matches_full(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid, (const i64*)indices, n_chunks, n_edges) &&
               graphop_spmm_pair_supported(dtype, n_chunks, n_edges, n_x, h, d, plan) && aligned16(out0) && aligned16(out1),
               "%s: not supported for these operands (fp32, one head, d in {64, 128, 256}, a plan of these arrays with sorted "
               "rows, 16-byte-aligned outputs): run the two passes separately", fn);
  GO_CHECK_ARG(n_out >= 0 && plan->info.max_row < n_out && n_out < 0x7fffffffLL, "%s: row id %lld but the outputs have %lld rows", fn,
               (long long)plan->info.max_row, (long long)n_out);
  const size_t row_b = sizeof(float) * (size_t)d;
  if (n_out * d == 0) return GRAPHOP_OK;
  GO_PTR(fn, out0); GO_PTR(fn, out1);
  if (n_chunks == 0) {
    GO_HIP(zero_async(out0, row_b * (size_t)n_out, st));
    GO_HIP(zero_async(out1, row_b * (size_t)n_out, st));
    return GRAPHOP_OK;
  }
  GO_PTR(fn, w2); GO_PTR(fn, X0); GO_PTR(fn, X1);
  // self-zeroing as in launch_spmm: rows a lane group owns are stored, edge-less rows between chunk rows zero-stored by the
  // group that sees the gap, the tail behind the last chunk row from here; not with gaps of more than 4 MB of rows
  const bool sz = tuning().spmm_selfzero && (double)plan->info.max_row_gap * (double)d * 4.0 <= 4.0 * 1048576.0;
  i64 covered = 0;
  if (sz) {
    covered = plan->info.max_row + 1;
    if (covered < n_out) {
      GO_HIP(zero_async((char*)out0 + row_b * (size_t)covered, row_b * (size_t)(n_out - covered), st));
      GO_HIP(zero_async((char*)out1 + row_b * (size_t)covered, row_b * (size_t)(n_out - covered), st));
    }
  } else {
    GO_HIP(zero_async(out0, row_b * (size_t)n_out, st));
    GO_HIP(zero_async(out1, row_b * (size_t)n_out, st));
  }
  const int cpg = cpg_for(n_chunks, tuning().spmm_flat_cpg, (int)d);
  ProfScope prof("spmm_pair_cols", st, "k_spmm_flat2_f32");
  GO_DISPATCH_LNV((int)d, {
    if constexpr (NV == 1 && L >= 16) {
      const i64 groups = ceil_div(n_chunks, cpg);
      const unsigned nb = blocks_for(groups, GroupCfg<L>::kGroupsPerBlock);
      if (sz) {
        if (groups > 1) {
          const unsigned zb = blocks_for(groups - 1, kFastBlock / L);
          hipLaunchKernelGGL((k_zero_shared_rows<L, NV>), dim3(zb), dim3(kFastBlock), 0, st, (const i64*)row, (float*)out0, n_chunks, cpg);
          hipLaunchKernelGGL((k_zero_shared_rows<L, NV>), dim3(zb), dim3(kFastBlock), 0, st, (const i64*)row, (float*)out1, n_chunks, cpg);
        }
        hipLaunchKernelGGL((k_spmm_flat2_f32<L, true>), dim3(nb), dim3(kFastBlock), 0, st, (const i64*)row, (const i64*)indptr,
                           (const i64*)eid, (const i64*)indices, (const float2*)w2, (const float*)X0, (const float*)X1,
                           (float*)out0, (float*)out1, n_chunks, cpg, covered);
      } else {
        hipLaunchKernelGGL((k_spmm_flat2_f32<L, false>), dim3(nb), dim3(kFastBlock), 0, st, (const i64*)row, (const i64*)indptr,
                           (const i64*)eid, (const i64*)indices, (const float2*)w2, (const float*)X0, (const float*)X1,
                           (float*)out0, (float*)out1, n_chunks, cpg, (i64)0);
      }
    }
  });
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

int graphop_node_mul_edge_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                  const int64_t* eid, const void* A, const void* B, void* y,
                                  int64_t n_chunks, int64_t n_edges, int64_t n_a, int64_t h,
                                  int64_t d, const graphop_plan_t* plan, void* stream) {
  const char* fn = "node_mul_edge_forward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, d));
  hipStream_t st = (hipStream_t)stream;
  GO_TRY(check_rows(fn, "A", plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid, n_chunks, n_edges) ? plan : nullptr, n_a));
  if (n_edges * h == 0) return GRAPHOP_OK;
  GO_PTR(fn, y);
  const bool covered = plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                                    n_chunks, n_edges) &&
                       plan->info.full_coverage && plan->info.eid_identity && plan->info.indptr_monotone;
  if (!covered) GO_HIP(zero_async(y, esize(dtype) * (size_t)(n_edges * h), st));
  if (n_chunks == 0) return GRAPHOP_OK;
  GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, A); GO_PTR(fn, B);
  if (nme_fast_ok(dtype, h, d, n_edges)) {
    ProfScope prof("node_mul_edge_fwd", st, "k_nme_fwd_f32");
    const int cpg = tuning().spmm_cpg;
    GO_DISPATCH_NME(d, h, {
      const unsigned nb = blocks_for(ceil_div(n_chunks, cpg), kFastBlock / LD);
      hipLaunchKernelGGL((k_nme_fwd_f32<LD, H>), dim3(nb), dim3(kFastBlock), 0, st, (const i64*)row,
                         (const i64*)indptr, (const i64*)eid, (const float*)A, (const float*)B,
                         (float*)y, n_chunks, cpg);
    });
    GO_LAUNCH_CHECK();
    return GRAPHOP_OK;
  }
  return launch_sddmm<true>("node_mul_edge_fwd", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                            (const i64*)eid, A, B, y, n_chunks, n_edges, n_edges, h, d, nullptr, st);
}

int graphop_node_mul_edge_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                   const int64_t* eid, const void* A, const void* B,
                                   const void* dy, void* dA, void* dB, int64_t n_chunks,
                                   int64_t n_edges, int64_t n_a, int64_t h, int64_t d,
                                   const graphop_plan_t* plan, void* stream) {
  const char* fn = "node_mul_edge_backward";
  GO_TRY(check_common(fn, dtype, n_chunks, n_edges, h, d));
  hipStream_t st = (hipStream_t)stream;
  const size_t es = esize(dtype);
  const bool covered = plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                                    n_chunks, n_edges) &&
                       plan->info.full_coverage && plan->info.eid_identity && plan->info.indptr_monotone;
  GO_TRY(check_rows(fn, "A / dA", plan_matches(plan, (const i64*)row, (const i64*)indptr, (const i64*)eid, n_chunks, n_edges) ? plan : nullptr, n_a));
  if (n_a * h * d > 0) { GO_PTR(fn, dA); GO_HIP(zero_async(dA, es * (size_t)(n_a * h * d), st)); }
  if (n_edges * d > 0) {   // every edge row is written when the chunks cover all slots: skip the E*d zero-fill
    GO_PTR(fn, dB);
    if (!covered) GO_HIP(zero_async(dB, es * (size_t)(n_edges * d), st));
  }
  if (n_chunks == 0 || h * d == 0) return GRAPHOP_OK;
  GO_PTR(fn, row); GO_PTR(fn, indptr); GO_PTR(fn, eid); GO_PTR(fn, A); GO_PTR(fn, B); GO_PTR(fn, dy);
  if (nme_fast_ok(dtype, h, d, n_edges)) {   // both gradients in one streaming pass over B and dy
    ProfScope prof("node_mul_edge_bwd", st, "k_nme_bwd_f32");
    const int cpg = tuning().spmm_cpg;
    GO_DISPATCH_NME(d, h, {
      const unsigned nb = blocks_for(ceil_div(n_chunks, cpg), kFastBlock / LD);
      hipLaunchKernelGGL((k_nme_bwd_f32<LD, H>), dim3(nb), dim3(kFastBlock), 0, st, (const i64*)row,
                         (const i64*)indptr, (const i64*)eid, (const float*)A, (const float*)B,
                         (const float*)dy, (float*)dA, (float*)dB, n_chunks, cpg);
    });
    GO_LAUNCH_CHECK();
    return GRAPHOP_OK;
  }
  // kernel_0 (graphop_kernel.cu:61-73): dA[row] += sum_k dy[eid[k], j/d] * B[eid[k], j%d]
  GO_TRY(launch_spmm<true>("node_mul_edge_bwd_dA", dtype, (const i64*)row, (const i64*)indptr, (const i64*)eid,
                           (const i64*)eid, dy, B, dA, n_chunks, n_edges, n_edges, h, d, nullptr, st));
  // kernel_1 (:79-94): dB[eid[k], j] = sum_ki dy[eid[k], ki] * A[row, ki, j]
  const unsigned nb = (unsigned)ceil_div(n_chunks, kGenericWavesPerBlock);
  if (dtype == GRAPHOP_F32)
    hipLaunchKernelGGL((k_node_mul_edge_bwd_b<float>), dim3(nb), dim3(kGenericBlock), 0, st,
                       (const i64*)row, (const i64*)indptr, (const i64*)eid, (const float*)A,
                       (const float*)dy, (float*)dB, n_chunks, h, d);
  else
    hipLaunchKernelGGL((k_node_mul_edge_bwd_b<double>), dim3(nb), dim3(kGenericBlock), 0, st,
                       (const i64*)row, (const i64*)indptr, (const i64*)eid, (const double*)A,
                       (const double*)dy, (double*)dB, n_chunks, h, d);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

}  // extern "C"
