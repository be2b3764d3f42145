// Host-side helpers shared by the translation units of libgraphop_hip (graphop_hip.hip defines
// them; attention.hip uses them).  Not part of the C ABI.
#pragma once
#include "common.h"

namespace graphop {

// brackets one kernel launch with two hipEvents when profiling is enabled (graphop_profile_enable)
struct ProfScope {
  hipEvent_t t0 = nullptr, t1 = nullptr;
  hipStream_t st;
  const char* name;
  const char* kernel;   // device kernel family (string literal), may be set after construction
  ProfScope(const char* n, hipStream_t s, const char* k = "");
  ~ProfScope();
};

struct Tuning {
  int sddmm_cpg;  // chunks per lane-group, SDDMM-type kernels
  int spmm_cpg;   // chunks per lane-group, SpMM-type kernels
  int force_generic;
  int sweep;            // use the window-sweep drivers when a plan allows it
  int window_kb;        // target bytes of gathered table per window (must sit in a 4 MiB L2)
  int mall_window_kb;   // window size for tables beyond the Infinity Cache
  int max_windows;
  int sweep_min_kb;     // tables smaller than this are L2-friendly enough for the chunk drivers
  int sweep_bpc;        // resident blocks per CU for the sweep drivers (3: with staged ids 2 % faster than the 4 the kernels are compiled for)
  int sweep_k;          // vrows per lane group (0 = auto)
  int vrow_t;           // vrow length cap (0 = auto from the mean row length)
  int sweep_min_granule;  // mean slots per (row, window) below which the sweep is not worth it
  int dense_blocks;       // use the fp32-MFMA block-dense drivers when the plan found a cover
  int dense_min_fill;     // ... whose 32x32 tiles hold at least this many percent edges
  int dense_detect_min_fill;  // plan creation keeps a block cover only above this fill (percent)
  int sweep_w;            // > 0: number of column windows (overrides window_kb)
  int spmm_window_scale;  // window-owner SpMM over identity-eid slots: windows this many times window_kb
  int attn_fused;         // attention_forward/backward: use the fused window kernels when they apply
  int attn_window_scale;  // fused kernels gather 2 packed rows per slot: windows of this many times window_kb
  int attn_k;             // vrows per lane group in the fused kernels (0 = auto)
  int attn_bpc;           // resident workgroups per CU of the fused kernels
  int staged_ids;         // window-owner passes: plan-time deal + contiguous ids staged through LDS (IdStage);
                          // bit 0: SDDMM, bit 1: SpMM (both orientations), bit 2: the fused backward passes
  int attn_fwd_walk;      // attention_forward as ONE walk-style pass (kernels_attn_walk.h) where it applies (fp32, h = 1, d = 64)
  int spmm_selfzero;      // SpMM-type passes on the chunk driver with row ownership leave their output fully defined
                          // themselves (no zero fill by the entry point) ...
  int spmm_selfzero_min_mb;  // ... for outputs of at least this many MB (below, the fill is noise)
  int spmm_flat;          // row-owning chunk driver in its slot-walking form (k_spmm_flat_f32) where chunks are short ...
  int spmm_flat_max_mean; // ... i.e. below this many slots per chunk on average
  int spmm_flat_cpg;      // chunks per lane group of that form (more than spmm_cpg: a group should see several id batches)
  int spmm_flat_min_chunks;  // ... and only for chunk lists at least this long
  int attn_max_d;         // widest row (floats) the fused window passes are chosen for: beyond 64 the two-row gathers
                          // dominate and the passes measure slower than the unfused ones (d=128: 18.9 vs 16.8 ms)
  int attn_rows;          // chunk-driver fused backward: -1 = by the cost rule, 0 = never, 1 = whenever legal
  int touch_sddmm;        // SDDMM strips: per-task id-line touches (kernels_fast.h: LineTouch): bit 0 ids, bit 1 edge ids
  int walk;               // walk drivers (kernels_walk.h): bit 1 SpMM-type passes over identity-eid (row-major) slots,
                          // bit 2 SpMM-type over permuted (column-major) slots (bit 0 was the SDDMM-type walk kernel of
                          // round 3: 1.63-1.73 ms against 1.51 on the window-owner strips, removed in round 4)
  int walk_window_kb;     // bytes of gathered table per window of the walk drivers, passes over identity-eid (row-major) slots
  int walk_window_kb_col; // ... passes over permuted (column-major) slots: their per-slot scalars are a gather whose lines
                          // share the L2 with the window (Reddit shape: 2.72 ms at 4 MB, 2.25 at 2 MB; row-major 1.71 / 1.78)
  int walk_drift;         // pacing steps a wave may run ahead of the slowest wave of its XCD (0 = free-running)
  int walk_steps;         // pacing steps per column window
  int walk_min_bin;       // fewest slots per (lane group, round) bin for the walk drivers to be chosen
  int walk_debug;         // 1: every walk launch is followed by a synchronisation and a line of pacing statistics on stderr
  int walk_fault;         // tests only: inject a hand-over fault into the walk kernel (kernels_walk.h: WalkView::fault)
  int walk_blocks;        // > 0: workgroups of the walk launches (tests: a small grid makes several rounds of sizeable bins)
  int plan_trim;          // 1: plans drop what only their layout builders read (32-bit mirrors, window tables) once an op has what
                          // it launches with; rebuilt on demand (plan.hip: plan_trim)
  int n_cu;
  Tuning() {
    sweep = env_int("GRAPHOP_SWEEP", 1);
    window_kb = env_int("GRAPHOP_WINDOW_KB", 4096);
    mall_window_kb = env_int("GRAPHOP_MALL_WINDOW_KB", 32768);
    max_windows = env_int("GRAPHOP_MAX_WINDOWS", 128);
    sweep_min_kb = env_int("GRAPHOP_SWEEP_MIN_KB", 4608);
    sweep_bpc = env_int("GRAPHOP_SWEEP_BPC", 3);
    sweep_k = env_int("GRAPHOP_SWEEP_K", 0);
    vrow_t = env_int("GRAPHOP_VROW_T", 0);
    sweep_min_granule = env_int("GRAPHOP_SWEEP_MIN_GRANULE", 4);
    sweep_w = env_int("GRAPHOP_SWEEP_W", 0);
    spmm_window_scale = env_int("GRAPHOP_SPMM_WINDOW_SCALE", 2);
    dense_blocks = env_int("GRAPHOP_DENSE_BLOCKS", 1);
    dense_min_fill = env_int("GRAPHOP_DENSE_MIN_FILL", 40);
    dense_detect_min_fill = env_int("GRAPHOP_DENSE_DETECT_MIN_FILL", 10);
    attn_fused = env_int("GRAPHOP_ATTN_FUSED", 1);
    attn_fwd_walk = env_int("GRAPHOP_ATTN_FWD_WALK", 1);
    spmm_selfzero = env_int("GRAPHOP_SPMM_SELFZERO", 1);
    spmm_selfzero_min_mb = env_int("GRAPHOP_SPMM_SELFZERO_MIN_MB", 256);
    spmm_flat = env_int("GRAPHOP_SPMM_FLAT", 1);
    spmm_flat_max_mean = env_int("GRAPHOP_SPMM_FLAT_MAX_MEAN", 10);
    spmm_flat_cpg = env_int("GRAPHOP_SPMM_FLAT_CPG", 32);
    if (spmm_flat_cpg < 1) spmm_flat_cpg = 1;
    spmm_flat_min_chunks = env_int("GRAPHOP_SPMM_FLAT_MIN_CHUNKS", 1 << 20);
    attn_window_scale = env_int("GRAPHOP_ATTN_WINDOW_SCALE", 2);
    attn_k = env_int("GRAPHOP_ATTN_K", 0);
    attn_bpc = env_int("GRAPHOP_ATTN_BPC", 0);
    attn_rows = env_int("GRAPHOP_ATTN_ROWS", -1);
    attn_max_d = env_int("GRAPHOP_ATTN_MAX_D", 64);
    staged_ids = env_int("GRAPHOP_STAGED_IDS", 7);
    touch_sddmm = env_int("GRAPHOP_TOUCH_SDDMM", 1);
    walk = env_int("GRAPHOP_WALK", 6);
    walk_window_kb = env_int("GRAPHOP_WALK_WINDOW_KB", 4096);
    walk_window_kb_col = env_int("GRAPHOP_WALK_WINDOW_KB_COL", 2048);
    walk_drift = env_int("GRAPHOP_WALK_DRIFT", 3);
    walk_steps = env_int("GRAPHOP_WALK_STEPS", 2);
    walk_min_bin = env_int("GRAPHOP_WALK_MIN_BIN", 1024);
    walk_blocks = env_int("GRAPHOP_WALK_BLOCKS", 0);
    walk_debug = env_int("GRAPHOP_WALK_DEBUG", 0);
    walk_fault = 0;
    plan_trim = env_int("GRAPHOP_PLAN_TRIM", 1);
    n_cu = 256;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        prop.multiProcessorCount > 0)
      n_cu = prop.multiProcessorCount;
    sddmm_cpg = env_int("GRAPHOP_SDDMM_CPG", 8);
    spmm_cpg = env_int("GRAPHOP_SPMM_CPG", 16);
    force_generic = env_int("GRAPHOP_FORCE_GENERIC", 0);
    if (sddmm_cpg < 1) sddmm_cpg = 1;
    if (spmm_cpg < 1) spmm_cpg = 1;
  }
};
Tuning& tuning_mut();
const Tuning& tuning();

// Stream-ordered zero fill (a kernel: runtime.hip, k_zero16).
hipError_t zero_async(void* ptr, size_t bytes, hipStream_t st);

struct SweepView;   // kernels_fast.h
}  // namespace graphop

#include "kernels_fast.h"

namespace graphop {

struct SweepLaunch {
  SweepView view;
  unsigned blocks;
  size_t lds_bytes;
};

// Overrides for kernels whose gathered rows / per-vrow LDS rows are not one F-float row
// (the fused attention kernels gather two packed rows per slot).
struct SweepOpts {
  i64 row_bytes = 0;      // bytes per gathered table row (0 = 16*L*NV)
  int K = 0;              // vrows per lane group (0 = auto)
  int window_scale = 0;   // > 0: windows of this many times window_kb, vrows as many times longer
  int bpc = 0;            // resident workgroups per CU (0 = tuning().sweep_bpc)
  int dry_run = 0;        // 1: decide and build the cached structure only (no task queue is taken)
  int touch = 0;          // SweepView::touch of the launch
  int staged = 0;         // 1: also fetch / build the dealt (window-major) layout and put it in the view
  int stage_lds_per_group = 0;   // bytes of LDS each lane group needs for staging (added to lds_bytes)
  int no_eids = 0;        // 1: the kernels never read edge ids (fused attention passes): the dealt layout keeps no copy of them
};

// Decide whether the window-sweep driver applies and fetch / build its structure.
// Returns 1 = use sweep, 0 = use the chunk driver, <0 = error code (negated).
int choose_sweep(const graphop_plan* plan, i64 n_table_rows, int L, int NV, hipStream_t st,
                 SweepLaunch* out, bool accumulating = false, const SweepOpts* opts = nullptr);

// plan.hip (setup kernels and the plan's cached structures)
int partition_count(const i64*, i64, i64, i64*, hipStream_t);
int partition_fill(const i64*, const i64*, i64, i64, i64, i64*, i64*, hipStream_t);
int plan_build(graphop_plan*, i64, hipStream_t, int);
int plan_get_sweep(graphop_plan*, int, i64, int, hipStream_t, const Sweep**);
int* plan_take_queue(graphop_plan*, const Sweep*);
int plan_get_dealt(graphop_plan*, const Sweep*, int, int, hipStream_t, const Sweep::Dealt**, bool want_eids = true);
void plan_free_sweeps(graphop_plan*);
int plan_get_walk(graphop_plan*, int, i64, int, int, int, int, hipStream_t, const Walk**, bool want_widx = true);
int plan_build_seg_eptr(graphop_plan*, hipStream_t);
int* plan_take_walk_sync(graphop_plan*, const Walk*);
int plan_n_sweeps(const graphop_plan* p);
const Sweep* plan_sweep_at(const graphop_plan* p, int i);
int plan_import_arrays(graphop_plan* p, const i64* seg_chunk, const int32_t* idx32, const int32_t* eid32,
                       const int32_t* long_segs, i64 n_long, const int32_t* blk_seg,
                       const int32_t* seg_e0, const int32_t* seg_row, hipStream_t st);
int plan_import_sweep(graphop_plan* p, int W, i64 win_cols, int T, int V, const int32_t* vr_row,
                      const int32_t* wp_lo, const int32_t* wp_hi, hipStream_t st);
void plan_init_sweeps(graphop_plan*);
int plan_ensure_mirrors(graphop_plan* p, hipStream_t st);
int plan_pin_mirrors(graphop_plan* p, hipStream_t st);
int plan_ensure_mirrors_locked(graphop_plan* p, hipStream_t st);
int plan_rebuild_sweep_tables(graphop_plan* p, const Sweep* sw, hipStream_t st);
int plan_pin_sweep_tables(graphop_plan* p, const Sweep* sw, hipStream_t st);
void plan_trim(graphop_plan* p);
// the fused attention passes' window structure for ONE orientation (attention.hip); dry run
int attn_prepare_plan(const graphop_plan* plan, i64 n_table_rows, i64 d, bool col, hipStream_t st);

// Fused attention forward on the walk (graphop_hip.hip; kernels_attn_walk.h): 1 = launched, 0 = does not apply,
// < 0 = error (negated).  dry_run: only decide / build the layout and report the workspace it needs.
int attn_fwd_walk(const graphop_plan* plan, i64 n_q, i64 n_k, i64 h, i64 d, int dtype, const void* Q, const void* K,
                  const void* V, void* o, void* stats, void* ws, i64 ws_bytes, hipStream_t st, bool dry_run,
                  size_t* ws_needed);

int softmax_forward_stats(int dtype, const i64* row, const i64* indptr, const i64* eid, const void* x,
                          void* y, i64 C, i64 E, i64 h, void* ws, i64 ws_rows,
                          const graphop_plan* plan, hipStream_t st, void* stats);

inline size_t esize(int dtype) { return dtype == GRAPHOP_F64 ? 8 : 4; }
inline bool pow2(i64 v) { return v > 0 && (v & (v - 1)) == 0; }
inline i64 pow2ceil(i64 v) { i64 p = 1; while (p < v) p <<= 1; return p; }

inline bool plan_matches_full(const graphop_plan* p, const i64* row, const i64* indptr,
                              const i64* eid, const i64* indices, i64 C, i64 E) {
  return p && p->row == (const int64_t*)row && p->indptr == (const int64_t*)indptr &&
         p->eid == (const int64_t*)eid && p->indices == (const int64_t*)indices &&
         p->info.n_chunks == C && p->info.n_edges == E;
}

#define GO_DISPATCH_LNV(F, ...)                                      \
  switch (F) {                                                       \
    case 16: { constexpr int L = 4, NV = 1; __VA_ARGS__; } break;    \
    case 32: { constexpr int L = 8, NV = 1; __VA_ARGS__; } break;    \
    case 64: { constexpr int L = 16, NV = 1; __VA_ARGS__; } break;   \
    case 128: { constexpr int L = 32, NV = 1; __VA_ARGS__; } break;  \
    case 256: { constexpr int L = 64, NV = 1; __VA_ARGS__; } break;  \
    case 512: { constexpr int L = 64, NV = 2; __VA_ARGS__; } break;  \
    case 1024: { constexpr int L = 64, NV = 4; __VA_ARGS__; } break; \
    default: break;                                                  \
  }

#define GO_PTR(fn, p) GO_CHECK_ARG((p) != nullptr, "%s: " #p " is NULL", fn)
#define GO_TRY(expr)                     \
  do {                                   \
    int _rc = (expr);                    \
    if (_rc != GRAPHOP_OK) return _rc;   \
  } while (0)

}  // namespace graphop
