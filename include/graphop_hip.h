/*
 * graphop_hip.h -- C ABI of the MI355X (gfx950) graph-attention operator library
 *                  (libgraphop_hip.so, built from custom_op_benchmark_amd/csrc/).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference exposes the
 * path as a pybind11 module `graphop` with eight functions on at::Tensor
 * (graphop/graphop.cpp:216-225).  Each entry point below replaces one of them, with the
 * tensors flattened to plain device pointers + sizes; a binding (ctypes / pybind / cgo ...)
 * recovers the reference signature exactly -- see INTEGRATION.md and
 * custom_op_benchmark_amd/graphop.py (the Python binding shipped here).
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor storage), contiguous,
 *     row-major; index arrays are int64 exactly as in the reference
 *     (graphop_kernel.cu:293-296); value arrays are `dtype` (GRAPHOP_F32 / GRAPHOP_F64, the
 *     reference's AT_DISPATCH_FLOATING_TYPES set, graphop_kernel.cu:291).
 *   - chunked CSR: row[n_chunks], indptr[n_chunks+1] as produced by partition_csr
 *     (part_csr.py:13-27); eid[n_edges], indices[n_edges] per CSR slot.
 *   - node tensors are (n, h, d) -> element (v,k,i) at ((v*h + k)*d + i); edge tensors are
 *     (n_edges, h).  h == 1 covers the reference's rank-reduced (n, d) / (n_edges) case.
 *   - outputs are caller-allocated and need NOT be initialised: the callee zero-fills them,
 *     reproducing the reference's at::zeros outputs (graphop_kernel.cu:284,379-380,429,482,
 *     527,571-572): slots no chunk covers read 0.
 *   - `stream` is a hipStream_t (NULL = default stream).  Calls enqueue work and return
 *     without synchronising, like the reference (graphop_kernel.cu:288,302).  No global state;
 *     thread-safe as long as a plan is not destroyed while in use.
 *   - return value: GRAPHOP_OK or an error code; graphop_last_error() gives the message of the
 *     calling thread's last failure (the reference throws c10::Error from AT_ASSERTM /
 *     THCudaCheck instead, graphop.cpp:4-6, graphop_kernel.cu:302).
 *   - `plan` may be NULL.  A plan (graphop_plan_create) caches per-graph derived structure
 *     (row segments, flags, 32-bit index mirrors) for one CSR orientation; with it the kernels
 *     take the row-owned fast paths.  With NULL they take the general path that is correct for
 *     ANY chunk layout (chunks of a row need not be adjacent), merging chunks of a row with
 *     float atomics like the reference's dgl::AtomicAdd (graphop/atomic.cuh:57-96).
 */
#ifndef GRAPHOP_HIP_H_
#define GRAPHOP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GRAPHOP_API __attribute__((visibility("default")))
#else
#define GRAPHOP_API
#endif

#define GRAPHOP_ABI_VERSION 7

#define GRAPHOP_F32 0
#define GRAPHOP_F64 1

#define GRAPHOP_OK 0
#define GRAPHOP_ERR_INVALID_ARGUMENT 1 /* bad size / dtype / null pointer / plan mismatch   */
#define GRAPHOP_ERR_HIP 2              /* a HIP runtime call or kernel launch failed        */
#define GRAPHOP_ERR_BAD_GRAPH 3        /* plan validation: index out of range, bad indptr   */

typedef struct graphop_plan graphop_plan_t; /* opaque */

/* Facts about one chunked-CSR orientation, filled by graphop_plan_create. */
typedef struct graphop_plan_info {
  int64_t n_chunks;
  int64_t n_edges;
  int64_t n_segments;      /* maximal runs of adjacent chunks with equal row id              */
  int64_t max_row;         /* largest row id (-1 if no chunks)                               */
  int64_t max_index;       /* largest value in indices (-1 if none / indices == NULL)        */
  int64_t max_segment_len; /* slots in the longest segment                                   */
  int32_t rows_sorted;     /* row[] non-decreasing => a row's chunks are adjacent            */
  int32_t indptr_monotone; /* indptr[] non-decreasing, within [0, n_edges]                   */
  int32_t eid_identity;    /* eid[k] == k for all k                                          */
  int32_t full_coverage;   /* indptr[0] == 0 && indptr[n_chunks] == n_edges                  */
  int32_t row_owned;       /* rows_sorted && indptr_monotone: fast paths enabled             */
  int32_t has_idx32;       /* 32-bit mirrors of eid / indices are cached                     */
  int32_t dense_fill_pct;  /* edges per 32x32 tile of the block-dense cover, in %; 0 = no cover */
  int32_t sorted_in_rows;  /* neighbour ids ascend inside every row segment (window drivers)  */
  int64_t n_dense_blocks;  /* blocks (<= 32 consecutive rows sharing one list of <= 32 ids)   */
  int64_t max_row_gap;     /* ABI 7: longest run of consecutive row ids without a chunk in front of a chunk's row   */
  int64_t n_geometry_fallbacks; /* ABI 7: passes that asked this plan for one window geometry more than it keeps (16) and
                              ran on the chunk drivers instead (2-3x slower on window-friendly shapes; also warned about
                              once per plan on stderr).  Live count: read it with graphop_plan_info after the passes. */
} graphop_plan_info_t;

GRAPHOP_API int graphop_abi_version(void);
GRAPHOP_API const char* graphop_last_error(void);
/* Device-side failures.  The walk kernels' hand-overs between worker and feeder waves are bounded spins (a launch
 * must not be able to hang the device); a spin whose bound expires does NOT fall through: the wave stores a code and
 * the launch's sequence number in a host-visible record, its workgroup aborts, and the outputs of that launch -- AND
 * of every launch that consumed them -- are invalid.  Launches are asynchronous, so, like a HIP error of a kernel, the
 * failure is seen by whoever looks next.  ABI 7: the record is STICKY -- every compute entry point looks at it first
 * and fails with GRAPHOP_ERR_HIP (message: which pass, on which device, which hand-over) for as long as it is set; only
 * graphop_check_device_errors() reports AND clears it (synchronise the stream first to learn about the launches before
 * it).  A binding should call it where results leave the library (the bundled ones do at the end of every step:
 * functions.attention_step, dist.ShardedAttention.step; bench.py before it prints).  One record per process: with
 * several devices the message names the device of the failed launch. */
GRAPHOP_API int graphop_check_device_errors(void);

/* ---- tuning knobs (also read once from the environment as GRAPHOP_<KEY>) ----------------------
 * keys: sddmm_cpg, spmm_cpg (chunks per lane group of the chunk drivers), force_generic,
 * sweep (0/1: window-owner drivers -- XCDs own column windows and waves pull (window, row tile) tasks),
 * window_kb, mall_window_kb, max_windows, sweep_min_kb, sweep_bpc, sweep_k, vrow_t, sweep_min_granule,
 * sweep_w, spmm_window_scale, staged_ids, dense_blocks (0/1: fp32-MFMA block-dense drivers when the plan found a
 * cover), dense_min_fill, dense_detect_min_fill (percent of a 32x32 tile), attn_fused (0/1),
 * attn_window_scale, attn_k, attn_bpc (fused attention kernels), touch_sddmm (per-task id-line
 * touches of the SDDMM strips), walk (bit 1 row-major SpMM-type, bit 2 column-major SpMM-type passes on
 * the walk drivers: lane groups own rows for a whole round and walk all column windows, nothing is
 * flushed per window), walk_window_kb, walk_window_kb_col, walk_drift, walk_steps, walk_min_bin,
 * walk_blocks, walk_debug, walk_fault (tests: hand-over fault injection), spmm_selfzero, spmm_selfzero_min_mb
 * (row-owning chunk driver defines every output row itself: no zero fill of outputs of at least that many MB),
 * plan_trim (0/1: plans drop builder inputs no kernel reads, see "device memory of plans"), spmm_flat (0/1), spmm_flat_max_mean, spmm_flat_min_chunks, spmm_flat_cpg (that driver in its slot-walking form below
 * that many slots per chunk on average, for chunk lists at least that long, with that many chunks per lane group).  Not thread-safe against
 * concurrent op calls; results never depend on them.  (Removed in ABI 6: sweep_mode, sweep_drift,
 * sweep_prefetch, transpose_scalars -- the paced vrow-owner sweep and the scalar transpose pre-pass,
 * both measured slower than what replaced them.) */
GRAPHOP_API int graphop_tune(const char* key, int value);
/* Every knob back to its default (the value at library load: built-in, or GRAPHOP_<KEY> from the
 * environment).  Tests that turn knobs restore them with this, never with literals. */
GRAPHOP_API int graphop_tune_reset(void);
/* Read a knob; enumerate the knob names (i = 0, 1, ...; NULL past the last one). */
GRAPHOP_API int graphop_tune_get(const char* key, int* value);
GRAPHOP_API const char* graphop_tune_key(int i);
/* Device bytes currently held through the library's allocator hook / hipMalloc: plans, their window
 * structures and id layouts, setup temporaries.  What a binding's plan cache budgets against. */
GRAPHOP_API int64_t graphop_memory_bytes(void);

/* ---- device memory of plans ------------------------------------------------------------------
 * Plans own device arrays (per orientation: 8 B per chunk and 4-8 B per edge per dealt / walk layout in use; the 32-bit
 * mirrors of the slot arrays (4-8 B per edge) and a window structure's tables (8 B x windows x rows) are builder inputs:
 * kept only while a per-batch window kernel reads them at run time, otherwise dropped once the layouts exist and
 * rebuilt on demand -- knob plan_trim.  Reddit-shape, both orientations: 2.3 GB once the 8-function step has run,
 * 3.8 GB with the fused op's layouts next to them; round 4: 5.7 GB).
 * By default they come from hipMalloc / hipFree (each a device-wide synchronisation).  A binding
 * may route them through its framework's allocator: alloc_fn(bytes, device, stream) returns a
 * device pointer usable on `stream` (NULL = out of memory), free_fn(ptr) releases it with
 * stream-ordered semantics.  The Python binding registers torch's caching allocator, so plan
 * memory shows up in (and can be reclaimed by) torch.cuda's accounting.  Pass NULL, NULL to
 * restore the default; pointers handed out earlier are still freed through the callback that
 * made them (or left to process exit once it is gone). */
typedef void* (*graphop_alloc_fn)(size_t bytes, int device, void* stream);
typedef void (*graphop_free_fn)(void* ptr);
GRAPHOP_API int graphop_set_allocator(graphop_alloc_fn alloc_fn, graphop_free_fn free_fn);

/* ---- per-kernel timing (measurement aid, off by default) -----------------------------------
 * When enabled, every hot-path kernel launch is bracketed by two hipEvents recorded on the
 * launch stream.  graphop_profile_read synchronises them, aggregates per pass tag
 * ("sddmm_fwd", "softmax_fwd", "spmm_fwd", "spmm_bwd_dedata", "spmm_bwd_dx", "softmax_bwd",
 * "sddmm_bwd_dA", "sddmm_bwd_dB", ...; output zero fills and task-queue resets are recorded under
 * "zero_fill"), clears the log and returns the number of tags. */
typedef struct graphop_profile_rec {
  char name[48];   /* pass tag */
  char kernel[48]; /* device kernel family that executed it (last launch under this tag) */
  int64_t calls;
  double total_ms;
  double min_ms;
  double max_ms;
} graphop_profile_rec_t;
GRAPHOP_API int graphop_profile_enable(int on);
GRAPHOP_API int graphop_profile_read(graphop_profile_rec_t* out, int cap);

/* ---- partition_csr (replaces part_csr.py:13-27 for device-resident indptr) -----------------
 * count: writes per-row chunk counts' exclusive prefix sum to first_chunk[n_rows+1]
 *        (first_chunk[n_rows] = C).  The caller reads C back (one 8-byte D2H copy) to size the
 *        outputs, then calls fill.  scratch: none.
 * fill:  row[C], indptr_out[C+1]. */
GRAPHOP_API int graphop_partition_csr_count(const int64_t* indptr, int64_t n_rows, int64_t chunk_size,
                                int64_t* first_chunk, void* stream);
GRAPHOP_API int graphop_partition_csr_fill(const int64_t* indptr, const int64_t* first_chunk, int64_t n_rows,
                               int64_t chunk_size, int64_t n_chunks, int64_t* row,
                               int64_t* indptr_out, void* stream);

/* ---- plan --------------------------------------------------------------------------------
 * Analyses (row, indptr, eid, indices) on the device, validates it (indices in [0,n_index_bound),
 * eid in [0,n_edges), indptr within range) and caches derived arrays.  Synchronises `stream`
 * once (setup path).  indices may be NULL (softmax / node_mul_edge only need row/indptr/eid).
 * n_index_bound <= 0 skips the range check of indices.  The arrays must stay alive and
 * unmodified while the plan is used.
 * RANGE CHECKS NEED A PLAN: an op entry point compares the plan's largest row id / neighbour id with
 * the operand sizes it is given (too few rows -> GRAPHOP_ERR_INVALID_ARGUMENT, no launch).  Called
 * with plan = NULL (or a plan of other arrays) it has nothing to compare with and keeps the
 * reference's behaviour: ids beyond an operand are out-of-bounds device accesses
 * (graphop_kernel.cu does no shape validation at all).  Bindings that cannot vouch for the ids should
 * always pass a plan; both bundled bindings do. */
GRAPHOP_API int graphop_plan_create(const int64_t* row, const int64_t* indptr, const int64_t* eid,
                        const int64_t* indices, int64_t n_chunks, int64_t n_edges,
                        int64_t n_index_bound, void* stream, graphop_plan_t** plan_out);
GRAPHOP_API int graphop_plan_info(const graphop_plan_t* plan, graphop_plan_info_t* info_out);
GRAPHOP_API void graphop_plan_destroy(graphop_plan_t* plan);

/* Build now every cached structure the ops would otherwise build on first use for node tensors of
 * n_table_rows x (h*d) values gathered through this plan (the column-window structures of the
 * SDDMM-type, SpMM-type and -- fused != 0 -- fused attention passes: fused = 1 for either side,
 * 2 when the plan is only ever the row-major side, 3 the column-major side).  After it no op call on
 * these shapes allocates or synchronises: required before capturing the ops into a HIP graph
 * (an op that would have to build one during capture fails with GRAPHOP_ERR_INVALID_ARGUMENT).
 * Also builds the walk layouts (csrc/kernels_walk.h) of the passes that take them.  Walk layouts are
 * not part of the persistence interface below: a plan re-created by graphop_plan_import rebuilds them
 * on first use (or in graphop_plan_prepare) from the imported arrays, a few milliseconds. */
GRAPHOP_API int graphop_plan_prepare(graphop_plan_t* plan, int dtype, int64_t n_table_rows, int64_t h,
                         int64_t d, int fused, void* stream);

/* ---- plan persistence (graph container) ---------------------------------------
 * A plan's derived arrays can be read out and a plan re-created from them WITHOUT analysing the
 * graph again (graphs.save_graph / load_graph keep them next to the eight index arrays).
 * graphop_plan_array: device pointer and byte size of one array.  sweep < 0: "seg_chunk" (int64
 * [n_segments+1]), "idx32", "eid32" (int32 [n_edges]), "long_segs" (int32), "blk_seg", "seg_e0",
 * "seg_row" (int32; block-dense cover); sweep = i >= 0: "vr_row" (int32 [V]), "wp_lo", "wp_hi"
 * (int32 [W*V]) of the i-th window structure.  Absent arrays give NULL / 0.
 * graphop_plan_import trusts its inputs (they come from an export of the same graph); all
 * pointers are device pointers and are copied. */
typedef struct graphop_sweep_info {
  int64_t win_cols; /* neighbour ids per column window */
  int32_t W;        /* windows */
  int32_t T;        /* row pieces are at most T slots long */
  int32_t V;        /* row pieces ("vrows") */
  int32_t n_dealt;  /* window-major id layouts built on this structure so far (export: see below) */
} graphop_sweep_info_t;
GRAPHOP_API int graphop_plan_n_sweeps(const graphop_plan_t* plan);
GRAPHOP_API int graphop_plan_sweep_info(const graphop_plan_t* plan, int sweep, graphop_sweep_info_t* out);
GRAPHOP_API int graphop_plan_array(const graphop_plan_t* plan, const char* name, int sweep, const void** ptr,
                       int64_t* bytes);
GRAPHOP_API int graphop_plan_import(const int64_t* row, const int64_t* indptr, const int64_t* eid,
                        const int64_t* indices, const graphop_plan_info_t* info,
                        const int64_t* seg_chunk, const int32_t* idx32, const int32_t* eid32,
                        const int32_t* long_segs, int64_t n_long, const int32_t* blk_seg,
                        const int32_t* seg_e0, const int32_t* seg_row, void* stream,
                        graphop_plan_t** plan_out);
GRAPHOP_API int graphop_plan_import_sweep(graphop_plan_t* plan, const graphop_sweep_info_t* info,
                              const int32_t* vr_row, const int32_t* wp_lo, const int32_t* wp_hi,
                              void* stream);
/* The window-owner kernels read their neighbour ids from a window-major copy ("dealt layout": the
 * granules of every task dealt to the lane groups once, each group's ids one contiguous run), one
 * per lane-group geometry (L lanes per group, K row pieces per group).  It is a pure function of
 * the window structure and the 32-bit mirrors, so a container stores only the (L, K) pairs:
 * graphop_plan_sweep_dealt reads the i-th pair of a structure, graphop_plan_sweep_build_dealt
 * re-creates the layout on the (imported) structure that matches `info`. */
GRAPHOP_API int graphop_plan_sweep_dealt(const graphop_plan_t* plan, int sweep, int i, int32_t* L, int32_t* K);
GRAPHOP_API int graphop_plan_sweep_build_dealt(graphop_plan_t* plan, const graphop_sweep_info_t* info, int32_t L,
                                   int32_t K, void* stream);

/* ---- SDDMM: maskedmm_csr_forward(row, indptr, eid, indices, A, B) -> y ----------------------
 * replaces graphop.cpp:16-30 / graphop_kernel.cu:269-304 (kernel :40-55).
 * y[eid[j], k] = <A[row[c], k, :], B[indices[j], k, :]> for every slot j of every chunk c.
 * A: (n_a,h,d)  B: (n_b,h,d)  y: (n_edges,h). */
GRAPHOP_API int graphop_maskedmm_csr_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                 const int64_t* eid, const int64_t* indices, const void* A,
                                 const void* B, void* y, int64_t n_chunks, int64_t n_edges,
                                 int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                 const graphop_plan_t* plan, void* stream);

/* ---- SDDMM over a SUBSET of the slots into a shared result array (ABI 7; not in the reference) -------------
 * Same arithmetic as graphop_maskedmm_csr_forward -- y[eid[j], k] = <A[row[c], k, :], B[indices[j], k, :]> for every
 * slot j of every chunk c (graphop_kernel.cu:45-52) -- but eid / indices hold n_slots entries of a SUB-GRAPH whose eid
 * values index a result array of n_y >= n_slots entries, and NOTHING is zero-filled: exactly the entries eid[] names
 * are written, each once.  Several chunk lists over disjoint slot sets can therefore compose one SDDMM into one y with
 * no fill and no add (each edge score is written exactly once, graphop_kernel.cu:51).  The sharded step uses it to run
 * the own-column half of its edges while the halo rows of B are still in flight (custom_op_benchmark_amd/dist.py).
 * No plan: graphop_plan_create bounds eid by the slot count; ids are not range-checked (as for any plan-less call). */
GRAPHOP_API int graphop_maskedmm_csr_forward_partial(int dtype, const int64_t* row, const int64_t* indptr,
                                         const int64_t* eid, const int64_t* indices, const void* A,
                                         const void* B, void* y, int64_t n_chunks, int64_t n_slots,
                                         int64_t n_y, int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                         void* stream);

/* ---- maskedmm_csr_backward(row, indptr_r, eid_r, indices_r, col, indptr_c, eid_c, indices_c,
 *                            A, B, dy) -> [dA, dB]
 * replaces graphop.cpp:108-131 / graphop_kernel.cu:355-409 (kernel :100-112, launched twice).
 * dA[row[c]]  += sum_k dy[eid_r[k]] * B[indices_r[k]]   (row-major CSR)
 * dB[col[c]]  += sum_k dy[eid_c[k]] * A[indices_c[k]]   (column-major CSR)
 * dA may be NULL when n_row_chunks == 0, dB when n_col_chunks == 0 (that half is skipped: the op can
 * be run one orientation at a time). */
GRAPHOP_API int graphop_maskedmm_csr_backward(int dtype, const int64_t* row, const int64_t* indptr_r,
                                  const int64_t* eid_r, const int64_t* indices_r,
                                  const int64_t* col, const int64_t* indptr_c,
                                  const int64_t* eid_c, const int64_t* indices_c, const void* A,
                                  const void* B, const void* dy, void* dA, void* dB,
                                  int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_edges,
                                  int64_t n_a, int64_t n_b, int64_t h, int64_t d,
                                  const graphop_plan_t* plan_r, const graphop_plan_t* plan_c,
                                  void* stream);

/* ---- sparse_softmax_forward(row, indptr, eid, x) -> y ---------------------------------------
 * replaces graphop.cpp:59-69 / graphop_kernel.cu:411-463 (kernels :170-202).
 * Per (row r, head t): m = max(-1e9, max x), y = exp(x-m) / sum exp(x-m) over all slots of all
 * chunks with row[c] == r (the -1e9 floor is the reference's max_val fill, :428).
 * workspace: only used when plan is NULL or not row_owned: 2*workspace_rows*h values of `dtype`,
 * workspace_rows > max(row[]) (the reference sizes it by n_edges, :420,426-427). */
GRAPHOP_API int graphop_sparse_softmax_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                   const int64_t* eid, const void* x, void* y, int64_t n_chunks,
                                   int64_t n_edges, int64_t h, void* workspace,
                                   int64_t workspace_rows, const graphop_plan_t* plan,
                                   void* stream);

/* ---- sparse_softmax_backward(row, indptr, eid, y, dy) -> dx ---------------------------------
 * replaces graphop.cpp:163-175 / graphop_kernel.cu:465-507 (kernels :208-230).
 * g[r,t] = sum dy*y ; dx = dy*y - g*y.  workspace: workspace_rows*h values (general path). */
GRAPHOP_API int graphop_sparse_softmax_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                    const int64_t* eid, const void* y, const void* dy, void* dx,
                                    int64_t n_chunks, int64_t n_edges, int64_t h,
                                    void* workspace, int64_t workspace_rows,
                                    const graphop_plan_t* plan, void* stream);

/* ---- vector_spmm_forward(row, indptr, eid, indices, edata, x) -> y --------------------------
 * replaces graphop.cpp:79-93 / graphop_kernel.cu:509-542 (kernel :118-130).
 * y[row[c], k, :] += sum_j edata[eid[j], k] * x[indices[j], k, :];  y: (n_y,h,d) where the
 * reference uses n_y = n_x (zeros_like(x), :527). */
GRAPHOP_API int graphop_vector_spmm_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                const int64_t* eid, const int64_t* indices, const void* edata,
                                const void* x, void* y, int64_t n_chunks, int64_t n_edges,
                                int64_t n_x, int64_t n_y, int64_t h, int64_t d,
                                const graphop_plan_t* plan, void* stream);

/* ---- vector_spmm_backward(row, indptr, eid, indices, col, indptr_t, eid_t, indices_t,
 *                           edata, dy, x) -> [dedata, dx]   (NB: dy before x)
 * replaces graphop.cpp:190-214 / graphop_kernel.cu:544-600 (kernels :135-163).
 * dedata[eid[j], k] = <dy[row[c], k, :], x[indices[j], k, :]>           (row-major CSR)
 * dx[col[c], k, :] += sum_j edata[eid_t[j], k] * dy[indices_t[j], k, :]  (column-major CSR)
 * Every column chunk is processed (the reference sizes that grid by the ROW chunk count,
 * graphop_kernel.cu:566,588 -- a latent bug when the counts differ).
 * dx may be NULL when n_col_chunks == 0 (that half is skipped: the op can be run one orientation at a time). */
GRAPHOP_API int graphop_vector_spmm_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                 const int64_t* eid, const int64_t* indices, const int64_t* col,
                                 const int64_t* indptr_t, const int64_t* eid_t,
                                 const int64_t* indices_t, const void* edata, const void* dy,
                                 const void* x, void* dedata, void* dx, int64_t n_row_chunks,
                                 int64_t n_col_chunks, int64_t n_edges, int64_t n_x, int64_t n_dy,
                                 int64_t h, int64_t d, const graphop_plan_t* plan_r,
                                 const graphop_plan_t* plan_c, void* stream);

/* ---- two SpMM-type passes over ONE chunked CSR as one launch (ABI 7; not in the reference) ---------------
 *   out0[row[c], :] += sum_j w2[eid[j], 0] * X0[indices[j], :]      out1[row[c], :] += sum_j w2[eid[j], 1] * X1[indices[j], :]
 * -- the arithmetic of two vector_spmm_forward-type kernel launches (graphop_kernel.cu:118-130; in the backward of the
 * composed step: dV = SpMM(a, dO) and dK = SpMM(ds, Q) over the column-major CSR, :151-163 and :100-112) that share
 * their slot list.  w2 is (n_edges, 2): a slot's two weights are ONE 8-byte read, and ids / edge ids / chunk metadata
 * are streamed once.  The sharded step uses it for its column-major side, whose per-slot weights are a random gather
 * (custom_op_benchmark_amd/dist.py).  X0, X1: (n_x, d); out0, out1: (n_out, d), need not be initialised.
 * Supported (graphop_spmm_pair_supported != 0): fp32, one head, d in {64, 128, 256}, a plan of these arrays with sorted
 * rows, 16-byte-aligned outputs; otherwise GRAPHOP_ERR_INVALID_ARGUMENT -- run the two passes separately.
 * Measured on the column side of a papers100M-shape 1/8 shard (tools/pair_columns_experiment.py): 56.9 ms for the two
 * separate launches, 52.4 for this one; writing the pairs in the CSR's slot order first so that the weights stream (built,
 * measured, removed) makes the launch 47.4 ms but the 200 M scattered 8-byte stores cost 8.3. */
/* out2[i] = (w0[i], w1[i]), i < n: two fp32 per-edge arrays interleaved into the (n, 2) pairs graphop_spmm_pair reads
 * (16-byte-aligned arrays; a streaming kernel). */
GRAPHOP_API int graphop_interleave_pairs(int dtype, const void* w0, const void* w1, void* out2, int64_t n, void* stream);
GRAPHOP_API int graphop_spmm_pair_supported(int dtype, int64_t n_chunks, int64_t n_edges, int64_t n_x, int64_t h,
                                int64_t d, const graphop_plan_t* plan);
GRAPHOP_API int graphop_spmm_pair(int dtype, const int64_t* row, const int64_t* indptr, const int64_t* eid,
                      const int64_t* indices, const void* w2, const void* X0, const void* X1, void* out0,
                      void* out1, int64_t n_chunks, int64_t n_edges, int64_t n_x, int64_t n_out, int64_t h,
                      int64_t d, const graphop_plan_t* plan, void* stream);
/* ---- node_mul_edge_forward(row, indptr, eid, A, B) -> y -------------------------------------
 * replaces graphop.cpp:39-51 / graphop_kernel.cu:235-266 (kernel :19-34).
 * y[eid[j], k] = <A[row[c], k, :], B[eid[j], :]>;  B: (n_edges, d) shared by all heads. */
GRAPHOP_API int graphop_node_mul_edge_forward(int dtype, const int64_t* row, const int64_t* indptr,
                                  const int64_t* eid, const void* A, const void* B, void* y,
                                  int64_t n_chunks, int64_t n_edges, int64_t n_a, int64_t h,
                                  int64_t d, const graphop_plan_t* plan, void* stream);

/* ---- node_mul_edge_backward(row, indptr, eid, A, B, dy) -> [dA, dB] -------------------------
 * replaces graphop.cpp:141-154 / graphop_kernel.cu:306-351 (kernels :61-94).
 * dA[row[c], k, i] += sum_j dy[eid[j], k] * B[eid[j], i];  dB[eid[j], i] = sum_k dy[eid[j],k]*A[row[c],k,i] */
GRAPHOP_API int graphop_node_mul_edge_backward(int dtype, const int64_t* row, const int64_t* indptr,
                                   const int64_t* eid, const void* A, const void* B,
                                   const void* dy, void* dA, void* dB, int64_t n_chunks,
                                   int64_t n_edges, int64_t n_a, int64_t h, int64_t d,
                                   const graphop_plan_t* plan, void* stream);

/* ---- fused attention step (EXTRA op, not one of the reference's eight) --------------------------
 * The composition the reference harness chains by hand -- MaskedMMCSR -> SparseSoftmax -> VectorSPMM
 * (wrapper.py:20-30, 8-18, 44-55) -- as one forward and one backward entry, so that the E-sized
 * intermediates s, a, da, ds never leave the library:
 *   forward : o[r]  = sum_j a[r,j] V[j],  a = row-softmax(<Q[r], K[j]>) over the row-major CSR;
 *             stats[(r*h + k)*2 + {0,1}] = (row max m, 1 / sum exp(s - m))   (n_q, h, 2)
 *   backward: dQ, dK, dV for a given dO, from (Q, K, V, o, stats): a and ds are recomputed per slot
 *             (a = exp(s - m) / sum, ds = a (<dO_r, V_j> - <dO_r, o_r>)).
 * Results equal the composition of the unfused entry points up to fp32 summation order.
 * Round 4: where it applies (fp32, h == 1, d == 64, a walkable row-major plan with identity eid) the forward is ONE
 * kernel -- a walk-style pass with an online softmax (csrc/kernels_attn_walk.h) -- and s / a are never materialised.
 * workspace: device scratch of graphop_attention_workspace_bytes(...) bytes (forward: piece records of the rows the
 * walk's bins share, a few MB, for the one-pass form; s and a for the composed form;
 * backward: packed operand tables when the fused passes apply -- fp32, h == 1, both plans given;
 * window-owner drivers for sweepable plans and tables beyond the L2, chunk drivers otherwise --
 * else s, a, da, ds of the composed path). */
GRAPHOP_API int graphop_attention_workspace_bytes(int dtype, int backward, int64_t n_edges, int64_t n_q,
                                      int64_t n_k, int64_t h, int64_t d,
                                      const graphop_plan_t* plan_r, const graphop_plan_t* plan_c,
                                      void* stream, int64_t* bytes_out);
/* 1 when graphop_attention_backward will run its fused passes (window-owner or chunk-driver form) for
 * these shapes / plans, 0 when
 * it will compose the unfused entry points (and recompute s and a first): a caller that can keep a
 * from its forward (the Python autograd class does) then prefers the unfused backward ops. */
GRAPHOP_API int graphop_attention_backward_is_fused(int dtype, int64_t n_edges, int64_t n_q, int64_t n_k, int64_t h,
                                        int64_t d, const graphop_plan_t* plan_r,
                                        const graphop_plan_t* plan_c, void* stream, int* fused_out);
GRAPHOP_API int graphop_attention_forward(int dtype, const int64_t* row, const int64_t* indptr, const int64_t* eid,
                              const int64_t* indices, const void* Q, const void* K, const void* V,
                              void* o, void* stats, int64_t n_chunks, int64_t n_edges, int64_t n_q,
                              int64_t n_k, int64_t h, int64_t d, void* workspace,
                              int64_t workspace_bytes, const graphop_plan_t* plan, void* stream);
GRAPHOP_API int graphop_attention_backward(int dtype, const int64_t* row, const int64_t* indptr_r,
                               const int64_t* eid_r, const int64_t* indices_r, const int64_t* col,
                               const int64_t* indptr_c, const int64_t* eid_c, const int64_t* indices_c,
                               const void* Q, const void* K, const void* V, const void* o,
                               const void* stats, const void* dO, void* dQ, void* dK, void* dV,
                               int64_t n_row_chunks, int64_t n_col_chunks, int64_t n_edges,
                               int64_t n_q, int64_t n_k, int64_t h, int64_t d, void* workspace,
                               int64_t workspace_bytes, const graphop_plan_t* plan_r,
                               const graphop_plan_t* plan_c, void* stream);

/* ---- halo pack / unpack of the node-range sharded step (not in the reference: single GPU) --------
 * gather_rows:      dst[i, :] = src[idx[i], :]          (send buffer of the rows peers gather from)
 * scatter_add_rows: dst[idx[i], :] += src[i, :]         (partial gradient rows coming home; idx may
 *                   repeat, native float atomics).  idx values must lie in [0, n_rows) (not checked:
 *                   the caller built them from its own range, custom_op_benchmark_amd/dist.py). */
GRAPHOP_API int graphop_gather_rows(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                        int64_t n_src_rows, int64_t row_elems, void* stream);
/* add_rows_unique: the same as scatter_add_rows for an idx run WITHOUT repeats (the rows served to ONE
 * peer are distinct): plain read-add-write at streaming rate instead of memory-side atomics; runs for
 * different peers must be issued one after the other on the same stream. */
GRAPHOP_API int graphop_add_rows_unique(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                            int64_t n_dst_rows, int64_t row_elems, void* stream);
GRAPHOP_API int graphop_scatter_add_rows(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                             int64_t n_dst_rows, int64_t row_elems, void* stream);
/* add_rows_grouped (ABI 6): the received rows grouped by the own row they belong to --
 * dst[grp_rows[g], :] += sum over p in [grp_ptr[g], grp_ptr[g + 1]) of src[grp_pos[p], :] -- ONE launch for the
 * rows of all peers (add_rows_unique needs one per peer), every own row read and written once, no atomics,
 * fixed summation order.  grp_rows must hold distinct rows; the grouping is built once per shard (dist.py). */
GRAPHOP_API int graphop_add_rows_grouped(int dtype, const void* src, const int64_t* grp_ptr, const int64_t* grp_rows,
                             const int64_t* grp_pos, void* dst, int64_t n_groups, int64_t n_dst_rows,
                             int64_t row_elems, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHOP_HIP_H_ */
