// libgraphop_hip: the plan entry points of the C ABI (create / info / persistence / destroy; include/graphop_hip.h).
// graphop_plan_prepare lives with the operator dispatch (graphop_hip.hip): it dry-runs the drivers' choices.
// Split from graphop_hip.hip in round 5.
#include <string.h>

#include "common.h"
#include "host.h"

using namespace graphop;

extern "C" {

int graphop_plan_create(const int64_t* row, const int64_t* indptr, const int64_t* eid,
                        const int64_t* indices, int64_t n_chunks, int64_t n_edges,
                        int64_t n_index_bound, void* stream, graphop_plan_t** plan_out) {
  GO_CHECK_ARG(plan_out != nullptr, "plan_create: plan_out is NULL");
  *plan_out = nullptr;
  GO_CHECK_ARG(indptr != nullptr && (row != nullptr || n_chunks == 0) &&
               (eid != nullptr || n_edges == 0), "plan_create: NULL pointer");
  GO_CHECK_ARG(n_chunks >= 0 && n_chunks < 0x7fffffffLL && n_edges >= 0,
               "plan_create: size out of range");
  GO_TRY(check_not_capturing((hipStream_t)stream, "plan_create"));
  graphop_plan* p = (graphop_plan*)calloc(1, sizeof(graphop_plan));
  GO_CHECK_ARG(p != nullptr, "plan_create: out of host memory");
  p->row = row; p->indptr = indptr; p->eid = eid; p->indices = indices;
  p->info.n_chunks = n_chunks;
  p->info.n_edges = n_edges;
  (void)hipGetDevice(&p->device);
  plan_init_sweeps(p);
  int rc = plan_build(p, n_index_bound, (hipStream_t)stream, tuning().dense_detect_min_fill);
  if (rc == GRAPHOP_OK) rc = plan_build_seg_eptr(p, (hipStream_t)stream);
  if (rc != GRAPHOP_OK) {
    graphop_plan_destroy(p);
    return rc;
  }
  *plan_out = p;
  return GRAPHOP_OK;
}

int graphop_plan_info(const graphop_plan_t* plan, graphop_plan_info_t* info_out) {
  GO_CHECK_ARG(plan && info_out, "plan_info: NULL pointer");
  *info_out = plan->info;
  return GRAPHOP_OK;
}

int graphop_plan_n_sweeps(const graphop_plan_t* plan) { return plan ? plan_n_sweeps(plan) : 0; }

int graphop_plan_sweep_info(const graphop_plan_t* plan, int sweep, graphop_sweep_info_t* out) {
  GO_CHECK_ARG(plan && out, "plan_sweep_info: NULL pointer");
  const Sweep* s = plan_sweep_at(plan, sweep);
  GO_CHECK_ARG(s != nullptr, "plan_sweep_info: no window structure %d", sweep);
  out->win_cols = s->win_cols; out->W = s->W; out->T = s->T; out->V = s->V; out->n_dealt = s->n_dealt;
  return GRAPHOP_OK;
}

int graphop_plan_array(const graphop_plan_t* plan, const char* name, int sweep, const void** ptr,
                       int64_t* bytes) {
  GO_CHECK_ARG(plan && name && ptr && bytes, "plan_array: NULL pointer");
  *ptr = nullptr; *bytes = 0;
  const graphop_plan_info_t& in = plan->info;
  auto give = [&](const void* p, size_t n) { if (p) { *ptr = p; *bytes = (int64_t)n; } return GRAPHOP_OK; };
  // (export is a setup path: arrays the plan dropped after building its layouts -- plan.hip: plan_trim -- are rebuilt
  // here on the default stream, so a container always holds the full derived state)
  if (sweep >= 0) {
    const Sweep* s = plan_sweep_at(plan, sweep);
    GO_CHECK_ARG(s != nullptr, "plan_array: no window structure %d", sweep);
    if (!s->wp_lo || !s->wp_hi || !s->vr_row) {
      const int rcw = plan_rebuild_sweep_tables(const_cast<graphop_plan*>(plan), s, nullptr);
      if (rcw != GRAPHOP_OK) return rcw;
    }
    if (!strcmp(name, "vr_row")) return give(s->vr_row, sizeof(int) * (size_t)s->V);
    if (!strcmp(name, "wp_lo")) return give(s->wp_lo, sizeof(int) * (size_t)s->V * s->W);
    if (!strcmp(name, "wp_hi")) return give(s->wp_hi, sizeof(int) * (size_t)s->V * s->W);
  } else {
    if (!strcmp(name, "idx32") || !strcmp(name, "eid32")) {
      const int rcm = plan_ensure_mirrors_locked(const_cast<graphop_plan*>(plan), nullptr);
      if (rcm != GRAPHOP_OK) return rcm;
    }
    if (!strcmp(name, "seg_chunk")) return give(plan->seg_chunk, sizeof(int64_t) * (size_t)(in.n_segments + 1));
    if (!strcmp(name, "idx32")) return give(plan->idx32, sizeof(int32_t) * (size_t)in.n_edges);
    if (!strcmp(name, "eid32")) return give(plan->eid32, sizeof(int32_t) * (size_t)in.n_edges);
    if (!strcmp(name, "long_segs")) return give(plan->long_segs, sizeof(int32_t) * (size_t)plan->n_long);
    if (!strcmp(name, "blk_seg")) return give(plan->blk_seg, sizeof(int32_t) * (size_t)(in.n_dense_blocks + 1));
    if (!strcmp(name, "seg_e0")) return give(plan->seg_e0, sizeof(int32_t) * (size_t)(in.n_segments + 1));
    if (!strcmp(name, "seg_row")) return give(plan->seg_row, sizeof(int32_t) * (size_t)in.n_segments);
  }
  set_error("plan_array: unknown array '%s'", name);
  return GRAPHOP_ERR_INVALID_ARGUMENT;
}

int graphop_plan_import(const int64_t* row, const int64_t* indptr, const int64_t* eid,
                        const int64_t* indices, const graphop_plan_info_t* info,
                        const int64_t* seg_chunk, const int32_t* idx32, const int32_t* eid32,
                        const int32_t* long_segs, int64_t n_long, const int32_t* blk_seg,
                        const int32_t* seg_e0, const int32_t* seg_row, void* stream,
                        graphop_plan_t** plan_out) {
  GO_CHECK_ARG(plan_out && info && indptr, "plan_import: NULL pointer");
  *plan_out = nullptr;
  GO_CHECK_ARG(info->n_chunks >= 0 && info->n_edges >= 0 && info->n_segments >= 0 && n_long >= 0,
               "plan_import: bad sizes");
  GO_TRY(check_not_capturing((hipStream_t)stream, "plan_import"));
  graphop_plan* p = (graphop_plan*)calloc(1, sizeof(graphop_plan));
  GO_CHECK_ARG(p != nullptr, "plan_import: out of host memory");
  p->row = row; p->indptr = indptr; p->eid = eid; p->indices = indices;
  p->info = *info;
  p->info.n_geometry_fallbacks = 0;   // (a live count of this plan object, not part of the persisted state)
  (void)hipGetDevice(&p->device);
  plan_init_sweeps(p);
  int rc = plan_import_arrays(p, (const i64*)seg_chunk, idx32, eid32, long_segs, n_long, blk_seg, seg_e0,
                              seg_row, (hipStream_t)stream);
  if (rc == GRAPHOP_OK) rc = plan_build_seg_eptr(p, (hipStream_t)stream);
  if (rc != GRAPHOP_OK) { graphop_plan_destroy(p); return rc; }
  *plan_out = p;
  return GRAPHOP_OK;
}

int graphop_plan_import_sweep(graphop_plan_t* plan, const graphop_sweep_info_t* info,
                              const int32_t* vr_row, const int32_t* wp_lo, const int32_t* wp_hi,
                              void* stream) {
  GO_CHECK_ARG(plan && info, "plan_import_sweep: NULL pointer");
  GO_TRY(check_not_capturing((hipStream_t)stream, "plan_import_sweep"));
  return plan_import_sweep(plan, info->W, info->win_cols, info->T, info->V, vr_row, wp_lo, wp_hi,
                           (hipStream_t)stream);
}

int graphop_plan_sweep_dealt(const graphop_plan_t* plan, int sweep, int i, int32_t* L, int32_t* K) {
  GO_CHECK_ARG(plan && L && K, "plan_sweep_dealt: NULL pointer");
  const Sweep* s = plan_sweep_at(plan, sweep);
  GO_CHECK_ARG(s != nullptr && i >= 0 && i < s->n_dealt, "plan_sweep_dealt: no layout %d of window structure %d", i, sweep);
  *L = s->dealt[i].L; *K = s->dealt[i].K;
  return GRAPHOP_OK;
}

int graphop_plan_sweep_build_dealt(graphop_plan_t* plan, const graphop_sweep_info_t* info, int32_t L,
                                   int32_t K, void* stream) {
  GO_CHECK_ARG(plan && info, "plan_sweep_build_dealt: NULL pointer");
  GO_CHECK_ARG(plan->info.has_idx32, "plan_sweep_build_dealt: the plan has no 32-bit mirrors");
  for (int i = 0; i < plan_n_sweeps(plan); ++i) {
    const Sweep* s = plan_sweep_at(plan, i);
    if (s->W == info->W && s->win_cols == info->win_cols && s->T == info->T) {
      const Sweep::Dealt* d = nullptr;
      return plan_get_dealt(plan, s, L, K, (hipStream_t)stream, &d);
    }
  }
  set_error("plan_sweep_build_dealt: no window structure W=%d T=%d", (int)info->W, (int)info->T);
  return GRAPHOP_ERR_INVALID_ARGUMENT;
}

void graphop_plan_destroy(graphop_plan_t* plan) {
  if (!plan) return;
  plan_free_sweeps(plan);
  if (plan->seg_chunk) go_free(plan->seg_chunk);
  if (plan->seg_eptr) go_free(plan->seg_eptr);
  if (plan->idx32) go_free(plan->idx32);
  if (plan->eid32) go_free(plan->eid32);
  if (plan->long_segs) go_free(plan->long_segs);
  if (plan->blk_seg) go_free(plan->blk_seg);
  if (plan->seg_e0) go_free(plan->seg_e0);
  if (plan->seg_row) go_free(plan->seg_row);
  free(plan);
}

}  // extern "C"
