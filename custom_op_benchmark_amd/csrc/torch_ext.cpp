// graphop_cpp: the reference's operator boundary as a compiled PyTorch-ROCm C++ extension.
//
// Same shape as the reference's graphop/graphop.cpp (:1-225): CHECK_CUDA / CHECK_CONTIGUOUS on every
// input (:4-6), eight functions on at::Tensor with the reference's positional signatures and return
// types (:16-30, :39-51, :59-69, :79-93, :108-131, :141-154, :163-175, :190-214), exported twice:
//   * PYBIND11_MODULE  -> importable module (the reference's only registration, :216-225)
//   * TORCH_LIBRARY(graphop, ...) -> torch.ops.graphop.* (north_star's surface; the reference has none)
// Where the reference forwards to its *_cuda_* launchers, this file forwards to the C ABI of
// libgraphop_hip.so (include/graphop_hip.h) on the current HIP stream.  It owns a per-graph plan cache
// (graphop_plan_create is the setup path: validation + derived arrays) under the same byte budget as
// the ctypes binding's; everything else --
// kernels, dispatch, workspaces' layout -- lives behind the C ABI.  The ctypes binding
// (custom_op_benchmark_amd/graphop.py) is the same boundary without a compiler.
#include <torch/extension.h>
#include <torch/library.h>
#include <c10/hip/HIPStream.h>

#include <algorithm>
#include <list>
#include <memory>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "graphop_hip.h"

#define CHECK_CUDA(x) TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor")            // graphop.cpp:4
#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")   // graphop.cpp:5
#define CHECK_INPUT(x) CHECK_CUDA(x); CHECK_CONTIGUOUS(x)                                // graphop.cpp:6
#define CHECK_INDEX(x) TORCH_CHECK((x).scalar_type() == at::kLong, "expected scalar type Long but found ", (x).scalar_type(), " (" #x ")")
// every value operand of one call has one dtype (the reference's data<scalar_t>() throws otherwise): the C ABI takes
// raw pointers plus ONE dtype code, so a mismatch here would be an out-of-bounds access on the device
#define CHECK_SAME_DTYPE(a, b) TORCH_CHECK((a).scalar_type() == (b).scalar_type(), "expected " #a " and " #b " to have the same dtype, got ", (a).scalar_type(), " and ", (b).scalar_type())
#define CHECK_EDGE_ROWS(x, e) TORCH_CHECK((x).dim() >= 1 && (x).size(0) >= (e), #x " must hold one entry per edge id: ", (x).size(0), " rows for ", (e), " edges")

namespace {

void check(int rc) { TORCH_CHECK(rc == GRAPHOP_OK, "graphop: ", graphop_last_error()); }

int dtype_code(const at::Tensor& t) {
  if (t.scalar_type() == at::kFloat) return GRAPHOP_F32;
  if (t.scalar_type() == at::kDouble) return GRAPHOP_F64;
  TORCH_CHECK(false, "graphop: not implemented for '", t.scalar_type(), "' (float32 / float64 only)");   // AT_DISPATCH_FLOATING_TYPES, graphop_kernel.cu:291
}

void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

const int64_t* ip(const at::Tensor& t) { return t.numel() ? t.data_ptr<int64_t>() : nullptr; }
void* vp(const at::Tensor& t) { return t.numel() ? t.data_ptr() : nullptr; }

// ---- plans: least-recently-used cache keyed by the identity and version of the index tensors ----------
// Two levels like the ctypes binding (_lib.get_plan): an entry per (row, indptr, eid) holds the plans of
// that orientation by `indices` tensor; a request without indices (softmax, node_mul_edge) reuses any
// of them.  Entries are handed out as shared_ptr: an eviction by another thread never destroys a plan
// an op is still using.  Budget: kMaxGraphs entries AND the library's device-byte count
// (graphop_memory_bytes: plans of BOTH bindings) against GRAPHOP_PLAN_CACHE_GB (default 96).
struct PlanKey {
  const void *row, *indptr, *eid;
  int64_t n_chunks, n_edges;
  uint32_t v0, v1, v2;
  int device;
  bool operator==(const PlanKey& o) const {
    return row == o.row && indptr == o.indptr && eid == o.eid && n_chunks == o.n_chunks && n_edges == o.n_edges &&
           v0 == o.v0 && v1 == o.v1 && v2 == o.v2 && device == o.device;
  }
};
struct PlanKeyHash {
  size_t operator()(const PlanKey& k) const {
    size_t h = std::hash<const void*>()(k.row);
    for (const void* p : {k.indptr, k.eid}) h = h * 1000003u ^ std::hash<const void*>()(p);
    return h ^ (size_t)k.n_edges ^ ((size_t)k.v0 << 7) ^ ((size_t)k.v2 << 13);
  }
};
struct PlanEntry {
  graphop_plan_t* plan = nullptr;
  graphop_plan_info_t info;
  const void* indices = nullptr;
  uint32_t v_indices = 0;
  std::vector<at::Tensor> keep;   // the arrays the plan points into stay alive with it
  ~PlanEntry() { if (plan) graphop_plan_destroy(plan); }
};
using PlanRef = std::shared_ptr<PlanEntry>;
struct GraphEntry {
  std::vector<PlanRef> plans;     // one per indices tensor (usually one)
  std::list<PlanKey>::iterator lru;
};
std::mutex g_mu;
std::unordered_map<PlanKey, GraphEntry, PlanKeyHash> g_graphs;
std::list<PlanKey> g_lru;
constexpr size_t kMaxGraphs = 64;
int64_t cache_budget_bytes() {
  static const int64_t b = [] {
    const char* v = getenv("GRAPHOP_PLAN_CACHE_GB");
    return (int64_t)((v && *v ? atof(v) : 96.0) * (double)(1LL << 30));
  }();
  return b;
}

PlanRef get_plan(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                 const at::Tensor* indices, int64_t bound) {
  PlanKey k{row.numel() ? row.data_ptr() : nullptr, indptr.data_ptr(), eid.numel() ? eid.data_ptr() : nullptr,
            row.numel(), eid.numel(), (uint32_t)row._version(), (uint32_t)indptr._version(), (uint32_t)eid._version(),
            (int)indptr.get_device()};
  const void* ix = indices && indices->numel() ? indices->data_ptr() : nullptr;
  PlanRef found;
  std::vector<PlanRef> evicted;   // destroyed after the lock is released (plan destruction frees device memory)
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_graphs.find(k);
    if (it != g_graphs.end()) {
      g_lru.splice(g_lru.begin(), g_lru, it->second.lru);
      for (auto& p : it->second.plans)
        if (!indices || (p->indices == ix && p->v_indices == (uint32_t)indices->_version())) { found = p; break; }
    }
    if (!found) {
      while (!g_lru.empty() && (g_graphs.size() >= kMaxGraphs || graphop_memory_bytes() > cache_budget_bytes())) {
        if (it != g_graphs.end() && g_lru.back() == k) break;      // never the graph being served
        auto victim = g_graphs.find(g_lru.back());
        for (auto& p : victim->second.plans) evicted.push_back(std::move(p));
        g_graphs.erase(victim);
        g_lru.pop_back();
        bool any_left = false;
        for (auto& e : evicted) any_left |= e.use_count() > 1;
        if (any_left) break;   // still in use elsewhere: its memory will not come back by evicting more
      }
    }
  }
  evicted.clear();
  if (!found) {
    auto e = std::make_shared<PlanEntry>();
    check(graphop_plan_create(ip(row), ip(indptr), ip(eid), indices ? ip(*indices) : nullptr, row.numel(), eid.numel(),
                              bound, stream_of(indptr), &e->plan));   // (not under g_mu: the allocator hook may need the GIL)
    check(graphop_plan_info(e->plan, &e->info));
    e->indices = ix;
    e->v_indices = indices ? (uint32_t)indices->_version() : 0;
    e->keep = {row, indptr, eid};
    if (indices) e->keep.push_back(*indices);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_graphs.find(k);
    if (it == g_graphs.end()) {
      g_lru.push_front(k);
      GraphEntry ge;
      ge.lru = g_lru.begin();
      it = g_graphs.emplace(k, std::move(ge)).first;
    }
    if (indices) {   // a plan with indices supersedes an index-less one of the same orientation
      auto& v = it->second.plans;
      v.erase(std::remove_if(v.begin(), v.end(), [](const PlanRef& p) { return p->indices == nullptr; }), v.end());
    }
    it->second.plans.push_back(e);
    found = e;
  }
  TORCH_CHECK(!indices || bound <= 0 || found->info.max_index < bound, "graphop: indices holds ", found->info.max_index,
              " but the gathered tensor has only ", bound, " rows");
  return found;
}
PlanRef get_plan(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid, const at::Tensor& indices,
                 int64_t bound) {
  return get_plan(row, indptr, eid, &indices, bound);
}
PlanRef get_plan3(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid) {
  return get_plan(row, indptr, eid, nullptr, 0);
}

struct DeviceGuard {   // the reference calls cudaSetDevice without restoring (graphop_kernel.cu:277)
  c10::DeviceGuard g;
  explicit DeviceGuard(const at::Tensor& t) : g(t.device()) {}
};

at::Tensor edge_out(const at::Tensor& like, int64_t e, int64_t h) {   // (e) if h == 1 else (e, h), graphop_kernel.cu:284
  return h == 1 ? at::empty({e}, like.options()) : at::empty({e, h}, like.options());
}

}  // namespace

// ---- the eight functions (graphop.cpp:16-214) ------------------------------------------------------
at::Tensor maskedmm_csr_forward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                const at::Tensor& indices, const at::Tensor& A, const at::Tensor& B) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(indices); CHECK_INPUT(A); CHECK_INPUT(B);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid); CHECK_INDEX(indices);
  CHECK_SAME_DTYPE(A, B);
  DeviceGuard dg(A);
  const int64_t e = eid.size(0), d = A.size(-1), h = A.dim() == 2 ? 1 : A.size(1);   // graphop_kernel.cu:282-283
  auto y = edge_out(A, e, h);
  const auto pp = get_plan(row, indptr, eid, indices, B.size(0));
  const auto& p = *pp;
  check(graphop_maskedmm_csr_forward(dtype_code(A), ip(row), ip(indptr), ip(eid), ip(indices), vp(A), vp(B), vp(y),
                                     row.size(0), e, A.size(0), B.size(0), h, d, p.plan, stream_of(A)));
  return y;
}

std::vector<at::Tensor> maskedmm_csr_backward(const at::Tensor& row, const at::Tensor& indptr_r,
                                              const at::Tensor& eid_r, const at::Tensor& indices_r,
                                              const at::Tensor& col, const at::Tensor& indptr_c,
                                              const at::Tensor& eid_c, const at::Tensor& indices_c,
                                              const at::Tensor& A, const at::Tensor& B, const at::Tensor& dy_) {
  CHECK_INPUT(row); CHECK_INPUT(indptr_r); CHECK_INPUT(eid_r); CHECK_INPUT(indices_r);
  CHECK_INPUT(col); CHECK_INPUT(indptr_c); CHECK_INPUT(eid_c); CHECK_INPUT(indices_c); CHECK_INPUT(A); CHECK_INPUT(B);
  CHECK_INDEX(row); CHECK_INDEX(indptr_r); CHECK_INDEX(eid_r); CHECK_INDEX(indices_r);
  CHECK_INDEX(col); CHECK_INDEX(indptr_c); CHECK_INDEX(eid_c); CHECK_INDEX(indices_c);
  CHECK_CUDA(dy_);
  const at::Tensor dy = dy_.contiguous();   // the reference forgets this check (graphop.cpp:120-129)
  CHECK_SAME_DTYPE(A, B); CHECK_SAME_DTYPE(A, dy); CHECK_EDGE_ROWS(dy, eid_r.size(0));
  DeviceGuard dg(A);
  const int64_t d = A.size(-1), h = dy.dim() == 2 ? dy.size(1) : 1;   // graphop_kernel.cu:373
  auto dA = at::empty_like(A), dB = at::empty_like(B);
  const auto ppr = get_plan(row, indptr_r, eid_r, indices_r, B.size(0));
  const auto ppc = get_plan(col, indptr_c, eid_c, indices_c, A.size(0));
  const auto &pr = *ppr, &pc = *ppc;
  check(graphop_maskedmm_csr_backward(dtype_code(A), ip(row), ip(indptr_r), ip(eid_r), ip(indices_r), ip(col),
                                      ip(indptr_c), ip(eid_c), ip(indices_c), vp(A), vp(B), vp(dy), vp(dA), vp(dB),
                                      row.size(0), col.size(0), eid_r.size(0), A.size(0), B.size(0), h, d, pr.plan,
                                      pc.plan, stream_of(A)));
  return {dA, dB};
}

at::Tensor node_mul_edge_forward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                 const at::Tensor& A, const at::Tensor& B) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(A); CHECK_INPUT(B);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid);
  CHECK_SAME_DTYPE(A, B);
  DeviceGuard dg(A);
  const int64_t e = eid.size(0), d = A.size(-1), h = A.dim() == 2 ? 1 : A.size(1);
  TORCH_CHECK(B.size(0) >= e && B.size(-1) == d, "node_mul_edge_forward: B must be (n_edges, d)");
  auto y = edge_out(A, e, h);
  const auto pp = get_plan3(row, indptr, eid);
  const auto& p = *pp;
  check(graphop_node_mul_edge_forward(dtype_code(A), ip(row), ip(indptr), ip(eid), vp(A), vp(B), vp(y), row.size(0), e,
                                      A.size(0), h, d, p.plan, stream_of(A)));
  return y;
}

std::vector<at::Tensor> node_mul_edge_backward(const at::Tensor& row, const at::Tensor& indptr,
                                               const at::Tensor& eid, const at::Tensor& A, const at::Tensor& B,
                                               const at::Tensor& dy_) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(A); CHECK_INPUT(B);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid);
  CHECK_CUDA(dy_);
  const at::Tensor dy = dy_.contiguous();
  CHECK_SAME_DTYPE(A, B); CHECK_SAME_DTYPE(A, dy); CHECK_EDGE_ROWS(dy, eid.size(0));
  DeviceGuard dg(A);
  const int64_t e = eid.size(0), d = A.size(-1), h = dy.dim() == 2 ? dy.size(1) : 1;
  TORCH_CHECK(B.size(0) == e && B.size(-1) == d, "node_mul_edge_backward: B must be (n_edges, d)");
  auto dA = at::empty_like(A), dB = at::empty_like(B);
  const auto pp = get_plan3(row, indptr, eid);
  const auto& p = *pp;
  check(graphop_node_mul_edge_backward(dtype_code(A), ip(row), ip(indptr), ip(eid), vp(A), vp(B), vp(dy), vp(dA),
                                       vp(dB), row.size(0), e, A.size(0), h, d, p.plan, stream_of(A)));
  return {dA, dB};
}

at::Tensor sparse_softmax_forward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                  const at::Tensor& x) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(x);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid);
  CHECK_EDGE_ROWS(x, eid.size(0));
  DeviceGuard dg(x);
  const int64_t h = x.dim() == 2 ? x.size(1) : 1;
  auto y = at::empty_like(x);
  const auto pp = get_plan3(row, indptr, eid);
  const auto& p = *pp;
  at::Tensor ws;
  int64_t ws_rows = 0;
  if (!p.info.row_owned) {   // general layout: max / sum scratch per row (the reference sizes it by E, :426-427)
    ws_rows = p.info.max_row + 1;
    ws = at::empty({2 * ws_rows * h}, x.options());
  }
  check(graphop_sparse_softmax_forward(dtype_code(x), ip(row), ip(indptr), ip(eid), vp(x), vp(y), row.size(0),
                                       eid.size(0), h, ws.defined() ? vp(ws) : nullptr, ws_rows, p.plan, stream_of(x)));
  return y;
}

at::Tensor sparse_softmax_backward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                   const at::Tensor& y, const at::Tensor& dy_) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(y);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid);
  CHECK_CUDA(dy_);
  const at::Tensor dy = dy_.contiguous();
  CHECK_SAME_DTYPE(y, dy); CHECK_EDGE_ROWS(y, eid.size(0)); CHECK_EDGE_ROWS(dy, eid.size(0));
  TORCH_CHECK(y.sizes() == dy.sizes(), "sparse_softmax_backward: y and dy must have the same shape");
  DeviceGuard dg(y);
  const int64_t h = dy.dim() == 2 ? dy.size(1) : 1;
  auto dx = at::empty_like(dy);
  const auto pp = get_plan3(row, indptr, eid);
  const auto& p = *pp;
  at::Tensor ws;
  int64_t ws_rows = 0;
  if (!p.info.row_owned) {
    ws_rows = p.info.max_row + 1;
    ws = at::empty({ws_rows * h}, y.options());
  }
  check(graphop_sparse_softmax_backward(dtype_code(y), ip(row), ip(indptr), ip(eid), vp(y), vp(dy), vp(dx), row.size(0),
                                        eid.size(0), h, ws.defined() ? vp(ws) : nullptr, ws_rows, p.plan, stream_of(y)));
  return dx;
}

at::Tensor vector_spmm_forward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                               const at::Tensor& indices, const at::Tensor& edata, const at::Tensor& x) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(indices); CHECK_INPUT(edata); CHECK_INPUT(x);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid); CHECK_INDEX(indices);
  CHECK_SAME_DTYPE(edata, x); CHECK_EDGE_ROWS(edata, eid.size(0));
  DeviceGuard dg(x);
  const int64_t h = edata.dim() == 2 ? edata.size(1) : 1, d = x.size(-1);   // graphop_kernel.cu:520
  auto y = at::empty_like(x);                                               // zeros_like(x), :527
  const auto pp = get_plan(row, indptr, eid, indices, x.size(0));
  const auto& p = *pp;
  check(graphop_vector_spmm_forward(dtype_code(x), ip(row), ip(indptr), ip(eid), ip(indices), vp(edata), vp(x), vp(y),
                                    row.size(0), eid.size(0), x.size(0), x.size(0), h, d, p.plan, stream_of(x)));
  return y;
}

std::vector<at::Tensor> vector_spmm_backward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                             const at::Tensor& indices, const at::Tensor& col,
                                             const at::Tensor& indptr_t, const at::Tensor& eid_t,
                                             const at::Tensor& indices_t, const at::Tensor& edata,
                                             const at::Tensor& dy, const at::Tensor& x) {   // NB dy before x, graphop.cpp:199-201
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(indices);
  CHECK_INPUT(col); CHECK_INPUT(indptr_t); CHECK_INPUT(eid_t); CHECK_INPUT(indices_t);
  CHECK_INPUT(edata); CHECK_INPUT(dy); CHECK_INPUT(x);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid); CHECK_INDEX(indices);
  CHECK_INDEX(col); CHECK_INDEX(indptr_t); CHECK_INDEX(eid_t); CHECK_INDEX(indices_t);
  CHECK_SAME_DTYPE(edata, x); CHECK_SAME_DTYPE(dy, x); CHECK_EDGE_ROWS(edata, eid.size(0));
  DeviceGuard dg(x);
  const int64_t h = edata.dim() == 2 ? edata.size(1) : 1, d = x.size(-1);
  auto dedata = at::empty_like(edata), dx = at::empty_like(x);
  const auto ppr = get_plan(row, indptr, eid, indices, x.size(0));
  const auto ppc = get_plan(col, indptr_t, eid_t, indices_t, dy.size(0));
  const auto &pr = *ppr, &pc = *ppc;
  check(graphop_vector_spmm_backward(dtype_code(x), ip(row), ip(indptr), ip(eid), ip(indices), ip(col), ip(indptr_t),
                                     ip(eid_t), ip(indices_t), vp(edata), vp(dy), vp(x), vp(dedata), vp(dx),
                                     row.size(0), col.size(0), eid.size(0), x.size(0), dy.size(0), h, d, pr.plan,
                                     pc.plan, stream_of(x)));
  return {dedata, dx};   // graphop_kernel.cu:599
}

// ---- the extra fused op (include/graphop_hip.h: graphop_attention_*) ------------------------------
std::vector<at::Tensor> attention_forward(const at::Tensor& row, const at::Tensor& indptr, const at::Tensor& eid,
                                          const at::Tensor& indices, const at::Tensor& Q, const at::Tensor& K,
                                          const at::Tensor& V) {
  CHECK_INPUT(row); CHECK_INPUT(indptr); CHECK_INPUT(eid); CHECK_INPUT(indices); CHECK_INPUT(Q); CHECK_INPUT(K); CHECK_INPUT(V);
  CHECK_INDEX(row); CHECK_INDEX(indptr); CHECK_INDEX(eid); CHECK_INDEX(indices);
  TORCH_CHECK(K.sizes() == V.sizes() && Q.sizes().slice(1) == K.sizes().slice(1),
              "attention_forward: Q (n_q,[h,]d), K and V (n_k,[h,]d) expected");
  CHECK_SAME_DTYPE(Q, K); CHECK_SAME_DTYPE(Q, V);
  DeviceGuard dg(Q);
  const int64_t e = eid.size(0), d = Q.size(-1), h = Q.dim() == 2 ? 1 : Q.size(1), n_q = Q.size(0), n_k = K.size(0);
  auto o = at::empty_like(Q);
  auto stats = at::empty({n_q, h, 2}, Q.options());
  const auto pp = get_plan(row, indptr, eid, indices, n_k);
  const auto& p = *pp;
  int64_t nbytes = 0;
  check(graphop_attention_workspace_bytes(dtype_code(Q), 0, e, n_q, n_k, h, d, p.plan, nullptr, stream_of(Q), &nbytes));
  auto ws = at::empty({std::max<int64_t>(nbytes, 1)}, Q.options().dtype(at::kByte));
  check(graphop_attention_forward(dtype_code(Q), ip(row), ip(indptr), ip(eid), ip(indices), vp(Q), vp(K), vp(V), vp(o),
                                  vp(stats), row.size(0), e, n_q, n_k, h, d, ws.data_ptr(), nbytes, p.plan, stream_of(Q)));
  return {o, stats};
}

std::vector<at::Tensor> attention_backward(const at::Tensor& row, const at::Tensor& indptr_r, const at::Tensor& eid_r,
                                           const at::Tensor& indices_r, const at::Tensor& col,
                                           const at::Tensor& indptr_c, const at::Tensor& eid_c,
                                           const at::Tensor& indices_c, const at::Tensor& Q, const at::Tensor& K,
                                           const at::Tensor& V, const at::Tensor& o, const at::Tensor& stats,
                                           const at::Tensor& dO_) {
  CHECK_INPUT(row); CHECK_INPUT(indptr_r); CHECK_INPUT(eid_r); CHECK_INPUT(indices_r);
  CHECK_INPUT(col); CHECK_INPUT(indptr_c); CHECK_INPUT(eid_c); CHECK_INPUT(indices_c);
  CHECK_INPUT(Q); CHECK_INPUT(K); CHECK_INPUT(V); CHECK_INPUT(o); CHECK_INPUT(stats);
  CHECK_INDEX(row); CHECK_INDEX(indptr_r); CHECK_INDEX(eid_r); CHECK_INDEX(indices_r);
  CHECK_INDEX(col); CHECK_INDEX(indptr_c); CHECK_INDEX(eid_c); CHECK_INDEX(indices_c);
  CHECK_CUDA(dO_);
  const at::Tensor dO = dO_.contiguous();
  CHECK_SAME_DTYPE(Q, K); CHECK_SAME_DTYPE(Q, V); CHECK_SAME_DTYPE(Q, o); CHECK_SAME_DTYPE(Q, stats); CHECK_SAME_DTYPE(Q, dO);
  TORCH_CHECK(K.sizes() == V.sizes() && Q.sizes().slice(1) == K.sizes().slice(1),
              "attention_backward: Q (n_q,[h,]d), K and V (n_k,[h,]d) expected");
  DeviceGuard dg(Q);
  const int64_t e = eid_r.size(0), d = Q.size(-1), h = Q.dim() == 2 ? 1 : Q.size(1), n_q = Q.size(0), n_k = K.size(0);
  TORCH_CHECK(o.sizes() == Q.sizes() && dO.sizes() == Q.sizes() && stats.numel() == n_q * h * 2,
              "attention_backward: o, dO must match Q and stats must be (n_q, h, 2)");
  auto dQ = at::empty_like(Q), dK = at::empty_like(K), dV = at::empty_like(V);
  const auto ppr = get_plan(row, indptr_r, eid_r, indices_r, n_k);
  const auto ppc = get_plan(col, indptr_c, eid_c, indices_c, n_q);
  const auto &pr = *ppr, &pc = *ppc;
  int64_t nbytes = 0;
  check(graphop_attention_workspace_bytes(dtype_code(Q), 1, e, n_q, n_k, h, d, pr.plan, pc.plan, stream_of(Q), &nbytes));
  auto ws = at::empty({std::max<int64_t>(nbytes, 1)}, Q.options().dtype(at::kByte));
  check(graphop_attention_backward(dtype_code(Q), ip(row), ip(indptr_r), ip(eid_r), ip(indices_r), ip(col), ip(indptr_c),
                                   ip(eid_c), ip(indices_c), vp(Q), vp(K), vp(V), vp(o), vp(stats), vp(dO), vp(dQ),
                                   vp(dK), vp(dV), row.size(0), col.size(0), e, n_q, n_k, h, d, ws.data_ptr(), nbytes,
                                   pr.plan, pc.plan, stream_of(Q)));
  return {dQ, dK, dV};
}

void clear_plan_cache() {
  std::vector<PlanRef> dead;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_graphs)
      for (auto& p : kv.second.plans) dead.push_back(std::move(p));
    g_graphs.clear();
    g_lru.clear();
  }
}

// drop the plans of the orientation whose chunk list is `row` (graphs.release)
void release_plans(const at::Tensor& row) {
  std::vector<PlanRef> dead;
  const void* r = row.numel() ? row.data_ptr() : nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_graphs.begin(); it != g_graphs.end();) {
    if (it->first.row == r) {
      for (auto& p : it->second.plans) dead.push_back(std::move(p));
      g_lru.erase(it->second.lru);
      it = g_graphs.erase(it);
    } else {
      ++it;
    }
  }
}

int64_t plan_cache_size() {
  std::lock_guard<std::mutex> lk(g_mu);
  return (int64_t)g_graphs.size();
}

// ---- registration 1: the reference's pybind11 module (graphop.cpp:216-225) ------------------------------
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("maskedmm_csr_forward", &maskedmm_csr_forward, "Masked Matrix Multiplication forward(CSR Format)");
  m.def("maskedmm_csr_backward", &maskedmm_csr_backward, "Masked Matrix Multiplication backward(CSR Format)");
  m.def("node_mul_edge_forward", &node_mul_edge_forward, "Node Multiply Edge forward");
  m.def("node_mul_edge_backward", &node_mul_edge_backward, "Node Multiply Edge backward");
  m.def("sparse_softmax_forward", &sparse_softmax_forward, "Sparse softmax forward");
  m.def("sparse_softmax_backward", &sparse_softmax_backward, "Sparse softmax backward");
  m.def("vector_spmm_forward", &vector_spmm_forward, "Vectorized SPMM forward");
  m.def("vector_spmm_backward", &vector_spmm_backward, "Vectorized SPMM backward");
  m.def("attention_forward", &attention_forward, "Fused SDDMM -> softmax -> SpMM forward (extra op)");
  m.def("attention_backward", &attention_backward, "Fused attention backward (extra op)");
  m.def("clear_plan_cache", &clear_plan_cache, "Destroy every cached per-graph plan");
  m.def("release_plans", &release_plans, "Drop the cached plans of the orientation whose chunk list is `row`");
  m.def("plan_cache_size", &plan_cache_size, "Graph orientations in the plan cache");
}

// ---- registration 2: torch.ops.graphop.* (schemas + CUDA(HIP) implementations) ------------------------------
TORCH_LIBRARY(graphop, m) {
  m.def("maskedmm_csr_forward(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor A, Tensor B) -> Tensor");
  m.def("maskedmm_csr_backward(Tensor row, Tensor indptr_r, Tensor eid_r, Tensor indices_r, Tensor col, Tensor indptr_c, Tensor eid_c, Tensor indices_c, Tensor A, Tensor B, Tensor dy) -> Tensor[]");
  m.def("node_mul_edge_forward(Tensor row, Tensor indptr, Tensor eid, Tensor A, Tensor B) -> Tensor");
  m.def("node_mul_edge_backward(Tensor row, Tensor indptr, Tensor eid, Tensor A, Tensor B, Tensor dy) -> Tensor[]");
  m.def("sparse_softmax_forward(Tensor row, Tensor indptr, Tensor eid, Tensor x) -> Tensor");
  m.def("sparse_softmax_backward(Tensor row, Tensor indptr, Tensor eid, Tensor y, Tensor dy) -> Tensor");
  m.def("vector_spmm_forward(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor edata, Tensor x) -> Tensor");
  m.def("vector_spmm_backward(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor col, Tensor indptr_t, Tensor eid_t, Tensor indices_t, Tensor edata, Tensor dy, Tensor x) -> Tensor[]");
  m.def("attention_forward(Tensor row, Tensor indptr, Tensor eid, Tensor indices, Tensor Q, Tensor K, Tensor V) -> Tensor[]");
  m.def("attention_backward(Tensor row, Tensor indptr_r, Tensor eid_r, Tensor indices_r, Tensor col, Tensor indptr_c, Tensor eid_c, Tensor indices_c, Tensor Q, Tensor K, Tensor V, Tensor o, Tensor stats, Tensor dO) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(graphop, CUDA, m) {
  m.impl("maskedmm_csr_forward", &maskedmm_csr_forward);
  m.impl("maskedmm_csr_backward", &maskedmm_csr_backward);
  m.impl("node_mul_edge_forward", &node_mul_edge_forward);
  m.impl("node_mul_edge_backward", &node_mul_edge_backward);
  m.impl("sparse_softmax_forward", &sparse_softmax_forward);
  m.impl("sparse_softmax_backward", &sparse_softmax_backward);
  m.impl("vector_spmm_forward", &vector_spmm_forward);
  m.impl("vector_spmm_backward", &vector_spmm_backward);
  m.impl("attention_forward", &attention_forward);
  m.impl("attention_backward", &attention_backward);
}

TORCH_LIBRARY_IMPL(graphop, CPU, m) {   // there is no CPU implementation: the reference's CHECK_CUDA message
  m.impl("maskedmm_csr_forward", &maskedmm_csr_forward);
  m.impl("maskedmm_csr_backward", &maskedmm_csr_backward);
  m.impl("node_mul_edge_forward", &node_mul_edge_forward);
  m.impl("node_mul_edge_backward", &node_mul_edge_backward);
  m.impl("sparse_softmax_forward", &sparse_softmax_forward);
  m.impl("sparse_softmax_backward", &sparse_softmax_backward);
  m.impl("vector_spmm_forward", &vector_spmm_forward);
  m.impl("vector_spmm_backward", &vector_spmm_backward);
  m.impl("attention_forward", &attention_forward);
  m.impl("attention_backward", &attention_backward);
}
