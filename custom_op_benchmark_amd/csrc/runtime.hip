// libgraphop_hip runtime: error text, allocator hook, per-kernel timing, tuning knobs, zero fill, the device error
// record and the C ABI entry points that are not operators (include/graphop_hip.h).  Split from graphop_hip.hip in round 5.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "host.h"
#include "kernels_walk.h"

namespace graphop {

// Zero fill as a kernel.  hipMemsetAsync is not used on the data path: as a memset node of a
// captured HIP graph (ROCm 7.2) it left half of the dwords of the byte range untouched on replay.
__global__ void k_zero16(uint4* __restrict__ p, i64 n16, unsigned char* __restrict__ tail, int n_tail) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  if (i < n_tail) tail[i] = 0;
  for (; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
__global__ void k_zero1(unsigned char* __restrict__ p, i64 n) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 stride = (i64)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0;
}

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

// ---- allocator hooks ---------------------------------------------------------------------------------
static graphop_alloc_fn g_alloc = nullptr;
static graphop_free_fn g_free = nullptr;
static std::mutex g_alloc_mu;
static std::map<void*, size_t> g_hooked;   // pointers that came from g_alloc (freed through g_free only) -> bytes
static std::map<void*, size_t> g_plain;    // pointers from hipMalloc -> bytes
static size_t g_bytes = 0;                 // device bytes currently held through go_malloc (plans, their layouts, temporaries)

hipError_t go_malloc(void** p, size_t bytes, hipStream_t st) {
  *p = nullptr;
  if (bytes == 0) bytes = 16;
  graphop_alloc_fn a;
  { std::lock_guard<std::mutex> lk(g_alloc_mu); a = g_alloc; }
  if (a) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    void* q = a(bytes, dev, (void*)st);
    if (!q) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_hooked[q] = bytes;
    g_bytes += bytes;
    *p = q;
    return hipSuccess;
  }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    g_plain[*p] = bytes;
    g_bytes += bytes;
  }
  return e;
}
void go_free(void* p) {
  if (!p) return;
  graphop_free_fn f = nullptr;
  bool hooked = false;
  {
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_hooked.find(p);
    if (it != g_hooked.end()) {
      hooked = true;
      g_bytes -= it->second;
      g_hooked.erase(it);
    } else {
      auto jt = g_plain.find(p);
      if (jt != g_plain.end()) { g_bytes -= jt->second; g_plain.erase(jt); }
    }
    f = g_free;
  }
  if (hooked) { if (f) f(p); return; }   // (allocator gone at shutdown: left to process exit)
  (void)hipFree(p);
}
bool go_alloc_stream_ordered() {
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  return g_alloc != nullptr;
}

int check_not_capturing(hipStream_t st, const char* what) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
    set_error("%s would allocate and synchronise while the stream is being captured into a HIP graph: "
              "build it first (graphop_plan_create / graphop_plan_prepare, Python: graphop.prepare(graph, h, d)) "
              "and capture afterwards", what);
    return GRAPHOP_ERR_INVALID_ARGUMENT;
  }
  (void)hipGetLastError();
  return GRAPHOP_OK;
}

// ---- optional per-kernel timing (hipEvents on the launch stream; off by default) -----------------
struct ProfRec { const char* name; const char* kernel; hipEvent_t t0, t1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

ProfScope::ProfScope(const char* n, hipStream_t s, const char* k) : st(s), name(n), kernel(k) {
  if (!g_prof_on) return;
  if (hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) { t0 = nullptr; return; }
  (void)hipEventRecord(t0, st);
}
ProfScope::~ProfScope() {
  if (!t0) return;
  (void)hipEventRecord(t1, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back({name, kernel, t0, t1});
}



Tuning& tuning_mut() {
  static Tuning t;
  return t;
}
const Tuning& tuning() { return tuning_mut(); }
int tuning_plan_trim() { return tuning().plan_trim; }

// Stream-ordered zero fill (a kernel: runtime.hip, k_zero16).
hipError_t zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  ProfScope prof("zero_fill", st, "k_zero16");
  unsigned char* p = (unsigned char*)ptr;
  const size_t head = (16 - ((uintptr_t)p & 15)) & 15;
  if (head >= bytes || bytes < 64) {
    hipLaunchKernelGGL(k_zero1, dim3((unsigned)ceil_div((i64)bytes, 256)), dim3(256), 0, st, p, (i64)bytes);
    return hipGetLastError();
  }
  if (head) hipLaunchKernelGGL(k_zero1, dim3(1), dim3(64), 0, st, p, (i64)head);
  p += head;
  const size_t body = bytes - head;
  const i64 n16 = (i64)(body / 16);
  const int n_tail = (int)(body % 16);
  i64 blocks = ceil_div(n16, 256 * 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_zero16, dim3((unsigned)blocks), dim3(256), 0, st, (uint4*)p, n16, p + n16 * 16, n_tail);
  return hipGetLastError();
}

// ---- device-side error record ---------------------------------------------------------------------------
// Host-mapped words written by kernels: [0] = code, [1] = sequence number of the walk launch that failed.  STICKY
// (ABI 7): entry points fail while [0] is set; only graphop_check_device_errors (clear = true) resets it.
static int* g_err_host = nullptr;   // hipHostMalloc'ed (mapped, coherent): written by kernels, read here
static int* g_err_dev = nullptr;
static std::mutex g_err_mu;
struct WalkLaunchRec { const char* tag; int device; unsigned seq; };
static WalkLaunchRec g_walk_ring[256];   // the last 256 walk launches: sequence number -> pass tag, device
static unsigned g_walk_seq = 0;
int* device_error_word(bool create) {
  std::lock_guard<std::mutex> lk(g_err_mu);
  if (!g_err_host && create) {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
      memset(h, 0, 64);
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, h, 0) == hipSuccess) { g_err_host = (int*)h; g_err_dev = (int*)d; }
      else (void)hipHostFree(h);
    }
    (void)hipGetLastError();
  }
  return g_err_dev;
}
// sequence number of the walk launch about to be made under pass tag `tag` (kept so that a failure can be named)
int walk_launch_id(const char* tag) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_err_mu);
  const unsigned seq = ++g_walk_seq;
  g_walk_ring[seq & 255] = WalkLaunchRec{tag, dev, seq};
  return (int)seq;
}
int check_async_error(bool clear) {
  int code = 0, seq = 0;
  WalkLaunchRec rec{nullptr, -1, 0};
  {
    std::lock_guard<std::mutex> lk(g_err_mu);
    if (!g_err_host) return GRAPHOP_OK;
    code = __atomic_load_n(g_err_host, __ATOMIC_ACQUIRE);
    if (code == 0) return GRAPHOP_OK;
    seq = __atomic_load_n(g_err_host + 1, __ATOMIC_ACQUIRE);
    if (g_walk_ring[(unsigned)seq & 255].seq == (unsigned)seq) rec = g_walk_ring[(unsigned)seq & 255];
    if (clear) {
      __atomic_store_n(g_err_host + 1, 0, __ATOMIC_RELEASE);
      __atomic_store_n(g_err_host, 0, __ATOMIC_RELEASE);
    }
  }
  const char* what = code == kWalkErrQuad ? "a (step, quad) unit waited for its quad's previous step"
                   : code == kWalkErrRing ? "a worker wave waited for a ring chunk of its feeder wave"
                   : "unknown code";
  char where[96];
  if (rec.tag) snprintf(where, sizeof(where), "pass '%s' on device %d, walk launch #%d", rec.tag, rec.device, seq);
  else snprintf(where, sizeof(where), "walk launch #%d", seq);
  set_error("a walk kernel (k_spmm_walk_* / k_attn_fwd_walk_f32) of an earlier launch aborted (%s): %s until its spin bound "
            "expired (device error code %d); the outputs of that launch and of every launch that consumed them are invalid%s",
            where, what, code,
            clear ? "" : " -- sticky: graphop_check_device_errors() acknowledges and clears it");
  return GRAPHOP_ERR_HIP;
}

}  // namespace graphop

using namespace graphop;

namespace {
struct TuneEntry { const char* k; int* p; };
// every knob of Tuning (host.h), by name
std::vector<TuneEntry> tune_table() {
  Tuning& t = tuning_mut();
  return {
      {"sddmm_cpg", &t.sddmm_cpg}, {"spmm_cpg", &t.spmm_cpg}, {"force_generic", &t.force_generic},
      {"sweep", &t.sweep}, {"window_kb", &t.window_kb}, {"mall_window_kb", &t.mall_window_kb}, {"max_windows", &t.max_windows},
      {"sweep_min_kb", &t.sweep_min_kb}, {"sweep_bpc", &t.sweep_bpc}, {"sweep_k", &t.sweep_k},
      {"vrow_t", &t.vrow_t}, {"sweep_min_granule", &t.sweep_min_granule}, {"sweep_w", &t.sweep_w}, {"spmm_window_scale", &t.spmm_window_scale}, {"dense_blocks", &t.dense_blocks}, {"dense_min_fill", &t.dense_min_fill},
      {"dense_detect_min_fill", &t.dense_detect_min_fill},
      {"attn_fused", &t.attn_fused},
      {"attn_window_scale", &t.attn_window_scale}, {"attn_k", &t.attn_k}, {"attn_bpc", &t.attn_bpc},
      {"attn_rows", &t.attn_rows}, {"attn_fwd_walk", &t.attn_fwd_walk}, {"spmm_selfzero", &t.spmm_selfzero},
      {"spmm_selfzero_min_mb", &t.spmm_selfzero_min_mb}, {"spmm_flat", &t.spmm_flat},
      {"spmm_flat_cpg", &t.spmm_flat_cpg}, {"spmm_flat_max_mean", &t.spmm_flat_max_mean},
      {"spmm_flat_min_chunks", &t.spmm_flat_min_chunks}, {"staged_ids", &t.staged_ids}, {"attn_max_d", &t.attn_max_d},
      {"touch_sddmm", &t.touch_sddmm}, {"walk", &t.walk}, {"walk_window_kb", &t.walk_window_kb}, {"walk_window_kb_col", &t.walk_window_kb_col},
      {"walk_drift", &t.walk_drift}, {"walk_min_bin", &t.walk_min_bin}, {"walk_blocks", &t.walk_blocks}, {"walk_debug", &t.walk_debug}, {"walk_fault", &t.walk_fault}, {"walk_steps", &t.walk_steps},
      {"plan_trim", &t.plan_trim}};
}
}  // namespace


extern "C" {


int graphop_abi_version(void) { return GRAPHOP_ABI_VERSION; }
const char* graphop_last_error(void) { return get_error(); }

int graphop_tune(const char* key, int value) {
  GO_CHECK_ARG(key != nullptr, "tune: key is NULL");
  for (auto& e : tune_table())
    if (strcmp(e.k, key) == 0) {
      *e.p = value;
      return GRAPHOP_OK;
    }
  set_error("tune: unknown key '%s'", key);
  return GRAPHOP_ERR_INVALID_ARGUMENT;
}

int graphop_tune_get(const char* key, int* value) {
  GO_CHECK_ARG(key != nullptr && value != nullptr, "tune_get: NULL pointer");
  for (auto& e : tune_table())
    if (strcmp(e.k, key) == 0) {
      *value = *e.p;
      return GRAPHOP_OK;
    }
  set_error("tune_get: unknown key '%s'", key);
  return GRAPHOP_ERR_INVALID_ARGUMENT;
}

const char* graphop_tune_key(int i) {
  static const std::vector<TuneEntry> tab = tune_table();   // (names only: the pointers are not used)
  return (i >= 0 && i < (int)tab.size()) ? tab[(size_t)i].k : nullptr;
}

int64_t graphop_memory_bytes(void) {
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  return (int64_t)g_bytes;
}

int graphop_check_device_errors(void) { return check_async_error(true); }

int graphop_tune_reset(void) {
  tuning_mut() = Tuning();   // the defaults (environment overrides included), as at library load
  return GRAPHOP_OK;
}

int graphop_set_allocator(graphop_alloc_fn alloc_fn, graphop_free_fn free_fn) {
  GO_CHECK_ARG((alloc_fn == nullptr) == (free_fn == nullptr), "set_allocator: give both callbacks or neither");
  std::lock_guard<std::mutex> lk(g_alloc_mu);
  g_alloc = alloc_fn;
  g_free = free_fn;
  return GRAPHOP_OK;
}

int graphop_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return GRAPHOP_OK;
}

// Synchronises the recorded events, aggregates per tag, clears the log.  Writes up to `cap`
// records; returns the number of distinct tags (or -1 on error).
int graphop_profile_read(graphop_profile_rec_t* out, int cap) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  std::map<std::string, graphop_profile_rec_t> agg;
  std::vector<std::string> order;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.t1) != hipSuccess || hipEventElapsedTime(&ms, r.t0, r.t1) != hipSuccess) {
      set_error("profile_read: event query failed");
      return -1;
    }
    (void)hipEventDestroy(r.t0);
    (void)hipEventDestroy(r.t1);
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      graphop_profile_rec_t rec;
      memset(&rec, 0, sizeof(rec));
      strncpy(rec.name, r.name, sizeof(rec.name) - 1);
      strncpy(rec.kernel, r.kernel ? r.kernel : "", sizeof(rec.kernel) - 1);
      rec.min_ms = ms;
      it = agg.emplace(r.name, rec).first;
      order.push_back(r.name);
    }
    it->second.calls += 1;
    it->second.total_ms += ms;
    if (ms < it->second.min_ms) it->second.min_ms = ms;
    if (ms > it->second.max_ms) it->second.max_ms = ms;
  }
  g_prof.clear();
  int n = 0;
  for (auto& k : order) {
    if (out && n < cap) out[n] = agg[k];
    ++n;
  }
  return n;
}

int graphop_partition_csr_count(const int64_t* indptr, int64_t n_rows, int64_t chunk_size,
                                int64_t* first_chunk, void* stream) {
  GO_CHECK_ARG(indptr && first_chunk, "partition_csr_count: NULL pointer");
  GO_CHECK_ARG(n_rows >= 0 && n_rows < 0x7ffffffeLL, "partition_csr_count: n_rows out of range");
  GO_CHECK_ARG(chunk_size >= 1, "partition_csr_count: chunk_size must be >= 1");
  return partition_count((const i64*)indptr, n_rows, chunk_size, (i64*)first_chunk,
                         (hipStream_t)stream);
}

int graphop_partition_csr_fill(const int64_t* indptr, const int64_t* first_chunk, int64_t n_rows,
                               int64_t chunk_size, int64_t n_chunks, int64_t* row,
                               int64_t* indptr_out, void* stream) {
  GO_CHECK_ARG(indptr && first_chunk && indptr_out && (row || n_chunks == 0),
               "partition_csr_fill: NULL pointer");
  GO_CHECK_ARG(n_rows >= 0 && n_chunks >= 0 && chunk_size >= 1, "partition_csr_fill: bad size");
  return partition_fill((const i64*)indptr, (const i64*)first_chunk, n_rows, chunk_size, n_chunks,
                        (i64*)row, (i64*)indptr_out, (hipStream_t)stream);
}

}  // extern "C"
