// Shared host/device helpers for libgraphop_hip (gfx950 / CDNA4 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <utility>

#include "graphop_hip.h"

namespace graphop {

using i64 = long long;  // same width as int64_t; HIP atomics are declared on (unsigned) long long

// ---- error reporting (thread-local message, C ABI returns a code) ------------------------------
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* get_error();

#define GO_CHECK_ARG(cond, ...)                \
  do {                                         \
    if (!(cond)) {                             \
      ::graphop::set_error(__VA_ARGS__);       \
      return GRAPHOP_ERR_INVALID_ARGUMENT;     \
    }                                          \
  } while (0)

#define GO_HIP(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::graphop::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                     \
      return GRAPHOP_ERR_HIP;                                                             \
    }                                                                                     \
  } while (0)

#define GO_LAUNCH_CHECK() GO_HIP(hipGetLastError())

inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

inline i64 ceil_div(i64 a, i64 b) { return (a + b - 1) / b; }

// ---- device memory of plans / setup temporaries -------------------------------------------------
// Goes through the allocator the binding registered (graphop_set_allocator: torch's caching
// allocator in the Python binding, so plan memory is visible to / reclaimable by the framework and
// frees are stream-ordered, no device-wide stall) or, without one, hipMalloc / hipFree.
hipError_t go_malloc(void** p, size_t bytes, hipStream_t st);
void go_free(void* p);
bool go_alloc_stream_ordered();   // true: a free needs no preceding stream synchronisation

// Setup work (plan / window-structure construction) allocates and synchronises: it cannot run while
// `st` is being captured into a HIP graph.  Returns GRAPHOP_ERR_INVALID_ARGUMENT with a message
// that names the remedy instead of letting the capture fail with an opaque HIP error.
int check_not_capturing(hipStream_t st, const char* what);

// Device-side failures (kernels_walk.h: walk_fail) are reported through one host-mapped word: kernels store a code,
// the host looks at it at the start of every entry point and in graphop_check_device_errors (graphop_hip.hip).
int* device_error_word(bool create);   // device-visible pointer (nullptr when there is none yet and !create)
int check_async_error(bool clear = false);   // GRAPHOP_OK, or GRAPHOP_ERR_HIP with the message set; the record is sticky unless clear
int tuning_plan_trim();                   // knob plan_trim (host.h: Tuning), for plan.hip
int walk_launch_id(const char* tag);     // sequence number of the walk launch about to be made under this pass tag

// ---- plan (host view) ---------------------------------------------------------------------------
struct PlanStats {  // device-resident while the analysis kernels run, then copied back
  i64 unsorted;          // #positions with row[c] < row[c-1]
  i64 bad_indptr;        // #positions with indptr[c] > indptr[c+1] or out of [0, E]
  i64 eid_not_identity;  // #k with eid[k] != k
  i64 bad_eid;           // #k with eid[k] outside [0, E)
  i64 bad_index;         // #k with indices[k] outside [0, bound)
  i64 max_row;
  i64 max_gap;           // most consecutive row ids without a chunk in front of a chunk's row (sorted rows: the longest run of edge-less rows)
  i64 max_index;
  i64 max_seg_len;
  i64 n_segments;
  i64 unsorted_ids;      // #adjacent slot pairs of one row with indices[k] > indices[k+1]
};

}  // namespace graphop

namespace graphop {
// Window-sweep structure of one plan for a given window geometry (see kernels_fast.h).
struct Sweep {
  int W = 0;            // number of column windows
  i64 win_cols = 0;     // ids per window
  int T = 0;            // longest vrow (rows longer than T slots are cut into pieces)
  int V = 0;            // number of vrows
  int* vr_row = nullptr;  // [V] owning row id
  int* wp_lo = nullptr;   // [W*V] first slot of vrow v inside window w
  int* wp_hi = nullptr;   // [W*V] one past its last slot inside window w
  int runtime_wp = 0;     // sticky: a per-batch (non-staged) kernel reads wp_lo / wp_hi at run time -- plan_trim leaves them alone
  int* queues = nullptr;  // [kQueueRing][8 * 64] task-queue heads of the window-owner drivers
  unsigned queue_next = 0;  // next ring slot (taken under the plan's sweep mutex)
  // Window-major ("dealt") layouts of the window-owner tasks, one per lane-group geometry, built on
  // first use: the granules of every (window, vrow tile) task are dealt to the wave's lane groups
  // once, here, and the neighbour ids of each group's strip are stored as ONE contiguous,
  // 16-byte-aligned run, so a strip fetches its ids with a few wide loads (kernels_fast.h, staged strips).
  struct Dealt {
    int L = 0, K = 0;       // lanes per group, vrows per group: tile = (64 / L) * K vrows per task
    int tiles = 0;          // tasks per window
    int* rec = nullptr;     // int4 [W * tiles * tile]: (first slot, length, row id, position in ids) per granule, dealt order
    int* ids = nullptr;     // neighbour ids, window-major dealt order (strips padded to 4 ints)
    int* eids = nullptr;    // same order: edge ids (nullptr when eid is the identity)
    long long n_ids = 0;
  };
  static constexpr int kMaxDealt = 4;
  Dealt dealt[kMaxDealt];
  int n_dealt = 0;
};
// Walk layout of one plan (kernels_walk.h): the slots of the CSR, read as one tape, are cut into
// waves * rounds BINS of equal length; bin (round r, wave q) is what wave q of the resident grid
// works on in round r.  A bin holds at most K * GW rows (whole rows, plus the pieces of the rows
// its two ends cut; a piece takes its share of the row's slots inside EVERY column window).  Inside
// every window the bin's slots are dealt to the wave's GW lane groups in equal contiguous shares; a
// lane group's shares of windows 0, 1, ... are stored as ONE contiguous run: per slot the neighbour
// id with the bin-local row number in the top bits, and the edge id (index of the per-edge scalar /
// result).  A lane group streams its run through LDS; the wave keeps the bin's rows (partial sums of
// an SpMM-type pass, A rows of an SDDMM-type pass) in LDS for the whole round: nothing is flushed
// per window.
struct Walk {
  int W = 0;
  i64 win_cols = 0;
  int groups = 0;         // lane groups of the resident grid the tape was cut for
  int GW = 0;             // lane groups per wave
  int K = 0;              // rows per lane group (<= kWalkK; fewer when the kernel's LDS also holds per-head weight rings)
  int rounds = 0;
  int xcd_slots = 0;      // 8 (groups % 8 == 0: every XCD slot owns a contiguous share of each round's tape) or 1
  long long n_slots = 0;  // ints in ids / widx (bins padded to 4, slack for whole-segment fetches)
  int* ids = nullptr;     // [(k << kWalkKShift) | neighbour id]
  int* widx = nullptr;    // edge id per slot
  int* bin_pos = nullptr; // [bins * GW + 1] first slot of every lane group's run
  int* bin_rows = nullptr;// [bins * K * GW] row id | (shared << 31) | (first piece of a shared row << 30), -1 = unused
  int* bin_cum = nullptr; // [bins * GW] slots in every lane group's run
  int* sync = nullptr;    // kWalkSyncRing sets of pacer counters of the walk kernels (a set is zeroed before the launch that takes it)
  long long sync_ints = 0;  // ints per set
  unsigned sync_next = 0;   // next set (taken under the plan's sweep mutex)
  int max_steps = 0;      // pacing steps per round the counters are sized for
  long long longest_run = 0;  // slots in the longest lane-group run (pacing steps are cut from it)
};
constexpr int kWalkSyncRing = 4;
constexpr int kWalkK = 15;        // most rows per lane group: 15 x 256 B + a 1 KB ring, x 32 lane groups = 152 KB of a CU's 160 KB LDS
constexpr int kWalkFirstPiece = 1 << 30;   // Walk::bin_rows: bit 31 = row shared with a neighbouring bin, bit 30 = its first piece
constexpr int kWalkRowMask = (1 << 30) - 1;
constexpr int kWalkKShift = 26;   // ids of a table < 4 GiB of >= 64-B rows need 26 bits; 6 bits of row-in-bin (<= 56)
constexpr long long kLongSegment = 1024;   // rows above this many slots are listed for the workgroup-per-row softmax
constexpr long long kLongSegmentBwd = 2048;  // ... which the backward uses only above this many (it caches 32 items per lane)
// Every window-owner launch takes the next of kQueueRing sets of queue heads, so launches that
// overlap on different streams never share one (a set is reused 64 launches later).
constexpr int kQueueRing = 64;
constexpr int kQueueInts = 8 * 64;        // 8 heads, one 256-B line each
}  // namespace graphop

struct graphop_plan {
  graphop_plan_info_t info;
  int sorted_in_rows;      // neighbour ids ascend inside every row segment
  void* sweeps;            // std::vector<graphop::Sweep>* (lazily built, guarded by sweep_mu)
  void* sweep_mu;          // std::mutex*
  void* walks;             // std::vector<graphop::Walk>* (lazily built, guarded by sweep_mu)
  const int64_t* row;      // identity of the arrays the plan was built from (not owned)
  const int64_t* indptr;
  const int64_t* eid;
  const int64_t* indices;
  int64_t* seg_chunk;      // [n_segments + 1] first chunk of each segment, then n_chunks (owned)
  int64_t* seg_eptr;       // [n_segments + 1] first slot of each segment, then the end of the last (owned, optional; row_owned plans)
  int32_t* idx32;          // [n_edges] (owned, optional)
  int32_t* eid32;          // [n_edges] (owned, optional; NULL when eid is the identity)
  int32_t* long_segs;      // [n_long] segments longer than kLongSegment slots (owned)
  int64_t n_long;
  int32_t* blk_seg;        // [n_dense_blocks + 1] first segment of every dense block (owned, optional)
  int32_t* seg_e0;         // [n_segments + 1] first slot of every segment (owned, with blk_seg)
  int32_t* seg_row;        // [n_segments] row id of every segment (owned, with blk_seg)
  int device;
  int mirrors_pinned;      // sticky: a kernel reads idx32 / eid32 at run time (per-batch window-owner kernels, block-dense kernels)
};

// ---- device-side helpers ------------------------------------------------------------------------
#ifdef __HIPCC__
namespace graphop {

constexpr int kWave = 64;

// DPP lane movement inside a 16-lane row (no LDS traffic).  ctrl: 0xB1 = quad_perm[1,0,3,2],
// 0x4E = quad_perm[2,3,0,1], 0x141 = row_half_mirror, 0x140 = row_mirror.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {   // two 32-bit DPP moves
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL> __device__ __forceinline__ float dpp_t(float v) { return dpp_f32<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double dpp_t(double v) { return dpp_f64<CTRL>(v); }

template <int G>
__device__ __forceinline__ double group_sum(double v) {
  if constexpr (G >= 2) v += dpp_f64<0xB1>(v);
  if constexpr (G >= 4) v += dpp_f64<0x4E>(v);
  if constexpr (G >= 8) v += dpp_f64<0x141>(v);
  if constexpr (G >= 16) v += dpp_f64<0x140>(v);
  if constexpr (G >= 32) v += __shfl_xor(v, 16);
  if constexpr (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

// Sum over aligned groups of G lanes (G = 1..64, power of two); every lane gets the total.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G >= 2) v += dpp_f32<0xB1>(v);
  if constexpr (G >= 4) v += dpp_f32<0x4E>(v);
  if constexpr (G >= 8) v += dpp_f32<0x141>(v);
  if constexpr (G >= 16) v += dpp_f32<0x140>(v);
  if constexpr (G >= 32) v += __shfl_xor(v, 16);
  if constexpr (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ float group_sum_rt(float v, int g) {  // g wave-uniform
  switch (g) {
    case 2: return group_sum<2>(v);
    case 4: return group_sum<4>(v);
    case 8: return group_sum<8>(v);
    case 16: return group_sum<16>(v);
    case 32: return group_sum<32>(v);
    case 64: return group_sum<64>(v);
    default: return v;
  }
}

// ---- compile-time loops and lane-group broadcasts ------------------------------------------------
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {   // f(integral_constant<int, 0>) ... f(<N-1>)
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Value held by lane U of every aligned group of L lanes, without an LDS round trip where the
// hardware allows: L = 16 is one DPP row (row_newbcast, and the compiler folds it into the consuming
// add: v_add_u32_dpp), L = 4 a quad_perm, L = 64 a readlane (scalar), L = 8 / 32 two of those and a
// select.  The strips issue one of these per slot; as ds_bpermute each cost an LDS-pipeline round
// trip (~100 cycles) in front of the row request it feeds.
template <int L, int U>
__device__ __forceinline__ int group_bcast(int v) {
  static_assert(U >= 0 && U < L, "lane out of range");
  if constexpr (L == 16) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, true);
  } else if constexpr (L == 4) {
    return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xF, 0xF, true);
  } else if constexpr (L == 8) {
    const int lo = __builtin_amdgcn_update_dpp(0, v, 0x150 + U, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, v, 0x150 + U + 8, 0xF, 0xF, true);
    return (threadIdx.x & 8) ? hi : lo;
  } else if constexpr (L == 64) {
    return __builtin_amdgcn_readlane(v, U);
  } else if constexpr (L == 32) {
    const int lo = __builtin_amdgcn_readlane(v, U);
    const int hi = __builtin_amdgcn_readlane(v, U + 32);
    return (threadIdx.x & 32) ? hi : lo;
  } else {
    return __shfl(v, U, L);
  }
}
template <int L, int U>
__device__ __forceinline__ unsigned group_bcast(unsigned v) { return (unsigned)group_bcast<L, U>((int)v); }
template <int L, int U>
__device__ __forceinline__ float group_bcast(float v) {
  return __builtin_bit_cast(float, group_bcast<L, U>(__builtin_bit_cast(int, v)));
}
template <int L, int U>
__device__ __forceinline__ double group_bcast(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = group_bcast<L, U>((int)b), hi = group_bcast<L, U>((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// p[u] = this lane's partial of slot u's dot product.  Returns, in lane u of every group, the sum of
// p[u] over the group's L lanes (other lanes: unspecified).  For a 16-lane group and 16 slots this is
// a transpose-reduce: every step halves the number of live values (the lane keeps the slots whose
// index bit matches its own lane bit and adds its partner's partials of those slots): 45 VALU
// instructions instead of 16 x (4 DPP adds + a select) = 80.
template <int L, int SB, typename T>
__device__ __forceinline__ T group_dots_to_owner(T (&p)[SB], int l) {
  if constexpr ((L == 16 || L == 32) && SB == 16) {
    // L == 32 (512-B rows): the two 16-lane halves of the group first exchange their partials (lane ^ 16), then each
    // half runs the 16-lane tree: lanes u and u + 16 both end up with slot u's sum -- 16 cross-row exchanges + 45
    // instructions instead of 16 x (4 DPP adds + a cross-row exchange + a select) = 96
    if constexpr (L == 32) {
#pragma unroll
      for (int u = 0; u < 16; ++u) p[u] += __shfl_xor(p[u], 16);
    }
    T t8[8], t4[4], t2[2];
    const bool b3 = l & 8, b2 = l & 4, b1 = l & 2, b0 = l & 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const T keep = b3 ? p[u + 8] : p[u], send = b3 ? p[u] : p[u + 8];
      t8[u] = keep + dpp_t<0x128>(send);        // row_ror:8 = lane ^ 8
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const T keep = b2 ? t8[u + 4] : t8[u], send = b2 ? t8[u] : t8[u + 4];
      t4[u] = keep + dpp_t<0x141>(send);        // row_half_mirror: flips bits 0-2, keeps bit 3
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const T keep = b1 ? t4[u + 2] : t4[u], send = b1 ? t4[u] : t4[u + 2];
      t2[u] = keep + dpp_t<0x4E>(send);         // quad_perm [2,3,0,1] = lane ^ 2
    }
    const T keep = b0 ? t2[1] : t2[0], send = b0 ? t2[0] : t2[1];
    return keep + dpp_t<0xB1>(send);            // quad_perm [1,0,3,2] = lane ^ 1
  } else if constexpr (L == 16 && SB == 8 && sizeof(T) == 4) {   // lanes u and u + 8 both end up with slot u's sum
    // Same summation tree as the 16-slot form (pairs {l, l^8} first, then ^7, ^2, ^1), so a dot
    // product recomputed by an 8-slot strip is BITWISE the value a 16-slot strip stored: the fused
    // attention backward re-derives exp(s - m) from the forward's row maxima.
    float t8[8], t4[4], t2[2];
    const bool b2 = l & 4, b1 = l & 2, b0 = l & 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) t8[u] = p[u] + dpp_f32<0x128>(p[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float keep = b2 ? t8[u + 4] : t8[u], send = b2 ? t8[u] : t8[u + 4];
      t4[u] = keep + dpp_f32<0x141>(send);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float keep = b1 ? t4[u + 2] : t4[u], send = b1 ? t4[u] : t4[u + 2];
      t2[u] = keep + dpp_f32<0x4E>(send);
    }
    const float keep = b0 ? t2[1] : t2[0], send = b0 ? t2[0] : t2[1];
    return keep + dpp_f32<0xB1>(send);
  } else {
    T res = 0;
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const T s = group_sum<L>(p[u]);
      if (l == u) res = s;
    }
    return res;
  }
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// Float max through integer atomics (no CAS loop): non-negative floats order like ints,
// negative floats order inversely as unsigned.
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  v += 0.f;   // -0.0f -> +0.0f: as an int, -0.0f is INT_MIN and would lose against every stored value
  if (v >= 0.f)
    atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else
    atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_float(double* addr, double v) {
  v += 0.0;
  if (v >= 0.0)
    atomicMax(reinterpret_cast<long long*>(addr), __double_as_longlong(v));
  else
    atomicMin(reinterpret_cast<unsigned long long*>(addr),
              static_cast<unsigned long long>(__double_as_longlong(v)));
}

__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }
// exp(a) for a <= 0 (softmax arguments: value - row maximum), full fp32 accuracy without expf's range handling:
// 2^t * (1 + ln2 * e) with t = fl(a * log2e) and e = the rounding error of that product + a * (log2e's low
// part).  Arguments below -200 (and the -inf of a padding lane) are clamped there: the result is 0 either way.
// Six full-rate instructions + one v_exp_f32 instead of expf's fourteen: the multi-head softmax is VALU-bound.
__device__ __forceinline__ float exp_nonpos(float a) {
  a = fmaxf(a, -200.f);
  const float log2e = 1.44269502162933349609375f, log2e_lo = 1.925963033500011e-08f;
  const float t = a * log2e;
  const float e = fmaf(a, log2e_lo, fmaf(a, log2e, -t));
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * 0.693147182464599609375f, r);
}
// exp of a non-positive argument by type
__device__ __forceinline__ float exp_le0(float x) { return exp_nonpos(x); }
__device__ __forceinline__ double exp_le0(double x) { return exp(x); }

}  // namespace graphop
#endif  // __HIPCC__
