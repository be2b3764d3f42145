// WALK drivers (plan; common.h: Walk): the flush-free loop order of the column-window passes.
//
// The window-owner drivers (kernels_fast.h) let every XCD own column windows; a row's partial sum
// then leaves the chip once per (row, window) -- memory-side float atomics, 1-1.4 GB per pass on
// the Reddit shape -- and the A rows of an SDDMM-type pass are re-read per (row, window).  Here the
// ownership is turned around: a lane group's LDS holds a BIN of rows for a whole round (plan.hip cuts
// the CSR, read as one tape of slots, into bins of equal length) -- partial sums of an SpMM-type pass,
// A rows of an SDDMM-type pass -- while ALL column windows are walked, window 0 first.  All waves of
// the chip therefore gather from the same part of the table at about the same time, which keeps
// that part in every XCD's L2; a per-XCD soft pacer (WalkPacer) bounds how far a CU may run ahead.
// Nothing is flushed per window: a row is written once per round (plain stores; atomics only for
// the <= 2 rows a bin shares with its neighbours).
//
// A bin's slots are stored window-major as one contiguous run of (row-in-bin | neighbour id, edge
// id) pairs per lane group: a strip is ONE flat list of full 16-slot batches for the whole round --
// no per-window round-up, no dealing.  Where a bin stands in the table after a given share of its
// slots varies from bin to bin like 1 / sqrt(slots per bin) (which windows a row's neighbours fall
// into is random); that spread is what the L2 has to hold.
// The layout is cut with bins per LANE GROUP: the rows are the lane group's own, partial sums are folded into LDS with
// a plain, deferred read-modify-write.  (Round 3 also had an SDDMM-type walk kernel with bins per wave, A rows in LDS
// for the round: 1.63-1.73 ms per pass against 1.51 on the window-owner strips, which have no flush to save; removed
// in round 4, numbers in profiles/r3_walk_experiments.txt.)
#pragma once
#include "kernels_fast.h"

namespace graphop {

struct WalkView {
  const int* ids;        // [(k << kWalkKShift) | neighbour id] per slot
  const int* widx;       // edge id per slot
  const int* bin_pos;    // [bins * GW + 1] first slot of every lane group's run
  const int* bin_rows;   // [bins * kWalkK * GW]
  const int* bin_cum;    // [bins * GW] slots in every lane group's run
  int* sync;             // pacer counters, zero at launch (nullptr = free-running)
  int W, groups, rounds;
  int K;                 // rows per lane group of the layout (Walk::K; LDS rows of a bin)
  int steps;             // pacing steps per round: a wave's step = position in its longest run / (that run's length / steps)
  int drift;             // a wave may start step s only once every wave of its XCD has left step s - drift
  int xcd_slots;         // grid % xcd_slots == 0; workgroup b serves XCD slot b % xcd_slots
  int stream_weights;    // 1: the per-slot weights are read in storage order (identity eid): nontemporal loads
  long long* dbg;        // diagnostics (knob walk_debug): per wave {cycles in the kernel, cycles waiting in the pacer, waits, XCC id}
  int* err;              // host-visible error record (host.h: device_error_word): a hand-over that timed out stores its code here
  int launch_id;         // ... and this launch's sequence number next to it (graphop_hip.hip: walk_launch_id), so the host can name the pass
  int fault;             // fault injection (knob walk_fault, tests only): 1 = the feeders never deliver chunk 1 of lane group 0,
                         // 2 = quad 0 never reports step 0 finished
};

// Hand-overs inside a walk workgroup (worker <-> feeder rings, quad steps) are CORRECTNESS-critical: a worker that
// read a ring chunk the feeder has not written, or started a unit before its quad's previous step was folded into
// LDS, would produce wrong numbers.  Their spins are bounded only so that a launch cannot hang the device; when a
// bound expires the wave stores a code in the host-visible error word, raises the workgroup's abort flag (every
// spin of the kernel polls it) and leaves.  The outputs of such a launch are garbage and the host reports
// GRAPHOP_ERR_HIP at the next library call / graphop_check_device_errors (graphop_hip.hip: check_async_error).
// The XCD pacer is different: results never depend on it, it gives up silently.
constexpr int kWalkErrQuad = 1;     // a (step, quad) unit waited for the quad's previous step
constexpr int kWalkErrRing = 2;     // a worker waited for a ring chunk of its feeder
constexpr int kWalkSpinQuad = 1 << 24;    // x s_sleep(1): ~1 s
constexpr int kWalkSpinRing = 1 << 22;    // x s_sleep(1): ~0.25 s
// the workgroup's abort flag as a WAVE-UNIFORM value: an exit that depends on a plain LDS read is divergent control flow
// to the compiler -- the feeder's bookkeeping (scalar registers, scalar branches) turned into exec-mask code and
// lane spills with it: 1.69 -> 2.33 ms per pass
__device__ __forceinline__ int walk_aborted(const int* wg_abort) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(wg_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// walk_fail raises the workgroup's abort flag (LDS: the code); the host-visible word is written ONCE, by the worker
// waves on their way out (walk_report).  No global store may sit inside the feeder's or the workers' loops: on gfx950
// loads and stores share the vmcnt counter, and with a store in the loop body -- however cold its path -- the
// compiler drains the loads the feeder keeps in flight across its turns (measured: +0.65 ms per pass).
__device__ __forceinline__ void walk_fail(int* wg_abort, int code) {
  __hip_atomic_store(wg_abort, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void walk_report(int* err, int launch_id, const int* wg_abort) {
  const int code = walk_aborted(wg_abort);
  if (code != 0 && err != nullptr && (threadIdx.x & (kWave - 1)) == 0) {
    __hip_atomic_store(err + 1, launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(err, code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

constexpr int kWalkIdMask = (1 << kWalkKShift) - 1;

// Soft pacing between the waves of one walk launch, per XCD (counters sharded by the hardware XCC id,
// layout as SweepPacer: sync[xcc * 64] = workgroups registered, then per (XCD, step) an arrival
// counter and a release word, one 256-B line each).  A wave SIGNALS the steps all of its lane groups
// have left and, before it gathers from step s, WAITS until every registered workgroup of its XCD
// has left step s - drift -- unless one of its own groups is still that far back (it cannot wait for
// itself; the slowest wave of an XCD never waits, so there is always progress).  Speed only: spins
// are bounded and a wave that times out stops pacing for good.
struct WalkPacer {
  int* ctr;
  int* reg;
  int* lds;        // [0..7] steps finished by each worker wave; [8] steps forwarded to the XCD counters; [9] highest released step + 1
  int drift;
  int done_next;   // first step this wave has not signalled yet (wave-uniform)
  bool active;
  long long t_wait = 0, n_wait = 0, t_feed = 0;   // diagnostics
  int n_waves;     // worker waves of the workgroup (<= 8)
  __device__ __forceinline__ WalkPacer(const WalkView& s, int* lds_words, int worker_waves = kFastBlock / kWave)
      : ctr(nullptr), reg(nullptr), lds(lds_words), drift(s.drift), done_next(0),
        active(s.sync != nullptr && s.drift > 0), n_waves(worker_waves) {
    if (!active) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= kSyncXcds - 1;
    const i64 steps = (i64)s.rounds * s.steps;
    reg = s.sync + (i64)xcc * kSyncStride;
    ctr = s.sync + (i64)kSyncStride * (kSyncXcds + 2 * (i64)xcc * steps);
    if (threadIdx.x < 12) lds[threadIdx.x] = 0;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(reg, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  // progress slot `slot` (a quad of the ticketed walk kernel; n_waves = number of slots) has finished all steps < upto
  __device__ __forceinline__ void signal_slot(int slot, int upto) {
    if (active && (threadIdx.x & (kWave - 1)) == 0) {
      __hip_atomic_store(lds + slot, upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int m = upto;
      for (int w = 0; w < n_waves; ++w) {
        const int o = __hip_atomic_load(lds + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        m = o < m ? o : m;
      }
      const int from = __hip_atomic_fetch_max(lds + 8, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (int st = from; st < m; ++st) {
        int* c = ctr + (i64)st * 2 * kSyncStride;
        const int prev = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int n = __hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 >= n) __hip_atomic_store(c + kSyncStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // about to gather from `step`; own_min = the step the slowest lane group of this wave is in
  __device__ __forceinline__ void wait_enter(int step, int own_min) {
    const int need = step - drift;
    if (!active || need < 0 || own_min <= need) return;
    int gave_up = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & (kWave - 1)) == 0 &&
        __hip_atomic_load(lds + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= need) {
      const int* rel = ctr + ((i64)need * 2 + 1) * kSyncStride;
      int it = 0;
      while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++it > 3000) { gave_up = 1; break; }   // ~2 ms without progress: give up pacing for good
      }
      if (!gave_up) __hip_atomic_fetch_max(lds + 9, need + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (__shfl(gave_up, 0)) active = false;
    t_wait += __builtin_amdgcn_s_memtime() - t0;
    n_wait += 1;
  }
  __device__ __forceinline__ void report(long long* dbg, long long t_start) const {
    if (dbg && (threadIdx.x & (kWave - 1)) == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      long long* d = dbg + ((long long)blockIdx.x * n_waves + (threadIdx.x >> 6)) * 4;
      d[0] = __builtin_amdgcn_s_memtime() - t_start; d[1] = t_wait; d[2] = n_wait + (t_feed << 16); d[3] = (active ? 0 : 16) + (xcc & 7);
    }
  }
};

template <int GPB>
__device__ __forceinline__ i64 walk_bin_index_of(const WalkView& s, int r, int g_in_blk) {   // bins per lane group
  const int slots = s.xcd_slots;
  const i64 x = blockIdx.x % slots, i = blockIdx.x / slots;
  const i64 per_slot = s.groups / slots;
  return ((i64)r * slots + x) * per_slot + i * GPB + g_in_blk;
}

template <int L>
__device__ __forceinline__ int wave_max_int(int v) {   // v group-uniform
  if constexpr (kWave / L == 1) return v;
  int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int g = 1; g < kWave / L; ++g) {
    const int o = __builtin_amdgcn_readlane(v, g * L);
    m = o > m ? o : m;
  }
  return m;
}

// ---- SpMM-type walk kernel: eight worker waves + four FEEDER waves per workgroup --------------------------
// Vector-memory operations of a wave return in issue order, so every small load between two batches
// of row requests -- the ids of the next batches, above all the per-slot weight w[edge] (a 4-byte
// gather over the whole edge array in the column-major passes) -- holds back the rows queued behind
// it for a trip to the Infinity Cache or HBM.  Here the worker waves issue row requests and nothing
// else: the feeder waves of the workgroup read the lane groups' (id, edge id) runs a chunk of
// kFeedChunk slots at a time, gathers the weights, and leaves ids and weights in a small LDS ring per
// lane group (kFeedRing chunks).  (Touching the next column window's share of the table ahead of the
// gathers was tried from the feeders and measured slower at every window size, 1.78 -> 2.0 ms per pass:
// two windows and the streams do not fit a 4 MiB L2 together.)
// Hand-over per (lane group, chunk) through two LDS words: `ready` = chunks written by the
// feeder, `done` = chunks the lane group has finished reading (LDS operations of a wave execute in
// order; the words are accessed with relaxed workgroup-scope atomics between compiler barriers).
constexpr int kFeedChunk = kWave; // slots per chunk (four batches): one per feeder lane
constexpr int kFeedRing = 2;      // chunks per ring
// One workgroup per CU: 8 worker waves + 4 feeder waves = three waves on every SIMD.  (Two workgroups
// of 5 or 6 waves do not pack: with <= 168 VGPRs a SIMD takes three waves, and a workgroup whose wave
// count is not a multiple of four leaves the second one waiting for a SIMD slot.)
constexpr int kWalkWorkers = 512;  // worker threads
constexpr int kFeeders = 4;        // feeder waves, each serving GPB / kFeeders lane groups
constexpr int kWalkThreads = kWalkWorkers + kFeeders * kWave;

// LDS of the SpMM-type walk kernel: per lane group K rows + a ring of (id, HV per-head weights) per slot
// (WS = 32-bit words per weight: 1 for fp32, 2 for fp64)
template <int L, int NV>
__host__ __device__ constexpr size_t spmm_walk_lds_bytes(int K, int HV, int WS = 1) {
  return (size_t)(kWalkWorkers / L) * ((size_t)K * L * NV * 16 + (size_t)kFeedChunk * kFeedRing * 4 * (1 + HV * WS));
}
// most rows per lane group that fit next to the rings (one workgroup per CU, 2 KB left to the static words)
template <int L, int NV>
__host__ __device__ constexpr int spmm_walk_rows(int HV, int WS = 1) {
  const long long per_group = (160LL * 1024 - 2048) / (kWalkWorkers / L) - (long long)kFeedChunk * kFeedRing * 4 * (1 + HV * WS);
  const long long k = per_group / ((long long)L * NV * 16);
  return (int)(k > kWalkK ? kWalkK : (k < 0 ? 0 : k));
}
typedef float walk_f4 __attribute__((ext_vector_type(4)));
typedef float walk_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int lds_ld(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  return v;
}
__device__ __forceinline__ void lds_st(int* p, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// out[row, f] = sum over the row's slots of w[edge, head(f)] * X[neighbour, f]   (graphop_kernel.cu:100-112, :118-130, :151-163)
// Bins per lane group (SH = 1): the LDS rows are the group's own.  HV = heads (a row is HV x d floats, d4 = d / 4
// float4s per head; w holds HV scalars per edge): the feeder stages all HV weights of a slot, a worker
// lane reads its head's.
// T = float or double (a row is 16 * L * NV bytes either way; several heads: fp32 only); OFF32 = false: tables of
// 4 GiB and more (the row offset is a 64-bit multiply-add per request instead of a 32-bit vector offset).
template <int L, int NV, int HV, typename T, bool OFF32>
__device__ __forceinline__ void spmm_walk_body(
    const WalkView& s, const T* __restrict__ wgt, const T* __restrict__ X, T* __restrict__ out, int d4) {
  using TR = RowT<T>;
  using vec = typename TR::vec;
  extern __shared__ float4 lds_raw[];
  vec* lds = reinterpret_cast<vec*>(lds_raw);
  constexpr int GPB = kWalkWorkers / L;
  constexpr i64 F4 = (i64)L * NV;
  constexpr int SB = StripCfg<L, NV>::SB;
  constexpr int RING = kFeedChunk * kFeedRing;
  constexpr int WS = TR::WORDS;                      // 32-bit words per weight
  constexpr int RSTRIDE = RING * (1 + HV * WS);      // ints per lane group's ring: ids, then HV weights per slot
  constexpr bool H1 = HV == 1;
  static_assert(sizeof(T) == 4 || HV == 1, "several heads: fp32 only");
  static_assert(L >= 16 && kWalkK <= L && SB == 16 && kFeedChunk % SB == 0, "lane k of a group holds the bin's k-th row");
  static_assert(HV == 1 || HV == 2 || HV == 4 || HV == 8, "heads");
  const int K = s.K;
  __shared__ int pace_words[16];
  __shared__ int feed_ready[GPB], feed_done[GPB];
  int* ring_base = reinterpret_cast<int*>(lds + (i64)GPB * K * F4);   // [GPB][RSTRIDE]
  constexpr int GW = kWave / L, NQ = GPB / GW;      // lane groups per wave; quads (see the worker waves) per workgroup
  static_assert(NQ <= 8, "pacer progress words");
  __shared__ int tk_next, quad_done[NQ], quad_len[NQ], bin_total[GPB], bin_seg[GPB];
  __shared__ int wg_abort;           // raised by walk_fail: every wave of the workgroup leaves
  if (threadIdx.x == 0) wg_abort = 0;
  if (threadIdx.x < GPB) { feed_ready[threadIdx.x] = 0; feed_done[threadIdx.x] = 0; bin_total[threadIdx.x] = 0; bin_seg[threadIdx.x] = 0; }
  if (threadIdx.x < NQ) { quad_done[threadIdx.x] = 0; quad_len[threadIdx.x] = SB; }
  if (threadIdx.x == 0) tk_next = 0;
  __syncthreads();                   // the zeroed hand-over words, whether or not the launch is paced
  const long long t_start = __builtin_amdgcn_s_memtime();
  WalkPacer pacer(s, pace_words, NQ);               // progress is kept per quad (below), not per wave

  if (threadIdx.x >= kWalkWorkers) {
    // ---------------- feeder wave ----------------
    // One wave instruction serves one lane group: a CHUNK of kFeedChunk = 64 slots, one per feeder lane.  Per
    // lane group the feeder holds two chunks in registers -- chunk c with its weights requested, chunk
    // c + 1 with its (id, edge id) pair requested -- and ADVANCES a group when the ring has room for
    // chunk c: write c, request the weights of c + 1, request the pair of c + 2.  Every request thus
    // has a whole chunk time (the group needs ~13 k cycles per chunk) to come back, and the groups
    // advance independently of each other.
    const int h = (threadIdx.x - kWalkWorkers) % kWave;
    constexpr int NG = GPB / kFeeders > 0 ? GPB / kFeeders : 1;   // lane groups of this feeder
    const int g0 = __builtin_amdgcn_readfirstlane(((threadIdx.x - kWalkWorkers) / kWave) * NG);   // (wave-uniform, and known to be)
    if (g0 >= GPB) return;
    int chunk_base[NG];                               // chunks of earlier rounds (wave-uniform)
#pragma unroll
    for (int g = 0; g < NG; ++g) chunk_base[g] = 0;
    long long t_space = 0;
    // identity-eid slots read every weight line once, a run at a time: streamed past the caches; permuted slots
    // share their lines with the neighbouring columns' lane groups of the same XCD: cached
    auto ld_w = [&](int wi, T (&o)[HV]) {
      if (wi < 0) {
#pragma unroll
        for (int i = 0; i < HV; ++i) o[i] = 0;
      } else if constexpr (HV == 1) {
        o[0] = s.stream_weights ? __builtin_nontemporal_load(wgt + wi) : wgt[wi];
      } else if constexpr (sizeof(T) != 4) {
      } else if constexpr (HV == 2) {
        const walk_f2* q = reinterpret_cast<const walk_f2*>(wgt) + wi;
        const walk_f2 t = s.stream_weights ? __builtin_nontemporal_load(q) : *q;
        o[0] = t.x; o[1] = t.y;
      } else {
#pragma unroll
        for (int i = 0; i < HV / 4; ++i) {
          const walk_f4* q = reinterpret_cast<const walk_f4*>(wgt) + (i64)wi * (HV / 4) + i;
          const walk_f4 t = s.stream_weights ? __builtin_nontemporal_load(q) : *q;
          o[4 * i] = t.x; o[4 * i + 1] = t.y; o[4 * i + 2] = t.z; o[4 * i + 3] = t.w;
        }
      }
    };
    for (int r = 0; r < s.rounds; ++r) {
      int pos0[NG], total[NG], nchunk[NG], c[NG];
      int idW[NG], idA[NG], wiA[NG], wi0[NG];
      T wvW[NG][HV];
      auto load_pair = [&](int g, int ck, int& idw, int& wi) {
        idw = 0; wi = -1;
        if (ck * kFeedChunk < total[g]) {
          const int j = ck * kFeedChunk + h;
          const int jc = j < total[g] ? j : total[g] - 1;   // the tail of the last chunk repeats the last neighbour, weight 0
          idw = __builtin_nontemporal_load(s.ids + pos0[g] + jc);
          if (j < total[g]) wi = __builtin_nontemporal_load(s.widx + pos0[g] + jc);
        }
      };
      int left = 0;
      {
        int p_l = 0, t_l = 0;                         // lane g < NG: run start / length of lane group g0 + g
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
      for (int g = 0; g < NG; ++g) load_pair(g, 0, idW[g], wi0[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) load_pair(g, 1, idA[g], wiA[g]);
#pragma unroll
      for (int g = 0; g < NG; ++g) ld_w(wi0[g], wvW[g]);
      while (left > 0) {
        // one TURN: every lane group whose ring has room advances by a chunk.  All ring writes first (they
        // use what the previous turn requested: one wait per turn), then all weight requests, then all
        // (id, edge id) requests.
        const int done_l = h < NG ? lds_ld(feed_done + g0 + h) : 0;
        unsigned adv = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (c[g] >= nchunk[g]) continue;                                // wave-uniform
          const int gc = chunk_base[g] + c[g];                            // running chunk number of this lane group
          if (gc - __builtin_amdgcn_readlane(done_l, g) >= kFeedRing) continue;   // ring full: the group is still reading
          if (s.fault == 1 && g0 + g == 0 && gc >= 1) continue;                   // (injected fault: chunk 1 of lane group 0 never comes)
          adv |= 1u << g;
          int* ring = ring_base + (g0 + g) * RSTRIDE;
          const int at = (gc % kFeedRing) * kFeedChunk + h;
          ring[at] = idW[g];
          if constexpr (HV == 1 && sizeof(T) == 4) {
            ring[RING + at] = __float_as_int(wvW[g][0]);
          } else if constexpr (HV == 1) {
            *reinterpret_cast<T*>(ring + RING + at * WS) = wvW[g][0];
          } else if constexpr (sizeof(T) != 4) {
          } else if constexpr (HV == 2) {
            *reinterpret_cast<float2*>(ring + RING + at * 2) = make_float2(wvW[g][0], wvW[g][1]);
          } else {
#pragma unroll
            for (int i = 0; i < HV / 4; ++i)
              *reinterpret_cast<float4*>(ring + RING + at * HV + 4 * i) =
                  make_float4(wvW[g][4 * i], wvW[g][4 * i + 1], wvW[g][4 * i + 2], wvW[g][4 * i + 3]);
          }
          lds_st(feed_ready + g0 + g, gc + 1);
        }
        if (adv == 0) {
          const long long t0 = s.dbg ? __builtin_amdgcn_s_memtime() : 0;
          __builtin_amdgcn_s_sleep(4);                    // (1 / 16 / 48 measured the same: 1.68 / 2.21 ms per pass)
          if (s.dbg) t_space += __builtin_amdgcn_s_memtime() - t0;
          // (no bound of its own: ring space comes from the worker waves, whose waits are all bounded and end in the
          // abort flag; a spin counter here -- one more live scalar and one more loop exit in the feeder's turn --
          // measured +0.65 ms per pass, profiles/r4_experiments.txt)
          if (walk_aborted(&wg_abort)) return;
          continue;
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (!((adv >> g) & 1)) continue;
          idW[g] = idA[g];
          ld_w(wiA[g], wvW[g]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (!((adv >> g) & 1)) continue;
          load_pair(g, c[g] + 2, idA[g], wiA[g]);
          ++c[g];
          --left;
        }
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) chunk_base[g] += nchunk[g];
    }
    if (s.dbg && h == 0 && g0 == 0) {
      long long* d = s.dbg + (long long)gridDim.x * 32 + (long long)blockIdx.x * 2;
      d[0] = __builtin_amdgcn_s_memtime() - t_start; d[1] = t_space;
    }
    return;
  }

  // ---------------- worker waves ----------------
  // The bins (lane groups' row sets, LDS-resident) belong to the WORKGROUP, not to a wave: a QUAD = the GW
  // bins one wave's lane groups process in lock step; (pacing step, quad) units are handed out through a
  // ticket counter in LDS, step-major.  A wave that is ahead simply takes the next unit -- of whichever
  // quad -- so the waves of a CU finish a step together however unevenly the memory system served them,
  // and the XCD pacer has to bound the spread of 32 CUs instead of 256 waves.  A unit starts when its
  // quad's previous step is finished (quad_done); between two units of a bin its state is in LDS: the
  // partial sums, the position in the run, the feeder ring.
  const int l = threadIdx.x % L;
  const int gq = (threadIdx.x / L) % GW;              // lane group inside the wave
  const int head = H1 ? 0 : l / d4;                    // NV == 1 when HV > 1: the lane's float4 belongs to one head
  static_assert(H1 || NV == 1, "several heads: one float4 per lane");
  const int n_steps = s.rounds * s.steps;
  for (;;) {
    int t = 0;
    if ((threadIdx.x & (kWave - 1)) == 0)
      t = __hip_atomic_fetch_add(&tk_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    t = __builtin_amdgcn_readfirstlane(t);
    const int gstep = t / NQ, q = t % NQ;             // step-major: the quads of a step are taken before the next step's
    if (gstep >= n_steps || walk_aborted(&wg_abort)) break;
    const int r = gstep / s.steps, sidx = gstep - r * s.steps;
    {
      int it = 0, dead = 0;                           // (every lane polls the same words: the values are made wave-uniform)
      const long long t0 = s.dbg ? __builtin_amdgcn_s_memtime() : 0;
      while (__builtin_amdgcn_readfirstlane(lds_ld(quad_done + q)) < gstep) {   // the quad's previous step is still being worked on
        __builtin_amdgcn_s_sleep(1);
        if (walk_aborted(&wg_abort)) { dead = 1; break; }
        if (++it > kWalkSpinQuad) { walk_fail(&wg_abort, kWalkErrQuad); dead = 1; break; }
      }
      if (s.dbg) pacer.t_feed += __builtin_amdgcn_s_memtime() - t0;
      if (dead) break;                                // never start a unit on a state that is not there
    }
    pacer.wait_enter(gstep, gstep);
    const int bin = q * GW + gq;
    vec* accs = lds + (i64)bin * K * F4;              // [K][NV][L]
    const int* ring = ring_base + bin * RSTRIDE;
    const i64 tb = walk_bin_index_of<GPB>(s, r, bin);
    int total, step_len;
    if (sidx == 0) {                                  // the bin's round starts with this unit
      total = s.bin_cum[tb];
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int v = 0; v < NV; ++v) accs[(k * NV + v) * L + l] = TR::zero();
      const int quad_total = wave_max_int<L>(total);
      // pacing steps = equal shares of the quad's longest run: equal positions in the runs mean nearly equal columns
      step_len = (((quad_total + s.steps - 1) / s.steps + SB - 1) / SB) * SB;
      step_len = step_len > 0 ? step_len : SB;
      if (l == 0) bin_total[bin] = total;
      if ((threadIdx.x & (kWave - 1)) == 0) quad_len[q] = step_len;
    } else {
      total = bin_total[bin];
      step_len = quad_len[q];
    }
    const int seg_base = bin_seg[bin];                // chunks of earlier rounds
    const int j0 = sidx * step_len;
    int j1 = j0 + step_len;
    {
      const int quad_total = wave_max_int<L>(total);
      j1 = j1 < quad_total ? j1 : quad_total;
    }
    vec acc[NV], pend_acc[NV], pend_rd[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = pend_acc[v] = pend_rd[v] = TR::zero();
    int k_cur = -1, pend_k = -1;
    // Row change: the partial sum of the row just left is folded into its LDS row by a DEFERRED
    // read-modify-write -- the LDS row is requested now, added and written back at the NEXT row
    // change -- so no LDS round trip sits between two slots' multiply-adds.  The bin is this lane
    // group's for the whole unit and a row's slots of one window are contiguous, so nobody touches the row in between.
    auto finish_pending = [&]() {
      if (pend_k >= 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          vec o = pend_rd[v];
          TR::add(o, pend_acc[v]);
          accs[(pend_k * NV + v) * L + l] = o;
        }
      }
    };
    auto row_change = [&](int kt) {
      finish_pending();
      pend_k = k_cur;
      if (k_cur >= 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          pend_acc[v] = acc[v];
          pend_rd[v] = accs[(k_cur * NV + v) * L + l];
          acc[v] = TR::zero();
        }
      }
      k_cur = kt;
    };
    struct Meta { int k; T w; };
    auto consume = [&](const vec (&x)[SB][NV], const Meta& m, const T (&wv)[H1 ? 1 : SB]) {
      static_for<SB>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        const int kt = group_bcast<L, u>(m.k);
        if (__builtin_expect(kt != k_cur, 0)) row_change(kt);   // group-uniform; about once per batch
        T w1;
        if constexpr (H1) w1 = group_bcast<L, u>(m.w);
        else w1 = wv[u];
#pragma unroll
        for (int v = 0; v < NV; ++v) TR::fma(acc[v], w1, x[u][v]);
      });
    };
    // one batch of rows in flight per lane group; id and weight of the next batch come from the feeder's ring
    vec x[SB][NV];
    Meta mc, mn;
    mc.k = mn.k = 0; mc.w = mn.w = 0;
    unsigned off_c = 0, off_n = 0;                    // OFF32: byte offset of the neighbour's row; else its row number
    // A chunk that never came (ring spin expired / workgroup aborting): the lane group carries on with neighbour 0
    // and weight 0 until the unit's loop ends -- the loop itself must stay free of exits: the compiler rotates it so
    // that the last row of a batch is consumed behind the first requests of the next one, and an exit test at the
    // loop end forces a full `s_waitcnt vmcnt(0)` drain per batch (measured: 1.69 -> 2.33 ms per pass) -- and the
    // wave leaves right behind the loop, before the rows are written or the step is reported.
    int ring_failed = 0;                              // per lane group
    auto stage = [&](int jb, Meta& m, unsigned& off) {
      if ((jb % kFeedChunk) == 0) {                   // entering a chunk: earlier ones are read, this one must be there
        const int gs = seg_base + jb / kFeedChunk;
        lds_st(feed_done + bin, gs);
        int it = 0;
        while (lds_ld(feed_ready + bin) <= gs) {
          __builtin_amdgcn_s_sleep(1);
          if (lds_ld(&wg_abort)) { ring_failed = 1; break; }
          if (++it > kWalkSpinRing) { walk_fail(&wg_abort, kWalkErrRing); ring_failed = 1; break; }
        }
      }
      const int at = (seg_base * kFeedChunk + jb + l) % RING;
      const int idw = ring_failed ? 0 : ring[at];
      m.k = (int)((unsigned)idw >> kWalkKShift);
      off = OFF32 ? (unsigned)(idw & kWalkIdMask) * (unsigned)(F4 * 16) : (unsigned)(idw & kWalkIdMask);
      if constexpr (H1 && sizeof(T) == 4) m.w = __int_as_float(ring[RING + at]);
      else if constexpr (H1) m.w = *reinterpret_cast<const T*>(ring + RING + at * WS);
      if constexpr (H1) m.w = ring_failed ? (T)0 : m.w;
    };
    if (j0 < total && j0 < j1) stage(j0, mc, off_c);
    for (int jb = j0; jb < j1; jb += SB) {
      if (jb < total) {
        static_for<SB>([&](auto uc) {
          constexpr int u = decltype(uc)::value;
          const unsigned o = group_bcast<L, u>(off_c);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            if constexpr (OFF32) x[u][v] = ld16_off<T>(X, o + (unsigned)((v * L + l) * 16));
            else x[u][v] = ld16_row64<T>(X, o, (unsigned)(F4 * 16), (unsigned)((v * L + l) * 16));
          }
        });
        T wv[H1 ? 1 : SB];
        if constexpr (!H1) {
          // this batch's weights of the lane's head -- read BEFORE the next batch is staged: entering a new chunk
          // there hands this batch's chunk back to the feeder (LDS operations of a wave execute in order)
          const int b0 = (seg_base * kFeedChunk + jb) % RING;    // a batch never wraps (RING % SB == 0)
#pragma unroll
          for (int u = 0; u < SB; ++u) wv[u] = (T)__int_as_float(ring[RING + (b0 + u) * HV + head]);
          asm volatile("" ::: "memory");
        } else {
          wv[0] = 0;
        }
        if (jb + SB < total && jb + SB < j1) stage(jb + SB, mn, off_n);
        consume(x, mc, wv);
        mc = mn; off_c = off_n;
      }
    }
    if (__any(ring_failed)) break;                    // (see above) nothing of this unit leaves the wave
    row_change(-1);      // the last row becomes the pending one ...
    finish_pending();    // ... and is folded in: the bin's state is in LDS again
    if (sidx == s.steps - 1) {
      // the bin's round is over: its rows leave the chip once -- plain stores for rows that are wholly inside
      // the bin, float atomics for the (at most two) rows it shares with its neighbours
      const int n_ch = (total + kFeedChunk - 1) / kFeedChunk;
      if (l == 0) bin_seg[bin] = seg_base + n_ch;
      lds_st(feed_done + bin, seg_base + n_ch);
      const int my_row = l < K ? s.bin_rows[tb * K + l] : -1;
      for (int k = 0; k < K; ++k) {
        const int rec = __shfl(my_row, k, L);
        if (rec == -1) continue;   // group-uniform
        const i64 row = rec & kWalkRowMask;
        vec a[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) a[v] = accs[(k * NV + v) * L + l];
        if (rec < 0) {
          atomic_flush_dense_t<L, NV, T>(out, row, a, l);
        } else {
#pragma unroll
          for (int v = 0; v < NV; ++v) reinterpret_cast<vec*>(out)[row * F4 + v * L + l] = a[v];
        }
      }
    }
    // this quad has finished the step (LDS operations of a wave execute in order: the state above is visible first)
    if ((threadIdx.x & (kWave - 1)) == 0 && !(s.fault == 2 && q == 0 && gstep == 0)) lds_st(quad_done + q, gstep + 1);
    pacer.signal_slot(q, gstep + 1);
  }
  walk_report(s.err, s.launch_id, &wg_abort);     // (the worker waves always get here: every spin of theirs polls the abort flag)
  pacer.report(s.dbg, t_start);
}

template <int L, int NV, int HV, bool OFF32 = true>
__global__ __launch_bounds__(kWalkThreads, 3) void k_spmm_walk_f32(   // (second bound: waves per SIMD)
    WalkView s, const float* __restrict__ wgt, const float* __restrict__ X, float* __restrict__ out, int d4) {
  spmm_walk_body<L, NV, HV, float, OFF32>(s, wgt, X, out, d4);
}
// fp64 (the reference dispatches both types through the same kernels, graphop_kernel.cu:291): one head, rows of 16 * L bytes
template <int L, int NV, bool OFF32 = true>
__global__ __launch_bounds__(kWalkThreads, 3) void k_spmm_walk_f64(
    WalkView s, const double* __restrict__ wgt, const double* __restrict__ X, double* __restrict__ out) {
  spmm_walk_body<L, NV, 1, double, OFF32>(s, wgt, X, out, L);
}

}  // namespace graphop
