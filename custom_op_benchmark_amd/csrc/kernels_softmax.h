#pragma once
#include "kernels_base.h"

namespace graphop {

// -------------------------------------------------------------------------------------------------
// Row-segment softmax (plan.row_owned).  Segment s = chunks [seg_chunk[s], seg_chunk[s+1]) =
// slots [indptr[c0], indptr[c1]); all of one row.  A group of G lanes owns a segment; items are
// the flattened (slot, head) pairs so that for eid == identity the reads are fully coalesced.
// Requires G % h == 0 (then a lane always sees the same head t = lane % h).
// Semantics: graphop_kernel.cu:170-202 (m starts at -1e9, :428).
// items per lane kept in registers (rows up to G*R items are read once).  Measured on Reddit-shape
// (mean row 492, 23 % of the rows above 512): forward best at 16, backward at 32.
// First slot of segment s: from the plan's per-segment array when it has one (one dependent load less in front of
// every row: short rows are bound by that chain), else through the chunk arrays.
__device__ __forceinline__ i64 seg_first(const i64* __restrict__ seg_eptr, const i64* __restrict__ seg_chunk,
                                         const i64* __restrict__ indptr, i64 s) {
  return seg_eptr ? seg_eptr[s] : indptr[seg_chunk[s]];
}

constexpr int kSoftmaxCacheFwd = 16;
constexpr int kSoftmaxCacheBwd = 32;

template <typename T>
__device__ __forceinline__ T neg_inf();
template <> __device__ __forceinline__ float neg_inf<float>() { return -INFINITY; }
template <> __device__ __forceinline__ double neg_inf<double>() { return -(double)INFINITY; }

// Segments longer than `long_len` slots are left to k_softmax_*_long (one workgroup per row).
template <typename T, int G, bool EID_ID>
__device__ __forceinline__ void softmax_fwd_seg_body(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ x, T* __restrict__ y, i64 n_seg, int h,
    i64 long_len, i64 block, const i64* __restrict__ row, T* __restrict__ stats) {
  constexpr int R = kSoftmaxCacheFwd;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const i64 items = len * h;
  const int t = l % h;

  auto offs = [&](i64 q) -> i64 {   // identity eid: (e0 + q/h)*h + q%h == e0*h + q
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  // whole row in registers: x read once, one exp per item.  Tiers by row length: the unrolled loops run all RR
  // iterations whatever the row holds
  auto in_regs = [&](auto rc) {
    constexpr int RR = decltype(rc)::value;
    T v[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const i64 q = l + (i64)r * G;
      v[r] = q < items ? x[offs(q)] : neg_inf<T>();
    }
    T m = (T)-1e9;
#pragma unroll
    for (int r = 0; r < RR; ++r) m = v[r] > m ? v[r] : m;
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) {
        const T m2 = __shfl_xor(m, mask, G);
        m = m > m2 ? m : m2;
      }
    T sum = 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      v[r] = (l + (i64)r * G) < items ? exp_le0(v[r] - m) : (T)0;
      sum += v[r];
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) sum += __shfl_xor(sum, mask, G);
    const T inv = (T)1 / sum;                     // one division per row; the items are scaled
#pragma unroll
    for (int r = 0; r < RR; ++r) {
      const i64 q = l + (i64)r * G;
      if (q < items) y[offs(q)] = v[r] * inv;
    }
    if (stats && l < h) {   // row statistics for the fused attention backward: (max, 1 / sum)
      const i64 o = (row[seg_chunk[s]] * h + l) * 2;
      stats[o] = m; stats[o + 1] = inv;
    }
  };
  if (items <= (i64)G * (R / 4)) { in_regs(std::integral_constant<int, R / 4>{}); return; }
  if (items <= (i64)G * (R / 2)) { in_regs(std::integral_constant<int, R / 2>{}); return; }
  if (items <= (i64)G * R) { in_regs(std::integral_constant<int, R>{}); return; }

  T m = (T)-1e9, sum = 0;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const T v = x[(EID_ID ? k : eid[k]) * h + t];
    if (v > m) {
      sum = sum * exp_le0(m - v) + (T)1;
      m = v;
    } else {
      sum += exp_le0(v - m);
    }
  }
#pragma unroll
  for (int mask = G / 2; mask >= 1; mask >>= 1) {
    if (mask >= h) {  // wave-uniform
      const T m2 = __shfl_xor(m, mask, G);
      const T s2 = __shfl_xor(sum, mask, G);
      const T mn = m > m2 ? m : m2;
      sum = sum * exp_le0(m - mn) + s2 * exp_le0(m2 - mn);
      m = mn;
    }
  }
  const T inv = (T)1 / sum;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    y[o] = exp_le0(x[o] - m) * inv;
  }
  if (stats && l < h) {
    const i64 o = (row[seg_chunk[s]] * h + l) * 2;
    stats[o] = m; stats[o + 1] = inv;
  }
}

// Backward: g = sum dy*y over the row; dx = dy*y - g*y   (graphop_kernel.cu:208-230)
template <typename T, int G, bool EID_ID>
__device__ __forceinline__ void softmax_bwd_seg_body(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ y, const T* __restrict__ dy,
    T* __restrict__ dx, i64 n_seg, int h, i64 long_len, i64 block) {
  // gathered through eid every cached item carries its own 64-bit offset: 8 per lane fit the
  // register file, 32 spilled (the identity form walks one base pointer with immediate offsets)
  constexpr int R = EID_ID ? kSoftmaxCacheBwd : 8;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const i64 items = len * h;
  const int t = l % h;

  auto offs = [&](i64 q) -> i64 {
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  auto in_regs = [&](auto rc) {
    constexpr int RR = decltype(rc)::value;
    T yy[RR], dd[RR];
    T g = 0;
    const int n_it = (int)items;
    if constexpr (EID_ID) {   // one base address + immediate offsets r*G
      const T* yp = y + e0 * h + l;
      const T* dp = dy + e0 * h + l;
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const bool ok = l + r * G < n_it;
        yy[r] = ok ? yp[r * G] : (T)0;
        dd[r] = ok ? dp[r * G] : (T)0;
      }
    } else {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const i64 q = l + (i64)r * G;
        yy[r] = 0; dd[r] = 0;
        if (q < items) {
          const i64 o = offs(q);
          yy[r] = y[o];
          dd[r] = dy[o];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) g += dd[r] * yy[r];
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= h) g += __shfl_xor(g, mask, G);
    if constexpr (EID_ID) {
      T* xp = dx + e0 * h + l;
#pragma unroll
      for (int r = 0; r < RR; ++r)
        if (l + r * G < n_it) xp[r * G] = dd[r] * yy[r] - g * yy[r];
    } else {
#pragma unroll
      for (int r = 0; r < RR; ++r) {
        const i64 q = l + (i64)r * G;
        if (q < items) dx[offs(q)] = dd[r] * yy[r] - g * yy[r];
      }
    }
  };
  if (items <= (i64)G * (R / 4)) { in_regs(std::integral_constant<int, R / 4>{}); return; }
  if (items <= (i64)G * (R / 2)) { in_regs(std::integral_constant<int, R / 2>{}); return; }
  if (items <= (i64)G * R) { in_regs(std::integral_constant<int, R>{}); return; }

  T g = 0;
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    g += dy[o] * y[o];
  }
#pragma unroll
  for (int mask = G / 2; mask >= 1; mask >>= 1)
    if (mask >= h) g += __shfl_xor(g, mask, G);
  for (i64 q = l; q < items; q += G) {
    const i64 k = e0 + q / h;
    const i64 o = (EID_ID ? k : eid[k]) * h + t;
    const T yy = y[o];
    dx[o] = dy[o] * yy - g * yy;
  }
}

// Long rows: one 256-thread workgroup per row segment listed in long_segs[] (rows above
// kLongSegment slots).  Up to 256*kBlockCache items are held in registers (one read of the inputs,
// one exp per item); longer rows loop twice.  Per-head partials are merged through LDS.
// Requires 256 % h == 0.
constexpr int kBlockCache = 8;

template <typename T, bool BWD>
__device__ __forceinline__ void block_merge(T& m, T& sum, T* sh_m, T* sh_s, int h) {
  const int tid = threadIdx.x;
  __syncthreads();                       // previous users of sh_m / sh_s are done reading
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= h; stride >>= 1) {   // tid and tid+stride share a head
    if (tid < stride) {
      if constexpr (!BWD) {
        const T m1 = sh_m[tid], m2 = sh_m[tid + stride];
        const T mn = m1 > m2 ? m1 : m2;
        sh_s[tid] = sh_s[tid] * exp_le0(m1 - mn) + sh_s[tid + stride] * exp_le0(m2 - mn);
        sh_m[tid] = mn;
      } else {
        sh_s[tid] += sh_s[tid + stride];
      }
    }
    __syncthreads();
  }
  m = sh_m[tid % h]; sum = sh_s[tid % h];
}

template <typename T, bool BWD, bool EID_ID>
__device__ __forceinline__ void softmax_long_body(
    const int* __restrict__ long_segs, const i64* __restrict__ seg_chunk,
    const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const i64* __restrict__ eid, const T* __restrict__ in0,
    const T* __restrict__ in1, T* __restrict__ out, int h, T* sh_m, T* sh_s, i64 long_len,
    const i64* __restrict__ row = nullptr, T* __restrict__ stats = nullptr) {
  constexpr int RB = kBlockCache;
  const i64 s = long_segs[blockIdx.x];
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  if (seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0 <= long_len) return;   // block-uniform: the per-row groups take it
  const i64 items = (seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0) * h;
  const int tid = threadIdx.x, t = tid % h;
  auto offs = [&](i64 q) -> i64 {   // 256 % h == 0, so q % h == t for every q of this thread
    if constexpr (EID_ID) return e0 * h + q;
    else return eid[e0 + q / h] * h + t;
  };
  if (items <= (i64)kFastBlock * RB) {
    T v[RB], u[BWD ? RB : 1];
    const int n_it = (int)items;
    // identity eid: one base address per array + immediate offsets r*256 (few address registers)
    const T* p0 = in0 + e0 * h + tid;
    const T* p1 = BWD ? in1 + e0 * h + tid : nullptr;
    T* po = out + e0 * h + tid;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int q = tid + r * kFastBlock;
      v[r] = BWD ? (T)0 : neg_inf<T>();
      if constexpr (BWD) u[r] = 0;
      if (q < n_it) {
        if constexpr (EID_ID) {
          v[r] = p0[r * kFastBlock];
          if constexpr (BWD) u[r] = p1[r * kFastBlock];
        } else {
          const i64 o = offs(q);
          v[r] = in0[o];
          if constexpr (BWD) u[r] = in1[o];
        }
      }
    }
    T m = (T)-1e9, sum = 0;
    if constexpr (!BWD) {
#pragma unroll
      for (int r = 0; r < RB; ++r) m = v[r] > m ? v[r] : m;
      // block max per head first, so every thread exponentiates against the final maximum
      __syncthreads();
      sh_m[tid] = m;
      __syncthreads();
      for (int stride = kFastBlock / 2; stride >= h; stride >>= 1) {
        if (tid < stride) { const T a = sh_m[tid], b2 = sh_m[tid + stride]; sh_m[tid] = a > b2 ? a : b2; }
        __syncthreads();
      }
      m = sh_m[t];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        v[r] = (tid + r * kFastBlock) < n_it ? exp_le0(v[r] - m) : (T)0;
        sum += v[r];
      }
      T mm = 0;
      block_merge<T, true>(mm, sum, sh_m, sh_s, h);
      const T inv = (T)1 / sum;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int q = tid + r * kFastBlock;
        if (q < n_it) {
          if constexpr (EID_ID) po[r * kFastBlock] = v[r] * inv;
          else out[offs(q)] = v[r] * inv;
        }
      }
      if (stats && tid < h) {
        const i64 o = (row[seg_chunk[s]] * h + tid) * 2;
        stats[o] = m; stats[o + 1] = inv;
      }
    } else {
#pragma unroll
      for (int r = 0; r < RB; ++r) sum += u[r] * v[r];
      block_merge<T, true>(m, sum, sh_m, sh_s, h);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int q = tid + r * kFastBlock;
        if (q < n_it) {
          if constexpr (EID_ID) po[r * kFastBlock] = u[r] * v[r] - sum * v[r];
          else out[offs(q)] = u[r] * v[r] - sum * v[r];
        }
      }
    }
    return;
  }
  T m = (T)-1e9, sum = 0;
  for (i64 q = tid; q < items; q += kFastBlock) {
    const i64 o = offs(q);
    if constexpr (!BWD) {
      const T v = in0[o];
      if (v > m) { sum = sum * exp_le0(m - v) + (T)1; m = v; }
      else sum += exp_le0(v - m);
    } else {
      sum += in1[o] * in0[o];
    }
  }
  block_merge<T, BWD>(m, sum, sh_m, sh_s, h);
  const T inv = BWD ? sum : (T)1 / sum;
  for (i64 q = tid; q < items; q += kFastBlock) {
    const i64 o = offs(q);
    if constexpr (!BWD) out[o] = exp_le0(in0[o] - m) * inv;
    else { const T yy = in0[o]; out[o] = in1[o] * yy - sum * yy; }
  }
  if constexpr (!BWD) {
    if (stats && tid < h) {
      const i64 o = (row[seg_chunk[s]] * h + tid) * 2;
      stats[o] = m; stats[o + 1] = inv;
    }
  }
}

// One launch: workgroups [0, n_long) take the hub rows (dispatched first, so their long serial
// loops overlap the bulk), the rest take kFastBlock/G ordinary row segments each.
template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_fwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ x, T* __restrict__ y, i64 n_seg, int h,
    i64 long_len, const int* __restrict__ long_segs, int n_long, const i64* __restrict__ row,
    T* __restrict__ stats) {
  __shared__ T sh_m[kFastBlock];
  __shared__ T sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_long_body<T, false, EID_ID>(long_segs, seg_chunk, indptr, seg_eptr, eid, x, (const T*)nullptr, y, h,
                                        sh_m, sh_s, long_len, row, stats);
  else
    softmax_fwd_seg_body<T, G, EID_ID>(seg_chunk, indptr, seg_eptr, eid, x, y, n_seg, h, long_len,
                                       (i64)blockIdx.x - n_long, row, stats);
}

template <typename T, int G, bool EID_ID>
__global__ __launch_bounds__(kFastBlock) void k_softmax_bwd_seg(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ y, const T* __restrict__ dy,
    T* __restrict__ dx, i64 n_seg, int h, i64 long_len, const int* __restrict__ long_segs,
    int n_long) {
  __shared__ T sh_m[kFastBlock];
  __shared__ T sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_long_body<T, true, EID_ID>(long_segs, seg_chunk, indptr, seg_eptr, eid, y, dy, dx, h, sh_m, sh_s, long_len);
  else
    softmax_bwd_seg_body<T, G, EID_ID>(seg_chunk, indptr, seg_eptr, eid, y, dy, dx, n_seg, h, long_len,
                                       (i64)blockIdx.x - n_long);
}

// -------------------------------------------------------------------------------------------------
// Several heads, identity eid, h % 4 == 0, fp32: the (slot, head) items of a row are len * h contiguous
// floats, read and written as float4s.  Component j of a lane's float4 belongs to head (4 * lane + j) % h
// for every float4 the lane touches (4 * G % h == 0), so a lane keeps four running statistics and lanes
// h / 4 apart are merged.  (The scalar form above reads a row of 492 slots x 8 heads twice with 4-byte
// loads in a latency-bound loop: 2.7 TB/s; this one holds rows up to G * 32 float4s in registers.)
__device__ __forceinline__ float4 f4_splat(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 f4_max(float4 a, float4 b) {
  return make_float4(a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z, a.w > b.w ? a.w : b.w);
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4_exp_sub(float4 a, float4 b) {
  return make_float4(exp_nonpos(a.x - b.x), exp_nonpos(a.y - b.y), exp_nonpos(a.z - b.z), exp_nonpos(a.w - b.w));
}
__device__ __forceinline__ float4 f4_rcp(float4 a) { return make_float4(1.f / a.x, 1.f / a.y, 1.f / a.z, 1.f / a.w); }
template <int G>
__device__ __forceinline__ float4 f4_shfl_xor(float4 a, int mask) {
  return make_float4(__shfl_xor(a.x, mask, G), __shfl_xor(a.y, mask, G), __shfl_xor(a.z, mask, G), __shfl_xor(a.w, mask, G));
}
// dx = dy * y - g * y
__device__ __forceinline__ float4 f4_bwd(float4 dy, float4 y, float4 g) {
  return make_float4(dy.x * y.x - g.x * y.x, dy.y * y.y - g.y * y.y, dy.z * y.z - g.z * y.z, dy.w * y.w - g.w * y.w);
}
// online softmax statistics of two partial rows merged: (m, sum) <- (m, sum) + (m2, s2)
__device__ __forceinline__ void f4_merge(float4& m, float4& sum, float4 m2, float4 s2) {
  const float4 mn = f4_max(m, m2);
  sum = f4_add(f4_mul(sum, f4_exp_sub(m, mn)), f4_mul(s2, f4_exp_sub(m2, mn)));
  m = mn;
}

constexpr int kVec4CacheFwd = 32;   // float4s per lane held in registers
constexpr int kVec4CacheBwd = 16;

template <int G, int R4, bool BWD>
__device__ __forceinline__ void softmax_vec4_regs(const float4* __restrict__ p0, const float4* __restrict__ p1,
                                                  float4* __restrict__ po, int n4, int l, int hq, float* st_row) {
    if constexpr (!BWD) {
      float4 v[R4];
#pragma unroll
      for (int r = 0; r < R4; ++r) v[r] = (l + r * G) < n4 ? p0[l + r * G] : f4_splat(-INFINITY);
      float4 m = f4_splat(-1e9f);
#pragma unroll
      for (int r = 0; r < R4; ++r) m = f4_max(m, v[r]);
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) m = f4_max(m, f4_shfl_xor<G>(m, mask));
      float4 sum = f4_splat(0.f);
#pragma unroll
      for (int r = 0; r < R4; ++r) {
        v[r] = (l + r * G) < n4 ? f4_exp_sub(v[r], m) : f4_splat(0.f);
        sum = f4_add(sum, v[r]);
      }
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) sum = f4_add(sum, f4_shfl_xor<G>(sum, mask));
      const float4 inv = f4_rcp(sum);              // one division per row and head; items are scaled
#pragma unroll
      for (int r = 0; r < R4; ++r)
        if ((l + r * G) < n4) po[l + r * G] = f4_mul(v[r], inv);
      if (st_row) {
        st_row[0] = m.x; st_row[1] = inv.x; st_row[2] = m.y; st_row[3] = inv.y;
        st_row[4] = m.z; st_row[5] = inv.z; st_row[6] = m.w; st_row[7] = inv.w;
      }
    } else {
      float4 yy[R4], dd[R4];
      float4 g = f4_splat(0.f);
#pragma unroll
      for (int r = 0; r < R4; ++r) {
        const bool ok = (l + r * G) < n4;
        yy[r] = ok ? p0[l + r * G] : f4_splat(0.f);
        dd[r] = ok ? p1[l + r * G] : f4_splat(0.f);
      }
#pragma unroll
      for (int r = 0; r < R4; ++r) g = f4_add(g, f4_mul(dd[r], yy[r]));
#pragma unroll
      for (int mask = G / 2; mask >= 1; mask >>= 1)
        if (mask >= hq) g = f4_add(g, f4_shfl_xor<G>(g, mask));
#pragma unroll
      for (int r = 0; r < R4; ++r)
        if ((l + r * G) < n4) po[l + r * G] = f4_bwd(dd[r], yy[r], g);
    }
}

// in0 = x (forward) | y (backward), in1 = dy.  Semantics as softmax_*_seg_body (graphop_kernel.cu:170-230).
// RMAX = most float4s per lane the register tiers may hold: the kernel's register count -- and with it how many
// waves a SIMD holds -- follows the largest tier compiled in.  Graphs of short rows (products-shape: 25 slots
// x 8 heads = 50 float4s per row) are bound by the chain of dependent loads per row (segment bounds, slot bounds,
// items), i.e. by resident waves: they take the RMAX = 8 instantiation (longer rows loop).
template <int G, bool BWD, int RMAX>
__device__ __forceinline__ void softmax_vec4_group(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ in0,
    const float* __restrict__ in1, float* __restrict__ out, i64 n_seg, int h, i64 long_len, i64 block,
    const i64* __restrict__ row, float* __restrict__ stats) {
  constexpr int R4 = RMAX;
  const int l = threadIdx.x % G;
  const i64 s = block * (kFastBlock / G) + threadIdx.x / G;
  if (s >= n_seg) return;                         // group-uniform
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len > long_len) return;
  const int n4 = (int)(len * h / 4);
  const int hq = h / 4;                           // lanes hq apart hold the same heads
  const float4* p0 = reinterpret_cast<const float4*>(in0 + e0 * h);
  const float4* p1 = BWD ? reinterpret_cast<const float4*>(in1 + e0 * h) : nullptr;
  float4* po = reinterpret_cast<float4*>(out + e0 * h);
  // whole row in registers: inputs read once, one exp per item.  Tiers by row length: the unrolled loops run all
  // R4 iterations whatever the row holds (a fixed 32 made the pass VALU-bound: 1.8 ms against 1.0 of traffic)
  float* st_row = (stats && l < hq) ? stats + (row[seg_chunk[s]] * h + 4 * l) * 2 : nullptr;
  if (n4 <= G * 4) { softmax_vec4_regs<G, 4, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  if (n4 <= G * 8) { softmax_vec4_regs<G, 8, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  if constexpr (R4 >= 16) {
    if (n4 <= G * 16) { softmax_vec4_regs<G, 16, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  }
  if constexpr (R4 > 16) {
    if (n4 <= G * R4) { softmax_vec4_regs<G, R4, BWD>(p0, p1, po, n4, l, hq, st_row); return; }
  }
  constexpr int U = 4;
  if constexpr (!BWD) {
    float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * G) < n4 ? p0[q0 + u * G] : f4_splat(-INFINITY);
      float4 mb = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mb = f4_max(mb, t[u]);
      sum = f4_mul(sum, f4_exp_sub(m, mb));
#pragma unroll
      for (int u = 0; u < U; ++u) sum = f4_add(sum, f4_exp_sub(t[u], mb));   // exp(-inf) = 0 past the end
      m = mb;
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= hq) f4_merge(m, sum, f4_shfl_xor<G>(m, mask), f4_shfl_xor<G>(sum, mask));
    const float4 inv = f4_rcp(sum);
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * G) < n4 ? p0[q0 + u * G] : f4_splat(0.f);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) po[q0 + u * G] = f4_mul(f4_exp_sub(t[u], m), inv);
    }
    if (stats && l < hq) {
      float* o = stats + (row[seg_chunk[s]] * h + 4 * l) * 2;
      o[0] = m.x; o[1] = inv.x; o[2] = m.y; o[3] = inv.y;
      o[4] = m.z; o[5] = inv.z; o[6] = m.w; o[7] = inv.w;
    }
  } else {
    float4 g = f4_splat(0.f);
    for (int q0 = l; q0 < n4; q0 += U * G) {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) g = f4_add(g, f4_mul(p1[q0 + u * G], p0[q0 + u * G]));
    }
#pragma unroll
    for (int mask = G / 2; mask >= 1; mask >>= 1)
      if (mask >= hq) g = f4_add(g, f4_shfl_xor<G>(g, mask));
    for (int q0 = l; q0 < n4; q0 += U * G) {
      float4 ty[U], td[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = (q0 + u * G) < n4;
        ty[u] = ok ? p0[q0 + u * G] : f4_splat(0.f);
        td[u] = ok ? p1[q0 + u * G] : f4_splat(0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * G) < n4) po[q0 + u * G] = f4_bwd(td[u], ty[u], g);
    }
  }
}

// A workgroup's row held in registers (up to 256 * R4 float4s): inputs read once, one exp per item -- every thread
// exponentiates against its OWN maximum, the (max, sum) pairs are merged through LDS, and the items are rescaled
// by exp(own max - row max) / row sum.
template <int R4, bool BWD>
__device__ __forceinline__ void softmax_vec4_long_regs(const float4* __restrict__ p0, const float4* __restrict__ p1,
                                                       float4* __restrict__ po, int n4, int hq, float4* sh_m,
                                                       float4* sh_s, float* st_row) {
  const int tid = threadIdx.x;
  float4 a[R4], b[BWD ? R4 : 1];
#pragma unroll
  for (int r = 0; r < R4; ++r) {
    const bool ok = (tid + r * kFastBlock) < n4;
    a[r] = ok ? p0[tid + r * kFastBlock] : f4_splat(BWD ? 0.f : -INFINITY);
    if constexpr (BWD) b[r] = ok ? p1[tid + r * kFastBlock] : f4_splat(0.f);
  }
  float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
  if constexpr (!BWD) {
#pragma unroll
    for (int r = 0; r < R4; ++r) m = f4_max(m, a[r]);
#pragma unroll
    for (int r = 0; r < R4; ++r) {
      a[r] = f4_exp_sub(a[r], m);                 // 0 for the -inf of a padding slot
      sum = f4_add(sum, a[r]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < R4; ++r) sum = f4_add(sum, f4_mul(b[r], a[r]));
  }
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= hq; stride >>= 1) {
    if (tid < stride) {
      if constexpr (!BWD) {
        float4 x = sh_m[tid], y = sh_s[tid];
        f4_merge(x, y, sh_m[tid + stride], sh_s[tid + stride]);
        sh_m[tid] = x; sh_s[tid] = y;
      } else {
        sh_s[tid] = f4_add(sh_s[tid], sh_s[tid + stride]);
      }
    }
    __syncthreads();
  }
  const float4 M = sh_m[tid % hq], S = sh_s[tid % hq];
  if constexpr (!BWD) {
    const float4 inv = f4_rcp(S);
    const float4 c = f4_mul(f4_exp_sub(m, M), inv);
#pragma unroll
    for (int r = 0; r < R4; ++r)
      if ((tid + r * kFastBlock) < n4) po[tid + r * kFastBlock] = f4_mul(a[r], c);
    if (st_row) {
      st_row[0] = M.x; st_row[1] = inv.x; st_row[2] = M.y; st_row[3] = inv.y;
      st_row[4] = M.z; st_row[5] = inv.z; st_row[6] = M.w; st_row[7] = inv.w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R4; ++r)
      if ((tid + r * kFastBlock) < n4) po[tid + r * kFastBlock] = f4_bwd(b[r], a[r], S);
  }
}

// Rows above long_len slots: one workgroup per row, float4 items, statistics merged through LDS.
template <bool BWD, int RMAX>
__device__ __forceinline__ void softmax_vec4_long(
    const int* __restrict__ long_segs, const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const float* __restrict__ in0, const float* __restrict__ in1, float* __restrict__ out, int h,
    float4* sh_m, float4* sh_s, i64 long_len, const i64* __restrict__ row, float* __restrict__ stats) {
  const i64 s = long_segs[blockIdx.x];
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s);
  const i64 len = seg_first(seg_eptr, seg_chunk, indptr, s + 1) - e0;
  if (len <= long_len) return;                    // block-uniform: the per-row groups take it
  const i64 n4 = len * h / 4;
  const int tid = threadIdx.x, hq = h / 4;
  const float4* p0 = reinterpret_cast<const float4*>(in0 + e0 * h);
  const float4* p1 = BWD ? reinterpret_cast<const float4*>(in1 + e0 * h) : nullptr;
  float4* po = reinterpret_cast<float4*>(out + e0 * h);
  if (n4 <= (RMAX >= 16 ? 16 : 8) * kFastBlock) {  // block-uniform
    float* st_row = (!BWD && stats && tid < hq) ? stats + (row[seg_chunk[s]] * h + 4 * tid) * 2 : nullptr;
    if (RMAX < 16 || n4 <= 8 * kFastBlock) softmax_vec4_long_regs<8, BWD>(p0, p1, po, (int)n4, hq, sh_m, sh_s, st_row);
    else softmax_vec4_long_regs<(RMAX >= 16 ? 16 : 8), BWD>(p0, p1, po, (int)n4, hq, sh_m, sh_s, st_row);
    return;
  }
  constexpr int U = 4;
  float4 m = f4_splat(-1e9f), sum = f4_splat(0.f);
  for (i64 q0 = tid; q0 < n4; q0 += U * kFastBlock) {
    if constexpr (!BWD) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = (q0 + u * kFastBlock) < n4 ? p0[q0 + u * kFastBlock] : f4_splat(-INFINITY);
      float4 mb = m;
#pragma unroll
      for (int u = 0; u < U; ++u) mb = f4_max(mb, t[u]);
      sum = f4_mul(sum, f4_exp_sub(m, mb));
#pragma unroll
      for (int u = 0; u < U; ++u) sum = f4_add(sum, f4_exp_sub(t[u], mb));
      m = mb;
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((q0 + u * kFastBlock) < n4) sum = f4_add(sum, f4_mul(p1[q0 + u * kFastBlock], p0[q0 + u * kFastBlock]));
    }
  }
  sh_m[tid] = m; sh_s[tid] = sum;
  __syncthreads();
  for (int stride = kFastBlock / 2; stride >= hq; stride >>= 1) {   // tid and tid + stride hold the same heads
    if (tid < stride) {
      if constexpr (!BWD) {
        float4 a = sh_m[tid], b = sh_s[tid];
        f4_merge(a, b, sh_m[tid + stride], sh_s[tid + stride]);
        sh_m[tid] = a; sh_s[tid] = b;
      } else {
        sh_s[tid] = f4_add(sh_s[tid], sh_s[tid + stride]);
      }
    }
    __syncthreads();
  }
  m = sh_m[tid % hq]; sum = sh_s[tid % hq];
  const float4 inv = BWD ? sum : f4_rcp(sum);
  for (i64 q0 = tid; q0 < n4; q0 += U * kFastBlock) {
    float4 t0[U], t1[BWD ? U : 1];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = (q0 + u * kFastBlock) < n4;
      t0[u] = ok ? p0[q0 + u * kFastBlock] : f4_splat(0.f);
      if constexpr (BWD) t1[u] = ok ? p1[q0 + u * kFastBlock] : f4_splat(0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((q0 + u * kFastBlock) < n4) {
        if constexpr (!BWD) po[q0 + u * kFastBlock] = f4_mul(f4_exp_sub(t0[u], m), inv);
        else po[q0 + u * kFastBlock] = f4_bwd(t1[u], t0[u], sum);
      }
  }
  if constexpr (!BWD) {
    if (stats && tid < hq) {
      float* o = stats + (row[seg_chunk[s]] * h + 4 * tid) * 2;
      o[0] = m.x; o[1] = inv.x; o[2] = m.y; o[3] = inv.y;
      o[4] = m.z; o[5] = inv.z; o[6] = m.w; o[7] = inv.w;
    }
  }
}

template <int G, int RMAX>
__global__ __launch_bounds__(kFastBlock) void k_softmax_fwd_vec4(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ x,
    float* __restrict__ y, i64 n_seg, int h, i64 long_len, const int* __restrict__ long_segs, int n_long,
    const i64* __restrict__ row, float* __restrict__ stats) {
  __shared__ float4 sh_m[kFastBlock];
  __shared__ float4 sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_vec4_long<false, RMAX>(long_segs, seg_chunk, indptr, seg_eptr, x, nullptr, y, h, sh_m, sh_s, long_len, row, stats);
  else
    softmax_vec4_group<G, false, RMAX>(seg_chunk, indptr, seg_eptr, x, nullptr, y, n_seg, h, long_len, (i64)blockIdx.x - n_long, row, stats);
}

template <int G, int RMAX>
__global__ __launch_bounds__(kFastBlock) void k_softmax_bwd_vec4(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr, const float* __restrict__ y,
    const float* __restrict__ dy, float* __restrict__ dx, i64 n_seg, int h, i64 long_len,
    const int* __restrict__ long_segs, int n_long) {
  __shared__ float4 sh_m[kFastBlock];
  __shared__ float4 sh_s[kFastBlock];
  if ((int)blockIdx.x < n_long)
    softmax_vec4_long<true, RMAX>(long_segs, seg_chunk, indptr, seg_eptr, y, dy, dx, h, sh_m, sh_s, long_len, nullptr, nullptr);
  else
    softmax_vec4_group<G, true, RMAX>(seg_chunk, indptr, seg_eptr, y, dy, dx, n_seg, h, long_len, (i64)blockIdx.x - n_long, nullptr, nullptr);
}

// Any h (G need not be a multiple of h): heads in an outer loop, strided reads.
template <typename T, bool BWD>
__global__ __launch_bounds__(kFastBlock) void k_softmax_seg_anyh(
    const i64* __restrict__ seg_chunk, const i64* __restrict__ indptr, const i64* __restrict__ seg_eptr,
    const i64* __restrict__ eid, const T* __restrict__ in0 /* x | y */,
    const T* __restrict__ in1 /* - | dy */, T* __restrict__ out, i64 n_seg, i64 h,
    const i64* __restrict__ row = nullptr, T* __restrict__ stats = nullptr) {
  const int lane = threadIdx.x & 63;
  const i64 s = (i64)blockIdx.x * (kFastBlock / kWave) + (threadIdx.x >> 6);
  if (s >= n_seg) return;
  const i64 e0 = seg_first(seg_eptr, seg_chunk, indptr, s), e1 = seg_first(seg_eptr, seg_chunk, indptr, s + 1);
  for (i64 t = 0; t < h; ++t) {
    if constexpr (!BWD) {
      T m = (T)-1e9;
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const T v = in0[eid[k] * h + t];
        m = v > m ? v : m;
      }
#pragma unroll
      for (int mask = 32; mask >= 1; mask >>= 1) {
        const T m2 = __shfl_xor(m, mask);
        m = m > m2 ? m : m2;
      }
      T sum = 0;
      for (i64 k = e0 + lane; k < e1; k += kWave) sum += exp_t(in0[eid[k] * h + t] - m);
      sum = wave_sum(sum);
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        out[o] = exp_t(in0[o] - m) / sum;
      }
      if (stats && lane == 0) {
        const i64 o = (row[seg_chunk[s]] * h + t) * 2;
        stats[o] = m; stats[o + 1] = (T)1 / sum;
      }
    } else {
      T g = 0;
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        g += in1[o] * in0[o];
      }
      g = wave_sum(g);
      for (i64 k = e0 + lane; k < e1; k += kWave) {
        const i64 o = eid[k] * h + t;
        out[o] = in1[o] * in0[o] - g * in0[o];
      }
    }
  }
}

}  // namespace graphop
