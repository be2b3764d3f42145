// Shared device helpers of the fp32 fast paths (gfx950, wave64).  Feature rows are walked as float4 (16 B / lane);
// a node row of F = h*d floats is covered by a GROUP of L lanes x NV float4 slots (F = 4*L*NV).  For the headline
// shape (d = 64, h = 1) L = 16: one 256-B row per 16-lane DPP row, four edges in flight per wave instruction,
// reductions by DPP inside the row.  Here: row / stream loads, the per-row slot-range loops of the chunk drivers
// (sddmm_range / spmm_range) and the float-atomic row flushes.
#pragma once
#include "common.h"

namespace graphop {

constexpr int kFastBlock = 256;

template <int L>
struct GroupCfg {
  static constexpr int kGroupsPerBlock = kFastBlock / L;
  static constexpr int kEdgeBatch = L < 16 ? L : 16;  // slots whose ids one index load covers
};

template <int NV>
struct Unroll {  // neighbour rows in flight per group (16*NV*U bytes per lane)
  static constexpr int value = NV == 1 ? 8 : (NV == 2 ? 4 : 2);
};

__device__ __forceinline__ float4 ld4(const float* base, i64 f4_index) {
  return reinterpret_cast<const float4*>(base)[f4_index];
}
typedef float vfloat4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* base, i64 f4_index) {   // streamed once: keep it out of the caches
  const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(base) + f4_index);
  return make_float4(v.x, v.y, v.z, v.w);
}

// Row slice of a gathered table.  OFF32: the table is < 4 GiB, so the byte offset fits 32 bits and
// the load uses the scalar-base + 32-bit vector-offset form (no 64-bit VALU address math).
template <bool OFF32>
__device__ __forceinline__ float4 ld_row(const float* base, int src, int f4_in_row, int row_f4) {
  if constexpr (OFF32) {
    const unsigned off = ((unsigned)src * (unsigned)row_f4 + (unsigned)f4_in_row) * 16u;
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
  } else {
    return reinterpret_cast<const float4*>(base)[(i64)src * row_f4 + f4_in_row];
  }
}

// 16 bytes at base + off, off < 4 GiB: scalar base + 32-bit vector offset (no 64-bit VALU add)
__device__ __forceinline__ float4 ld4_off(const float* base, unsigned off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (size_t)off);
}

// ---- rows by scalar type ---------------------------------------------------------------------------------
// The plan-driven kernels that exist for fp32 AND fp64 (the reference dispatches both through the same kernels,
// graphop_kernel.cu:291) are written against 16-byte row pieces: a lane holds `vec` = float4 or double2, a row of
// 16 * L * NV bytes is covered by L lanes x NV pieces whatever the scalar type (fp64 d = 64 = 512 B = the fp32
// d = 128 lane-group shape).  RowT<float> spells out exactly the operations the fp32 kernels always used.
template <typename T> struct RowT;
template <> struct RowT<float> {
  using vec = float4;
  static constexpr int N = 4;            // scalars per 16-byte piece
  static constexpr int WORDS = 1;        // 32-bit words per scalar
  static __device__ __forceinline__ vec zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void fma(vec& acc, float w, const vec& x) {
    acc.x = fmaf(w, x.x, acc.x); acc.y = fmaf(w, x.y, acc.y); acc.z = fmaf(w, x.z, acc.z); acc.w = fmaf(w, x.w, acc.w);
  }
  static __device__ __forceinline__ void add(vec& a, const vec& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
  static __device__ __forceinline__ float dot(const vec& a, const vec& b) { return dot4(a, b); }
  static __device__ __forceinline__ float comp(const vec& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)); }
};
template <> struct RowT<double> {
  using vec = double2;
  static constexpr int N = 2;
  static constexpr int WORDS = 2;
  static __device__ __forceinline__ vec zero() { return make_double2(0.0, 0.0); }
  static __device__ __forceinline__ void fma(vec& acc, double w, const vec& x) { acc.x = ::fma(w, x.x, acc.x); acc.y = ::fma(w, x.y, acc.y); }
  static __device__ __forceinline__ void add(vec& a, const vec& b) { a.x += b.x; a.y += b.y; }
  static __device__ __forceinline__ double dot(const vec& a, const vec& b) { return ::fma(a.y, b.y, a.x * b.x); }
  static __device__ __forceinline__ double comp(const vec& a, int i) { return i == 0 ? a.x : a.y; }
};
// 16 bytes at base + off, off < 4 GiB (scalar base + 32-bit vector offset), by scalar type
template <typename T>
__device__ __forceinline__ typename RowT<T>::vec ld16_off(const T* base, unsigned off) {
  return *reinterpret_cast<const typename RowT<T>::vec*>(reinterpret_cast<const char*>(base) + (size_t)off);
}
// 16 bytes at base + row * row_bytes + off: tables of 4 GiB and more (one 64-bit multiply-add per request)
template <typename T>
__device__ __forceinline__ typename RowT<T>::vec ld16_row64(const T* base, unsigned row, unsigned row_bytes, unsigned off) {
  return *reinterpret_cast<const typename RowT<T>::vec*>(reinterpret_cast<const char*>(base) + ((size_t)row * row_bytes + off));
}
template <typename T>
__device__ __forceinline__ typename RowT<T>::vec ld16(const T* base, i64 piece) {
  return reinterpret_cast<const typename RowT<T>::vec*>(base)[piece];
}

template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st_stream(float* p, float v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// -------------------------------------------------------------------------------------------------
// y[e*h + k] = <a[k], B[src, k, :]> for slots [lo, hi) (all of one row whose features are in a[]).
//   EDGE_B = false: src = idx[j], B (n_b, h, d)        (graphop_kernel.cu:40-55, :135-149)
//   EDGE_B = true : src = eid[j], B (n_edges, d), h==1  (node_mul_edge, :19-34)
//   EID_ID: eid[j] == j (skip the load).  IT: int64 API arrays or the plan's int32 mirrors.
//   H1: h == 1 -> results are collected across lanes and stored coalesced.
template <int L, int NV, bool H1, bool EDGE_B, bool EID_ID, bool NT, typename IT>
__device__ __forceinline__ void sddmm_range(const float4 (&a)[NV], i64 lo, i64 hi,
                                            const IT* __restrict__ eid,
                                            const IT* __restrict__ idx,
                                            const float* __restrict__ B, float* __restrict__ y,
                                            int h, int d4, int l) {
  constexpr int EB = GroupCfg<L>::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  for (i64 jb = lo; jb < hi; jb += EB) {
    const int nb = (hi - jb) < EB ? (int)(hi - jb) : EB;
    int my_e = -1, my_src = 0;
    if (l < nb) {
      my_e = EID_ID ? (int)(jb + l) : (int)ld_stream<NT>(eid + jb + l);
      my_src = EDGE_B ? my_e : (int)ld_stream<NT>(idx + jb + l);
    }
    float res = 0.f;
    for (int t = 0; t < nb; t += U) {
      float4 b[U][NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tt = (t + u) < nb ? (t + u) : (nb - 1);
        const i64 src = __shfl(my_src, tt, L);
#pragma unroll
        for (int v = 0; v < NV; ++v) b[u][v] = ld4(B, src * F4 + v * L + l);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (H1) {
          float p = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) p += dot4(a[v], b[u][v]);
          p = group_sum<L>(p);
          if (l == t + u) res = p;
        } else {
          const bool live = (t + u) < nb;
          const int tt = live ? (t + u) : (nb - 1);
          const i64 e = __shfl(my_e, tt, L);
          if (d4 >= L) {  // a head spans d4/L whole slots: add them, then reduce the group
            const int sph = d4 / L;
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              acc += dot4(a[v], b[u][v]);
              if ((v + 1) % sph == 0) {
                const float s = group_sum<L>(acc);
                if (live && l == 0) y[e * h + v / sph] = s;
                acc = 0.f;
              }
            }
          } else {  // a slot holds L/d4 heads: reduce sub-groups of d4 lanes
            const int hps = L / d4;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              const float s = group_sum_rt(dot4(a[v], b[u][v]), d4);
              if (live && (l % d4) == 0) y[e * h + v * hps + l / d4] = s;
            }
          }
        }
      }
    }
    if constexpr (H1) {
      if (my_e >= 0) st_stream<NT>(y + my_e, res);
    }
  }
}

// acc[f] += sum_{k in [lo,hi)} w[eid[k]*h + head(f)] * X[idx[k], f]
//   (graphop_kernel.cu:100-112 dA/dB, :118-130 forward, :151-163 dx)
template <int L, int NV, bool H1, bool EID_ID, bool NT, typename IT>
__device__ __forceinline__ void spmm_range(float4 (&acc)[NV], i64 lo, i64 hi,
                                           const IT* __restrict__ eid, const IT* __restrict__ idx,
                                           const float* __restrict__ w,
                                           const float* __restrict__ X, int h,
                                           const int (&hv)[NV], int l) {
  constexpr int EB = GroupCfg<L>::kEdgeBatch;
  constexpr int U = Unroll<NV>::value < EB ? Unroll<NV>::value : EB;
  constexpr i64 F4 = (i64)L * NV;
  for (i64 jb = lo; jb < hi; jb += EB) {
    const int nb = (hi - jb) < EB ? (int)(hi - jb) : EB;
    int my_e = 0, my_src = 0;
    float my_w = 0.f;
    if (l < nb) {
      my_e = EID_ID ? (int)(jb + l) : (int)ld_stream<NT>(eid + jb + l);
      my_src = (int)ld_stream<NT>(idx + jb + l);
      if constexpr (H1) my_w = EID_ID ? ld_stream<NT>(w + my_e) : w[my_e];
    }
    for (int t = 0; t < nb; t += U) {
      float4 x[U][NV];
      float wt[U][H1 ? 1 : NV];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = (t + u) < nb;
        const int tt = live ? (t + u) : (nb - 1);
        const i64 src = __shfl(my_src, tt, L);
        if constexpr (H1) {
          const float ww = __shfl(my_w, tt, L);
          wt[u][0] = live ? ww : 0.f;
        } else {
          const i64 e = __shfl(my_e, tt, L);
#pragma unroll
          for (int v = 0; v < NV; ++v) wt[u][v] = live ? w[e * h + hv[v]] : 0.f;
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) x[u][v] = ld4(X, src * F4 + v * L + l);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float ww = wt[u][H1 ? 0 : v];
          acc[v].x = fmaf(ww, x[u][v].x, acc[v].x);
          acc[v].y = fmaf(ww, x[u][v].y, acc[v].y);
          acc[v].z = fmaf(ww, x[u][v].z, acc[v].z);
          acc[v].w = fmaf(ww, x[u][v].w, acc[v].w);
        }
    }
  }
}

template <int L, int NV>
__device__ __forceinline__ void atomic_flush(float* __restrict__ out, i64 row,
                                             const float4 (&acc)[NV], int l) {
  constexpr i64 F4 = (i64)L * NV;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float* p = out + (row * F4 + v * L + l) * 4;
    atomicAdd(p + 0, acc[v].x);
    atomicAdd(p + 1, acc[v].y);
    atomicAdd(p + 2, acc[v].z);
    atomicAdd(p + 3, acc[v].w);
  }
}

// Same sum, but every atomic wave-instruction covers CONSECUTIVE dwords of the row (a group's L
// lanes add L consecutive floats = whole 64-B memory-side atomic requests) instead of one dword
// out of every 16 B: for flushes that are frequent enough to load the memory-side atomic units.
// Group-uniform call (all L lanes of the group active).
template <int L, int NV>
__device__ __forceinline__ void atomic_flush_dense(float* __restrict__ out, i64 row,
                                                   const float4 (&acc)[NV], int l) {
  constexpr i64 F = 4LL * L * NV;
  float* base = out + row * F;
  if constexpr (L >= 4) {
    const int comp = l & 3;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int src = i * (L / 4) + (l >> 2);
        const float x = __shfl(acc[v].x, src, L), y = __shfl(acc[v].y, src, L);
        const float z = __shfl(acc[v].z, src, L), w = __shfl(acc[v].w, src, L);
        const float val = comp == 0 ? x : (comp == 1 ? y : (comp == 2 ? z : w));
        atomicAdd(base + (v * 4 + i) * L + l, val);
      }
    }
  } else {
    atomic_flush<L, NV>(out, row, acc, l);
  }
}

// atomic_flush_dense by scalar type: every atomic wave-instruction of the group covers L consecutive scalars of the row
template <int L, int NV, typename T>
__device__ __forceinline__ void atomic_flush_dense_t(T* __restrict__ out, i64 row, const typename RowT<T>::vec (&acc)[NV], int l) {
  if constexpr (sizeof(T) == 4) {
    atomic_flush_dense<L, NV>(out, row, acc, l);
  } else {
    constexpr int N = RowT<T>::N;                 // 2 doubles per lane and piece
    constexpr i64 F = (i64)N * L * NV;
    T* base = out + row * F;
    const int comp = l & (N - 1);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int src = i * (L / N) + (l / N);
        const T x = __shfl(acc[v].x, src, L), y = __shfl(acc[v].y, src, L);
        unsafeAtomicAdd(base + (v * N + i) * L + l, comp == 0 ? x : y);
      }
    }
  }
}

}  // namespace graphop
