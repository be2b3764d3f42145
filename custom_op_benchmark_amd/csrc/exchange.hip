// Halo pack / unpack kernels of the node-range sharded step (custom_op_benchmark_amd/dist.py).
// Not in the reference (single GPU): north_star's multi-GPU path sends the node rows
// other ranks gather from (K, V) and returns partial gradient rows (dK, dV) by all-to-all; these two
// kernels move rows between the node tensors and the contiguous send / receive buffers.
#include "common.h"
#include "host.h"

namespace graphop {
namespace {

// dst[i, :] = src[idx[i], :]; rows of `row16` 16-byte units, one lane per unit (coalesced both ways)
__global__ __launch_bounds__(256) void k_gather_rows16(const uint4* __restrict__ src,
                                                       const i64* __restrict__ idx, uint4* __restrict__ dst,
                                                       i64 n_idx, i64 row16) {
  const i64 total = n_idx * row16;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 i = t / row16, c = t - i * row16;
    dst[t] = src[idx[i] * row16 + c];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_gather_rows(const T* __restrict__ src, const i64* __restrict__ idx,
                                                     T* __restrict__ dst, i64 n_idx, i64 row_elems) {
  const i64 total = n_idx * row_elems;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 i = t / row_elems, c = t - i * row_elems;
    dst[t] = src[idx[i] * row_elems + c];
  }
}
// dst[idx[i], :] += src[i, :]; idx may repeat (a row served to several peers): native float atomics,
// consecutive lanes on consecutive dwords of a row (whole 64-B memory-side requests)
template <typename T>
__global__ __launch_bounds__(256) void k_scatter_add_rows(const T* __restrict__ src, const i64* __restrict__ idx,
                                                          T* __restrict__ dst, i64 n_idx, i64 row_elems) {
  const i64 total = n_idx * row_elems;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 i = t / row_elems, c = t - i * row_elems;
    atomicAdd(dst + idx[i] * row_elems + c, src[t]);
  }
}

// the same for a run of idx WITHOUT repeats (the rows one peer was served): plain read-add-write
template <typename T>
__global__ __launch_bounds__(256) void k_add_rows_unique(const T* __restrict__ src, const i64* __restrict__ idx,
                                                         T* __restrict__ dst, i64 n_idx, i64 row_elems) {
  const i64 total = n_idx * row_elems;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 i = t / row_elems, c = t - i * row_elems;
    T* p = dst + idx[i] * row_elems + c;
    *p = *p + src[t];
  }
}

// dst[rows[g], :] += sum over p in [ptr[g], ptr[g + 1]) of src[pos[p], :] -- the received rows grouped by the own row
// they belong to (grouping built once per shard, dist.py): ONE launch for all peers instead of one plain-add launch
// per peer, every own row read and written once however many peers it was served to, no atomics, and the
// summation order (peer order) is fixed.  16-byte units, one lane per unit.
__global__ __launch_bounds__(256) void k_add_rows_grouped16(const float4* __restrict__ src, const i64* __restrict__ ptr,
                                                            const i64* __restrict__ rows, const i64* __restrict__ pos,
                                                            float4* __restrict__ dst, i64 n_groups, i64 row16) {
  const i64 total = n_groups * row16;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 g = t / row16, c = t - g * row16;
    float4* d = dst + rows[g] * row16 + c;
    float4 acc = *d;
    for (i64 p = ptr[g]; p < ptr[g + 1]; ++p) {
      const float4 v = src[pos[p] * row16 + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *d = acc;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_add_rows_grouped(const T* __restrict__ src, const i64* __restrict__ ptr,
                                                          const i64* __restrict__ rows, const i64* __restrict__ pos,
                                                          T* __restrict__ dst, i64 n_groups, i64 row_elems) {
  const i64 total = n_groups * row_elems;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (i64)gridDim.x * blockDim.x) {
    const i64 g = t / row_elems, c = t - g * row_elems;
    T* d = dst + rows[g] * row_elems + c;
    T acc = *d;
    for (i64 p = ptr[g]; p < ptr[g + 1]; ++p) acc += src[pos[p] * row_elems + c];
    *d = acc;
  }
}

inline unsigned grid_of(i64 total) {
  i64 g = ceil_div(total > 0 ? total : 1, 256);
  return (unsigned)(g > 65536 ? 65536 : g);
}

// out2[i] = (w0[i], w1[i]): two per-edge arrays interleaved into the (n, 2) pairs graphop_spmm_pair reads (streaming;
// torch.stack does the same at a third of the rate)
__global__ __launch_bounds__(256) void k_interleave_pairs(const float4* __restrict__ w0, const float4* __restrict__ w1,
                                                          float4* __restrict__ out2, i64 n4, const float* __restrict__ t0,
                                                          const float* __restrict__ t1, float2* __restrict__ tout, int n_tail) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (i64)gridDim.x * blockDim.x) {
    const float4 a = w0[i], b = w1[i];
    out2[2 * i] = make_float4(a.x, b.x, a.y, b.y);
    out2[2 * i + 1] = make_float4(a.z, b.z, a.w, b.w);
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tout[threadIdx.x] = make_float2(t0[threadIdx.x], t1[threadIdx.x]);
}

}  // namespace
}  // namespace graphop

using namespace graphop;

extern "C" {

int graphop_interleave_pairs(int dtype, const void* w0, const void* w1, void* out2, int64_t n, void* stream) {
  const char* fn = "interleave_pairs";
  GO_CHECK_ARG(dtype == GRAPHOP_F32, "%s: fp32 only", fn);
  GO_CHECK_ARG(n >= 0, "%s: negative size", fn);
  if (n == 0) return GRAPHOP_OK;
  GO_PTR(fn, w0); GO_PTR(fn, w1); GO_PTR(fn, out2);
  GO_CHECK_ARG((((uintptr_t)w0 | (uintptr_t)w1 | (uintptr_t)out2) & 15) == 0, "%s: 16-byte-aligned arrays", fn);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("interleave_pairs", st, "k_interleave_pairs");
  const i64 n4 = n / 4;
  const int tail = (int)(n - 4 * n4);
  hipLaunchKernelGGL(k_interleave_pairs, dim3(grid_of(n4 > 0 ? n4 : 1)), dim3(256), 0, st, (const float4*)w0, (const float4*)w1,
                     (float4*)out2, n4, (const float*)w0 + 4 * n4, (const float*)w1 + 4 * n4, (float2*)out2 + 4 * n4, tail);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

int graphop_gather_rows(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                        int64_t n_src_rows, int64_t row_elems, void* stream) {
  const char* fn = "gather_rows";
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_idx >= 0 && n_src_rows >= 0 && row_elems >= 0, "%s: negative size", fn);
  if (n_idx * row_elems == 0) return GRAPHOP_OK;
  GO_PTR(fn, src); GO_PTR(fn, idx); GO_PTR(fn, dst);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("halo_pack", st, "k_gather_rows16");
  const size_t row_bytes = esize(dtype) * (size_t)row_elems;
  if (row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    const i64 row16 = (i64)(row_bytes / 16);
    hipLaunchKernelGGL(k_gather_rows16, dim3(grid_of(n_idx * row16)), dim3(256), 0, st, (const uint4*)src,
                       (const i64*)idx, (uint4*)dst, n_idx, row16);
  } else if (dtype == GRAPHOP_F32) {
    hipLaunchKernelGGL((k_gather_rows<float>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const float*)src, (const i64*)idx, (float*)dst, n_idx, row_elems);
  } else {
    hipLaunchKernelGGL((k_gather_rows<double>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const double*)src, (const i64*)idx, (double*)dst, n_idx, row_elems);
  }
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

int graphop_add_rows_unique(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                            int64_t n_dst_rows, int64_t row_elems, void* stream) {
  const char* fn = "add_rows_unique";
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_idx >= 0 && n_dst_rows >= 0 && row_elems >= 0, "%s: negative size", fn);
  if (n_idx * row_elems == 0) return GRAPHOP_OK;
  GO_PTR(fn, src); GO_PTR(fn, idx); GO_PTR(fn, dst);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("halo_unpack_add", st, "k_add_rows_unique");
  if (dtype == GRAPHOP_F32)
    hipLaunchKernelGGL((k_add_rows_unique<float>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const float*)src, (const i64*)idx, (float*)dst, n_idx, row_elems);
  else
    hipLaunchKernelGGL((k_add_rows_unique<double>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const double*)src, (const i64*)idx, (double*)dst, n_idx, row_elems);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

int graphop_add_rows_grouped(int dtype, const void* src, const int64_t* grp_ptr, const int64_t* grp_rows,
                             const int64_t* grp_pos, void* dst, int64_t n_groups, int64_t n_dst_rows,
                             int64_t row_elems, void* stream) {
  const char* fn = "add_rows_grouped";
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_groups >= 0 && n_dst_rows >= 0 && row_elems >= 0, "%s: negative size", fn);
  if (n_groups * row_elems == 0) return GRAPHOP_OK;
  GO_PTR(fn, src); GO_PTR(fn, grp_ptr); GO_PTR(fn, grp_rows); GO_PTR(fn, grp_pos); GO_PTR(fn, dst);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("halo_unpack_add", st, "k_add_rows_grouped");
  const size_t row_bytes = esize(dtype) * (size_t)row_elems;
  if (dtype == GRAPHOP_F32 && row_bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    const i64 row16 = (i64)(row_bytes / 16);
    hipLaunchKernelGGL(k_add_rows_grouped16, dim3(grid_of(n_groups * row16)), dim3(256), 0, st, (const float4*)src,
                       (const i64*)grp_ptr, (const i64*)grp_rows, (const i64*)grp_pos, (float4*)dst, n_groups, row16);
  } else if (dtype == GRAPHOP_F32) {
    hipLaunchKernelGGL((k_add_rows_grouped<float>), dim3(grid_of(n_groups * row_elems)), dim3(256), 0, st, (const float*)src,
                       (const i64*)grp_ptr, (const i64*)grp_rows, (const i64*)grp_pos, (float*)dst, n_groups, row_elems);
  } else {
    hipLaunchKernelGGL((k_add_rows_grouped<double>), dim3(grid_of(n_groups * row_elems)), dim3(256), 0, st, (const double*)src,
                       (const i64*)grp_ptr, (const i64*)grp_rows, (const i64*)grp_pos, (double*)dst, n_groups, row_elems);
  }
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

int graphop_scatter_add_rows(int dtype, const void* src, const int64_t* idx, void* dst, int64_t n_idx,
                             int64_t n_dst_rows, int64_t row_elems, void* stream) {
  const char* fn = "scatter_add_rows";
  GO_CHECK_ARG(dtype == GRAPHOP_F32 || dtype == GRAPHOP_F64, "%s: bad dtype", fn);
  GO_CHECK_ARG(n_idx >= 0 && n_dst_rows >= 0 && row_elems >= 0, "%s: negative size", fn);
  if (n_idx * row_elems == 0) return GRAPHOP_OK;
  GO_PTR(fn, src); GO_PTR(fn, idx); GO_PTR(fn, dst);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof("halo_unpack_add", st, "k_scatter_add_rows");
  if (dtype == GRAPHOP_F32)
    hipLaunchKernelGGL((k_scatter_add_rows<float>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const float*)src, (const i64*)idx, (float*)dst, n_idx, row_elems);
  else
    hipLaunchKernelGGL((k_scatter_add_rows<double>), dim3(grid_of(n_idx * row_elems)), dim3(256), 0, st,
                       (const double*)src, (const i64*)idx, (double*)dst, n_idx, row_elems);
  GO_LAUNCH_CHECK();
  return GRAPHOP_OK;
}

}  // extern "C"
