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

inline unsigned grid_of(i64 total) {
  i64 g = ceil_div(total > 0 ? total : 1, 256);
  return (unsigned)(g > 65536 ? 65536 : g);
}

}  // namespace
}  // namespace graphop

using namespace graphop;

extern "C" {

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
