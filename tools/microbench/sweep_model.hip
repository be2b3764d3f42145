// Microbenchmark: the SDDMM sweep's inner loop with every row request hitting L2 (one 3.7 MB
// window for everybody, no pacing, no window steps), features switched on one by one:
//   mode 0: LCG row ids, sum only            (= l2_gather)
//   mode 1: + coalesced nontemporal id stream (E ints) instead of the LCG
//   mode 2: + per-edge ds_read_b128 of an "A row" from LDS and a dot product with DPP reduction
//   mode 3: + nontemporal store of one float per edge
// E = 114,615,892 edges, 1024 workgroups x 256 threads, 16 rows in flight per 16-lane group.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
  v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v); return v;
}

template <int MODE, int PF>
__global__ __launch_bounds__(256, 4) void k_model(const float4* __restrict__ table, unsigned n_rows,
                                                  const int* __restrict__ ids, long n_edges,
                                                  float* __restrict__ y, float* __restrict__ out) {
  __shared__ float4 rows[16 * 8 * 16];   // 16 groups x 8 "A rows" x 16 lanes
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (int i = threadIdx.x; i < 16 * 8 * 16; i += 256) rows[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const long n_groups = (long)gridDim.x * 16;
  const long gid = (long)blockIdx.x * 16 + g;
  const long per = (n_edges + n_groups - 1) / n_groups;
  const long e0 = gid * per, e1 = e0 + per < n_edges ? e0 + per : n_edges;
  unsigned rnd = (unsigned)gid * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  float res = 0.f;
  int q[PF > 0 ? PF : 1];
#pragma unroll
  for (int p = 0; p < PF; ++p) q[p] = (e0 + p * 16 + l < e1) ? __builtin_nontemporal_load(ids + e0 + p * 16 + l) : 0;
  for (long jb = e0; jb < e1; jb += 16) {
    int my = 0;
    if (PF > 0) {
      my = q[0];
#pragma unroll
      for (int p = 0; p + 1 < PF; ++p) q[p] = q[p + 1];
    } else if (MODE >= 1) { if (jb + l < e1) my = __builtin_nontemporal_load(ids + jb + l); }
    else { rnd = rnd * 1664525u + 1013904223u; my = (int)((rnd >> 4) % n_rows); }
    float4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int row = __shfl(my, u, 16);
      b[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + ((unsigned)row * 256u + l * 16u));
    }
    if (PF > 0) q[PF - 1] = (jb + PF * 16 + l < e1) ? __builtin_nontemporal_load(ids + jb + PF * 16 + l) : 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE >= 2) {
        const float4 a = rows[(g * 8 + (u & 7)) * 16 + l];
        float p = a.x * b[u].x + a.y * b[u].y + a.z * b[u].z + a.w * b[u].w;
        p = sum16(p);
        if (l == u) res = p;
      } else { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
    }
    if (MODE >= 3) { if (jb + l < e1) __builtin_nontemporal_store(res, y + jb + l); }
  }
  if (acc.x + acc.y + acc.z + acc.w + res == 1234.5f) out[0] = acc.x;
}

// Staged ids: a group fetches the ids of a whole task (TB batches) with dwordx4 loads one task
// ahead, parks them in LDS at the task switch and reads one id per lane and batch with ds_read --
// between two batches of row requests the vector memory pipeline sees no other load.
template <int MODE, int TB>
__global__ __launch_bounds__(256, 4) void k_model_staged(const float4* __restrict__ table, unsigned n_rows,
                                                         const int* __restrict__ ids, long n_edges,
                                                         float* __restrict__ y, float* __restrict__ out) {
  __shared__ float4 rows[16 * 4 * 16];
  __shared__ int idbuf[16][TB * 16];
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  for (int i = threadIdx.x; i < 16 * 4 * 16; i += 256) rows[i] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  const long n_groups = (long)gridDim.x * 16;
  const long gid = (long)blockIdx.x * 16 + g;
  long per = (n_edges + n_groups - 1) / n_groups;
  per = (per + TB * 16 - 1) / (TB * 16) * (TB * 16);
  const long e0 = gid * per, e1 = e0 + per < n_edges ? e0 + per : n_edges;
  float res = 0.f;
  float4 acc = make_float4(0, 0, 0, 0);
  constexpr int NQ = TB / 4;
  int4 nx[NQ];
  auto fetch = [&](long base) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const long at = base + k * 64 + l * 4;
      nx[k] = (at + 3 < e1) ? *reinterpret_cast<const int4*>(ids + at) : make_int4(0, 0, 0, 0);
    }
  };
  if (e0 < e1) fetch(e0);
  for (long base = e0; base < e1; base += TB * 16) {
#pragma unroll
    for (int k = 0; k < NQ; ++k) *reinterpret_cast<int4*>(&idbuf[g][k * 64 + l * 4]) = nx[k];
    fetch(base + TB * 16);
    for (int bt = 0; bt < TB; ++bt) {
      const long jb = base + bt * 16;
      if (jb >= e1) break;
      const int my = idbuf[g][bt * 16 + l];
      float4 b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int row = __shfl(my, u, 16);
        b[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + ((unsigned)row * 256u + l * 16u));
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (MODE >= 2) {
          const float4 a = rows[(g * 4 + (u & 3)) * 16 + l];
          float p = a.x * b[u].x + a.y * b[u].y + a.z * b[u].z + a.w * b[u].w;
          p = sum16(p);
          if (l == u) res = p;
        } else { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
      }
      if (MODE >= 3) { if (jb + l < e1) __builtin_nontemporal_store(res, y + jb + l); }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w + res == 1234.5f) out[0] = acc.x;
}

int main() {
  const long E = 114615892;
  const unsigned n_rows = 3700u * 1024 / 256;
  float4* table; int* ids; float *y, *out;
  CK(hipMalloc(&table, (size_t)n_rows * 256)); CK(hipMalloc(&ids, E * 4)); CK(hipMalloc(&y, E * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(table, 0, (size_t)n_rows * 256));
  std::vector<int> h(E);
  unsigned r = 1;
  for (long i = 0; i < E; ++i) { r = r * 1664525u + 1013904223u; h[i] = (int)((r >> 4) % n_rows); }
  CK(hipMemcpy(ids, h.data(), E * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 14; ++mode) {
    auto launch = [&]() {
      switch (mode) {
        case 0: hipLaunchKernelGGL((k_model<0, 0>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 1: hipLaunchKernelGGL((k_model<1, 0>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 2: hipLaunchKernelGGL((k_model<2, 0>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 3: hipLaunchKernelGGL((k_model<3, 0>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 4: hipLaunchKernelGGL((k_model<3, 1>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 5: hipLaunchKernelGGL((k_model<3, 2>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 6: hipLaunchKernelGGL((k_model<3, 3>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 7: hipLaunchKernelGGL((k_model<3, 4>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 8: hipLaunchKernelGGL((k_model_staged<1, 8>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 9: hipLaunchKernelGGL((k_model_staged<1, 16>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 10: hipLaunchKernelGGL((k_model_staged<2, 16>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 11: hipLaunchKernelGGL((k_model_staged<3, 8>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        case 12: hipLaunchKernelGGL((k_model_staged<3, 16>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
        default: hipLaunchKernelGGL((k_model_staged<3, 32>), dim3(1024), dim3(256), 0, 0, table, n_rows, ids, E, y, out); break;
      }
    };
    launch(); launch();
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    static const char* names[14] = {"", "", "", "", " + ids 1 batch ahead", " + ids 2 ahead", " + ids 3 ahead", " + ids 4 ahead",
                                    " ids staged in LDS per 8 batches (mode 1 work)", " ids staged per 16 batches (mode 1 work)",
                                    " ids staged per 16 batches (mode 2 work)", " ids staged per 8 batches", " ids staged per 16 batches",
                                    " ids staged per 32 batches"};
    printf("mode %d%s : %.3f ms  (row gather %.1f TB/s)\n", mode < 4 ? mode : (mode < 8 || mode > 10 ? 3 : (mode == 10 ? 2 : 1)), names[mode],
           ms, (double)E * 256 / ms / 1e9);
  }
  return 0;
}
