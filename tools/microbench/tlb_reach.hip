// Does the rate of a random 512-byte-row gather depend on the SIZE of the table once it is far beyond every cache?
// (If it does, address translation reach is part of the HBM random-row rate the sharded passes run at, and an order of
// the gathers with page locality would buy something; if not, 5.7-5.9 TB/s is the memory system's rate.)
// 64 M rows of 512 B gathered from tables of 0.5 ... 64 GB; ids = a hash of the slot number, masked to the table; eight
// rows in flight per 32-lane group, 5 workgroups per CU resident (the chunk driver's shape).
// Measured (profiles/r4_tlb_reach.txt): 7.14 TB/s at 0.5 GB (half of it Infinity-Cache hits), 6.22 at 4 GB, 6.04 at 8,
// 5.96 at 16, 5.91 at 32 and 64 GB -- flat within 2 % from 8 GB up: no translation cliff.
//   hipcc --offload-arch=gfx950 -O3 tlb_reach.hip -o tlb_reach && ./tlb_reach
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

__global__ __launch_bounds__(256, 5) void k_gather(const float4* __restrict__ tab, float* __restrict__ out,
                                                   long long n_slots, unsigned long long span_rows) {
  const int l = threadIdx.x % 32;
  const long long g = ((long long)blockIdx.x * 256 + threadIdx.x) / 32, ng = (long long)gridDim.x * 8;
  float acc = 0.f;
  for (long long j0 = g * 8; j0 < n_slots; j0 += ng * 8) {
    float4 x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = tab[(long long)(mix((unsigned long long)(j0 + u)) & (span_rows - 1)) * 32 + l];   // span: a power of two
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += x[u].x + x[u].y + x[u].z + x[u].w;
  }
  if (acc == 123.456f) out[g] = acc;
}

int main() {
  const long long max_bytes = 64LL << 30;
  float4* tab; CK(hipMalloc(&tab, max_bytes)); CK(hipMemset(tab, 0, max_bytes));
  float* out; CK(hipMalloc(&out, 1 << 24));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const long long n_slots = 64LL << 20;
  for (long long gb2 = 1; gb2 <= 128; gb2 *= 2) {
    const unsigned long long span_rows = (unsigned long long)gb2 * (1ULL << 29) / 512;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(k_gather, dim3(256 * 5 * 4), dim3(256), 0, 0, tab, out, n_slots, span_rows);
      CK(hipGetLastError());
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      best = ms < best ? ms : best;
    }
    printf("table %6.1f GB, uniformly random rows: %.3f ms = %.2f TB/s\n", span_rows * 512.0 / (1 << 30), best, n_slots * 512.0 / best / 1e9);
    fflush(stdout);
  }
  return 0;
}
