// Microbenchmark: what does the L2 -> CU path deliver for the sweep's access pattern?  Every
// 16-lane group gathers random 256-B rows (float4 per lane) from a window of `win_kb` KB, `SB` rows
// in flight per group, 4 groups per wave, `bpc` 256-thread workgroups per CU; no other traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int SB>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ table, unsigned n_rows, int iters,
                                                float* __restrict__ out, unsigned miss_per_1024 = 0,
                                                unsigned big_rows = 0) {
  const int l = threadIdx.x & 15;
  unsigned rnd = (blockIdx.x * 256u + (threadIdx.x >> 4)) * 2654435761u + 12345u;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 b[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      rnd = rnd * 1664525u + 1013904223u;
      unsigned r0 = (rnd >> 4) % n_rows;
      if (miss_per_1024 && ((rnd >> 20) & 1023u) < miss_per_1024) r0 = n_rows + (rnd >> 3) % big_rows;   // far row: L2 miss
      const unsigned row = __shfl((int)r0, 0, 16);
      b[u] = table[(size_t)row * 16 + l];
    }
#pragma unroll
    for (int u = 0; u < SB; ++u) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int bpc = argc > 1 ? atoi(argv[1]) : 4;
  float4* table; float* out;
  const size_t max_bytes = 512u << 20;
  CK(hipMalloc(&table, max_bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(table, 0, max_bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int blocks = 256 * bpc, iters = 200;
  for (int win_kb : {1024, 2048, 3700, 8192, 61440, 491520}) {
    const unsigned n_rows = (unsigned)((size_t)win_kb * 1024 / 256);
    for (int sb : {8, 16}) {
      auto launch = [&]() {
        if (sb == 8) hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, table, n_rows, iters * 2, out);
        else hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, table, n_rows, iters, out);
      };
      launch(); launch();
      CK(hipEventRecord(a));
      for (int r = 0; r < 5; ++r) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      const double bytes = (double)blocks * 16 * iters * 16 * 256.0;
      printf("window %7d KB  bpc %d  rows in flight/group %2d : %.3f ms  %.2f TB/s\n", win_kb, bpc, sb, ms, bytes / ms / 1e9);
    }
  }
  // mix: a fraction of the rows comes from a far (Infinity-Cache / HBM) region -> in-order return stalls
  for (int win_kb : {3700}) {
    const unsigned n_rows = (unsigned)((size_t)win_kb * 1024 / 256);
    const unsigned big_rows = (unsigned)((max_bytes - (size_t)win_kb * 1024) / 256);
    for (unsigned miss : {0u, 10u, 20u, 51u, 102u, 205u}) {
      auto launch = [&]() { hipLaunchKernelGGL(k_gather<16>, dim3(blocks), dim3(256), 0, 0, table, n_rows, iters, out, miss, big_rows); };
      launch(); launch();
      CK(hipEventRecord(a));
      for (int r = 0; r < 5; ++r) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      const double bytes = (double)blocks * 16 * iters * 16 * 256.0;
      printf("window %d KB + %.1f %% far rows (HBM) : %.3f ms  %.2f TB/s\n", win_kb, miss / 10.24, ms, bytes / ms / 1e9);
    }
  }
  return 0;
}
