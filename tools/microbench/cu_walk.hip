// Microbenchmark: can ALL XCDs walk the same column windows in step (accumulators stay on chip, no
// partial-row flushes) and still gather at the L2-resident rate?
//   one workgroup of 1024 threads per CU owns a fixed set of "rows" (accumulators in LDS) and, for
//   w = 0 .. W-1, gathers its share of random 256-B rows of window w (ids from an LCG, 16 rows in
//   flight per 16-lane group), adds every GRAN-slot partial sum to one of its LDS rows (ds_add_f32),
//   optionally paced per XCD (a workgroup may run at most `drift` windows ahead of the slowest one).
//   mode 0: every workgroup walks all W windows (the flush-free loop order)
//   mode 1: a workgroup only visits the windows its XCD owns (w % 8 == xcc): the shipped order,
//           same total work, partial sums still only to LDS (so the difference is the walk, not the flush)
// usage: cu_walk [W] [drift] [gran]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRows = 384;          // accumulator rows per workgroup (96 KB of LDS)
constexpr int kStride = 64;         // ints between counters

template <int MODE>
__global__ __launch_bounds__(1024) void k_walk(const float4* __restrict__ table, unsigned n_rows, int W,
                                               long slots_per_wg_window, int gran, int drift,
                                               int* __restrict__ sync, float* __restrict__ out) {
  extern __shared__ float acc_lds[];   // [kRows][64]
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;   // 64 groups
  for (int i = threadIdx.x; i < kRows * 64; i += 1024) acc_lds[i] = 0.f;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  int* reg = sync + xcc * kStride;
  int* ctr = sync + (8 + xcc * 512) * kStride;   // per (xcc, window) arrival counters
  if (threadIdx.x == 0 && drift > 0) atomicAdd(reg, 1);
  __syncthreads();
  const unsigned win_rows = n_rows / W;
  unsigned rnd = (blockIdx.x * 64u + g) * 2654435761u + 12345u;
  const long per_group = slots_per_wg_window / 64;
  for (int w = 0; w < W; ++w) {
    if (MODE == 1 && (w & 7) != (int)xcc) continue;
    if (drift > 0 && w >= drift) {      // wait until every workgroup of this XCD finished window w - drift
      if (threadIdx.x == 0) {
        const int n = __hip_atomic_load(reg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        while (__hip_atomic_load(ctr + (w - drift) * kStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
          __builtin_amdgcn_s_sleep(8);
          if (++it > 100000) break;
        }
      }
      __syncthreads();
    }
    const unsigned base = (unsigned)w * win_rows;
    const long n_slots = MODE == 1 ? per_group * 8 : per_group;   // same total work in both modes
    float4 a = make_float4(0, 0, 0, 0);
    int since = 0;
    for (long jb = 0; jb < n_slots; jb += 16) {
      rnd = rnd * 1664525u + 1013904223u;
      const unsigned my = base + ((rnd >> 4) + l * 40503u) % win_rows;
      float4 b[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned row = __shfl(my, u, 16);
        b[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + ((size_t)row * 256u + l * 16u));
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
      since += 16;
      if (since >= gran) {
        float* r = acc_lds + ((rnd >> 8) % kRows) * 64 + l * 4;
        atomicAdd(r + 0, a.x); atomicAdd(r + 1, a.y); atomicAdd(r + 2, a.z); atomicAdd(r + 3, a.w);
        a = make_float4(0, 0, 0, 0);
        since = 0;
      }
    }
    if (drift > 0) {
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr + w * kStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < kRows * 64; i += 1024) s += acc_lds[i];
  if (s == 1234.5f) out[0] = s;
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 16;
  const int drift = argc > 2 ? atoi(argv[2]) : 0;
  const int gran = argc > 3 ? atoi(argv[3]) : 32;
  const long E = 114615892;
  const unsigned n_rows = 232960;   // 59.6 MB table
  float4* table; int* sync; float* out;
  CK(hipMalloc(&table, (size_t)n_rows * 256)); CK(hipMalloc(&sync, sizeof(int) * kStride * (8 + 8 * 512))); CK(hipMalloc(&out, 4));
  CK(hipMemset(table, 0, (size_t)n_rows * 256));
  const int n_wg = 256;
  long per = E / ((long)n_wg * W);
  per = per / (64 * 16) * (64 * 16);
  const double slots = (double)per * n_wg * W;
  CK(hipFuncSetAttribute((const void*)k_walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kRows * 256));
  CK(hipFuncSetAttribute((const void*)k_walk<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kRows * 256));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode) {
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemsetAsync(sync, 0, sizeof(int) * kStride * (8 + 8 * 512)));
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL((k_walk<0>), dim3(n_wg), dim3(1024), kRows * 256, 0, table, n_rows, W, per, gran, drift, sync, out);
      else hipLaunchKernelGGL((k_walk<1>), dim3(n_wg), dim3(1024), kRows * 256, 0, table, n_rows, W, per, gran, mode == 1 ? 0 : drift, sync, out);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep > 0) { sum += ms; if (ms < best) best = ms; }
    }
    printf("W=%d drift=%d gran=%d  %s : mean %.3f ms  best %.3f ms  (%.1f TB/s of row gathers)\n", W, drift, gran,
           mode == 0 ? "all XCDs walk all windows" : "XCD-owned windows      ", sum / 5, best, slots * 256 / (sum / 5) / 1e9);
  }
  return 0;
}
