// Can the scalar data cache carry the per-slot scalar gather of the column-major passes?
// Four "feeder-like" waves per CU (optionally more) gather 4-byte scalars w[idx[j]] whose addresses fall into a
// sliding 16 MB region of a 458 MB array (what one window step of the column-major walk touches), (a) with one
// vector load per 64 slots (64 scattered dwords = 64 L1 tag lookups), (b) with 64 scalar loads (v_readlane ->
// s_load_dword; the scalar cache is a separate L2 client, the vector memory pipeline never sees them).
//   hipcc --offload-arch=gfx950 -O3 smem_gather.hip -o smem_gather && ./smem_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0 plain vector load, 1 nontemporal, 2 agent-scope relaxed atomic load (sc1), 3 scalar cache
__global__ __launch_bounds__(1024) void k_gather(const float* __restrict__ w, const int* __restrict__ idx, float* __restrict__ out,
                                                 long long per_wave) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int* my = idx + wave * per_wave;
  float acc = 0.f;
  for (long long j = 0; j < per_wave; j += 64) {
    const int e = my[j + lane];
    if constexpr (MODE == 3) {
      float v = 0.f;
#pragma unroll
      for (int i0 = 0; i0 < 64; i0 += 32) {         // 32 scalar loads in flight per wave, one wait
        float x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const int off = __builtin_amdgcn_readlane(e, i0 + u) << 2;
          asm volatile("s_load_dword %0, %1, %2" : "=s"(x[u]) : "s"(w), "s"(off));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          asm volatile("" : "+s"(x[u]));             // (consumed after the wait)
          v = lane == i0 + u ? x[u] : v;
        }
      }
      acc += v;
    } else if constexpr (MODE == 1) {
      acc += __builtin_nontemporal_load(w + e);
    } else if constexpr (MODE == 2) {
      acc += __hip_atomic_load(w + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      acc += w[e];
    }
  }
  out[wave * 64 + lane] = acc;
}

int main() {
  const long long E = 114615892;
  const int cus = 256;
  float* w; CK(hipMalloc(&w, E * 4)); CK(hipMemset(w, 0, E * 4));
  for (int waves_per_cu : {4, 8, 16}) {
    const long long waves = (long long)cus * waves_per_cu;
    const long long per_wave = ((E / waves) / 64) * 64;
    std::vector<int> h((size_t)(waves * per_wave));
    // wave q, position j: all waves sweep the array together; at a given share of the run the addresses lie within a
    // 16 MB (4 M floats) region that slides from the start of the array to its end
    unsigned long long s = 88172645463325252ULL;
    for (long long q = 0; q < waves; ++q)
      for (long long j = 0; j < per_wave; ++j) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const long long base = (long long)((double)j / per_wave * (E - 4200000));
        h[(size_t)(q * per_wave + j)] = (int)(base + (long long)(s % 4000000ULL));
      }
    int* idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float* out; CK(hipMalloc(&out, waves * 64 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        const dim3 g(cus), t(64 * waves_per_cu);
        if (mode == 0) hipLaunchKernelGGL(k_gather<0>, g, t, 0, 0, w, idx, out, per_wave);
        else if (mode == 1) hipLaunchKernelGGL(k_gather<1>, g, t, 0, 0, w, idx, out, per_wave);
        else if (mode == 2) hipLaunchKernelGGL(k_gather<2>, g, t, 0, 0, w, idx, out, per_wave);
        else hipLaunchKernelGGL(k_gather<3>, g, t, 0, 0, w, idx, out, per_wave);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      const char* names[4] = {"vector plain ", "vector nt    ", "vector sc1   ", "scalar cache "};
      printf("waves/CU %2d  %s gather of %lld scalars: %.3f ms  (%.1f G scalars/s)\n", waves_per_cu, names[mode],
             waves * per_wave, best, waves * per_wave / best / 1e6);
    }
    CK(hipFree(idx)); CK(hipFree(out));
  }
  return 0;
}
