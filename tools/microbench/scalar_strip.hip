// Microbenchmark: "wave per row, scalar control" inner loops for d = 64 (one float per lane).
// A wave walks a contiguous run of slots; the neighbour ids come in through SCALAR loads
// (s_load_dwordx16: their own cache, their own counter -- nothing sits in the in-order vector-memory
// return queue between two batches of row requests), every row is ONE buffer_load_dword whose
// row offset is a SCALAR operand (soffset = id << 8; 64 lanes x 4 B = the 256-B row): no VALU
// address work and no LDS at all.
//   mode 0: SDDMM flavour  y[e] = <a, B[id[e]]>: 1 multiply per slot, 16 slots reduced together with
//           v_permlane32_swap / v_permlane16_swap + DPP (2.2 VALU per slot), 16 results stored coalesced
//   mode 1: SpMM flavour   acc += w[e] * X[id[e]]: the weight is a scalar operand of the FMA
//           (1 VALU per slot), the 256-B sum is flushed every `flush` slots with float atomics
//   mode 2: as 1 but the weights are gathered through a second scalar id stream (w[eid[e]], the
//           column-major passes' transposed scalar gather), random 4-B scalar loads
// Same table / id set as l2_gather_ids.hip (3.7 MB table: L2-resident in every XCD), so the rates
// compare directly with that file's modes (lane-group strips: 22 TB/s with per-batch id loads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swap32_add(float x, float y) {   // lanes 0-31: x.lo + x.hi ; 32-63: y.lo + y.hi
  v2i r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, x), __builtin_bit_cast(int, y), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float swap16_add(float x, float y) {
  v2i r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, x), __builtin_bit_cast(int, y), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}

// p[u] = lane's product of slot u; returns slot (lane & 15)'s... see mapping below: after the
// two swaps lane l holds partial sums of slot ((l>>5)*8 + ((l>>4)&1)*4 + u') for u' < 4; the DPP
// steps then select by lane bits 3, 2 and finish with two full steps: lane l ends with the sum of
// slot  8*(l>>5) + 4*((l>>4)&1) + 2*((l>>3)&1) + ((l>>2)&1).
__device__ __forceinline__ float reduce16(float (&p)[16], int lane) {
  float t8[8], t4[4], t2[2];
#pragma unroll
  for (int u = 0; u < 8; ++u) t8[u] = swap32_add(p[u], p[u + 8]);
#pragma unroll
  for (int u = 0; u < 4; ++u) t4[u] = swap16_add(t8[u], t8[u + 4]);
  const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const float keep = b3 ? t4[u + 2] : t4[u], send = b3 ? t4[u] : t4[u + 2];
    t2[u] = keep + dpp<0x128>(send);
  }
  const float keep = b2 ? t2[1] : t2[0], send = b2 ? t2[0] : t2[1];
  float r = keep + dpp<0x141>(send);      // row_half_mirror: flips bits 0-2 (bit 2 is what we need)
  r += dpp<0x4E>(r);                      // xor 2
  r += dpp<0xB1>(r);                      // xor 1
  return r;
}

template <int MODE, int BPC>
__global__ __launch_bounds__(256, BPC) void k(const float* __restrict__ table, const int* __restrict__ ids,
                                              const int* __restrict__ eids, const float* __restrict__ w,
                                              long per_wave, int flush, float* __restrict__ y,
                                              float* __restrict__ out, unsigned n_rows) {
  const int lane = threadIdx.x & 63;
  const long wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const long e0 = wave * per_wave, e1 = e0 + per_wave;
  const unsigned voff = (unsigned)lane * 4u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, n_rows * 256u, 0x00020000);
  const float a = 0.5f + lane;   // the wave's own row (one float per lane)
  float acc = 0.f;
  v16i nid = *reinterpret_cast<const v16i*>(ids + e0);
  v16f nw; v16i neid;
  if (MODE == 1) nw = *reinterpret_cast<const v16f*>(w + e0);
  if (MODE == 2) neid = *reinterpret_cast<const v16i*>(eids + e0);
  int since = 0;
  for (long jb = e0; jb < e1; jb += 16) {
    const v16i id = nid;
    float b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)   // buffer_load_dword v, v_lane4, s[rsrc], s_rowoff offen: no VALU per slot
      b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, id[u] << 8, 0));
    v16f wt;
    if (MODE == 1) wt = nw;
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u) wt[u] = w[neid[u]];     // scalar loads, random 4 B
    }
    const long jn = jb + 16 < e1 ? jb + 16 : e0;
    nid = *reinterpret_cast<const v16i*>(ids + jn);
    if (MODE == 1) nw = *reinterpret_cast<const v16f*>(w + jn);
    if (MODE == 2) neid = *reinterpret_cast<const v16i*>(eids + jn);
    if (MODE == 0) {
      float p[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) p[u] = a * b[u];
      const float r = reduce16(p, lane);
      // lanes with (lane & 3) == 0 hold one slot each: 16 results; store them (slot order permuted the
      // same way in every batch -- a real kernel permutes the 16 ids' order instead)
      if ((lane & 3) == 0) y[jb + (lane >> 2)] = r;
    } else {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(wt[u], b[u], acc);
      since += 16;
      if (since >= flush) {
        atomicAdd(out + ((size_t)((unsigned)id[0] % n_rows) << 6) + lane, acc);
        acc = 0.f; since = 0;
      }
    }
  }
  if (MODE != 0 && acc == 1234.5f) out[0] = acc;
}

int main(int argc, char** argv) {
  const unsigned n_rows = 3700u * 1024 / 256;
  const int flush = argc > 1 ? atoi(argv[1]) : 32;
  const long waves = 1024L * 4 * 2, per_wave = 14336, E = waves * per_wave;   // 117.4 M slots
  float *table, *out, *w, *y; int *ids, *eids;
  CK(hipMalloc(&table, (size_t)n_rows * 256)); CK(hipMemset(table, 0, (size_t)n_rows * 256));
  CK(hipMalloc(&out, (size_t)n_rows * 256)); CK(hipMemset(out, 0, (size_t)n_rows * 256));
  CK(hipMalloc(&ids, (size_t)(E + 1024) * 4)); CK(hipMalloc(&eids, (size_t)(E + 1024) * 4));
  CK(hipMalloc(&w, (size_t)(E + 1024) * 4)); CK(hipMemset(w, 0, (size_t)(E + 1024) * 4));
  CK(hipMalloc(&y, (size_t)(E + 1024) * 4));
  int* h = (int*)malloc((size_t)E * 4);
  unsigned s = 12345u;
  for (long i = 0; i < E; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 8) % n_rows); }
  CK(hipMemcpy(ids, h, (size_t)E * 4, hipMemcpyHostToDevice));
  for (long i = 0; i < E; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)(((unsigned long long)(s >> 4) * (unsigned long long)E) >> 28); }
  CK(hipMemcpy(eids, h, (size_t)E * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int bpc = 4; bpc <= 8; bpc += 4)
    for (int mode = 0; mode < 3; ++mode) {
      const long pw = bpc == 8 ? per_wave : per_wave * 2;
      const unsigned grid = bpc == 8 ? 2048 : 1024;
      auto launch = [&]() {
#define GO(M, B) hipLaunchKernelGGL((k<M, B>), dim3(grid), dim3(256), 0, 0, table, ids, eids, w, pw, flush, y, out, n_rows)
        if (bpc == 4) { if (mode == 0) GO(0, 4); else if (mode == 1) GO(1, 4); else GO(2, 4); }
        else { if (mode == 0) GO(0, 8); else if (mode == 1) GO(1, 8); else GO(2, 8); }
#undef GO
      };
      launch(); launch();
      CK(hipEventRecord(a));
      for (int r = 0; r < 5; ++r) launch();
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
      printf("waves/SIMD %d mode %d : %.3f ms  (%.1f TB/s of row gathers)\n", bpc, mode, ms, (double)E * 256 / ms / 1e9);
    }
  return 0;
}
