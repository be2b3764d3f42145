// Microbenchmark: does it matter WHERE the gathered row ids come from?  Same loop as l2_gather.hip
// (16-lane groups, 16 random 256-B rows of a 3.7 MB window in flight per group), ids either
//   mode 0: computed in registers (hash of a counter),
//   mode 1: one coalesced 64-B load per batch (lane u holds the id of slot u), consumed right away,
//   mode 2: as 1, but the load for batch i+1 is issued before batch i's rows are consumed,
//   mode 3: 16 batches of ids fetched at once (one dwordx4 x 4 per lane = 256 ids per group),
//   mode 4: as 2, but four batches ahead (four registers),
//   mode 5: 4 batches of ids fetched at once (one dwordx4 per lane = 64 ids per group), one block ahead,
//   mode 6: every 4 batches, four dword loads back to back (ids of the NEXT four batches),
//   mode 7: every 16 batches, sixteen dword loads back to back (ids of the next sixteen batches).
// `./l2_gather_ids 1` wraps the id indices inside 1 MB, i.e. an L2-RESIDENT id stream: modes 1 and 2
// then run at 29.6 / 30.5 TB/s instead of 21.5 / 23.0 with the ids streamed from HBM -- vector-memory
// loads return in issue order, so an id load that misses L2 holds back the row loads behind it
// (round 2; what kernels_fast.h: LineTouch acts on).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const float4* __restrict__ table, unsigned n_rows, const int* __restrict__ ids_,
                                            long per_group, float* __restrict__ out, long id_mask) {
  struct Wrap { const int* p; long m; __device__ const int& operator[](long i) const { return p[i & m]; } __device__ const int* operator+(long i) const { return p + (i & m); } } ids{ids_, id_mask};
  const int l = threadIdx.x & 15;
  const long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const long e0 = g * per_group, e1 = e0 + per_group;
  float4 acc = make_float4(0, 0, 0, 0);
  int nxt = 0;
  if (MODE == 2) nxt = ids[e0 + l];
  int4 blk[4];
  int q4[4] = {0, 0, 0, 0};
  if (MODE == 4) { for (int p = 0; p < 4; ++p) q4[p] = ids[e0 + p * 16 + l]; }
  int4 cur4 = make_int4(0, 0, 0, 0), nxt4 = make_int4(0, 0, 0, 0);
  if (MODE == 5) nxt4 = *reinterpret_cast<const int4*>(ids + e0 + l * 4);
  int c6[4] = {0, 0, 0, 0}, n6[4] = {0, 0, 0, 0};
  if (MODE == 6) { for (int p = 0; p < 4; ++p) n6[p] = ids[e0 + p * 16 + l]; }
  int c7[16], n7[16];
  if (MODE == 7) { for (int p = 0; p < 16; ++p) { n7[p] = ids[e0 + p * 16 + l]; c7[p] = 0; } }
  for (long jb = e0; jb < e1; jb += 16) {
    int my;
    if (MODE == 0) my = (int)(hash32((unsigned)(jb + l)) % n_rows);
    else if (MODE == 1) my = ids[jb + l];
    else if (MODE == 2) my = nxt;
    else if (MODE == 3) {
      const int b16 = (int)((jb - e0) >> 4) & 15;          // batch inside the 256-id block
      if (b16 == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) blk[q] = *reinterpret_cast<const int4*>(ids + jb + (l * 4 + q) * 4);   // lane l holds ids [16l, 16l+16)
      }
      // slot u of batch b16 = id index b16*16 + u = lane b16, element u
      my = 0;   // filled below through shuffles
      int tmp[16] = {blk[0].x, blk[0].y, blk[0].z, blk[0].w, blk[1].x, blk[1].y, blk[1].z, blk[1].w,
                     blk[2].x, blk[2].y, blk[2].z, blk[2].w, blk[3].x, blk[3].y, blk[3].z, blk[3].w};
      int mine = 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int v = __shfl(tmp[u], b16, 16); if (l == u) mine = v; }
      my = mine;
    } else if (MODE == 4) {
      my = q4[0]; q4[0] = q4[1]; q4[1] = q4[2]; q4[2] = q4[3];
    } else if (MODE == 6) {
      const int b4 = (int)((jb - e0) >> 4) & 3;
      if (b4 == 0) { for (int p = 0; p < 4; ++p) c6[p] = n6[p]; }
      my = b4 == 0 ? c6[0] : (b4 == 1 ? c6[1] : (b4 == 2 ? c6[2] : c6[3]));
    } else if (MODE == 7) {
      const int b16 = (int)((jb - e0) >> 4) & 15;
      if (b16 == 0) { for (int p = 0; p < 16; ++p) c7[p] = n7[p]; }
      my = c7[0];
#pragma unroll
      for (int p = 1; p < 16; ++p) my = b16 == p ? c7[p] : my;
    } else {   // MODE 5: lane l holds ids [4l, 4l+4) of the 64-id block; slot u of batch b4 = index b4*16+u = lane (b4*4 + u/4), element u%4
      const int b4 = (int)((jb - e0) >> 4) & 3;
      if (b4 == 0) { cur4 = nxt4; }
      const int src = b4 * 4 + (l >> 2);
      const int x = __shfl(cur4.x, src, 16), y = __shfl(cur4.y, src, 16), z = __shfl(cur4.z, src, 16), w = __shfl(cur4.w, src, 16);
      const int c = l & 3;
      my = c == 0 ? x : (c == 1 ? y : (c == 2 ? z : w));
    }
    float4 b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned row = (unsigned)__shfl(my, u, 16);
      b[u] = table[(size_t)row * 16 + l];
    }
    if (MODE == 2) nxt = (jb + 16 < e1) ? ids[jb + 16 + l] : 0;
    if (MODE == 4) q4[3] = (jb + 64 < e1) ? ids[jb + 64 + l] : 0;
    if (MODE == 6) { if ((((jb - e0) >> 4) & 3) == 0 && jb + 64 < e1) { for (int p = 0; p < 4; ++p) n6[p] = ids[jb + 64 + p * 16 + l]; } }
    if (MODE == 7) { if ((((jb - e0) >> 4) & 15) == 0 && jb + 256 < e1) { for (int p = 0; p < 16; ++p) n7[p] = ids[jb + 256 + p * 16 + l]; } }
    if (MODE == 5) { if ((((jb - e0) >> 4) & 3) == 0 && jb + 64 < e1) nxt4 = *reinterpret_cast<const int4*>(ids + jb + 64 + l * 4); }
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc.x += b[u].x; acc.y += b[u].y; acc.z += b[u].z; acc.w += b[u].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const long id_mask = (argc > 1 && atoi(argv[1])) ? ((1L << 18) - 1) : ~0L;   // 1: ids wrap inside 1 MB (an L2-resident id stream)
  printf("id stream: %s\n", id_mask == ~0L ? "470 MB from HBM" : "1 MB, L2-resident");
  const unsigned n_rows = 3700u * 1024 / 256;
  const long groups = 1024L * 16, per_group = 7168, E = groups * per_group;   // 117.4 M slots; multiple of 256 (mode 3 reads whole 256-id blocks)
  float4* table; float* out; int* ids;
  CK(hipMalloc(&table, (size_t)n_rows * 256)); CK(hipMemset(table, 0, (size_t)n_rows * 256));
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&ids, (size_t)(E + 1024) * 4)); CK(hipMemset(ids, 0, (size_t)(E + 1024) * 4));
  int* h = (int*)malloc((size_t)E * 4);
  unsigned s = 12345u;
  for (long i = 0; i < E; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 8) % n_rows); }
  CK(hipMemcpy(ids, h, (size_t)E * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 8; ++mode) {
    auto launch = [&]() {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
        default: hipLaunchKernelGGL(k<7>, dim3(1024), dim3(256), 0, 0, table, n_rows, ids, per_group, out, id_mask); break;
      }
    };
    launch(); launch();
    CK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("mode %d : %.3f ms  (%.1f TB/s of row gathers)\n", mode, ms, (double)E * 256 / ms / 1e9);
  }
  return 0;
}
