// Model of the L2-TILE form of the two-sided transpose of the column-major passes' per-edge scalars, exactly as the
// round-3 verdict (item 1a) words it: tiles of Rt rows x Ct columns of the edge matrix whose DESTINATION region
// (the tile's scalars in the column plan's slot order, S2 = Rt * r scalars, 1-4 MB) is written by ONE XCD, so that the
// 4-byte stores of different rows into one destination line meet in that XCD's L2 and leave it as whole lines.
// (tile_transpose.hip models the other form: tiles that fit a CU's LDS, both sides coalesced.)
//
// Shape (as in tile_transpose.hip): 262,144 rows x 448 slots = 117.4 M scalars, row-major.  Inside a tile the slots of
// row i are ONE run of r scalars (r = Ct * 492 / N on the Reddit shape: 32-64 for Ct = 15-30 k columns); the lanes of a
// run read consecutive addresses (one or two L2 requests per run).  Destination: scalar o of row i goes to
// tile_base + o * (Rt + pad) + i -- the column's slots of consecutive rows are adjacent (what the walk order is: per
// column the slots of a row window are one run), the lanes of one store instruction hit r different lines.
//   mode 0: destination index computed (lower bound: no metadata stream)
//   mode 1: destination index streamed from a 4-byte-per-slot array in source order (what a plan would hold)
// Launch: XCD slot x = blockIdx % 8 takes the tiles x, x + 8, ...; the XCD's workgroups split the tile's rows.
//   hipcc --offload-arch=gfx950 -O3 tile_transpose_l2.hip -o tile_transpose_l2 && ./tile_transpose_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kPad = 33;   // keeps the lanes of a store instruction off one L2 channel

// spread = 1: the 32 slots of a destination line come from rows spread evenly over the tile's Rt rows (the real walk
// order: a column's slots inside a row window are its ~17 source rows in ascending order, anywhere in the window), so a
// line stays partially written for the whole sweep of the tile instead of being filled by 32 consecutive rows at once
__global__ void k_setup(int* __restrict__ dst32, long long n_rows, long long row_len, int Rt, int r, int spread = 0) {
  const long long n = n_rows * row_len;
  const int G = (int)(row_len / r);                      // column groups (tiles per row window)
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const long long row = p / row_len;
    const int s = (int)(p % row_len), g = s / r, o = s % r;
    const long long win = row / Rt;
    int i = (int)(row % Rt);
    if (spread) i = (i % (Rt / 32)) * 32 + i / (Rt / 32);
    const long long tile = win * G + g;
    dst32[p] = (int)(tile * ((long long)(Rt + kPad) * r) + (long long)o * (Rt + kPad) + i);
  }
}

template <int MODE, int U>
__global__ __launch_bounds__(256) void k_l2tile(const float* __restrict__ w, const int* __restrict__ dst32,
                                                float* __restrict__ out, long long row_len, int Rt, int r, int G,
                                                int W, int waves_per_xcd) {
  const int x = blockIdx.x % 8;
  const int wave_in_xcd = (blockIdx.x / 8) * 4 + threadIdx.x / 64;
  const int lane = threadIdx.x % 64;
  const int rows_per_inst = 64 / r;                      // r <= 64, power of two
  const int sub = lane / r, o = lane % r;
  const long long n_tiles = (long long)W * G;
  for (long long t = x; t < n_tiles; t += 8) {
    const long long win = t / G;
    const int g = (int)(t % G);
    const long long tile_base = t * ((long long)(Rt + kPad) * r);
    // this wave's rows: wave_in_xcd * rows_per_inst + sub, stepping by waves_per_xcd * rows_per_inst, U at a time
    for (int i0 = wave_in_xcd * rows_per_inst; i0 < Rt; i0 += waves_per_xcd * rows_per_inst * U) {
      float v[U];
      int d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * waves_per_xcd * rows_per_inst + sub;
        const long long src = (win * Rt + (i < Rt ? i : 0)) * row_len + (long long)g * r + o;
        v[u] = __builtin_nontemporal_load(w + src);
        if constexpr (MODE == 1) d[u] = __builtin_nontemporal_load(dst32 + src);
        else d[u] = (int)(tile_base + (long long)o * (Rt + kPad) + i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * waves_per_xcd * rows_per_inst + sub;
        if (i < Rt) out[d[u]] = v[u];
      }
    }
  }
}

// V4: every lane moves FOUR consecutive scalars of a run (one 16-byte load of w and of dst32 per lane): four times the
// bytes in flight per wave at the same number of resident waves (more waves per CU spread the rows of a destination line
// over more time and lose the combining, see the WGs/CU sweep).
typedef int vint4 __attribute__((ext_vector_type(4)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));
template <int MODE, int U>
__global__ __launch_bounds__(256) void k_l2tile_v4(const float* __restrict__ w, const int* __restrict__ dst32,
                                                   float* __restrict__ out, long long row_len, int Rt, int r, int G,
                                                   int W, int waves_per_xcd) {
  const int x = blockIdx.x % 8;
  const int wave_in_xcd = (blockIdx.x / 8) * 4 + threadIdx.x / 64;
  const int lane = threadIdx.x % 64;
  const int lanes_per_row = r / 4;                       // r >= 4
  const int rows_per_inst = 64 / lanes_per_row;
  const int sub = lane / lanes_per_row, o = (lane % lanes_per_row) * 4;
  const long long n_tiles = (long long)W * G;
  for (long long t = x; t < n_tiles; t += 8) {
    const long long win = t / G;
    const int g = (int)(t % G);
    const long long tile_base = t * ((long long)(Rt + kPad) * r);
    for (int i0 = wave_in_xcd * rows_per_inst; i0 < Rt; i0 += waves_per_xcd * rows_per_inst * U) {
      vfloat4 v[U];
      vint4 d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * waves_per_xcd * rows_per_inst + sub;
        const long long src = (win * Rt + (i < Rt ? i : 0)) * row_len + (long long)g * r + o;
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(w + src));
        if constexpr (MODE == 1) d[u] = __builtin_nontemporal_load(reinterpret_cast<const vint4*>(dst32 + src));
        else {
          const int b = (int)(tile_base + (long long)o * (Rt + kPad) + i);
          d[u] = vint4{b, b + (Rt + kPad), b + 2 * (Rt + kPad), b + 3 * (Rt + kPad)};
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * waves_per_xcd * rows_per_inst + sub;
        if (i < Rt) { out[d[u].x] = v[u].x; out[d[u].y] = v[u].y; out[d[u].z] = v[u].z; out[d[u].w] = v[u].w; }
      }
    }
  }
}

int main() {
  const long long n_rows = 262144, row_len = 448;
  const long long E = n_rows * row_len;
  float* w; CK(hipMalloc(&w, E * 4)); CK(hipMemset(w, 0, E * 4));
  const long long out_elems = E + E / 256 + (1 << 20);       // room for the pad rows
  float* out; CK(hipMalloc(&out, out_elems * 4));
  int* dst32; CK(hipMalloc(&dst32, E * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct Cfg { int Rt, r; };
  const Cfg cfgs[] = {{32768, 32}, {16384, 64}, {16384, 32}, {8192, 64}, {8192, 32}, {4096, 64}, {16384, 16}, {32768, 8}};
  for (const Cfg& c : cfgs) {
    const int G = (int)(row_len / c.r), W = (int)(n_rows / c.Rt);
    hipLaunchKernelGGL(k_setup, dim3(4096), dim3(256), 0, 0, dst32, n_rows, row_len, c.Rt, c.r);
    CK(hipDeviceSynchronize());
    for (int wgs_per_cu = 2; wgs_per_cu <= 8; wgs_per_cu *= 2) {
      const int blocks = 256 * wgs_per_cu;
      for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(a));
          if (mode == 0)
            hipLaunchKernelGGL((k_l2tile<0, 4>), dim3(blocks), dim3(256), 0, 0, w, dst32, out, row_len, c.Rt, c.r, G, W, blocks / 8 * 4);
          else
            hipLaunchKernelGGL((k_l2tile<1, 4>), dim3(blocks), dim3(256), 0, 0, w, dst32, out, row_len, c.Rt, c.r, G, W, blocks / 8 * 4);
          CK(hipGetLastError());
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          best = ms < best ? ms : best;
        }
        printf("L2 tile Rt=%5d rows x r=%2d (destination region %.2f MB) %d WGs/CU  %s: %.3f ms for %lld scalars (x 0.976 at 114.6 M = %.3f ms)\n",
               c.Rt, c.r, (double)c.Rt * c.r * 4 / 1048576.0, wgs_per_cu,
               mode == 0 ? "index computed" : "index streamed", best, E, best * 114615892.0 / E);
        fflush(stdout);
      }
    }
    for (int pass = 0; pass < 2; ++pass)
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
      const int blocks = 256 * wgs_per_cu;
      if (pass == 1 && wgs_per_cu == 1) {
        hipLaunchKernelGGL(k_setup, dim3(4096), dim3(256), 0, 0, dst32, n_rows, row_len, c.Rt, c.r, 1);
        CK(hipDeviceSynchronize());
      }
      for (int variant = pass; variant < 4; variant += 1 + pass) {   // second pass: the streamed-index variants on the spread layout      // (mode, U): (0,1) (1,1) (0,2) (1,2)
        const int mode = variant & 1, U = variant < 2 ? 1 : 2;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipEventRecord(a));
#define GOV(MM, UU) hipLaunchKernelGGL((k_l2tile_v4<MM, UU>), dim3(blocks), dim3(256), 0, 0, w, dst32, out, row_len, c.Rt, c.r, G, W, blocks / 8 * 4)
          if (variant == 0) GOV(0, 1); else if (variant == 1) GOV(1, 1); else if (variant == 2) GOV(0, 2); else GOV(1, 2);
#undef GOV
          CK(hipGetLastError());
          CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
          float ms; CK(hipEventElapsedTime(&ms, a, b));
          best = ms < best ? ms : best;
        }
        printf("L2 tile Rt=%5d rows x r=%2d V4 (16-B loads) U=%d %d WGs/CU  %s%s: %.3f ms (x 0.976 at 114.6 M = %.3f ms)\n",
               c.Rt, c.r, U, wgs_per_cu, mode == 0 ? "index computed" : "index streamed", pass ? ", lines filled over the whole sweep" : "",
               best, best * 114615892.0 / E);
        fflush(stdout);
      }
    }
  }
  return 0;
}
