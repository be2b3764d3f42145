// Model of a TWO-SIDED tile transpose of the per-edge scalars of the column-major walk passes (round-3 verdict,
// item 1a): can a pre-pass produce w in the column plan's slot order for <= 0.25 ms per array, so that the walk's
// feeder waves stream the weights instead of gathering one 4-byte scalar per L2 request (0.54 ms per pass)?
//
// Shape modelled (Reddit: N = 232,965 rows of ~492 slots, E = 114.6 M; the model uses 262,144 rows x 448 slots):
// the destination is the walk order -- per (bin of 15 columns, window of R rows) one contiguous run -- the source
// is the row-major edge order, where the slots of row i inside a group of C columns are ONE run of
// r = C * 492 / N scalars.  A tile = (R rows) x (C columns) = S = R * r scalars that fit a CU's LDS:
//   phase 1: gather the tile's R source runs into LDS (lanes of one run share a 128-B line: one L2 request per run
//            instead of one per scalar), through a 4-byte source index per slot (src32, streamed);
//   phase 2: every destination slot reads its scalar from LDS through a 16-bit in-tile index (perm16, streamed)
//            and the tile leaves in coalesced 16-byte stores.
// Tiles of one window are swept by ONE XCD (blockIdx % 8), column group after column group, so the source lines a
// run shares with the neighbouring column groups are L2 hits.  Metadata: 6 B per slot (src32 + perm16); the
// "lean" variant derives the source index from one 4-byte run start per run (4 / r B per slot) instead of src32.
// Reference point in the same harness: r = 1 (every scalar its own request) = the one-sided gather.
//   hipcc --offload-arch=gfx950 -O3 tile_transpose.hip -o tile_transpose && ./tile_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int vint4 __attribute__((ext_vector_type(4)));
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef unsigned short vushort4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned mix_bits(unsigned q, int bits, unsigned salt) {   // a bijection on `bits`-bit words
  const unsigned mask = (1u << bits) - 1;
  q = (q * 0x9E3779B1u + salt) & mask;
  q ^= q >> (bits / 2);
  q = (q * 0x85EBCA6Bu) & mask;
  q ^= q >> (bits / 3 + 1);
  q = (q * 0xC2B2AE35u + 1u) & mask;   // odd multipliers / xor-shifts: every step is invertible modulo 2^bits
  return q;
}

// setup: the metadata a plan would hold
__global__ void k_setup(int* __restrict__ src32, int* __restrict__ run_start, unsigned short* __restrict__ perm16,
                        long long n_tiles, int S, int sbits, int r, int R, int G, long long row_len) {
  const long long n = n_tiles * S;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const long long t = p / S;
    const int q = (int)(p % S);
    const long long win = t / G, g = t % G;
    const int i = q / r, o = q % r;                         // row inside the window, position in the run
    const long long s0 = (win * R + i) * row_len + g * r;   // the run of row i inside column group g
    src32[p] = (int)(s0 + o);
    if (o == 0) run_start[t * R + i] = (int)s0;
    perm16[p] = (unsigned short)mix_bits((unsigned)q, sbits, (unsigned)t * 2654435761u);
  }
}

// MODE 0: src32 per slot; MODE 1 ("lean"): one run start per run, lanes of a run derive their index
template <int T, int MODE>
__global__ __launch_bounds__(T) void k_transpose(const float* __restrict__ w, const int* __restrict__ src32,
                                                 const int* __restrict__ run_start,
                                                 const unsigned short* __restrict__ perm16, float* __restrict__ out,
                                                 int S, int r, int rshift, int R, int G, int W, int wgs_per_xcd) {
  extern __shared__ float tile[];
  // XCD slot x sweeps the windows x, x + 8, ...; its workgroups take consecutive column groups of the window
  const int x = blockIdx.x % 8, i_in_x = blockIdx.x / 8;
  for (int win = x; win < W; win += 8) {
    for (int g = i_in_x; g < G; g += wgs_per_xcd) {
      const long long t = (long long)win * G + g;
      const long long base = t * S;
      // phase 1: source runs -> LDS
      for (int q0 = threadIdx.x * 4; q0 < S; q0 += T * 4) {
        vint4 s;
        if constexpr (MODE == 0) {
          s = __builtin_nontemporal_load(reinterpret_cast<const vint4*>(src32 + base + q0));
        } else {
          // run length r = 1 << rshift (power of two in this model): slot q belongs to run q >> rshift
          const int* rs = run_start + t * R;
          s.x = rs[(q0 + 0) >> rshift] + ((q0 + 0) & (r - 1));
          s.y = rs[(q0 + 1) >> rshift] + ((q0 + 1) & (r - 1));
          s.z = rs[(q0 + 2) >> rshift] + ((q0 + 2) & (r - 1));
          s.w = rs[(q0 + 3) >> rshift] + ((q0 + 3) & (r - 1));
        }
        vfloat4 v;
        v.x = w[s.x]; v.y = w[s.y]; v.z = w[s.z]; v.w = w[s.w];
        *reinterpret_cast<vfloat4*>(tile + q0) = v;
      }
      __syncthreads();
      // phase 2: LDS -> destination order, coalesced
      for (int q0 = threadIdx.x * 4; q0 < S; q0 += T * 4) {
        const vushort4 p = __builtin_nontemporal_load(reinterpret_cast<const vushort4*>(perm16 + base + q0));
        vfloat4 v;
        v.x = tile[p.x]; v.y = tile[p.y]; v.z = tile[p.z]; v.w = tile[p.w];
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(out + base + q0));
      }
      __syncthreads();
    }
  }
}

int main() {
  const long long n_rows = 262144, row_len = 448;            // 117.4 M slots
  const long long E = n_rows * row_len;
  float* w; CK(hipMalloc(&w, E * 4)); CK(hipMemset(w, 0, E * 4));
  float* out; CK(hipMalloc(&out, E * 4));
  int* src32; CK(hipMalloc(&src32, E * 4));
  unsigned short* perm16; CK(hipMalloc(&perm16, E * 2));
  int* run_start; CK(hipMalloc(&run_start, E * 4));          // (E / r entries used)
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct Cfg { int S, r, T, wgs_per_cu; };
  // S = scalars per tile (LDS), r = source run length; rows per tile R = S / r; column groups per window G = row_len / r
  const Cfg cfgs[] = {
      {32768, 1, 1024, 1},  {32768, 2, 1024, 1},  {32768, 4, 1024, 1},  {32768, 8, 1024, 1},
      {32768, 16, 1024, 1}, {32768, 32, 1024, 1}, {16384, 4, 512, 2},   {16384, 8, 512, 2},
      {16384, 16, 512, 2},  {8192, 4, 256, 4},    {8192, 8, 256, 4},
  };
  for (const Cfg& c : cfgs) {
    const int R = c.S / c.r, G = (int)(row_len / c.r), W = (int)(n_rows / R);
    const long long n_tiles = (long long)W * G;
    int sbits = 0; while ((1 << sbits) < c.S) ++sbits;
    int rshift = 0; while ((1 << rshift) < c.r) ++rshift;
    hipLaunchKernelGGL(k_setup, dim3(4096), dim3(256), 0, 0, src32, run_start, perm16, n_tiles, c.S, sbits, c.r, R, G, row_len);
    CK(hipDeviceSynchronize());
    const int blocks = 256 * c.wgs_per_cu;
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
#define GO(TT, MM) do { auto kfn = k_transpose<TT, MM>;                                                              \
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, c.S * 4));                  \
    hipLaunchKernelGGL(kfn, dim3(blocks), dim3(TT), c.S * 4, 0, w, src32, run_start, perm16, out, c.S, c.r, rshift, \
                       R, G, W, blocks / 8); } while (0)
        if (c.T == 1024) { if (mode == 0) GO(1024, 0); else GO(1024, 1); }
        else if (c.T == 512) { if (mode == 0) GO(512, 0); else GO(512, 1); }
        else { if (mode == 0) GO(256, 0); else GO(256, 1); }
#undef GO
        CK(hipGetLastError());
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      const double bytes = (double)E * (4 + 4 + 2 + (mode == 0 ? 4.0 : 4.0 / c.r));
      printf("tile S=%5d (R=%5d rows x r=%2d) T=%4d x%d/CU  %s: %.3f ms for %lld scalars (x %.3f at 114.6 M = %.3f ms; %.2f TB/s of payload + metadata)\n",
             c.S, R, c.r, c.T, c.wgs_per_cu, mode == 0 ? "src32 per slot " : "run starts only", best, E,
             114615892.0 / E, best * 114615892.0 / E, bytes / best / 1e9);
    }
  }
  return 0;
}
