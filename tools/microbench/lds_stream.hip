// Microbenchmark (go/no-go for an LDS-window streaming driver): every CU streams the whole
// N x 256 B table through two 64 KB LDS buffers (1024-thread workgroup, one per CU), one barrier
// per window; optionally does `reads` random ds_read_b128 row reads + 4 FMAs per thread per window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int WIN_ROWS = 256;                 // 64 KB per buffer
constexpr int THREADS = 1024;

__global__ __launch_bounds__(THREADS) void k_stream(const float4* __restrict__ table, int n_rows,
                                                    float* __restrict__ out, int reads) {
  __shared__ float4 buf[2][WIN_ROWS * 16];
  const int tid = threadIdx.x;
  const int n_win = (n_rows + WIN_ROWS - 1) / WIN_ROWS;
  float4 acc = make_float4(0, 0, 0, 0);
  float4 stage[4];
  // prologue: window 0
  for (int i = 0; i < 4; ++i) {
    const long idx = (long)tid + i * THREADS;
    stage[i] = idx < (long)n_rows * 16 ? table[idx] : make_float4(0, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) buf[0][tid + i * THREADS] = stage[i];
  __syncthreads();
  unsigned rnd = tid * 2654435761u + blockIdx.x;
  for (int w = 0; w < n_win; ++w) {
    const int cur = w & 1;
    if (w + 1 < n_win)
      for (int i = 0; i < 4; ++i) {
        const long idx = (long)(w + 1) * WIN_ROWS * 16 + tid + i * THREADS;
        stage[i] = idx < (long)n_rows * 16 ? table[idx] : make_float4(0, 0, 0, 0);
      }
    for (int r = 0; r < reads; ++r) {       // each 16-lane group reads one random row of the window
      rnd = rnd * 1664525u + 1013904223u;
      const int row = (__shfl((int)(rnd >> 8), 0, 16)) & (WIN_ROWS - 1);
      const float4 x = buf[cur][row * 16 + (tid & 15)];
      acc.x = fmaf(x.x, 1.0001f, acc.x); acc.y = fmaf(x.y, 1.0001f, acc.y);
      acc.z = fmaf(x.z, 1.0001f, acc.z); acc.w = fmaf(x.w, 1.0001f, acc.w);
    }
    if (w + 1 < n_win)
      for (int i = 0; i < 4; ++i) buf[cur ^ 1][tid + i * THREADS] = stage[i];
    __syncthreads();
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

int main(int argc, char** argv) {
  const int n_rows = argc > 1 ? atoi(argv[1]) : 232965;
  const int blocks = argc > 2 ? atoi(argv[2]) : 256;
  float4* table; float* out;
  CK(hipMalloc(&table, (size_t)n_rows * 256)); CK(hipMalloc(&out, 4));
  std::vector<float> h((size_t)n_rows * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 1e-3f;
  CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int reads : {0, 1, 2, 4, 8}) {
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(THREADS), 0, 0, table, n_rows, out, reads);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(THREADS), 0, 0, table, n_rows, out, reads);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double fill = (double)blocks * n_rows * 256.0;
    const double edges = (double)blocks * 64.0 * reads * ((n_rows + WIN_ROWS - 1) / WIN_ROWS);
    printf("rows=%d blocks=%d reads/thread/window=%d : %.3f ms  fill %.1f TB/s  lds-row-reads %.1f M (%.2f G rows/s)\n",
           n_rows, blocks, reads, ms, fill / ms / 1e9, edges / 1e6, edges / ms / 1e6);
  }
  return 0;
}
