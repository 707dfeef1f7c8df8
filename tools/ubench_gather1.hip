// ubench_gather1: the gather rate of ONE workgroup -- what a lone query's scoring phase can pull through one CU.
// Random 256-byte rows, one 16-byte load per lane, 16 lanes per row, U rows in flight per lane, 1024 threads;
// grids of 1 / 8 / 64 / 256 workgroups (one per CU).  Prints GB/s per workgroup.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gather1.hip -o /tmp/ubench_gather1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

template <int U, int NT>
__global__ __launch_bounds__(NT) void k_gather(const uint4* __restrict__ table, uint32_t row_mask,
                                               uint32_t rows_per_wave, uint32_t* out) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const uint32_t wave_global = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  uint32_t acc = 0;
  for (uint32_t i0 = 0; i0 < rows_per_wave; i0 += 4 * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row = mix(wave_global * rows_per_wave + i0 + u * 4 + grp) & row_mask;
      v[u] = table[(size_t)row * 16 + sub];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x + v[u].y + v[u].z + v[u].w;
  }
  out[blockIdx.x * NT + threadIdx.x] = acc;
}

template <int U, int NT>
static void run(const uint4* table, uint32_t* out, int lg, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t rpw = 8192;  // rows per wavefront
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<U, NT>), dim3(grid), dim3(NT), 0, 0, table, (1u << lg) - 1u, rpw, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double bytes_per_wg = (double)rpw * (NT / 64) * 256;
  printf("rows=2^%d grid=%3d threads=%4d U=%2d  %.3f ms  %.1f GB/s per workgroup  (%.0f GB/s total)\n", lg, grid, NT, U, ms,
         bytes_per_wg / (ms * 1e-3) / 1e9, bytes_per_wg * grid / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t max_rows = 1u << 22;  // 1 GiB
  uint4* table; uint32_t* out;
  if (hipMalloc(&table, max_rows * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 256 * 1024 * 4);
  hipMemset(table, 1, max_rows * 256);
  for (int lg : {20, 22})
    for (int grid : {1, 8, 64, 256}) {
      run<4, 1024>(table, out, lg, grid);
      run<8, 1024>(table, out, lg, grid);
      run<16, 1024>(table, out, lg, grid);
      run<8, 512>(table, out, lg, grid);
      run<16, 512>(table, out, lg, grid);
    }
  return 0;
}
