// ubench_gather: what does the chip deliver for the access pattern of the scoring phase --
// random 256-byte rows (128 x f16), one 16-byte load per lane, 16 lanes per row, 8 rows in
// flight per lane -- as a function of table size (Infinity-Cache resident or not) and of the
// number of 1024-thread workgroups per CU?  Prints GB/s; build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

template <int U>
__global__ __launch_bounds__(1024) void k_gather(const uint4* __restrict__ table, uint32_t row_mask,
                                                 uint32_t rows_per_wave, uint32_t* out) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const uint32_t wave_global = blockIdx.x * 16 + (threadIdx.x >> 6);
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
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
  const size_t max_rows = 1u << 24;  // 16M rows x 256 B = 4 GiB
  uint4* table; uint32_t* out;
  if (hipMalloc(&table, max_rows * 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 4096 * 1024 * 4);
  hipMemset(table, 1, max_rows * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const uint32_t total_rows = 32u << 20;  // 32M row reads = 8 GiB per launch
  for (int lg : {20, 22, 24}) {
    for (int grid : {256, 512, 1024}) {
      const uint32_t rpw = total_rows / (grid * 16);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(1024), 0, 0, table, (1u << lg) - 1u, rpw, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("rows=2^%d (%4zu MiB) grid=%4d U=8  %.3f ms  %.0f GB/s\n", lg, ((size_t)256 << lg) >> 20, grid,
                        ms, (double)total_rows * 256 / (ms * 1e-3) / 1e9);
      }
    }
  }
  for (int lg : {20, 24}) {  // twice the loads in flight per lane
    const int grid = 256;
    const uint32_t rpw = total_rows / (grid * 16);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(1024), 0, 0, table, (1u << lg) - 1u, rpw, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("rows=2^%d (%4zu MiB) grid=%4d U=16 %.3f ms  %.0f GB/s\n", lg, ((size_t)256 << lg) >> 20, grid,
                      ms, (double)total_rows * 256 / (ms * 1e-3) / 1e9);
    }
  }
  return 0;
}
