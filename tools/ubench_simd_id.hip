// ubench_simd_id: which SIMD does each wavefront of a 512-thread workgroup run on?  (HW_ID bits 5:4.)
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_simd_id.hip -o tools/_build/ubench_simd_id
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(int* out) {
  const int hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}
int main() {
  int* d; hipMalloc(&d, 1024 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1024), dim3(512), 0, 0, d);
  static int h[1024 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int b = 0; b < 1024; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: SIMD 0..3 counts %d %d %d %d   (block 0: hw_id 0x%08x simd %d wave_slot %d)\n", w, hist[w][0], hist[w][1],
                                     hist[w][2], hist[w][3], h[w], (h[w] >> 4) & 3, h[w] & 15);
  return 0;
}
