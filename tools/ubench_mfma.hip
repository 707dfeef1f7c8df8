// ubench_mfma: what does a SIMD's matrix pipe deliver for v_mfma_f32_32x32x16_f16 as a function of
// (wavefronts per SIMD) x (independent accumulators per wavefront) x (chain or not)?  Prints shader
// cycles per MFMA per SIMD (s_memtime around the loop of wave 0) and the effective clock.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NT>
__global__ __launch_bounds__(NT) void k_mfma(int iters, float* out, long long* ticks) {
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = (float)(threadIdx.x + a + r);
  f16x8 x, y;
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (float)(threadIdx.x + i)); y[i] = (_Float16)(0.002f * (float)(threadIdx.x ^ i)); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NACC, int NT>
static void run(const char* name, float* out, long long* ticks) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mfma<NACC, NT>), dim3(blocks), dim3(NT), 0, 0, 10, out, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_mfma<NACC, NT>), dim3(blocks), dim3(NT), 0, 0, iters, out, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
  const double per_simd = (double)iters * 32 * (NT / 64) / 4;  // MFMAs one SIMD issued
  printf("%-34s %2d waves/SIMD %d acc: %.3f ms, %.1f ticks/MFMA/SIMD, wall %.2f ns/MFMA/SIMD, tick rate %.2f GHz, %.0f TFLOP/s\n",
         name, NT / 256, NACC, ms, mean / per_simd, ms * 1e6 / per_simd, mean / (ms * 1e6),
         256.0 * 4 * per_simd * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 8);
  run<1, 256>("chain, 1 wave per SIMD", out, ticks);
  run<2, 256>("2 accumulators, 1 wave per SIMD", out, ticks);
  run<4, 256>("4 accumulators, 1 wave per SIMD", out, ticks);
  run<1, 512>("chain, 2 waves per SIMD", out, ticks);
  run<2, 512>("2 accumulators, 2 waves per SIMD", out, ticks);
  run<4, 512>("4 accumulators, 2 waves per SIMD", out, ticks);
  run<4, 1024>("4 accumulators, 4 waves per SIMD", out, ticks);
  return 0;
}
