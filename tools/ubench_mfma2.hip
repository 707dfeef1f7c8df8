// ubench_mfma2: is the issue rate of v_mfma_f32_32x32x16_f16 DATA dependent?  The split-f16 MLP scorer runs its
// MFMAs at ~64 shader cycles each in two unrelated mappings (nann_mlp.h, nann_mlp2.h) while a bare loop over constant
// operands (tools/ubench_mfma.hip) issues one per 32.  Same loop here with operand sets of different content:
//   mode 0  small smooth values (the round-2 ubench), one operand pair reused by every MFMA
//   mode 1  random f16 operands ~ N(0, 1), one pair reused
//   mode 2  random operands, 8 pairs rotated (a different A and B fragment every MFMA, as the scorer has)
//   mode 3  all-zero operands
// Prints shader ticks per MFMA per SIMD (s_memtime of wave 0), wall ns per MFMA per SIMD and the tick rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma2.hip -o /tmp/ubench_mfma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NSETS, int NT>
__global__ __launch_bounds__(NT) void k_mfma(int iters, const uint4* data, float* out, long long* ticks) {
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  f16x8 x[NSETS], y[NSETS];
#pragma unroll
  for (int s = 0; s < NSETS; ++s) {
    union { uint4 u; f16x8 h; } cx, cy;
    cx.u = data[(2 * s) * 1024 + (threadIdx.x & 1023)];
    cy.u = data[(2 * s + 1) * 1024 + (threadIdx.x & 1023)];
    x[s] = cx.h; y[s] = cy.h;
  }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u)
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[u % NSETS], y[(u / 2) % NSETS], acc[u % NACC], 0, 0, 0);
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

static uint16_t f2h(float f) {  // round-to-nearest-even, normal range
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int e = (int)((x >> 23) & 0xff) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (e <= 0) return (uint16_t)sign;
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);
  uint32_t h = sign | ((uint32_t)e << 10) | (m >> 13);
  if ((m & 0x1fffu) > 0x1000u || ((m & 0x1fffu) == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)h;
}

template <int NACC, int NSETS, int NT>
static void run(const char* name, const uint4* data, float* out, long long* ticks) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mfma<NACC, NSETS, NT>), dim3(blocks), dim3(NT), 0, 0, 50, data, out, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_mfma<NACC, NSETS, NT>), dim3(blocks), dim3(NT), 0, 0, iters, data, out, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
  const double per_simd = (double)iters * 32 * (NT / 64) / 4;
  printf("%-44s %d w/SIMD %d acc %d sets: %.1f ticks/MFMA/SIMD, %.2f ns/MFMA/SIMD, tick rate %.2f GHz, %.0f TFLOP/s\n",
         name, NT / 256, NACC, NSETS, mean / per_simd, ms * 1e6 / per_simd, mean / (ms * 1e6),
         256.0 * 4 * per_simd * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out; long long* ticks; uint4* data[4];
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 8);
  const size_t n16 = 16 * 1024 * 8;  // 16 fragments x 1024 threads x 8 halves
  std::vector<uint16_t> h(n16);
  srand(7);
  auto gauss = [] { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; return s - 6.0f; };
  for (int mode = 0; mode < 4; ++mode) {
    for (size_t i = 0; i < n16; ++i) {
      float v = 0.0f;
      if (mode == 0) v = 0.001f * (float)((i / 8) % 1024 + i % 8);
      else if (mode == 1 || mode == 2) v = gauss();
      h[i] = mode == 3 ? 0 : f2h(v);
    }
    hipMalloc(&data[mode], n16 * 2);
    hipMemcpy(data[mode], h.data(), n16 * 2, hipMemcpyHostToDevice);
  }
  run<1, 1, 256>("smooth small values, chain", data[0], out, ticks);
  run<1, 1, 256>("zeros, chain", data[3], out, ticks);
  run<1, 1, 256>("random N(0,1), one pair, chain", data[1], out, ticks);
  run<2, 1, 256>("random N(0,1), one pair", data[1], out, ticks);
  run<2, 8, 256>("random N(0,1), 8 pairs rotated", data[2], out, ticks);
  run<4, 8, 256>("random N(0,1), 8 pairs rotated", data[2], out, ticks);
  run<2, 8, 512>("random N(0,1), 8 pairs rotated", data[2], out, ticks);
  run<4, 8, 1024>("random N(0,1), 8 pairs rotated", data[2], out, ticks);
  run<2, 8, 256>("smooth, 8 pairs rotated", data[0], out, ticks);
  return 0;
}
