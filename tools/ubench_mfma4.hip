// ubench_mfma4: do the VALU stretches of one wavefront run under the MFMAs of ANOTHER wavefront of the same SIMD?
// Each wavefront loops over: 16 chained MFMAs, V1 dependence-free VALU, 24 MFMAs (4 accumulators), V2 VALU -- the
// shape of one hidden tile of the split-f16 MLP scorer -- with W wavefronts per SIMD.  If the stretches overlap, a
// SIMD's time per iteration is max(W x 40 x 32, 1280 + 2.4 (V1 + V2)); if they do not, the sum.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma4.hip -o tools/_build/ubench_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PIN __builtin_amdgcn_sched_barrier(0)

template <int V1, int V2, int NT, int STAGGER>
__global__ __launch_bounds__(NT) void k_tile(int iters, const uint4* data, float* out, long long* ticks) {
  f32x16 acc[5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  f16x8 x[8], y[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    union { uint4 u; f16x8 h; } cx, cy;
    cx.u = data[(2 * s) * 1024 + (threadIdx.x & 1023)];
    cy.u = data[(2 * s + 1) * 1024 + (threadIdx.x & 1023)];
    x[s] = cx.h; y[s] = cy.h;
  }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  __syncthreads();
  if (STAGGER && (threadIdx.x >> 8) == 1) {  // the second wavefront of each SIMD starts half a tile late
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[u % 8], y[(u / 2) % 8], acc[4], 0, 0, 0); PIN; }
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[u % 8], y[(u / 2) % 8], acc[0], 0, 0, 0); PIN; }
#pragma unroll
    for (int k = 0; k < V1; ++k) v[k % 16] = __builtin_fmaf(v[k % 16], 1.0001f, 0.5f);
    PIN;
#pragma unroll
    for (int u = 0; u < 24; ++u) { acc[1 + u / 6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[u % 8], y[(u / 3) % 8], acc[1 + u / 6], 0, 0, 0); PIN; }
#pragma unroll
    for (int k = 0; k < V2; ++k) v[k % 16] = __builtin_fmaf(v[k % 16], 0.9999f, 0.25f);
    PIN;
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * NT + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * (NT / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

static uint16_t f2h(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int e = (int)((x >> 23) & 0xff) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (e <= 0) return (uint16_t)sign;
  if (e >= 31) return (uint16_t)(sign | 0x7bffu);
  return (uint16_t)(sign | ((uint32_t)e << 10) | (m >> 13));
}

template <int V1, int V2, int NT, int STAGGER>
static void run(const char* name, const uint4* data, float* out, long long* ticks) {
  const int iters = 400, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_tile<V1, V2, NT, STAGGER>), dim3(blocks), dim3(NT), 0, 0, 20, data, out, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_tile<V1, V2, NT, STAGGER>), dim3(blocks), dim3(NT), 0, 0, iters, data, out, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h((size_t)blocks * (NT / 64));
  hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto t : h) mean += (double)t; mean /= (double)h.size();
  const int W = NT / 256;
  printf("%-40s W=%d V1=%3d V2=%3d: %.0f ticks/iteration (pipe needs %d, a wave alone %d + %.0f), %.2f us, tick rate %.2f GHz\n", name, W, V1, V2,
         mean / iters, W * 40 * 32, 40 * 32, 2.4 * (V1 + V2), ms * 1e3 / iters, mean / (ms * 1e6));
}

int main() {
  float* out; long long* ticks; uint4* data;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 16 * 8);
  const size_t n16 = 16 * 1024 * 8;
  std::vector<uint16_t> h(n16);
  srand(7);
  for (auto& v : h) { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; v = f2h(s - 6.0f); }
  hipMalloc(&data, n16 * 2); hipMemcpy(data, h.data(), n16 * 2, hipMemcpyHostToDevice);
  run<0, 0, 256, 0>("MFMAs only", data, out, ticks);
  run<130, 0, 256, 0>("+ 130 VALU after layer 1", data, out, ticks);
  run<0, 0, 512, 0>("MFMAs only", data, out, ticks);
  run<130, 0, 512, 0>("+ 130 VALU after layer 1", data, out, ticks);
  run<130, 130, 512, 0>("+ 130 VALU after each layer", data, out, ticks);
  run<260, 260, 512, 0>("+ 260 VALU after each layer", data, out, ticks);
  run<130, 130, 512, 1>("same, second wavefront half a tile late", data, out, ticks);
  run<260, 260, 512, 1>("same, second wavefront half a tile late", data, out, ticks);
  run<130, 130, 768, 0>("+ 130 VALU after each layer", data, out, ticks);
  run<130, 130, 1024, 0>("+ 130 VALU after each layer", data, out, ticks);
  return 0;
}
