// ubench_mfma3: what may sit between two MFMAs of ONE wavefront without costing matrix-pipe time?  32 MFMAs
// (v_mfma_f32_32x32x16_f16, random operands, 8 operand pairs rotated) per iteration, one wavefront per SIMD, with a
// pattern of other instructions pinned between them (sched_barrier).  Prints shader ticks per MFMA.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma3.hip -o tools/_build/ubench_mfma3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PIN __builtin_amdgcn_sched_barrier(0)

// PAT: 0 chain            1 chain + ds_read      2 chain + 2 VALU       3 two acc + ds_read   4 two acc + 4 VALU
//      5 two acc + 8 VALU  6 two acc + 12 VALU    7 chain, A operand = ds_read issued 8 MFMAs earlier
//      8 two acc + 4 v_accvgpr-style reads of the OTHER accumulator (completed)   9 four acc + 8 VALU
template <int PAT>
__global__ __launch_bounds__(256) void k_pat(int iters, const uint4* data, float* out, long long* ticks) {
  __shared__ uint4 lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = data[i];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
  f16x8 x[8], y[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    union { uint4 u; f16x8 h; } cx, cy;
    cx.u = data[(2 * s) * 1024 + threadIdx.x];
    cy.u = data[(2 * s + 1) * 1024 + threadIdx.x];
    x[s] = cx.h; y[s] = cy.h;
  }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x + i) * 1e-3f;
  uint4 ring[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = lds[i * 64 + lane];
  float sink = 0.0f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      constexpr int NACC = (PAT == 0 || PAT == 1 || PAT == 2 || PAT == 7) ? 1 : (PAT == 9 ? 4 : 2);
      f16x8 a_op = x[u % 8];
      if constexpr (PAT == 7) { union { uint4 q; f16x8 h; } c; c.q = ring[u % 8]; a_op = c.h; }
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_op, y[(u / 2) % 8], acc[u % NACC], 0, 0, 0);
      PIN;
      if constexpr (PAT == 1 || PAT == 3) {
        uint4 t = lds[((u + it) % 16) * 64 + lane];
        asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
        ring[u % 8] = t;
      }
      if constexpr (PAT == 7) ring[u % 8] = lds[((u + it) % 16) * 64 + lane];
      constexpr int NV = PAT == 2 ? 2 : PAT == 4 ? 4 : (PAT == 5 || PAT == 9) ? 8 : PAT == 6 ? 12 : 0;
#pragma unroll
      for (int k = 0; k < NV; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      if constexpr (PAT == 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sink += acc[(u + 1) % 2][(4 * u + k) % 16];
      }
      PIN;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = sink;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)ring[i].x;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
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

template <int PAT>
static void run(const char* name, const uint4* data, float* out, long long* ticks) {
  const int iters = 1000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_pat<PAT>), dim3(blocks), dim3(256), 0, 0, 50, data, out, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_pat<PAT>), dim3(blocks), dim3(256), 0, 0, iters, data, out, ticks);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
  const double n = (double)iters * 32;
  printf("PAT %d %-52s %.1f ticks/MFMA  %.2f ns/MFMA  tick rate %.2f GHz\n", PAT, name, mean / n, ms * 1e6 / n, mean / (ms * 1e6));
}

int main() {
  float* out; long long* ticks; uint4* data;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
  const size_t n16 = 16 * 1024 * 8;
  std::vector<uint16_t> h(n16);
  srand(7);
  for (auto& v : h) { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; v = f2h(s - 6.0f); }
  hipMalloc(&data, n16 * 2); hipMemcpy(data, h.data(), n16 * 2, hipMemcpyHostToDevice);
  run<0>("chain", data, out, ticks);
  run<1>("chain + 1 ds_read_b128 between", data, out, ticks);
  run<2>("chain + 2 VALU between", data, out, ticks);
  run<3>("two accumulators + 1 ds_read_b128 between", data, out, ticks);
  run<4>("two accumulators + 4 VALU between", data, out, ticks);
  run<5>("two accumulators + 8 VALU between", data, out, ticks);
  run<6>("two accumulators + 12 VALU between", data, out, ticks);
  run<7>("chain, A operand from a ds_read 8 MFMAs earlier", data, out, ticks);
  run<8>("two accumulators + 4 reads of the other accumulator", data, out, ticks);
  run<9>("four accumulators + 8 VALU between", data, out, ticks);
  return 0;
}
