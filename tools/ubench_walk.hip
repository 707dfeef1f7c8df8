// ubench_walk.hip -- isolates the cost of the single-wavefront walker (nann_device.h) on the
// GPU box:  hipcc -O3 --offload-arch=gfx950 -I nann_amd/csrc tools/ubench_walk.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nann_device.h"
using namespace nann;

// MODE 0: full wave_walk_span; 1: ids + bitmap ops only (no resolution); 2: resolution without
// the global store (kept ids to LDS); 3: full, with 15 other wavefronts streaming LDS writes
template <int MODE>
__global__ __launch_bounds__(1024) void k(const int32_t* ids, int n, uint32_t n_items, uint32_t bm_words,
                                          int32_t* out, long long* ticks, int* kept_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* bm = reinterpret_cast<uint32_t*>(smem);
  int32_t* stage = reinterpret_cast<int32_t*>(smem + (size_t)bm_words * 4);  // 4096 ids
  int32_t* lout = stage + 4096;                                             // 4096 ids
  for (uint32_t i = threadIdx.x; i < bm_words; i += 1024) bm[i] = (i * 2654435761u) & ((i * 40503u) << 7);  // ~25 % set
  for (int i = threadIdx.x; i < n; i += 1024) stage[i] = ids[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave == 0) {
    int err = 0;
    const long long t0 = clock64();
    int base = 0;
    if (MODE == 0 || MODE == 3) {
      base = wave_walk_span<true>(stage, n, bm, n_items, out, 0, &err);
    } else if (MODE == 1) {
      uint32_t acc = 0;
      for (int c = 0; c < n; c += 64) {
        const int32_t x = stage[min(c + lane, n - 1)];
        uint32_t* w = bm + ((uint32_t)x >> 5);
        const uint32_t bit = 1u << (x & 31);
        const uint32_t pre = *w;
        const uint32_t old = atomicOr(w, bit);
        acc += (pre ^ old) & bit;
      }
      base = (int)acc;
    } else {
      base = wave_walk_span<true>(stage, n, bm, n_items, lout, 0, &err);
    }
    const long long t1 = clock64();
    if (lane == 0) { ticks[0] = t1 - t0; kept_out[0] = base; }
  } else if (MODE == 3) {
    // LDS write traffic from the other wavefronts while the walker runs (bounded)
    for (int it = 0; it < 64; ++it) lout[(threadIdx.x * 7 + it * 1024) & 4095] = it;
  }
}

int main() {
  const int n = 4096;
  const uint32_t n_items = 1000000, bm_words = 31252;
  std::vector<int32_t> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (i % 7 == 3 && i > 8) ? h[i - 5] : (int32_t)(((long long)rand() * 7919) % n_items);
  int32_t *d_ids, *d_out; long long* d_t; int* d_k;
  hipMalloc(&d_ids, n * 4); hipMalloc(&d_out, n * 4); hipMalloc(&d_t, 64); hipMalloc(&d_k, 64);
  hipMemcpy(d_ids, h.data(), n * 4, hipMemcpyHostToDevice);
  const size_t lds = (size_t)bm_words * 4 + 2 * 4096 * 4;
  auto run = [&](auto kern, const char* name) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long long t = 0; int kept = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, 0, d_ids, n, n_items, bm_words, d_out, d_t, d_k);
      hipDeviceSynchronize();
      hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost); hipMemcpy(&kept, d_k, 4, hipMemcpyDeviceToHost);
    }
    printf("%-28s %8lld ticks  %6.1f ticks/step  kept %d  (%s)\n", name, t, (double)t / (n / 64), kept,
           hipGetErrorString(hipGetLastError()));
  };
  run(k<0>, "full walker");
  run(k<1>, "LDS ops only (unbatched)");
  run(k<2>, "full, kept ids to LDS");
  run(k<3>, "full + LDS traffic");
  return 0;
}
