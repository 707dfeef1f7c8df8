// ubench_lds_atomic: how fast does one CU execute random-address LDS operations -- plain 4-byte reads, CAS with return
// (ds_cmpst_rtn_b32), ds_min_u32 without return -- with 8 or 16 wavefronts issuing 4 or 8 independent operations per
// lane and step (the shape of wg_expand_hash's probe loop)?  Prints shader cycles per wave-step and lane-ops per cycle.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_atomic.hip -o tools/_build/ubench_lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

template <int OP, int PER, int NT, int SLOTS>
__global__ __launch_bounds__(NT) void k_lds(int steps, uint32_t* out, long long* ticks) {
  extern __shared__ uint32_t vis[];
  for (int i = threadIdx.x; i < SLOTS; i += NT) vis[i] = 0xffffffffu;
  __syncthreads();
  uint32_t acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < steps; ++s) {
    uint32_t h[PER], c[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) h[j] = mix((s * PER + j) * NT + threadIdx.x + blockIdx.x * 7919u) & (SLOTS - 1);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (OP == 0) c[j] = vis[h[j]];
      else if (OP == 1) c[j] = atomicCAS(&vis[h[j]], 0xfffffffeu, (uint32_t)s);  // never matches: pure traffic
      else { atomicMin(&vis[h[j]], 0xffffff00u | (uint32_t)j); c[j] = 0; }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) acc += c[j];
    if (OP == 2) __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * NT + threadIdx.x] = acc;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int OP, int PER, int NT, int SLOTS>
static void run(const char* name, uint32_t* out, long long* ticks) {
  const int steps = 2000, grid = 256;
  auto kern = k_lds<OP, PER, NT, SLOTS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SLOTS * 4);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), SLOTS * 4, 0, steps, out, ticks);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), SLOTS * 4, 0, steps, out, ticks);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (auto t : h) mean += (double)t; mean /= 256;
  printf("%-22s %2d waves x %d ops/lane/step, %5d slots: %7.0f cycles per step, %.2f lane-ops per cycle per CU\n", name, NT / 64, PER,
         SLOTS, mean / steps, (double)NT * PER / (mean / steps));
}

int main() {
  uint32_t* out; long long* ticks;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 256 * 8);
  run<0, 8, 512, 16384>("ds_read_b32", out, ticks);
  run<1, 8, 512, 16384>("ds_cmpst_rtn_b32", out, ticks);
  run<2, 8, 512, 16384>("ds_min_u32 (no return)", out, ticks);
  run<0, 4, 1024, 32768>("ds_read_b32", out, ticks);
  run<1, 4, 1024, 32768>("ds_cmpst_rtn_b32", out, ticks);
  run<2, 4, 1024, 32768>("ds_min_u32 (no return)", out, ticks);
  run<1, 1, 1024, 32768>("ds_cmpst_rtn_b32", out, ticks);
  run<1, 8, 64, 16384>("ds_cmpst_rtn_b32", out, ticks);
  run<0, 8, 64, 16384>("ds_read_b32", out, ticks);
  return 0;
}
