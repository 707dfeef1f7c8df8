// ubench_walk2.hip -- which part of a walker step costs what (one wavefront, bitmap in LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// FLAGS: 1 = batched (U=8), 2 = dup loop, 4 = store to global, 8 = store to LDS, 16 = ballot/compaction
template <int FLAGS>
__device__ int walk(const int32_t* src, int n, uint32_t* bm, uint32_t n_items, int32_t* gout, int32_t* lout) {
  constexpr int U = (FLAGS & 1) ? 8 : 1;
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull;
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 64 * U) {
    int32_t x[U]; uint32_t pre[U], old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = src[min(c0 + u * 64 + lane, n - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool inr = (c0 + u * 64 + lane) < n && (uint32_t)x[u] < n_items;
      uint32_t* w = bm + (inr ? ((uint32_t)x[u] >> 5) : 0u);
      const uint32_t bit = inr ? (1u << (x[u] & 31)) : 0u;
      pre[u] = *w;
      old[u] = atomicOr(w, bit);
    }
    if (FLAGS & 32) {
      bool keepv[U]; bool freshv[U]; uint64_t dm[U]; uint64_t any = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool inr = (c0 + u * 64 + lane) < n && (uint32_t)x[u] < n_items;
        const uint32_t bit = 1u << (x[u] & 31);
        freshv[u] = inr && !(pre[u] & bit);
        keepv[u] = inr && !(old[u] & bit);
        dm[u] = __ballot(freshv[u] && !keepv[u]);
        any |= dm[u];
      }
      if (any) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          uint64_t dupl = dm[u];
          while (dupl) {
            const int l = __ffsll((unsigned long long)dupl) - 1;
            const int32_t xv = __builtin_amdgcn_readlane(x[u], l);
            const bool mine = freshv[u] && x[u] == xv;
            const uint64_t same = __ballot(mine);
            const int first = __ffsll((unsigned long long)same) - 1;
            if (mine) keepv[u] = (lane == first);
            dupl &= ~same;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t m = __ballot(keepv[u]);
        if (keepv[u]) gout[base + __popcll(m & lt)] = x[u];
        base += __popcll(m);
      }
      continue;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool inr = (c0 + u * 64 + lane) < n && (uint32_t)x[u] < n_items;
      const uint32_t bit = 1u << (x[u] & 31);
      const bool fresh = inr && !(pre[u] & bit);
      bool keep = inr && !(old[u] & bit);
      if (FLAGS & 2) {
        uint64_t dupl = __ballot(fresh && !keep);
        while (dupl) {
          const int l = __ffsll((unsigned long long)dupl) - 1;
          const int32_t xv = __builtin_amdgcn_readlane(x[u], l);
          const bool mine = fresh && x[u] == xv;
          const uint64_t same = __ballot(mine);
          const int first = __ffsll((unsigned long long)same) - 1;
          if (mine) keep = (lane == first);
          dupl &= ~same;
        }
      }
      if (FLAGS & 16) {
        const uint64_t m = __ballot(keep);
        if (FLAGS & 4) { if (keep) gout[base + __popcll(m & lt)] = x[u]; }
        if (FLAGS & 8) { if (keep) lout[(base + __popcll(m & lt)) & 4095] = x[u]; }
        base += __popcll(m);
      } else {
        base += keep ? 1 : 0;
      }
    }
  }
  return base;
}

template <int FLAGS>
__global__ __launch_bounds__(1024) void k(const int32_t* ids, int n, uint32_t n_items, uint32_t bm_words,
                                          int32_t* out, long long* ticks, int* kept_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* bm = reinterpret_cast<uint32_t*>(smem);
  int32_t* stage = reinterpret_cast<int32_t*>(smem + (size_t)bm_words * 4);
  int32_t* lout = stage + 4096;
  for (uint32_t i = threadIdx.x; i < bm_words; i += 1024) bm[i] = (i * 2654435761u) & ((i * 40503u) << 7);
  for (int i = threadIdx.x; i < n; i += 1024) stage[i] = ids[i];
  __syncthreads();
  if ((threadIdx.x >> 6) == 0) {
    const long long t0 = clock64();
    const int base = walk<FLAGS>(stage, n, bm, n_items, out, lout);
    const long long t1 = clock64();
    if (lane_id() == 0) { ticks[0] = t1 - t0; kept_out[0] = base; }
  }
}

int main() {
  const int n = 4096;
  const uint32_t n_items = 1000000, bm_words = 31252;
  std::vector<int32_t> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (int32_t)(((long long)rand() * 7919) % n_items);
  int32_t *d_ids, *d_out; long long* d_t; int* d_k;
  (void)hipMalloc(&d_ids, n * 4); (void)hipMalloc(&d_out, n * 4); (void)hipMalloc(&d_t, 64); (void)hipMalloc(&d_k, 64);
  (void)hipMemcpy(d_ids, h.data(), n * 4, hipMemcpyHostToDevice);
  const size_t lds = (size_t)bm_words * 4 + 2 * 4096 * 4;
  auto run = [&](auto kern, const char* name) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    long long t = 0; int kept = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, 0, d_ids, n, n_items, bm_words, d_out, d_t, d_k);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&kept, d_k, 4, hipMemcpyDeviceToHost);
    }
    printf("%-44s %8lld ticks %7.1f /step kept %d\n", name, t, (double)t / (n / 64), kept);
  };
  run(k<0>, "unbatched, count only");
  run(k<1>, "batched U=8, count only");
  run(k<1 | 16>, "batched, ballot compaction, no store");
  run(k<1 | 16 | 8>, "batched, compaction, LDS store");
  run(k<1 | 16 | 4>, "batched, compaction, global store");
  run(k<1 | 16 | 4 | 2>, "batched, compaction, global store, dup loop");
  run(k<16 | 4 | 2>, "unbatched, compaction, global store, dup loop");
  run(k<1 | 32>, "batched U=8, batch-level dup check, global store");
  return 0;
}
