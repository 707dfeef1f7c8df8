// ubench_mlp2: where does a 256-row pass of the split-f16 MLP scorer's second mapping (nann_mlp2.h) spend its time?
// The stand-alone scorer over random rows / random weights, in timing variants with parts compiled out (VAR bits of
// wg_score_mlp_split2), each: wall us per pass per CU, shader ticks per pass (s_memtime of wave 0), tick rate.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=off tools/ubench_mlp2.hip -o tools/_build/ubench_mlp2
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rejected/nann_mlp3_streamed_layer2.h"

using namespace nann;

template <int D, int VAR>
__global__ __launch_bounds__(kMlp2NT, 1) void k_var(MlpParams P, const void* table, uint32_t n_rows, const int32_t* ids,
                                                   int per_block, const float* qv, float* scores, long long* ticks) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(Mlp2Scratch<D>)];
  Mlp2Scratch<D>* S = reinterpret_cast<Mlp2Scratch<D>*>(scratch);
  wg_mlp2_stage_setup<kMlp2NT>(P, wg_mlp_query_u<kMlp2NT>(P, qv), &S->v);
  if (VAR & 2) {  // no staging: the tiles' fragments land once
    for (int i = threadIdx.x; i < 2 * Mlp2Scratch<D>::kTile; i += kMlp2NT) (&S->buf[0][0])[i] = P.p1[i % Mlp2Scratch<D>::kL1];
    __syncthreads();
  }
  const long long t0 = __builtin_readcyclecounter();
  wg_score_mlp_split2<D, DT_F16, VAR>(P, table, n_rows, ids + (size_t)blockIdx.x * per_block, per_block, S,
                                      scores + (size_t)blockIdx.x * per_block);
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// the pre-projected form (nann_mlp3.h): rows of the f32 [rows, 256] table; `row_mask` bounds the rows touched (a small
// table stays in L2: instruction-bound time; the full one comes from HBM)
__global__ __launch_bounds__(kMlp2NT, 1) void k_proj(MlpParams P, const float* proj, uint32_t n_rows, const int32_t* ids,
                                                    int per_block, const float* qv, float* scores, long long* ticks) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(Mlp3Scratch)];
  Mlp3Scratch* S = reinterpret_cast<Mlp3Scratch*>(scratch);
  wg_mlp2_stage_setup<kMlp2NT>(P, wg_mlp_query_u<kMlp2NT>(P, qv), &S->v);
  const long long t0 = __builtin_readcyclecounter();
  wg_score_mlp_proj(P, proj, n_rows, ids + (size_t)blockIdx.x * per_block, per_block, S, scores + (size_t)blockIdx.x * per_block);
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static uint16_t f2h(float f) {
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
static float gauss() { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; return s - 6.0f; }

struct Dev {
  MlpParams P;
  void* table; int32_t* ids; float* q; float* scores; long long* ticks;
};

template <int D, int VAR>
static void run(const char* name, const Dev& d, int passes) {
  const int blocks = 256, per_block = passes * 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_var<D, VAR>), dim3(blocks), dim3(kMlp2NT), 0, 0, d.P, d.table, 1u << 20, d.ids, per_block, d.q, d.scores, d.ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  long long h[256]; hipMemcpy(h, d.ticks, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)h[i]; mean /= blocks;
  printf("d=%d VAR=%2d %-58s %.2f us/pass/CU  %.0f ticks/pass  (%.1f ticks per MFMA slot)  tick rate %.2f GHz\n", D, VAR, name,
         best * 1e3 / passes, mean / passes, mean / passes / (D == 128 ? 640.0 : 512.0), mean / (best * 1e6));
}

int main() {
  srand(11);
  Dev d;
  const int D = 128;
  const size_t n_tab = 1u << 20;
  std::vector<uint16_t> tab(n_tab * D);
  for (auto& v : tab) v = f2h(0.3f * gauss());
  hipMalloc(&d.table, tab.size() * 2); hipMemcpy(d.table, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
  const int passes = 48, n = 256 * passes * 256;
  std::vector<int32_t> ids(n);
  for (auto& v : ids) v = (int32_t)(((unsigned)rand() * 32768u + (unsigned)rand()) % n_tab);
  hipMalloc(&d.ids, (size_t)n * 4); hipMemcpy(d.ids, ids.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  hipMalloc(&d.scores, (size_t)n * 4); hipMalloc(&d.ticks, 256 * 8);
  // weights: f32 block [w1 2d x 256 | b1 | alpha1 | w2 256 x 128 | b2 | alpha2 | w3] + packed f16 planes (random normals x 2^7 scale)
  const size_t n_w1 = 2 * (size_t)D * 256, n_w2 = 256 * 128;
  std::vector<float> w(n_w1 + 256 + 256 + n_w2 + 128 + 128 + 128);
  for (auto& v : w) v = 0.08f * gauss();
  for (size_t i = n_w1 + 256; i < n_w1 + 512; ++i) w[i] = 0.25f;
  float* dw; hipMalloc(&dw, w.size() * 4); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  const size_t n_p = (size_t)8 * (D / 16) * 2 * 64 * 8 + (size_t)8 * 2 * 4 * 2 * 64 * 8;
  std::vector<uint16_t> planes(n_p);
  for (size_t i = 0; i < n_p; ++i) planes[i] = f2h(((i / 512) & 1) ? 0.004f * gauss() : 10.0f * gauss());  // hi planes ~ w x 2^7, lo planes small
  uint4* dp; hipMalloc(&dp, n_p * 2); hipMemcpy(dp, planes.data(), n_p * 2, hipMemcpyHostToDevice);
  d.P = MlpParams{};
  d.P.w1 = dw; d.P.b1 = dw + n_w1; d.P.alpha1 = d.P.b1 + 256; d.P.w2 = d.P.alpha1 + 256; d.P.b2 = d.P.w2 + n_w2;
  d.P.alpha2 = d.P.b2 + 128; d.P.w3 = d.P.alpha2 + 128; d.P.d = D; d.P.h1 = 256; d.P.h2 = 128;
  d.P.p1 = dp; d.P.p2 = dp + (size_t)8 * (D / 16) * 2 * 64;
  std::vector<float> q(D);
  for (auto& v : q) v = 0.3f * gauss();
  hipMalloc(&d.q, D * 4); hipMemcpy(d.q, q.data(), D * 4, hipMemcpyHostToDevice);
  {  // pre-projected form: 1M x 256 f32 table (1 GB), random rows; then rows folded into 4096 (L2-resident)
    float* proj; hipMalloc(&proj, (size_t)n_tab * 256 * 4);
    hipMemset(proj, 0, (size_t)n_tab * 256 * 4);
    int32_t* ids_small; hipMalloc(&ids_small, (size_t)n * 4);
    std::vector<int32_t> sm(ids); for (auto& v : sm) v &= 4095;
    hipMemcpy(ids_small, sm.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float best = 1e30f;
      for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_proj, dim3(256), dim3(kMlp2NT), 0, 0, d.P, proj, (uint32_t)n_tab, which ? ids_small : d.ids, passes * 256, d.q, d.scores, d.ticks);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
      }
      long long h[256]; hipMemcpy(h, d.ticks, sizeof(h), hipMemcpyDeviceToHost);
      double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
      printf("pre-projected form, %-44s %.2f us/pass/CU  %.0f ticks/pass  tick rate %.2f GHz\n", which ? "rows from a 4 MB window (L2)" : "rows from the 1 GB table (HBM)",
             best * 1e3 / passes, mean / passes, mean / (best * 1e6));
    }
  }
  run<128, 0>("full", d, passes);
  run<128, 1>("no PReLU / split arithmetic", d, passes);
  run<128, 2>("no weight staging (no fetch / LDS store / barrier)", d, passes);
  run<128, 3>("no split, no staging", d, passes);
  run<128, 6>("no staging, fragments read once", d, passes);
  run<128, 7>("no split, no staging, fragments read once", d, passes);
  run<128, 15>("MFMAs + row loads only", d, passes);
  run<128, 8>("no output layer", d, passes);
  return 0;
}
