// ubench_valu_rate (round 5): issue cost of the vector instructions the traversal's hot loops are made of, per SIMD, with
// 1 / 2 / 4 wavefronts per SIMD.  The L2 scoring loop turned out to be bound by vector-instruction ISSUE; its phase-repeat
// measurement (profiles/rd5ad_*) says ~11.5 cycles per row and CU where a count of 4 cycles per instruction gives 5.3 -- so
// which of its instructions are not 4-cycle ones?  Every instruction is `asm volatile` over 16 independent registers.
// Prints shader cycles per instruction and SIMD (total cycles / (instructions x wavefronts per SIMD)).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu_rate.hip -o tools/_build/ubench_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND, int NT>
__global__ __launch_bounds__(NT) void k_rate(int iters, float* out, long long* ticks) {
  float v[16], w[16];
  uint32_t h[16];
  unsigned long long q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = (float)(threadIdx.x + i) * 1e-3f; w[i] = 1.0f + (float)i;
    h[i] = 0x3c003800u + threadIdx.x + i; q[i] = (unsigned long long)threadIdx.x * 7u + i;
    asm volatile("" : "+v"(v[i]), "+v"(w[i]), "+v"(h[i]), "+v"(q[i]));
  }
  unsigned long long sm = 0x0001000100010001ull;
  asm volatile("" : "+s"(sm));
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define I_FMA(i) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(v[i]) : "v"(w[i]));
#define I_SUB(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "v"(w[i]));
#define I_CVT(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(h[i]));
#define I_MIX(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[i]) : "v"(h[i]), "v"(w[i]));
#define I_MIXH(i) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(v[i]) : "v"(h[i]), "v"(w[i]));
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define I_DPPQ(i) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(w[i]));
#define I_DPPM(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(w[i]));
#define I_DPPB(i) asm volatile("v_or_b32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(h[i]) : "v"(h[(i + 5) & 15]), "v"(h[(i + 9) & 15]));
#define I_CND(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "s"(sm));
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(h[i]) : "v"(h[(i + 5) & 15]), "v"(h[(i + 9) & 15]));
#define I_LSHADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define I_MAD64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(h[i]), "v"(h[(i + 1) & 15]) : "vcc");
#define I_RDLANE(i) { uint32_t s_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s_) : "v"(h[i])); }
#define I_CMP64(i) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(q[i]), "v"(q[(i + 1) & 15]) : "vcc");
#define I_BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h[i]) : "v"(h[(i + 5) & 15]));
#define I_MIN(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(v[i]) : "v"(w[i]));
#define I_CVTPK(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(v[i]), "v"(w[i]));
#define I_MUL24(i) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(h[i]) : "v"(h[(i + 5) & 15]), "v"(h[(i + 9) & 15]));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(h[i]) : "v"(h[(i + 5) & 15]), "v"(h[(i + 9) & 15]));
    if constexpr (KIND == 0) { REP16(I_FMA) REP16(I_FMA) }
    if constexpr (KIND == 1) { REP16(I_SUB) REP16(I_SUB) }
    if constexpr (KIND == 2) { REP16(I_CVT) REP16(I_CVT) }
    if constexpr (KIND == 3) { REP16(I_MIX) REP16(I_MIXH) }
    if constexpr (KIND == 4) { REP16(I_PKADD) REP16(I_PKADD) }
    if constexpr (KIND == 5) { REP16(I_PKFMA) REP16(I_PKFMA) }
    if constexpr (KIND == 6) { REP16(I_DPPQ) REP16(I_DPPQ) }
    if constexpr (KIND == 7) { REP16(I_DPPM) REP16(I_DPPM) }
    if constexpr (KIND == 8) { REP16(I_DPPB) REP16(I_DPPB) }
    if constexpr (KIND == 9) { REP16(I_CND) REP16(I_CND) }
    if constexpr (KIND == 10) { REP16(I_MAD24) REP16(I_MAD24) }
    if constexpr (KIND == 11) { REP16(I_LSHADD64) REP16(I_LSHADD64) }
    if constexpr (KIND == 12) { REP16(I_MAD64) REP16(I_MAD64) }
    if constexpr (KIND == 13) { REP16(I_RDLANE) REP16(I_RDLANE) }
    if constexpr (KIND == 14) { REP16(I_CMP64) REP16(I_CMP64) }
    if constexpr (KIND == 15) { REP16(I_BCNT) REP16(I_BCNT) }
    if constexpr (KIND == 16) { REP16(I_MIN) REP16(I_MIN) }
    if constexpr (KIND == 17) { REP16(I_CVTPK) REP16(I_CVTPK) }
    if constexpr (KIND == 18) { REP16(I_MUL24) REP16(I_MUL24) }
    if constexpr (KIND == 19) { REP16(I_MULLO) REP16(I_MULLO) }
    // the scoring loop's row chunk as it is: 8 fma_mix + 8 fma (dependent chain per row, 2 rows interleaved)
    if constexpr (KIND == 20) {
#define I_ROW(i) asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" \
                              : "=&v"(v[i]), "=&v"(v[(i + 8) & 15]) : "v"(h[i]), "v"(w[i]), "v"(w[(i + 1) & 15]));              \
                 asm volatile("v_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %0, %2, %2, %0" : "+v"(w[(i + 3) & 15]) : "v"(v[i]), "v"(v[(i + 8) & 15]));
      I_ROW(0) I_ROW(1) I_ROW(2) I_ROW(3) I_ROW(4) I_ROW(5) I_ROW(6) I_ROW(7)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc += v[i] + w[i] + (float)h[i] + (float)(uint32_t)q[i];
  out[blockIdx.x * NT + threadIdx.x] = acc;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static double g_ns = 0.0;  // the last run: nanoseconds per instruction and SIMD by HIP events (the cycle counter's unit is not
                           // necessarily the shader clock)
template <int KIND, int NT>
static double run(int iters, float* out, long long* ticks, int blocks) {
  hipLaunchKernelGGL((k_rate<KIND, NT>), dim3(blocks), dim3(NT), 0, 0, iters, out, ticks);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_rate<KIND, NT>), dim3(blocks), dim3(NT), 0, 0, iters, out, ticks);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.0f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> t(blocks);
  hipMemcpy(t.data(), ticks, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (long long x : t) s += (double)x;
  g_ns = (double)ms * 1e6 / ((double)iters * 32) / (NT / 256.0);
  return s / blocks / ((double)iters * 32) / (NT / 256.0);  // counter units per instruction and SIMD
}

int main() {
  const int blocks = 256, iters = 20000;
  float* out; long long* ticks;
  hipMalloc(&out, blocks * 1024 * sizeof(float));
  hipMalloc(&ticks, blocks * sizeof(long long));
  const char* names[] = {"v_fma_f32", "v_sub_f32", "v_cvt_f32_f16", "v_fma_mix_f32 (lo, hi)", "v_pk_add_f32", "v_pk_fma_f32", "v_add_f32_dpp quad_perm",
                         "v_add_f32_dpp row_mirror", "v_or_b32_dpp row_newbcast", "v_cndmask_b32_e64 (sgpr mask)", "v_mad_u32_u24", "v_lshl_add_u64",
                         "v_mad_i64_i32", "v_readlane_b32", "v_cmp_gt_u64", "v_bcnt_u32_b32", "v_min_f32", "v_cvt_pkrtz_f16_f32", "v_mul_u32_u24", "v_mul_lo_u32",
                         "scoring row chunk (16 fma_mix + 16 fma per 32)"};
  printf("%-48s %10s %10s %10s %12s  (readcyclecounter units per instruction and SIMD at 1, 2, 4 wavefronts per SIMD; ns by HIP events at 4)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "ns at 4");
#define ROW(K) { const double a = run<K, 256>(iters, out, ticks, blocks), b = run<K, 512>(iters, out, ticks, blocks), c = run<K, 1024>(iters, out, ticks, blocks); \
                 printf("%-48s %10.2f %10.2f %10.2f %12.3f\n", names[K], a, b, c, g_ns); }
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20)
  return 0;
}
