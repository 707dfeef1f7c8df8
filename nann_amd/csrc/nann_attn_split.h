// nann_attn_split.h -- the reference scorer model (nann_attn.h) on the 16-bit matrix instruction.
//
// Same model, same mapping idea as nann_attn_kernels.h (one wavefront = 32 candidates, every layer's output
// stays in the 32x32 C/D register layout and is the next layer's B operand in place), but with the operand
// split of nann_mlp.h's split form: every f32 operand v is carried as hi + lo f16 (22 significant bits),
// weights and keys pre-scaled by 2^7, activations by 2^4 (exact), products on v_mfma_f32_32x32x16_f16 with
// f32 accumulation:
//     exact f16 operand (table rows e, user sequence u):  W_hi.x + W_lo.x            2 MFMAs per 16 k
//     f32 activation x:                                   W_hi.x_hi + W_hi.x_lo + W_lo.x_hi   3 MFMAs
// 540 MFMAs of 32 cycles per 32 candidates (d = 128) instead of 1632 of 64.  Logits agree with the f32 form
// to ~1e-6 relative; tests hold them to 1e-5 like every attention-scorer test.
//
// A fragments (weights; per user: keys and the sequence) are packed ahead of time in MFMA lane order, hi and
// lo planes, one <= 16 KB slice per (layer, output tile): nann_hip.hip packs the weights at scorer creation
// (pack_attn_frags), k_attn_prepare_split the per-user side.  "cd order" of a k index: the C/D register
// order of the tile that produced it -- chunk q, lane group g, element i <-> unit (i&3) + 16q + 8(i>>2) + 4g.
//
// Staging: two 16 KB LDS buffers; slice s+1 travels L2 -> registers -> LDS while slice s feeds the MFMAs;
// one barrier per slice.  The pre-scaled small vectors sit in LDS behind the buffers.  NANN_ATTN_TIMING builds
// stamp the shader clock per section (tools/attn_rate.py prints them).
#pragma once
#include "nann_attn.h"
#include "nann_mlp.h"

namespace nann {

constexpr float kAttnWS = 128.0f;  // weights, keys x 2^7
constexpr float kAttnHS = 16.0f;   // activations x 2^4

// offsets (floats) into AttnParams::pvec, the pre-scaled small vectors of the split form
enum : int {
  PV_BQ1 = 0,            // bq1 * WS                        [128]
  PV_AQ = 128,           // aq * HS / WS                    [128]
  PV_BQ2 = 256,          // bq2 * WS * HS                   [256]
  PV_B1 = 512,           // b1 * WS * HS                    [128]
  PV_S1 = 640,           // bn_scale1 / WS  (bn output x 2^4) [128]
  PV_T1 = 768,           // bn_shift1 * HS                  [128]
  PV_A1 = 896,           // alpha1 - 1  (prelu: w + A min(w, 0)) [128]
  PV_B2 = 1024, PV_S2 = 1088, PV_T2 = 1152, PV_A2 = 1216,  // [64] each, scaled like layer 1's
  PV_B3 = 1280, PV_S3 = 1312, PV_T3 = 1344, PV_A3 = 1376,  // [32] each; A3 = alpha3 (unscaled)
  PV_W4 = 1408,          // w4                              [32]
  PV_COUNT = 1440
};

// uint4 offsets of the per-user A fragments inside the kt / upad buffers of nann_attn_prepare
//   kt  : [q_ tile t (8)][pos tile p (2)][chunk q (2)][plane (2)][64 lanes]     = 4096 uint4 (64 KB)
//   upad: [unit tile m (2)][pos tile p (2)][chunk q (2)][64 lanes]              =  512 uint4 ( 8 KB)

__device__ __forceinline__ int cd_unit(int q, int g, int i) { return (i & 3) + 16 * q + 8 * (i >> 2) + 4 * g; }

#ifdef NANN_ATTN_SPLIT_TU
// per user: k_l exactly as k_attn_prepare (f32 fmaf chains), then packed.  256 threads per user.
__global__ __launch_bounds__(256) void k_attn_prepare_split(AttnParams P, const uint16_t* __restrict__ user_seq_f16,
                                                            uint16_t* __restrict__ kt, uint16_t* __restrict__ ua) {
  __shared__ float u[kAttnLP * kAttnE];      // 16 KB
  __shared__ float k1[kAttnLP * 128];        // 32 KB
  __shared__ uint16_t ubits[kAttnLP * kAttnE];  // 8 KB
  const int tid = threadIdx.x;
  const size_t user = blockIdx.x;
  user_seq_f16 += user * (size_t)P.L * kAttnE;
  kt += user * (size_t)256 * kAttnLP * 2;    // the f32 [256][64] buffer, as halves
  ua += user * (size_t)kAttnLP * kAttnE * 2;  // the f32 [64][64] buffer, as halves
  for (int i = tid; i < kAttnLP * kAttnE; i += 256) {
    const int l = i / kAttnE;
    const uint16_t b = l < P.L ? user_seq_f16[i] : (uint16_t)0;
    ubits[i] = b;
    u[i] = half_bits_to_float(b);
  }
  __syncthreads();
  for (int i = tid; i < P.L * 128; i += 256) {  // model_util.py:84
    const int l = i >> 7, j = i & 127;
    float acc = P.bk1[j];
    for (int k = 0; k < kAttnE; ++k) acc = __fmaf_rn(u[l * kAttnE + k], P.wk1[k * 128 + j], acc);
    k1[i] = prelu(acc, P.ak[j]);
  }
  __syncthreads();
  // keys: A rows = positions, k = the units of q_ tile t in cd order; hi / lo planes, x 2^7
  for (int o = tid; o < 8 * 2 * 2 * 64 * 8; o += 256) {
    const int i = o & 7, lane = (o >> 3) & 63, q = (o >> 9) & 1, p = (o >> 10) & 1, t = o >> 11;
    const int l = 32 * p + (lane & 31), j = 32 * t + cd_unit(q, lane >> 5, i);
    float acc = 0.0f;
    if (l < P.L) {
      acc = P.bk2[j];
      for (int k = 0; k < 128; ++k) acc = __fmaf_rn(k1[l * 128 + k], P.wk2[k * 256 + j], acc);  // :85
    }
    const float v = acc * kAttnWS;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const size_t base = ((size_t)(((t * 2 + p) * 2 + q) * 2) * 64 + lane) * 8 + i;
    union { _Float16 h; uint16_t b; } ch, cl;
    ch.h = hi; cl.h = lo;
    kt[base] = ch.b;
    kt[base + 64 * 8] = cl.b;
  }
  // the sequence: A rows = the 64 embedding units, k = positions in cd order of the attention tiles; exact f16
  for (int o = tid; o < 2 * 2 * 2 * 64 * 8; o += 256) {
    const int i = o & 7, lane = (o >> 3) & 63, q = (o >> 9) & 1, p = (o >> 10) & 1, m = o >> 11;
    const int l = 32 * p + cd_unit(q, lane >> 5, i), unit = 32 * m + (lane & 31);
    ua[o] = ubits[l * kAttnE + unit];  // (zero beyond L)
  }
}
#endif  // NANN_ATTN_SPLIT_TU

// hi / lo split of 16 f32 values (x already carries the activation scale) into the two 8-wide B fragments of a finished
// tile, 1.5 instructions per value: hi cut toward zero by one packed conversion per pair, lo = f16(x - hi) by ONE
// v_fma_mix{lo,hi}_f16 per value, which converts its f16 operand itself (nann_mlp2.h prelu_split_pair; the first form,
// cvt + sub + cvt, cost ~3 per value and twice the registers: 123 -> 8 spilled registers in the traversal kernels).
__device__ __forceinline__ void split_tile(const f32x16& x, f16x8 (&h)[2], f16x8 (&l)[2]) {
  typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float x0 = x[8 * q + 2 * k], x1 = x[8 * q + 2 * k + 1];
      const h2_t hv = __builtin_amdgcn_cvt_pkrtz(x0, x1);
      hw[k] = __builtin_bit_cast(uint32_t, hv);
      uint32_t lo;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hw[k]), "v"(x0));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hw[k]), "v"(x1));
      lw[k] = lo;
    }
    uint4 hh, ll;
    hh.x = hw[0]; hh.y = hw[1]; hh.z = hw[2]; hh.w = hw[3];
    ll.x = lw[0]; ll.y = lw[1]; ll.z = lw[2]; ll.w = lw[3];
    h[q] = as_f16x8(hh);
    l[q] = as_f16x8(ll);
  }
}

// exp(x) for the softmax's x <= 0 in six instructions (expf() is ~10): 2^(x log2 e) with the product carried in two
// floats -- t = x * L, e = the multiply's rounding error plus x * (log2 e - L) -- and 2^(t + e) = 2^t (1 + e ln 2);
// v_exp_f32 is good to ~1 ulp, the result to ~2 ulp (the logits are held to 1e-5)
__device__ __forceinline__ float exp_nonpos(float x) {
  x = fmaxf(x, -128.0f);  // the padding positions carry -inf (inf - inf below would be NaN); 2^-184 flushes to 0 like exp(-inf)
  constexpr float L = 1.44269502162933349609375f;       // float(log2 e)
  constexpr float L_lo = 1.92596299112661746e-08f;      // log2 e - L
  const float t = x * L;
  const float e = __builtin_fmaf(x, L_lo, __builtin_fmaf(x, L, -t));
  const float r = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(r * 0.693147182464599609375f, e, r);
}

// the lane's 16 rows of a 32-unit vector tile: four runs of 4 consecutive floats
__device__ __forceinline__ void load_tile_vec(const float* v, int g, float (&out)[16]) {
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const float4 x = *reinterpret_cast<const float4*>(v + 8 * rr + 4 * g);
    out[4 * rr] = x.x; out[4 * rr + 1] = x.y; out[4 * rr + 2] = x.z; out[4 * rr + 3] = x.w;
  }
}

#define NANN_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)

// wg_score_attn_split: as wg_score_attn.  kt / ua = the packed per-user fragments (k_attn_prepare_split);
// `slice` = kAttnSlice + kAttnVecFloats floats of LDS (two 16 KB buffers + the small vectors).
template <int D, int DT, int NT>
__device__ __forceinline__ void wg_score_attn_split(const AttnParams& P, const uint4* __restrict__ kt,
                                                    const uint4* __restrict__ ua, const void* table,
                                                    long long n_table_rows, const int32_t* indices, long long n,
                                                    float* slice_f, float* scores) {
  static_assert(D == 64 || D == 128, "item embedding dim");
  static_assert(DT == DT_F16 || DT == DT_BF16, "item rows are f16 or bf16");
  static_assert(NT == 512, "two uint4 per thread per 16 KB slice");
  constexpr int KC = D / 16;                 // 16-deep chunks of an item row
  constexpr int CPP = (NT / 64) * 32;
  constexpr int NS = 4 + 16 + 1 + 8 + 2 + 1;  // slices per pass
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  uint4* buf = reinterpret_cast<uint4*>(slice_f);  // [2][1024]
  float* pv = slice_f + kAttnSlice;                // the small vectors, in LDS: a step's seeds must not wait on L2
  static_assert(PV_COUNT <= kAttnVecFloats, "vector block");
  __syncthreads();
  for (int k = tid; k < PV_COUNT / 4; k += NT)
    reinterpret_cast<float4*>(pv)[k] = reinterpret_cast<const float4*>(P.pvec)[k];
  // (visible after the barriers that open the first pass)
  const float att_scale = (1.0f / sqrtf(256.0f)) / (kAttnWS * kAttnHS);  // model_util.py:89-91, and the operand scales

  // slice s of a pass -> (global source, number of uint4)
  auto slice_src = [&](int s, int* cnt) -> const uint4* {
    if (s < 4) { *cnt = KC * 128; return P.pq1 + (size_t)s * KC * 128; }                  // q1 tile s
    s -= 4;
    if (s < 16) {
      const int t = s >> 1;
      if ((s & 1) == 0) { *cnt = 1024; return P.pq2 + (size_t)t * 1024; }                 // Wq2 tile t
      *cnt = 512; return kt + (size_t)t * 512;                                            // keys for q_ tile t
    }
    s -= 16;
    if (s < 1) { *cnt = 512; return ua; }                                                 // the sequence
    s -= 1;
    if (s < 8) {
      const int m = s >> 1;
      if ((s & 1) == 0) { *cnt = 512; return P.pw1a + (size_t)m * 512; }                  // W1 rows of a, tile m
      *cnt = KC * 128; return P.pw1e + (size_t)m * KC * 128;                              // W1 rows of e, tile m
    }
    s -= 8;
    if (s < 2) { *cnt = 1024; return P.pw2 + (size_t)s * 1024; }
    *cnt = 512; return P.pw3;
  };
  uint4 pre0, pre1;
  auto fetch = [&](int s) {
    int cnt;
    const uint4* src = slice_src(s, &cnt);
    pre0 = src[min(tid, cnt - 1)];
    pre1 = src[min(tid + NT, cnt - 1)];
  };
  auto row_of = [&](long long c0) -> size_t {
    const long long i = c0 + wave * 32 + cand;
    const long long ic = i < n ? i : n - 1;
    const long long rid = indices ? (long long)indices[ic] : ic;
    return (rid >= 0 && rid < n_table_rows) ? (size_t)rid : 0u;
  };
  uint4 ev[KC];  // B fragments of the item row: chunk kc = elements 16 kc + 8 g .. + 8
  auto load_row = [&](size_t row) {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const char*>(table) + row * D * 2) + g;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) ev[kc] = src[2 * kc];
  };

  // one step of the slice pipeline: slice s is in buf[s & 1]; the next one travels L2 -> registers while this
  // one feeds the MFMAs, then registers -> the other buffer, one barrier
  auto step_begin = [&](int s) -> const uint4* {
    if (s + 1 < NS) fetch(s + 1);
    return buf + (s & 1) * 1024;
  };
  auto step_end = [&](int s) {
    if (s + 1 < NS) {
      uint4* nb = buf + ((s + 1) & 1) * 1024;
      nb[tid] = pre0;
      nb[tid + NT] = pre1;
      __syncthreads();
    }
  };

  for (long long c0 = 0; c0 < n; c0 += CPP) {
    const long long i = c0 + wave * 32 + cand;
    const size_t row = row_of(c0);
    load_row(row);
    fetch(0);
    __syncthreads();  // the previous pass (or the caller) is done with both buffers
    buf[tid] = pre0;
    buf[tid + NT] = pre1;
    __syncthreads();

    f32x16 acc;
#ifdef NANN_ATTN_TIMING  // timing build: shader-clock stamps per section, written over the first scores by thread 0
    long long tk[6];
    tk[0] = __builtin_readcyclecounter();
#endif
    // ---- q1 = prelu(e Wq1 + bq1), x 2^4, split: the B fragments of the next layer
    f16x8 q1h[4][2], q1l[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4* A = step_begin(m);
      float seed[16];
      load_tile_vec(pv + PV_BQ1 + 32 * m, g, seed);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = seed[r];
      f16x8 W[2 * KC];
      load_frags(A, lane, W);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const f16x8 b = row_chunk_f16<DT>(ev[kc]);
        acc = NANN_MFMA16(W[kc * 2 + 0], b, acc);
        acc = NANN_MFMA16(W[kc * 2 + 1], b, acc);
      }
      float al_[16];
      load_tile_vec(pv + PV_AQ + 32 * m, g, al_);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = acc[r] * (acc[r] > 0.0f ? kAttnHS / kAttnWS : al_[r]);
      split_tile(acc, q1h[m], q1l[m]);
      step_end(m);
    }
#ifdef NANN_ATTN_TIMING
    tk[1] = __builtin_readcyclecounter();
#endif
    // ---- attention logits, q_ tile by q_ tile: att[l] += sum_{j in tile} q_[j] k_l[j].  A ROLLED loop: measured
    // 15 % faster than the unrolled form (61.6 -> 52.1 us per pass; spills 724 -> 236 B/lane, half the code);
    // rolling the q1 and DNN-1 tile loops as well (fragments stored through uniform branches) LOSES 20 %.
    f32x16 att[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) att[p][r] = 0.0f;
#pragma unroll 1
    for (int t = 0; t < 8; ++t) {
      f16x8 qh[2], ql[2];
      {  // q_ tile t = q1 Wq2 + bq2
        const uint4* A = step_begin(4 + 2 * t);
        float seed[16];
        load_tile_vec(pv + PV_BQ2 + 32 * t, g, seed);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = seed[r];
        f16x8 W[16];
        load_frags(A, lane, W);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          acc = NANN_MFMA16(W[kc * 2], q1h[kc >> 1][kc & 1], acc);
          acc = NANN_MFMA16(W[kc * 2], q1l[kc >> 1][kc & 1], acc);
          acc = NANN_MFMA16(W[kc * 2 + 1], q1h[kc >> 1][kc & 1], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= 1.0f / kAttnWS;  // q_ x 2^4
        split_tile(acc, qh, ql);
        step_end(4 + 2 * t);
      }
      {
        const uint4* A = step_begin(5 + 2 * t);
        f16x8 K[8];
        load_frags(A, lane, K);
        // product-major, the two position tiles alternating
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) att[p] = NANN_MFMA16(K[(p * 2 + q) * 2], qh[q], att[p]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) att[p] = NANN_MFMA16(K[(p * 2 + q) * 2], ql[q], att[p]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int p = 0; p < 2; ++p) att[p] = NANN_MFMA16(K[(p * 2 + q) * 2 + 1], qh[q], att[p]);
        step_end(5 + 2 * t);
      }
    }
#ifdef NANN_ATTN_TIMING
    tk[2] = __builtin_readcyclecounter();
#endif
    // ---- softmax over the L positions (:93); positions >= L are padding of the layout
    f16x8 ph[2][2], pl[2][2];  // softmax weights x 2^4, split
    {
      float mx = -INFINITY;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int l = 32 * p + cd_unit(r >> 3, g, r & 7);
          att[p][r] = l < P.L ? att[p][r] * att_scale : -INFINITY;
          mx = fmaxf(mx, att[p][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.0f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          att[p][r] = exp_nonpos(att[p][r] - mx);
          sum += att[p][r];
        }
      sum += __shfl_xor(sum, 32);
      const float inv = kAttnHS / sum;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int r = 0; r < 16; ++r) att[p][r] *= inv;
        split_tile(att[p], ph[p], pl[p]);
      }
    }
    // ---- a = sum_l p_l u_l (:95, model.py:204-206): u is exact f16
    f16x8 ah[2][2], al[2][2];  // a x 2^4, split
    {
      const uint4* A = step_begin(20);
      f16x8 U[8];
      load_frags(A, lane, U);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            acc = NANN_MFMA16(U[(m * 2 + p) * 2 + q], ph[p][q], acc);
            acc = NANN_MFMA16(U[(m * 2 + p) * 2 + q], pl[p][q], acc);
          }
        split_tile(acc, ah[m], al[m]);
      }
      load_row(row);  // the item row again for DNN layer 1 (L1 / L2 hit): not held through the attention loop
      step_end(20);
    }
#ifdef NANN_ATTN_TIMING
    tk[3] = __builtin_readcyclecounter();
#endif
    // ---- DNN layer 1 on [a ; e] (model.py:211-214)
    f16x8 h1h[4][2], h1l[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      {
        const uint4* A = step_begin(21 + 2 * m);
        float seed[16];
        load_tile_vec(pv + PV_B1 + 32 * m, g, seed);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = seed[r];
        f16x8 W[8];
        load_frags(A, lane, W);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          acc = NANN_MFMA16(W[kc * 2], ah[kc >> 1][kc & 1], acc);
          acc = NANN_MFMA16(W[kc * 2], al[kc >> 1][kc & 1], acc);
          acc = NANN_MFMA16(W[kc * 2 + 1], ah[kc >> 1][kc & 1], acc);
        }
        step_end(21 + 2 * m);
      }
      {
        const uint4* A = step_begin(22 + 2 * m);
        f32x16 acc_e;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_e[r] = 0.0f;
        f16x8 W[2 * KC];
        load_frags(A, lane, W);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const f16x8 b = row_chunk_f16<DT>(ev[kc]);
          acc_e = NANN_MFMA16(W[kc * 2], b, acc_e);
          acc_e = NANN_MFMA16(W[kc * 2 + 1], b, acc_e);
        }
        float sc[16], sh[16], al_[16];
        load_tile_vec(pv + PV_S1 + 32 * m, g, sc);
        load_tile_vec(pv + PV_T1 + 32 * m, g, sh);
        load_tile_vec(pv + PV_A1 + 32 * m, g, al_);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float w = __fmaf_rn(__fmaf_rn(acc_e[r], kAttnHS, acc[r]), sc[r], sh[r]);  // bn(x W + b) x 2^4
          acc[r] = __fmaf_rn(neg_part(w), al_[r], w);  // prelu: w + (alpha - 1) min(w, 0)
        }
        split_tile(acc, h1h[m], h1l[m]);
        step_end(22 + 2 * m);
      }
    }
#ifdef NANN_ATTN_TIMING
    tk[4] = __builtin_readcyclecounter();
#endif
    // ---- layer 2
    f16x8 h2h[2][2], h2l[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const uint4* A = step_begin(29 + m);
      float seed[16];
      load_tile_vec(pv + PV_B2 + 32 * m, g, seed);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = seed[r];
      f16x8 W[16];
      load_frags(A, lane, W);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        acc = NANN_MFMA16(W[kc * 2], h1h[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2], h1l[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2 + 1], h1h[kc >> 1][kc & 1], acc);
      }
      float sc[16], sh[16], al_[16];
      load_tile_vec(pv + PV_S2 + 32 * m, g, sc);
      load_tile_vec(pv + PV_T2 + 32 * m, g, sh);
      load_tile_vec(pv + PV_A2 + 32 * m, g, al_);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w = __fmaf_rn(acc[r], sc[r], sh[r]);
        acc[r] = __fmaf_rn(neg_part(w), al_[r], w);
      }
      split_tile(acc, h2h[m], h2l[m]);
      step_end(29 + m);
    }
    // ---- layer 3 and the bias-free output (:218-219)
    float logit = 0.0f;
    {
      const uint4* A = step_begin(31);
      float seed[16];
      load_tile_vec(pv + PV_B3, g, seed);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = seed[r];
      f16x8 W[8];
      load_frags(A, lane, W);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        acc = NANN_MFMA16(W[kc * 2], h2h[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2], h2l[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2 + 1], h2h[kc >> 1][kc & 1], acc);
      }
      float sc[16], sh[16], al_[16], w4[16];
      load_tile_vec(pv + PV_S3, g, sc);
      load_tile_vec(pv + PV_T3, g, sh);
      load_tile_vec(pv + PV_A3, g, al_);
      load_tile_vec(pv + PV_W4, g, w4);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = __fmaf_rn(acc[r], sc[r], sh[r]);
        logit = __fmaf_rn(v > 0.0f ? v : al_[r] * v, w4[r], logit);
      }
      step_end(31);
    }
    logit += __shfl_xor(logit, 32);
    if (g == 0 && i < n) scores[i] = logit;
#ifdef NANN_ATTN_TIMING
    tk[5] = __builtin_readcyclecounter();
    __syncthreads();
    if (tid == 0 && c0 + CPP >= n)
      for (int k = 0; k < 5; ++k) scores[k] = (float)(tk[k + 1] - tk[k]);
#endif
  }
  __syncthreads();
}

#undef NANN_MFMA16

}  // namespace nann
