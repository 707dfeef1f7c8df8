// nann_attn_proj.h -- the reference scorer model (nann_attn.h) with its ITEM-ONLY layers taken out of the traversal.
//
// Of the model's layers, three read nothing but the candidate's row e (model.py:189-233, model_util.py:70-97):
//     q1 = prelu(e Wq1 + bq1),   q_ = q1 Wq2 + bq2,   and the rows of DNN layer 1 that multiply e in x = [a ; e].
// In the split-f16 form (nann_attn_split.h) they are 12 of a pass's 32 weight slices and 320 of its 540 MFMAs per 32
// candidates.  They depend on the (model, index) pair only, so the first search of a pair computes, for every item,
//     T[i] = [ q_(e_i) x 2^4  (256 floats) ;  (e_i W1e) x 2^7 x 2^4  (128 floats) ]          (k_attn_preproject, f32 chains)
// -- a resident f32 [N, 384] table owned by the scorer, 1.5 GB per million items of HBM's 288, the same scheme and
// the same ownership rule as the MLP's pre-projection (nann_mlp3.h) -- and the traversal's scorer starts at the
// attention logits: a lane gathers the four 16-byte pieces of its row of T that hold its 16 units of q_ tile t in the
// C/D layout (the tile arrives where q2's accumulators would have been), splits it and multiplies it with the keys;
// DNN layer 1 seeds its accumulators with b1 + the row's W1e part and runs the attention half only.
// Per 32 candidates: 220 MFMAs, 6 slices of 16 KB per pass (keys 4 x two tiles, W2 x 2); the sequence, W1a and W3 stay in LDS.
// Scores agree with the f32 form within the same 1e-5 every attention test holds (the table is an f32 chain; what is
// split into f16 planes afterwards is what the split form splits at the same place).
#pragma once
#include "nann_attn_split.h"

namespace nann {

constexpr int kAttnProjWidth = 384;  // floats per item: q_ x 2^4 [256] ; (e W1e) x 2^11 [128]
// A fragments that stay in LDS for a whole scoring call instead of travelling once per pass: the user's sequence
// (512 uint4), W1a (4 tiles x 512) and W3 (512) -- 48 KB behind the slice buffers and the vectors; the keys (64 KB)
// and W2 (32 KB) keep streaming
constexpr int kAttnResidentU4 = 512 + 2048 + 512;
constexpr int kAttnResidentBytes = kAttnResidentU4 * 16;

#ifdef NANN_ATTN_SPLIT_TU
// 16 rows per step of a 256-thread workgroup; weights stream from L2 (they are re-read once per 16 rows: 256 KB).
template <int DT>
__global__ __launch_bounds__(256) void k_attn_preproject(AttnParams P, const void* __restrict__ emb, long long n_rows,
                                                         float* __restrict__ proj) {
  constexpr int R = 16;
  __shared__ float es[R][128];   // the rows, f32
  __shared__ float q1[R][128];
  const int tid = threadIdx.x, d = P.d;
  const int j = tid & 127, rh = tid >> 7;  // unit, row half (8 rows each) of the 128-unit layers
  const long long steps = (n_rows + R - 1) / R;
  for (long long s = blockIdx.x; s < steps; s += gridDim.x) {
    const long long r0 = s * R;
    __syncthreads();
    for (int i = tid; i < R * d; i += 256) {
      const int r = i / d, k = i - r * d;
      const long long row = r0 + r < n_rows ? r0 + r : n_rows - 1;
      const uint16_t b = static_cast<const uint16_t*>(emb)[(size_t)row * d + k];
      es[r][k] = DT == DT_F16 ? half_bits_to_float(b) : __uint_as_float((uint32_t)b << 16);
    }
    __syncthreads();
    {  // q1 = prelu(e Wq1 + bq1)  (model_util.py:79)
      float acc[8];
      const float b = P.bq1[j];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = b;
      for (int k = 0; k < d; ++k) {
        const float w = P.wq1[k * 128 + j];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = __fmaf_rn(es[rh * 8 + r][k], w, acc[r]);
      }
      const float al = P.aq[j];
#pragma unroll
      for (int r = 0; r < 8; ++r) q1[rh * 8 + r][j] = prelu(acc[r], al);
    }
    {  // the rows of DNN layer 1 that multiply e: x = [a ; e], W1 rows 64 .. 64 + d  (model.py:211)
      float acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = 0.0f;
      for (int k = 0; k < d; ++k) {
        const float w = P.w1[(size_t)(kAttnE + k) * 128 + j];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = __fmaf_rn(es[rh * 8 + r][k], w, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (r0 + rh * 8 + r < n_rows) proj[(size_t)(r0 + rh * 8 + r) * kAttnProjWidth + 256 + j] = acc[r] * (kAttnWS * kAttnHS);
    }
    __syncthreads();
    {  // q_ = q1 Wq2 + bq2  (model_util.py:80): thread = unit, all 16 rows
      float acc[R];
      const float b = P.bq2[tid];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = b;
      for (int k = 0; k < 128; ++k) {
        const float w = P.wq2[k * 256 + tid];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = __fmaf_rn(q1[r][k], w, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r0 + r < n_rows) proj[(size_t)(r0 + r) * kAttnProjWidth + tid] = acc[r] * kAttnHS;
    }
  }
}
#endif  // NANN_ATTN_SPLIT_TU

#define NANN_MFMA16(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, c_, 0, 0, 0)

// wg_score_attn_proj: as wg_score_attn_split, with `proj` = the (model, index) table instead of the embedding rows.
template <int NT>
__device__ __forceinline__ void wg_score_attn_proj(const AttnParams& P, const uint4* __restrict__ kt,
                                                   const uint4* __restrict__ ua, const float* __restrict__ proj,
                                                   long long n_table_rows, const int32_t* indices, long long n,
                                                   float* slice_f, float* scores) {
  static_assert(NT == 512, "two uint4 per thread per 16 KB slice");
  constexpr int CPP = (NT / 64) * 32;
  // slices per pass: the keys, two tiles per 16 KB slice (one barrier per pair), and W2; the rest is resident
  constexpr int S_W2 = 4, NS = 6;
  constexpr int R_SEQ = 0, R_W1 = 512, R_W3 = 2560;  // uint4 offsets in the resident block
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  uint4* buf = reinterpret_cast<uint4*>(slice_f);  // [2][1024]
  float* pv = slice_f + kAttnSlice;                // the small vectors, in LDS
  uint4* res = reinterpret_cast<uint4*>(slice_f + kAttnSlice + kAttnVecFloats);  // [kAttnResidentU4]
  __syncthreads();
  for (int k = tid; k < PV_COUNT / 4; k += NT)
    reinterpret_cast<float4*>(pv)[k] = reinterpret_cast<const float4*>(P.pvec)[k];
  for (int k = tid; k < kAttnResidentU4; k += NT)
    res[k] = k < R_W1 ? ua[k] : k < R_W3 ? P.pw1a[k - R_W1] : P.pw3[k - R_W3];
  // (visible after the barriers that open the first pass)
  const float att_scale = (1.0f / sqrtf(256.0f)) / (kAttnWS * kAttnHS);  // model_util.py:89-91, and the operand scales

  auto slice_src = [&](int s, int* cnt) -> const uint4* {
    *cnt = 1024;
    if (s < S_W2) return kt + (size_t)s * 1024;                                       // keys for q_ tiles 2s, 2s + 1
    return P.pw2 + (size_t)(s - S_W2) * 1024;
  };
  uint4 pre0, pre1;
  auto fetch = [&](int s) {
    int cnt;
    const uint4* src = slice_src(s, &cnt);
    pre0 = src[min(tid, cnt - 1)];
    pre1 = src[min(tid + NT, cnt - 1)];
  };
  // the slices of a pass form a ring: the last step of a pass brings in slice 0 of the next one (the same keys -- one
  // call scores one user's candidates), so only the first pass waits for its first slice
  static_assert(NS % 2 == 0, "slice s lives in buffer s & 1 in every pass");
  auto step_begin = [&](int s) -> const uint4* {
    fetch(s + 1 < NS ? s + 1 : 0);
    return buf + (s & 1) * 1024;
  };
  auto step_end = [&](int s) {
    uint4* nb = buf + ((s + 1) & 1) * 1024;
    nb[tid] = pre0;
    nb[tid + NT] = pre1;
    __syncthreads();
  };
  // the lane's 16 values of a 32-unit tile of its row: four runs of 4 consecutive floats (load_tile_vec's pattern)
  auto gather_tile = [&](const float* tile, float4 (&x)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) x[rr] = *reinterpret_cast<const float4*>(tile + 8 * rr + 4 * g);
  };
  auto as_tile = [&](const float4 (&x)[4], f32x16& v) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { v[4 * rr] = x[rr].x; v[4 * rr + 1] = x[rr].y; v[4 * rr + 2] = x[rr].z; v[4 * rr + 3] = x[rr].w; }
  };

  auto row_id = [&](long long i) -> long long {
    const long long ic = i < n ? i : n - 1;
    return indices ? (long long)indices[ic] : ic;
  };
  auto row_of = [&](long long rid) -> const float* {
    return proj + ((rid >= 0 && rid < n_table_rows) ? (size_t)rid : 0u) * kAttnProjWidth;
  };
  if (n <= 0) return;
  float4 qa[4], qb[4];  // q_ tiles in flight: two tiles ahead of their use (four: spills, slower)
  fetch(0);
  __syncthreads();  // the caller is done with both buffers
  buf[tid] = pre0;
  buf[tid + NT] = pre1;
  __syncthreads();

  for (long long c0 = 0; c0 < n; c0 += CPP) {
    const long long i = c0 + wave * 32 + cand;
    const float* T = row_of(row_id(i));
    gather_tile(T, qa);
    gather_tile(T + 32, qb);

    f32x16 acc;
    // ---- attention logits, q_ tile by q_ tile: att[l] += sum_{j in tile} q_[j] k_l[j]
    f32x16 att[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) att[p][r] = 0.0f;
    auto keys_tile = [&](const uint4* A, float4 (&x)[4], int refill) {  // refill: the tile x holds next, or < 0
      f16x8 qh[2], ql[2];
      as_tile(x, acc);
      split_tile(acc, qh, ql);
      if (refill >= 0) gather_tile(T + 32 * refill, x);
      f16x8 K[8];
      load_frags(A, lane, K);
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
    };
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {  // tiles 2 tp, 2 tp + 1
      const uint4* A = step_begin(tp);
      keys_tile(A, qa, tp < 3 ? 2 * tp + 2 : -1);
      keys_tile(A + 512, qb, tp < 3 ? 2 * tp + 3 : -1);
      step_end(tp);
    }
    // the row's part of DNN layer 1: in flight under the softmax
    float4 dA[4], dB[4];  // tiles 0 and 1 now, 2 and 3 when these have been used
    gather_tile(T + 256, dA);
    gather_tile(T + 256 + 32, dB);
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
      f16x8 U[8];
      load_frags(res + R_SEQ, lane, U);
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
    }
    // ---- DNN layer 1 on [a ; e] (model.py:211-214): the e rows come from the table
    f16x8 h1h[4][2], h1l[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4* A = res + R_W1 + 512 * m;
      float seed[16];
      load_tile_vec(pv + PV_B1 + 32 * m, g, seed);
      f32x16 dv;
      as_tile((m & 1) ? dB : dA, dv);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = seed[r] + dv[r];
      if (m < 2) gather_tile(T + 256 + 32 * (m + 2), (m & 1) ? dB : dA);
      f16x8 W[8];
      load_frags(A, lane, W);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        acc = NANN_MFMA16(W[kc * 2], ah[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2], al[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2 + 1], ah[kc >> 1][kc & 1], acc);
      }
      float sc[16], sh[16], al_[16];
      load_tile_vec(pv + PV_S1 + 32 * m, g, sc);
      load_tile_vec(pv + PV_T1 + 32 * m, g, sh);
      load_tile_vec(pv + PV_A1 + 32 * m, g, al_);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w = __fmaf_rn(acc[r], sc[r], sh[r]);  // bn(x W + b) x 2^4
        acc[r] = __fmaf_rn(neg_part(w), al_[r], w);  // prelu: w + (alpha - 1) min(w, 0)
      }
      split_tile(acc, h1h[m], h1l[m]);
    }
    // ---- layer 2
    f16x8 h2h[2][2], h2l[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const uint4* A = step_begin(S_W2 + m);
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
      step_end(S_W2 + m);
    }
    // ---- layer 3 and the bias-free output (:218-219)
    float logit = 0.0f;
    {
      const uint4* A = res + R_W3;
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
    }
    logit += __shfl_xor(logit, 32);
    if (g == 0 && i < n) scores[i] = logit;
  }
  __syncthreads();
}

// wg_score_attn_res (round 4): wg_score_attn_proj with EVERYTHING a scoring call reads resident in LDS -- the user's keys
// (64 KB of fragments) in the place of the 16K-slot visited set, which the caller parks in the slot around the call
// (nann_mlp5.h's scheme), W2 (32 KB) in the place of the two slice buffers, the sequence / W1a / W3 fragments and the
// vectors where they were.  No slice ring, no barrier inside a call: a wavefront takes every eighth 32-candidate block
// and runs at its own pace (the proj form met at six barriers per 256 candidates).  Same arithmetic, same scores.
template <int NT>
__device__ __forceinline__ void wg_score_attn_res(const AttnParams& P, const uint4* __restrict__ kt,
                                                  const uint4* __restrict__ ua, const float* __restrict__ proj,
                                                  long long n_table_rows, const int32_t* indices, long long n,
                                                  uint4* keys, float* slice_f, float* scores) {
  static_assert(NT == 512, "two uint4 per thread per 16 KB slice");
  constexpr int NW = NT / 64;
  constexpr int R_SEQ = 0, R_W1 = 512, R_W3 = 2560;  // uint4 offsets in the resident block
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  uint4* w2 = reinterpret_cast<uint4*>(slice_f);  // [2][1024]: the two output tiles of layer 2 (where the proj form's slice buffers are)
  float* pv = slice_f + kAttnSlice;                // the small vectors, in LDS
  uint4* res = reinterpret_cast<uint4*>(slice_f + kAttnSlice + kAttnVecFloats);  // [kAttnResidentU4]
  __syncthreads();
  for (int k = tid; k < PV_COUNT / 4; k += NT)
    reinterpret_cast<float4*>(pv)[k] = reinterpret_cast<const float4*>(P.pvec)[k];
  for (int k = tid; k < kAttnResidentU4; k += NT)
    res[k] = k < R_W1 ? ua[k] : k < R_W3 ? P.pw1a[k - R_W1] : P.pw3[k - R_W3];
  for (int k = tid; k < 4096; k += NT) keys[k] = kt[k];   // q_ tiles 2 s, 2 s + 1 at keys + 1024 s
  for (int k = tid; k < 2048; k += NT) w2[k] = P.pw2[k];
  __syncthreads();
  const float att_scale = (1.0f / sqrtf(256.0f)) / (kAttnWS * kAttnHS);  // model_util.py:89-91, and the operand scales

  // the lane's 16 values of a 32-unit tile of its row: four runs of 4 consecutive floats (load_tile_vec's pattern)
  auto gather_tile = [&](const float* tile, float4 (&x)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) x[rr] = *reinterpret_cast<const float4*>(tile + 8 * rr + 4 * g);
  };
  auto as_tile = [&](const float4 (&x)[4], f32x16& v) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) { v[4 * rr] = x[rr].x; v[4 * rr + 1] = x[rr].y; v[4 * rr + 2] = x[rr].z; v[4 * rr + 3] = x[rr].w; }
  };

  auto row_id = [&](long long i) -> long long {
    const long long ic = i < n ? i : n - 1;
    return indices ? (long long)indices[ic] : ic;
  };
  auto row_of = [&](long long rid) -> const float* {
    return proj + ((rid >= 0 && rid < n_table_rows) ? (size_t)rid : 0u) * kAttnProjWidth;
  };
  if (n <= 0) return;
  float4 qa[4], qb[4];  // q_ tiles in flight: two tiles ahead of their use (four: spills, slower)
  const long long nblk = (n + 31) >> 5;
  for (long long blk = wave; blk < nblk; blk += NW) {
    const long long i = blk * 32 + cand;
    const float* T = row_of(row_id(i));
    gather_tile(T, qa);
    gather_tile(T + 32, qb);

    f32x16 acc;
    // ---- attention logits, q_ tile by q_ tile: att[l] += sum_{j in tile} q_[j] k_l[j]
    f32x16 att[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) att[p][r] = 0.0f;
    auto keys_tile = [&](const uint4* A, float4 (&x)[4], int refill) {  // refill: the tile x holds next, or < 0
      f16x8 qh[2], ql[2];
      as_tile(x, acc);
      split_tile(acc, qh, ql);
      if (refill >= 0) gather_tile(T + 32 * refill, x);
      f16x8 K[8];
      load_frags(A, lane, K);
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
    };
#pragma unroll 1
    for (int tp = 0; tp < 4; ++tp) {  // tiles 2 tp, 2 tp + 1
      const uint4* A = keys + 1024 * tp;
      keys_tile(A, qa, tp < 3 ? 2 * tp + 2 : -1);
      keys_tile(A + 512, qb, tp < 3 ? 2 * tp + 3 : -1);
    }
    // the row's part of DNN layer 1: in flight under the softmax
    float4 dA[4], dB[4];  // tiles 0 and 1 now, 2 and 3 when these have been used
    gather_tile(T + 256, dA);
    gather_tile(T + 256 + 32, dB);
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
      f16x8 U[8];
      load_frags(res + R_SEQ, lane, U);
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
    }
    // ---- DNN layer 1 on [a ; e] (model.py:211-214): the e rows come from the table
    f16x8 h1h[4][2], h1l[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const uint4* A = res + R_W1 + 512 * m;
      float seed[16];
      load_tile_vec(pv + PV_B1 + 32 * m, g, seed);
      f32x16 dv;
      as_tile((m & 1) ? dB : dA, dv);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = seed[r] + dv[r];
      if (m < 2) gather_tile(T + 256 + 32 * (m + 2), (m & 1) ? dB : dA);
      f16x8 W[8];
      load_frags(A, lane, W);
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        acc = NANN_MFMA16(W[kc * 2], ah[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2], al[kc >> 1][kc & 1], acc);
        acc = NANN_MFMA16(W[kc * 2 + 1], ah[kc >> 1][kc & 1], acc);
      }
      float sc[16], sh[16], al_[16];
      load_tile_vec(pv + PV_S1 + 32 * m, g, sc);
      load_tile_vec(pv + PV_T1 + 32 * m, g, sh);
      load_tile_vec(pv + PV_A1 + 32 * m, g, al_);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float w = __fmaf_rn(acc[r], sc[r], sh[r]);  // bn(x W + b) x 2^4
        acc[r] = __fmaf_rn(neg_part(w), al_[r], w);  // prelu: w + (alpha - 1) min(w, 0)
      }
      split_tile(acc, h1h[m], h1l[m]);
    }
    // ---- layer 2
    f16x8 h2h[2][2], h2l[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const uint4* A = w2 + 1024 * m;
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
    }
    // ---- layer 3 and the bias-free output (:218-219)
    float logit = 0.0f;
    {
      const uint4* A = res + R_W3;
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
    }
    logit += __shfl_xor(logit, 32);
    if (g == 0 && i < n) scores[i] = logit;
  }
  __syncthreads();
}

#undef NANN_MFMA16

}  // namespace nann
