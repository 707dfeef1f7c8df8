// nann_mlp.h -- the candidate-scoring MLP on the matrix cores (BASELINE configs 3-5).
//
// Scorer behind the BlazeXlaOp contract (UO/blaze_op/blaze_xla_kernel.cc:24-33,
// blaze_xla_predictor.cc:360-459: rows scored independently, f32 logits):
//     x = [q ; e] in R^{2d}  ->  H1 -> PReLU -> H2 -> PReLU -> 1 (no bias)
// (PReLU = max(0,x) + alpha*min(0,x), model_util.py:9-11; last layer
// bias-free, model.py:218-219).  north_star asks for scores within 1e-5 of fp32, so the
// contraction runs on the f32-input MFMA v_mfma_f32_32x32x2_f32, which is bit-for-bit a
// k-ordered fmaf chain; the oracle (oracle/nann_oracle.c, ORDER_E / ORDER_H / ORDER_O)
// walks k in the same order, so scores are bit-identical, not merely close.
//
// Mapping (one wavefront = 32 candidates):
//   layer 1, tile t (32 hidden units):  D1_t[j][c] = u[j] + sum_k W1e[k][32t+j] * e[c][k]
//       A operand = W1e^T tile (lane l: hidden unit l&31, k-slot l>>5), streamed through LDS
//       B operand = the candidate's embedding (lane l: candidate l&31, its half row in registers)
//       the per-query part u = b1 + W1q^T q is hoisted out and seeds the accumulators
//   layer 2, tile m: D2_m[o][c] = b2[o] + sum_j W2[j][32m+o] * h1[c][j]
//       B operand = the layer-1 accumulators THEMSELVES: in the 32x32 C/D layout lane l
//       holds, for candidate l&31, exactly the hidden units its k-slot needs, so h1 never
//       leaves the register file (no LDS transpose, no HBM round trip)
//   output: per-lane fma chain over its 64 outputs with w3, then the two k-slots are added.
// Weights are staged per 16 KB slice (layer 1: 128 k-rows x 32 columns; layer 2: 32 k-rows x
// 128 columns) into LDS by all waves, register-double-buffered so the L2 fetch of slice s+1
// hides under the MFMAs of slice s.
#pragma once
#include "nann_device.h"

namespace nann {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpParams {  // device pointers
  const float* w1;  // [2d, h1]
  const float* b1;
  const float* alpha1;
  const float* w2;  // [h1, h2]
  const float* b2;
  const float* alpha2;
  const float* w3;
  int d, h1, h2;
};

constexpr int kMlpNT = 512;  // 8 wavefronts: 2 per SIMD -> 256 VGPRs each for the accumulators
constexpr int kMlpSlice = 4096;  // floats per weight slice in LDS (128 rows x 32 columns)

struct MlpScratch {
  float slice[kMlpSlice];
  float u[512];       // b1 + W1q^T q of the current query
  float alpha1[512];
  float b2[256];
  float alpha2[256];
  float w3[256];
};
static_assert(sizeof(MlpScratch) <= kPhaseScratch, "phase scratch too small for the MLP");

__device__ __forceinline__ float prelu(float x, float a) {
  const float pos = x > 0.0f ? x : 0.0f;
  const float neg = x < 0.0f ? x : 0.0f;
  return pos + a * neg;
}

// Per query: u[j] = b1[j] + sum_k q[k] * W1[k][j] (k ascending, fmaf) and the small vectors
// into LDS.  All NT threads; ends with a barrier.
template <int NT>
__device__ __forceinline__ void wg_mlp_query_setup(const MlpParams& P, const float* qv, MlpScratch* S) {
  const int tid = local_tid();
  for (int j = tid; j < P.h1; j += NT) {
    float acc = P.b1[j];
    for (int k = 0; k < P.d; ++k) acc = __fmaf_rn(qv[k], P.w1[(size_t)k * P.h1 + j], acc);
    S->u[j] = acc;
    S->alpha1[j] = P.alpha1[j];
  }
  for (int m = tid; m < P.h2; m += NT) {
    S->b2[m] = P.b2[m];
    S->alpha2[m] = P.alpha2[m];
    S->w3[m] = P.w3[m];
  }
  __syncthreads();
}

template <int DT>
__device__ __forceinline__ float packed_elem(const uint4* r, int k) {  // k is a compile-time constant after unrolling
  if constexpr (DT == DT_F32) {
    const uint4 v = r[k >> 2];
    const uint32_t w = (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
    return __uint_as_float(w);
  } else {
    const uint4 v = r[k >> 3];
    const int wi = (k >> 1) & 3;
    const uint32_t w = wi == 0 ? v.x : wi == 1 ? v.y : wi == 2 ? v.z : v.w;
    const uint32_t h = (k & 1) ? (w >> 16) : (w & 0xffffu);
    return DT == DT_F16 ? half_bits_to_float(h) : bf16_bits_to_float(h);
  }
}

// wg_score_mlp: scores[i] for candidates ids[i], i < n (ids == nullptr: row i).
// D = embedding dim, H1T = h1/32, H2T = h2/32 (compile time: accumulators live in
// registers).  All NT threads (NT/64 wavefronts x 32 candidates per pass).
// wg_mlp_query_setup must have run for this query.
// Rows outside [0, n_table_rows) are read as row 0 (the caller reports them).
//
// Schedule: hidden tile t of layer 1 is finished (KS1 weight slices), PReLU'd, and at once
// contracted into ALL layer-2 tiles (one more slice: rows 32t..32t+31 of W2, contiguous)
// before tile t+1 starts, so only one layer-1 tile (16 registers) is live next to the H2T
// layer-2 accumulators.  Every layer-2 output still sees its k rows in ascending tile / row
// order, i.e. the same fmaf chain as contracting after all of layer 1 (ORDER_H).
template <int D, int H1T, int H2T, int DT, int NT>
__device__ __forceinline__ void wg_score_mlp(const MlpParams& P, const void* __restrict__ table,
                                             uint32_t n_table_rows, const int32_t* ids, int n,
                                             MlpScratch* S, float* scores) {
  constexpr int NWV = NT / 64;
  constexpr int CPP = NWV * 32;                       // candidates per pass
  constexpr int HALF = D / 2;                         // elements per k-slot
  constexpr int EB = (DT == DT_F32) ? 4 : 2;          // bytes per element
  constexpr int NV = HALF * EB / 16;                  // uint4 per lane for its half row
  constexpr int KS1 = (HALF + 63) / 64;               // layer-1 slices per tile (64 kk each)
  constexpr int KK1 = HALF < 64 ? HALF : 64;          // kk per layer-1 slice
  constexpr int SPT = KS1 + 1;                        // slices per hidden tile: layer 1, then its layer-2 rows
  constexpr int NSLICE = H1T * SPT;
  static_assert(H2T * 32 * 32 == kMlpSlice, "a layer-2 slice is 32 rows of W2 (h2 = 128)");
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, slot = lane >> 5;
  const int h1 = H1T * 32, h2 = H2T * 32;

  // slice s -> global source of float4 number f (0..1023)
  auto slice_src = [&](int s, int f) -> const float4* {
    const int t = s / SPT, ks = s % SPT;
    if (ks < KS1) {  // layer 1: 128 rows x 8 float4 = rows {kk0+kkl} of slot 0 then of slot 1, columns of tile t
      const int row = f >> 3, c4 = f & 7;
      const int sl = row >> 6, kkl = row & 63;  // LDS row = slot*64 + kk_local
      const int kk = min(ks * 64 + kkl, HALF - 1);
      const size_t grow = (size_t)D + (size_t)sl * HALF + kk;
      return reinterpret_cast<const float4*>(P.w1 + grow * h1 + t * 32) + c4;
    }
    // layer 2: rows [32t, 32t+32) of W2, all h2 columns: one contiguous 16 KB block
    return reinterpret_cast<const float4*>(P.w2 + (size_t)t * 32 * h2) + f;
  };

  for (int i0 = 0; i0 < n; i0 += CPP) {
    const int i = i0 + wave * 32 + cand;
    const int ic = min(i, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    const size_t row = rid < n_table_rows ? rid : 0u;
    // this lane's half of the candidate row
    uint4 ev[NV];
    {
      const uint4* src = reinterpret_cast<const uint4*>(static_cast<const char*>(table) +
                                                        (row * D + (size_t)slot * HALF) * EB);
#pragma unroll
      for (int v = 0; v < NV; ++v) ev[v] = src[v];
    }
    f32x16 acc1;
    f32x16 acc2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mt][r] = S->b2[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * slot];
    float4 pre0 = *slice_src(0, tid), pre1 = *slice_src(0, tid + NT);
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
      const int t = s / SPT, ks = s % SPT;
      __syncthreads();  // every wave is done with the previous slice
      reinterpret_cast<float4*>(S->slice)[tid] = pre0;
      reinterpret_cast<float4*>(S->slice)[tid + NT] = pre1;
      __syncthreads();
      if (s + 1 < NSLICE) {  // next slice from L2 while this one feeds the MFMAs
        pre0 = *slice_src(s + 1, tid);
        pre1 = *slice_src(s + 1, tid + NT);
      }
      if (ks == 0) {  // the per-query part seeds the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = S->u[32 * t + (r & 3) + 8 * (r >> 2) + 4 * slot];
      }
      if (ks < KS1) {
#pragma unroll
        for (int kkl = 0; kkl < KK1; ++kkl) {
          const float a = S->slice[slot * 2048 + kkl * 32 + cand];
          const float b = packed_elem<DT>(ev, ks * 64 + kkl);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
        if (ks == KS1 - 1) {  // tile complete: PReLU in place
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc1[r] = prelu(acc1[r], S->alpha1[32 * t + (r & 3) + 8 * (r >> 2) + 4 * slot]);
        }
      } else {
        // the accumulators of tile t ARE the B operand: lane (cand, slot) holds hidden units
        // 32t + (r&3) + 8(r>>2) + 4 slot, exactly the k rows its k-slot contributes
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int krow = (r & 3) + 8 * (r >> 2) + 4 * slot;  // row of this slice
#pragma unroll
          for (int mt = 0; mt < H2T; ++mt) {
            const float a = S->slice[krow * (H2T * 32) + 32 * mt + cand];
            acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, acc1[r], acc2[mt], 0, 0, 0);
          }
        }
      }
    }
    // PReLU of layer 2 and the bias-free output layer: per-lane chain over its 64 outputs
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * slot;
        part = __fmaf_rn(prelu(acc2[mt][r], S->alpha2[m]), S->w3[m], part);
      }
    const float other = __shfl_xor(part, 32);
    const float p0 = slot == 0 ? part : other, p1 = slot == 0 ? other : part;
    if (slot == 0 && i < n) scores[i] = p0 + p1;
  }
  __syncthreads();
}

}  // namespace nann
