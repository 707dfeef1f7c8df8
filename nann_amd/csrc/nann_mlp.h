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
//   layer 1, tile t (32 hidden units):  D1_t[j][c] = u[j] + P[j][c],  P[j][c] = sum_k W1e[k][32t+j] * e[c][k]
//       A operand = W1e^T tile (lane l: hidden unit l&31, k-slot l>>5), streamed through LDS
//       B operand = the candidate's embedding (lane l: candidate l&31, its half row in registers)
//       P is a chain from 0 (ORDER_E) -- the very bits nann_search's pre-projected table holds (nann_mlp3.h,
//       nann_mlp5.h) -- and the per-query part u = b1 + W1q^T q, hoisted out, is added ONCE (round 4; rounds 1-3
//       seeded the accumulators with u)
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
  // split-f16 form (NANN_MLP_SPLIT_F16): the item half of W1 and all of W2, times 2^7, as f16 (hi, lo) planes
  // packed on the host in MFMA A-fragment order (pack_split_weights in nann_hip.hip)
  const uint4* p1;  // [h1/32 tiles][d/16 chunks][hi, lo][64 lanes] x 8 halves
  const uint4* p2;  // [h1/32 tiles][2 chunks][h2/32 tiles][hi, lo][64 lanes] x 8 halves
  // exact form with W2 resident in LDS (nann_mlp5.h): f32 A fragments of v_mfma_f32_32x32x2_f32, four chain steps per 16 bytes
  const float4* p2x;  // [h1/32 tiles][h2/32 tiles][4][64 lanes] x 4 floats
};

constexpr int kMlpNT = 512;  // 8 wavefronts: 2 per SIMD -> 256 VGPRs each for the accumulators
constexpr int kMlpSlice = 4096;  // floats per weight slice in LDS (128 rows x 32 columns)

struct MlpVectors {
  float u[512];       // b1 + W1q^T q of the current query
  float alpha1[512];
  float b2[256];
  float alpha2[256];
  float w3[256];
};
struct MlpScratch {
  float slice[kMlpSlice];
  MlpVectors v;
};
static_assert(sizeof(MlpScratch) <= kPhaseScratch, "phase scratch too small for the MLP");
// split-f16 form: two slice buffers (slice s+1 lands in one while slice s feeds the MFMAs from the other)
struct MlpSplitScratch {
  uint4 buf[4][1024];  // d <= 128: [tile parity][layer-1 slice, layer-2 slice]; d = 256: two of them, one slice each
  MlpVectors v;
};

__device__ __forceinline__ float prelu(float x, float a) {
  const float pos = x > 0.0f ? x : 0.0f;
  const float neg = x < 0.0f ? x : 0.0f;
  return pos + a * neg;
}

// Per query: u[j] = b1[j] + sum_k q[k] * W1[k][j] (k ascending, fmaf).  Thread j < h1 returns u[j] and keeps
// it in a register for the whole traversal of the query (h1 <= NT); the loads of a run of 8 rows are issued
// together, the chain stays in k order.
template <int NT>
__device__ __forceinline__ float wg_mlp_query_u(const MlpParams& P, const float* qv) {
  const int j = local_tid();
  if (j >= P.h1) return 0.0f;
  float acc = P.b1[j];
  const float* w = P.w1 + j;
  int k = 0;
  for (; k + 8 <= P.d; k += 8) {
    float wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = w[(size_t)(k + i) * P.h1];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __fmaf_rn(qv[k + i], wv[i], acc);
  }
  for (; k < P.d; ++k) acc = __fmaf_rn(qv[k], w[(size_t)k * P.h1], acc);
  return acc;
}

// Per scoring call (the phase scratch was reused since the last one): u and the small vectors into LDS.
// All NT threads; ends with a barrier.
// u_scale / b2_scale / alpha1_scale: the split-f16 form keeps these pre-multiplied by its accumulator scales
// (exact powers of two)
template <int NT>
__device__ __forceinline__ void wg_mlp_stage_setup(const MlpParams& P, float u, MlpVectors* V, float u_scale = 1.0f,
                                                   float b2_scale = 1.0f, float alpha1_scale = 1.0f) {
  const int tid = local_tid();
  static_assert(NT >= 256, "one hidden unit per thread");
  if (tid < P.h1) {
    V->u[tid] = u * u_scale;
    V->alpha1[tid] = P.alpha1[tid] * alpha1_scale;
  }
  for (int m = tid; m < P.h2; m += NT) {
    V->b2[m] = P.b2[m] * b2_scale;
    V->alpha2[m] = P.alpha2[m];
    V->w3[m] = P.w3[m];
  }
  __syncthreads();
}

template <int NT>
__device__ __forceinline__ void wg_mlp_query_setup(const MlpParams& P, const float* qv, MlpVectors* V,
                                                   float u_scale = 1.0f, float b2_scale = 1.0f,
                                                   float alpha1_scale = 1.0f) {
  wg_mlp_stage_setup<NT>(P, wg_mlp_query_u<NT>(P, qv), V, u_scale, b2_scale, alpha1_scale);
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
  const MlpVectors* V = &S->v;
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
      for (int r = 0; r < 16; ++r) acc2[mt][r] = V->b2[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * slot];
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
      if (ks == 0) {  // the item part P = W1e^T e is its own chain from 0 (round 4: the same bits as the pre-projected table)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
      }
      if (ks < KS1) {
#pragma unroll
        for (int kkl = 0; kkl < KK1; ++kkl) {
          const float a = S->slice[slot * 2048 + kkl * 32 + cand];
          const float b = packed_elem<DT>(ev, ks * 64 + kkl);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc1, 0, 0, 0);
        }
        if (ks == KS1 - 1) {  // tile complete: a1 = u + P (the per-query part, one add), PReLU in place
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int unit = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * slot;
            acc1[r] = prelu(V->u[unit] + acc1[r], V->alpha1[unit]);
          }
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
        part = __fmaf_rn(prelu(acc2[mt][r], V->alpha2[m]), V->w3[m], part);
      }
    const float other = __shfl_xor(part, 32);
    const float p0 = slot == 0 ? part : other, p1 = slot == 0 ? other : part;
    if (slot == 0 && i < n) scores[i] = p0 + p1;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Split-f16 form of the same scorer (NANN_MLP_SPLIT_F16): the f32-input MFMA runs at 1/16 of the
// 16-bit rate, and north_star only asks for scores within 1e-5 of fp32.  Every f32 operand v is
// carried as two f16 values, v = hi + lo with hi = f16(v), lo = f16(v - hi) -- 22 significant bits --
// and products go to v_mfma_f32_32x32x16_f16 with f32 accumulation:
//   layer 1   W1e . e    : e is an f16 table row already (bf16 rows convert) -> Whi.e + Wlo.e         2 MFMAs per 16 k
//   layer 2   W2  . h1   : Whi.hh + Whi.hl + Wlo.hh    (Wlo.hl is below 2^-22 of the product: dropped)  3 MFMAs per 16 k
// For lo to be a NORMAL f16 (full 11 bits) the operands are pre-scaled by powers of two, which is
// exact: weights by 2^7 (host side, at packing), activations h1 by 2^4; the accumulators then carry
// 2^7 (layer 1) and 2^11 (layer 2) and are scaled back once.  One accumulator per output, as in the
// f32 form.  Per 32 candidates a wavefront issues 8 x (2 d/16 + 24) MFMAs of 32 cycles instead of
// 8 x (d/2 + 128) of 64: 6.4x less matrix time at d = 128.  Scores differ from the fp32 chain by
// ~1e-6 relative (tests hold them to 1e-5), so this form is checked with tolerances and tie-aware
// ids; the f32 form above stays the bit-exact one.
//
// The register layout trick carries over: in the 32x32 C/D layout lane (c, g) holds hidden units
// (r & 3) + 8 (r >> 2) + 4 g of tile t; registers 0..7 / 8..15 of the finished layer-1 tile ARE the
// B fragments (8 k-values per lane) of two 16-deep layer-2 steps, provided W2's A fragments are
// packed with the same k order -- which the host packing does.  (The hardware pairs A's and B's
// per-lane k slots; any k order shared by both sides gives the same dot product.)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float kSplitWScale = 128.0f;  // weights x 2^7 (|w| <= 511 or the scorer is refused)
constexpr float kSplitHScale = 16.0f;   // hidden activations x 2^4 (|h| <= 4094)

// min(x, 0) as ONE v_min_f32 (fminf() costs a second instruction: the compiler canonicalises its operand first; the
// accumulators hold no signalling NaNs to quiet)
__device__ __forceinline__ float neg_part(float x) {
  float r;
  asm("v_min_f32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

__device__ __forceinline__ f16x8 as_f16x8(const uint4& v) {
  union { uint4 u; f16x8 h; } c;
  c.u = v;
  return c.h;
}

template <int DT>
__device__ __forceinline__ f16x8 row_chunk_f16(const uint4& v) {  // 8 table elements -> f16
  if constexpr (DT == DT_F16) {
    return as_f16x8(v);
  } else {  // bf16 -> f32 (exact) -> f16 (exact above 2^-14 in magnitude, ~1e-8 absolute below)
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    f16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      r[2 * i] = (_Float16)bf16_bits_to_float(w[i] & 0xffffu);
      r[2 * i + 1] = (_Float16)bf16_bits_to_float(w[i] >> 16);
    }
    return r;
  }
}

// All A fragments of a step are read from LDS in one burst before its first MFMA (and the compiler is kept from
// sinking them back next to their uses): one exposed LDS latency per step instead of one per chunk.
template <int N>
__device__ __forceinline__ void load_frags(const uint4* A, int lane, f16x8 (&f)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) f[k] = as_f16x8(A[k * 64 + lane]);
  __builtin_amdgcn_sched_barrier(0);
}

// Where a 256-row pass spends its 27 us at d = 128 (tools/mlp_rate.py on the stand-alone scorer, timing builds with
// parts compiled out, profiles/r2_mlp_split_experiments.md): the 320 MFMAs per wavefront alone take 17-19 us =
// 26.5-30 ns per MFMA per SIMD, where a bare loop of this MFMA holds 18.4 ns on this chip (tools/ubench_mfma.hip:
// one per 32 shader cycles, chained or not, at the ~1.7 GHz the chip clocks under it -- 1.8 PFLOP/s, not the
// 2.5 of 2.4 GHz); everything else alone -- weight slices L2 -> LDS -> registers, barriers, PReLU / split
// arithmetic -- takes 5 us, and the two overlap poorly.  Halving the LDS reads, dropping the barriers or the vector
// arithmetic, a second slice buffer with one barrier per slice, burst reads of a step's fragments, rolling or
// unrolling the tile loop, giving neighbouring MFMAs different accumulators: each moved the pass by under 7 %.
// PMC (profiles/r2_mlp_split_pmc_head.txt): matrix pipe 46 % busy; a wavefront spends 50 % of its cycles stalled
// at issue behind its partner's MFMAs or its own chain, 29 % parked at waits and barriers, 21 % issuing -- the two
// wavefronts of a SIMD run the same phase at the same time, so nothing fills the pipe while both split a tile.
template <int D, int H1T, int H2T, int DT, int NT>
__device__ __forceinline__ void wg_score_mlp_split(const MlpParams& P, const void* __restrict__ table,
                                                   uint32_t n_table_rows, const int32_t* ids, int n,
                                                   MlpSplitScratch* S, float* scores) {
  const MlpVectors* V = &S->v;
  static_assert(DT == DT_F16 || DT == DT_BF16, "split form: 16-bit table rows");
  static_assert(H2T == 4, "a layer-2 slice is [2 chunks][4 tiles][2 planes] KB");
  constexpr int NWV = NT / 64;
  constexpr int CPP = NWV * 32;            // candidates per pass
  constexpr int KC = D / 16;               // 16-deep chunks of layer 1
  constexpr int KCS = KC < 8 ? KC : 8;     // chunks per layer-1 slice (16 KB = 8 chunks x 2 planes x 1 KB)
  constexpr int KS1 = (KC + 7) / 8;        // layer-1 slices per hidden tile
  constexpr int SPT = KS1 + 1;             // + the tile's layer-2 slice
  constexpr int NSLICE = H1T * SPT;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  uint4* buf = &S->buf[0][0];  // [2][1024] x 16 B

  auto slice_src = [&](int s, int f) -> const uint4* {  // uint4 number f (0..1023) of slice s
    const int t = s / SPT, ks = s % SPT;
    if (ks < KS1) return P.p1 + ((size_t)(t * KC + ks * 8) * 2) * 64 + min(f, KCS * 128 - 1);
    return P.p2 + (size_t)t * 1024 + f;
  };
  auto row_of = [&](int i0) -> size_t {
    const int ic = min(i0 + wave * 32 + cand, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    return rid < n_table_rows ? rid : 0u;
  };
  // B fragments of layer 1: chunk kc of this lane = elements 16 kc + 8 g .. + 8 of the row (one 16-B load)
  uint4 ev[KC];
  auto load_row = [&](size_t row) {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const char*>(table) + row * D * 2) + g;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) ev[kc] = src[2 * kc];
  };
  if (n > 0) load_row(row_of(0));
  // Slice pipeline.  A GROUP of G slices is handed over per barrier: group gi sits in buffer gi & 1 while the next
  // group (of this pass or the first of the next) travels L2 -> registers, then registers -> the other buffer.
  // G = 2 = a whole hidden tile when its two slices fit twice (d <= 128): with ONE barrier per TILE the two
  // wavefronts of a SIMD no longer meet after every slice, and the one that lost the matrix pipe during layer 1
  // runs its MFMAs under the other's PReLU / split arithmetic instead of beside it.  (Timing build: with a barrier
  // per slice the older wavefront of a SIMD issues its chain at 32 cycles per MFMA, then waits 31 % of the pass at
  // barriers for the younger one, whose chain could only start when the first had released the pipe.)
  constexpr int G = SPT == 2 ? 2 : 1;
  constexpr int NG = NSLICE / G;
  static_assert(NSLICE % G == 0 && NG % 2 == 0, "group 0 of the next pass lands in buffer 0 again");
  uint4 pre0, pre1, pre2, pre3;  // (scalars: an array here ends up in scratch memory)
  auto fetch_group = [&](int gi) {
    pre0 = *slice_src(gi * G, tid);
    pre1 = *slice_src(gi * G, tid + NT);
    if constexpr (G == 2) {
      pre2 = *slice_src(gi * G + 1, tid);
      pre3 = *slice_src(gi * G + 1, tid + NT);
    }
  };
  auto write_group = [&](int b) {
    buf[(b * G) * 1024 + tid] = pre0;
    buf[(b * G) * 1024 + tid + NT] = pre1;
    if constexpr (G == 2) {
      buf[(b * G + 1) * 1024 + tid] = pre2;
      buf[(b * G + 1) * 1024 + tid + NT] = pre3;
    }
  };
  fetch_group(0);
  __syncthreads();  // the caller is done with the scratch (the vectors were staged before, behind a barrier)
  write_group(0);
  __syncthreads();

  for (int i0 = 0; i0 < n; i0 += CPP) {
    const int i = i0 + wave * 32 + cand;
    const size_t next_row = i0 + CPP < n ? row_of(i0 + CPP) : 0;
    f32x16 a2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        a2[mt][4 * rr] = v.x; a2[mt][4 * rr + 1] = v.y; a2[mt][4 * rr + 2] = v.z; a2[mt][4 * rr + 3] = v.w;
      }
    f32x16 a1;
    f16x8 bh[2], bl[2];
    // the hidden-tile loop is ROLLED: same speed as the unrolled form, an eighth of the code, and k_search keeps its
    // registers (d = 256: 468 -> 68 B/lane of spills)
#pragma unroll 1
    for (int t = 0; t < H1T; ++t)
#pragma unroll
    for (int ks = 0; ks < SPT; ++ks) {
      const int s = t * SPT + ks;
      const int gi = s / G, gj = s % G;
      const uint4* A = buf + ((gi & 1) * G + gj) * 1024;
      if (gj == 0) fetch_group(gi + 1 < NG ? gi + 1 : 0);  // next group from L2 while this one feeds the MFMAs
      if (ks == 0) {  // the per-query part seeds the tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float4 v = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
          a1[4 * rr] = v.x; a1[4 * rr + 1] = v.y; a1[4 * rr + 2] = v.z; a1[4 * rr + 3] = v.w;
        }
      }
      if (ks < KS1) {
        f16x8 W[2 * KCS];
        load_frags(A, lane, W);
#pragma unroll
        for (int kl = 0; kl < KCS; ++kl) {
          const int kc = ks * 8 + kl;
          if (kc < KC) {
            const f16x8 b = row_chunk_f16<DT>(ev[kc < KC ? kc : 0]);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[kl * 2], b, a1, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[kl * 2 + 1], b, a1, 0, 0, 0);
          }
        }
        if (t == H1T - 1 && ks == KS1 - 1 && i0 + CPP < n) load_row(next_row);  // the rows are consumed: fetch the next pass's
        if (ks == KS1 - 1) {  // tile complete: PReLU, scale (x 2^-7 x 2^4, folded into alpha1 and the positive
          // branch: PReLU commutes with a positive factor) and split into the layer-2 B fragments.  hi is cut
          // with round-toward-zero (one packed conversion per pair); lo = h - hi is exact in f32 and rounds
          // into 11 more bits.
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
            const float2 al = *reinterpret_cast<const float2*>(&V->alpha1[32 * t + 8 * (r >> 2) + 4 * g + (r & 3)]);
            const float x0 = a1[r], x1 = a1[r + 1];
            const float h0 = x0 * (x0 > 0.0f ? kSplitHScale / kSplitWScale : al.x);
            const float h1 = x1 * (x1 > 0.0f ? kSplitHScale / kSplitWScale : al.y);
            const h2_t hi = __builtin_amdgcn_cvt_pkrtz(h0, h1);
            const h2_t lo = __builtin_amdgcn_cvt_pkrtz(h0 - (float)hi[0], h1 - (float)hi[1]);
            bh[r >> 3][r & 7] = (_Float16)hi[0]; bh[r >> 3][(r & 7) + 1] = (_Float16)hi[1];
            bl[r >> 3][r & 7] = (_Float16)lo[0]; bl[r >> 3][(r & 7) + 1] = (_Float16)lo[1];
          }
        }
      } else {
        f16x8 W[4 * H2T];
        load_frags(A, lane, W);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int mt = 0; mt < H2T; ++mt) {
            a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[(q * H2T + mt) * 2], bh[q], a2[mt], 0, 0, 0);
            a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[(q * H2T + mt) * 2], bl[q], a2[mt], 0, 0, 0);
            a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[(q * H2T + mt) * 2 + 1], bh[q], a2[mt], 0, 0, 0);
          }
      }
      if (gj == G - 1) {  // hand the next group over
        write_group((gi + 1) & 1);
        __syncthreads();
      }
    }
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g;
        part = __fmaf_rn(prelu(a2[mt][r], V->alpha2[m]), V->w3[m], part);  // still x 2^11: scaled back once below
      }
    const float other = __shfl_xor(part, 32);
    const float p0 = g == 0 ? part : other, p1 = g == 0 ? other : part;
    if (g == 0 && i < n) scores[i] = (p0 + p1) * (1.0f / (kSplitWScale * kSplitHScale));
  }
  __syncthreads();
}

}  // namespace nann
