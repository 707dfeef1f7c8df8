// nann_mlp5.h -- the MLP scorer of the traversal with ALL OF LAYER 2 RESIDENT IN LDS (round 4).
//
// What bounded the pre-projected form of round 3 (nann_mlp3.h, wg_score_mlp_proj1): the matrix pipe was
// 36 % busy.  Per 256-row pass a SIMD spent 12.3 k cycles issuing MFMAs, ~5 k on vector instructions, ~2 k on LDS /
// memory instructions and ~9 k STALLED: the layer-2 weight slices travelled L2 -> registers -> LDS once per pass
// (128 KB per 256 rows), handed over behind four workgroup barriers, so the eight wavefronts met four times per pass
// and the two wavefronts of a SIMD ran the same phase at the same time -- both splitting operands, then both queueing
// for the matrix pipe.
//
// Here W2 stays in LDS for the whole scoring call: 128 KB (hi + lo f16 planes x 2^7 in MFMA A-fragment order for the
// split-f16 form; f32 in A-fragment order for the exact form).  It does not fit NEXT to the visited set, so it takes
// the set's place: a query's set is only read and written by the expand phase, hence the workgroup parks the 64 KB
// set in its slot's HBM scratch (L2-resident) before the scoring calls that need it afterwards -- rounds 2 and 3; the
// set is empty before round 0, cleared after round 1 and dead after round 4 -- loads W2 over [set | phase scratch],
// scores, and brings the set back: ~256 KB of L2 traffic per parked call, ~2 us of a ~130 us call, against the
// 128 KB per 256 ROWS the streamed form moved.  With the weights resident the scoring loop has NO barrier: a
// wavefront takes every eighth 32-row block and runs tile after tile at its own pace, so the two wavefronts of a SIMD
// drift apart and one's operand split runs under the other's MFMAs (MI355X_MICROARCH.md, "Two waves per SIMD": the
// matrix pipe and the VALU are separate pipes, the partner's VALU issues under a wavefront's MFMAs).
//
// Two scorers share the skeleton:
//   wg_score_mlp_res    split-f16 (NANN_MLP_SPLIT_F16): nann_mlp3.h's arithmetic, 192 x v_mfma_f32_32x32x16_f16 per
//                       32 rows; scores within 1e-5 of the fp32 chain.
//   wg_score_mlp_xres   exact (NANN_MLP_EXACT_F32) ON THE PRE-PROJECTED TABLE: the canonical order of layer 1 is
//                       a1 = u + P with P the ORDER_E fmaf chain from 0 that k_mlp_preproject stores (oracle:
//                       score_mlp_row), so the traversal runs layer 2 only: 512 x v_mfma_f32_32x32x2_f32 per 32 rows
//                       instead of 1024, bit-identical to the oracle.
#pragma once
#include "nann_mlp3.h"

namespace nann {

constexpr int kMlpResW2Bytes = 131072;                                // W2 resident: 8 hidden tiles x 16 KB
constexpr int kMlpResW2Vec = kMlpResW2Bytes / 16;                     // uint4
constexpr int kMlpResBytes = kMlpResW2Bytes + (int)sizeof(Mlp2Vectors);  // + the per-query vectors behind it
static_assert(sizeof(Mlp2Vectors) % 256 == 0, "vectors keep the regions behind them aligned");

// Vectors of the exact form: the same struct, unscaled (u, alpha1, b2, alpha2, w3).
template <int NT>
__device__ __forceinline__ void wg_mlp_xres_vectors(const MlpParams& P, float u, Mlp2Vectors* V) {
  const int tid = local_tid();
  static_assert(NT >= 256, "one hidden unit per thread");
  if (tid < 256) {
    V->u[tid] = u;
    V->beta1[tid] = P.alpha1[tid];
  }
  if (tid < 128) {
    V->b2[tid] = P.b2[tid];
    V->beta2[tid] = P.alpha2[tid];
    V->w3[tid] = P.w3[tid];
  }
}
template <int NT>
__device__ __forceinline__ void wg_mlp_res_vectors(const MlpParams& P, float u, Mlp2Vectors* V) {
  const int tid = local_tid();
  if (tid < 256) {
    V->u[tid] = u * kSplit2Scale;
    V->beta1[tid] = P.alpha1[tid] - 1.0f;
  }
  if (tid < 128) {
    V->b2[tid] = P.b2[tid] * (kSplit2Scale * kSplit2Scale);
    V->beta2[tid] = P.alpha2[tid] - 1.0f;
    V->w3[tid] = P.w3[tid];
  }
}

// Park `park_vec` uint4 of LDS at `lds` (the visited set) in global memory and load the resident weights over
// lds[0 .. kMlpResW2Vec).  A thread parks exactly the addresses it then overwrites, in program order, so no barrier
// sits between the two; the caller's barrier before this call covers the earlier phases, the one after it the loads.
template <int NT>
__device__ __forceinline__ void wg_mlp_res_enter(uint4* lds, const uint4* __restrict__ w2, uint4* park, int park_vec) {
  const int tid = local_tid();
  static_assert(kMlpResW2Vec % (NT * 8) == 0, "two batches of eight 16-byte loads per thread");
  if (park != nullptr)
    for (int i = tid; i < park_vec; i += NT) park[i] = lds[i];
#pragma unroll
  for (int b = 0; b < kMlpResW2Vec / (NT * 8); ++b) {
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w2[(b * 8 + j) * NT + tid];
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[(b * 8 + j) * NT + tid] = v[j];
  }
}
template <int NT>
__device__ __forceinline__ void wg_mlp_res_leave(uint4* lds, const uint4* park, int park_vec) {
  if (park == nullptr) return;
  for (int i = local_tid(); i < park_vec; i += NT) lds[i] = park[i];
}

// ---------------------------------------------------------------------------------------------------------------
// split-f16, W2 resident.  W2 = the scorer's p2 planes ([t][q][m][hi, lo][lane] uint4, nann_hip.hip pack_split_weights)
// in LDS; V staged by wg_mlp_res_vectors.  No barrier inside; every wavefront returns on its own.
template <int NT>
__device__ __forceinline__ void wg_score_mlp_res(const float* __restrict__ proj, uint32_t n_table_rows,
                                                 const int32_t* ids, int n, const uint4* W2, const Mlp2Vectors* V,
                                                 float* scores) {
  constexpr int H1T = 8, H2T = 4, NW = NT / 64;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  const int nblk = (n + 31) >> 5;
  if (wave >= nblk) return;
  auto row_ptr = [&](int i) -> const float* {
    const int ic = min(i, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    return proj + (size_t)(rid < n_table_rows ? rid : 0u) * kMlpProjWidth + 4 * g;
  };
  // this lane's 16 pre-activations of tile t of a row: four 16-byte pieces, units 32 t + 8 rr + 4 g + 0..3
  auto load_tile = [&](const float* row, int t, float4 (&p)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) p[rr] = *reinterpret_cast<const float4*>(row + 32 * t + 8 * rr);
  };
  const float* row = row_ptr(wave * 32 + cand);
  float4 pE[4], pO[4];  // even / odd tiles, refilled two tiles ahead
  load_tile(row, 0, pE);
  load_tile(row, 1, pO);
  for (int b = wave; b < nblk; b += NW) {
    const int i = b * 32 + cand;
    const float* next = (b + NW < nblk) ? row_ptr(i + NW * 32) : row;
    f32x16 a2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        a2[mt][4 * rr] = v.x; a2[mt][4 * rr + 1] = v.y; a2[mt][4 * rr + 2] = v.z; a2[mt][4 * rr + 3] = v.w;
      }
    auto tile = [&](int t, float4 (&x)[4]) {
      const uint4* L2 = W2 + t * 1024 + lane;
      // the tile's sixteen A fragments leave LDS while the operand split runs
      f16x8 Wf[4 * H2T];
#pragma unroll
      for (int k = 0; k < 4 * H2T; ++k) Wf[k] = as_f16x8(L2[k * 64]);
      f16x8 bh[2], bl[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint4 h, l;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int rr = 2 * q + half;
          const float4 u = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
          const float4 be = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
          uint32_t h0, l0, h1, l1;
          prelu_split_pair_pk(f32x2{x[rr].x, x[rr].y}, f32x2{u.x, u.y}, f32x2{be.x, be.y}, h0, l0);
          prelu_split_pair_pk(f32x2{x[rr].z, x[rr].w}, f32x2{u.z, u.w}, f32x2{be.z, be.w}, h1, l1);
          if (half == 0) { h.x = h0; h.y = h1; l.x = l0; l.y = l1; } else { h.z = h0; h.w = h1; l.z = l0; l.w = l1; }
        }
        bh[q] = as_f16x8(h); bl[q] = as_f16x8(l);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const bool wrap = t + 2 >= H1T;
        load_tile(wrap ? next : row, wrap ? t + 2 - H1T : t + 2, x);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          const f16x8 wh = Wf[(q * H2T + mt) * 2], wl = Wf[(q * H2T + mt) * 2 + 1];
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[q], a2[mt], 0, 0, 0);
        }
    };
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) {
      tile(t, pE);
      tile(t + 1, pO);
    }
    row = next;
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 be = *reinterpret_cast<const float4*>(&V->beta2[32 * mt + 8 * rr + 4 * g]);
        const float4 w3 = *reinterpret_cast<const float4*>(&V->w3[32 * mt + 8 * rr + 4 * g]);
        const float bes[4] = {be.x, be.y, be.z, be.w}, w3s[4] = {w3.x, w3.y, w3.z, w3.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xa = a2[mt][4 * rr + e];
          part = __builtin_fmaf(__builtin_fmaf(neg_part(xa), bes[e], xa), w3s[e], part);
        }
      }
    const float other = __shfl_xor(part, 32);
    constexpr float kUnscale = 1.0f / (kSplit2Scale * kSplit2Scale);
    if (g == 0 && i < n) scores[i] = (part + other) * kUnscale;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// exact f32, W2 resident, layer 1 from the pre-projected table.  W2 = the scorer's p2x (f32 A fragments of
// v_mfma_f32_32x32x2_f32, four consecutive chain steps per 16 bytes:
//     p2x[t][mt][j][lane][i] = W2[32 t + i + 8 j + 4 (lane >> 5)][32 mt + (lane & 31)]
// i.e. step r = 4 j + i of tile t in ORDER_H); V staged by wg_mlp_xres_vectors.  Bit for bit the oracle's
// score_mlp_row: a1 = u + P (P = table / 2^7, exact), prelu, the ORDER_H fmaf chain on the matrix cores, prelu, ORDER_O.
template <int NT>
__device__ __forceinline__ void wg_score_mlp_xres(const float* __restrict__ proj, uint32_t n_table_rows,
                                                  const int32_t* ids, int n, const float4* W2, const Mlp2Vectors* V,
                                                  float* scores) {
  constexpr int H1T = 8, H2T = 4, NW = NT / 64;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  const int nblk = (n + 31) >> 5;
  if (wave >= nblk) return;
  auto row_ptr = [&](int i) -> const float* {
    const int ic = min(i, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    return proj + (size_t)(rid < n_table_rows ? rid : 0u) * kMlpProjWidth + 4 * g;
  };
  auto load_tile = [&](const float* row, int t, float4 (&p)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) p[rr] = *reinterpret_cast<const float4*>(row + 32 * t + 8 * rr);
  };
  const float* row = row_ptr(wave * 32 + cand);
  float4 pE[4], pO[4];
  load_tile(row, 0, pE);
  load_tile(row, 1, pO);
  for (int b = wave; b < nblk; b += NW) {
    const int i = b * 32 + cand;
    const float* next = (b + NW < nblk) ? row_ptr(i + NW * 32) : row;
    f32x16 acc2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        acc2[mt][4 * rr] = v.x; acc2[mt][4 * rr + 1] = v.y; acc2[mt][4 * rr + 2] = v.z; acc2[mt][4 * rr + 3] = v.w;
      }
    auto tile = [&](int t, float4 (&x)[4]) {
      // h1 of this lane's 16 hidden units of tile t (register r = 4 rr + e <-> unit 32 t + 8 rr + 4 g + e)
      float h[16];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 u = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
        const float4 al = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
        constexpr float kInv = 1.0f / kSplit2Scale;  // the table holds 2^7 P: exact both ways
        h[4 * rr + 0] = prelu(u.x + x[rr].x * kInv, al.x);
        h[4 * rr + 1] = prelu(u.y + x[rr].y * kInv, al.y);
        h[4 * rr + 2] = prelu(u.z + x[rr].z * kInv, al.z);
        h[4 * rr + 3] = prelu(u.w + x[rr].w * kInv, al.w);
      }
      {
        const bool wrap = t + 2 >= H1T;
        load_tile(wrap ? next : row, wrap ? t + 2 - H1T : t + 2, x);
      }
      const float4* A = W2 + (size_t)t * (H2T * 4 * 64) + lane;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 f[H2T];
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) f[mt] = A[(mt * 4 + j) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < H2T; ++mt) {
            const float a = e == 0 ? f[mt].x : e == 1 ? f[mt].y : e == 2 ? f[mt].z : f[mt].w;
            acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, h[4 * j + e], acc2[mt], 0, 0, 0);
          }
      }
    };
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) {
      tile(t, pE);
      tile(t + 1, pO);
    }
    row = next;
    // PReLU of layer 2 and the bias-free output layer: per-lane chain over its 64 outputs (wg_score_mlp's epilogue)
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g;
        part = __fmaf_rn(prelu(acc2[mt][r], V->beta2[m]), V->w3[m], part);
      }
    const float other = __shfl_xor(part, 32);
    const float p0 = g == 0 ? part : other, p1 = g == 0 ? other : part;
    if (g == 0 && i < n) scores[i] = p0 + p1;
  }
}

}  // namespace nann
