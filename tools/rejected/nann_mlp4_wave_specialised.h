// REJECTED EXPERIMENT (round 3; not part of the build -- kept for the record).  It ran, bit-identical
// to nann_mlp3.h, at 3.61 ms per 1024 queries against 3.20 ms: a wavefront's vector instruction occupies its SIMD for four
// cycles, so the two producers need ~2.4 k cycles per step for the six blocks the consumers finish in ~0.8 k.
//
// nann_mlp4.h -- the pre-projected split-f16 MLP scorer (nann_mlp3.h) with its two kinds of work on DIFFERENT SIMDs.
//
// Measured on gfx950: a SIMD never overlaps vector instructions with its matrix instructions -- not
// inside a wavefront, not across the two wavefronts it hosts.  In nann_mlp3.h every wavefront alternates between the
// two (gather a tile of P, add the query's part, PReLU, split into f16 planes: ~4.5 vector instructions per value;
// then 24 MFMAs), so each SIMD's matrix pipe idles while its vector pipe works and the other way round: 43 % busy.
// The four SIMDs of a CU are independent, though.  Here the two wavefronts that share a SIMD with wavefront 0 are
// PRODUCERS: they do all the vector work of a step -- the tile of six 32-row blocks -- and leave the finished B
// fragments (hi / lo planes in MFMA lane order) in LDS; the six wavefronts on the other three SIMDs are CONSUMERS:
// B fragments and the layer-2 weight fragments from LDS, 24 MFMAs per step, nothing else until the output layer.
// One workgroup barrier per step hands a tile over (step s is produced during step s - 1, two buffers).
// Which wavefronts share a SIMD is read from HW_ID (observed: w and w + 4; the first SIMD rotates per workgroup); if
// the placement is ever not two per SIMD the caller falls back to nann_mlp3.h (uniform decision).
// A pass = 6 x 32 = 192 candidates, 8 steps.  The arithmetic per value is nann_mlp3.h's: scores are bit-identical.
#pragma once
#include "nann_mlp3.h"

namespace nann {

constexpr int kMlp4Blocks = 6;  // consumer wavefronts = 32-row blocks per pass
struct Mlp4Scratch {
  uint4 wbuf[2][1024];               // layer-2 fragments of hidden tile t ([q][output tile][hi, lo][lane]) in wbuf[step & 1]
  uint4 bbuf[2][kMlp4Blocks][256];   // B fragments of block b ([hi0, lo0, hi1, lo1][lane]) in bbuf[step & 1][b]
  Mlp2Vectors v;
  int simd[8];
};

// Returns false (nothing done) when the workgroup's wavefronts are not placed two per SIMD.
__device__ __forceinline__ bool wg_score_mlp_ws(const MlpParams& P, const float* __restrict__ proj, uint32_t n_table_rows,
                                                const int32_t* ids, int n, Mlp4Scratch* S, float* scores) {
  constexpr int NT = 512, H1T = 8, H2T = 4, NB = kMlp4Blocks, CPP = 32 * NB;
  const Mlp2Vectors* V = &S->v;
  const int tid = local_tid(), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cand = lane & 31, g = lane >> 5;
  if (n <= 0) return true;
  // ---- roles: producers = the wavefronts on wavefront 0's SIMD (HW_ID bits 5:4)
  if (lane == 0) S->simd[wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  __syncthreads();
  int n_prod = 0, prod_ix = -1, cons_ix = -1, n_cons = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const bool p = S->simd[w] == S->simd[0];
    if (w == wave) { prod_ix = p ? n_prod : -1; cons_ix = p ? -1 : n_cons; }
    n_prod += p ? 1 : 0;
    n_cons += p ? 0 : 1;
  }
  if (n_prod != 2) return false;  // uniform
  const bool producer = prod_ix >= 0;
  const int n_pass = (n + CPP - 1) / CPP;
  const int n_steps = n_pass * H1T;

  auto row_ptr = [&](int i) -> const float* {
    const int ic = min(i, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    return proj + (size_t)(rid < n_table_rows ? rid : 0u) * kMlpProjWidth + 4 * g;
  };
  auto load_tile = [&](const float* row, int t, float4 (&p)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) p[rr] = *reinterpret_cast<const float4*>(row + 32 * t + 8 * rr);
  };

  if (producer) {
    // ===== producers: blocks 3 prod_ix .. + 2; the P tiles of step k sit in pre[k & 1], fetched two steps ahead
    const int b0 = 3 * prod_ix;
    const float* row[3];
    const float* nxt[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { row[j] = row_ptr(32 * (b0 + j) + cand); nxt[j] = row[j]; }
    float4 pre[2][3][4];
#pragma unroll
    for (int j = 0; j < 3; ++j) { load_tile(row[j], 0, pre[0][j]); load_tile(row[j], 1, pre[1][j]); }
    auto produce = [&](int k, float4 (&x)[3][4]) {  // step k -> bbuf[k & 1]; then refill x with step k + 2
      const int t = k & (H1T - 1);
      float4 u[4], be[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        u[rr] = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
        be[rr] = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        uint4* dst = &S->bbuf[k & 1][b0 + j][lane];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          uint4 h, l;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int rr = 2 * q + half;
            uint32_t h0, l0, h1, l1;
            prelu_split_pair(x[j][rr].x + u[rr].x, x[j][rr].y + u[rr].y, be[rr].x, be[rr].y, h0, l0);
            prelu_split_pair(x[j][rr].z + u[rr].z, x[j][rr].w + u[rr].w, be[rr].z, be[rr].w, h1, l1);
            if (half == 0) { h.x = h0; h.y = h1; l.x = l0; l.y = l1; } else { h.z = h0; h.w = h1; l.z = l0; l.w = l1; }
          }
          dst[(2 * q) * 64] = h;
          dst[(2 * q + 1) * 64] = l;
        }
      }
      // refill: step k + 2 (tile t + 2 of this pass, or of the next one)
      const bool wrap = t + 2 >= H1T;
#pragma unroll
      for (int j = 0; j < 3; ++j) load_tile(wrap ? nxt[j] : row[j], (t + 2) & (H1T - 1), x[j]);
    };
    // prologue: step 0 before the consumers start
    {
      const bool more = CPP < n;
#pragma unroll
      for (int j = 0; j < 3; ++j) nxt[j] = more ? row_ptr(CPP + 32 * (b0 + j) + cand) : row[j];
    }
    produce(0, pre[0]);
    __syncthreads();
    for (int s = 0; s < n_steps; s += 2) {  // (two steps per iteration: the buffers alternate by name)
      // step s: produce s + 1
      if (s + 1 < n_steps) produce(s + 1, pre[1]);
      __syncthreads();
      // step s + 1: produce s + 2 (tile 0 of the next pass when s + 2 is a multiple of 8: its rows become current)
      if (((s + 2) & (H1T - 1)) == 0) {
        const int pass_next = (s + 2) / H1T;  // the pass whose tile 0 is produced now
#pragma unroll
        for (int j = 0; j < 3; ++j) row[j] = nxt[j];
        const bool more = (pass_next + 1) * CPP < n;
#pragma unroll
        for (int j = 0; j < 3; ++j) nxt[j] = more ? row_ptr((pass_next + 1) * CPP + 32 * (b0 + j) + cand) : row[j];
      }
      if (s + 2 < n_steps) produce(s + 2, pre[0]);
      __syncthreads();
    }
  } else {
    // ===== consumers: block cons_ix; the first four also move the layer-2 slice of the next step L2 -> LDS
    const bool stager = cons_ix < 4;
    const int st_lane = cons_ix * 64 + lane;  // 0..255: four uint4 each
    uint4 st0, st1, st2, st3;
    auto fetch_w = [&](int t) {
      const uint4* src = P.p2 + (size_t)t * 1024 + st_lane;
      st0 = src[0]; st1 = src[256]; st2 = src[512]; st3 = src[768];
    };
    auto store_w = [&](int b) {
      uint4* dst = &S->wbuf[b][st_lane];
      dst[0] = st0; dst[256] = st1; dst[512] = st2; dst[768] = st3;
    };
    if (stager) { fetch_w(0); store_w(0); }
    f32x16 a2[H2T];
    auto seed = [&]() {
#pragma unroll
      for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
          a2[mt][4 * rr] = v.x; a2[mt][4 * rr + 1] = v.y; a2[mt][4 * rr + 2] = v.z; a2[mt][4 * rr + 3] = v.w;
        }
    };
    seed();
    __syncthreads();  // (the producers' prologue barrier)
    for (int s = 0; s < n_steps; ++s) {
      const int t = s & (H1T - 1);
      if (stager && s + 1 < n_steps) fetch_w((t + 1) & (H1T - 1));
      const uint4* Bf = &S->bbuf[s & 1][cons_ix][lane];
      const uint4* L2 = &S->wbuf[s & 1][0];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f16x8 bh = as_f16x8(Bf[(2 * q) * 64]);
        const f16x8 bl = as_f16x8(Bf[(2 * q + 1) * 64]);
        f16x8 W2[2 * H2T];
#pragma unroll
        for (int k = 0; k < 2 * H2T; ++k) W2[k] = as_f16x8(L2[(q * 2 * H2T + k) * 64 + lane]);
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2], bh, a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2], bl, a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2 + 1], bh, a2[mt], 0, 0, 0);
        }
      }
      if (stager && s + 1 < n_steps) store_w((s + 1) & 1);
      if (t == H1T - 1) {  // the block's output layer (model: PReLU, then the 128 -> 1 layer)
        const int i = (s / H1T) * CPP + 32 * cons_ix + cand;
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
        seed();
      }
      __syncthreads();
    }
  }
  __syncthreads();
  return true;
}

}  // namespace nann
