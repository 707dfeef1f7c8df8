// tools/rejected/nann_mlp3_streamed_layer2.h -- RETIRED in round 5; not compiled into the library.
//
// Round 3's split-f16 MLP scorers on the pre-projected table with LAYER 2 STREAMED through LDS once per 256-row pass
// (wg_score_mlp_proj: 256 threads, two 32-row blocks per wavefront; wg_score_mlp_proj1: 512 threads, one block per
// wavefront, slices handed over a pair of hidden tiles at a time).  configs[2]: 268 k / 323-325 k queries/s
// (profiles/r3i_bench_mlp.json, r3h_bench_mlp*.json), matrix pipe 0.36 busy, ~9 k of a pass's 28.7 k cycles stalled on the
// four barriers of the slice hand-over.  Superseded by nann_amd/csrc/nann_mlp5.h (all of W2 resident in LDS, no barrier in
// the loop: 383-410 k); from round 4 on they were reachable only through NANN_MLP_MAPPING=3|4 in the environment
// (scorer kind kScorerMlpProj = 5, launch_search_mlp_proj).  Kept as the A/B's other arm: to build it again, include this
// header behind nann_mlp3.h and restore the kScorerMlpProj branches of search_one (git show 033bcc0:nann_amd/csrc/nann_search.h).
#pragma once
#include "../../nann_amd/csrc/nann_mlp3.h"

namespace nann {

struct Mlp3Scratch {
  // layer-2 slices ([q][output tile][hi, lo][lane], 16 KB per hidden tile), handed over a PAIR of tiles at a time:
  // tiles 2 p, 2 p + 1 in buf[p & 1] -- four barriers per pass instead of eight
  uint4 buf[2][2048];
  Mlp2Vectors v;
};
static_assert(sizeof(Mlp3Scratch) <= sizeof(MlpSplitScratch), "fits the phase scratch of the first mapping");

// wg_score_mlp_proj: scores[i] for candidates ids[i], i < n (ids == nullptr: row i), from the pre-projected table.
// All 256 threads; 4 wavefronts x (32 + 32) candidates per pass.  wg_mlp2_stage_setup must have run for this query.
// Rows outside [0, n_table_rows) are read as row 0 (the caller reports them).
__device__ __forceinline__ void wg_score_mlp_proj(const MlpParams& P, const float* __restrict__ proj, uint32_t n_table_rows,
                                                  const int32_t* ids, int n, Mlp3Scratch* S, float* scores) {
  constexpr int NT = kMlp2NT, H1T = 8, H2T = 4, CPP = 256;
  const Mlp2Vectors* V = &S->v;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  if (n <= 0) return;
  // layer-2 slice of tile t: P.p2 + t * 1024 uint4; 4 per thread
  uint4 st0, st1, st2, st3;
  auto fetch_slice = [&](int t) {
    const uint4* src = P.p2 + (size_t)t * 1024 + tid;
    st0 = src[0]; st1 = src[NT]; st2 = src[2 * NT]; st3 = src[3 * NT];
  };
  auto store_slice = [&](int b) {
    uint4* dst = &S->buf[b][tid];
    dst[0] = st0; dst[NT] = st1; dst[2 * NT] = st2; dst[3 * NT] = st3;
  };
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
  const float* rowA = row_ptr(wave * 64 + cand);
  const float* rowB = row_ptr(wave * 64 + 32 + cand);
  // two tile buffers per block: even tiles in (pA, pB), odd tiles in (qA, qB); a buffer is refilled for tile t + 2 as
  // soon as tile t's operand split has consumed it, i.e. ~1.7 tiles (~4k cycles) ahead of its use -- the gathers are
  // random 128-byte accesses whose HBM latency under load is 2-3k cycles (one tile ahead, the first version, stalled
  // every tile: 50.8k cycles per pass against the second mapping's 41.7k)
  float4 pA[4], pB[4], qA[4], qB[4];
  load_tile(rowA, 0, pA);
  load_tile(rowB, 0, pB);
  load_tile(rowA, 1, qA);
  load_tile(rowB, 1, qB);
  fetch_slice(0);
  __syncthreads();  // the caller is done with the scratch (the vectors were staged before, behind a barrier)
  store_slice(0);
  __syncthreads();

  for (int i0 = 0; i0 < n; i0 += CPP) {
    const int iA = i0 + wave * 64 + cand, iB = iA + 32;
    const bool more = i0 + CPP < n;
    const float* nextA = more ? row_ptr(iA + CPP) : rowA;
    const float* nextB = more ? row_ptr(iB + CPP) : rowB;
    f32x16 a2A[H2T], a2B[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        a2A[mt][4 * rr] = v.x; a2A[mt][4 * rr + 1] = v.y; a2A[mt][4 * rr + 2] = v.z; a2A[mt][4 * rr + 3] = v.w;
        a2B[mt][4 * rr] = v.x; a2B[mt][4 * rr + 1] = v.y; a2B[mt][4 * rr + 2] = v.z; a2B[mt][4 * rr + 3] = v.w;
      }
    // one hidden tile: its pieces are in (xA, xB); they are refilled with tile t + 2 (of these rows, or of the next
    // pass's) right after the split
    auto tile = [&](int t, float4 (&xA)[4], float4 (&xB)[4]) {
      const uint4* L2 = &S->buf[t & 1][0];
      fetch_slice(t + 1 < H1T ? t + 1 : 0);
      f16x8 W2[4 * H2T];
#pragma unroll
      for (int k = 0; k < 4 * H2T; ++k) W2[k] = as_f16x8(L2[k * 64 + lane]);
      // h = PReLU(P + u), split into the layer-2 B fragments (hi / lo) -- both blocks
      f16x8 bhA[2], blA[2], bhB[2], blB[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {  // fragment q = registers 8 q .. 8 q + 7 of the tile = pieces rr = 2 q, 2 q + 1
        uint4 hA, lA, hB, lB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int rr = 2 * q + half;
          const float4 u = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
          const float4 be = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
          uint32_t h0, l0, h1, l1;
          prelu_split_pair(xA[rr].x + u.x, xA[rr].y + u.y, be.x, be.y, h0, l0);
          prelu_split_pair(xA[rr].z + u.z, xA[rr].w + u.w, be.z, be.w, h1, l1);
          if (half == 0) { hA.x = h0; hA.y = h1; lA.x = l0; lA.y = l1; } else { hA.z = h0; hA.w = h1; lA.z = l0; lA.w = l1; }
          prelu_split_pair(xB[rr].x + u.x, xB[rr].y + u.y, be.x, be.y, h0, l0);
          prelu_split_pair(xB[rr].z + u.z, xB[rr].w + u.w, be.z, be.w, h1, l1);
          if (half == 0) { hB.x = h0; hB.y = h1; lB.x = l0; lB.y = l1; } else { hB.z = h0; hB.w = h1; lB.z = l0; lB.w = l1; }
        }
        bhA[q] = as_f16x8(hA); blA[q] = as_f16x8(lA);
        bhB[q] = as_f16x8(hB); blB[q] = as_f16x8(lB);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const bool wrap = t + 2 >= H1T;
        const int nt = wrap ? t + 2 - H1T : t + 2;
        load_tile(wrap ? nextA : rowA, nt, xA);
        load_tile(wrap ? nextB : rowB, nt, xB);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- layer 2, both blocks: every fragment pair feeds six MFMAs
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          const f16x8 wh = W2[(q * H2T + mt) * 2], wl = W2[(q * H2T + mt) * 2 + 1];
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bhA[q], a2A[mt], 0, 0, 0);
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bhB[q], a2B[mt], 0, 0, 0);
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, blA[q], a2A[mt], 0, 0, 0);
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, blB[q], a2B[mt], 0, 0, 0);
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bhA[q], a2A[mt], 0, 0, 0);
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bhB[q], a2B[mt], 0, 0, 0);
        }
      store_slice((t + 1) & 1);
      __syncthreads();
    };
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) {
      tile(t, pA, pB);
      tile(t + 1, qA, qB);
    }
    rowA = nextA;
    rowB = nextB;
    // PReLU of layer 2 and the bias-free output layer, both blocks from one read of the vectors
    float partA = 0.0f, partB = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 be = *reinterpret_cast<const float4*>(&V->beta2[32 * mt + 8 * rr + 4 * g]);
        const float4 w3 = *reinterpret_cast<const float4*>(&V->w3[32 * mt + 8 * rr + 4 * g]);
        const float bes[4] = {be.x, be.y, be.z, be.w}, w3s[4] = {w3.x, w3.y, w3.z, w3.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xa = a2A[mt][4 * rr + e], xb = a2B[mt][4 * rr + e];
          partA = __builtin_fmaf(__builtin_fmaf(neg_part(xa), bes[e], xa), w3s[e], partA);
          partB = __builtin_fmaf(__builtin_fmaf(neg_part(xb), bes[e], xb), w3s[e], partB);
        }
      }
    const float oa = __shfl_xor(partA, 32), ob = __shfl_xor(partB, 32);
    constexpr float kUnscale = 1.0f / (kSplit2Scale * kSplit2Scale);
    if (g == 0) {
      if (iA < n) scores[iA] = (partA + oa) * kUnscale;
      if (iB < n) scores[iB] = (partB + ob) * kUnscale;
    }
  }
  __syncthreads();
}

// The same scorer for a 512-thread workgroup: 8 wavefronts x ONE 32-row block, two wavefronts per SIMD (<= 256
// registers each).  A wavefront issues in order and nothing of its own overlaps (see the header), so with one wavefront
// per SIMD every latency of the tile loop -- the weight slice's trip from L2, the LDS write + barrier of its hand-over,
// the fragment reads, the gathers -- is SIMD idle time (measured: 36 k cycles per 256-row pass even with the rows
// L2-resident, against ~17 k of instruction issue); a second wavefront on the SIMD fills it.  The price is one LDS
// fragment read per three MFMAs instead of per six.
__device__ __forceinline__ void wg_score_mlp_proj1(const MlpParams& P, const float* __restrict__ proj, uint32_t n_table_rows,
                                                   const int32_t* ids, int n, Mlp3Scratch* S, float* scores) {
  constexpr int NT = 512, H1T = 8, H2T = 4, CPP = 256;
  const Mlp2Vectors* V = &S->v;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  if (n <= 0) return;
  uint4 st0, st1, st2, st3;  // the layer-2 slices of a pair of tiles: 2048 uint4, four per thread
  auto fetch_pair = [&](int pr) {
    const uint4* src = P.p2 + (size_t)pr * 2048 + tid;
    st0 = src[0]; st1 = src[NT]; st2 = src[2 * NT]; st3 = src[3 * NT];
  };
  auto store_pair = [&](int b) {
    uint4* dst = &S->buf[b][tid];
    dst[0] = st0; dst[NT] = st1; dst[2 * NT] = st2; dst[3 * NT] = st3;
  };
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
  float4 pE[4], pO[4];  // even / odd tiles, refilled two tiles ahead
  load_tile(row, 0, pE);
  load_tile(row, 1, pO);
  fetch_pair(0);
  __syncthreads();  // the caller is done with the scratch (the vectors were staged before, behind a barrier)
  store_pair(0);
  __syncthreads();

  for (int i0 = 0; i0 < n; i0 += CPP) {
    const int i = i0 + wave * 32 + cand;
    const bool more = i0 + CPP < n;
    const float* next = more ? row_ptr(i + CPP) : row;
    f32x16 a2[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        a2[mt][4 * rr] = v.x; a2[mt][4 * rr + 1] = v.y; a2[mt][4 * rr + 2] = v.z; a2[mt][4 * rr + 3] = v.w;
      }
    auto tile = [&](int t, float4 (&x)[4]) {
      const uint4* L2 = &S->buf[(t >> 1) & 1][(t & 1) * 1024];
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
      for (int q = 0; q < 2; ++q) {
        f16x8 W2[2 * H2T];
#pragma unroll
        for (int k = 0; k < 2 * H2T; ++k) W2[k] = as_f16x8(L2[(q * 2 * H2T + k) * 64 + lane]);
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2], bh[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2], bl[q], a2[mt], 0, 0, 0);
          a2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[mt * 2 + 1], bh[q], a2[mt], 0, 0, 0);
        }
      }
    };
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) {
      fetch_pair(t + 2 < H1T ? (t >> 1) + 1 : 0);  // the next pair's slices (or pair 0 for the next pass) underneath this one
      tile(t, pE);
      tile(t + 1, pO);
      store_pair(((t >> 1) + 1) & 1);
      __syncthreads();
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
  __syncthreads();
}


}  // namespace nann
