// nann_mlp2.h -- the split-f16 MLP scorer, second mapping (round 3): 64 candidate rows per wavefront at ONE
// wavefront per SIMD.
//
// Same arithmetic as wg_score_mlp_split (nann_mlp.h: every f32 operand as hi + lo f16, products on
// v_mfma_f32_32x32x16_f16 with f32 accumulation; layer-1 accumulators become layer-2 B fragments in place), other
// mapping.  What bounded the first mapping (8 wavefronts x 32 rows, two wavefronts per SIMD): a
// wavefront's stream was [16 chained MFMAs] -> [~130 VALU of PReLU + operand split, all dependent on the chain] ->
// [24 MFMAs]; the two wavefronts of a SIMD ran that stream in lockstep, so the matrix pipe idled whenever both were
// in their VALU stretch (matrix pipe 44 % busy by counter), and every MFMA paid one ds_read_b128 of A fragment.
// Here a wavefront owns TWO independent 32-row blocks (A, B) and 512 registers:
//   * every A fragment (weights, LDS) feeds two MFMAs -- half the LDS reads per MFMA;
//   * the in-order instruction stream itself interleaves: block A's operand split sits between block B's layer-1
//     MFMAs, block B's between block A's layer-2 MFMAs (the matrix pipe executes an MFMA for 32 cycles while the
//     VALU issues ~14 independent instructions underneath it), pinned with sched_group_barrier;
//   * the VALU work per element is 3.5 instructions instead of ~8: PReLU as x + (alpha - 1) min(x, 0) (one v_min,
//     one v_fma; the activation scale equals the accumulator's, so no rescale multiply), hi = cvt_pkrtz(h0, h1),
//     lo = f16(h - hi) as ONE v_fma_mix{lo,hi}_f16 per element (f16 source operand converted by the instruction).
// One barrier per hidden tile (the weight slices of tile t + 1 travel L2 -> registers -> LDS underneath tile t).
// 256 threads: the traversal kernel that hosts it runs 4 wavefronts per CU (k_search<.., 256>).
#pragma once
#include <type_traits>
#include "nann_mlp.h"

namespace nann {

constexpr int kMlp2NT = 256;
constexpr float kSplit2Scale = 128.0f;  // weights x 2^7 (host packing, shared with the first mapping) AND activations x 2^7:
                                        // |w| <= 511, |h1| <= 511; layer-2 accumulators carry 2^14

struct Mlp2Vectors {
  float u[256];      // (b1 + W1q^T q) x 2^7 of the current query
  float beta1[256];  // alpha1 - 1
  float b2[128];     // b2 x 2^14
  float beta2[128];  // alpha2 - 1
  float w3[128];
};

template <int D>
struct Mlp2Scratch {
  static constexpr int KC = D / 16;        // 16-deep chunks of layer 1
  static constexpr int kL1 = KC * 2 * 64;  // uint4 per layer-1 slice of a hidden tile: [chunk][hi, lo][lane]
  static constexpr int kL2 = 2 * 4 * 2 * 64;  // uint4 per layer-2 slice: [q][output tile][hi, lo][lane]
  static constexpr int kTile = kL1 + kL2;
  uint4 buf[2][kTile];  // tile t in buf[t & 1]
  Mlp2Vectors v;
};

// per scoring call: the query's u and the small vectors into LDS, pre-scaled.  All NT threads; ends with a barrier.
template <int NT>
__device__ __forceinline__ void wg_mlp2_stage_setup(const MlpParams& P, float u, Mlp2Vectors* V) {
  const int tid = local_tid();
  static_assert(NT >= 256, "one hidden unit per thread");
  if (tid < 256) {
    V->u[tid] = u * kSplit2Scale;
    V->beta1[tid] = P.alpha1[tid] - 1.0f;
  }
  if (tid < 128) {
    V->b2[tid] = P.b2[tid] * (kSplit2Scale * kSplit2Scale);
    V->beta2[tid] = P.alpha2[tid] - 1.0f;
    V->w3[tid] = P.w3[tid];
  }
  __syncthreads();
}

// h = x + beta min(x, 0) for a pair of accumulator values, split into f16 hi (round toward zero) and lo = f16(h - hi):
// 2 v_min + 2 v_fma + cvt_pkrtz + v_fma_mixlo_f16 + v_fma_mixhi_f16
__device__ __forceinline__ void prelu_split_pair(float x0, float x1, float b0, float b1, uint32_t& hi, uint32_t& lo) {
  typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
  const float h0 = __builtin_fmaf(neg_part(x0), b0, x0);
  const float h1 = __builtin_fmaf(neg_part(x1), b1, x1);
  const h2_t hv = __builtin_amdgcn_cvt_pkrtz(h0, h1);
  hi = __builtin_bit_cast(uint32_t, hv);
  uint32_t l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(h0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(h1));
  lo = l;
}

// The same for h = (x + u) + beta min(x + u, 0) with the add and the fma as PACKED f32 instructions (v_pk_add_f32,
// v_pk_fma_f32: two values per instruction at the scalar ones' issue cost -- a wavefront's vector instruction holds its
// SIMD for four cycles, and this scorer's time is two thirds issue): 3.5 instructions per value instead of 4.5.
// Element for element the same IEEE operations as prelu_split_pair(x0 + u0, x1 + u1, ...): bit-identical.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void prelu_split_pair_pk(f32x2 x, f32x2 u, f32x2 b, uint32_t& hi, uint32_t& lo) {
  typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
  const f32x2 xs = x + u;
  // (plain min, not neg_part's asm: an add's result needs no canonicalising, and asm outputs land in unrelated
  //  registers, which would scalarise the packed fma that consumes them)
  const f32x2 m = __builtin_elementwise_min(xs, f32x2{0.0f, 0.0f});
  const f32x2 h = __builtin_elementwise_fma(m, b, xs);
  const h2_t hv = __builtin_amdgcn_cvt_pkrtz(h.x, h.y);
  hi = __builtin_bit_cast(uint32_t, hv);
  uint32_t l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(h.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(h.y));
  lo = l;
}

// the finished layer-1 tile of one block -> the B fragments of its two 16-deep layer-2 steps (bh / bl [q]);
// beta[r] = (alpha1 - 1) of the hidden unit register r holds
__device__ __forceinline__ void split_tile(const f32x16& a1, const float (&beta)[16], f16x8 (&bh)[2], f16x8 (&bl)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    uint4 h, l;
    prelu_split_pair(a1[8 * q + 0], a1[8 * q + 1], beta[8 * q + 0], beta[8 * q + 1], h.x, l.x);
    prelu_split_pair(a1[8 * q + 2], a1[8 * q + 3], beta[8 * q + 2], beta[8 * q + 3], h.y, l.y);
    prelu_split_pair(a1[8 * q + 4], a1[8 * q + 5], beta[8 * q + 4], beta[8 * q + 5], h.z, l.z);
    prelu_split_pair(a1[8 * q + 6], a1[8 * q + 7], beta[8 * q + 6], beta[8 * q + 7], h.w, l.w);
    bh[q] = as_f16x8(h);
    bl[q] = as_f16x8(l);
  }
}

// wg_score_mlp_split2: scores[i] for candidates ids[i], i < n (ids == nullptr: row i).  All 256 threads; 4 wavefronts x
// (32 + 32) candidates per pass.  wg_mlp2_stage_setup must have run for this query.  Rows outside [0, n_table_rows)
// are read as row 0 (the caller reports them).
// VAR (timing builds of tools/ubench_mlp2.hip only; the product instantiates VAR = 0): bit 0 = no PReLU / operand
// split arithmetic, bit 1 = no weight staging (fetch, LDS stores, barrier), bit 2 = A fragments read from LDS once per
// pass instead of per tile, bit 3 = no output layer.  Wrong scores, same control flow.
template <int D, int DT, int VAR = 0>
__device__ __forceinline__ void wg_score_mlp_split2(const MlpParams& P, const void* __restrict__ table,
                                                    uint32_t n_table_rows, const int32_t* ids, int n,
                                                    Mlp2Scratch<D>* S, float* scores) {
  static_assert(DT == DT_F16 || DT == DT_BF16, "split form: 16-bit table rows");
  static_assert(D == 64 || D == 128, "two tile buffers next to the visited set: d <= 128");
  using Scr = Mlp2Scratch<D>;
  constexpr int NT = kMlp2NT, KC = Scr::KC, H1T = 8, H2T = 4;
  constexpr int CPP = (NT / 64) * 64;    // candidates per pass
  constexpr int PERT = Scr::kTile / NT;  // uint4 per thread per staged tile
  static_assert(Scr::kTile % NT == 0, "tile slices divide over the threads");
  const Mlp2Vectors* V = &S->v;
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;

  // tile t in global memory: its layer-1 fragments (P.p1, [t][kc][plane][lane]) then its layer-2 ones (P.p2, [t][q][m][plane][lane])
  auto tile_src = [&](int t, int f) -> const uint4* {
    return f < Scr::kL1 ? P.p1 + (size_t)t * Scr::kL1 + f : P.p2 + (size_t)t * Scr::kL2 + (f - Scr::kL1);
  };
  // staging registers of the next tile: named scalars (an array here ends up in scratch memory)
  static_assert(PERT == 6 || PERT == 8, "d = 64: 6, d = 128: 8 uint4 per thread per tile");
  uint4 st0, st1, st2, st3, st4, st5, st6, st7;
  auto fetch_tile = [&](int t) {
    st0 = *tile_src(t, 0 * NT + tid); st1 = *tile_src(t, 1 * NT + tid); st2 = *tile_src(t, 2 * NT + tid);
    st3 = *tile_src(t, 3 * NT + tid); st4 = *tile_src(t, 4 * NT + tid); st5 = *tile_src(t, 5 * NT + tid);
    if constexpr (PERT == 8) { st6 = *tile_src(t, 6 * NT + tid); st7 = *tile_src(t, 7 * NT + tid); }
  };
  auto store_tile = [&](int b) {
    uint4* dst = &S->buf[b][tid];
    dst[0 * NT] = st0; dst[1 * NT] = st1; dst[2 * NT] = st2; dst[3 * NT] = st3; dst[4 * NT] = st4; dst[5 * NT] = st5;
    if constexpr (PERT == 8) { dst[6 * NT] = st6; dst[7 * NT] = st7; }
  };
  auto row_of = [&](int i) -> size_t {
    const int ic = min(i, n - 1);
    const uint32_t rid = ids ? (uint32_t)ids[ic] : (uint32_t)ic;
    return rid < n_table_rows ? rid : 0u;
  };
  // B fragments of layer 1: chunk kc of this lane = elements 16 kc + 8 g .. + 8 of its row (one 16-byte load)
  uint4 evA[KC], evB[KC];
  auto load_rows = [&](size_t ra, size_t rb) {
    const uint4* sa = reinterpret_cast<const uint4*>(static_cast<const char*>(table) + ra * D * 2) + g;
    const uint4* sb = reinterpret_cast<const uint4*>(static_cast<const char*>(table) + rb * D * 2) + g;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) { evA[kc] = sa[2 * kc]; evB[kc] = sb[2 * kc]; }
  };
  if (n <= 0) return;
  fetch_tile(0);
  load_rows(row_of(wave * 64 + cand), row_of(wave * 64 + 32 + cand));
  __syncthreads();  // the caller is done with the scratch (the vectors were staged before, behind a barrier)
  store_tile(0);
  __syncthreads();

  for (int i0 = 0; i0 < n; i0 += CPP) {
    const int iA = i0 + wave * 64 + cand, iB = iA + 32;
    const bool more = i0 + CPP < n;
    const size_t nextA = more ? row_of(iA + CPP) : 0, nextB = more ? row_of(iB + CPP) : 0;
    f32x16 a2A[H2T], a2B[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 v = *reinterpret_cast<const float4*>(&V->b2[32 * mt + 8 * rr + 4 * g]);
        a2A[mt][4 * rr] = v.x; a2A[mt][4 * rr + 1] = v.y; a2A[mt][4 * rr + 2] = v.z; a2A[mt][4 * rr + 3] = v.w;
        a2B[mt][4 * rr] = v.x; a2B[mt][4 * rr + 1] = v.y; a2B[mt][4 * rr + 2] = v.z; a2B[mt][4 * rr + 3] = v.w;
      }
    // one hidden tile; LAST (compile time) = tile H1T - 1, after whose layer 1 the rows are dead and the next pass's are fetched
    auto tile = [&](int t, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const uint4* L1 = &S->buf[(VAR & 4) ? 0 : (t & 1)][0];
      const uint4* L2 = L1 + Scr::kL1;
      if constexpr (!(VAR & 2)) fetch_tile(LAST ? 0 : t + 1);  // the next tile (of this pass, or tile 0 for the next) from L2 underneath this one
      f32x16 a1A, a1B;
      float beta[16];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {  // the per-query part seeds both blocks' tile; the PReLU slopes of its 16 units
        const float4 v = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
        a1A[4 * rr] = v.x; a1A[4 * rr + 1] = v.y; a1A[4 * rr + 2] = v.z; a1A[4 * rr + 3] = v.w;
        a1B[4 * rr] = v.x; a1B[4 * rr + 1] = v.y; a1B[4 * rr + 2] = v.z; a1B[4 * rr + 3] = v.w;
        const float4 b = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
        beta[4 * rr] = b.x; beta[4 * rr + 1] = b.y; beta[4 * rr + 2] = b.z; beta[4 * rr + 3] = b.w;
      }
      f16x8 W1[2 * KC];
#pragma unroll
      for (int k = 0; k < 2 * KC; ++k) W1[k] = as_f16x8(L1[k * 64 + lane]);
      __builtin_amdgcn_sched_barrier(0);
      // ---- layer 1, block A: 2 KC chained MFMAs; the layer-2 fragments of this tile are read underneath
      f16x8 W2[4 * H2T];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const f16x8 b = row_chunk_f16<DT>(evA[kc]);
        a1A = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[2 * kc], b, a1A, 0, 0, 0);
        a1A = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[2 * kc + 1], b, a1A, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 4 * H2T; ++k) W2[k] = as_f16x8(L2[k * 64 + lane]);
#pragma unroll
      for (int k = 0; k < 2 * KC; ++k) {  // pin: one MFMA, then a 16-byte LDS read (16 reads under the first 16 MFMAs)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k < 4 * H2T) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (2 * KC < 4 * H2T) __builtin_amdgcn_sched_group_barrier(0x100, 4 * H2T - 2 * KC, 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- layer 1, block B; block A's PReLU + operand split rides underneath these MFMAs
      f16x8 bhA[2], blA[2], bhB[2], blB[2];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const f16x8 b = row_chunk_f16<DT>(evB[kc]);
        a1B = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[2 * kc], b, a1B, 0, 0, 0);
        a1B = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[2 * kc + 1], b, a1B, 0, 0, 0);
      }
      if constexpr (VAR & 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          union { float f[4]; f16x8 h; } c;
          c.f[0] = a1A[8 * q]; c.f[1] = a1A[8 * q + 1]; c.f[2] = a1A[8 * q + 2]; c.f[3] = a1A[8 * q + 3];
          bhA[q] = c.h;
          c.f[0] = a1A[8 * q + 4]; c.f[1] = a1A[8 * q + 5]; c.f[2] = a1A[8 * q + 6]; c.f[3] = a1A[8 * q + 7];
          blA[q] = c.h;
        }
      } else {
        split_tile(a1A, beta, bhA, blA);
      }
      // pin: two MFMAs first (block A's last layer-1 result is still in the pipe), then VALU in groups of 8 per MFMA
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int k = 2; k < 2 * KC; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LAST) { if (more) load_rows(nextA, nextB); }  // the rows are consumed: fetch the next pass's
      // ---- layer 2, block A; block B's split underneath
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2], bhA[q], a2A[mt], 0, 0, 0);
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2], blA[q], a2A[mt], 0, 0, 0);
          a2A[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2 + 1], bhA[q], a2A[mt], 0, 0, 0);
        }
      if constexpr (VAR & 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          union { float f[4]; f16x8 h; } c;
          c.f[0] = a1B[8 * q]; c.f[1] = a1B[8 * q + 1]; c.f[2] = a1B[8 * q + 2]; c.f[3] = a1B[8 * q + 3];
          bhB[q] = c.h;
          c.f[0] = a1B[8 * q + 4]; c.f[1] = a1B[8 * q + 5]; c.f[2] = a1B[8 * q + 6]; c.f[3] = a1B[8 * q + 7];
          blB[q] = c.h;
        }
      } else {
        split_tile(a1B, beta, bhB, blB);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
      for (int k = 2; k < 6 * H2T; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- layer 2, block B; the next tile's slices go to LDS underneath
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2], bhB[q], a2B[mt], 0, 0, 0);
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2], blB[q], a2B[mt], 0, 0, 0);
          a2B[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[(q * H2T + mt) * 2 + 1], bhB[q], a2B[mt], 0, 0, 0);
        }
      if constexpr (!(VAR & 2)) {
        store_tile((t + 1) & 1);
#pragma unroll
        for (int k = 0; k < PERT; ++k) {  // pin: the stores spread under the MFMAs, two MFMAs apart
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
      }
    };
#pragma unroll 1
    for (int t = 0; t < H1T - 1; ++t) tile(t, std::false_type{});
    tile(H1T - 1, std::true_type{});
    // PReLU of layer 2 and the bias-free output layer, both blocks from one read of the vectors
    float partA = 0.0f, partB = 0.0f;
    if constexpr (VAR & 8) { partA = a2A[0][0] + a2A[1][1] + a2A[2][2] + a2A[3][3]; partB = a2B[0][0] + a2B[1][1] + a2B[2][2] + a2B[3][3]; }
    else
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
          partA = __builtin_fmaf(__builtin_fmaf(__builtin_fminf(xa, 0.0f), bes[e], xa), w3s[e], partA);
          partB = __builtin_fmaf(__builtin_fmaf(__builtin_fminf(xb, 0.0f), bes[e], xb), w3s[e], partB);
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

}  // namespace nann
