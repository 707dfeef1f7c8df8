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
#include <cstddef>

#include "nann_mlp3.h"

#ifndef NANN_RES_XSKEW
#define NANN_RES_XSKEW 32  // the same for the exact form (a tile = 64 f32 MFMAs of 64 cycles)
#endif
#ifndef NANN_RES_VAR
#define NANN_RES_VAR 0  // timing builds only (tools/build_res_variant.py): bit 0 no split arithmetic, bit 1 no gathers, bit 2 no epilogue
#endif

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

// Park `park_vec` uint4 of LDS at `lds` (the visited set) in global memory and reload the head of the resident weights,
// lds[0 .. RELOAD_VEC) -- what the set / the phase scratch overwrote since the last scoring call; the rest of W2 was
// loaded once per launch (k_search).  A thread parks exactly the addresses it then overwrites, in program order, so no barrier
// sits between the two; the caller's barrier before this call covers the earlier phases, the one after it the loads.
template <int NT, int RELOAD_VEC>
__device__ __forceinline__ void wg_mlp_res_enter(uint4* lds, const uint4* __restrict__ w2, uint4* park, int park_vec) {
  const int tid = local_tid();
  static_assert(RELOAD_VEC % (NT * 4) == 0, "batches of four 16-byte loads per thread");
  if (park != nullptr)
    for (int i = tid; i < park_vec; i += NT) park[i] = lds[i];
#pragma unroll
  for (int b = 0; b < RELOAD_VEC / (NT * 4); ++b) {
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = w2[(b * 4 + j) * NT + tid];
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[(b * 4 + j) * NT + tid] = v[j];
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
//
// What the loop costs, in shader cycles per scored row of the workgroup (timing builds r4b / r4c, normalised by the rows
// each build scored -- a build with wrong scores walks other beams): 48 of MFMA issue (192 x 32 cycles per 32 rows on
// four SIMDs); +22..31 for the PReLU + operand-split arithmetic -- a SIMD does not overlap its vector and matrix
// instructions, not even across the two wavefronts it hosts (confirmed here: the build without
// that arithmetic runs 76.8 instead of 96.9) --; +11..22 for the gathers of the table rows; the rest LDS reads, the
// per-call weight load and the ragged last blocks of a call.
//
// The eight hidden tiles of a 32-row block are UNROLLED (NANN_RES_ROLLED = 1: two tiles per trip of a rolled loop, the
// form of r4b): hipcc puts `s_waitcnt vmcnt(0)` at the head of a rolled loop that carries gathers in flight, i.e. every
// trip waited for the gathers issued one tile earlier (a random 128-byte access takes 2-3 k cycles under this load, a
// tile of a wavefront ~1.6 k).  Unrolled, the waits inside a block are exact (`vmcnt(4)`: this tile's four loads, not
// the next tile's).  For the unrolled body to keep its registers the LDS addresses are formed from THREE opaque bases
// (weights below / above the 64 KB an instruction's offset field reaches, and the vectors) + immediate offsets; left to
// itself the compiler hoists ~60 address registers out of the block loop and spills them into it.
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) u32x4v* lds_u4_ptr;
typedef const __attribute__((address_space(3))) f32x4v* lds_f4_ptr;
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {  // a generic pointer into LDS -> its LDS byte address
  return (uint32_t)(size_t)p;  // (the LDS aperture's low 32 bits are the LDS address)
}
// ---------------------------------------------------------------------------------------------------------------
// split-f16 scoring of a sequence of 32-row blocks by ONE wavefront as a software pipeline (round 4, second half; used by
// the fused kernel below and by the scoring launch of the pipeline of phases, nann_mlp6.h).
//
// A block is 8 tiles x 2 steps (16 k each) x 12 MFMAs (hi.hi, hi.lo, lo.hi for four 32-unit output tiles).  Two
// symmetric wavefronts per SIMD that alternate "convert a tile" / "multiply a tile" drift into step and add their
// vector time to their MFMA time (profiles/r4d: 88.7 cycles per row against an MFMA floor of 48).  Here every MFMA
// carries a slice of the NEXT step's other work in its shadow, in source order, fenced by sched_barrier:
//   hi.hi of output tile mt   -> PReLU of pair mt of the next tile's step (packed f32 by hand)
//   hi.lo                     -> its f16 halves; the hi fragment of W2 for the next step into the register just freed
//                                (ds_read); one ds_read of the next conversion's u / beta
//   lo.hi                     -> the lo fragment likewise; step 0: one 16-byte gather of the tile after next
// Per accumulator the order of products is the older loop's (hi.hi, hi.lo, lo.hi per step): same bits in layer 2.
//
// advance(k, row, next, change, u_next): called at the top of block k with this block's row pointer; sets the row
// pointer of the block behind it (this lane's row + 4 g floats; `row` again when there is none), and, when that block
// belongs to another query, change = true and u_next = this lane's 16 bytes of that query's u x 2^7 (written to L.u_wr
// behind the last read of the current u).  store(k, score): every lane, the block's score of its row (lanes g and g ^ 1
// hold the same value).  VAR: timing builds (nann_mlp6.h NANN_PHASE_VAR; 16 = the PReLU decomposition priced in round 6).  KW3B >= 0: the packed output layer (below).
struct SplitPipeLds {  // opaque LDS byte addresses (an `asm volatile("" : "+v"(x))` behind each, see wg_score_mlp_res)
  uint32_t w_lo, w_hi;  // W2 fragments below / above 64 KB, + lane * 16
  uint32_t v_at;        // Mlp2Vectors, + g * 16
  uint32_t u_at;        // the current query's u x 2^7 [256], + g * 16 (the fused kernel: = v_at)
  uint32_t u_wr;        // where this lane writes its 16 bytes of the next query's u (fused kernel: unused)
  const float* seed_base = nullptr;  // VAR & 16 only (round 6 pricing build): a table whose rows stand in for the second
  uint32_t seed_rows = 0;            // pre-projected table of the PReLU decomposition, [seed_rows, 256] f32
};
template <int VAR, int KW3B, class Advance, class Store>
__device__ __forceinline__ void wave_mlp_split_pipeline(const SplitPipeLds& L, const float* row, int n_blocks,
                                                        Advance advance, Store store) {
  constexpr int H1T = 8, H2T = 4;
  constexpr int kBeta1 = 256, kB2 = 512, kBeta2 = 640, kW3 = 768;  // Mlp2Vectors, in floats
  const uint32_t w_lo = L.w_lo, w_hi = L.w_hi, v_at = L.v_at, u_at = L.u_at;
  auto vec4 = [&](int float_index) -> f32x4v { return *reinterpret_cast<lds_f4_ptr>(v_at + 4 * float_index); };
  auto uvec4 = [&](int float_index) -> f32x4v { return *reinterpret_cast<lds_f4_ptr>(u_at + 4 * float_index); };
  auto fragt = [&](int t, int k) -> f16x8 {
    const u32x4v v = *reinterpret_cast<lds_u4_ptr>((t < 4 ? w_lo : w_hi) + (t & 3) * 16384 + k * 1024);
    return __builtin_bit_cast(f16x8, v);
  };
  auto lds_write_u = [&](float4 v) {
    typedef __attribute__((address_space(3))) f32x4v* lds_f4_wptr;
    *reinterpret_cast<lds_f4_wptr>(L.u_wr) = f32x4v{v.x, v.y, v.z, v.w};
  };
  f32x4v x[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) x[t][rr] = *reinterpret_cast<const f32x4v*>(row + 32 * t + 8 * rr);
  f16x8 Wf[2 * H2T];
  uint32_t Bh[2][2][4], Bl[2][2][4];  // [tile parity][step][pair]: the B fragments of a tile, high and low halves
  f32x4v cu[2], cb[2];                // u / beta of the step being converted
#pragma unroll
  for (int k = 0; k < 2 * H2T; ++k) Wf[k] = fragt(0, k);
  // PReLU + split of one pair in two halves, so that each rides in another MFMA's shadow: a = PReLU(x + u) ...
  auto convert_a = [&](const f32x4v (&xt)[4], int q, int p) -> f32x2 {
    const int half = p >> 1;
    const f32x4v xv = xt[2 * q + half], u = cu[half], be = cb[half];
    const f32x2 xp = (p & 1) ? f32x2{xv.z, xv.w} : f32x2{xv.x, xv.y};
    const f32x2 up = (p & 1) ? f32x2{u.z, u.w} : f32x2{u.x, u.y};
    const f32x2 bp = (p & 1) ? f32x2{be.z, be.w} : f32x2{be.x, be.y};
    // packed f32 forms by hand (left to itself hipcc scalarises about half of them; the vector pipe's issue slots
    // are what bounds this loop): x + u, min(., 0) per half (there is no packed f32 min), (alpha - 1) min + (x + u)
    const f32x2 xs = xp + up;
    if constexpr ((VAR & 16) != 0) {
      // pricing build of VERDICT r5 item 2: prelu(x) = a x + (1 - a) max(x, 0), the linear part precomputed per item (a second
      // table, gathered below) and per query, so the activation beside the MFMAs is ONE v_max per element and no beta read
      (void)bp;
      return __builtin_elementwise_max(xs, f32x2{0.0f, 0.0f});
    }
    const f32x2 m = __builtin_elementwise_min(xs, f32x2{0.0f, 0.0f});
    return __builtin_elementwise_fma(m, bp, xs);
  };
  // ... and its f16 halves: hi = rtz(a), lo = a - hi (prelu_split_pair_pk's arithmetic, nann_mlp2.h)
  auto convert_b = [&](f32x2 h, uint32_t& hi, uint32_t& lo) {
    typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(h.x, h.y));
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi), "v"(h.x));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi), "v"(h.y));
    lo = l;
  };
  auto convert_pair = [&](const f32x4v (&xt)[4], int q, int p, uint32_t& h, uint32_t& l) { convert_b(convert_a(xt, q, p), h, l); };
#pragma unroll
  for (int q = 0; q < 2; ++q) {  // the first block's tile 0, outside the pipeline
    cu[0] = uvec4(8 * (2 * q)); cu[1] = uvec4(8 * (2 * q + 1));
    cb[0] = vec4(kBeta1 + 8 * (2 * q)); cb[1] = vec4(kBeta1 + 8 * (2 * q + 1));
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) convert_pair(x[0], q, pp, Bh[0][q][pp], Bl[0][q][pp]);
  }
  cu[0] = uvec4(32); cu[1] = uvec4(32 + 8); cb[0] = vec4(kBeta1 + 32); cb[1] = vec4(kBeta1 + 32 + 8);  // tile 1, step 0
  for (int k = 0; k < n_blocks; ++k) {
    const float* next = row;
    bool change = false;
    float4 u_next = float4{0.0f, 0.0f, 0.0f, 0.0f};
    advance(k, row, next, change, u_next);
    f32x16 acc[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4v v = vec4(kB2 + 32 * mt + 8 * rr);
        acc[mt][4 * rr] = v.x; acc[mt][4 * rr + 1] = v.y; acc[mt][4 * rr + 2] = v.z; acc[mt][4 * rr + 3] = v.w;
      }
    f32x4v seed[(VAR & 16) ? 16 : 1];
    if constexpr ((VAR & 16) != 0) {
      // the item's linear part (a (.) P_i) W2: 128 f32 = 512 more bytes per scored row, issued at the top of the block, added to
      // the accumulators behind its last MFMA (the products are linear in it).  Stand-in rows: another row of the table per row.
      const uint32_t rid = (uint32_t)((row - L.seed_base) >> 8);
      const uint32_t rid2 = (uint32_t)(((unsigned long long)rid * 2654435761ull + 12345ull) % L.seed_rows);
      const float* srow = L.seed_base + (size_t)rid2 * 256 + ((row - L.seed_base) & 255);
#pragma unroll
      for (int i = 0; i < 16; ++i) seed[i] = *reinterpret_cast<const f32x4v*>(srow + 8 * i);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < H1T; ++t) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int cbuf = t & 1, nbuf = (t + 1) & 1;
        const int nt = (q ? t + 1 : t) & (H1T - 1), nq = q ^ 1;  // the next step: its W2 fragments ...
        const int ct = ((q ? t + 1 : t) + 1) & (H1T - 1);        // ... and the tile whose conversion rides on it
        // the next block's query takes over the wavefront's u: behind the last read of this block's (step (6, 0)),
        // in front of the first read for the next block's tile 0 (below)
        if (t == H1T - 2 && q == 1 && change) lds_write_u(u_next);
        const f16x8 bh = as_f16x8(uint4{Bh[cbuf][q][0], Bh[cbuf][q][1], Bh[cbuf][q][2], Bh[cbuf][q][3]});
        const f16x8 bl = as_f16x8(uint4{Bl[cbuf][q][0], Bl[cbuf][q][1], Bl[cbuf][q][2], Bl[cbuf][q][3]});
        f32x2 hv[H2T] = {};
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          if (!(VAR & 8)) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[2 * mt], bh, acc[mt], 0, 0, 0);
          if (!(VAR & 4)) hv[mt] = convert_a(x[nbuf], q, mt);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          if (!(VAR & 8)) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[2 * mt], bl, acc[mt], 0, 0, 0);
          if (!(VAR & 2)) Wf[2 * mt] = fragt(nt, nq * 2 * H2T + 2 * mt);
          const int rr = 2 * nq + (mt & 1);  // u / beta of the next step's conversion (this step's were read above)
          if (!(VAR & 4)) {
            if (mt < 2) cu[mt & 1] = uvec4(32 * ct + 8 * rr); else if (!(VAR & 16)) cb[mt & 1] = vec4(kBeta1 + 32 * ct + 8 * rr);
            convert_b(hv[mt], Bh[nbuf][q][mt], Bl[nbuf][q][mt]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt) {
          if (!(VAR & 8)) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[2 * mt + 1], bh, acc[mt], 0, 0, 0);
          if (!(VAR & 2)) Wf[2 * mt + 1] = fragt(nt, nq * 2 * H2T + 2 * mt + 1);
          if (q == 0 && !(VAR & 1))  // tile t + 2 (of the next block behind tile 5) into the buffer tile t was converted from
            x[cbuf][mt] = *reinterpret_cast<const f32x4v*>((t + 2 >= H1T ? next : row) + 32 * ((t + 2) & (H1T - 1)) + 8 * mt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr ((VAR & 16) != 0) {
#pragma unroll
      for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const f32x4v v = seed[4 * mt + rr];
          acc[mt][4 * rr] += v.x; acc[mt][4 * rr + 1] += v.y; acc[mt][4 * rr + 2] += v.z; acc[mt][4 * rr + 3] += v.w;
        }
    }
    // PReLU of layer 2 and the bias-free output layer
    float part;
    if constexpr (KW3B >= 0) {
      // sum_j w3_j (x_j + beta2_j min(x_j, 0)) as two packed-f32 dot products, w3 . x and (w3 beta2) . min(x, 0) -- 2 vector
      // instructions per unit instead of 3 (the caller staged w3 beta2 at float KW3B of the vectors)
      f32x2 dot[4] = {};
#pragma unroll
      for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const f32x4v w3 = vec4(kW3 + 32 * mt + 8 * rr), wb = vec4(KW3B + 32 * mt + 8 * rr);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 xa = f32x2{acc[mt][4 * rr + e], acc[mt][4 * rr + e + 1]};
            const f32x2 w3p = e ? f32x2{w3.z, w3.w} : f32x2{w3.x, w3.y}, wbp = e ? f32x2{wb.z, wb.w} : f32x2{wb.x, wb.y};
            f32x2 m;
            asm("v_min_f32 %0, 0, %1" : "=v"(m.x) : "v"(xa.x));
            asm("v_min_f32 %0, 0, %1" : "=v"(m.y) : "v"(xa.y));
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dot[e >> 1]) : "v"(xa), "v"(w3p));
            asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(dot[2 + (e >> 1)]) : "v"(m), "v"(wbp));
          }
        }
      part = ((dot[0].x + dot[0].y) + (dot[1].x + dot[1].y)) + ((dot[2].x + dot[2].y) + (dot[3].x + dot[3].y));
    } else {  // one chain, unit by unit (the fused kernel since round 4's first half)
      part = 0.0f;
#pragma unroll
      for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const f32x4v be = vec4(kBeta2 + 32 * mt + 8 * rr), w3 = vec4(kW3 + 32 * mt + 8 * rr);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xa = acc[mt][4 * rr + e];
            part = __builtin_fmaf(__builtin_fmaf(neg_part(xa), be[e], xa), w3[e], part);
          }
        }
    }
    const float other = __shfl_xor(part, 32);
    constexpr float kUnscale = 1.0f / (kSplit2Scale * kSplit2Scale);
    store(k, (part + other) * kUnscale);
    row = next;
  }
}

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
  // LDS bases (byte addresses), made opaque so that every read is `base + immediate`
  uint32_t w_lo = lds_offset_of(W2) + (uint32_t)lane * 16u;
  uint32_t w_hi = w_lo + 65536u;
  uint32_t v_at = lds_offset_of(V) + (uint32_t)g * 16u;
  asm volatile("" : "+v"(w_lo), "+v"(w_hi), "+v"(v_at));
  auto frag = [&](int t, int k) -> f16x8 {  // fragment k (0..15: [q][m][hi, lo]) of tile t
    const uint32_t base = t < 4 ? w_lo : w_hi;
    const int off = (t & 3) * 16384 + k * 1024;
    const u32x4v v = *reinterpret_cast<lds_u4_ptr>(base + off);
    return __builtin_bit_cast(f16x8, v);
  };
  auto vec4 = [&](int float_index) -> f32x4v {  // four floats of the vectors at float_index + 4 g
    return *reinterpret_cast<lds_f4_ptr>(v_at + 4 * float_index);
  };
  constexpr int kU = 0, kBeta1 = 256, kB2 = 512, kBeta2 = 640, kW3 = 768;  // Mlp2Vectors, in floats
  static_assert(offsetof(Mlp2Vectors, beta1) == 4 * kBeta1 && offsetof(Mlp2Vectors, b2) == 4 * kB2 &&
                offsetof(Mlp2Vectors, beta2) == 4 * kBeta2 && offsetof(Mlp2Vectors, w3) == 4 * kW3, "Mlp2Vectors layout");
  {  // the software pipeline (round 4, second half; the tile-phased loop of r4a-r4j: tools/rejected/nann_mlp5_tile_phased_loop.h)
    SplitPipeLds L;
    L.w_lo = w_lo; L.w_hi = w_hi; L.v_at = v_at; L.u_at = v_at; L.u_wr = 0u;  // (kU = 0: the query's u heads the vectors)
    int i_cur = 0;
    wave_mlp_split_pipeline<0, -1>(
        L, row_ptr(wave * 32 + cand), (nblk - wave + NW - 1) / NW,
        [&](int k, const float* row_k, const float*& next, bool&, float4&) {
          const int b = wave + k * NW;
          i_cur = b * 32 + cand;
          next = (b + NW < nblk) ? row_ptr(i_cur + NW * 32) : row_k;
        },
        [&](int, float score) {
          if (g == 0 && i_cur < n) scores[i_cur] = score;
        });
    return;
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
  if (NANN_RES_XSKEW > 0 && wave >= NW / 2) __builtin_amdgcn_s_sleep(NANN_RES_XSKEW);  // SIMD partners half a tile apart (see the split form)
  auto load_ub = [&](int t, float4 (&ub)[8]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      ub[rr] = *reinterpret_cast<const float4*>(&V->u[32 * t + 8 * rr + 4 * g]);
      ub[4 + rr] = *reinterpret_cast<const float4*>(&V->beta1[32 * t + 8 * rr + 4 * g]);
    }
  };
  const float* row = row_ptr(wave * 32 + cand);
  float4 pE[4], pO[4];
  load_tile(row, 0, pE);
  load_tile(row, 1, pO);
  float4 ubE[8], ubO[8];  // u / alpha1 of the even / odd tile, read underneath the previous tile's MFMAs
  load_ub(0, ubE);
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
    auto tile = [&](int t, float4 (&x)[4], float4 (&cur)[8], float4 (&nxt)[8]) {
      // h1 of this lane's 16 hidden units of tile t (register r = 4 rr + e <-> unit 32 t + 8 rr + 4 g + e)
      float h[16];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float4 u = cur[rr], al = cur[4 + rr];
        constexpr float kInv = 1.0f / kSplit2Scale;  // the table holds 2^7 P: exact both ways
#if (NANN_RES_VAR & 1)  // timing build: no layer-1 arithmetic
        h[4 * rr + 0] = x[rr].x + u.x; h[4 * rr + 1] = x[rr].y + al.x; h[4 * rr + 2] = x[rr].z; h[4 * rr + 3] = x[rr].w;
#else
        h[4 * rr + 0] = prelu(u.x + x[rr].x * kInv, al.x);
        h[4 * rr + 1] = prelu(u.y + x[rr].y * kInv, al.y);
        h[4 * rr + 2] = prelu(u.z + x[rr].z * kInv, al.z);
        h[4 * rr + 3] = prelu(u.w + x[rr].w * kInv, al.w);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#if !(NANN_RES_VAR & 2)
      {
        const bool wrap = t + 2 >= H1T;
        load_tile(wrap ? next : row, wrap ? t + 2 - H1T : t + 2, x);
      }
#endif
      load_ub((t + 1) & (H1T - 1), nxt);
      __builtin_amdgcn_sched_barrier(0);
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
      tile(t, pE, ubE, ubO);
      tile(t + 1, pO, ubO, ubE);
    }
    row = next;
    // PReLU of layer 2 and the bias-free output layer: per-lane chain over its 64 outputs (wg_score_mlp's epilogue)
    float part = 0.0f;
#if (NANN_RES_VAR & 4)  // timing build: no output chain
    part = acc2[0][0] + acc2[1][1] + acc2[2][2] + acc2[3][3];
#else
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g;
        part = __fmaf_rn(prelu(acc2[mt][r], V->beta2[m]), V->w3[m], part);
      }
#endif
    const float other = __shfl_xor(part, 32);
    const float p0 = g == 0 ? part : other, p1 = g == 0 ? other : part;
    if (g == 0 && i < n) scores[i] = p0 + p1;
  }
}

}  // namespace nann
