// nann_mlp3.h -- the table of PRE-PROJECTED ITEM HALVES OF LAYER 1 that the MLP traversals score from (k_mlp_preproject).
// (Round 3's scorers on it -- layer 2 streamed through LDS per 256-row pass, wg_score_mlp_proj / wg_score_mlp_proj1 -- were
// retired in round 5: tools/rejected/nann_mlp3_streamed_layer2.h.  What runs now: nann_mlp5.h, all of layer 2 resident.)
//
// x = [q ; e]  ->  W1^T x + b1 = (b1 + W1q^T q) + W1e^T e.  The first bracket depends on the query only and has been
// hoisted out of the scoring loop since round 1 (once per query).  The second depends on the ITEM only: it is the
// same vector every time item e is scored, by whatever query.  nann_search therefore computes P[i] = W1e^T e_i for
// every item ONCE per (scorer, index) pair (k_mlp_preproject: f32, k-ordered fmaf chain, stored x 2^7 -- a resident
// table of 256 f32 per item, 1 GB per million items of HBM's 288), and the traversal's scorer no longer runs layer
// 1 at all: it gathers the 1 KB row of P instead of the 2 d-byte embedding row, adds the query's part, applies
// PReLU, splits into f16 hi + lo (nann_mlp2.h's arithmetic) and goes straight to layer 2 on the matrix cores.
//
// Why (profiles/r3_ubench_mfma*.txt, r3b_ubench_mlp2_a.txt): on gfx950 a SIMD does not overlap its VALU work with
// its MFMAs -- not inside a wavefront, not across the two wavefronts it hosts: every VALU instruction adds ~2.4
// cycles to the SIMD's time whatever is in the matrix pipe -- and with operands of real data the chip clocks
// 1.45-1.6 GHz under a saturated pipe.  A scoring pass is therefore (MFMA count) x 32 cycles + (everything else),
// and the only way down is fewer instructions: layer 1 was 128 of the 320 MFMAs per 32 rows, half of the weight
// slices staged through LDS, and all of the accumulator <-> vector register traffic of the operand split.  Cost:
// 4x the bytes gathered per scored row (1 KB instead of 256 B at d = 128; ~11 MB per query at ef = 128, under the
// HBM roofline at the rates this scorer reaches) and the table itself.
//
// Mapping as nann_mlp2.h: 256 threads, a wavefront owns two 32-row blocks (A, B); per hidden tile t a lane reads
// the four 16-byte pieces of its row of P that hold ITS 16 hidden units in the 32x32 C/D register layout
// (units 32 t + 8 rr + 4 g + 0..3) -- the tile arrives in the registers exactly where layer 1's accumulators would
// have been; the next tile's pieces are in flight underneath.  One barrier per tile (layer-2 slice hand-over).
// Scores: within 1e-5 of the fp32 chain like the other split forms (P is the exact-form chain's item part rounded
// once to f32).
#pragma once
#include "nann_mlp2.h"

namespace nann {

constexpr int kMlpProjWidth = 256;  // f32 per item in the pre-projected table (h1)

// P'[i][j] = 2^7 sum_k e_i[k] W1[d + k][j] (fmaf chain from 0 in ORDER_E), rows i0 .. i0 + kRows of one workgroup.
// 256 threads = hidden units; the rows of the block are staged in LDS as f32, W1's item half streams through L2.
template <int DT>
__global__ __launch_bounds__(256) void k_mlp_preproject(const void* __restrict__ emb, long long n_rows, int d,
                                                        const float* __restrict__ w1 /*[2d, 256]*/, float* __restrict__ proj) {
  constexpr int kRows = 32;
  __shared__ float e[kRows][256 + 4];  // d <= 256; +4: rows on different banks
  const int j = threadIdx.x;
  const long long i0 = (long long)blockIdx.x * kRows;
  for (int idx = threadIdx.x; idx < kRows * d; idx += 256) {
    const int r = idx / d, k = idx - r * d;
    const long long row = i0 + r < n_rows ? i0 + r : n_rows - 1;
    const uint32_t bits = static_cast<const uint16_t*>(emb)[(size_t)row * d + k];
    e[r][k] = DT == DT_F16 ? half_bits_to_float(bits) : bf16_bits_to_float(bits);
  }
  __syncthreads();
  float acc[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) acc[r] = 0.0f;
  const float* w = w1 + (size_t)d * 256 + j;  // item half: rows d .. 2d - 1
  const int hd = d >> 1;
  // ORDER_E (oracle/nann_oracle.c): k = 0, d/2, 1, d/2 + 1, ... -- the order in which the f32 MFMA of the stand-alone
  // exact scorer (wg_score_mlp) consumes a row, so that table and scorer hold the same bits
  for (int kk = 0; kk < hd; kk += 2) {
    const float wa0 = w[(size_t)kk * 256], wb0 = w[(size_t)(hd + kk) * 256];
    const float wa1 = w[(size_t)(kk + 1) * 256], wb1 = w[(size_t)(hd + kk + 1) * 256];
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const float2 ea = *reinterpret_cast<const float2*>(&e[r][kk]);
      const float2 eb = *reinterpret_cast<const float2*>(&e[r][hd + kk]);
      acc[r] = __fmaf_rn(ea.x, wa0, acc[r]);
      acc[r] = __fmaf_rn(eb.x, wb0, acc[r]);
      acc[r] = __fmaf_rn(ea.y, wa1, acc[r]);
      acc[r] = __fmaf_rn(eb.y, wb1, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r)
    if (i0 + r < n_rows) proj[(size_t)(i0 + r) * 256 + j] = acc[r] * kSplit2Scale;
}

}  // namespace nann
