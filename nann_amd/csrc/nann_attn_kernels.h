// nann_attn_kernels.h -- device code of the reference scorer model (see nann_attn.h for the model and
// the mapping).  The workgroup-level scorer (wg_score_attn) is a template any unit may include; the
// non-template kernel k_attn_prepare is only defined in the unit that sets NANN_ATTN_KERNELS_TU
// (nann_attn_inst.hip).
#pragma once
#include "nann_attn.h"
#include "nann_mlp.h"

namespace nann {

// row of the 32-unit tile that register r of lane slot s holds
__device__ __forceinline__ int cd_row(int r, int slot) { return (r & 3) + 8 * (r >> 2) + 4 * slot; }

// all threads: rows x cols floats (cols % 4 == 0) from src (row stride src_stride floats) into LDS
template <int NT>
__device__ __forceinline__ void attn_stage(float* slice, const float* __restrict__ src, int rows, int cols,
                                           int src_stride) {
  const int c4n = cols >> 2;
  __syncthreads();  // the previous slice is no longer read
  for (int f = local_tid(); f < rows * c4n; f += NT) {
    const int row = f / c4n, c4 = f - row * c4n;
    reinterpret_cast<float4*>(slice)[f] = *reinterpret_cast<const float4*>(src + (size_t)row * src_stride + c4 * 4);
  }
  __syncthreads();
}

// out[m0 + m] += sum over the KT k-tiles of `in` (tiles in0..) of W[k][32 m + unit] * in[k], for the
// MT output tiles held in the staged slice ([KT*32 rows][NC floats], row-major)
//
// The A operands travel in groups of 4 k-rows x MT tiles, the next group's LDS reads issued in front of this group's MFMAs
// and the schedule pinned group by group: left to itself hipcc hoists the reads of a whole layer (up to 256) above its
// first MFMA and spills what they occupy (the f32 attention traversal ran with 1.3 KB of scratch per lane).
template <int KT, int MT, int NC, int NIN, int NOUT>
__device__ __forceinline__ void attn_mma(const float* slice, const f32x16 (&in)[NIN], int in0,
                                         f32x16 (&out)[NOUT], int out0, int lane) {
  const int unit = lane & 31, slot = lane >> 5;
  const float* base = slice + 4 * slot * NC + unit;  // + (32 t + (r & 3) + 8 (r >> 2)) NC + 32 m
  constexpr int G = KT * 4;  // groups: (t, r >> 2)
  float a[2][4][MT];
  auto load_group = [&](int g, float (&dst)[4][MT]) {
    const int t = g >> 2, rq = g & 3;
#pragma unroll
    for (int rl = 0; rl < 4; ++rl)
#pragma unroll
      for (int m = 0; m < MT; ++m) dst[rl][m] = base[(32 * t + rl + 8 * rq) * NC + 32 * m];
  };
  load_group(0, a[0]);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g + 1 < G) load_group(g + 1, a[(g + 1) & 1]);
    const int t = g >> 2, rq = g & 3;
#pragma unroll
    for (int rl = 0; rl < 4; ++rl)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        out[out0 + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][rl][m], in[in0 + t][4 * rq + rl], out[out0 + m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int N>
__device__ __forceinline__ void attn_fill(f32x16 (&x)[N], int tile, const float* v, int slot) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[tile][r] = v ? v[32 * tile + cd_row(r, slot)] : 0.0f;
}

// x = prelu(x * scale + shift; alpha) (scale == nullptr: no batch norm)
template <int N>
__device__ __forceinline__ void attn_act(f32x16 (&x)[N], int tile, int vec_tile, const float* scale,
                                         const float* shift, const float* alpha, int slot) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = 32 * vec_tile + cd_row(r, slot);
    float v = x[tile][r];
    if (scale) v = __fmaf_rn(v, scale[j], shift[j]);
    x[tile][r] = prelu(v, alpha[j]);
  }
}

#ifdef NANN_ATTN_KERNELS_TU
// ---- per user: kt[j][l] = k_l[j] (f32 [256][64], l >= L zero) and upad[l][k] (f32 [64][64]) ---------
// One workgroup of 256 threads per user.  Same fmaf order as the oracle's dense(): bit-identical.
__global__ __launch_bounds__(256) void k_attn_prepare(AttnParams P, const uint16_t* __restrict__ user_seq_f16,
                                                      float* __restrict__ kt, float* __restrict__ upad) {
  __shared__ float u[kAttnLP * kAttnE];   // 16 KB
  __shared__ float k1[kAttnLP * 128];     // 32 KB
  const int tid = threadIdx.x;
  const size_t user = blockIdx.x;
  user_seq_f16 += user * (size_t)P.L * kAttnE;
  kt += user * (size_t)256 * kAttnLP;
  upad += user * (size_t)kAttnLP * kAttnE;
  for (int i = tid; i < kAttnLP * kAttnE; i += 256) {
    const int l = i / kAttnE;
    const float v = l < P.L ? half_bits_to_float(user_seq_f16[i]) : 0.0f;
    u[i] = v;
    upad[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < P.L * 128; i += 256) {  // model_util.py:84
    const int l = i >> 7, j = i & 127;
    float acc = P.bk1[j];
    for (int k = 0; k < kAttnE; ++k) acc = __fmaf_rn(u[l * kAttnE + k], P.wk1[k * 128 + j], acc);
    k1[i] = prelu(acc, P.ak[j]);
  }
  __syncthreads();
  for (int i = tid; i < 256 * kAttnLP; i += 256) {  // :85, stored transposed
    const int j = i >> 6, l = i & 63;
    float acc = 0.0f;
    if (l < P.L) {
      acc = P.bk2[j];
      for (int k = 0; k < 128; ++k) acc = __fmaf_rn(k1[l * 128 + k], P.wk2[k * 256 + j], acc);
    }
    kt[i] = acc;
  }
}

#endif  // NANN_ATTN_KERNELS_TU

// ---- candidates -------------------------------------------------------------------------------
// wg_score_attn: logits of candidates ids[0..n) (ids == nullptr: rows 0..n of `table`) for ONE user
// (kt / upad of that user), by all NT threads (NT/64 wavefronts x 32 candidates per pass); `slice` =
// kAttnSlice floats of LDS.  Rows outside [0, n_table_rows) are read as row 0 (the caller reports
// them).  Shared by the stand-alone scorer (k_score_attn) and the fused traversal (k_search).
//
// PROJ (round 4): `table` is the (model, index) pair's table of pre-projected item-only layers (nann_attn_proj.h:
// f32 [n, 384] = q_ x 2^4 ; (e W1e) x 2^11, f32 chains) instead of the embedding rows: q1, q_ and the rows of DNN layer 1
// that multiply e are LOOKED UP -- 608 instead of 1632 v_mfma_f32_32x32x2_f32 per 32 candidates, 6 staged slices instead
// of 24 (the keys four q_ tiles per slice).  Layer 1 then sums b1 + (e W1e) first and the attention half behind it (the
// f32 form on rows: b1, attention half, e half): the same 1e-5 agreement with the oracle every attention test holds.
//
// RES (with PROJ, the 16K-slot hash plan of the fused traversal): everything a scoring call reads is resident in LDS --
// the user's keys (f32 [256][64] = 64 KB) in the place of the visited set, which the caller parks around the call
// (nann_mlp5.h's scheme), the padded sequence, W1's attention rows, W2 and W3 (88 KB) in the scratch -- loaded once per
// call; no staged slice, no barrier inside the call, every wavefront at its own pace.
template <int D, int DT, int NT, bool PROJ = false, bool RES = false>
__device__ __forceinline__ void wg_score_attn(const AttnParams& P, const float* __restrict__ kt,
                                              const float* __restrict__ upad, const void* table,
                                              long long n_table_rows, const int32_t* indices, long long n,
                                              float* slice, float* scores, float* keys = nullptr) {
  static_assert(!RES || PROJ, "the resident form runs on the pre-projected table");
  static_assert(D == 64 || D == 128, "item embedding dim");
  static_assert(DT == DT_F16 || DT == DT_BF16, "item rows are f16 or bf16");
  static_assert(NT == kAttnNT, "attn_stage strides by kAttnNT");
  constexpr int ET = D / 32;  // tiles of the candidate row
  constexpr int kProjW = 384;  // (= kAttnProjWidth, nann_attn_proj.h)
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, slot = lane >> 5;
  constexpr int CPP = (NT / 64) * 32;
  const float inv_sqrt_dk = 1.0f / sqrtf(256.0f);  // model_util.py:89-91
  // PROJ: W2 and W3 (40 KB) stay in LDS for the whole call, behind the slice and the split form's vectors -- the part of
  // the attention kernels' scratch (kAttnScratch, nann_search.h) no other phase of the traversal writes
  float* res_up = slice;                  // RES: [upad 16 KB | W1a 32 KB | W2 32 KB | W3 8 KB] = kAttnXResFloats
  float* res_w1a = slice + kAttnLP * kAttnE;
  float* res_w2 = RES ? res_w1a + kAttnE * 128 : slice + kAttnSlice + kAttnVecFloats;
  float* res_w3 = res_w2 + 128 * 64;
  if constexpr (RES) {
    __syncthreads();  // the caller is done with the set and the scratch
    for (int f = tid; f < 256 * kAttnLP / 4; f += NT) reinterpret_cast<float4*>(keys)[f] = reinterpret_cast<const float4*>(kt)[f];
    for (int f = tid; f < kAttnLP * kAttnE / 4; f += NT) reinterpret_cast<float4*>(res_up)[f] = reinterpret_cast<const float4*>(upad)[f];
    for (int f = tid; f < kAttnE * 128 / 4; f += NT) reinterpret_cast<float4*>(res_w1a)[f] = reinterpret_cast<const float4*>(P.w1)[f];
  }
  if constexpr (PROJ) {
    for (int f = tid; f < 128 * 64 / 4; f += NT) reinterpret_cast<float4*>(res_w2)[f] = reinterpret_cast<const float4*>(P.w2)[f];
    for (int f = tid; f < 64 * 32 / 4; f += NT) reinterpret_cast<float4*>(res_w3)[f] = reinterpret_cast<const float4*>(P.w3)[f];
  }
  if constexpr (RES) __syncthreads();
  for (long long c0 = 0; c0 < n; c0 += CPP) {
    if (RES && c0 + wave * 32 >= n) break;  // (no barrier below: a wavefront without rows is done)
    const long long i = c0 + wave * 32 + cand;
    const long long ic = i < n ? i : n - 1;
    const long long rid = indices ? (long long)indices[ic] : ic;
    const size_t row = (rid >= 0 && rid < n_table_rows) ? (size_t)rid : 0u;
    f32x16 e[ET];
    f32x16 q1[4];
    const float* prow = PROJ ? static_cast<const float*>(table) + row * kProjW + 4 * slot : nullptr;
    if constexpr (!PROJ) {
    // the candidate row straight into the C/D layout: 4 consecutive elements per (tile, r >> 2)
    {
      const uint16_t* src = static_cast<const uint16_t*>(table) + row * D;
#pragma unroll
      for (int t = 0; t < ET; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint2 v = *reinterpret_cast<const uint2*>(src + 32 * t + 8 * g + 4 * slot);
          const uint32_t w[2] = {v.x, v.y};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t h = (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu);
            e[t][4 * g + k] = DT == DT_F16 ? half_bits_to_float(h) : bf16_bits_to_float(h);
          }
        }
    }
    // ---- q1 = prelu(e Wq1 + bq1) : [128] = 4 tiles; Wq1 staged in two column halves
#pragma unroll
    for (int m = 0; m < 4; ++m) attn_fill(q1, m, P.bq1, slot);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      attn_stage<kAttnNT>(slice, P.wq1 + 64 * half, D, 64, 128);
      attn_mma<ET, 2, 64>(slice, e, 0, q1, 2 * half, lane);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) attn_act(q1, m, m, nullptr, nullptr, P.aq, slot);
    }
    // ---- attention logits, q_ tile by q_ tile: att[l] += sum_{j in tile} q_[j] k_l[j]
    f32x16 att[2];
    attn_fill(att, 0, nullptr, slot);
    attn_fill(att, 1, nullptr, slot);
    if constexpr (PROJ) {
      float4 nx[4];  // the next q_ tile of the row, in flight under this tile's MFMAs
#pragma unroll
      for (int g = 0; g < 4; ++g) nx[g] = *reinterpret_cast<const float4*>(prow + 8 * g);
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // the keys of four q_ tiles per slice: kt[128 half .., :] = 32 KB
        if constexpr (!RES) attn_stage<kAttnNT>(slice, kt + (size_t)128 * half * kAttnLP, 128, kAttnLP, kAttnLP);
        const float* kslice = RES ? keys + (size_t)128 * half * kAttnLP : slice;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          f32x16 qt[1];
          constexpr float kInv = 1.0f / 16.0f;  // the table holds q_ x 2^4; the tile arrives in the C/D layout: four floats per (lane, g)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            qt[0][4 * g + 0] = nx[g].x * kInv; qt[0][4 * g + 1] = nx[g].y * kInv; qt[0][4 * g + 2] = nx[g].z * kInv; qt[0][4 * g + 3] = nx[g].w * kInv;
          }
          const int tn = 4 * half + tt + 1;
          if (tn < 8) {
#pragma unroll
            for (int g = 0; g < 4; ++g) nx[g] = *reinterpret_cast<const float4*>(prow + 32 * tn + 8 * g);
          }
          attn_mma<1, 2, kAttnLP>(kslice + tt * 32 * kAttnLP, qt, 0, att, 0, lane);
        }
      }
    } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x16 qt[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) qt[0][r] = P.bq2[32 * t + cd_row(r, slot)];
      attn_stage<kAttnNT>(slice, P.wq2 + 32 * t, 128, 32, 256);          // Wq2[:, 32t..32t+32): 16 KB
      attn_mma<4, 1, 32>(slice, q1, 0, qt, 0, lane);
      attn_stage<kAttnNT>(slice, kt + (size_t)32 * t * kAttnLP, 32, kAttnLP, kAttnLP);  // kt[32t.., :]: 8 KB
      attn_mma<1, 2, kAttnLP>(slice, qt, 0, att, 0, lane);
    }
    }
    // ---- softmax over the L positions (:93); positions >= L are padding of the layout, not of the sequence
    float mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int l = 32 * p + cd_row(r, slot);
        att[p][r] = l < P.L ? att[p][r] * inv_sqrt_dk : -INFINITY;
        mx = fmaxf(mx, att[p][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        att[p][r] = expf(att[p][r] - mx);
        sum += att[p][r];
      }
    sum += __shfl_xor(sum, 32);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) att[p][r] = att[p][r] / sum;
    // ---- a = sum_l p_l u_l (:95, model.py:204-206): [64] = 2 tiles, K = positions
    f32x16 x[2];
    attn_fill(x, 0, nullptr, slot);
    attn_fill(x, 1, nullptr, slot);
    if constexpr (RES) {
      attn_mma<2, 2, kAttnE>(res_up, att, 0, x, 0, lane);
    } else {
      attn_stage<kAttnNT>(slice, upad, kAttnLP, kAttnE, kAttnE);  // 16 KB
      attn_mma<2, 2, kAttnE>(slice, att, 0, x, 0, lane);
    }
    // ---- DNN layer 1 on [a ; e] (model.py:211-214): [128] = 4 tiles
    f32x16 h1[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) attn_fill(h1, m, P.b1, slot);
    if constexpr (PROJ) {  // + the row's (e W1e), looked up
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(prow + 256 + 32 * m + 8 * g);
          constexpr float kInv = 1.0f / 2048.0f;  // x 2^11 in the table
          h1[m][4 * g + 0] += v.x * kInv; h1[m][4 * g + 1] += v.y * kInv; h1[m][4 * g + 2] += v.z * kInv; h1[m][4 * g + 3] += v.w * kInv;
        }
    }
    if constexpr (RES) {
      attn_mma<2, 4, 128>(res_w1a, x, 0, h1, 0, lane);
    } else {
      attn_stage<kAttnNT>(slice, P.w1, kAttnE, 128, 128);  // rows of a: 32 KB
      attn_mma<2, 4, 128>(slice, x, 0, h1, 0, lane);
    }
    if constexpr (!PROJ) {
#pragma unroll
    for (int part = 0; part < ET / 2; ++part) {  // rows of e, 64 at a time
      attn_stage<kAttnNT>(slice, P.w1 + (size_t)(kAttnE + 64 * part) * 128, 64, 128, 128);
      attn_mma<2, 4, 128>(slice, e, 2 * part, h1, 0, lane);
    }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) attn_act(h1, m, m, P.s1, P.t1, P.a1, slot);
    // ---- layer 2: [64] = 2 tiles
    f32x16 h2[2];
    attn_fill(h2, 0, P.b2, slot);
    attn_fill(h2, 1, P.b2, slot);
    if constexpr (PROJ) {
      attn_mma<4, 2, 64>(res_w2, h1, 0, h2, 0, lane);
    } else {
      attn_stage<kAttnNT>(slice, P.w2, 128, 64, 64);  // 32 KB
      attn_mma<4, 2, 64>(slice, h1, 0, h2, 0, lane);
    }
    attn_act(h2, 0, 0, P.s2, P.t2, P.a2, slot);
    attn_act(h2, 1, 1, P.s2, P.t2, P.a2, slot);
    // ---- layer 3: [32] = 1 tile
    f32x16 h3[1];
    attn_fill(h3, 0, P.b3, slot);
    if constexpr (PROJ) {
      attn_mma<2, 1, 32>(res_w3, h2, 0, h3, 0, lane);
    } else {
      attn_stage<kAttnNT>(slice, P.w3, 64, 32, 32);  // 8 KB
      attn_mma<2, 1, 32>(slice, h2, 0, h3, 0, lane);
    }
    attn_act(h3, 0, 0, P.s3, P.t3, P.a3, slot);
    // ---- logit = h3 . w4 (no bias, :218-219): each lane its 16 units, then the other k-slot
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = __fmaf_rn(h3[0][r], P.w4[cd_row(r, slot)], part);
    part += __shfl_xor(part, 32);
    if (slot == 0 && i < n) scores[i] = part;
  }
  __syncthreads();
}

template <int D, int DT>
__global__ __launch_bounds__(kAttnNT) void k_score_attn(AttnParams P, const float* __restrict__ kt,
                                                        const float* __restrict__ upad, const void* table,
                                                        long long n_table_rows, const int32_t* indices,
                                                        long long n, float* scores, long long* bad_i) {
  __shared__ __attribute__((aligned(16))) float slice[kAttnSlice];
  const int tid = local_tid();
  constexpr int CPP = (kAttnNT / 64) * 32;
  if (indices) {  // bounds first (gather_op.cc:170-175)
    for (long long i = (long long)blockIdx.x * kAttnNT + tid; i < n; i += (long long)gridDim.x * kAttnNT) {
      const long long r = indices[i];
      if (r < 0 || r >= n_table_rows) atomicMin(reinterpret_cast<unsigned long long*>(bad_i), (unsigned long long)i);
    }
  }
  for (long long c0 = (long long)blockIdx.x * CPP; c0 < n; c0 += (long long)gridDim.x * CPP) {
    const long long cnt = (n - c0) < CPP ? (n - c0) : CPP;
    if (indices) {
      wg_score_attn<D, DT, kAttnNT>(P, kt, upad, table, n_table_rows, indices + c0, cnt, slice, scores + c0);
    } else {  // rows c0.. of `table` itself
      wg_score_attn<D, DT, kAttnNT>(P, kt, upad, static_cast<const uint16_t*>(table) + (size_t)c0 * D,
                                    n_table_rows - c0, nullptr, cnt, slice, scores + c0);
    }
  }
}

}  // namespace nann
