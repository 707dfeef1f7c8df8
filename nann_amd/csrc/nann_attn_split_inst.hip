// nann_attn_split_inst.hip -- kernels of the reference scorer model in its split-f16 form (nann_attn_split.h)
// and their launchers: per-user packing, the stand-alone scorer, the fused traversal.
#define NANN_ATTN_SPLIT_TU 1
#include <algorithm>
#include "nann_search.h"

namespace nann {

template <int D, int DT>
__global__ __launch_bounds__(kAttnNT) void k_score_attn_split(AttnParams P, const float* __restrict__ kt,
                                                              const float* __restrict__ upad, const void* table,
                                                              long long n_table_rows, const int32_t* indices,
                                                              long long n, float* scores, long long* bad_i) {
  __shared__ __attribute__((aligned(16))) float slice[kAttnSlice + kAttnVecFloats];
  const int tid = local_tid();
  constexpr int CPP = (kAttnNT / 64) * 32;
  if (indices) {  // bounds first (gather_op.cc:170-175)
    for (long long i = (long long)blockIdx.x * kAttnNT + tid; i < n; i += (long long)gridDim.x * kAttnNT) {
      const long long r = indices[i];
      if (r < 0 || r >= n_table_rows) atomicMin(reinterpret_cast<unsigned long long*>(bad_i), (unsigned long long)i);
    }
  }
  const uint4* k4 = reinterpret_cast<const uint4*>(kt);
  const uint4* u4 = reinterpret_cast<const uint4*>(upad);
  // each workgroup takes a contiguous run of passes: one call, so that the slice pipeline stays primed
  const long long passes = (n + CPP - 1) / CPP, per = (passes + gridDim.x - 1) / gridDim.x;
  const long long c0 = (long long)blockIdx.x * per * CPP;
  if (c0 >= n) return;
  const long long cnt = (n - c0) < per * CPP ? (n - c0) : per * CPP;
  if (indices) {
    wg_score_attn_split<D, DT, kAttnNT>(P, k4, u4, table, n_table_rows, indices + c0, cnt, slice, scores + c0);
  } else {  // rows c0.. of `table` itself
    wg_score_attn_split<D, DT, kAttnNT>(P, k4, u4, static_cast<const uint16_t*>(table) + (size_t)c0 * D,
                                        n_table_rows - c0, nullptr, cnt, slice, scores + c0);
  }
}

int launch_attn_prepare_split(hipStream_t st, const AttnParams& P, const void* user_seq_f16, long long n_users,
                              float* kt, float* upad) {
  hipLaunchKernelGGL(k_attn_prepare_split, dim3((unsigned)n_users), dim3(256), 0, st, P,
                     static_cast<const uint16_t*>(user_seq_f16), reinterpret_cast<uint16_t*>(kt),
                     reinterpret_cast<uint16_t*>(upad));
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

int launch_score_attn_split(int dt, unsigned blocks, hipStream_t st, const AttnParams& P, const float* kt,
                            const float* upad, const void* table, long long n_table_rows, const int32_t* indices,
                            long long n, float* scores, long long* bad_i) {
#define NANN_ATTN_CASE(D_, DT_)                                                                            \
  hipLaunchKernelGGL((k_score_attn_split<D_, DT_>), dim3(blocks), dim3(kAttnNT), 0, st, P, kt, upad, table, \
                     n_table_rows, indices, n, scores, bad_i)
  if (P.d == 64 && dt == NANN_F16) NANN_ATTN_CASE(64, DT_F16);
  else if (P.d == 64 && dt == NANN_BF16) NANN_ATTN_CASE(64, DT_BF16);
  else if (P.d == 128 && dt == NANN_F16) NANN_ATTN_CASE(128, DT_F16);
  else if (P.d == 128 && dt == NANN_BF16) NANN_ATTN_CASE(128, DT_BF16);
  else return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
#undef NANN_ATTN_CASE
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

template <int D, int DT>
static int launch_attn_split_vis(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  constexpr int LPR = D / 8;
  if (vis == VIS_LDS_HASH) return launch_search_as<LPR, DT, VIS_LDS_HASH, kScorerAttnSplit, kAttnNT>(slots, lds_bytes, a, st);
  if (vis == VIS_LDS_BITMAP || vis == VIS_HBM_BITMAP)
    return launch_search_bitmap<LPR, DT, kScorerAttnSplit, kAttnNT>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention traversal: no kernel for this plan");
}

int launch_search_attn_split(int d, int dt, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  if (d == 64 && dt == NANN_F16) return launch_attn_split_vis<64, DT_F16>(vis, slots, lds_bytes, a, st);
  if (d == 64 && dt == NANN_BF16) return launch_attn_split_vis<64, DT_BF16>(vis, slots, lds_bytes, a, st);
  if (d == 128 && dt == NANN_F16) return launch_attn_split_vis<128, DT_F16>(vis, slots, lds_bytes, a, st);
  if (d == 128 && dt == NANN_BF16) return launch_attn_split_vis<128, DT_BF16>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
}

int launch_search_attn_proj(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  // the scorer never reads the embedding table: the <16, f16> instance serves every d and row dtype
  if (vis == VIS_LDS_HASH) return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerAttnProj, kAttnNT>(slots, lds_bytes, a, st);
  if (vis == VIS_LDS_BITMAP || vis == VIS_HBM_BITMAP)
    return launch_search_bitmap<16, DT_F16, kScorerAttnProj, kAttnNT>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention traversal: no kernel for this plan");
}

int launch_search_attn_res(int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerAttnRes, kAttnNT>(slots, lds_bytes, a, st);
}

int launch_attn_preproject(int dt, const AttnParams& P, const void* emb, long long n_rows, float* proj, hipStream_t st) {
  if (dt != NANN_F16 && dt != NANN_BF16) return fail(NANN_ERR_UNSUPPORTED, "attention scorer: rows f16 or bf16");
  if (P.d <= 0 || P.d > 128) return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d <= 128");
  const long long steps = (n_rows + 15) / 16;
  const unsigned blocks = (unsigned)std::min<long long>(steps, 256 * 8);
  if (blocks == 0) return NANN_OK;
  if (dt == NANN_F16) hipLaunchKernelGGL((k_attn_preproject<DT_F16>), dim3(blocks), dim3(256), 0, st, P, emb, n_rows, proj);
  else hipLaunchKernelGGL((k_attn_preproject<DT_BF16>), dim3(blocks), dim3(256), 0, st, P, emb, n_rows, proj);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

}  // namespace nann
