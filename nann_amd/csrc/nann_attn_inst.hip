// nann_attn_inst.hip -- kernels of the reference scorer model (nann_attn.h) and their launchers.
#define NANN_ATTN_KERNELS_TU 1
#include "nann_search.h"

namespace nann {

int launch_attn_prepare(hipStream_t st, const AttnParams& P, const void* user_seq_f16, long long n_users,
                        float* kt, float* upad) {
  hipLaunchKernelGGL(k_attn_prepare, dim3((unsigned)n_users), dim3(256), 0, st, P,
                     static_cast<const uint16_t*>(user_seq_f16), kt, upad);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

int launch_score_attn(int dt, unsigned blocks, hipStream_t st, const AttnParams& P, const float* kt,
                      const float* upad, const void* table, long long n_table_rows, const int32_t* indices,
                      long long n, float* scores, long long* bad_i) {
#define NANN_ATTN_CASE(D_, DT_)                                                                      \
  hipLaunchKernelGGL((k_score_attn<D_, DT_>), dim3(blocks), dim3(kAttnNT), 0, st, P, kt, upad, table, \
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
static int launch_attn_vis(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  constexpr int LPR = D / 8;
  if (vis == VIS_LDS_HASH) return launch_search_as<LPR, DT, VIS_LDS_HASH, kScorerAttn, kAttnNT>(slots, lds_bytes, a, st);
  if (vis == VIS_LDS_BITMAP || vis == VIS_HBM_BITMAP)
    return launch_search_bitmap<LPR, DT, kScorerAttn, kAttnNT>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention traversal: no kernel for this plan");
}

int launch_search_attn(int d, int dt, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  if (d == 64 && dt == NANN_F16) return launch_attn_vis<64, DT_F16>(vis, slots, lds_bytes, a, st);
  if (d == 64 && dt == NANN_BF16) return launch_attn_vis<64, DT_BF16>(vis, slots, lds_bytes, a, st);
  if (d == 128 && dt == NANN_F16) return launch_attn_vis<128, DT_F16>(vis, slots, lds_bytes, a, st);
  if (d == 128 && dt == NANN_BF16) return launch_attn_vis<128, DT_BF16>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
}

int launch_search_attn_xproj(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  // the f32 form on the pre-projected table never reads the embedding rows: one instance serves every d and row dtype
  if (vis == VIS_LDS_HASH) return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerAttnXProj, kAttnNT>(slots, lds_bytes, a, st);
  if (vis == VIS_LDS_BITMAP || vis == VIS_HBM_BITMAP)
    return launch_search_bitmap<16, DT_F16, kScorerAttnXProj, kAttnNT>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention traversal: no kernel for this plan");
}

}  // namespace nann
