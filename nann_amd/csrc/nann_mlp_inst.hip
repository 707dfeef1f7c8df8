// nann_mlp_inst.hip -- MLP-scorer instantiations of the fused traversal and of the
// stand-alone scorer for ONE embedding dim (-DNANN_MLP_D=64|128|256), so that the three
// heavy objects (1024 unrolled MFMAs each) compile in parallel.
#include "nann_eval.h"

#ifndef NANN_MLP_D
#error "compile with -DNANN_MLP_D=64|128|256"
#endif
#define NANN_CAT2(a, b) a##b
#define NANN_CAT(a, b) NANN_CAT2(a, b)

namespace nann {

#if NANN_MLP_D == 128
int launch_mlp_preproject(int dt, const void* emb, long long n_rows, int d, const float* w1, float* proj, hipStream_t st) {
  if (dt != NANN_F16 && dt != NANN_BF16) return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
  if (d > 256 || d % 4) return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: d <= 256");
  const unsigned blocks = (unsigned)((n_rows + 31) / 32);
  if (dt == NANN_F16) hipLaunchKernelGGL((k_mlp_preproject<DT_F16>), dim3(blocks), dim3(256), 0, st, emb, n_rows, d, w1, proj);
  else hipLaunchKernelGGL((k_mlp_preproject<DT_BF16>), dim3(blocks), dim3(256), 0, st, emb, n_rows, d, w1, proj);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}
#endif

int NANN_CAT(launch_search_mlp_d, NANN_MLP_D)(int dt, int split, int vis, int slots, size_t lds_bytes,
                                              const SearchArgs& a, hipStream_t st) {
  constexpr int LPR = NANN_MLP_D / 8;
  if (dt != NANN_F16 && dt != NANN_BF16) return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
  if (split && vis == VIS_LDS_HASH) {  // one workgroup per CU: 16K-slot set + the weight-slice buffers
#if NANN_MLP_D <= 128
    // second mapping (nann_mlp2.h): 256 threads, 64 rows per wavefront at one wavefront per SIMD
    if (dt == NANN_F16) return launch_search_as<LPR, DT_F16, VIS_LDS_HASH, kScorerMlpSplit, kMlp2NT>(slots, lds_bytes, a, st);
    return launch_search_as<LPR, DT_BF16, VIS_LDS_HASH, kScorerMlpSplit, kMlp2NT>(slots, lds_bytes, a, st);
#else
    if (dt == NANN_F16) return launch_search_as<LPR, DT_F16, VIS_LDS_HASH, kScorerMlpSplit, kMlpNT>(slots, lds_bytes, a, st);
    return launch_search_as<LPR, DT_BF16, VIS_LDS_HASH, kScorerMlpSplit, kMlpNT>(slots, lds_bytes, a, st);
#endif
  }
  if (vis != VIS_LDS_BITMAP && vis != VIS_HBM_BITMAP) return fail(NANN_ERR_UNSUPPORTED, "MLP traversal: no kernel for this plan");
  if (split) {
    if (dt == NANN_F16) return launch_search_bitmap<LPR, DT_F16, kScorerMlpSplit, kMlpNT>(vis, slots, lds_bytes, a, st);
    return launch_search_bitmap<LPR, DT_BF16, kScorerMlpSplit, kMlpNT>(vis, slots, lds_bytes, a, st);
  }
  if (dt == NANN_F16) return launch_search_bitmap<LPR, DT_F16, NANN_SCORER_MLP, kMlpNT>(vis, slots, lds_bytes, a, st);
  return launch_search_bitmap<LPR, DT_BF16, NANN_SCORER_MLP, kMlpNT>(vis, slots, lds_bytes, a, st);
}

// evaluation-graph traversal (nann_eval.h), f32 MFMA scorer
int NANN_CAT(launch_eval_mlp_d, NANN_MLP_D)(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  constexpr int LPR = NANN_MLP_D / 8;
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_MLP, kMlpNT, false>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_MLP, kMlpNT, false>(slots, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
}

int NANN_CAT(launch_score_mlp_d, NANN_MLP_D)(int dt, int split, unsigned blocks, hipStream_t st, const MlpParams& P,
                                             const void* table, long long n_table_rows,
                                             const int32_t* indices, long long n, const float* q,
                                             float* out, OpResult* res) {
#define NANN_LAUNCH_SCORE(DT_, SPLIT_)                                                                       \
  hipLaunchKernelGGL((k_score_mlp<NANN_MLP_D, DT_, SPLIT_>), dim3(blocks), dim3(kMlpNT), 0, st, P, table, \
                     n_table_rows, indices, n, q, out, res)
#if NANN_MLP_D <= 128
#define NANN_LAUNCH_SCORE2(DT_)                                                                               \
  hipLaunchKernelGGL((k_score_mlp2<NANN_MLP_D, DT_>), dim3(blocks), dim3(kMlp2NT), 0, st, P, table, n_table_rows, \
                     indices, n, q, out, res)
  if (split && dt == NANN_F16) NANN_LAUNCH_SCORE2(DT_F16);
  else if (split && dt == NANN_BF16) NANN_LAUNCH_SCORE2(DT_BF16);
  else
#undef NANN_LAUNCH_SCORE2
#endif
  if (dt == NANN_F16 && split) NANN_LAUNCH_SCORE(DT_F16, true);
  else if (dt == NANN_F16) NANN_LAUNCH_SCORE(DT_F16, false);
  else if (dt == NANN_BF16 && split) NANN_LAUNCH_SCORE(DT_BF16, true);
  else if (dt == NANN_BF16) NANN_LAUNCH_SCORE(DT_BF16, false);
  else return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
#undef NANN_LAUNCH_SCORE
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

}  // namespace nann
