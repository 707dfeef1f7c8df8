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

int NANN_CAT(launch_search_mlp_d, NANN_MLP_D)(int dt, int split, int vis, int slots, size_t lds_bytes,
                                              const SearchArgs& a, hipStream_t st) {
  constexpr int LPR = NANN_MLP_D / 8;
  if (dt != NANN_F16 && dt != NANN_BF16) return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
  if (split && vis == VIS_LDS_HASH) {  // one 512-thread workgroup per CU: 16K-slot set + two weight-slice buffers
    if (dt == NANN_F16) return launch_search_as<LPR, DT_F16, VIS_LDS_HASH, kScorerMlpSplit, kMlpNT>(slots, lds_bytes, a, st);
    return launch_search_as<LPR, DT_BF16, VIS_LDS_HASH, kScorerMlpSplit, kMlpNT>(slots, lds_bytes, a, st);
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
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_MLP, kMlpNT>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_MLP, kMlpNT>(slots, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
}

int NANN_CAT(launch_score_mlp_d, NANN_MLP_D)(int dt, int split, unsigned blocks, hipStream_t st, const MlpParams& P,
                                             const void* table, long long n_table_rows,
                                             const int32_t* indices, long long n, const float* q,
                                             float* out, OpResult* res) {
#define NANN_LAUNCH_SCORE(DT_, SPLIT_)                                                                       \
  hipLaunchKernelGGL((k_score_mlp<NANN_MLP_D, DT_, SPLIT_>), dim3(blocks), dim3(kMlpNT), 0, st, P, table, \
                     n_table_rows, indices, n, q, out, res)
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
