// nann_mlp_inst.hip -- MLP-scorer instantiations of the fused traversal and of the
// stand-alone scorer for ONE embedding dim (-DNANN_MLP_D=64|128|256), so that the three
// heavy objects (1024 unrolled MFMAs each) compile in parallel.
#include "nann_search.h"

#ifndef NANN_MLP_D
#error "compile with -DNANN_MLP_D=64|128|256"
#endif
#define NANN_CAT2(a, b) a##b
#define NANN_CAT(a, b) NANN_CAT2(a, b)

namespace nann {

#if NANN_COMPACT
// the compact library variant carries the L2 traversal only (its phase scratch is too small for the MLP)
int NANN_CAT(launch_search_mlp_d, NANN_MLP_D)(int, const SearchPlan&, const SearchArgs&, hipStream_t) {
  return fail(NANN_ERR_UNSUPPORTED, "compact build: L2 scorer only");
}
int NANN_CAT(launch_score_mlp_d, NANN_MLP_D)(int, unsigned, hipStream_t, const MlpParams&, const void*, long long,
                                             const int32_t*, long long, const float*, float*, OpResult*) {
  return fail(NANN_ERR_UNSUPPORTED, "compact build: L2 scorer only");
}
}  // namespace nann
#else

int NANN_CAT(launch_search_mlp_d, NANN_MLP_D)(int dt, const SearchPlan& p, const SearchArgs& a, hipStream_t st) {
  constexpr int LPR = NANN_MLP_D / 8;
  if (dt == NANN_F16) return launch_search<LPR, DT_F16, NANN_SCORER_MLP, kMlpNT>(p, a, st);
  if (dt == NANN_BF16) return launch_search<LPR, DT_BF16, NANN_SCORER_MLP, kMlpNT>(p, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
}

int NANN_CAT(launch_score_mlp_d, NANN_MLP_D)(int dt, unsigned blocks, hipStream_t st, const MlpParams& P,
                                             const void* table, long long n_table_rows,
                                             const int32_t* indices, long long n, const float* q,
                                             float* out, OpResult* res) {
  if (dt == NANN_F16)
    hipLaunchKernelGGL((k_score_mlp<NANN_MLP_D, DT_F16>), dim3(blocks), dim3(kMlpNT), 0, st, P, table,
                       n_table_rows, indices, n, q, out, res);
  else if (dt == NANN_BF16)
    hipLaunchKernelGGL((k_score_mlp<NANN_MLP_D, DT_BF16>), dim3(blocks), dim3(kMlpNT), 0, st, P, table,
                       n_table_rows, indices, n, q, out, res);
  else
    return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

}  // namespace nann
#endif
