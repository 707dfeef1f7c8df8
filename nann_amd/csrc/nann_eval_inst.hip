// nann_eval_inst.hip -- L2 and attention-model instantiations of the evaluation-graph traversal
// (nann_eval.h); the MLP ones are built with the other MLP kernels (nann_mlp_inst.hip).
#include "nann_eval.h"
#ifndef NANN_EVAL_DEV
#define NANN_EVAL_DEV 0
#endif

namespace nann {

template <int LPR, bool SEEN_LDS>
static int eval_l2(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_L2, kNT, SEEN_LDS>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_L2, kNT, SEEN_LDS>(slots, a, st);
  return launch_eval_as<LPR, DT_F32, NANN_SCORER_L2, kNT, SEEN_LDS>(slots, a, st);
}

size_t eval_l2_lds_base() { return eval_lds_base<NANN_SCORER_L2, kNT>(); }
// every instance has at least the top-k scratch of kEvalMaxK (eval_scratch_bytes is a max over it)
size_t eval_dirty_room() { return (sizeof(TopkScratchT<kEvalMaxK>) + 255) & ~(size_t)255; }

int launch_eval_l2(int lpr, int dt, int seen_lds, int slots, const EvalArgs& a, hipStream_t st) {
#if NANN_EVAL_DEV  // kernel iteration (tools/build_res_variant.py ... -DNANN_EVAL_DEV=1): the 128-d f16 instances only, a fifth of the compile
  if (lpr != 16 || dt != NANN_F16) return fail(NANN_ERR_UNSUPPORTED, "NANN_EVAL_DEV build: 128-d f16 only");
  return seen_lds ? launch_eval_as<16, DT_F16, NANN_SCORER_L2, kNT, true>(slots, a, st)
                  : launch_eval_as<16, DT_F16, NANN_SCORER_L2, kNT, false>(slots, a, st);
#endif
#define NANN_EVAL_L2(LPR_) return seen_lds ? eval_l2<LPR_, true>(dt, slots, a, st) : eval_l2<LPR_, false>(dt, slots, a, st)
  switch (lpr) {
    case 8: NANN_EVAL_L2(8);
    case 16: NANN_EVAL_L2(16);
    case 32: NANN_EVAL_L2(32);
    default: NANN_EVAL_L2(64);
  }
#undef NANN_EVAL_L2
}

int launch_eval_attn(int d, int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (d == 64 && dt == NANN_F16) return launch_eval_as<8, DT_F16, kScorerAttn, kAttnNT, false>(slots, a, st);
  if (d == 64 && dt == NANN_BF16) return launch_eval_as<8, DT_BF16, kScorerAttn, kAttnNT, false>(slots, a, st);
  if (d == 128 && dt == NANN_F16) return launch_eval_as<16, DT_F16, kScorerAttn, kAttnNT, false>(slots, a, st);
  if (d == 128 && dt == NANN_BF16) return launch_eval_as<16, DT_BF16, kScorerAttn, kAttnNT, false>(slots, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
}

}  // namespace nann
