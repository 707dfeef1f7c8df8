// nann_eval_inst.hip -- L2 and attention-model instantiations of the evaluation-graph traversal
// (nann_eval.h); the MLP ones are built with the other MLP kernels (nann_mlp_inst.hip).
#include "nann_eval.h"

namespace nann {

template <int LPR>
static int eval_l2(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_L2, kNT>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_L2, kNT>(slots, a, st);
  return launch_eval_as<LPR, DT_F32, NANN_SCORER_L2, kNT>(slots, a, st);
}

int launch_eval_l2(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st) {
  switch (lpr) {
    case 8: return eval_l2<8>(dt, slots, a, st);
    case 16: return eval_l2<16>(dt, slots, a, st);
    case 32: return eval_l2<32>(dt, slots, a, st);
    default: return eval_l2<64>(dt, slots, a, st);
  }
}

int launch_eval_attn(int d, int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (d == 64 && dt == NANN_F16) return launch_eval_as<8, DT_F16, kScorerAttn, kAttnNT>(slots, a, st);
  if (d == 64 && dt == NANN_BF16) return launch_eval_as<8, DT_BF16, kScorerAttn, kAttnNT>(slots, a, st);
  if (d == 128 && dt == NANN_F16) return launch_eval_as<16, DT_F16, kScorerAttn, kAttnNT>(slots, a, st);
  if (d == 128 && dt == NANN_BF16) return launch_eval_as<16, DT_BF16, kScorerAttn, kAttnNT>(slots, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
}

}  // namespace nann
