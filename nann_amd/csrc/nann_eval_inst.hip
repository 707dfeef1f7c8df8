// nann_eval_inst.hip -- slot-form L2 and attention-model instantiations of the evaluation-graph traversal
// (nann_eval.h).  The LDS form's L2 instances are their own unit (nann_eval_lds_inst.hip), the MLP ones are built with the
// other MLP kernels (nann_mlp_inst.hip).
#include "nann_eval.h"
#ifndef NANN_EVAL_DEV
#define NANN_EVAL_DEV 0  // kernel iteration (tools/build_res_variant.py ... -DNANN_EVAL_DEV=1): the 128-d f16 L2 instances only
#endif

namespace nann {

template <int LPR>
static int eval_l2_slot(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_L2, kNT, 0>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_L2, kNT, 0>(slots, a, st);
  return launch_eval_as<LPR, DT_F32, NANN_SCORER_L2, kNT, 0>(slots, a, st);
}

size_t eval_l2_lds_base() { return eval_lds_base<NANN_SCORER_L2, kNT>(); }
// every instance has at least the top-k scratch of kEvalMaxK (eval_scratch_bytes is a max over it)
size_t eval_dirty_room() { return (sizeof(TopkScratchT<kEvalMaxK>) + 255) & ~(size_t)255; }

int launch_eval_l2(int lpr, int dt, int seen_lds, int slots, const EvalArgs& a, hipStream_t st) {
  if (seen_lds) return a.n_windows > 1 ? launch_eval_l2_win(lpr, dt, slots, a, st) : launch_eval_l2_lds(lpr, dt, slots, a, st);
#if NANN_EVAL_DEV
  if (lpr != 16 || dt != NANN_F16) return fail(NANN_ERR_UNSUPPORTED, "NANN_EVAL_DEV build: 128-d f16 only");
  return launch_eval_as<16, DT_F16, NANN_SCORER_L2, kNT, 0>(slots, a, st);
#else
  switch (lpr) {
    case 8: return eval_l2_slot<8>(dt, slots, a, st);
    case 16: return eval_l2_slot<16>(dt, slots, a, st);
    case 32: return eval_l2_slot<32>(dt, slots, a, st);
    default: return eval_l2_slot<64>(dt, slots, a, st);
  }
#endif
}

int launch_eval_attn(int d, int dt, int slots, const EvalArgs& a, hipStream_t st) {
#if NANN_EVAL_DEV
  return fail(NANN_ERR_UNSUPPORTED, "NANN_EVAL_DEV build: no attention instances");
#else
  if (d == 64 && dt == NANN_F16) return launch_eval_as<8, DT_F16, kScorerAttn, kAttnNT, 0>(slots, a, st);
  if (d == 64 && dt == NANN_BF16) return launch_eval_as<8, DT_BF16, kScorerAttn, kAttnNT, 0>(slots, a, st);
  if (d == 128 && dt == NANN_F16) return launch_eval_as<16, DT_F16, kScorerAttn, kAttnNT, 0>(slots, a, st);
  if (d == 128 && dt == NANN_BF16) return launch_eval_as<16, DT_BF16, kScorerAttn, kAttnNT, 0>(slots, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d in {64, 128}, rows f16 or bf16");
#endif
}

}  // namespace nann
