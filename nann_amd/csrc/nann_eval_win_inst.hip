// nann_eval_win_inst.hip -- the LDS form of the evaluation-graph traversal sweeping the id space in windows (search_eval_lds<.., MULTI>,
// nann_eval.h): L2 scorer, shards of ~1 M to ~8 M items.  Its own translation unit, like the one-window instances.
#include "nann_eval.h"
#ifndef NANN_EVAL_DEV
#define NANN_EVAL_DEV 0  // kernel iteration: the 128-d f16 instance only
#endif

namespace nann {

template <int LPR>
static int eval_l2_win(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_L2, kNT, 2>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_L2, kNT, 2>(slots, a, st);
  return launch_eval_as<LPR, DT_F32, NANN_SCORER_L2, kNT, 2>(slots, a, st);
}

int launch_eval_l2_win(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st) {
#if NANN_EVAL_DEV
  if (lpr != 16 || dt != NANN_F16) return fail(NANN_ERR_UNSUPPORTED, "NANN_EVAL_DEV build: 128-d f16 only");
  return launch_eval_as<16, DT_F16, NANN_SCORER_L2, kNT, 2>(slots, a, st);
#else
  switch (lpr) {
    case 8: return eval_l2_win<8>(dt, slots, a, st);
    case 16: return eval_l2_win<16>(dt, slots, a, st);
    case 32: return eval_l2_win<32>(dt, slots, a, st);
    default: return eval_l2_win<64>(dt, slots, a, st);
  }
#endif
}

}  // namespace nann
