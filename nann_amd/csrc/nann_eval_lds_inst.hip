// nann_eval_lds_inst.hip -- the LDS form of the evaluation-graph traversal (search_eval_lds, nann_eval.h): L2 scorer, shards of
// up to 2^20 items.  Its own translation unit: twelve instances of the longest kernel of the library.
#include "nann_eval.h"
#ifndef NANN_EVAL_DEV
#define NANN_EVAL_DEV 0  // kernel iteration: the 128-d f16 instance only
#endif

namespace nann {

template <int LPR>
static int eval_l2_lds(int dt, int slots, const EvalArgs& a, hipStream_t st) {
  if (dt == NANN_F16) return launch_eval_as<LPR, DT_F16, NANN_SCORER_L2, kNT, 1>(slots, a, st);
  if (dt == NANN_BF16) return launch_eval_as<LPR, DT_BF16, NANN_SCORER_L2, kNT, 1>(slots, a, st);
  return launch_eval_as<LPR, DT_F32, NANN_SCORER_L2, kNT, 1>(slots, a, st);
}

int launch_eval_l2_lds(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st) {
#if NANN_EVAL_DEV
  if (lpr != 16 || dt != NANN_F16) return fail(NANN_ERR_UNSUPPORTED, "NANN_EVAL_DEV build: 128-d f16 only");
  return launch_eval_as<16, DT_F16, NANN_SCORER_L2, kNT, 1>(slots, a, st);
#else
  switch (lpr) {
    case 8: return eval_l2_lds<8>(dt, slots, a, st);
    case 16: return eval_l2_lds<16>(dt, slots, a, st);
    case 32: return eval_l2_lds<32>(dt, slots, a, st);
    default: return eval_l2_lds<64>(dt, slots, a, st);
  }
#endif
}

}  // namespace nann
