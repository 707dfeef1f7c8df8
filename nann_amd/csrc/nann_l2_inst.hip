// nann_l2_inst.hip -- L2-scorer instantiations of the fused traversal for ONE row dtype
// (-DNANN_L2_DT=0 f16 | 1 bf16 | 2 f32, -DNANN_L2_NAME=f16|bf16|f32): one object each so that
// they compile in parallel.
#include "nann_search.h"

#if !defined(NANN_L2_DT) || !defined(NANN_L2_NAME)
#error "compile with -DNANN_L2_DT=<0|1|2> -DNANN_L2_NAME=<f16|bf16|f32>"
#endif
#define NANN_CAT2(a, b) a##b
#define NANN_CAT(a, b) NANN_CAT2(a, b)

namespace nann {

int NANN_CAT(launch_search_l2_, NANN_L2_NAME)(int lpr, const SearchPlan& p, const SearchArgs& a, hipStream_t st) {
#if NANN_COMPACT
  // compact library variant: only the 512-thread kernel with the visited set in LDS exists
  if (!(p.nt == 512 && p.lds_bitmap)) return fail(NANN_ERR_UNSUPPORTED, "compact build: plan not supported");
  switch (lpr) {
    case 8: return launch_search_lds<8, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
    case 16: return launch_search_lds<16, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
    case 32: return launch_search_lds<32, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
    default: return launch_search_lds<64, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
  }
#else
  if (p.nt == 512) {  // global-bitmap variant: two half-size workgroups per CU
    switch (lpr) {
      case 8: return launch_search_global<8, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
      case 16: return launch_search_global<16, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
      case 32: return launch_search_global<32, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
      default: return launch_search_global<64, NANN_L2_DT, NANN_SCORER_L2, 512>(p, a, st);
    }
  }
  switch (lpr) {
    case 8: return launch_search<8, NANN_L2_DT, NANN_SCORER_L2, kNT>(p, a, st);
    case 16: return launch_search<16, NANN_L2_DT, NANN_SCORER_L2, kNT>(p, a, st);
    case 32: return launch_search<32, NANN_L2_DT, NANN_SCORER_L2, kNT>(p, a, st);
    default: return launch_search<64, NANN_L2_DT, NANN_SCORER_L2, kNT>(p, a, st);
  }
#endif
}

}  // namespace nann
