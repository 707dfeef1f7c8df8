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

#define NANN_LAUNCH_L2 NANN_CAT(launch_l2_as_, NANN_L2_NAME)  // (one name per row dtype: two dtypes may share a translation unit)
template <int LPR>
static int NANN_LAUNCH_L2(int vis, int nt, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  if (vis == VIS_LDS_HASH && nt == 512)  // two queries per CU
    return launch_search_as<LPR, NANN_L2_DT, VIS_LDS_HASH, NANN_SCORER_L2, 512>(slots, lds_bytes, a, st);
  if (vis == VIS_LDS_HASH32 && nt == kNT)  // wide beams: one query per CU, 32K-slot set
    return launch_search_as<LPR, NANN_L2_DT, VIS_LDS_HASH32, NANN_SCORER_L2, kNT>(slots, lds_bytes, a, st);
  if ((vis == VIS_LDS_BITMAP || vis == VIS_HBM_BITMAP) && nt == kNT)
    return launch_search_bitmap<LPR, NANN_L2_DT, NANN_SCORER_L2, kNT>(vis, slots, lds_bytes, a, st);
  return fail(NANN_ERR_UNSUPPORTED, "L2 traversal: no kernel for this plan");
}

int NANN_CAT(launch_search_l2_, NANN_L2_NAME)(int lpr, int vis, int nt, int slots, size_t lds_bytes,
                                              const SearchArgs& a, hipStream_t st) {
  switch (lpr) {
    case 8: return NANN_LAUNCH_L2<8>(vis, nt, slots, lds_bytes, a, st);
    case 16: return NANN_LAUNCH_L2<16>(vis, nt, slots, lds_bytes, a, st);
    case 32: return NANN_LAUNCH_L2<32>(vis, nt, slots, lds_bytes, a, st);
    default: return NANN_LAUNCH_L2<64>(vis, nt, slots, lds_bytes, a, st);
  }
}

#undef NANN_LAUNCH_L2

}  // namespace nann
