// nann_mlp_res_inst.hip -- the traversal with the MLP scorer's layer 2 resident in LDS (nann_mlp5.h): split-f16 and
// exact f32, on the 16K-slot hash-set plan and on the HBM-bitmap plan (wide beams, large shards, and the rerun of
// queries the set handed back).  An object of its own so that the four kernels compile next to the others.
#include "nann_search.h"

namespace nann {

int launch_search_mlp_res(int exact, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  // the scorers never read the embedding table: one instance per (precision, plan) serves every d and row dtype
  if (vis == VIS_LDS_HASH) {
    if (exact) return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerMlpXRes, 512>(slots, lds_bytes, a, st);
    return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerMlpRes, 512>(slots, lds_bytes, a, st);
  }
  if (vis != VIS_HBM_BITMAP) return fail(NANN_ERR_UNSUPPORTED, "MLP traversal with resident layer 2: hash-set or HBM-bitmap plan");
  if (exact) return launch_search_as<16, DT_F16, VIS_HBM_BITMAP, kScorerMlpXRes, 512>(slots, lds_bytes, a, st);
  return launch_search_as<16, DT_F16, VIS_HBM_BITMAP, kScorerMlpRes, 512>(slots, lds_bytes, a, st);
}

}  // namespace nann
