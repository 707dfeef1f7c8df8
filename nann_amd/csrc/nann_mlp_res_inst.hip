// nann_mlp_res_inst.hip -- the traversal with the MLP scorer's layer 2 resident in LDS (nann_mlp5.h): split-f16 and
// exact f32, on the 16K-slot hash-set plan and on the HBM-bitmap plan (wide beams, large shards, and the rerun of
// queries the set handed back).  An object of its own so that the four kernels compile next to the others.
#include <cstdlib>

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

int launch_search_mlp_phase(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  // the traversal stages read no embedding row and score nothing: one instance per set size (16K slots: two 512-thread
  // workgroups per CU; 32K slots, wide beams: one of 1024)
  if (vis == VIS_LDS_HASH32) return launch_search_as<16, DT_F16, VIS_LDS_HASH32, kScorerMlpPhase, kNT>(slots, lds_bytes, a, st);
  return launch_search_as<16, DT_F16, VIS_LDS_HASH, kScorerMlpPhase, 512>(slots, lds_bytes, a, st);
}

static void phase_offsets(const SearchArgs& a, unsigned long long off[9]) {
  slot_layout(a.max_cand, a.max_raw, a.pool_cap, 0u, off);  // (offsets do not depend on the last region's size)
}

int launch_mlp_phase_score(int exact, const SearchArgs& a, int round, int workgroups, hipStream_t st) {
  if (a.n_queries > kPhaseChunk) return fail(NANN_ERR_BAD_ARGUMENT, "phased MLP traversal: chunks of at most 1024 queries");
  unsigned long long off[9];
  phase_offsets(a, off);
  PhaseScoreArgs p;
  p.ws = a.ws; p.slot_bytes = a.slot_bytes;
  p.off_cand_ids = off[0]; p.off_cand_scores = off[1]; p.off_state = off[8];
  p.enter = a.enter; p.proj = a.proj; p.n_items = a.n_items; p.n_queries = a.n_queries;
  p.round = round; p.mlp = a.mlp; p.dry = 0;
  auto launch = [&](auto kern) -> int {
    NANN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPhaseScoreLds));
    hipLaunchKernelGGL(kern, dim3((unsigned)workgroups), dim3(512), kPhaseScoreLds, st, p);
    NANN_HIP_TRY(hipGetLastError());
    return NANN_OK;
  };
  const int rc = exact ? launch(k_mlp_phase_score<true>) : launch(k_mlp_phase_score<false>);
  // timing only (tools/gpu_r4.sh phase_vars): the same lists once more through a dry launch -- of the timing build's
  // reduced kernel when there is one (NANN_PHASE_VAR, nann_mlp6.h)
  static const bool shadow = [] { const char* e = std::getenv("NANN_PHASE_SHADOW"); return e && e[0] == '1'; }();
  if (rc || !shadow) return rc;
  p.dry = 1;
  return exact ? launch(k_mlp_phase_score<true>) : launch(k_mlp_phase_score<false, NANN_PHASE_VAR>);
}

}  // namespace nann
