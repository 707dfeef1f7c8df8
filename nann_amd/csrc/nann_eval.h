// nann_eval.h -- the evaluation graph's traversal (Model.retrieval + search_level,
// NANN_impls/nann/model.py:299-362) as ONE kernel per batch of users.
//
// It differs from the serving schedule (nann_search.h) in three places, all of them set semantics:
//   * the neighbours of a frontier are taken as a SET and scored in ascending-id order
//     (tf.unique + tf.sets.difference, :316-319);
//   * top_k keeps min(k, n) (:268) and an exhausted frontier is not an error (plain TF scoring);
//   * the next frontier = the new nodes scoring at least the worst kept result (:330-331).
// Ascending sets want a bitmap, not a list: `seen` collects the neighbours that are not in
// `visited` (one atomicOr each), and a word-order scan of `seen` IS the ascending, duplicate-free
// list -- no sort.  Both bitmaps live in the slot's HBM scratch (they stay in L2): this job runs a
// few thousand users per evaluation, it shares the scorers and the top-k with the serving kernel
// but not its LDS budget.
#pragma once
#include "nann_search.h"

namespace nann {

struct EvalArgs {
  const void* emb;
  const int64_t* item_ids;
  const int32_t* nbv[2];
  const int64_t* nbrs[2];
  const int32_t* enter;
  int n_enter;
  uint32_t n_items;
  int d;
  const float* q;
  int n_queries;
  int num_scoring[3];  // rounds per level (index = level; [2] must be 1, model.py:347)
  int top_k[3];        // kept results per level
  int topk_eval;       // rows of the outputs
  unsigned char* ws;
  unsigned long long slot_bytes;
  uint32_t bm_words;   // padded to a multiple of 4
  int cat_cap;         // entries of the result||next arrays
  int64_t* out_ids;    // [n_queries, topk_eval]
  float* out_scores;
  int32_t* out_index;
  int32_t* n_out;      // [n_queries] rows that are valid (min(topk_eval, results))
  int32_t* status;
  MlpParams mlp;
  AttnParams attn;
  const float* kt;
  const float* upad;
};

struct EvalSlot {
  uint32_t* visited;
  uint32_t* seen;
  int32_t* cat_ids;
  float* cat_sc;
  int32_t* res_ids;
  float* res_sc;
  int32_t* cand;
};

__host__ __device__ inline unsigned long long eval_slot_layout(uint32_t bm_words, int cat_cap, unsigned long long off[7]) {
  unsigned long long o = 0;
  auto put = [&](int i, unsigned long long bytes) { off[i] = o; o += (bytes + 255ull) & ~255ull; };
  put(0, 4ull * bm_words);
  put(1, 4ull * bm_words);
  put(2, 4ull * cat_cap);
  put(3, 4ull * cat_cap);
  put(4, 4ull * kMaxK);
  put(5, 4ull * kMaxK);
  put(6, 4ull * kMaxK);
  return o;
}

struct EvalScanScratch {
  uint32_t wave_tot[kNW];
  uint32_t total;
  int flags[2];
};

// exclusive prefix of v over the workgroup (thread order); *total = the sum.  Two barriers.
template <int NT>
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t v, EvalScanScratch* S, uint32_t* total) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const uint32_t inc = wave_scan_add(v);
  if (lane == 63) S->wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t t = S->wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ uint32_t ld_word(const uint32_t* p) {  // past the L1: the word may have been changed by an atomic
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// barrier between phases that hand the HBM bitmaps from atomics to plain loads / stores and back: the
// stores are in L2 before it, the L1 is dropped after it
__device__ __forceinline__ void wg_sync_mem() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int LPR, int DT, int SC, int NT>
__device__ __forceinline__ void eval_score(const EvalArgs& a, int qi, const int32_t* ids, int n, float* out,
                                           unsigned char* scratch, const float* qv, float mlp_u) {
  const int tid = local_tid();
  if constexpr (SC == NANN_SCORER_L2) {
    wg_score_l2_part<LPR, DT, NT / 64>(a.emb, a.d, ids, 0, n, qv, out, tid >> 6);
  } else if constexpr (SC == kScorerAttn) {
    wg_score_attn<LPR * 8, DT, NT>(a.attn, a.kt + (size_t)qi * 256 * kAttnLP, a.upad + (size_t)qi * kAttnLP * kAttnE,
                                   a.emb, (long long)a.n_items, ids, (long long)n,
                                   reinterpret_cast<float*>(scratch), out);
  } else {
    MlpScratch* M = reinterpret_cast<MlpScratch*>(scratch);
    wg_mlp_stage_setup<NT>(a.mlp, mlp_u, &M->v);
    wg_score_mlp<LPR * 8, 8, 4, DT, NT>(a.mlp, a.emb, a.n_items, ids, n, M, out);
  }
  __syncthreads();
}

template <int LPR, int DT, int SC, int NT>
__device__ __forceinline__ int search_eval_one(const EvalArgs& a, int qi, const EvalSlot& sv, unsigned char* scratch,
                                               float* qv, int* n_result) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  EvalScanScratch* SS = reinterpret_cast<EvalScanScratch*>(scratch);
  if constexpr (SC != kScorerAttn) {
    for (int k = tid; k < a.d; k += NT) qv[k] = a.q[(size_t)qi * a.d + k];
  }
  wg_zero_words(sv.seen, a.bm_words);
  wg_sync_mem();
  float mlp_u = 0.0f;
  if constexpr (SC == NANN_SCORER_MLP) mlp_u = wg_mlp_query_u<NT>(a.mlp, qv);

  // start level: score every enter point, keep min(k, n) (:349-353)
  const int E = a.n_enter;
  if (E <= 0) return NANN_ERR_EMPTY_SCORE_BATCH;
  eval_score<LPR, DT, SC, NT>(a, qi, a.enter, E, sv.cat_sc, scratch, qv, mlp_u);
  int n_res = min(a.top_k[2], E);
  int st = wg_topk<NT>(a.enter, sv.cat_sc, nullptr, E, n_res, nullptr, sv.res_ids, sv.res_sc, nullptr, nullptr, scratch);
  if (st) return st;

  // words of the bitmaps each thread scans: a contiguous run, so that thread order = id order
  const uint32_t per_thread = ((a.bm_words + NT - 1) / NT + 3u) & ~3u;
  const uint32_t w_lo = min((uint32_t)tid * per_thread, a.bm_words), w_hi = min(w_lo + per_thread, a.bm_words);

  for (int level = 1; level >= 0; --level) {  // search_level (:299-337)
    wg_zero_words(sv.visited, a.bm_words);
    if (tid < 2) SS->flags[tid] = 0;
    wg_sync_mem();
    // visited = idx_ep (:311); result -> front of the concat arrays; candidates = result
    for (int i = tid; i < n_res; i += NT) {
      const int32_t id = sv.res_ids[i];
      if ((uint32_t)id >= a.n_items) { SS->flags[1] = 1; continue; }
      atomicOr(&sv.visited[(uint32_t)id >> 5], 1u << (id & 31));
      sv.cand[i] = id;
      sv.cat_ids[i] = id;
      sv.cat_sc[i] = sv.res_sc[i];
    }
    wg_sync_mem();
    if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
    int n_cand = n_res;
    const int32_t* __restrict__ values = a.nbv[level];
    const int64_t* __restrict__ rs = a.nbrs[level];
    for (int it = 0; it < a.num_scoring[level]; ++it) {
      // neighbours of the candidates that are not visited -> bits of `seen` (one wavefront per row)
      __syncthreads();
      if (tid < 2) SS->flags[tid] = 0;  // (the scratch was reused since)
      __syncthreads();
      for (int i = wave; i < n_cand; i += NT / 64) {
        const int32_t c = sv.cand[i];
        const int64_t s = rs[c], e = rs[c + 1];
        for (int64_t j = s + lane; j < e; j += 64) {
          const int32_t v = values[j];
          if ((uint32_t)v >= a.n_items) { SS->flags[1] = 1; continue; }
          const uint32_t bit = 1u << (v & 31);
          if (!(ld_word(&sv.visited[(uint32_t)v >> 5]) & bit)) atomicOr(&sv.seen[(uint32_t)v >> 5], bit);
        }
      }
      wg_sync_mem();
      if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
      // ascending list of the set (:316-319), visited |= it (:321), seen = 0
      uint32_t cnt = 0;
      for (uint32_t w = w_lo; w < w_hi; w += 4) {
        const uint4 s4 = *reinterpret_cast<const uint4*>(&sv.seen[w]);
        cnt += __popc(s4.x) + __popc(s4.y) + __popc(s4.z) + __popc(s4.w);
      }
      uint32_t total;
      uint32_t at = wg_excl_scan<NT>(cnt, SS, &total);
      const int n_next = (int)total;
      if (n_res + n_next > a.cat_cap) return NANN_ERR_CAPACITY;
      if (cnt) {
        int32_t* dst = sv.cat_ids + n_res;
        for (uint32_t w = w_lo; w < w_hi; ++w) {
          uint32_t s = sv.seen[w];
          if (!s) continue;
          sv.visited[w] |= s;
          sv.seen[w] = 0u;
          while (s) {
            const int b = __ffs(s) - 1;
            s &= s - 1;
            dst[at++] = (int32_t)(w * 32u + (uint32_t)b);
          }
        }
      }
      wg_sync_mem();
      if (n_next == 0) {  // plain TF ops score an empty batch as an empty tensor: the result is cut to min(k, n), no candidate is left
        n_res = min(a.top_k[level], n_res);
        n_cand = 0;
        continue;
      }
      eval_score<LPR, DT, SC, NT>(a, qi, sv.cat_ids + n_res, n_next, sv.cat_sc + n_res, scratch, qv, mlp_u);  // :323
      const int n_cat = n_res + n_next;
      const int k = min(a.top_k[level], n_cat);
      st = wg_topk<NT>(sv.cat_ids, sv.cat_sc, nullptr, n_cat, k, nullptr, sv.res_ids, sv.res_sc, nullptr, nullptr,
                       scratch);  // :326-328
      if (st) return st;
      // next frontier: new nodes scoring at least the worst kept result, in id order (:330-331)
      const float worst = sv.res_sc[k - 1];
      uint32_t n_new = 0;
      for (int base = 0; base < n_next; base += NT) {
        const int i = base + tid;
        const bool keep = i < n_next && sv.cat_sc[n_res + i] >= worst;
        uint32_t tot;
        const uint32_t pos = n_new + wg_excl_scan<NT>(keep ? 1u : 0u, SS, &tot);
        if (keep && pos < (uint32_t)kMaxK) sv.cand[pos] = sv.cat_ids[n_res + i];
        n_new += tot;
      }
      if (n_new > (uint32_t)kMaxK) return NANN_ERR_CAPACITY;  // more ties at the threshold than a frontier holds
      __syncthreads();
      n_cand = (int)n_new;
      n_res = k;
      for (int i = tid; i < k; i += NT) {
        sv.cat_ids[i] = sv.res_ids[i];
        sv.cat_sc[i] = sv.res_sc[i];
      }
      __syncthreads();
    }
  }
  *n_result = n_res;
  return NANN_OK;
}

template <int LPR, int DT, int SC, int NT>
__global__ __launch_bounds__(NT) void k_search_eval(EvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kScratchBytes = phase_scratch<VIS_HBM_BITMAP, SC, NT>();
  unsigned char* scratch = smem;
  float* qv = reinterpret_cast<float*>(scratch + kScratchBytes);
  int* misc = reinterpret_cast<int*>(qv + kMaxD);

  unsigned long long off[7];
  eval_slot_layout(a.bm_words, a.cat_cap, off);
  unsigned char* slot = a.ws + 256 + (unsigned long long)blockIdx.x * a.slot_bytes;
  EvalSlot sv;
  sv.visited = reinterpret_cast<uint32_t*>(slot + off[0]);
  sv.seen = reinterpret_cast<uint32_t*>(slot + off[1]);
  sv.cat_ids = reinterpret_cast<int32_t*>(slot + off[2]);
  sv.cat_sc = reinterpret_cast<float*>(slot + off[3]);
  sv.res_ids = reinterpret_cast<int32_t*>(slot + off[4]);
  sv.res_sc = reinterpret_cast<float*>(slot + off[5]);
  sv.cand = reinterpret_cast<int32_t*>(slot + off[6]);
  WsHeader* hdr = reinterpret_cast<WsHeader*>(a.ws);
  const int K = a.topk_eval;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) misc[0] = (int)atomicAdd(&hdr->queue, 1u);
    __syncthreads();
    const int qi = misc[0];
    if (qi >= a.n_queries) break;
    int n_res = 0;
    const int st = search_eval_one<LPR, DT, SC, NT>(a, qi, sv, scratch, qv, &n_res);
    __syncthreads();
    const int n = st ? 0 : min(K, n_res);  // results[:topk_eval] (:358), item ids (:360)
    for (int i = threadIdx.x; i < K; i += NT) {
      const int32_t r = i < n ? sv.res_ids[i] : 0;
      a.out_ids[(size_t)qi * K + i] = i < n ? a.item_ids[r] : 0;
      if (a.out_scores) a.out_scores[(size_t)qi * K + i] = i < n ? sv.res_sc[i] : 0.0f;
      if (a.out_index) a.out_index[(size_t)qi * K + i] = r;
    }
    if (threadIdx.x == 0) {
      a.status[qi] = st;
      a.n_out[qi] = n;
    }
  }
}

template <int LPR, int DT, int SC, int NT>
inline int launch_eval_as(int slots, const EvalArgs& a, hipStream_t st) {
  auto kern = k_search_eval<LPR, DT, SC, NT>;
  const size_t lds_bytes = (size_t)phase_scratch<VIS_HBM_BITMAP, SC, NT>() + kMaxD * 4 + 256;
  if (lds_bytes > 48 * 1024)
    NANN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(slots), dim3(NT), lds_bytes, st, a);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

// instantiations: nann_eval_inst.hip (L2, attention model), nann_mlp_inst.hip (MLP, f32 MFMA)
int launch_eval_l2(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_attn(int d, int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d64(int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d128(int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d256(int dt, int slots, const EvalArgs& a, hipStream_t st);

}  // namespace nann
