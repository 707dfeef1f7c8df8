// nann_search.h -- the fused traversal kernel (build_opt_graph.py:109-149) and its launch
// helper, shared by the translation units that instantiate it (nann_hip.hip: L2 scorer;
// nann_mlp_inst.hip: MLP scorer, one object per embedding dim so they compile in parallel).
#pragma once
#ifndef NANN_REPEAT_SCORE
#define NANN_REPEAT_SCORE 0  // measurement builds only
#endif
#ifndef NANN_REPEAT_TOPK
#define NANN_REPEAT_TOPK 0
#endif
#include <type_traits>
#include "../../include/nann_hip.h"
#include "nann_device.h"
#include "nann_mlp.h"
#include "nann_mlp2.h"
#include "nann_mlp3.h"
#include "nann_mlp5.h"
#include "nann_mlp6.h"
#include "nann_attn_kernels.h"
#include "nann_attn_split.h"
#include "nann_attn_proj.h"

#include <string>

namespace nann {

// small result block shared between a kernel and the host for calls that
// return data-dependent counts (they synchronise anyway)
struct OpResult {
  long long n_out;
  long long n_out_splits;
  long long bad_i;
  int code;  // ragged validation code 1/2/3
  int err;   // nann_status
};


int fail(int code, const std::string& msg);  // sets nann_last_error(); defined in nann_hip.hip

#define NANN_HIP_TRY(expr)                                                                 \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ::nann::fail(NANN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// stand-alone MLP scorer: one query vector, n rows; each workgroup takes passes of 256 rows
template <int D, int DT, bool SPLIT>
__global__ __launch_bounds__(kMlpNT) void k_score_mlp(MlpParams P, const void* table, long long n_table_rows,
                                                      const int32_t* indices, long long n, const float* qv,
                                                      float* scores, OpResult* res) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[SPLIT ? sizeof(MlpSplitScratch) : sizeof(MlpScratch)];
  typedef typename std::conditional<SPLIT, MlpSplitScratch, MlpScratch>::type Scratch;
  Scratch* S = reinterpret_cast<Scratch*>(scratch);
  constexpr int CPP = (kMlpNT / 64) * 32;
  if (indices) {  // bounds first (gather_op.cc:170-175)
    for (long long i = (long long)blockIdx.x * kMlpNT + threadIdx.x; i < n; i += (long long)gridDim.x * kMlpNT) {
      const long long r = indices[i];
      if (r < 0 || r >= n_table_rows)
        atomicMin(reinterpret_cast<unsigned long long*>(&res->bad_i), (unsigned long long)i);
    }
  }
  if constexpr (SPLIT) wg_mlp_query_setup<kMlpNT>(P, qv, &S->v, kSplitWScale, kSplitWScale * kSplitHScale, kSplitHScale / kSplitWScale);
  else wg_mlp_query_setup<kMlpNT>(P, qv, &S->v);
  // each workgroup takes ONE contiguous run of passes: the slice pipeline and the row prefetch stay primed
  const long long passes = (n + CPP - 1) / CPP, per = (passes + gridDim.x - 1) / gridDim.x;
  const long long c0 = (long long)blockIdx.x * per * CPP;
  if (c0 < n) {
    const int cnt = (int)((n - c0) < per * CPP ? (n - c0) : per * CPP);
    const void* tab = indices ? table : static_cast<const char*>(table) + (size_t)c0 * D * (DT == DT_F32 ? 4 : 2);
    const uint32_t n_tab = indices ? (uint32_t)n_table_rows : (uint32_t)(n_table_rows - c0);  // rows c0.. of `table` itself
    const int32_t* idx = indices ? indices + c0 : nullptr;
    if constexpr (SPLIT) wg_score_mlp_split<D, 8, 4, DT, kMlpNT>(P, tab, n_tab, idx, cnt, S, scores + c0);
    else wg_score_mlp<D, 8, 4, DT, kMlpNT>(P, tab, n_tab, idx, cnt, S, scores + c0);
  }
}

// the split-f16 scorer's second mapping (nann_mlp2.h: 4 wavefronts x 64 rows, d <= 128) as the stand-alone op
template <int D, int DT>
__global__ __launch_bounds__(kMlp2NT, 1) void k_score_mlp2(MlpParams P, const void* table, long long n_table_rows,
                                                          const int32_t* indices, long long n, const float* qv,
                                                          float* scores, OpResult* res) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(Mlp2Scratch<D>)];
  Mlp2Scratch<D>* S = reinterpret_cast<Mlp2Scratch<D>*>(scratch);
  constexpr int CPP = kMlp2NT;  // 4 wavefronts x 64 rows
  if (indices) {  // bounds first (gather_op.cc:170-175)
    for (long long i = (long long)blockIdx.x * kMlp2NT + threadIdx.x; i < n; i += (long long)gridDim.x * kMlp2NT) {
      const long long r = indices[i];
      if (r < 0 || r >= n_table_rows)
        atomicMin(reinterpret_cast<unsigned long long*>(&res->bad_i), (unsigned long long)i);
    }
  }
  wg_mlp2_stage_setup<kMlp2NT>(P, wg_mlp_query_u<kMlp2NT>(P, qv), &S->v);
  // each workgroup takes ONE contiguous run of passes: the slice pipeline and the row prefetch stay primed
  const long long passes = (n + CPP - 1) / CPP, per = (passes + gridDim.x - 1) / gridDim.x;
  const long long c0 = (long long)blockIdx.x * per * CPP;
  if (c0 < n) {
    const int cnt = (int)((n - c0) < per * CPP ? (n - c0) : per * CPP);
    const void* tab = indices ? table : static_cast<const char*>(table) + (size_t)c0 * D * 2;
    const uint32_t n_tab = indices ? (uint32_t)n_table_rows : (uint32_t)(n_table_rows - c0);
    wg_score_mlp_split2<D, DT>(P, tab, n_tab, indices ? indices + c0 : nullptr, cnt, S, scores + c0);
  }
}

// ---------------------------------------------------------------------------
// the fused traversal (build_opt_graph.py:109-149)
struct SearchArgs {
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
  int t[6];                // level_topn of the launch; with tq: the per-launch maxima (workspace, plan, output row stride t[5])
  const int32_t* tq;       // optional device i32[n_queries, 6]: level_topn PER QUERY (the reference feeds it per request,
                           // build_opt_graph.py:75,151-159); each entry in [0, t[i]], else the query fails with BAD_ARGUMENT
  unsigned char* ws;
  unsigned long long slot_bytes;
  uint32_t bm_words;  // padded to a multiple of 4
  int max_cand, max_raw, pool_cap;
  int64_t* out_ids;
  float* out_scores;
  int32_t* out_index;
  int32_t* status;
  int32_t* counters;
  long long* phase_ticks;  // optional [n_queries, NANN_NUM_PHASES] shader-clock ticks
  MlpParams mlp;           // NANN_SCORER_MLP only
  const float* proj;       // kScorerMlpProj: the pre-projected item half of layer 1, f32 [n_items, 256] (nann_mlp3.h);
                           // kScorerMlpRes / kScorerMlpXRes: the same table (nann_mlp5.h);
                           // kScorerAttnProj: the item-only layers of the attention model, f32 [n_items, 384] (nann_attn_proj.h)
  AttnParams attn;         // kScorerAttn only
  const float* kt;         //   per-query projected keys f32 [n_queries, 256, 64] (k_attn_prepare)
  const float* upad;       //   per-query padded sequence f32 [n_queries, 64, 64]
  int id_bits;             // VIS_LDS_HASH*: bits of the shard's id space (set entries are (remainder, step) tags: vis_key)
  int redo;                // 1: fallback launch, only queries with status NANN_ERR_CAPACITY
  int phase;               // kScorerMlpPhase: which traversal stage this launch runs (0: up to the entry layer's scoring call,
                           // p = 1..5: from behind round p - 1's scoring call up to round p's, 5: to the end); nann_mlp6.h
};

static_assert(PH_COUNT == NANN_NUM_PHASES, "phase list out of sync with include/nann_hip.h");

struct SlotView {
  int32_t* cand_ids;
  float* cand_scores;
  int32_t* raw;
  int32_t* beam_ids;
  float* beam_scores;
  int32_t* pool_ids;
  float* pool_scores;
  uint32_t* gbitmap;
  PhaseState* phase;  // kScorerMlpPhase: what the query carries between launches
};

__host__ __device__ inline unsigned long long slot_layout(int max_cand, int max_raw, int pool_cap,
                                                          uint32_t gbm_words, unsigned long long off[9]) {
  unsigned long long o = 0;
  auto put = [&](int i, unsigned long long bytes) { off[i] = o; o += (bytes + 255ull) & ~255ull; };
  put(0, 4ull * max_cand);  // cand_ids
  put(1, 4ull * max_cand);  // cand_scores
  put(2, 4ull * max_raw);   // raw
  put(3, 4ull * kMaxK);     // beam_ids
  put(4, 4ull * kMaxK);     // beam_scores
  put(5, 4ull * pool_cap);  // pool_ids
  put(6, 4ull * pool_cap);  // pool_scores
  put(8, sizeof(PhaseState));  // (fixed size, hence in front of the last region: a kernel that passes gbm_words = 0 still finds it)
  put(7, 4ull * gbm_words); // LAST: bitmap in HBM (large shards only); where a set is parked (nann_mlp5.h, nann_mlp6.h)
  return o;
}

// scorer template values: NANN_SCORER_L2 (0), NANN_SCORER_MLP (1, f32 MFMA, bit-exact) and the MLP's split-f16 form
constexpr int kScorerMlpSplit = 2;
constexpr int kScorerAttn = 3;  // the reference's attention + DNN model (nann_attn.h); "query" = kt / upad of the user
constexpr int kScorerAttnSplit = 4;  //   the same on the 16-bit MFMA with split operands (nann_attn_split.h)
// (5 was round 3's split-f16 MLP on the table with layer 2 streamed per pass: retired in round 5, tools/rejected/nann_mlp3_streamed_layer2.h)
constexpr int kScorerAttnProj = 6;   // split-f16 attention model with its item-only layers pre-projected per (model, index) (nann_attn_proj.h)
constexpr int kScorerMlpRes = 7;     // split-f16 MLP on the pre-projected table with ALL of layer 2 resident in LDS (nann_mlp5.h)
constexpr int kScorerMlpXRes = 8;    // exact f32 MLP on the pre-projected table, layer 2 resident in LDS: bit-identical to the oracle
constexpr int kScorerMlpPhase = 9;   // the MLP traversal as a pipeline of phases (nann_mlp6.h): this instance runs the traversal stages BETWEEN
                                     // scoring calls (k_mlp_phase_score runs those), one slot per QUERY, two workgroups per CU
constexpr bool is_mlp_res(int sc) { return sc == kScorerMlpRes || sc == kScorerMlpXRes; }
// bytes at the head of the resident weights that another phase overwrites between two scoring calls (reloaded per call):
// the 16K-slot set (hash plan) or the bitmap filter's phase scratch (HBM-bitmap plan)
constexpr int mlp_res_reload_bytes(bool hash) { return hash ? 65536 : 32768; }
static_assert(kPhaseScratch <= 32768, "the bitmap kernels' phase scratch lies over the first two weight tiles");
constexpr int kScorerAttnRes = 11;   // split-f16 attention model on its table with everything a scoring call reads resident in LDS (nann_attn_proj.h)
constexpr int kScorerAttnXProj = 10;  // the f32-MFMA attention model on the same pre-projected table (nann_attn_kernels.h, PROJ)
constexpr bool is_attn(int sc) { return sc == kScorerAttn || sc == kScorerAttnSplit || sc == kScorerAttnProj || sc == kScorerAttnXProj || sc == kScorerAttnRes; }

// where a query's visited set lives
enum : int {
  VIS_LDS_BITMAP = 0,  // 1 bit per item in LDS (N <= ~1.07M): one 1024-thread workgroup per CU
  VIS_HBM_BITMAP = 1,  // 1 bit per item in the slot's HBM scratch (larger shards)
  VIS_LDS_HASH = 2,    // exact hash set of visited ids in LDS, 16K slots (64 KB): two 512-thread workgroups per CU
  VIS_LDS_HASH32 = 3   // the same with 32K slots (128 KB): one 1024-thread workgroup per CU (wide beams)
};
constexpr int vis_slots(int vis) { return vis == VIS_LDS_HASH ? 16384 : vis == VIS_LDS_HASH32 ? 32768 : 0; }

// LDS layout of the hash-set traversal: [set | phase scratch | q f32[kMaxD] | misc]
template <int NT, int SLOTS>
constexpr int hash_phase_scratch() {
  constexpr int a = (int)sizeof(TopkScratch), b = (int)sizeof(ExpandHashScratch<NT, SLOTS>);
  return ((a > b ? a : b) + 255) & ~255;
}
// phase scratch of a traversal kernel: the attention scorer stages 32 KB weight slices
constexpr int kAttnSplitScratch = (kAttnSlice + kAttnVecFloats) * 4 + kAttnResidentBytes;  // split forms: slice buffers, vectors, resident fragments (nann_attn_proj.h)
constexpr int kAttnScratch = kAttnSplitScratch > kAttnXResFloats * 4 ? kAttnSplitScratch : kAttnXResFloats * 4;  // (f32 form, resident: nann_attn_kernels.h)
constexpr int kMlpSplitScratch = (int)((sizeof(MlpSplitScratch) + 255) & ~(size_t)255);  // two slice buffers + the vectors
static_assert(sizeof(Mlp2Scratch<128>) <= sizeof(MlpSplitScratch), "the second mapping's tile buffers fit the same phase scratch");
template <int VIS, int SC, int NT>
constexpr int phase_scratch() {
  constexpr bool hash = VIS == VIS_LDS_HASH || VIS == VIS_LDS_HASH32;
  constexpr int base = hash ? hash_phase_scratch<NT, vis_slots(VIS) ? vis_slots(VIS) : 16384>() : kPhaseScratch;
  // resident layer 2 (nann_mlp5.h), LDS = [W2 128 KB | vectors | ...]:
  //   hash plan        the 16K-slot set lies over W2's first 64 KB, the phase scratch BEHIND the vectors: a scoring call
  //                    reloads 64 KB of weights, the other 64 KB stay for the whole launch
  //   HBM-bitmap plan  the (larger) phase scratch of the bitmap filter lies over W2's first 32 KB: 32 KB reloaded per call
  if (is_mlp_res(SC)) return hash ? base : kMlpResBytes;
  if (is_attn(SC) && base < kAttnScratch) return kAttnScratch;
  if (SC == kScorerMlpSplit && base < kMlpSplitScratch) return kMlpSplitScratch;
  return base;
}

template <int LPR, int DT, int VIS, int SC, int NT>
__device__ __forceinline__ int search_one(const SearchArgs& a, int qi, const SlotView& sv, uint32_t* bm,
                                          unsigned char* scratch, float* qv, int32_t* ctr,
                                          long long* ticks) {
  constexpr bool HASH = VIS == VIS_LDS_HASH || VIS == VIS_LDS_HASH32;
  constexpr bool LDSBM = VIS == VIS_LDS_BITMAP;
  constexpr int SLOTS = HASH ? vis_slots(VIS) : 16384;
  const int tid = local_tid();
  const int k5 = a.t[5];  // output row stride
  if constexpr (SC == kScorerMlpPhase) {
    // a query an earlier stage finished (a failed request, or one handed back for the bitmap rerun) is skipped here, on the
    // round trip that fetches the rest of its state -- not by a load of its own in the queue loop
    if (a.phase > 0 && sv.phase->status != kPhasePending) return kPhaseSkip;
  }
  int t[6];               // this query's level_topn (uniform)
#pragma unroll
  for (int i = 0; i < 6; ++i) t[i] = a.tq ? a.tq[(size_t)qi * 6 + i] : a.t[i];
  if (a.tq) {
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) bad |= t[i] < 0 || t[i] > a.t[i];
    if (bad) return NANN_ERR_BAD_ARGUMENT;
  }
  // bitmap kernels, L2: candidate scores are mirrored in LDS for the selection; the MLP uses that
  // space for its weight slices (and its selection time is negligible next to the MFMAs)
  float* lds_scores = (SC == NANN_SCORER_L2 && !HASH) ? reinterpret_cast<float*>(scratch + kLdsScoresOff) : nullptr;
  PhaseTimer timer;
  timer.start(ticks, a.phase_ticks != nullptr);
  const SubTimer pt{ticks, a.phase_ticks != nullptr};
  auto mark = [&](int phase) { timer.mark(phase); };

  if constexpr (!is_attn(SC)) {
    if (SC != kScorerMlpPhase || a.phase == 0)  // (the later stages of the pipeline of phases score nothing: no query vector)
      for (int k = tid; k < a.d; k += NT) qv[k] = a.q[(size_t)qi * a.d + k];
  }
  __syncthreads();
  constexpr int H1T = 8, H2T = 4;  // 256-128-1 (BASELINE configs 3-5)
  float mlp_u = 0.0f;              // MLP: thread j's per-query part of hidden unit j, once per query
  if constexpr (SC == NANN_SCORER_MLP || SC == kScorerMlpSplit || is_mlp_res(SC)) mlp_u = wg_mlp_query_u<NT>(a.mlp, qv);
  constexpr bool PHASED = SC == kScorerMlpPhase;
  bool resumed = false;  // PHASED, stage > 0: this launch starts BEHIND the scoring call of round a.phase - 1
  if constexpr (PHASED) {
    if (a.phase == 0) {  // the query's part of layer 1, once: the scoring launches read it from the slot
      const float u = wg_mlp_query_u<NT>(a.mlp, qv);
      if (tid < 256) sv.phase->u[tid] = u;
    } else {
      resumed = true;
      if (tid < 3 * NANN_NUM_ROUNDS) ctr[tid] = sv.phase->ctr[tid];
    }
  }

  // top-k ranks its selected pairs bin by bin (nann_device.h, round 6) in the L2 traversals: ~11 of a lone query's 144 us were the
  // all-pairs ranking (the register-starved MLP / attention kernels keep it: their selection is a percent of their time)
#ifndef NANN_TOPK_BINS
#define NANN_TOPK_BINS 1
#endif
  constexpr bool kTopkBins = NANN_TOPK_BINS && SC == NANN_SCORER_L2;
  // The schedule of build_opt_graph.py:109-149 as six stages with ONE call site per
  // building block: stage 0 = entry layer (:111-112), 1 = level 1 (:114-127),
  // 2..4 = the three level-0 rounds (:129-141), 5 = final top-k (:143-149).
  int vis_count = 0;  // hash set: ids in it (uniform)
  const int E = a.n_enter;
  int nP = 0;                         // pool size so far
  const int32_t* frontier = nullptr;  // beam walked by the next stage
  int nB = 0;
  int r_first = 0;
  if constexpr (PHASED) {
    if (resumed) {
      r_first = a.phase - 1;
      nP = sv.phase->nP;
      vis_count = sv.phase->vis_count;
      if constexpr (HASH) {
        if (r_first == 2 || r_first == 3) {  // the level-0 set goes on: back from where it was parked
          const uint4* park = reinterpret_cast<const uint4*>(sv.gbitmap);
          for (int i = tid; i < SLOTS / 4; i += NT) reinterpret_cast<uint4*>(bm)[i] = park[i];
        }
      }
      __syncthreads();
    }
  }
  for (int r = r_first; r <= NANN_NUM_ROUNDS; ++r) {
    const int32_t* sc_ids = nullptr;  // what this stage scores
    float* sc_out = nullptr;
    int sc_n = 0, base_off = 0;
    const bool behind_score = PHASED && resumed && r == r_first;  // this round's lists were built and scored by earlier launches
    if (behind_score) {
      sc_n = sv.phase->sc_n;
      base_off = sv.phase->base_off;
    } else
    if (r == 0) {
      sc_ids = a.enter; sc_out = sv.cand_scores; sc_n = E;
      if (tid == 0) ctr[2 * NANN_NUM_ROUNDS + 0] = E;
    } else if (r < NANN_NUM_ROUNDS) {
      const int level = (r == 1) ? 1 : 0;
      int nC = 0, G = 0;
      // sub-step 0 ("mark", only when a level starts): fresh visited set, then the current
      // result set goes through BitmapRefDifference (:115-120, :131-133).
      // sub-step 1: neighbours of the frontier, filtered (:116,121-122 / :136-137).
      for (int ss = (r <= 2) ? 0 : 1; ss < 2; ++ss) {
        const int32_t* src;
        const int64_t* rs;
        int n_in;
        int32_t* dst;
        if (ss == 0) {
          mark(PH_OTHER);
          if constexpr (HASH) { wg_vis_clear<SLOTS>(bm); vis_count = 0; }
          else wg_zero_words(bm, a.bm_words);
          __syncthreads();
          mark(PH_ZERO);
          src = (r == 1) ? sv.beam_ids : sv.pool_ids;
          n_in = (r == 1) ? t[0] : t[1];
          dst = (r == 1) ? sv.cand_ids : sv.beam_ids;
          rs = nullptr;
        } else {
          src = a.nbv[level]; rs = a.nbrs[level]; n_in = nB; dst = sv.cand_ids + base_off;
        }
        int gathered = 0;
        int kept = -3;  // not done yet
        if constexpr (!HASH) {
          if (ss == 0) {
            // The list to mark is a TopKV2 output over distinct nodes, hence duplicate-free, and
            // the bitmap is empty: BitmapRefDifference returns the list unchanged and the order
            // in which bits are set does not matter -> every thread ORs its bit.  The returned
            // words prove the premise; if it ever failed, redo the step with the ordered filter.
            int* flags = reinterpret_cast<int*>(scratch);  // [0] duplicate seen, [1] id out of range
            if (tid < 2) flags[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n_in; i += NT) {
              const int32_t id = src[i];
              if ((uint32_t)id < a.n_items) {
                const uint32_t bit = 1u << (id & 31);
                if (atomicOr(&bm[(uint32_t)id >> 5], bit) & bit) flags[0] = 1;
                dst[i] = id;
              } else {
                flags[1] = 1;
              }
            }
            __syncthreads();
            const int dup = flags[0], oob = flags[1];
            __syncthreads();
            if (oob) return NANN_ERR_INDEX_OUT_OF_RANGE;
            if (!dup) {
              kept = n_in;
            } else {
              wg_zero_words(bm, a.bm_words);
              __syncthreads();
            }
          }
          if (kept == -3)
            kept = wg_expand_walk<LDSBM, NT>(ss == 0 ? nullptr : frontier, n_in, src, rs, a.n_items, bm, dst,
                                             scratch, &gathered, ss == 0 ? no_timer() : pt);
        } else {
          if (ss == 0) {  // distinct ids into an empty set: one CAS each; a duplicate redoes it in order
            kept = wg_mark_hash<NT, SLOTS>(src, n_in, a.n_items, bm, a.id_bits, dst, scratch);
            if (kept == -4) { wg_vis_clear<SLOTS>(bm); __syncthreads(); kept = -3; }
            else if (kept >= 0) vis_count = kept;
          }
          if (kept == -3)
            kept = wg_expand_hash<NT, SLOTS>(ss == 0 ? nullptr : frontier, n_in, src, rs, a.n_items, bm, a.id_bits,
                                             vis_count, dst, scratch, &gathered, ss == 0 ? no_timer() : pt);
        }
        mark(ss == 0 ? PH_WALK : PH_EXPAND);
        if (kept == -2) return NANN_ERR_CAPACITY;  // the hash set could overflow: the host reruns the query on a bitmap kernel
        if (kept < 0) return NANN_ERR_INDEX_OUT_OF_RANGE;
        if (ss == 0) {
          if (r == 1) {
            if (kept != t[0]) return NANN_ERR_BAD_ARGUMENT;  // duplicate enter points
            base_off = kept;
          }
          frontier = sv.beam_ids;  // r == 1: the entry winners; r == 2: diff(P) written there
          nB = kept;
        } else {
          nC = kept; G = gathered;
        }
      }
      if (r == 1) {  // sR in front of sC (:125-126); after the walk, whose staging shares this LDS
        for (int i = tid; i < base_off; i += NT) {
          const float v = sv.beam_scores[i];
          sv.cand_scores[i] = v;
          if (lds_scores != nullptr && i < kLdsScores) lds_scores[i] = v;
        }
      }
      if (tid == 0) { ctr[0 * 5 + r] = nB; ctr[1 * 5 + r] = G; ctr[2 * 5 + r] = nC; }
      sc_ids = sv.cand_ids + base_off; sc_out = sv.cand_scores + base_off; sc_n = nC;
    }
    if (r < NANN_NUM_ROUNDS && !behind_score) {  // forward(): GatherV2 + scorer (:91-107)
      if (sc_n == 0) return NANN_ERR_EMPTY_SCORE_BATCH;
      mark(PH_OTHER);
      if constexpr (PHASED) {
        // the scoring call is another launch's (k_mlp_phase_score over every query's list): leave what the stage behind
        // it needs in the slot -- and the set, when a later round still reads it (rounds 2 and 3; see nann_mlp5.h)
        __syncthreads();
        if (tid == 0) {
          sv.phase->r = r; sv.phase->sc_n = sc_n; sv.phase->base_off = base_off; sv.phase->nP = nP;
          sv.phase->vis_count = vis_count;
        }
        if (tid < 3 * NANN_NUM_ROUNDS) sv.phase->ctr[tid] = ctr[tid];
        if constexpr (HASH) {
          if (r == 2 || r == 3) {
            uint4* park = reinterpret_cast<uint4*>(sv.gbitmap);
            for (int i = tid; i < SLOTS / 4; i += NT) park[i] = reinterpret_cast<const uint4*>(bm)[i];
          }
        }
        return kPhasePending;
      }
      if constexpr (!PHASED) {
      if constexpr (SC == NANN_SCORER_L2) {
        wg_score_l2_part<LPR, DT, NT / 64>(a.emb, a.d, sc_ids, 0, sc_n, qv, sc_out, tid >> 6, (unsigned long long)a.n_items * (unsigned)(a.d * 2) <= 0xffffffffull && a.n_items <= (1u << 24));
#if NANN_REPEAT_SCORE  // measurement builds (tools/build_res_variant.py --unit nann_l2_inst.hip): a phase run twice costs what it costs once
        for (int rep = 0; rep < NANN_REPEAT_SCORE; ++rep) {
          __syncthreads();
          wg_score_l2_part<LPR, DT, NT / 64>(a.emb, a.d, sc_ids, 0, sc_n, qv, sc_out, tid >> 6, (unsigned long long)a.n_items * (unsigned)(a.d * 2) <= 0xffffffffull && a.n_items <= (1u << 24));
        }
#endif
        if (lds_scores != nullptr) {
          __syncthreads();
          for (int i = tid; i < sc_n && base_off + i < kLdsScores; i += NT)  // LDS mirror for the selection
            lds_scores[base_off + i] = sc_out[i];
        }
      } else if constexpr (SC == kScorerAttn) {
        wg_score_attn<LPR * 8, DT, NT>(a.attn, a.kt + (size_t)qi * 256 * kAttnLP, a.upad + (size_t)qi * kAttnLP * kAttnE,
                                       a.emb, (long long)a.n_items, sc_ids, (long long)sc_n,
                                       reinterpret_cast<float*>(scratch), sc_out);
      } else if constexpr (SC == kScorerAttnXProj) {
        if constexpr (VIS == VIS_LDS_HASH) {  // everything resident for the call: the keys over the (parked) visited set
          uint4* set4 = reinterpret_cast<uint4*>(bm);
          uint4* park = (r == 2 || r == 3) ? reinterpret_cast<uint4*>(sv.gbitmap) : nullptr;
          __syncthreads();
          if (park != nullptr)
            for (int i = tid; i < SLOTS / 4; i += NT) park[i] = set4[i];
          wg_score_attn<128, DT_F16, NT, true, true>(a.attn, a.kt + (size_t)qi * 256 * kAttnLP, a.upad + (size_t)qi * kAttnLP * kAttnE,
                                                     a.proj, (long long)a.n_items, sc_ids, (long long)sc_n,
                                                     reinterpret_cast<float*>(scratch), sc_out, reinterpret_cast<float*>(bm));
          if (park != nullptr)
            for (int i = tid; i < SLOTS / 4; i += NT) set4[i] = park[i];
        } else {
          wg_score_attn<128, DT_F16, NT, true>(a.attn, a.kt + (size_t)qi * 256 * kAttnLP, a.upad + (size_t)qi * kAttnLP * kAttnE,
                                               a.proj, (long long)a.n_items, sc_ids, (long long)sc_n,
                                               reinterpret_cast<float*>(scratch), sc_out);
        }
      } else if constexpr (SC == kScorerAttnSplit) {
        wg_score_attn_split<LPR * 8, DT, NT>(a.attn, reinterpret_cast<const uint4*>(a.kt + (size_t)qi * 256 * kAttnLP),
                                             reinterpret_cast<const uint4*>(a.upad + (size_t)qi * kAttnLP * kAttnE),
                                             a.emb, (long long)a.n_items, sc_ids, (long long)sc_n,
                                             reinterpret_cast<float*>(scratch), sc_out);
      } else if constexpr (SC == kScorerAttnRes) {
        // the user's keys take the place of the visited set for the call (the set parked in the slot when a later stage
        // still needs it: stages 2 and 3, as for the MLP with resident layer 2)
        static_assert(VIS == VIS_LDS_HASH, "the resident attention scorer: the 16K-slot set's 64 KB hold the keys");
        uint4* set4 = reinterpret_cast<uint4*>(bm);
        uint4* park = (r == 2 || r == 3) ? reinterpret_cast<uint4*>(sv.gbitmap) : nullptr;
        __syncthreads();
        if (park != nullptr)
          for (int i = tid; i < SLOTS / 4; i += NT) park[i] = set4[i];
        wg_score_attn_res<NT>(a.attn, reinterpret_cast<const uint4*>(a.kt + (size_t)qi * 256 * kAttnLP),
                              reinterpret_cast<const uint4*>(a.upad + (size_t)qi * kAttnLP * kAttnE),
                              a.proj, (long long)a.n_items, sc_ids, (long long)sc_n, set4,
                              reinterpret_cast<float*>(scratch), sc_out);
        if (park != nullptr)
          for (int i = tid; i < SLOTS / 4; i += NT) set4[i] = park[i];
      } else if constexpr (SC == kScorerAttnProj) {
        wg_score_attn_proj<NT>(a.attn, reinterpret_cast<const uint4*>(a.kt + (size_t)qi * 256 * kAttnLP),
                               reinterpret_cast<const uint4*>(a.upad + (size_t)qi * kAttnLP * kAttnE),
                               a.proj, (long long)a.n_items, sc_ids, (long long)sc_n,
                               reinterpret_cast<float*>(scratch), sc_out);
      } else if constexpr (is_mlp_res(SC)) {
        // layer 2 resident in LDS over [visited set | phase scratch] (nann_mlp5.h).  The set is parked in the slot's HBM
        // scratch when a later stage still needs it: stages 2 and 3 (it is empty before stage 0, cleared after stage 1
        // -- :129-133 -- and dead after stage 4)
        static_assert(NT == 512, "the resident scorers: eight wavefronts, two per SIMD");
        static_assert(VIS == VIS_LDS_HASH || VIS == VIS_HBM_BITMAP, "resident layer 2: 16K-slot set or HBM bitmap");
        uint4* lds0 = reinterpret_cast<uint4*>(HASH ? reinterpret_cast<unsigned char*>(bm) : scratch);  // = the kernel's LDS base
        Mlp2Vectors* V = reinterpret_cast<Mlp2Vectors*>(reinterpret_cast<unsigned char*>(lds0) + kMlpResW2Bytes);
        constexpr int kReload = mlp_res_reload_bytes(HASH) / 16;  // uint4 of W2 that the set / the phase scratch overwrote
        uint4* park = (HASH && (r == 2 || r == 3)) ? reinterpret_cast<uint4*>(sv.gbitmap) : nullptr;
        __syncthreads();
        if constexpr (SC == kScorerMlpXRes) {
          wg_mlp_res_enter<NT, kReload>(lds0, reinterpret_cast<const uint4*>(a.mlp.p2x), park, SLOTS / 4);
          wg_mlp_xres_vectors<NT>(a.mlp, mlp_u, V);
          __syncthreads();
          wg_score_mlp_xres<NT>(a.proj, a.n_items, sc_ids, sc_n, reinterpret_cast<const float4*>(lds0), V, sc_out);
        } else {
          wg_mlp_res_enter<NT, kReload>(lds0, a.mlp.p2, park, SLOTS / 4);
          wg_mlp_res_vectors<NT>(a.mlp, mlp_u, V);
          __syncthreads();
          wg_score_mlp_res<NT>(a.proj, a.n_items, sc_ids, sc_n, lds0, V, sc_out);
#if NANN_REPEAT_SCORE
          for (int rep = 0; rep < NANN_REPEAT_SCORE; ++rep) {
            __syncthreads();
            wg_score_mlp_res<NT>(a.proj, a.n_items, sc_ids, sc_n, lds0, V, sc_out);
          }
#endif
        }
        __syncthreads();
        wg_mlp_res_leave<NT>(lds0, park, SLOTS / 4);
      } else {
        // (the phase scratch was reused since the last stage)
        if constexpr (SC == kScorerMlpSplit && NT == kMlp2NT) {  // second mapping: 4 wavefronts x 64 rows (nann_mlp2.h)
          Mlp2Scratch<LPR * 8>* M = reinterpret_cast<Mlp2Scratch<LPR * 8>*>(scratch);
          wg_mlp2_stage_setup<NT>(a.mlp, mlp_u, &M->v);
          wg_score_mlp_split2<LPR * 8, DT>(a.mlp, a.emb, a.n_items, sc_ids, sc_n, M, sc_out);
        } else if constexpr (SC == kScorerMlpSplit) {
          MlpSplitScratch* M = reinterpret_cast<MlpSplitScratch*>(scratch);
          wg_mlp_stage_setup<NT>(a.mlp, mlp_u, &M->v, kSplitWScale, kSplitWScale * kSplitHScale, kSplitHScale / kSplitWScale);
          wg_score_mlp_split<LPR * 8, H1T, H2T, DT, NT>(a.mlp, a.emb, a.n_items, sc_ids, sc_n, M, sc_out);
        } else {
          MlpScratch* M = reinterpret_cast<MlpScratch*>(scratch);
          wg_mlp_stage_setup<NT>(a.mlp, mlp_u, &M->v);
          wg_score_mlp<LPR * 8, H1T, H2T, DT, NT>(a.mlp, a.emb, a.n_items, sc_ids, sc_n, M, sc_out);
        }
      }
      }  // !PHASED
      __syncthreads();
      mark(PH_SCORE);
    }
    if (r < NANN_NUM_ROUNDS && sc_n == 1) return NANN_ERR_TOPK_SCALAR_INPUT;
    // top_k(): TopKV2 + Gather of the ids (:52-66)
    const int32_t* tk_ids; const float* tk_sc; int tk_n, tk_k;
    int32_t* tk_out_ids; float* tk_out_sc; const int64_t* tk_map = nullptr; int64_t* tk_out_map = nullptr;
    if (r == 0) {          // R, sR = topk(EP, s, t0)                      :112
      tk_ids = a.enter; tk_sc = sv.cand_scores; tk_n = E; tk_k = t[0];
      tk_out_ids = sv.beam_ids; tk_out_sc = sv.beam_scores;
    } else if (r == 1) {   // P, sP = topk(R || C, sR || sC, t1)           :125-127
      tk_ids = sv.cand_ids; tk_sc = sv.cand_scores; tk_n = base_off + sc_n; tk_k = t[1];
      tk_out_ids = sv.pool_ids; tk_out_sc = sv.pool_scores;
    } else if (r < NANN_NUM_ROUNDS) {  // B, sB = topk(C, sC, t[r]); appended to the pool  :139-141
      tk_ids = sv.cand_ids; tk_sc = sv.cand_scores; tk_n = sc_n; tk_k = t[r];
      tk_out_ids = sv.pool_ids + nP; tk_out_sc = sv.pool_scores + nP;
    } else {               // final: topk(pool, t5) -> item_ids           :143-149
      tk_ids = sv.pool_ids; tk_sc = sv.pool_scores; tk_n = nP; tk_k = t[5];
      tk_out_ids = a.out_index ? a.out_index + (size_t)qi * k5 : nullptr;
      tk_out_sc = a.out_scores ? a.out_scores + (size_t)qi * k5 : nullptr;
      tk_map = a.item_ids; tk_out_map = a.out_ids + (size_t)qi * k5;
    }
    mark(PH_OTHER);
#if NANN_REPEAT_TOPK
    for (int rep = 0; rep < NANN_REPEAT_TOPK; ++rep) {
      (void)wg_topk<NT, kMaxK, kTopkBins>(tk_ids, tk_sc, (r < NANN_NUM_ROUNDS) ? lds_scores : nullptr, tk_n, tk_k,
                        nullptr, tk_out_ids, tk_out_sc, tk_map, tk_out_map, scratch, pt);
      __syncthreads();
    }
#endif
    const int st = wg_topk<NT, kMaxK, kTopkBins>(tk_ids, tk_sc, (r < NANN_NUM_ROUNDS) ? lds_scores : nullptr, tk_n, tk_k,
                               nullptr, tk_out_ids, tk_out_sc, tk_map, tk_out_map, scratch, pt);
    mark(PH_TOPK);
    if (st) return st;
    if (r == 1) {
      nP = t[1];
    } else if (r >= 2 && r < NANN_NUM_ROUNDS) {
      frontier = sv.pool_ids + nP;  // the beam = best NEW nodes only
      nB = t[r];
      nP += nB;
    }
  }
  mark(PH_OTHER);
  return NANN_OK;
}

// header of the workspace (256 bytes, zeroed by the host before the first launch of a call)
struct WsHeader {
  unsigned int queue;        // next query of the main launch
  unsigned int pad0[15];
  unsigned int redo_queue;   // next query of the fallback launch
  unsigned int pad1[15];
  unsigned int n_redo;       // queries the hash-set kernel handed back (NANN_ERR_CAPACITY)
  unsigned int pad2[15];
  unsigned int pqueue[8];    // next query of traversal stage p of the phased MLP pipeline (nann_mlp6.h)
};
static_assert(sizeof(WsHeader) <= 256, "workspace header");

// Second launch bound = waves per SIMD the register allocation must leave room for: the 16K-slot
// hash-set kernel lives off TWO 512-thread workgroups per CU (16 waves = 4 per SIMD -> at most 128
// VGPRs; at 130 the second workgroup silently stops fitting and the kernel runs at half occupancy).
template <int LPR, int DT, int VIS, int SC, int NT>
__global__ __launch_bounds__(NT, ((VIS == VIS_LDS_HASH && (SC == NANN_SCORER_L2 || SC == kScorerMlpPhase)) ? 2 : 1) * NT / 256) void k_search(SearchArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool HASH = VIS == VIS_LDS_HASH || VIS == VIS_LDS_HASH32;
  constexpr int kScratchBytes = phase_scratch<VIS, SC, NT>();
  uint32_t* bm_lds = reinterpret_cast<uint32_t*>(smem);
  unsigned char* scratch = smem + (VIS == VIS_LDS_BITMAP ? (size_t)a.bm_words * 4
                                   : (is_mlp_res(SC) && HASH) ? (size_t)kMlpResBytes : (size_t)vis_slots(VIS) * 4);
  float* qv = reinterpret_cast<float*>(scratch + kScratchBytes);
  int* misc = reinterpret_cast<int*>(qv + kMaxD);  // [0] next query
  int32_t* s_ctr = misc + 2;                        // [3 * NANN_NUM_ROUNDS]
  long long* s_ticks = reinterpret_cast<long long*>(misc + 20);  // [NANN_NUM_PHASES]

  unsigned long long off[9];
  slot_layout(a.max_cand, a.max_raw, a.pool_cap, VIS == VIS_HBM_BITMAP ? a.bm_words : 0u, off);  // (only the offsets are used)
  constexpr bool PHASED = SC == kScorerMlpPhase;  // one slot per QUERY: its state outlives the launch (nann_mlp6.h)
  SlotView sv;
  auto bind_slot = [&](unsigned long long index) {
  unsigned char* slot = a.ws + 256 + index * a.slot_bytes;
  sv.cand_ids = reinterpret_cast<int32_t*>(slot + off[0]);
  sv.cand_scores = reinterpret_cast<float*>(slot + off[1]);
  sv.raw = reinterpret_cast<int32_t*>(slot + off[2]);
  sv.beam_ids = reinterpret_cast<int32_t*>(slot + off[3]);
  sv.beam_scores = reinterpret_cast<float*>(slot + off[4]);
  sv.pool_ids = reinterpret_cast<int32_t*>(slot + off[5]);
  sv.pool_scores = reinterpret_cast<float*>(slot + off[6]);
  sv.gbitmap = reinterpret_cast<uint32_t*>(slot + off[7]);
  sv.phase = reinterpret_cast<PhaseState*>(slot + off[8]);
  };
  bind_slot(blockIdx.x);
  uint32_t* bm = VIS == VIS_HBM_BITMAP ? sv.gbitmap : bm_lds;
  const int k5 = a.t[5];
  WsHeader* hdr = reinterpret_cast<WsHeader*>(a.ws);
  // fallback launch (a.redo != 0): only the queries the hash-set kernel handed back
  if (a.redo) {
    if (__hip_atomic_load(&hdr->n_redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
  }
  unsigned int* queue = a.redo ? &hdr->redo_queue : PHASED ? &hdr->pqueue[a.phase] : &hdr->queue;
  if constexpr (is_mlp_res(SC)) {  // the part of the resident weights no other phase overwrites: once per launch
    const uint4* w2 = SC == kScorerMlpXRes ? reinterpret_cast<const uint4*>(a.mlp.p2x) : a.mlp.p2;
    constexpr int keep_from = mlp_res_reload_bytes(HASH) / 16;
    for (int i = keep_from + (int)threadIdx.x; i < kMlpResW2Vec; i += NT) reinterpret_cast<uint4*>(smem)[i] = w2[i];
  }

  // queries are pulled from one device-wide counter: a slot that finishes early takes
  // the next request instead of idling until the slowest slot is done
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      int qn = (int)atomicAdd(queue, 1u);
      if (a.redo)  // skip the queries that are done
        while (qn < a.n_queries && a.status[qn] != NANN_ERR_CAPACITY) qn = (int)atomicAdd(queue, 1u);
      misc[0] = qn;
    }
    if (threadIdx.x < 3 * NANN_NUM_ROUNDS) s_ctr[threadIdx.x] = 0;
    if (threadIdx.x < NANN_NUM_PHASES) s_ticks[threadIdx.x] = 0;
    __syncthreads();
    const int qi = misc[0];
    if (qi >= a.n_queries) break;
    if constexpr (PHASED) bind_slot((unsigned long long)qi);
    const int st = search_one<LPR, DT, VIS, SC, NT>(a, qi, sv, bm, scratch, qv, s_ctr, s_ticks);
    __syncthreads();
    if constexpr (PHASED) {
      if (st == kPhaseSkip) continue;  // (finished by an earlier stage)
      if (threadIdx.x == 0) sv.phase->status = st;
      if (a.phase_ticks && threadIdx.x < NANN_NUM_PHASES) {  // ticks add up over the stages (scoring launches: not in here)
        long long* dst = &a.phase_ticks[(size_t)qi * NANN_NUM_PHASES + threadIdx.x];
        s_ticks[threadIdx.x] += a.phase ? *dst : 0;
        if (st == kPhasePending) *dst = s_ticks[threadIdx.x];
      }
      if (st == kPhasePending) continue;  // its scoring call is the next launch's
    }
    // a request the reference would fail: zeroed outputs + its code; per-query level_topn: the row's tail is zero
    const int k_done = st ? 0 : (a.tq ? a.tq[(size_t)qi * 6 + 5] : k5);
    if (k_done < k5) {
      for (int i = k_done + (int)threadIdx.x; i < k5; i += NT) {
        a.out_ids[(size_t)qi * k5 + i] = 0;
        if (a.out_scores) a.out_scores[(size_t)qi * k5 + i] = 0.0f;
        if (a.out_index) a.out_index[(size_t)qi * k5 + i] = 0;
      }
    }
    if (threadIdx.x == 0) {
      a.status[qi] = st;
      if (HASH && st == NANN_ERR_CAPACITY) atomicAdd(&hdr->n_redo, 1u);
    }
    if (a.counters && threadIdx.x < 3 * NANN_NUM_ROUNDS)
      a.counters[(size_t)qi * 3 * NANN_NUM_ROUNDS + threadIdx.x] = s_ctr[threadIdx.x];
    if (a.phase_ticks && threadIdx.x < NANN_NUM_PHASES)
      a.phase_ticks[(size_t)qi * NANN_NUM_PHASES + threadIdx.x] = s_ticks[threadIdx.x];
  }
}

struct SearchPlan {
  int max_cand, max_raw, pool_cap;
  int vis;             // VIS_*
  int id_bits;         // VIS_LDS_HASH*: bits of the id space the set's tags are cut from
  size_t lds_bytes;
  unsigned long long slot_bytes;
  int slots;
  int nt;              // threads per workgroup
  // VIS_LDS_HASH: the bitmap plan that reruns queries whose set would overflow
  int fb_vis;
  size_t fb_lds_bytes;
  int fb_slots;
  // MLP with a pre-projected table, beams that fit the 16K-slot set: the pipeline of phases (nann_mlp6.h) -- traversal
  // stages at this geometry (one slot per query of a chunk), scoring launches between them
  bool phased;
  size_t phase_lds_bytes;
  int phase_slots;
  int phase_vis, phase_per_cu;  // VIS_LDS_HASH (two 512-thread workgroups per CU) | VIS_LDS_HASH32 (one of 1024: wide beams)
  int phase_score_wgs;          // workgroups of a scoring launch (one per CU, fewer with a slot reserve)
  float est_visited, worst_visited;  // the planner's estimate of a level's visited ids / the bound from max degrees
};

template <int LPR, int DT, int VIS, int SC, int NT>
inline int launch_search_as(int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  auto kern = k_search<LPR, DT, VIS, SC, NT>;
  if (lds_bytes > 48 * 1024)
    NANN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(slots), dim3(NT), lds_bytes, st, a);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

// bitmap kernels (either residence) at NT threads
template <int LPR, int DT, int SC, int NT>
inline int launch_search_bitmap(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st) {
  if (vis == VIS_LDS_BITMAP) return launch_search_as<LPR, DT, VIS_LDS_BITMAP, SC, NT>(slots, lds_bytes, a, st);
  return launch_search_as<LPR, DT, VIS_HBM_BITMAP, SC, NT>(slots, lds_bytes, a, st);
}

// L2 instantiations live in nann_l2_inst.hip (one object per row dtype): (vis, nt) in
// {(VIS_LDS_BITMAP, 1024), (VIS_HBM_BITMAP, 1024), (VIS_LDS_HASH, 512), (VIS_LDS_HASH32, 1024)}
int launch_search_l2_f16(int lpr, int vis, int nt, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_l2_bf16(int lpr, int vis, int nt, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_l2_f32(int lpr, int vis, int nt, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
// MLP instantiations live in nann_mlp_inst.hip (one object per embedding dim): bitmap kernels, 512 threads
int launch_search_mlp_d64(int dt, int split, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_mlp_d128(int dt, int split, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_mlp_d256(int dt, int split, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
// launch_mlp_preproject fills the table of pre-projected item halves of layer 1 (nann_mlp3.h); lives in the d = 128 object
// layer 2 resident in LDS (nann_mlp5.h), split-f16 (exact = 0) or exact f32 (exact = 1); vis in {VIS_LDS_HASH, VIS_HBM_BITMAP}
int launch_search_mlp_res(int exact, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
// the pipeline of phases (nann_mlp6.h): a traversal stage (a.phase), a round's scoring launch
int launch_search_mlp_phase(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_mlp_phase_score(int exact, const SearchArgs& a, int round, int workgroups, hipStream_t st);
int launch_mlp_preproject(int dt, const void* emb, long long n_rows, int d, const float* w1, float* proj, hipStream_t st);
// attention-scorer instantiations live in nann_attn_inst.hip: (vis, 512 threads) for vis in
// {VIS_LDS_HASH (one workgroup per CU), VIS_LDS_BITMAP, VIS_HBM_BITMAP}
int launch_search_attn(int d, int dt, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_attn_split(int d, int dt, int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);  // nann_attn_split_inst.hip
// the pre-projected form (nann_attn_proj.h): one instantiation per plan for every d / row dtype (it never reads the
// embedding table); launch_attn_preproject fills the f32 [n_rows, kAttnProjWidth] table.  nann_attn_split_inst.hip
int launch_search_attn_proj(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);
int launch_search_attn_xproj(int vis, int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);  // nann_attn_inst.hip
int launch_search_attn_res(int slots, size_t lds_bytes, const SearchArgs& a, hipStream_t st);  // (16K-slot hash plan only)
int launch_attn_preproject(int dt, const AttnParams& P, const void* emb, long long n_rows, float* proj, hipStream_t st);
int launch_score_mlp_d64(int dt, int split, unsigned blocks, hipStream_t st, const MlpParams& P, const void* table,
                         long long n_table_rows, const int32_t* indices, long long n, const float* q,
                         float* out, OpResult* res);
int launch_score_mlp_d128(int dt, int split, unsigned blocks, hipStream_t st, const MlpParams& P, const void* table,
                          long long n_table_rows, const int32_t* indices, long long n, const float* q,
                          float* out, OpResult* res);
int launch_score_mlp_d256(int dt, int split, unsigned blocks, hipStream_t st, const MlpParams& P, const void* table,
                          long long n_table_rows, const int32_t* indices, long long n, const float* q,
                          float* out, OpResult* res);

}  // namespace nann
