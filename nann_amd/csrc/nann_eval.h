// nann_eval.h -- the evaluation graph's traversal (Model.retrieval + search_level,
// NANN_impls/nann/model.py:299-362) as ONE kernel per batch of users.
//
// It differs from the serving schedule (nann_search.h) in three places, all of them set semantics:
//   * the neighbours of a frontier are taken as a SET and scored in ascending-id order
//     (tf.unique + tf.sets.difference, :316-319);
//   * top_k keeps min(k, n) (:268) and an exhausted frontier is not an error (plain TF scoring);
//   * the next frontier = the new nodes scoring at least the worst kept result (:330-331).
// Ascending sets want a bitmap, not a list: `seen` takes one bit per neighbour of the frontier, visited or not (atomics only), and a
// word-order scan of `seen` against `visited` IS the ascending, duplicate-free list of new nodes -- no sort; every bitmap word has
// ONE owner thread, which alone reads and writes that word of `visited` (plain loads / stores to the slot: no atomic, no fence)
// and clears its word of `seen`.  Three forms of that (history and measurements: DESIGN.md 4.4):
//   * search_eval_lds  -- L2 scorer, shards of up to ~1 M items (round 4; rebuilt in round 6): `seen` in LDS, thread t owning
//     words [32 t, 32 t + 32) of both bitmaps; `seen` is all zero outside gather -> emit, so its LDS is the round's staging area for
//     ids, scores and top-k's output; the frontier and its row bounds in the phase scratch; a level's marks a list; LDS-only
//     barriers behind fire-and-forget copies to the slot; top-k ranked bin by bin over the candidates that can still enter;
//   * search_eval_win  -- the same sweeping the id space in windows of 992 x 32 bitmap words: shards of up to ~8 M items;
//   * search_eval_slot -- `seen` in the slot (atomics performed in L2), a second-level bitmap of touched words in LDS: the MLP and
//     attention scorers (whose weights own the LDS) and larger shards.
// All three produce the same sets in the same orders with the same arithmetic: bit-identical to oracle_search_eval.
// top_k_per_level / topk_eval up to kEvalMaxK = 2048 (the reference's are defaults, config.py:50-58).
#pragma once
#ifndef NANN_REPEAT_SCORE
#define NANN_REPEAT_SCORE 0  // measurement builds only
#endif
#ifndef NANN_REPEAT_TOPK
#define NANN_REPEAT_TOPK 0
#endif
#ifndef NANN_EVAL_TICKS
#define NANN_EVAL_TICKS 0  // measurement builds: 100 MHz ticks per phase of a user, summed over the launch (EvalArgs.ticks; NANN_EVAL_TICKS=1 in the environment prints them)
#endif
#ifndef NANN_REPEAT_GATHER
#define NANN_REPEAT_GATHER 0  // the walk of the frontier's rows into `seen` twice (idempotent: the bits are ORed)
#endif
#include "nann_search.h"

namespace nann {

constexpr int kEvalMaxK = 2048;  // largest top_k_per_level / topk_eval / frontier

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
  uint32_t vis_words;  // words of a slot's `visited` region (the transposed layout rounds the owners' runs up)
  uint32_t lds_words;  // LDS form: words of the region that holds `seen` and, from a round's emit to its end, the staged scores / results
  uint32_t win_owners; // LDS form: owners (threads' worth of 32 bitmap words) of a window; n_windows of them cover the index
  int n_windows;
  unsigned long long* ticks;  // measurement builds (NANN_EVAL_TICKS): 16 accumulators, else unused
  int use_dirty;       // slot form: the second-level bitmap fits the phase scratch (eval_plan); the LDS form always has it
  int64_t* out_ids;    // [n_queries, topk_eval]
  float* out_scores;
  int32_t* out_index;
  int32_t* n_out;      // [n_queries] rows that are valid (min(topk_eval, results))
  int32_t* status;
  int32_t* counters;   // [n_queries, 3] or NULL: rows walked (F), neighbours gathered (G), rows scored (S), summed over a user's
                       // rounds (S includes the enter points): what SURVEY.md 8(d)'s byte formula is evaluated on
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

// The LDS form sweeps the id space in WINDOWS of kEvalWinOwners x 32 words of the bitmaps (round 6: shards beyond 2^20 items):
// thread t of the workgroup owns words [32 t, 32 t + 32) of the window, the window's `seen` bits live in LDS, and a round walks
// its frontier once per window.  One window (the 1M-item shards of configs[1-4]): every owner of the index in one sweep.
constexpr uint32_t kEvalWinOwners = 992;  // (992 x 33 words of skewed bitmap + the phase scratch + q: 157 KB of the CU's 160)
constexpr uint32_t kEvalMaxWindows = 8;   // shards of up to 8.1 M items; beyond: the slot form
__host__ __device__ inline uint32_t eval_owners(uint32_t bm_words) { return (bm_words + 31u) >> 5; }  // threads' worth of words
__host__ __device__ inline uint32_t eval_win_owners(uint32_t bm_words, uint32_t owners_max = kEvalWinOwners) {
  const uint32_t dw = eval_owners(bm_words);
  return dw < owners_max ? dw : owners_max;
}
__host__ __device__ inline uint32_t eval_windows(uint32_t bm_words, uint32_t owners_max = kEvalWinOwners) {
  const uint32_t dw = eval_owners(bm_words), ow = eval_win_owners(bm_words, owners_max);
  return ow ? (dw + ow - 1u) / ow : 1u;
}
// words of `visited` in its transposed layout: per window 32 words for each of the workgroup's (at most) 1 024 threads (word j of
// thread t of window w at (32 w + j) NT + t); the slot form's runs (thread t: dirty words [t D, (t + 1) D)) fit the same space
__host__ __device__ inline uint32_t eval_vis_words(uint32_t bm_words, uint32_t owners_max = kEvalWinOwners) {
  return eval_windows(bm_words, owners_max) * 1024u * 32u;
}
// bytes of the LDS region of `seen`: a window's bitmap skewed by one word per 32 (search_eval_lds), and at least the staging area
// of a small round (the kept results + 12 K scores) -- the region is `seen` from a round's gather to its emit and staging behind it
constexpr int kEvalStageMinWords = 2 * 2048 + 12288;
__host__ __device__ inline size_t eval_seen_lds_bytes(uint32_t bm_words, uint32_t owners_max = kEvalWinOwners,
                                                      size_t stage_min_words = kEvalStageMinWords) {
  const size_t skewed = (size_t)eval_win_owners(bm_words, owners_max) * 33u;  // (every owner's run of 32 words whole)
  return ((skewed > stage_min_words ? skewed : stage_min_words) * 4 + 255) & ~(size_t)255;
}
// (owners_max / stage_min_words: tools/rejected/nann_eval_pair_variant.patch -- two 512-thread workgroups per CU with windows of 352
//  owners measured 10-12 % slower than one 1 024-thread workgroup, profiles/r6j_eval_pair_variant_ab.jsonl)

__host__ __device__ inline unsigned long long eval_slot_layout(uint32_t bm_words, uint32_t vis_words, int cat_cap, unsigned long long off[7]) {
  unsigned long long o = 0;
  auto put = [&](int i, unsigned long long bytes) { off[i] = o; o += (bytes + 255ull) & ~255ull; };
  put(0, 4ull * vis_words);
  put(1, 4ull * bm_words);
  put(2, 4ull * cat_cap);
  put(3, 4ull * cat_cap);
  put(4, 4ull * kEvalMaxK);
  put(5, 4ull * kEvalMaxK);
  put(6, 4ull * kEvalMaxK);
  return o;
}

struct EvalScanScratch {
  uint32_t wave_tot[kNW];
  uint32_t total;
  int flags[2];
};

// exclusive prefix of v over the workgroup (thread order); *total = the sum.  Two barriers.
// A barrier that orders LDS only: it does not wait for this thread's global stores (a round's copies to the slot are read by
// nobody before the next full barrier, which a __syncthreads() -- a workgroup-scope fence -- would make every wavefront sit out)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT, bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t wg_excl_scan(uint32_t v, EvalScanScratch* S, uint32_t* total) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const uint32_t inc = wave_scan_add(v);
  if (lane == 63) S->wave_tot[wave] = inc;
  if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < NT / 64; ++w) {
    const uint32_t t = S->wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  if constexpr (LDS_ONLY) lds_barrier(); else __syncthreads();
  *total = tot;
  return base + inc - v;
}


__device__ __forceinline__ uint32_t ld_word(const uint32_t* p) {  // past the L1: the word is changed by atomics performed in L2
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS scratch of the evaluation kernel: the scorer's phase scratch or the (larger-k) top-k scratch, whichever is larger
template <int SC, int NT>
constexpr int eval_scratch_bytes() {
  constexpr int a = phase_scratch<VIS_HBM_BITMAP, SC, NT>();
  constexpr int b = (int)((sizeof(TopkScratchT<kEvalMaxK>) + 255) & ~(size_t)255);
  return a > b ? a : b;
}

template <int LPR, int DT, int SC, int NT>
__device__ __forceinline__ void eval_score(const EvalArgs& a, int qi, const int32_t* ids, int n, float* out,
                                           unsigned char* scratch, const float* qv, float mlp_u) {
  const int tid = local_tid();
  if constexpr (SC == NANN_SCORER_L2) {
    wg_score_l2_part<LPR, DT, NT / 64>(a.emb, a.d, ids, 0, n, qv, out, tid >> 6, (unsigned long long)a.n_items * (unsigned)(a.d * 2) <= 0xffffffffull && a.n_items <= (1u << 24));
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

#if NANN_EVAL_TICKS
#define EVAL_TICK_DECL long long tk_last = wall_clock64(); unsigned long long tk[10] = {}
#define EVAL_TICK(i) do { __syncthreads(); const long long now_ = wall_clock64(); tk[i] += (unsigned long long)(now_ - tk_last); tk_last = now_; } while (0)
#define EVAL_TICK_FLUSH do { if (local_tid() == 0 && a.ticks) for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&a.ticks[i_], tk[i_]); } while (0)
#else
#define EVAL_TICK_DECL do { } while (0)
#define EVAL_TICK(i) do { } while (0)
#define EVAL_TICK_FLUSH do { } while (0)
#endif

// ---- the slot form: `seen` in the slot (atomics performed in L2); MLP / attention scorers, shards whose bitmap does not fit LDS
template <int LPR, int DT, int SC, int NT>
__device__ __forceinline__ int search_eval_slot(const EvalArgs& a, int qi, const EvalSlot& sv, uint32_t* seen,
                                                unsigned char* scratch, float* qv, int* n_result, bool clear_seen) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  int ctr_f = 0, ctr_s = 0;  // (uniform)
  int ctr_g = 0;             // this lane's share: row lengths it fetched (lanes 0-7 of every wavefront)
  constexpr int NW = NT / 64;
  EvalScanScratch* SS = reinterpret_cast<EvalScanScratch*>(scratch);
  EVAL_TICK_DECL;
  if constexpr (SC != kScorerAttn) {
    for (int k = tid; k < a.d; k += NT) qv[k] = a.q[(size_t)qi * a.d + k];
  }
  // ---- the second-level bitmap (round 6).  dirty word d, bit b <-> word 32 d + b of `seen` was touched this round (the lane
  // whose atomic OR found the word still zero reports it).  Thread t owns dirty words [t D, (t + 1) D) and with them words
  // [32 t D, 32 (t + 1) D) of `seen` and `visited`: its dirty words list ITS touched words in ascending order, and thread order is
  // word order.  The dirty words live in the phase scratch behind the scan scratch: valid from a round's gather to its emit
  // (scoring and top-k reuse the scratch; they are zeroed again behind them).  use_dirty: they fit (eval_plan) -- else the full
  // scans of round 4, every bitmap word with one owner (wavefront w, trip j, lane l -> w C + 64 j + l).
  const bool use_dirty = a.use_dirty;
  const uint32_t DW = (a.bm_words + 31u) >> 5;
  const int D = (int)((DW + NT - 1) / NT);
  uint32_t* dirty = reinterpret_cast<uint32_t*>(scratch + 256);
  const uint32_t d0 = (uint32_t)tid * (uint32_t)D;
  const uint32_t C = (((a.bm_words + NW - 1) / NW) + 63u) & ~63u;
  const int J = (int)(C >> 6);
  const uint32_t w0 = (uint32_t)wave * C + (uint32_t)lane;
  // `visited` (the slot, only ever touched by its owners) is TRANSPOSED when the dirty words are in use: word j of thread t's
  // run r at (32 r + j) NT + t, so that a wavefront's loads of "my j-th word" are 256 contiguous bytes
  auto vp = [&](uint32_t w) -> uint32_t {
    const uint32_t d = w >> 5, t = d / (uint32_t)D, r = d - t * (uint32_t)D;
    return (r * 32u + (w & 31u)) * (uint32_t)NT + t;
  };
  auto seen_load = [&](uint32_t w) -> uint32_t { return ld_word(&seen[w]); };
  auto seen_or = [&](uint32_t id) {
    const uint32_t bit = 1u << (id & 31), w = id >> 5;
    if (use_dirty) {
      if (__hip_atomic_fetch_or(&seen[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(&dirty[w >> 5], 1u << (w & 31));
    } else {
      __hip_atomic_fetch_or(&seen[w], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto zero_dirty = [&]() {
    for (uint32_t d = (uint32_t)tid; d < DW; d += NT) dirty[d] = 0u;
  };
  // the touched words of this thread in ascending order, four at a time (so that their loads are in flight together):
  // body(w[4], n) with n <= 4 valid word indices
  auto for_dirty4 = [&](auto&& body) {
    for (int dd = 0; dd < D; ++dd) {
      const uint32_t d = d0 + (uint32_t)dd;
      uint32_t dw = d < DW ? dirty[d] : 0u;
      while (dw) {
        uint32_t w[4];
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (dw) { w[k] = d * 32u + (uint32_t)(__ffs(dw) - 1); dw &= dw - 1; n = k + 1; } else w[k] = 0u;
        }
        body(w, n);
      }
    }
  };
  if (clear_seen) {  // the slot's first user, or the one behind a user that failed with bits set: every other user leaves `seen` zero
    uint4* s4 = reinterpret_cast<uint4*>(seen);
    for (uint32_t i = (uint32_t)tid; i < a.bm_words / 4u; i += NT) s4[i] = uint4{0u, 0u, 0u, 0u};
  }
  if (tid < 2) SS->flags[tid] = 0;
  __syncthreads();
  float mlp_u = 0.0f;
  if constexpr (SC == NANN_SCORER_MLP) mlp_u = wg_mlp_query_u<NT>(a.mlp, qv);

  // start level: score every enter point, keep min(k, n) (:349-353)
  const int E = a.n_enter;
  if (E <= 0) return NANN_ERR_EMPTY_SCORE_BATCH;
  eval_score<LPR, DT, SC, NT>(a, qi, a.enter, E, sv.cat_sc, scratch, qv, mlp_u);
  ctr_s += E;
  EVAL_TICK(0);
  int n_res = min(a.top_k[2], E);
  int st = wg_topk<NT, kEvalMaxK>(a.enter, sv.cat_sc, nullptr, E, n_res, nullptr, sv.res_ids, sv.res_sc, nullptr, nullptr, scratch);
  if (st) return st;
  EVAL_TICK(1);

  for (int level = 1; level >= 0; --level) {  // search_level (:299-337)
    if (tid < 2) SS->flags[tid] = 0;  // (the scratch was reused since)
    if (use_dirty) {
      zero_dirty();
      // visited = {} for the level: plain 16-byte stores by everybody (bm_words is a multiple of 4); the owners' stores of the
      // marks below are ordered behind them by the barrier
      uint4* v4 = reinterpret_cast<uint4*>(sv.visited);
      for (uint32_t i = (uint32_t)tid; i < a.vis_words / 4u; i += NT) v4[i] = uint4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // visited = idx_ep (:311): the marks go through `seen`; result -> front of the concat arrays; candidates = result
    for (int i = tid; i < n_res; i += NT) {
      const int32_t id = sv.res_ids[i];
      if ((uint32_t)id >= a.n_items) { SS->flags[1] = 1; continue; }
      seen_or((uint32_t)id);
      sv.cand[i] = id;
      sv.cat_ids[i] = id;
      sv.cat_sc[i] = sv.res_sc[i];
    }
    __syncthreads();
    if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
    if (use_dirty) {  // the owners of the marked words: visited = marks, seen = 0
      for_dirty4([&](const uint32_t (&w)[4], int n) {
        uint32_t sw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sw[k] = k < n ? seen_load(w[k]) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k < n) { sv.visited[vp(w[k])] = sw[k]; seen[w[k]] = 0u; }
      });
      __syncthreads();  // (every owner has read its dirty words)
      zero_dirty();
    } else {
      for (int j = 0; j < J; ++j) {  // the owners: visited = marks, seen = 0
        const uint32_t w = w0 + 64u * j;
        if (w < a.bm_words) {
          const uint32_t s = seen_load(w);
          sv.visited[w] = s;
          if (s) seen[w] = 0u;
        }
      }
    }
    __syncthreads();
    EVAL_TICK(2);
    int n_cand = n_res;
    const int32_t* __restrict__ values = a.nbv[level];
    const int64_t* __restrict__ rs = a.nbrs[level];
    for (int it = 0; it < a.num_scoring[level]; ++it) {
      // ---- neighbours of the candidates -> bits of `seen`: eight rows per wavefront and trip
      ctr_f += n_cand;
      for (int rep = 0; rep <= NANN_REPEAT_GATHER; ++rep) {
        int bad = 0;
        long long s_nx = 0, e_nx = 0;  // the bounds of the NEXT trip's rows are fetched under this trip's values and atomics
        if (lane < 8 && wave * 8 + lane < n_cand) {
          const int32_t c = sv.cand[wave * 8 + lane];
          s_nx = rs[c]; e_nx = rs[c + 1];
        }
        for (int base = wave * 8; base < n_cand; base += NW * 8) {
          const long long s = s_nx, e = e_nx;
          if (rep == 0) ctr_g += (int)(e - s);
          s_nx = 0; e_nx = 0;
          if (lane < 8 && base + NW * 8 + lane < n_cand) {
            const int32_t c = sv.cand[base + NW * 8 + lane];
            s_nx = rs[c]; e_nx = rs[c + 1];
          }
          int32_t v[8];
          long long sr[8], er[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            sr[r] = __shfl(s, r); er[r] = __shfl(e, r);
            v[r] = (sr[r] + lane < er[r]) ? values[sr[r] + lane] : -1;
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            if (sr[r] + lane < er[r]) {
              if ((uint32_t)v[r] < a.n_items) seen_or((uint32_t)v[r]); else bad = 1;
            }
            for (long long j = sr[r] + 64 + lane; j < er[r]; j += 64) {  // (rows of more than 64 neighbours)
              const int32_t x = values[j];
              if ((uint32_t)x < a.n_items) seen_or((uint32_t)x); else bad = 1;
            }
          }
        }
        if (bad) SS->flags[1] = 1;
      }
      __syncthreads();
      if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
      EVAL_TICK(3);
      // ---- new = seen & ~visited, in ascending id order (:316-319); visited |= new (:321), seen = 0
      int n_next = 0;
      if (use_dirty) {
        // pass 1, this thread's touched words only: count the new bits
        uint32_t cnt = 0;
        for_dirty4([&](const uint32_t (&w)[4], int n) {
          uint32_t vis[4], sw[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { vis[k] = k < n ? sv.visited[vp(w[k])] : 0u; sw[k] = k < n ? seen_load(w[k]) : 0u; }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < n) cnt += (uint32_t)__popc(sw[k] & ~vis[k]);
        });
        uint32_t total;
        uint32_t at = wg_excl_scan<NT>(cnt, SS, &total);  // thread order = word order = ascending ids
        n_next = (int)total;
        ctr_s += n_next;
        if (n_res + n_next > a.cat_cap) return NANN_ERR_CAPACITY;
        // pass 2: the ids; seen = 0 for the next round
        int32_t* dst = sv.cat_ids + n_res;
        for_dirty4([&](const uint32_t (&w)[4], int n) {
          // both words again (L2 hits) instead of a store -> load of the same word through L2
          uint32_t nw[4], vis[4], sw[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { vis[k] = k < n ? sv.visited[vp(w[k])] : 0u; sw[k] = k < n ? seen_load(w[k]) : 0u; }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            nw[k] = sw[k] & ~vis[k];
            if (k < n && nw[k]) sv.visited[vp(w[k])] = vis[k] | nw[k];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < n) {
              seen[w[k]] = 0u;
              uint32_t x = nw[k];
              while (x) {
                dst[at++] = (int32_t)(w[k] * 32u + (uint32_t)(__ffs(x) - 1));
                x &= x - 1;
              }
            }
        });
        __syncthreads();  // (every owner has read its dirty words)
        zero_dirty();
      } else {
        uint32_t cnt = 0;
        for (int j = 0; j < J; ++j) {
          const uint32_t w = w0 + 64u * j;
          if (w < a.bm_words) {
            const uint32_t s = seen_load(w);
            if (s) cnt += (uint32_t)__popc(s & ~sv.visited[w]);
          }
        }
        const uint32_t wtot = wave_total(wave_scan_add(cnt));
        if (lane == 0) SS->wave_tot[wave] = wtot;
        __syncthreads();
        uint32_t run = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
          const uint32_t t = SS->wave_tot[w];
          if (w < wave) run += t;
          total += t;
        }
        n_next = (int)total;
        ctr_s += n_next;
        if (n_res + n_next > a.cat_cap) return NANN_ERR_CAPACITY;
        int32_t* dst = sv.cat_ids + n_res;
        for (int j = 0; j < J; ++j) {  // word order = ascending ids; every bitmap word has ONE owner thread
          const uint32_t w = w0 + 64u * j;
          uint32_t nw = 0;
          if (w < a.bm_words) {
            const uint32_t s = seen_load(w);
            if (s) {
              const uint32_t vis = sv.visited[w];
              nw = s & ~vis;
              sv.visited[w] = vis | s;
              seen[w] = 0u;
            }
          }
          const uint32_t c = (uint32_t)__popc(nw);
          const uint32_t inc = wave_scan_add(c);
          uint32_t at = run + inc - c;
          while (nw) {
            const int b = __ffs(nw) - 1;
            nw &= nw - 1;
            dst[at++] = (int32_t)(w * 32u + (uint32_t)b);
          }
          run += wave_total(inc);
        }
      }
      __syncthreads();
      EVAL_TICK(4);
      if (n_next == 0) {  // plain TF ops score an empty batch as an empty tensor: the result is cut to min(k, n), no candidate is left
        n_res = min(a.top_k[level], n_res);
        n_cand = 0;
        continue;
      }
      eval_score<LPR, DT, SC, NT>(a, qi, sv.cat_ids + n_res, n_next, sv.cat_sc + n_res, scratch, qv, mlp_u);  // :323
#if NANN_REPEAT_SCORE  // measurement builds (tools/build_res_variant.py --unit nann_eval_inst.hip): a phase run twice costs what it costs once
      eval_score<LPR, DT, SC, NT>(a, qi, sv.cat_ids + n_res, n_next, sv.cat_sc + n_res, scratch, qv, mlp_u);
#endif
      EVAL_TICK(5);
      const int n_cat = n_res + n_next;
      const int k = min(a.top_k[level], n_cat);
#if NANN_REPEAT_TOPK
      (void)wg_topk<NT, kEvalMaxK>(sv.cat_ids, sv.cat_sc, nullptr, n_cat, k, nullptr, sv.res_ids, sv.res_sc, nullptr, nullptr,
                                  scratch);
      __syncthreads();
#endif
      st = wg_topk<NT, kEvalMaxK>(sv.cat_ids, sv.cat_sc, nullptr, n_cat, k, nullptr, sv.res_ids, sv.res_sc, nullptr, nullptr,
                                  scratch);  // :326-328
      if (st) return st;
      EVAL_TICK(6);
      // next frontier: new nodes scoring at least the worst kept result, in id order (:330-331).  Round 6: every thread takes a
      // CONTIGUOUS run of the new nodes, so one workgroup scan places them (it was one scan, two barriers, per 1024 nodes)
      const float worst = sv.res_sc[k - 1];
      if (it + 1 < a.num_scoring[level]) {  // (a level's last round: its frontier is never walked -- and cannot overflow)
        const int per = (n_next + NT - 1) / NT;
        const int lo = min(tid * per, n_next), hi = min(lo + per, n_next);
        uint32_t mine = 0;
        for (int i = lo; i < hi; ++i) mine += sv.cat_sc[n_res + i] >= worst ? 1u : 0u;
        uint32_t n_new;
        uint32_t pos = wg_excl_scan<NT>(mine, SS, &n_new);
        if (n_new > (uint32_t)kEvalMaxK) return NANN_ERR_CAPACITY;  // more ties at the threshold than a frontier holds
        for (int i = lo; i < hi; ++i)
          if (sv.cat_sc[n_res + i] >= worst) sv.cand[pos++] = sv.cat_ids[n_res + i];
        n_cand = (int)n_new;
      }
      __syncthreads();  // (the copy below overwrites cat_ids[n_res ..] that the selection above reads)
      n_res = k;
      for (int i = tid; i < k; i += NT) {
        sv.cat_ids[i] = sv.res_ids[i];
        sv.cat_sc[i] = sv.res_sc[i];
      }
      if (tid < 2) SS->flags[tid] = 0;  // (the scratch was reused since)
      if (use_dirty) zero_dirty();
      __syncthreads();
      EVAL_TICK(7);
    }
  }
  EVAL_TICK_FLUSH;
  *n_result = n_res;
  if (a.counters) {  // (the kernel zeroed the user's three words before the call)
    if (tid == 0) { atomicAdd(&a.counters[(size_t)qi * 3 + 0], ctr_f); atomicAdd(&a.counters[(size_t)qi * 3 + 2], ctr_s); }
    if (ctr_g) atomicAdd(&a.counters[(size_t)qi * 3 + 1], ctr_g);
  }
  return NANN_OK;
}

__device__ __forceinline__ long long readlane64(long long v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v & 0xffffffffull), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), l);
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// ---- the LDS form (L2 scorer, shards of up to 2^20 items): `seen` in LDS, one dirty word per thread (its own 32 words of both
// bitmaps).  Round 6, third step: with one 1 024-thread workgroup per CU nothing hides a dependent trip to the slot (1-3 us under
// the other CUs' scoring), and a round took ~20 of them -- the frontier, its row bounds and rows one after the other, `visited`,
// the new ids out and in again, their scores out and in again for top-k, its results out and in again for the threshold, the
// frontier out, the results to the front of the concat arrays.  `seen` is ALL ZERO outside gather -> emit, so its LDS doubles as a
// STAGING AREA from a round's emit to its end: the round's scores (result || new, n_cat <= lds_words - 4096: 28 K for 2^20 items,
// else they stay in the slot) and top-k's output live there, the frontier lives in the phase scratch; the extents used are zeroed
// again at the round's end.  Only ids cross to the slot (emit -> score / top-k gather / threshold: prefetched, off the critical
// path), the results' copy to the front of the concat arrays is fire-and-forget behind an LDS-only barrier, and the walk of a
// frontier fetches ALL its row bounds, then 32 rows per wavefront in flight together.  Same sets, same orders, same arithmetic:
// bit-identical to the oracle as before.
template <int LPR, int DT, int NT>
__device__ __forceinline__ int search_eval_lds(const EvalArgs& a, int qi, const EvalSlot& sv, uint32_t* seen,
                                               unsigned char* scratch, float* qv, int* n_result, bool clear_seen) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  int ctr_f = 0, ctr_s = 0;  // (uniform)
  int ctr_g = 0;             // this lane's share: row lengths it fetched (lanes 0-7 of every wavefront)
  constexpr int NW = NT / 64;
  constexpr int NF = kEvalMaxK / NT;  // kept results per thread
  static_assert(kEvalMaxK % NT == 0 && NF >= 1, "a thread carries kEvalMaxK / NT scores of the kept results");
  // phase scratch: [scan scratch 256 | frontier: kEvalMaxK ids | ...] (top-k's scratch overlays all of it: the frontier is
  // written behind it)
  EvalScanScratch* SS = reinterpret_cast<EvalScanScratch*>(scratch);
  int32_t* cand = reinterpret_cast<int32_t*>(scratch + 256);
  // the frontier's row bounds for the walk: start (48 bits) | length (16 bits; eval_plan: the LDS form takes indices whose rows
  // are shorter than 2^16)
  unsigned long long* bnd = reinterpret_cast<unsigned long long*>(scratch + 256 + kEvalMaxK * 4);
  static_assert(256 + kEvalMaxK * 12 <= kPhaseScratch, "phase scratch too small for the frontier and its row bounds");
  // staging area (the region of `seen`): [kept ids: kEvalMaxK | kept scores: kEvalMaxK | scores of result || new: CAP]
  int32_t* st_res_ids = reinterpret_cast<int32_t*>(seen);
  float* st_res_sc = reinterpret_cast<float*>(seen + kEvalMaxK);
  float* st_cat_sc = reinterpret_cast<float*>(seen + 2 * kEvalMaxK);
  const int CAP = (int)a.lds_words - 2 * kEvalMaxK;  // scores only: result || new up to here
  const int CAP2 = CAP / 2;                          // scores and ids: [scores: CAP2 | ids: CAP2]
  int32_t* st_cat_ids = reinterpret_cast<int32_t*>(seen + 2 * kEvalMaxK + CAP2);
  // the ascending list of new ids is written while `seen` is still being read: into the phase scratch (the frontier is dead
  // by then), and moved to the staging area behind the barrier
  int32_t* emit_lds = cand;
  constexpr int kEmitCap = (kPhaseScratch - 256) / 4;
  EVAL_TICK_DECL;
  for (int k = tid; k < a.d; k += NT) qv[k] = a.q[(size_t)qi * a.d + k];
  // Thread t owns words [32 t, 32 t + 32) of both bitmaps (t < DW <= NT, eval_plan), so thread order is word order.  `seen` is
  // SKEWED by one word per 32 (word w at w + w / 32 = 33 t + j): the owners read their j-th words together, lanes 33 words apart
  // -- conflict-free; without the skew, two LDS banks for the whole wavefront.  `visited` (the slot, only ever touched by its
  // owners) is TRANSPOSED: word j of thread t at j NT + t, so that a wavefront's loads of "my j-th word" are 256 contiguous
  // bytes.  An owner reads ALL 32 of its words of `seen`, eight at a time in flight (a word's read behind the test of a
  // second-level "touched" bit was one LDS round trip per word, 32 in a row; and the bit cost the walk a second atomic per word).
  const uint32_t DW = (a.bm_words + 31u) >> 5;
  const bool owner = (uint32_t)tid < DW;
  const uint32_t own = (uint32_t)tid * 33u;
  auto seen_or = [&](uint32_t id) {
    const uint32_t w = id >> 5;
    atomicOr(&seen[w + (w >> 5)], 1u << (id & 31));
  };
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(sv.visited, 0, (int)(a.vis_words * 4u), 0x00020000);
  if (clear_seen) {  // the slot's first user, or the one behind a user that failed with the region in use: every other user leaves it zero
    for (uint32_t w = (uint32_t)tid; w < a.lds_words; w += NT) seen[w] = 0u;
  }
  if (tid < 2) SS->flags[tid] = 0;
  __syncthreads();

  // the kept results leave the staging area: ids and scores to the front of the slot's concat arrays (fire and forget: the next
  // reader is a full barrier away), the ids to the frontier when the results ARE the next frontier (level start, :308); then the
  // extents of the region that the round used are zeroed -- `seen` again
  auto publish = [&](int k, int n_staged, int n_staged_ids, bool to_cand) {
    for (int i = tid; i < k; i += NT) {
      const int32_t id = st_res_ids[i];
      sv.cat_ids[i] = id;
      sv.cat_sc[i] = st_res_sc[i];
      if (to_cand) { cand[i] = id; sv.res_ids[i] = id; }  // (the next level's frontier and its marks, below)
    }
    lds_barrier();
    for (int i = tid; i < k; i += NT) { seen[i] = 0u; seen[kEvalMaxK + i] = 0u; }
    for (int i = tid; i < n_staged; i += NT) seen[2 * kEvalMaxK + i] = 0u;
    for (int i = tid; i < n_staged_ids; i += NT) seen[2 * kEvalMaxK + CAP2 + i] = 0u;
    if (tid < 2) SS->flags[tid] = 0;
    lds_barrier();
  };

  // start level: score every enter point, keep min(k, n) (:349-353)
  const int E = a.n_enter;
  if (E <= 0) return NANN_ERR_EMPTY_SCORE_BATCH;
  const bool near = (unsigned long long)a.n_items * (unsigned)(a.d * 2) <= 0xffffffffull && a.n_items <= (1u << 24);
  wg_score_l2_part<LPR, DT, NW>(a.emb, a.d, a.enter, 0, E, qv, sv.cat_sc, wave, near);
  __syncthreads();
  ctr_s += E;
  EVAL_TICK(0);
  int n_res = min(a.top_k[2], E);
  int st = wg_topk_binned<NT, kEvalMaxK>(a.enter, sv.cat_sc, E, n_res, st_res_ids, st_res_sc, scratch);
  if (st) return st;
  EVAL_TICK(1);
  publish(n_res, 0, 0, true);
  bool cand_is_result = true;  // (uniform) the frontier array holds the kept ids

  for (int level = 1; level >= 0; --level) {  // search_level (:299-337)
    if (!cand_is_result) {  // (a level that ended on an empty round, or ran no round: its result, cut, from the slot)
      __syncthreads();
      for (int i = tid; i < n_res; i += NT) {
        const int32_t id = sv.cat_ids[i];
        cand[i] = id;
        sv.res_ids[i] = id;
      }
      __syncthreads();
    }
    // visited = idx_ep (:311), candidates = result.  The level's marks -- its starting result, <= top_k ids -- are NOT put into
    // `visited`: they stay a list (the slot's result array; a thread reads back the entries it wrote) and are taken out of
    // `seen` behind every walk, one atomic each.  So a level starts without a pass over the bitmaps, and its first round needs
    // no word of `visited`: it WRITES all of them (visited = new), which is also the level's visited = {}.
    const int n_marks = n_res;
    EVAL_TICK(2);
    int n_cand = n_res;
    const int32_t* __restrict__ values = a.nbv[level];
    const int64_t* __restrict__ rs = a.nbrs[level];
    for (int it = 0; it < a.num_scoring[level]; ++it) {
      // ---- neighbours of the candidates -> bits of `seen`.  Every thread fetches the bounds of its rows of the frontier (one
      // trip for all of them) into LDS; then wavefront w walks rows [8 (w + NW t), + 8) for t = 0, 1, ..., four trips at a time:
      // 32 rows per wavefront in flight together.  The level's marks ride along, for the removal behind the walk.
      ctr_f += n_cand;
      const bool first = it == 0;
      int32_t mk[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) mk[j] = sv.res_ids[min(tid + j * NT, n_marks - 1)];
      {
        long long s[NF], e[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = tid + j * NT;
          s[j] = 0; e[j] = 0;
          if (r < n_cand) {
            const int32_t c = cand[r];
            s[j] = rs[c]; e[j] = rs[c + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = tid + j * NT;
          if (r < n_cand) {
            bnd[r] = (unsigned long long)s[j] | ((unsigned long long)(e[j] - s[j]) << 48);
            ctr_g += (int)(e[j] - s[j]);
          }
        }
      }
      lds_barrier();
      for (int rep = 0; rep <= NANN_REPEAT_GATHER; ++rep) {
        int bad = 0;
        for (int t0 = 0; t0 * NW * 8 < n_cand; t0 += 4) {
          long long s4[4];
          int len4[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = (t0 + t) * NW * 8 + wave * 8 + lane;
            unsigned long long b = 0ull;
            if (lane < 8 && r < n_cand) b = bnd[r];
            s4[t] = (long long)(b & 0xffffffffffffull); len4[t] = (int)(b >> 48);
          }
          int32_t v[4][8];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const long long sr = readlane64(s4[t], r);
              const int len = __builtin_amdgcn_readlane(len4[t], r);
              v[t][r] = lane < len ? values[sr + lane] : 0;
            }
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const int len = __builtin_amdgcn_readlane(len4[t], r);
              if (lane < len) {
                if ((uint32_t)v[t][r] < a.n_items) seen_or((uint32_t)v[t][r]); else bad = 1;
              }
              if (len > 64) {  // (rows of more than 64 neighbours)
                const long long sr = readlane64(s4[t], r);
                for (int j = 64 + lane; j < len; j += 64) {
                  const int32_t x = values[sr + j];
                  if ((uint32_t)x < a.n_items) seen_or((uint32_t)x); else bad = 1;
                }
              }
            }
        }
        if (bad) SS->flags[1] = 1;
      }
      lds_barrier();  // (the walk loads; what is still on its way to the slot -- the last round's results, an owner's words of
                      //  `visited` -- is read back by the thread that stored it)
#pragma unroll
      for (int j = 0; j < NF; ++j)  // seen \= marks
        if (tid + j * NT < n_marks) {
          const uint32_t id = (uint32_t)mk[j];
          if (id >= a.n_items) SS->flags[1] = 1;
          else atomicAnd(&seen[(id >> 5) + (id >> 10)], ~(1u << (id & 31)));
        }
      lds_barrier();
      if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
      EVAL_TICK(3);
      // ---- new = seen & ~visited, in ascending id order (:316-319); visited |= new (:321), seen = 0
      // ALL 32 of the thread's words of `visited` in ONE batch (touched words only, 4 / 8 / 16 at a time, were 5 / 4 / 2 dependent
      // trips; the level's first round has none to fetch) -- and with them this thread's ids and scores of the kept results (the
      // front of the concat arrays), which go to the staging area behind the emit.  Buffer loads: one VGPR of offset for all of
      // them, the word's stride in an SGPR.
      int32_t fi[NF];
      float fs[NF];
      uint32_t cnt = 0, nd = 0;  // new bits of this thread's words; which of its words have any
      auto owners = [&](auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        uint32_t vis[32];
        if constexpr (!FIRST) {
#pragma unroll
          for (int j = 0; j < 32; ++j) vis[j] = __builtin_amdgcn_raw_buffer_load_b32(vrs, tid * 4, j * NT * 4, 0);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          fi[j] = sv.cat_ids[min(tid + j * NT, n_res - 1)];
          fs[j] = sv.cat_sc[min(tid + j * NT, n_res - 1)];
        }
        if (owner) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            uint32_t sw[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sw[i] = seen[own + 8 * b + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int j = 8 * b + i;
              if constexpr (FIRST) {  // visited = new: every word written
                __builtin_amdgcn_raw_buffer_store_b32(sw[i], vrs, tid * 4, j * NT * 4, 0);
                if (sw[i]) { nd |= 1u << j; cnt += (uint32_t)__popc(sw[i]); }
              } else {
                const uint32_t nw = sw[i] & ~vis[j];
                if (sw[i] != nw) seen[own + j] = nw;  // the NEW bits stay, for the emit behind the scan
                if (nw) {
                  __builtin_amdgcn_raw_buffer_store_b32(vis[j] | nw, vrs, tid * 4, j * NT * 4, 0);
                  nd |= 1u << j;
                  cnt += (uint32_t)__popc(nw);
                }
              }
            }
          }
        }
      };
      if (first) owners(std::true_type{}); else owners(std::false_type{});
      EVAL_TICK(8);
      uint32_t total;
      uint32_t at = wg_excl_scan<NT, true>(cnt, SS, &total);  // thread order = word order = ascending ids
      const int n_next = (int)total;
      ctr_s += n_next;
      if (n_res + n_next > a.cat_cap) return NANN_ERR_CAPACITY;
      EVAL_TICK(9);
      if (n_next == 0) {  // plain TF ops score an empty batch as an empty tensor: the result is cut to min(k, n), no candidate is left
        __syncthreads();  // (`seen` is zero again: the owners wrote their words' new bits, none)
        n_res = min(a.top_k[level], n_res);
        n_cand = 0;
        cand_is_result = false;
        continue;
      }
      const int n_cat = n_res + n_next;
      // where the round's ids and scores live: 2 = both in the staging area, 1 = the scores, 0 = neither (the slot, as the slot form)
      const int mode = (n_next <= kEmitCap && n_cat <= CAP2) ? 2 : n_cat <= CAP ? 1 : 0;
      auto emit = [&](int32_t* dst) {  // this thread's words with new bits, four reads in flight; seen = 0 behind them
        uint32_t m = nd;
        while (m) {
          uint32_t j[4], x[4];
          int n = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (m) { j[u] = (uint32_t)(__ffs(m) - 1); m &= m - 1; n = u + 1; } else j[u] = j[0];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) x[u] = seen[own + j[u]];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (u < n) {
              seen[own + j[u]] = 0u;
              const uint32_t w = (uint32_t)tid * 32u + j[u];
              uint32_t y = x[u];
              while (y) {
                dst[at++] = (int32_t)(w * 32u + (uint32_t)(__ffs(y) - 1));
                y &= y - 1;
              }
            }
        }
      };
      if (mode == 2) {
        if (cnt) emit(emit_lds);
        lds_barrier();  // `seen` is zero: the region is the staging area from here
        for (int i = tid; i < n_next; i += NT) st_cat_ids[n_res + i] = emit_lds[i];
#pragma unroll
        for (int j = 0; j < NF; ++j)
          if (tid + j * NT < n_res) { st_cat_ids[tid + j * NT] = fi[j]; st_cat_sc[tid + j * NT] = fs[j]; }
        lds_barrier();
      } else {
        if (cnt) emit(sv.cat_ids + n_res);
        __syncthreads();  // the new ids are in the slot, `seen` is zero: the region is the staging area from here
        if (mode == 1) {
#pragma unroll
          for (int j = 0; j < NF; ++j)
            if (tid + j * NT < n_res) st_cat_sc[tid + j * NT] = fs[j];
        }
      }
      EVAL_TICK(4);
      const int k = min(a.top_k[level], n_cat);
      const bool last = it + 1 == a.num_scoring[level];  // the level's last round: its frontier is never walked
      auto rest = [&](auto mode_c) -> int {
        constexpr int MODE = decltype(mode_c)::value;
        const int32_t* cat_ids;
        float* cat_sc;
        if constexpr (MODE == 2) cat_ids = st_cat_ids; else cat_ids = sv.cat_ids;
        if constexpr (MODE >= 1) cat_sc = st_cat_sc; else cat_sc = sv.cat_sc;
        for (int rep = 0; rep <= NANN_REPEAT_SCORE; ++rep) {  // :323
          wg_score_l2_part<LPR, DT, NW>(a.emb, a.d, cat_ids + n_res, 0, n_next, qv, cat_sc + n_res, wave, near);
          __syncthreads();
        }
        EVAL_TICK(5);
        int rc = 0;
        // (the kept results are sorted and, once there are k of them, only candidates that beat the worst one can enter)
        const bool full = n_res >= k;
        const uint32_t floor_key = full ? score_key(cat_sc[n_res - 1]) : 0u;
        for (int rep = 0; rep <= NANN_REPEAT_TOPK; ++rep) {  // :326-328
          rc = wg_topk_binned<NT, kEvalMaxK>(cat_ids, cat_sc, n_cat, k, st_res_ids, st_res_sc, scratch, full ? n_res : 0x7fffffff, floor_key);
          if (rc) return rc;
        }
        EVAL_TICK(6);
        if (!last) {
          // next frontier: new nodes scoring at least the worst kept result, in id order (:330-331): every thread takes a
          // CONTIGUOUS run of the new nodes, one workgroup scan places them
          const float worst = st_res_sc[k - 1];
          const int per = (n_next + NT - 1) / NT;
          const int lo = min(tid * per, n_next), hi = min(lo + per, n_next);
          uint32_t mine = 0;
          for (int i0 = lo; i0 < hi; i0 += 8) {  // (eight scores in flight: they may be in the slot)
            float sc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sc[u] = cat_sc[n_res + min(i0 + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) mine += (i0 + u < hi && sc[u] >= worst) ? 1u : 0u;
          }
          uint32_t n_new;
          uint32_t pos = wg_excl_scan<NT, true>(mine, SS, &n_new);
          if (n_new > (uint32_t)kEvalMaxK) return NANN_ERR_CAPACITY;  // more ties at the threshold than a frontier holds
          if (mine) {
            for (int i0 = lo; i0 < hi; i0 += 8) {  // the POSITIONS of the frontier's rows first ...
              float sc[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) sc[u] = cat_sc[n_res + min(i0 + u, hi - 1)];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (i0 + u < hi && sc[u] >= worst) cand[pos++] = i0 + u;
            }
          }
          lds_barrier();
          for (int p = tid; p < (int)n_new; p += NT) cand[p] = cat_ids[n_res + cand[p]];  // ... then their ids, one trip for all
          n_cand = (int)n_new;
          __syncthreads();  // (the copy below overwrites cat_ids[n_res ..] that the selection above reads)
        } else {
          n_cand = k;
        }
        publish(k, MODE >= 1 ? n_cat : 0, MODE == 2 ? n_cat : 0, last);
        return 0;
      };
      st = mode == 2 ? rest(std::integral_constant<int, 2>{}) : mode == 1 ? rest(std::integral_constant<int, 1>{}) : rest(std::integral_constant<int, 0>{});
      if (st) return st;
      n_res = k;
      cand_is_result = last;
      EVAL_TICK(7);
    }
  }
  EVAL_TICK_FLUSH;
  *n_result = n_res;
  if (a.counters) {  // (the kernel zeroed the user's three words before the call)
    if (tid == 0) { atomicAdd(&a.counters[(size_t)qi * 3 + 0], ctr_f); atomicAdd(&a.counters[(size_t)qi * 3 + 2], ctr_s); }
    if (ctr_g) atomicAdd(&a.counters[(size_t)qi * 3 + 1], ctr_g);
  }
  return NANN_OK;
}

// ---- the LDS form sweeping the id space in WINDOWS (shards of ~1 M to ~8 M items; round 6).  The one-window function above with
// the walk -> marks -> owners' pass -> scan -> emit of a round once per window of kEvalWinOwners x 32 bitmap words: the window's
// `seen` bits in LDS, its words of `visited` in the slot, the frontier's bounds (LDS) fetched once and its rows walked once per
// window (L2 hits), the new ids to the slot window after window -- ascending across windows as inside one.  A function of its
// own: as a flag of the one-window function the sweep's loop cost that kernel 15 registers, 60 B of spills and 3 %.
template <int LPR, int DT, int NT>
__device__ __forceinline__ int search_eval_win(const EvalArgs& a, int qi, const EvalSlot& sv, uint32_t* seen,
                                               unsigned char* scratch, float* qv, int* n_result, bool clear_seen) {
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  int ctr_f = 0, ctr_s = 0;  // (uniform)
  int ctr_g = 0;             // this lane's share: row lengths it fetched (lanes 0-7 of every wavefront)
  constexpr int NW = NT / 64;
  constexpr int NF = kEvalMaxK / NT;  // kept results per thread
  static_assert(kEvalMaxK % NT == 0 && NF >= 1, "a thread carries kEvalMaxK / NT scores of the kept results");
  // phase scratch: [scan scratch 256 | frontier: kEvalMaxK ids | ...] (top-k's scratch overlays all of it: the frontier is
  // written behind it)
  EvalScanScratch* SS = reinterpret_cast<EvalScanScratch*>(scratch);
  int32_t* cand = reinterpret_cast<int32_t*>(scratch + 256);
  // the frontier's row bounds for the walk: start (48 bits) | length (16 bits; eval_plan: the LDS form takes indices whose rows
  // are shorter than 2^16)
  unsigned long long* bnd = reinterpret_cast<unsigned long long*>(scratch + 256 + kEvalMaxK * 4);
  static_assert(256 + kEvalMaxK * 12 <= kPhaseScratch, "phase scratch too small for the frontier and its row bounds");
  // staging area (the region of `seen`): [kept ids: kEvalMaxK | kept scores: kEvalMaxK | scores of result || new: CAP]
  int32_t* st_res_ids = reinterpret_cast<int32_t*>(seen);
  float* st_res_sc = reinterpret_cast<float*>(seen + kEvalMaxK);
  float* st_cat_sc = reinterpret_cast<float*>(seen + 2 * kEvalMaxK);
  const int CAP = (int)a.lds_words - 2 * kEvalMaxK;  // scores only: result || new up to here
  const int CAP2 = CAP / 2;                          // scores and ids: [scores: CAP2 | ids: CAP2]
  int32_t* st_cat_ids = reinterpret_cast<int32_t*>(seen + 2 * kEvalMaxK + CAP2);
  // the ascending list of new ids is written while `seen` is still being read: into the phase scratch (the frontier is dead
  // by then), and moved to the staging area behind the barrier
  int32_t* emit_lds = cand;
  constexpr int kEmitCap = (kPhaseScratch - 256) / 4;
  EVAL_TICK_DECL;
  for (int k = tid; k < a.d; k += NT) qv[k] = a.q[(size_t)qi * a.d + k];
  // Thread t owns words [32 t, 32 t + 32) of both bitmaps' current window (t < OW <= NT), so thread order is word order.  `seen` is
  // SKEWED by one word per 32 (word w at w + w / 32 = 33 t + j): the owners read their j-th words together, lanes 33 words apart
  // -- conflict-free; without the skew, two LDS banks for the whole wavefront.  `visited` (the slot, only ever touched by its
  // owners) is TRANSPOSED: word j of thread t at j NT + t, so that a wavefront's loads of "my j-th word" are 256 contiguous
  // bytes.  An owner reads ALL 32 of its words of `seen`, eight at a time in flight (a word's read behind the test of a
  // second-level "touched" bit was one LDS round trip per word, 32 in a row; and the bit cost the walk a second atomic per word).
  const uint32_t DW = (a.bm_words + 31u) >> 5;   // owners' worth of bitmap words in the whole index
  const uint32_t OW = a.win_owners;  // ... in a window (kEvalWinOwners, or all of them)
  constexpr bool MULTI = true;
  const int W = a.n_windows;  // windows of a round
  const uint32_t own = (uint32_t)tid * 33u;
  const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(sv.visited, 0, (int)(a.vis_words * 4u), 0x00020000);
  if (clear_seen) {  // the slot's first user, or the one behind a user that failed with the region in use: every other user leaves it zero
    for (uint32_t w = (uint32_t)tid; w < a.lds_words; w += NT) seen[w] = 0u;
  }
  if (tid < 2) SS->flags[tid] = 0;
  __syncthreads();

  // the kept results leave the staging area: ids and scores to the front of the slot's concat arrays (fire and forget: the next
  // reader is a full barrier away), the ids to the frontier when the results ARE the next frontier (level start, :308); then the
  // extents of the region that the round used are zeroed -- `seen` again
  auto publish = [&](int k, int n_staged, int n_staged_ids, bool to_cand) {
    for (int i = tid; i < k; i += NT) {
      const int32_t id = st_res_ids[i];
      sv.cat_ids[i] = id;
      sv.cat_sc[i] = st_res_sc[i];
      if (to_cand) { cand[i] = id; sv.res_ids[i] = id; }  // (the next level's frontier and its marks, below)
    }
    lds_barrier();
    for (int i = tid; i < k; i += NT) { seen[i] = 0u; seen[kEvalMaxK + i] = 0u; }
    for (int i = tid; i < n_staged; i += NT) seen[2 * kEvalMaxK + i] = 0u;
    for (int i = tid; i < n_staged_ids; i += NT) seen[2 * kEvalMaxK + CAP2 + i] = 0u;
    if (tid < 2) SS->flags[tid] = 0;
    lds_barrier();
  };

  // start level: score every enter point, keep min(k, n) (:349-353)
  const int E = a.n_enter;
  if (E <= 0) return NANN_ERR_EMPTY_SCORE_BATCH;
  const bool near = (unsigned long long)a.n_items * (unsigned)(a.d * 2) <= 0xffffffffull && a.n_items <= (1u << 24);
  wg_score_l2_part<LPR, DT, NW>(a.emb, a.d, a.enter, 0, E, qv, sv.cat_sc, wave, near);
  __syncthreads();
  ctr_s += E;
  EVAL_TICK(0);
  int n_res = min(a.top_k[2], E);
  int st = wg_topk_binned<NT, kEvalMaxK>(a.enter, sv.cat_sc, E, n_res, st_res_ids, st_res_sc, scratch);
  if (st) return st;
  EVAL_TICK(1);
  publish(n_res, 0, 0, true);
  bool cand_is_result = true;  // (uniform) the frontier array holds the kept ids

  for (int level = 1; level >= 0; --level) {  // search_level (:299-337)
    if (!cand_is_result) {  // (a level that ended on an empty round, or ran no round: its result, cut, from the slot)
      __syncthreads();
      for (int i = tid; i < n_res; i += NT) {
        const int32_t id = sv.cat_ids[i];
        cand[i] = id;
        sv.res_ids[i] = id;
      }
      __syncthreads();
    }
    // visited = idx_ep (:311), candidates = result.  The level's marks -- its starting result, <= top_k ids -- are NOT put into
    // `visited`: they stay a list (the slot's result array; a thread reads back the entries it wrote) and are taken out of
    // `seen` behind every walk, one atomic each.  So a level starts without a pass over the bitmaps, and its first round needs
    // no word of `visited`: it WRITES all of them (visited = new), which is also the level's visited = {}.
    const int n_marks = n_res;
    EVAL_TICK(2);
    int n_cand = n_res;
    const int32_t* __restrict__ values = a.nbv[level];
    const int64_t* __restrict__ rs = a.nbrs[level];
    for (int it = 0; it < a.num_scoring[level]; ++it) {
      // ---- neighbours of the candidates -> bits of `seen`.  Every thread fetches the bounds of its rows of the frontier (one
      // trip for all of them) into LDS; then wavefront w walks rows [8 (w + NW t), + 8) for t = 0, 1, ..., four trips at a time:
      // 32 rows per wavefront in flight together.  The level's marks ride along, for the removal behind the walk.
      ctr_f += n_cand;
      const bool first = it == 0;
      int32_t mk[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) mk[j] = sv.res_ids[min(tid + j * NT, n_marks - 1)];
      {
        long long s[NF], e[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = tid + j * NT;
          s[j] = 0; e[j] = 0;
          if (r < n_cand) {
            const int32_t c = cand[r];
            s[j] = rs[c]; e[j] = rs[c + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          const int r = tid + j * NT;
          if (r < n_cand) {
            bnd[r] = (unsigned long long)s[j] | ((unsigned long long)(e[j] - s[j]) << 48);
            ctr_g += (int)(e[j] - s[j]);
          }
        }
      }
      lds_barrier();
      int32_t fi[NF];
      float fs[NF];
      int n_next = 0, mode = 0;
      // one sweep of the id space per window: walk, take the marks out, the owners' pass, the scan, the emit
      for (int w = 0; w < W; ++w) {
        const uint32_t w_lo = (uint32_t)w * OW * 32u;            // first bitmap word of the window
        const uint32_t w_n = min(OW, DW - (uint32_t)w * OW) * 32u;  // its words
        const bool owner = (uint32_t)tid * 32u < w_n;
        auto seen_or = [&](uint32_t id) {
          const uint32_t lw = (id >> 5) - w_lo;  // (one window: every id is inside)
          if (!MULTI || lw < w_n) atomicOr(&seen[lw + (lw >> 5)], 1u << (id & 31));
        };
        for (int rep = 0; rep <= NANN_REPEAT_GATHER; ++rep) {
          int bad = 0;
          for (int t0 = 0; t0 * NW * 8 < n_cand; t0 += 4) {
            long long s4[4];
            int len4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int r = (t0 + t) * NW * 8 + wave * 8 + lane;
              unsigned long long b = 0ull;
              if (lane < 8 && r < n_cand) b = bnd[r];
              s4[t] = (long long)(b & 0xffffffffffffull); len4[t] = (int)(b >> 48);
            }
            int32_t v[4][8];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const long long sr = readlane64(s4[t], r);
                const int len = __builtin_amdgcn_readlane(len4[t], r);
                v[t][r] = lane < len ? values[sr + lane] : 0;
              }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int r = 0; r < 8; ++r) {
                const int len = __builtin_amdgcn_readlane(len4[t], r);
                if (lane < len) {
                  if ((uint32_t)v[t][r] < a.n_items) seen_or((uint32_t)v[t][r]); else bad = 1;
                }
                if (len > 64) {  // (rows of more than 64 neighbours)
                  const long long sr = readlane64(s4[t], r);
                  for (int j = 64 + lane; j < len; j += 64) {
                    const int32_t x = values[sr + j];
                    if ((uint32_t)x < a.n_items) seen_or((uint32_t)x); else bad = 1;
                  }
                }
              }
          }
          if (bad) SS->flags[1] = 1;
        }
        lds_barrier();  // (the walk loads; what is still on its way to the slot -- the last round's results, an owner's words
                        //  of `visited` -- is read back by the thread that stored it)
#pragma unroll
        for (int j = 0; j < NF; ++j)  // seen \= marks
          if (tid + j * NT < n_marks) {
            const uint32_t id = (uint32_t)mk[j];
            const uint32_t lw = (id >> 5) - w_lo;
            if (id >= a.n_items) SS->flags[1] = 1;
            else if (!MULTI || lw < w_n) atomicAnd(&seen[lw + (lw >> 5)], ~(1u << (id & 31)));
          }
        lds_barrier();
        if (SS->flags[1]) return NANN_ERR_INDEX_OUT_OF_RANGE;
        EVAL_TICK(3);
        // ---- new = seen & ~visited, in ascending id order (:316-319); visited |= new (:321), seen = 0
        // ALL 32 of the thread's words of the window's `visited` in ONE batch (touched words only, 4 / 8 / 16 at a time, were
        // 5 / 4 / 2 dependent trips; the level's first round has none to fetch) -- and with them this thread's ids and scores of
        // the kept results (the front of the concat arrays), which go to the staging area behind the emit.  Buffer loads: one
        // VGPR of offset for all of them, the word's stride in an SGPR.
        uint32_t cnt = 0, nd = 0;  // new bits of this thread's words; which of its words have any
        const int vbase = w * 32 * NT * 4;  // the window's words of `visited` (bytes)
        auto owners = [&](auto first_c) {
          constexpr bool FIRST = decltype(first_c)::value;
          uint32_t vis[32];
          if constexpr (!FIRST) {
#pragma unroll
            for (int j = 0; j < 32; ++j) vis[j] = __builtin_amdgcn_raw_buffer_load_b32(vrs, tid * 4, vbase + j * NT * 4, 0);
          }
          if (w == 0) {
#pragma unroll
            for (int j = 0; j < NF; ++j) {
              fi[j] = sv.cat_ids[min(tid + j * NT, n_res - 1)];
              fs[j] = sv.cat_sc[min(tid + j * NT, n_res - 1)];
            }
          }
          if (owner) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              uint32_t sw[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) sw[i] = seen[own + 8 * b + i];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int j = 8 * b + i;
                if constexpr (FIRST) {  // visited = new: every word written
                  __builtin_amdgcn_raw_buffer_store_b32(sw[i], vrs, tid * 4, vbase + j * NT * 4, 0);
                  if (sw[i]) { nd |= 1u << j; cnt += (uint32_t)__popc(sw[i]); }
                } else {
                  const uint32_t nw = sw[i] & ~vis[j];
                  if (sw[i] != nw) seen[own + j] = nw;  // the NEW bits stay, for the emit behind the scan
                  if (nw) {
                    __builtin_amdgcn_raw_buffer_store_b32(vis[j] | nw, vrs, tid * 4, vbase + j * NT * 4, 0);
                    nd |= 1u << j;
                    cnt += (uint32_t)__popc(nw);
                  }
                }
              }
            }
          }
        };
        if (first) owners(std::true_type{}); else owners(std::false_type{});
        EVAL_TICK(8);
        uint32_t total;
        uint32_t at = wg_excl_scan<NT, true>(cnt, SS, &total);  // thread order = word order = ascending ids
        ctr_s += (int)total;
        if (n_res + n_next + (int)total > a.cat_cap) return NANN_ERR_CAPACITY;
        EVAL_TICK(9);
        auto emit = [&](int32_t* dst) {  // this thread's words with new bits, four reads in flight; seen = 0 behind them
          uint32_t m = nd;
          while (m) {
            uint32_t j[4], x[4];
            int n = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (m) { j[u] = (uint32_t)(__ffs(m) - 1); m &= m - 1; n = u + 1; } else j[u] = j[0];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = seen[own + j[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (u < n) {
                seen[own + j[u]] = 0u;
                const uint32_t wd = w_lo + (uint32_t)tid * 32u + j[u];
                uint32_t y = x[u];
                while (y) {
                  dst[at++] = (int32_t)(wd * 32u + (uint32_t)(__ffs(y) - 1));
                  y &= y - 1;
                }
              }
          }
        };
        if constexpr (MULTI) {  // several windows: the ids to the slot, window after window (the region is `seen` again for the next one)
          if (cnt) emit(sv.cat_ids + n_res + n_next);
          n_next += (int)total;
          lds_barrier();
          continue;
        }
        n_next = (int)total;
        if (n_next == 0) break;
        // where the round's ids and scores live: 2 = both in the staging area, 1 = the scores, 0 = neither (the slot, as the slot form)
        mode = (n_next <= kEmitCap && n_res + n_next <= CAP2) ? 2 : n_res + n_next <= CAP ? 1 : 0;
        if (mode == 2) {
          if (cnt) emit(emit_lds);
          lds_barrier();  // `seen` is zero: the region is the staging area from here
          for (int i = tid; i < n_next; i += NT) st_cat_ids[n_res + i] = emit_lds[i];
#pragma unroll
          for (int j = 0; j < NF; ++j)
            if (tid + j * NT < n_res) { st_cat_ids[tid + j * NT] = fi[j]; st_cat_sc[tid + j * NT] = fs[j]; }
          lds_barrier();
        } else {
          if (cnt) emit(sv.cat_ids + n_res);
        }
      }
      if (n_next == 0) {  // plain TF ops score an empty batch as an empty tensor: the result is cut to min(k, n), no candidate is left
        __syncthreads();  // (`seen` is zero again: the owners wrote their words' new bits, none)
        n_res = min(a.top_k[level], n_res);
        n_cand = 0;
        cand_is_result = false;
        continue;
      }
      const int n_cat = n_res + n_next;
      if constexpr (MULTI) mode = n_cat <= CAP ? 1 : 0;
      if (mode != 2) {
        __syncthreads();  // the new ids are in the slot, `seen` is zero: the region is the staging area from here
        if (mode == 1) {
#pragma unroll
          for (int j = 0; j < NF; ++j)
            if (tid + j * NT < n_res) st_cat_sc[tid + j * NT] = fs[j];
        }
      }
      EVAL_TICK(4);
      const int k = min(a.top_k[level], n_cat);
      const bool last = it + 1 == a.num_scoring[level];  // the level's last round: its frontier is never walked
      auto rest = [&](auto mode_c) -> int {
        constexpr int MODE = decltype(mode_c)::value;
        const int32_t* cat_ids;
        float* cat_sc;
        if constexpr (MODE == 2) cat_ids = st_cat_ids; else cat_ids = sv.cat_ids;
        if constexpr (MODE >= 1) cat_sc = st_cat_sc; else cat_sc = sv.cat_sc;
        for (int rep = 0; rep <= NANN_REPEAT_SCORE; ++rep) {  // :323
          wg_score_l2_part<LPR, DT, NW>(a.emb, a.d, cat_ids + n_res, 0, n_next, qv, cat_sc + n_res, wave, near);
          __syncthreads();
        }
        EVAL_TICK(5);
        int rc = 0;
        // (the kept results are sorted and, once there are k of them, only candidates that beat the worst one can enter)
        const bool full = n_res >= k;
        const uint32_t floor_key = full ? score_key(cat_sc[n_res - 1]) : 0u;
        for (int rep = 0; rep <= NANN_REPEAT_TOPK; ++rep) {  // :326-328
          rc = wg_topk_binned<NT, kEvalMaxK>(cat_ids, cat_sc, n_cat, k, st_res_ids, st_res_sc, scratch, full ? n_res : 0x7fffffff, floor_key);
          if (rc) return rc;
        }
        EVAL_TICK(6);
        if (!last) {
          // next frontier: new nodes scoring at least the worst kept result, in id order (:330-331): every thread takes a
          // CONTIGUOUS run of the new nodes, one workgroup scan places them
          const float worst = st_res_sc[k - 1];
          const int per = (n_next + NT - 1) / NT;
          const int lo = min(tid * per, n_next), hi = min(lo + per, n_next);
          uint32_t mine = 0;
          for (int i0 = lo; i0 < hi; i0 += 8) {  // (eight scores in flight: they may be in the slot)
            float sc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sc[u] = cat_sc[n_res + min(i0 + u, hi - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) mine += (i0 + u < hi && sc[u] >= worst) ? 1u : 0u;
          }
          uint32_t n_new;
          uint32_t pos = wg_excl_scan<NT, true>(mine, SS, &n_new);
          if (n_new > (uint32_t)kEvalMaxK) return NANN_ERR_CAPACITY;  // more ties at the threshold than a frontier holds
          if (mine) {
            for (int i0 = lo; i0 < hi; i0 += 8) {  // the POSITIONS of the frontier's rows first ...
              float sc[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) sc[u] = cat_sc[n_res + min(i0 + u, hi - 1)];
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (i0 + u < hi && sc[u] >= worst) cand[pos++] = i0 + u;
            }
          }
          lds_barrier();
          for (int p = tid; p < (int)n_new; p += NT) cand[p] = cat_ids[n_res + cand[p]];  // ... then their ids, one trip for all
          n_cand = (int)n_new;
          __syncthreads();  // (the copy below overwrites cat_ids[n_res ..] that the selection above reads)
        } else {
          n_cand = k;
        }
        publish(k, MODE >= 1 ? n_cat : 0, MODE == 2 ? n_cat : 0, last);
        return 0;
      };
      st = mode == 2 ? rest(std::integral_constant<int, 2>{}) : mode == 1 ? rest(std::integral_constant<int, 1>{}) : rest(std::integral_constant<int, 0>{});
      if (st) return st;
      n_res = k;
      cand_is_result = last;
      EVAL_TICK(7);
    }
  }
  EVAL_TICK_FLUSH;
  *n_result = n_res;
  if (a.counters) {  // (the kernel zeroed the user's three words before the call)
    if (tid == 0) { atomicAdd(&a.counters[(size_t)qi * 3 + 0], ctr_f); atomicAdd(&a.counters[(size_t)qi * 3 + 2], ctr_s); }
    if (ctr_g) atomicAdd(&a.counters[(size_t)qi * 3 + 1], ctr_g);
  }
  return NANN_OK;
}

// LDS: [seen bitmap (SEEN_LDS) | scratch | q | misc].  SEEN_LDS: 0 = the slot form, 1 = the LDS form, one window (shards of up to
// ~1 M items), 2 = the LDS form sweeping the id space in windows
template <int LPR, int DT, int SC, int NT, int SEEN_LDS>
__global__ __launch_bounds__(NT) void k_search_eval(EvalArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kScratchBytes = eval_scratch_bytes<SC, NT>();
  const size_t bm_bytes = SEEN_LDS ? (((size_t)a.lds_words * 4 + 255) & ~(size_t)255) : 0;
  unsigned char* scratch = smem + bm_bytes;
  float* qv = reinterpret_cast<float*>(scratch + kScratchBytes);
  int* misc = reinterpret_cast<int*>(qv + kMaxD);

  unsigned long long off[7];
  eval_slot_layout(a.bm_words, a.vis_words, a.cat_cap, off);
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
  bool clear_seen = true;  // (uniform)
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) misc[0] = (int)atomicAdd(&hdr->queue, 1u);
    __syncthreads();
    const int qi = misc[0];
    if (qi >= a.n_queries) break;
    if (a.counters && threadIdx.x < 3) a.counters[(size_t)qi * 3 + threadIdx.x] = 0;
    int n_res = 0;
    int st;
    // `seen` is all-zero behind a user that succeeded (the emit pass clears what the round touched); the slot's first user
    // and the one behind a failure clear it whole.  (Without the dirty words the owners clear as they scan: same invariant.)
    if constexpr (SEEN_LDS) {
      static_assert(SC == NANN_SCORER_L2, "the LDS form is the L2 scorer's");
      if constexpr (SEEN_LDS == 2) st = search_eval_win<LPR, DT, NT>(a, qi, sv, reinterpret_cast<uint32_t*>(smem), scratch, qv, &n_res, clear_seen);
      else st = search_eval_lds<LPR, DT, NT>(a, qi, sv, reinterpret_cast<uint32_t*>(smem), scratch, qv, &n_res, clear_seen);
    } else {
      st = search_eval_slot<LPR, DT, SC, NT>(a, qi, sv, sv.seen, scratch, qv, &n_res, clear_seen);
    }
    clear_seen = st != 0;
    __syncthreads();
    // the kept results: the LDS form leaves them at the front of the concat arrays (its result arrays are in LDS)
    const int32_t* res_ids = SEEN_LDS ? sv.cat_ids : sv.res_ids;
    const float* res_sc = SEEN_LDS ? sv.cat_sc : sv.res_sc;
    const int n = st ? 0 : min(K, n_res);  // results[:topk_eval] (:358), item ids (:360)
    for (int i = threadIdx.x; i < K; i += NT) {
      const int32_t r = i < n ? res_ids[i] : 0;
      a.out_ids[(size_t)qi * K + i] = i < n ? a.item_ids[r] : 0;
      if (a.out_scores) a.out_scores[(size_t)qi * K + i] = i < n ? res_sc[i] : 0.0f;
      if (a.out_index) a.out_index[(size_t)qi * K + i] = r;
    }
    if (threadIdx.x == 0) {
      a.status[qi] = st;
      a.n_out[qi] = n;
    }
  }
}

// LDS bytes of an instance: the scratch + q + misc, + the bitmap when `seen` lives there
template <int SC, int NT>
constexpr size_t eval_lds_base() { return (size_t)eval_scratch_bytes<SC, NT>() + kMaxD * 4 + 256; }

template <int LPR, int DT, int SC, int NT, int SEEN_LDS>
inline int launch_eval_as(int slots, const EvalArgs& a, hipStream_t st) {
  auto kern = k_search_eval<LPR, DT, SC, NT, SEEN_LDS>;
  const size_t lds_bytes = eval_lds_base<SC, NT>() + (SEEN_LDS ? (((size_t)a.lds_words * 4 + 255) & ~(size_t)255) : 0);
  if (lds_bytes > 48 * 1024)
    NANN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(slots), dim3(NT), lds_bytes, st, a);
  NANN_HIP_TRY(hipGetLastError());
  return NANN_OK;
}

// instantiations: nann_eval_inst.hip (L2 slot form, attention model), nann_eval_lds_inst.hip (L2, LDS form), nann_mlp_inst.hip (MLP, f32 MFMA).  seen_lds: the L2
// instances only (eval_l2_lds_bytes() = what the plan checks against the CU's LDS)
size_t eval_l2_lds_base();
size_t eval_dirty_room();  // bytes of the phase scratch the second-level bitmap may take (the smallest instance's)
int launch_eval_l2(int lpr, int dt, int seen_lds, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_l2_lds(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st);  // nann_eval_lds_inst.hip (one window)
int launch_eval_l2_win(int lpr, int dt, int slots, const EvalArgs& a, hipStream_t st);  // nann_eval_win_inst.hip (several)
int launch_eval_attn(int d, int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d64(int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d128(int dt, int slots, const EvalArgs& a, hipStream_t st);
int launch_eval_mlp_d256(int dt, int slots, const EvalArgs& a, hipStream_t st);

}  // namespace nann
