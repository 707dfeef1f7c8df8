// tools/rejected/nann_coop_small_batches.h -- NOT PART OF THE BUILD.  Round 4's experiment for VERDICT r3 item 8 ("a path for
// few concurrent requests"): several workgroups per query at small batches.  Kept as it ran (it was spliced into
// csrc/nann_search.h: the building blocks below, a branch in search_one's L2 scoring call, a helper loop ahead of
// k_search's query loop, block -> (query, participant) mapping, `helpers` / `grid` in the plan).  Results were bit-identical
// to the one-workgroup launch (25 parity tests) and SLOWER at every batch size: profiles/r4ij_coop_small_batches_ab.txt.
//
// search_one, L2 scorer on the 32K-slot plan:
//     if (a.helpers > 1) coop_owner_score<LPR, DT, NT>(job, r + 1, a.emb, a.d, sc_ids, sc_n, base_off, r == 0, qv, sc_out, box);
// k_search, before the query loop (block b = participant (b / 8) % helpers of query (b % 8) + 8 (b / (8 helpers))):
//     if (coop_h > 0) for (;;) { poll job->seq (relaxed) ; kCoopExit -> return ; lane-0 acquire ; barrier ;
//                                coop_score_chunks(job, seq, ..., from_enter ? a.enter : cand_ids + off, n, qv, cand_scores + off) ; }
//     owner, after search_one: job->seq = kCoopExit
#pragma once

// ---- several workgroups per query when the chip is mostly idle (round 4) -----------------------------------------
// One query = one workgroup leaves 255 CUs idle at B = 1 and three quarters of the chip at B = 64, and 42 % of such a
// query's time is the five scoring calls: ~2 k random 256-byte rows each through ONE CU's memory pipe (~100 GB/s).
// With `helpers` workgroups per query the OWNER (participant 0) runs the traversal as always; at a scoring call it
// publishes the candidate list it already holds in its slot's scratch (ids in, scores out: the arrays every stage uses
// anyway), and owner and helpers claim 512-row chunks of it until none is left.  Rows are scored by the same code
// whoever claims them, so results are bit-identical to the one-workgroup launch.
//   * claims are a CAS on (stage << 16 | next chunk): a helper that is late for a stage finds another stage number and
//     claims nothing; the owner resets the word FIRST when it opens the next stage;
//   * the owner only ever waits for chunks that a RUNNING helper has claimed (done == chunks): a helper that was never
//     scheduled claims nothing, so nothing can deadlock, whatever else occupies the GPU;
//   * helpers of a query sit on the same XCD as their owner (observed placement: block b -> XCD b % 8), so the flag
//     hops stay inside one L2; correctness does not depend on it (agent-scope release / acquire).
struct CoopJob {
  unsigned int seq;    // stage number, > 0; kCoopExit: the query is done
  unsigned int n;      // rows of the stage
  unsigned int off;    // ids = slot cand_ids + off (or the enter points when from_enter), scores = slot cand_scores + off
  unsigned int from_enter;
  unsigned int next;   // stage << 16 | next unclaimed chunk
  unsigned int done;   // chunks finished
  unsigned int pad[10];
};
constexpr unsigned int kCoopExit = 0xffffffffu;
constexpr int kCoopChunk = 512;  // rows: one full iteration of wg_score_l2_part at 1024 threads, d = 128 (8 row loads per lane in flight)

// Hand-offs follow cdna_hip_programming.md Guideline 16: plain payload stores -> barrier -> ONE lane's agent release
// -> drained -> relaxed flag store; the consumer polls RELAXED (an acquire load per poll drops its CU's L1 every time:
// 2-3x slower hops, and hundreds of such pollers cost the chip half its bandwidth -- the first version of this code,
// profiles/r4i_coop_ab.txt), then ONE lane's agent acquire -> barrier -> plain loads.
__device__ __forceinline__ unsigned int coop_ld(const unsigned int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_release_lane0() {  // caller: behind a barrier, lane 0 of the workgroup only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (ROCm 7.2 can drop the wait behind buffer_wbl2: restate it)
}

// owner and helpers: claim chunks of the open stage and score them.  All NT threads; `box` = one LDS int.
template <int LPR, int DT, int NT>
__device__ __forceinline__ void coop_score_chunks(CoopJob* job, unsigned int stage, const void* emb, int d, const int32_t* ids,
                                                  int n, const float* qv, float* out, int* box) {
  const int tid = local_tid();
  const int n_chunks = (n + kCoopChunk - 1) / kCoopChunk;
  int mine = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      int got = -1;
      unsigned int cur = __hip_atomic_load(&job->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while ((cur >> 16) == (stage & 0xffffu) && (int)(cur & 0xffffu) < n_chunks) {
        if (__hip_atomic_compare_exchange_strong(&job->next, &cur, cur + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT)) { got = (int)(cur & 0xffffu); break; }
      }
      *box = got;
    }
    __syncthreads();
    const int c = *box;
    if (c < 0) break;
    const int lo = c * kCoopChunk, hi = min(n, lo + kCoopChunk);
    wg_score_l2_part<LPR, DT, NT / 64>(emb, d, ids, lo, hi, qv, out, tid >> 6);
    ++mine;
  }
  if (mine) {  // the scores are written back before the count says so
    __syncthreads();
    if (tid == 0) {
      coop_release_lane0();
      __hip_atomic_fetch_add(&job->done, (unsigned int)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// owner: open a stage, work on it, wait for the chunks the helpers claimed
template <int LPR, int DT, int NT>
__device__ __forceinline__ void coop_owner_score(CoopJob* job, unsigned int stage, const void* emb, int d, const int32_t* ids,
                                                 int n, unsigned int off, bool from_enter, const float* qv, float* out, int* box) {
  const int tid = local_tid();
  const int n_chunks = (n + kCoopChunk - 1) / kCoopChunk;
  __syncthreads();
  if (tid == 0) {
    if (!from_enter) coop_release_lane0();  // the candidate ids this workgroup wrote (the enter points are the index's)
    __hip_atomic_store(&job->next, (stage & 0xffffu) << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // first: closes the old stage for late helpers
    __hip_atomic_store(&job->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&job->n, (unsigned int)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&job->off, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&job->from_enter, from_enter ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the job's fields before its number
    __hip_atomic_store(&job->seq, stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  coop_score_chunks<LPR, DT, NT>(job, stage, emb, d, ids, n, qv, out, box);
  __syncthreads();
  if (tid == 0) {
    while ((int)coop_ld(&job->done) < n_chunks) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the helpers' scores: this CU's L1 is dropped once
  }
  __syncthreads();
}


