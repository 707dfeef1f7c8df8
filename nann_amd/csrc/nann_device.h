// nann_device.h -- workgroup-level building blocks of the gfx950 retrieval path.
//
// Written for CDNA4 only: 64-lane wavefronts (ballot masks are 64-bit, lane
// arithmetic is & 63), 160 KiB LDS per CU, one 1024-thread workgroup per CU.
// Each block below states which reference loop it replaces (paths relative to
// /root/reference/, UO/ = tensorflow/tensorflow/core/user_ops/).
//
//   wave_walk_span  BitmapRefDifference::Differ  UO/bitmap_op/bitmap_ops.cc:221-234 (one wavefront)
//   wg_filter_chunk the same scan by the whole workgroup, 2048 ids per step
//   wg_expand_walk  GroupGather::Fill + Differ   UO/beam_search_op/GroupGather_kernel.cc:137-168
//   wg_score     GatherV2 + scorer             core/kernels/gather_functor.h:96-103 + BlazeXlaOp
//   wg_topk      TopKV2 (+ Gather of ids)      core/kernels/topk_op.cc:104-205
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace nann {

constexpr int kNT = 1024;            // threads per traversal workgroup (bitmap kernels)
constexpr int kNW = kNT / 64;        // 16 wavefronts
constexpr int kTopkEPT = 16;         // top-k keys held in registers per thread (n <= 16384)
constexpr int kMaxK = 1024;          // largest k / frontier a workgroup handles
constexpr int kPhaseScratch = 27648; // LDS bytes shared by the phases of the bitmap kernels
constexpr int kMaxD = 512;
// bitmap kernels: candidate scores of a round are mirrored in LDS behind the top-k scratch
constexpr int kLdsScoresOff = 10752;
constexpr int kLdsScores = 4096;

enum : int { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

// threadIdx.x behind an opaque asm.  Every building block of the fused kernel starts from
// its own copy: what it derives from the thread id (lane masks, LDS addresses, ~tid, ...) is
// then recomputed inside the block instead of being hoisted out of the round / query loops
// and kept live -- and spilled -- across all the other blocks.
__device__ __forceinline__ int local_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < 1024);
  return t;
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }  // per-op kernels
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }

// ---------------------------------------------------------------------------
// cross-lane primitives on DPP (a VALU modifier, a few cycles) instead of ds_bpermute
// (an LDS round trip of ~100 cycles per step): row_shr within rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 to carry across rows (gfx9 family, incl. gfx950).
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t src) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}
// inclusive prefix sum over the 64 lanes (lane 63 = total)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += dpp_u32<0x111>(0, v);
  v += dpp_u32<0x112>(0, v);
  v += dpp_u32<0x114>(0, v);
  v += dpp_u32<0x118>(0, v);
  v += dpp_u32<0x142, 0xA>(0, v);
  v += dpp_u32<0x143, 0xC>(0, v);
  return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t scanned) {
  return (uint32_t)__builtin_amdgcn_readlane((int)scanned, 63);
}
// OR / AND of a value over the wavefront (result valid in lane 63, broadcast by readlane)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
  v |= dpp_u32<0x111>(0, v);
  v |= dpp_u32<0x112>(0, v);
  v |= dpp_u32<0x114>(0, v);
  v |= dpp_u32<0x118>(0, v);
  v |= dpp_u32<0x142, 0xA>(0, v);
  v |= dpp_u32<0x143, 0xC>(0, v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_and(uint32_t v) {
  v &= dpp_u32<0x111>(0xffffffffu, v);
  v &= dpp_u32<0x112>(0xffffffffu, v);
  v &= dpp_u32<0x114>(0xffffffffu, v);
  v &= dpp_u32<0x118>(0xffffffffu, v);
  v &= dpp_u32<0x142, 0xA>(0xffffffffu, v);
  v &= dpp_u32<0x143, 0xC>(0xffffffffu, v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
  v = min(v, dpp_u32<0x111>(0xffffffffu, v));
  v = min(v, dpp_u32<0x112>(0xffffffffu, v));
  v = min(v, dpp_u32<0x114>(0xffffffffu, v));
  v = min(v, dpp_u32<0x118>(0xffffffffu, v));
  v = min(v, dpp_u32<0x142, 0xA>(0xffffffffu, v));
  v = min(v, dpp_u32<0x143, 0xC>(0xffffffffu, v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
  v = max(v, dpp_u32<0x111>(0, v));
  v = max(v, dpp_u32<0x112>(0, v));
  v = max(v, dpp_u32<0x114>(0, v));
  v = max(v, dpp_u32<0x118>(0, v));
  v = max(v, dpp_u32<0x142, 0xA>(0, v));
  v = max(v, dpp_u32<0x143, 0xC>(0, v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int CTRL>
__device__ __forceinline__ int32_t dpp_i32(int32_t v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// ---------------------------------------------------------------------------
// optional per-phase time attribution (thread 0 only; off unless a buffer is given)
enum { PH_ZERO = 0, PH_WALK, PH_EXPAND, PH_SCORE, PH_TOPK, PH_OTHER,
       PH_TK_LOAD, PH_TK_SEARCH, PH_TK_COLLECT, PH_TK_SORT, PH_EX_PASS1, PH_EX_LOOP, PH_EX_WALKBUSY,
       PH_EX_LOOKUP, PH_EX_LOAD, PH_EX_INSERT, PH_EX_BARA, PH_EX_CHECK, PH_EX_RANK,  // hash-set expand, per piece
       PH_COUNT };
struct PhaseTimer {
  long long* ticks;  // LDS, [PH_COUNT]
  long long last;
  bool on;
  __device__ __forceinline__ void start(long long* t, bool enable) {
    ticks = t; on = enable; last = enable ? (long long)clock64() : 0;
  }
  // attribute the time since the previous mark to `phase`; sub-phases (>= PH_TK_LOAD)
  // do not reset the clock of the enclosing phase
  __device__ __forceinline__ void mark(int phase) {
    if (on && threadIdx.x == 0) {
      const long long now = (long long)clock64();
      ticks[phase] += now - last;
      last = now;
    }
  }
  __device__ __forceinline__ void sub(int phase, long long& t0) {
    if (on && threadIdx.x == 0) {
      const long long now = (long long)clock64();
      ticks[phase] += now - t0;
      t0 = now;
    }
  }
  __device__ __forceinline__ long long now() const { return on ? (long long)clock64() : 0; }
};
// The view of the timer the building blocks take BY VALUE (a pointer to the PhaseTimer object,
// nullable, kept it in scratch memory: every mark cost a scratch load even with timing off).
struct SubTimer {
  long long* ticks;
  bool on;
  __device__ __forceinline__ long long now() const { return on ? (long long)clock64() : 0; }
  __device__ __forceinline__ void sub(int phase, long long& t0) const {
    if (on && threadIdx.x == 0) {
      const long long t = (long long)clock64();
      ticks[phase] += t - t0;
      t0 = t;
    }
  }
};
__device__ __forceinline__ SubTimer no_timer() { return SubTimer{nullptr, false}; }

// ---------------------------------------------------------------------------
// scalar conversions (exact)
__device__ __forceinline__ float half_bits_to_float(uint32_t h) {
  return __half2float(__ushort_as_half((unsigned short)h));
}
__device__ __forceinline__ float bf16_bits_to_float(uint32_t h) { return __uint_as_float(h << 16); }

// monotone map f32 -> u32 (larger score -> larger key).  `+ 0.0f` folds -0 into
// +0 so that equal floats have equal keys, as TopKV2's comparator sees them
// (topk_op.cc:134-142).  NaNs are outside the contract.
__device__ __forceinline__ uint32_t score_key(float s) {
  s = s + 0.0f;
  const uint32_t u = __float_as_uint(s);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------
// LDS / global fill
__device__ __forceinline__ void wg_zero_words(uint32_t* p, uint32_t n_words) {
  // n_words is padded to a multiple of 4 by the caller (16-byte stores)
  uint4* p4 = reinterpret_cast<uint4*>(p);
  const uint32_t n4 = n_words >> 2;
  for (uint32_t i = (uint32_t)local_tid(); i < n4; i += blockDim.x) p4[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// wave_walk_span: ordered first-occurrence filter against the visited bitmap.
//
// Reference semantics (bitmap_ops.cc:224-232): scan ids in order; keep an id
// iff its bit is clear, then set the bit.  Here ONE wavefront walks a span 64
// ids per step, U steps per batch.  Per step every lane reads its bitmap word
// (pre) and ORs its bit in (old).  All 2U LDS operations of a batch are issued
// back to back: the LDS executes one wave's instructions in order, so step
// c+1 observes every bit step c set -- the serial scan order is preserved
// across steps while the atomics stay pipelined, and only the ballots below
// wait for them.  Inside a step, lanes holding the same fresh id are resolved
// to the LOWEST lane with ballots (which lane the LDS arbiter served first
// does not matter).  Kept ids are compacted with ballot + popcount (stable),
// so the output order is the serial order.
//
// kLdsBm=false walks a bitmap in global memory (shards too large for LDS):
// the pre-read is then an atomic OR of 0 so that it is served by L2 like the
// update, one step at a time.
//
// Must be called by all lanes of exactly one wavefront.  `src` may be LDS or
// global.  Returns base + number of ids kept (wave-uniform).  *err is set to 1
// if an id is outside [0, n_items).
template <bool kLdsBm>
__device__ __forceinline__ int wave_walk_span(const int32_t* src, int n, uint32_t* bm,
                                              uint32_t n_items, int32_t* out, int base, int* err) {
  constexpr int U = kLdsBm ? 8 : 1;  // steps per batch
  if (n <= 0) return base;
  const int lane = local_tid() & 63;
  const uint64_t lt = lanemask_lt(lane);
  bool bad = false;
  for (int c0 = 0; c0 < n; c0 += 64 * U) {
    int32_t x[U];
    uint32_t pre[U], old[U];
    // branch-free: lanes past the span / out of range read word 0 and OR in nothing
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = src[min(c0 + u * 64 + lane, n - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool valid = (c0 + u * 64 + lane) < n;
      const bool inr = valid && (uint32_t)x[u] < n_items;
      bad |= valid && !inr;
      uint32_t* w = bm + (inr ? ((uint32_t)x[u] >> 5) : 0u);
      const uint32_t bit = inr ? (1u << (x[u] & 31)) : 0u;
      pre[u] = kLdsBm ? *w : atomicOr(w, 0u);
      old[u] = atomicOr(w, bit);
    }
    bool fresh[U], keep[U];
    uint64_t dm[U];
    uint64_t any_dup = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool inr = (c0 + u * 64 + lane) < n && (uint32_t)x[u] < n_items;
      const uint32_t bit = 1u << (x[u] & 31);
      fresh[u] = inr && !(pre[u] & bit);
      keep[u] = inr && !(old[u] & bit);
      dm[u] = __ballot(fresh[u] && !keep[u]);
      any_dup |= dm[u];
    }
    if (any_dup) {  // some step holds the same fresh id twice: the lowest lane keeps it
#pragma unroll
      for (int u = 0; u < U; ++u) {
        uint64_t dupl = dm[u];
        while (dupl) {
          const int l = __ffsll((unsigned long long)dupl) - 1;     // wave-uniform (scalar)
          const int32_t xv = __builtin_amdgcn_readlane(x[u], l);   // v_readlane: no LDS round trip
          const bool mine = fresh[u] && x[u] == xv;
          const uint64_t same = __ballot(mine);
          const int first = __ffsll((unsigned long long)same) - 1;
          if (mine) keep[u] = (lane == first);
          dupl &= ~same;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {  // stable compaction
      const uint64_t m = __ballot(keep[u]);
      if (keep[u]) out[base + popc64(m & lt)] = x[u];
      base += popc64(m);
    }
  }
  if (__ballot(bad) != 0ull) {
    if (lane == 0) *err = 1;
  }
  return base;
}

// ---------------------------------------------------------------------------
// wg_expand_walk: GroupGather (one group) fused with BitmapRefDifference.
//
//   GroupGather_kernel.cc:137-168 -- concatenate the CSR rows of a frontier in
//   frontier order, duplicates kept (build_opt_graph.py:39-49);
//   bitmap_ops.cc:221-234        -- ordered visited-set filter of that list.
//
// The concatenation is never materialised in HBM.  Pass 1 (count, :137-145):
// one thread per frontier node reads its two row_splits; a block-wide
// exclusive scan gives every row its offset in the virtual list.  Pass 2 walks
// the list in pieces of kChunk ids: the rows of piece c+1 are fetched from the
// CSR into registers (one wavefront per row, one coalesced <=256-byte load per
// row) while piece c, already staged in LDS, goes through wg_filter_chunk.
//
// wg_filter_chunk is the reference's serial scan (keep an id iff its bit is
// clear, then set it) done by the whole workgroup, kChunk ids at a time:
//   1. every thread reads the bitmap words of its ids       -> visited before the piece
//   2. barrier; the unvisited ones OR their bit in           -> exactly one "winner" per id
//   3. the losers of step 2 are later/earlier copies of an id that is new in this piece:
//      they record min(position) per id in a small LDS hash table (slot -> position, ids
//      compared through the staged list); winners that find their id there join the min
//   4. an id is kept at the smallest position it occurs at -- the serial scan's choice
//   5. kept ids are compacted in position order (threads own consecutive positions:
//      DPP wave scan + one LDS exchange of wave totals), so the output is the serial order.
// Five barriers per kChunk ids instead of kChunk/64 dependent steps of one wavefront.
//
// List mode (row_splits == nullptr): `values[0..n_frontier)` is itself the
// list to filter (the "mark" calls, build_opt_graph.py:119-120,132-133).
//
// All NT threads.  n_frontier <= kMaxK in CSR mode.  Outputs (uniform):
// *gathered = length of the virtual list, return value = ids kept (appended
// to out[0..)); -1 on an out-of-range frontier id or neighbour id.
constexpr int kChunk = 2048;
constexpr uint32_t kHashEmpty = 0xffffffffu;
constexpr int kChunkTab = 64;
constexpr int kHashSlots = kChunk;
struct ExpandWalkScratch {
  uint32_t off[kMaxK + 1];
  uint32_t rowstart[kMaxK];
  uint32_t wave_tot[kNW];
  int bad;
  int any_contested;
  int pad[2];
  int chunk_first[kChunkTab];  // first row of piece c (c < kChunkTab), filled by pass 1
  int32_t stage[kChunk];   // the piece being filtered
  uint32_t hash[kHashSlots];  // slot -> smallest position (in the piece) of the id hashed there
};

__device__ __forceinline__ uint32_t chunk_hash(int32_t x) {
  return ((uint32_t)x * 2654435761u) >> 21;  // 11 bits: kChunk slots
}
static_assert(kChunk == 2048, "chunk_hash yields log2(kChunk) bits");


template <bool kLdsBm, int NT>
__device__ __forceinline__ int wg_filter_chunk(ExpandWalkScratch* S, int n_c, uint32_t* bm,
                                               uint32_t n_items, int32_t* out, int base) {
  constexpr int PER = kChunk / NT;  // consecutive positions per thread
  constexpr int NWV = NT / 64;
  static_assert(PER == 2 || PER == 4, "NT must be 1024 or 512");
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  int32_t x[PER];
  if constexpr (PER == 2) {
    const int2 v = reinterpret_cast<const int2*>(S->stage)[tid];
    x[0] = v.x; x[1] = v.y;
  } else {
    const int4 v = reinterpret_cast<const int4*>(S->stage)[tid];
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  }
  // ---- 1. visited before this piece?
  uint32_t* w[PER];
  uint32_t bit[PER];
  bool fresh[PER];
  bool bad = false;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const bool valid = tid * PER + e < n_c;
    const bool inr = valid && (uint32_t)x[e] < n_items;
    bad |= valid && !inr;
    w[e] = bm + (inr ? ((uint32_t)x[e] >> 5) : 0u);
    bit[e] = inr ? (1u << (x[e] & 31)) : 0u;
    const uint32_t pre = kLdsBm ? *w[e] : atomicOr(w[e], 0u);  // global bitmap: served by L2 like the update
    fresh[e] = inr && !(pre & bit[e]);
  }
#pragma unroll
  for (int e = 1; e < PER; ++e)  // the same new id twice inside one thread: the later one is a copy
#pragma unroll
    for (int e2 = 0; e2 < e; ++e2)
      if (fresh[e] && fresh[e2] && x[e] == x[e2]) fresh[e] = false;
  if (bad) S->bad = 1;
  __syncthreads();
  // ---- 2. set the bits; one winner per new id
  bool won[PER], cont[PER], keep[PER];
  uint32_t slot[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    won[e] = false;
    if (fresh[e]) won[e] = !(atomicOr(w[e], bit[e]) & bit[e]);
    cont[e] = fresh[e] && !won[e];
    slot[e] = kHashEmpty;
  }
  // ---- 3. the other copies of a new id record the smallest position
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    if (cont[e]) {
      const uint32_t pos = (uint32_t)(tid * PER + e);
      S->any_contested = 1;
      uint32_t h = chunk_hash(x[e]);
      for (;;) {
        uint32_t cur = S->hash[h];
        if (cur == kHashEmpty) {
          cur = atomicCAS(&S->hash[h], kHashEmpty, pos);
          if (cur == kHashEmpty) break;
        }
        if (S->stage[cur] == x[e]) { atomicMin(&S->hash[h], pos); break; }
        h = (h + 1) & (kChunk - 1);
      }
      slot[e] = h;
    }
  }
  __syncthreads();
  const bool anyc = S->any_contested != 0;
  if (anyc) {
    // ---- 4. winners whose id has other copies join the minimum
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      if (won[e]) {
        const uint32_t pos = (uint32_t)(tid * PER + e);
        uint32_t h = chunk_hash(x[e]);
        for (;;) {
          const uint32_t cur = S->hash[h];
          if (cur == kHashEmpty) break;  // no other copy
          if (S->stage[cur] == x[e]) { atomicMin(&S->hash[h], pos); slot[e] = h; break; }
          h = (h + 1) & (kChunk - 1);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const uint32_t pos = (uint32_t)(tid * PER + e);
      keep[e] = (won[e] && slot[e] == kHashEmpty) || ((won[e] || cont[e]) && slot[e] != kHashEmpty && S->hash[slot[e]] == pos);
    }
  } else {
#pragma unroll
    for (int e = 0; e < PER; ++e) keep[e] = won[e];
  }
  // ---- 5. ordered compaction
  uint32_t cnt = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) cnt += keep[e] ? 1u : 0u;
  const uint32_t inc = wave_scan_add(cnt);
  if (lane == 63) S->wave_tot[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0, tot = 0;
#pragma unroll
  for (int wv = 0; wv < NWV; ++wv) {
    const uint32_t v = S->wave_tot[wv];
    if (wv < wave) wbase += v;
    tot += v;
  }
  int o = base + (int)(wbase + inc - cnt);
#pragma unroll
  for (int e = 0; e < PER; ++e)
    if (keep[e]) out[o++] = x[e];
  if (anyc) {  // leave the table empty for the next piece (read again only after two barriers)
#pragma unroll
    for (int e = 0; e < PER; ++e)
      if (cont[e]) S->hash[slot[e]] = kHashEmpty;
    if (tid == 0) S->any_contested = 0;
  }
  return base + (int)tot;
}


template <bool kLdsBm, int NT = kNT>
__device__ __forceinline__ int wg_expand_walk(const int32_t* frontier, int n_frontier,
                                              const int32_t* __restrict__ values,
                                              const int64_t* __restrict__ row_splits,
                                              uint32_t n_items, uint32_t* bm, int32_t* out,
                                              unsigned char* scratch, int* gathered,
                                              SubTimer pt = no_timer()) {
  ExpandWalkScratch* S = reinterpret_cast<ExpandWalkScratch*>(scratch);
  long long tsub = pt.now();
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const bool list_mode = row_splits == nullptr;
  const int n_rows = list_mode ? 1 : n_frontier;
  if (tid == 0) { S->bad = 0; S->any_contested = 0; }
  for (int i = tid; i < kHashSlots; i += NT) S->hash[i] = kHashEmpty;
  __syncthreads();
  // ---- pass 1: row lengths -> offsets --------------------------------------
  constexpr int NWV = NT / 64;
  uint32_t total = 0;
  for (int t0 = 0; t0 < n_rows; t0 += NT) {
    const int t = t0 + tid;
    uint32_t len = 0, start = 0;
    if (list_mode) {
      if (t == 0) len = (uint32_t)n_frontier;
    } else if (t < n_frontier) {
      const int32_t node = frontier[t];
      if ((uint32_t)node < n_items) {
        const int64_t s = row_splits[node], e = row_splits[node + 1];
        start = (uint32_t)s;
        len = (uint32_t)(e - s);
      } else {
        S->bad = 1;
      }
    }
    const uint32_t inc = wave_scan_add(len);
    if (lane == 63) S->wave_tot[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const uint32_t v = S->wave_tot[w];
      if (w < wave) wbase += v;
      tot += v;
    }
    if (t < n_rows) {
      const uint32_t my_off = total + wbase + inc - len;
      S->off[t] = my_off;
      S->rowstart[t] = start;
      // pieces that begin inside this row
      for (uint32_t c = (my_off + kChunk - 1) / kChunk; c < (uint32_t)kChunkTab && c * kChunk < my_off + len; ++c)
        S->chunk_first[c] = t;
    }
    total += tot;
    __syncthreads();
  }
  if (tid == 0) S->off[n_rows] = total;
  __syncthreads();
  *gathered = (int)total;
  pt.sub(PH_EX_PASS1, tsub);
  if (S->bad) return -1;
  // ---- pass 2: fetch piece c+1 || filter piece c -------------------------------
  const int G = (int)total;
  const int n_chunks = (G + kChunk - 1) / kChunk;
  constexpr int RB = 64 / NWV;  // rows per wavefront in the register-staged batch: 64 rows per piece
  struct RowBatch {
    int dst_off[RB];  // index into the staging buffer, -1 = nothing
    int32_t v[RB];
    uint32_t long_rows;
    int r0;
  };
  // rows r0, r0 + NWV, ... of piece c: one coalesced load per row into registers
  auto issue = [&](int c, int r0, RowBatch& B) {
    const uint32_t lo = (uint32_t)c * kChunk;
    const uint32_t hi = min((uint32_t)G, lo + kChunk);
    uint32_t src_off[RB];  // index into values[] (0 when the lane has nothing to copy)
    B.long_rows = 0;
    B.r0 = r0;
#pragma unroll
    for (int j = 0; j < RB; ++j) {  // row descriptors (LDS)
      const int r = r0 + j * NWV;
      src_off[j] = 0; B.dst_off[j] = -1;
      if (r < n_rows) {
        const uint32_t o = S->off[r];
        const uint32_t l = S->off[r + 1] - o;
        const uint32_t p = o + lane;
        if (lane < l && p >= lo && p < hi) {
          src_off[j] = S->rowstart[r] + lane;
          B.dst_off[j] = (int)(p - lo);
        }
        if (l > 64 && o < hi) B.long_rows |= 1u << j;
      }
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) B.v[j] = values[src_off[j]];
  };
  auto commit = [&](int c, const RowBatch& B) {
    const uint32_t lo = (uint32_t)c * kChunk;
    const uint32_t hi = min((uint32_t)G, lo + kChunk);
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if (B.dst_off[j] >= 0) S->stage[B.dst_off[j]] = B.v[j];
    if (B.long_rows) {  // rows longer than 64 (not produced by HNSW with M <= 32): remaining pieces
      for (int j = 0; j < RB; ++j) {
        if (!((B.long_rows >> j) & 1u)) continue;
        const int r = B.r0 + j * NWV;
        const uint32_t o = S->off[r], l = S->off[r + 1] - o, st = S->rowstart[r];
        for (uint32_t cc = 64 + lane; cc < l; cc += 64) {
          const uint32_t pp = o + cc;
          if (pp >= lo && pp < hi) S->stage[pp - lo] = values[(size_t)st + cc];
        }
      }
    }
  };
  // first row whose end lies beyond the start of piece c: upper_bound over off[1..n_rows]
  auto first_row = [&](int c) {
    if (c < kChunkTab) return S->chunk_first[c];
    const uint32_t lo = (uint32_t)c * kChunk;
    int a = 0, b = n_rows;
    while (a < b) {
      const int m = (a + b) >> 1;
      if (S->off[m + 1] > lo) b = m; else a = m + 1;
    }
    return a;
  };
  // rows of piece c beyond the register-staged batch (pieces made of many short rows)
  auto rest = [&](int c, int r_first) {
    const uint32_t hi = min((uint32_t)G, (uint32_t)c * kChunk + kChunk);
    for (int r0 = r_first + NWV * RB; r0 < n_rows && S->off[r0] < hi; r0 += NWV * RB) {
      RowBatch B2;
      issue(c, r0, B2);
      commit(c, B2);
    }
  };
  RowBatch B;
  if (n_chunks > 0) {
    issue(0, first_row(0) + wave, B);
    commit(0, B);
    rest(0, B.r0);
  }
  __syncthreads();
  int base = 0;
  for (int c = 0; c < n_chunks; ++c) {
    const bool more = c + 1 < n_chunks;
    if (more) issue(c + 1, first_row(c + 1) + wave, B);  // adjacency loads in flight underneath the filter
    const int n_c = min(kChunk, G - c * kChunk);
    long long tw = pt.now();
    base = wg_filter_chunk<kLdsBm, NT>(S, n_c, bm, n_items, out, base);
    pt.sub(PH_EX_WALKBUSY, tw);
    if (more) {
      commit(c + 1, B);
      rest(c + 1, B.r0);
    }
    __syncthreads();
  }
  pt.sub(PH_EX_LOOP, tsub);
  const int bad = S->bad;
  __syncthreads();
  return bad ? -1 : base;
}

// ---------------------------------------------------------------------------
// wg_expand_hash: GroupGather (one group) + BitmapRefDifference for the traversal that keeps TWO
// queries per CU: the visited set is an exact open-addressing hash set of ids in LDS (64 KB)
// instead of the 1-bit-per-item bitmap (125 KB at 1M items), so the memory phases of one query
// overlap the LDS/VALU phases of the other.
//
// A slot holds (tag << 12) | pos.  pos = 0 marks an id visited before the current piece;
// the copies of an id inside the current piece meet in ONE slot (claimed by CAS on the empty
// value, joined by tag match) and ds_min_u32 leaves the SMALLEST position there -- the copy the
// reference's serial scan keeps (bitmap_ops.cc:224-232).  After one barrier a position is kept
// iff its own value survived in its slot, and the keeper resets pos to 0.  The virtual list
// (CSR rows of the frontier, concatenated) is never staged: thread t owns positions
// j * NT + t of the piece (coalesced over a wavefront), finds its row in O(1) -- a bit per
// list position marks where a row starts, so the row of position p is a prefix popcount -- and
// fetches its id straight from the CSR, the ids of piece c+1 flying underneath piece c.  A probe
// step is ONE LDS round trip (the CAS itself reports who holds the slot), the steps of a thread's
// PER ids are issued together.  Kept ids are compacted in position order with the ballots
// themselves: wave w's j-th ballot IS the keep-mask of positions [j * NT + 64 w, +64).  Two
// barriers per piece of up to 4095 ids (five per 2048 in wg_filter_chunk), no per-piece table.
//
// Up to 2^20 items the tag IS the id (id << 12 | pos).  Beyond, it is not (rounds 1-2 stored the id anyway: 22-bit ids
// of a 4M-item shard left 10 position bits, i.e. pieces of 1023 ids -- four times the barriers and latency chains of a
// 1M-item shard, 34 % of a query at 4M x 256-d; and the tag form costs configs[1] 3 %, so small shards keep the id).  With
// B = bits(n_items), p = id * odd mod 2^B is a bijection of the id space; its top HB bits are the home slot h0, its
// low 14 bits t cover the remainder (B - HB <= 14 bits) and drive the probe stride.  An id that settles k steps down
// its sequence is stored as tag = (t << 6) | k: from (slot, t, k) follow stride(t), h0 = slot - k stride, p, and the
// id -- the set stays EXACT with 20 tag bits whatever the shard size, and every shard gets 12 position bits.
// Every copy of an id walks the same sequence and slots are never vacated, so a copy finds its id at the same k.
// A sequence longer than 62 steps hands the query to the bitmap kernel like a full set (probability ~load^63; step
// 63 is never stored, so no entry equals the empty value 0xffffffff).
// Capacity: the set must stay below SLOTS - 64 entries; a piece that could exceed it is cut to the room left, and with
// less than kVisMinPiece of room returns -2 and the host reruns
// that query on the bitmap kernel.
constexpr uint32_t kVisEmpty = 0xffffffffu;
constexpr int kVisPosBits = 12;   // position in the piece + 1 (0 = visited before the piece)
constexpr int kVisStepBits = 6;   // probe steps an entry may sit from its home slot
template <int NT, int SLOTS>
struct ExpandHashScratch {
  static constexpr int PER = 4096 / NT;        // positions per thread and piece
  static constexpr int kMaskBits = 2 * SLOTS;  // longest virtual list of a round (longer: bitmap kernel)
  static constexpr int kMaskWords = kMaskBits / 32;
  uint32_t rowdelta[kMaxK];                // per non-empty row, in list order: CSR start - offset in the virtual list
  uint32_t startmask[kMaskWords];          // bit p: a row starts at position p of the virtual list
  unsigned short maskprefix[kMaskWords];   // rows that start before word w
  unsigned long long kept[64];             // keep-masks of the piece: word j * (NT/64) + wave
  uint32_t wave_tot[2 * kNW];
  int flags[4];                            // [0] bad id, [1] duplicate in a mark list, [2] probe sequence too long
};

// SLOTS = 16384 (64 KB: two 512-thread workgroups per CU) or 32768 (128 KB: one 1024-thread
// workgroup per CU, for beams whose visited set outgrows the small table).
// vis_key: home slot h0 and the 14 low bits t of id x's image p in a B-bit id space (HB < B <= HB + 14).  Three
// bijections of [0, 2^B) in a row -- multiply by an odd constant, fold the high half onto the low half, multiply
// again -- so that both halves of p depend on every bit of the id.
constexpr int kVisTagBits = 14;
constexpr int kVisMinPiece = 512;  // shortest piece the insert loop is run for when the set is nearly full
constexpr int kVisDirectBits = 32 - kVisPosBits;  // id spaces of up to 20 bits: the entry holds the id itself (rounds 1-2)
// -> home slot h0 and the entry's high part `hi` (everything but step and position)
template <int SLOTS, bool TAG>
__device__ __forceinline__ void vis_key(int32_t x, int id_bits, uint32_t& h0, uint32_t& hi) {
  static_assert(SLOTS == 16384 || SLOTS == 32768, "14 or 15 hash bits");
  constexpr int HB = SLOTS == 16384 ? 14 : 15;
  if constexpr (!TAG) {
    h0 = ((uint32_t)x * 2654435761u) >> (32 - HB);
    hi = (uint32_t)x << kVisPosBits;
    return;
  }
  const uint32_t mask = (1u << id_bits) - 1u;
  uint32_t p = ((uint32_t)x * 2654435761u) & mask;
  p ^= p >> (id_bits >> 1);
  p = (p * 0x85ebca6bu) & mask;
  h0 = p >> (id_bits - HB);
  hi = (p & ((1u << kVisTagBits) - 1u)) << (kVisStepBits + kVisPosBits);
}
// Double hashing: the probe sequence of an id is h0, h0 + s, h0 + 2s, ... with an odd stride s(t) (odd:
// the sequence visits every slot of the power-of-two table).  A wavefront probes until its SLOWEST
// lane is done, i.e. for the longest of 512 probe sequences; with linear probing that tail is the
// longest cluster (measured: 14 k cycles per 4095-id piece), with 2^13 strides it is ~log(512) / log(1 / load)
// steps.  The stride may only depend on bits the slot stores -- that is what lets (slot, tag) name the id --, and
// it needs all 14 of them: strides from the 6 remainder bits of a 1M-item shard (64 strides) or key-independent
// triangular steps cluster (measured at configs[1]: insert 69 k -> 98 k / 111 k cycles per query).
template <int SLOTS>
__device__ __forceinline__ uint32_t vis_stride(uint32_t t) {  // t = the entry's id (direct) or its 14 stored bits
  return ((t * 0x85ebca6bu) >> (SLOTS == 16384 ? 18 : 17)) | 1u;
}

template <int SLOTS>
__device__ __forceinline__ void wg_vis_clear(uint32_t* vis) {
  uint4* p4 = reinterpret_cast<uint4*>(vis);
  for (int i = local_tid(); i < SLOTS / 4; i += (int)blockDim.x)
    p4[i] = make_uint4(kVisEmpty, kVisEmpty, kVisEmpty, kVisEmpty);
}

// The "mark" calls (build_opt_graph.py:119-120,132-133) on the hash set: the list is a TopKV2 output
// over distinct nodes and the set is empty, so BitmapRefDifference returns the list unchanged and
// every id goes in with position 0 ("visited before").  One CAS per probe step.  Returns n, or -1
// on an out-of-range id, -2 when a probe sequence is too long for its tag (the bitmap kernel reruns the query),
// or -4 if an id occurs twice (premise violated: the caller clears the set and runs the ordered
// filter instead).  All NT threads; ends with barriers.
template <int NT, int SLOTS>
__device__ __forceinline__ int wg_mark_hash(const int32_t* list, int n, uint32_t n_items, uint32_t* vis,
                                            int id_bits, int32_t* out, unsigned char* scratch) {
  auto* S = reinterpret_cast<ExpandHashScratch<NT, SLOTS>*>(scratch);
  const int tid = local_tid();
  if (tid < 3) S->flags[tid] = 0;
  __syncthreads();
  constexpr uint32_t kStepOne = 1u << kVisPosBits, kStepEnd = ((1u << kVisStepBits) - 2u) << kVisPosBits;
  const bool direct = id_bits <= kVisDirectBits;  // uniform
  const uint32_t step_inc = direct ? 0u : kStepOne;
  const int tag_shift = direct ? kVisPosBits : kVisStepBits + kVisPosBits;
  for (int i = tid; i < n; i += NT) {
    const int32_t id = list[i];
    if ((uint32_t)id < n_items) {
      uint32_t h, val;
      if (direct) vis_key<SLOTS, false>(id, id_bits, h, val);
      else vis_key<SLOTS, true>(id, id_bits, h, val);
      const uint32_t step = vis_stride<SLOTS>(val >> tag_shift);
      for (;;) {
        const uint32_t c = atomicCAS(&vis[h], kVisEmpty, val);
        if (c == kVisEmpty) break;
        if ((c ^ val) < kStepOne) { S->flags[1] = 1; break; }
        if (step_inc && (val & kStepEnd) == kStepEnd) { S->flags[2] = 1; break; }
        h = (h + step) & (SLOTS - 1);
        val += step_inc;
      }
      out[i] = id;
    } else {
      S->flags[0] = 1;
    }
  }
  __syncthreads();
  const int bad = S->flags[0], dup = S->flags[1], far = S->flags[2];
  __syncthreads();
  return bad ? -1 : far ? -2 : dup ? -4 : n;
}

// All NT threads.  Contract as wg_expand_walk; vis_count (uniform, in/out) = ids in the set.
// Returns ids kept (appended to out[0..)), -1 on an out-of-range id, -2 when the set could overflow
// (or the round's list is longer than the start-bit mask).
template <int NT, int SLOTS, bool TAG>
__device__ __forceinline__ int wg_expand_hash_impl(const int32_t* frontier, int n_frontier,
                                              const int32_t* __restrict__ values,
                                              const int64_t* __restrict__ row_splits, uint32_t n_items,
                                              uint32_t* vis, int id_bits, int& vis_count, int32_t* out,
                                              unsigned char* scratch, int* gathered, SubTimer pt) {
  using Scratch = ExpandHashScratch<NT, SLOTS>;
  constexpr int PER = Scratch::PER;
  constexpr int NWV = NT / 64;
  static_assert(PER * NWV == 64, "one keep-mask word per lane");
  Scratch* S = reinterpret_cast<Scratch*>(scratch);
  long long tsub = pt.now();
  const int tid = local_tid(), lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: indexes readlane below
  const uint64_t lt = lanemask_lt(lane);
  const bool list_mode = row_splits == nullptr;
  const int n_rows = list_mode ? 1 : n_frontier;
  if (tid < 4) S->flags[tid] = 0;
  for (int i = tid; i < Scratch::kMaskWords; i += NT) S->startmask[i] = 0u;
  __syncthreads();
  // ---- pass 1: row lengths -> offsets of the virtual list; a start bit per non-empty row, and per
  //      non-empty row (in list order) the difference between its CSR address and its list offset,
  //      so that position p of the list is values[p + rowdelta[#rows starting at or before p - 1]]
  uint32_t total = 0, rows_ne = 0;
  for (int t0 = 0; t0 < n_rows; t0 += NT) {
    const int t = t0 + tid;
    uint32_t len = 0, start = 0;
    if (list_mode) {
      if (t == 0) len = (uint32_t)n_frontier;
    } else if (t < n_frontier) {
      const int32_t node = frontier[t];
      if ((uint32_t)node < n_items) {
        const int64_t s = row_splits[node], e = row_splits[node + 1];
        start = (uint32_t)s;
        len = (uint32_t)(e - s);
      } else {
        S->flags[0] = 1;
      }
    }
    const uint32_t ne = len ? 1u : 0u;
    const uint32_t inc = wave_scan_add(len), inc_ne = wave_scan_add(ne);
    if (lane == 63) { S->wave_tot[wave] = inc; S->wave_tot[kNW + wave] = inc_ne; }
    __syncthreads();
    uint32_t wbase = 0, tot = 0, wbase_ne = 0, tot_ne = 0;
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
      const uint32_t v = S->wave_tot[w], vn = S->wave_tot[kNW + w];
      if (w < wave) { wbase += v; wbase_ne += vn; }
      tot += v; tot_ne += vn;
    }
    const uint32_t my_off = total + wbase + inc - len;
    if (len && my_off < (uint32_t)Scratch::kMaskBits) {
      S->rowdelta[rows_ne + wbase_ne + inc_ne - 1u] = start - my_off;
      atomicOr(&S->startmask[my_off >> 5], 1u << (my_off & 31));
    }
    total += tot;
    rows_ne += tot_ne;
    __syncthreads();
  }
  *gathered = (int)total;
  const int G = (int)total;
  if (S->flags[0]) return -1;
  if (G > Scratch::kMaskBits) return -2;
  {  // rows that start before each mask word (exclusive prefix of the words' popcounts)
    const int words = (G + 31) >> 5;
    uint32_t run = 0;
    for (int w0 = 0; w0 < words; w0 += NT) {
      const int w = w0 + tid;
      const uint32_t cnt = w < words ? (uint32_t)__popc(S->startmask[w]) : 0u;
      const uint32_t inc = wave_scan_add(cnt);
      if (lane == 63) S->wave_tot[wave] = inc;
      __syncthreads();
      uint32_t wbase = 0, tot = 0;
#pragma unroll
      for (int wv = 0; wv < NWV; ++wv) {
        const uint32_t v = S->wave_tot[wv];
        if (wv < wave) wbase += v;
        tot += v;
      }
      if (w < words) S->maskprefix[w] = (unsigned short)(run + wbase + inc - cnt);
      run += tot;
      __syncthreads();
    }
  }
  pt.sub(PH_EX_PASS1, tsub);
  // ---- pass 2: pieces of PL positions; the ids of piece c+1 are fetched underneath piece c
  constexpr int PL = NT * PER - 1;  // 4095 positions (+1: 12 bits)
  static_assert(PL < (1 << kVisPosBits), "a piece's positions fit the position field");
  constexpr uint32_t pmask = (1u << kVisPosBits) - 1u;
  constexpr uint32_t kStepOne = 1u << kVisPosBits;
  constexpr uint32_t step_inc = TAG ? kStepOne : 0u;  // !TAG: the entry holds the id itself, no step field
  constexpr int tag_shift = TAG ? kVisStepBits + kVisPosBits : kVisPosBits;
  auto fetch = [&](int c0, int n_c, int32_t (&xx)[PER]) {
    uint32_t src[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int pl = j * NT + tid;
      const uint32_t p = (uint32_t)(c0 + min(pl, n_c - 1));  // positions past the piece re-read its last id (dropped)
      const uint32_t w = p >> 5;
      const uint32_t m = S->startmask[w];
      const uint32_t pre = S->maskprefix[w];
      const uint32_t ord = pre + (uint32_t)__popc(m & (0xffffffffu >> (31u - (p & 31u)))) - 1u;
      src[j] = p + S->rowdelta[ord];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) xx[j] = values[src[j]];
  };
  int base = 0;
  bool bad = false;
  int32_t x[PER], xn[PER];
  if (G > 0) fetch(0, min(PL, G), x);
  for (int c0 = 0; c0 < G;) {
    // A piece may add as many ids as it has positions.  One that could pass the set's capacity is CUT to the room that is
    // left (piece boundaries are free: a later piece finds the ids of an earlier one in the set) instead of giving the
    // query up -- on graphs whose rows are all at the cap most of a piece is visited already, and the uncut test handed
    // back queries whose sets ended at 12 k of 16 k entries (round 5: 39 of 4096 on the exact k-NN graph, each a serial
    // tail on the bitmap kernel: profiles/rd5u_knn_graph.txt).  Less than kMinPiece of room: the bitmap kernel.
    int n_c = min(PL, G - c0);
    const int room = SLOTS - 64 - vis_count;
    const bool cut = n_c > room;  // uniform
    if (cut) {
      if (room < kVisMinPiece) return -2;
      n_c = room;
    }
    const int c_next = c0 + n_c;
    long long tw = pt.now();
    if (!cut && c_next < G) fetch(c_next, min(PL, G - c_next), xn);
    pt.sub(PH_EX_LOOKUP, tw);
    // 1. test-and-insert: (tag << 12) | (position in piece + 1), tag = (t << 6) | probe step.  One CAS per
    //    probe step: an empty slot is claimed, a slot with the same tag (= the same id) is joined with ds_min (the
    //    smallest position of an id stays), anything else sends the lane to the next slot of its sequence.
    uint32_t val[PER], h[PER];
    bool act[PER], in_set[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const bool valid = j * NT + tid < n_c;
      const bool inr = (uint32_t)x[j] < n_items;
      bad |= valid && !inr;
      act[j] = in_set[j] = valid && inr;
      uint32_t hi;
      vis_key<SLOTS, TAG>(x[j], id_bits, h[j], hi);
      val[j] = hi | (uint32_t)(j * NT + tid + 1);
    }
    if (pt.on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pt.sub(PH_EX_LOAD, tw);
    // (The loop is written as selects around ONE predicated ds_min: as an if / else-if / else chain the compiler
    //  nests three exec-mask regions with branches per id -- ~45 instructions each, 8 ids per lane and step -- and the
    //  insert phase is bound by that, not by the LDS: `profiles/r4d_*`.)
    for (int it = 0;; ++it) {
      uint32_t c[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) c[j] = act[j] ? atomicCAS(&vis[h[j]], kVisEmpty, val[j]) : kVisEmpty;
      bool any = false;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const bool occupied = c[j] != kVisEmpty;  // (an idle lane and a lane that just claimed its slot: false)
        const bool same = (c[j] >> kVisPosBits) == (val[j] >> kVisPosBits);  // the same tag: the same id
        if (occupied && same) atomicMin(&vis[h[j]], val[j]);
        const bool adv = occupied && !same;
        const uint32_t hn = (h[j] + vis_stride<SLOTS>(TAG ? val[j] >> tag_shift : (uint32_t)x[j])) & (SLOTS - 1);
        h[j] = adv ? hn : h[j];
        if constexpr (TAG) val[j] += adv ? step_inc : 0u;
        act[j] = adv;
        any |= adv;
      }
      if (__ballot(any) == 0ull) break;
      if constexpr (TAG) {
        // every probing lane is `it + 1` steps from home: at 62 its tag cannot name the next slot (uniform test)
        if (it + 1 == (1 << kVisStepBits) - 2) {
          if (any) S->flags[2] = 1;
          break;
        }
      }
    }
    pt.sub(PH_EX_INSERT, tw);
    __syncthreads();
    pt.sub(PH_EX_BARA, tw);
    // 2. a position is kept iff its value survived; the keeper marks the id "visited before"
    uint64_t km[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const bool keep = in_set[j] && vis[h[j]] == val[j];
      if (keep) vis[h[j]] = val[j] & ~pmask;
      km[j] = __ballot(keep);
      if (lane == 0) S->kept[j * NWV + wave] = km[j];
    }
    __syncthreads();
    // a probe sequence outran its tag: the piece's result is void, the bitmap kernel reruns the query (uniform: the flag
    // is only set before the barrier above; read here, behind the check's own LDS traffic, not on the barrier's heels)
    if (TAG && S->flags[2]) return -2;
    pt.sub(PH_EX_CHECK, tw);
    // 3. ordered compaction: exclusive prefix over the 64 mask words (every wavefront on its own)
    const uint32_t cnt = (uint32_t)popc64(S->kept[lane]);
    const uint32_t inc = wave_scan_add(cnt);
    const uint32_t exc = inc - cnt;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint32_t wb = (uint32_t)__builtin_amdgcn_readlane((int)exc, j * NWV + wave);
      if ((km[j] >> lane) & 1ull) out[base + (int)wb + popc64(km[j] & lt)] = x[j];
    }
    const int tot = (int)wave_total(inc);
    base += tot;
    vis_count += tot;
    if (cut) {  // (rare) the piece behind a cut one starts where no prefetch looked
      if (c_next < G) fetch(c_next, min(PL, G - c_next), x);
    } else {
#pragma unroll
      for (int j = 0; j < PER; ++j) x[j] = xn[j];
    }
    c0 = c_next;
    pt.sub(PH_EX_RANK, tw);
  }
  pt.sub(PH_EX_LOOP, tsub);
  if (__ballot(bad) != 0ull && lane == 0) S->flags[0] = 1;
  __syncthreads();
  const int any_bad = S->flags[0];
  __syncthreads();
  return any_bad ? -1 : base;
}

template <int NT, int SLOTS>
__device__ __forceinline__ int wg_expand_hash(const int32_t* frontier, int n_frontier,
                                              const int32_t* __restrict__ values,
                                              const int64_t* __restrict__ row_splits, uint32_t n_items,
                                              uint32_t* vis, int id_bits, int& vis_count, int32_t* out,
                                              unsigned char* scratch, int* gathered, SubTimer pt) {
  if (id_bits <= kVisDirectBits)  // uniform: entries hold the id (two loop bodies: the tag form costs configs[1] 2 %)
    return wg_expand_hash_impl<NT, SLOTS, false>(frontier, n_frontier, values, row_splits, n_items, vis, id_bits,
                                                 vis_count, out, scratch, gathered, pt);
  return wg_expand_hash_impl<NT, SLOTS, true>(frontier, n_frontier, values, row_splits, n_items, vis, id_bits,
                                              vis_count, out, scratch, gathered, pt);
}

// ---------------------------------------------------------------------------
// Row scorers.  LPR = lanes per row = d/8: each lane owns 8 consecutive
// elements (one 16-byte load for f16/bf16), so a wavefront scores 64/LPR rows
// per load instruction and every row is fetched as one contiguous run.
//
// Canonical L2 order (identical in oracle/nann_oracle.c, so scores are
// bit-identical): per lane acc = fma(t_k, t_k, acc) for k = 0..7 with
// t_k = q_k - x_k, then an xor butterfly over the LPR lanes (strides 1, 2, 4,
// ...), score = 0 - sum.
template <int DT>
struct RowChunk {  // 8 consecutive elements of a row
  uint4 a;
  uint4 b;  // second half, f32 rows only
};

template <int DT>
__device__ __forceinline__ RowChunk<DT> load_chunk(const void* table, size_t row, int d, int sub) {
  RowChunk<DT> r;
  if constexpr (DT == DT_F32) {
    const uint4* p = reinterpret_cast<const uint4*>(static_cast<const float*>(table) + row * d + sub * 8);
    r.a = p[0];
    r.b = p[1];
  } else {
    const uint4* p = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(table) + row * d + sub * 8);
    r.a = p[0];
    r.b = make_uint4(0, 0, 0, 0);
  }
  return r;
}

template <int DT>
__device__ __forceinline__ void chunk_to_float(const RowChunk<DT>& r, float x[8]) {
  if constexpr (DT == DT_F32) {
    x[0] = __uint_as_float(r.a.x); x[1] = __uint_as_float(r.a.y);
    x[2] = __uint_as_float(r.a.z); x[3] = __uint_as_float(r.a.w);
    x[4] = __uint_as_float(r.b.x); x[5] = __uint_as_float(r.b.y);
    x[6] = __uint_as_float(r.b.z); x[7] = __uint_as_float(r.b.w);
  } else {
    const uint32_t w[4] = {r.a.x, r.a.y, r.a.z, r.a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (DT == DT_F16) {
        x[2 * i] = half_bits_to_float(w[i] & 0xffffu);
        x[2 * i + 1] = half_bits_to_float(w[i] >> 16);
      } else {
        x[2 * i] = bf16_bits_to_float(w[i] & 0xffffu);
        x[2 * i + 1] = bf16_bits_to_float(w[i] >> 16);
      }
    }
  }
}

template <int LPR>
__device__ __forceinline__ float l2_finish(const float q[8], const float x[8]) {
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = q[k] - x[k];
    acc = __fmaf_rn(t, t, acc);
  }
  // xor butterfly over the LPR lanes of the row.  Strides 1 and 2 are quad permutes; for
  // strides 4 and 8 the mirror patterns pair every lane with a lane of the partner group,
  // all of whose lanes already hold the same partial sum -- the same tree as p[l] + p[l^s].
  acc = acc + dpp_f32<0xB1>(acc);                      // quad_perm [1,0,3,2]  (l ^ 1)
  acc = acc + dpp_f32<0x4E>(acc);                      // quad_perm [2,3,0,1]  (l ^ 2)
  if constexpr (LPR >= 8) acc = acc + dpp_f32<0x141>(acc);   // row_half_mirror      (l ^ 4)
  if constexpr (LPR >= 16) acc = acc + dpp_f32<0x140>(acc);  // row_mirror           (l ^ 8)
  if constexpr (LPR >= 32) acc = acc + __shfl_xor(acc, 16);
  if constexpr (LPR >= 64) acc = acc + __shfl_xor(acc, 32);
  return 0.0f - acc;
}

// The same score from the 16 bytes of an f16 row as they were loaded: t_k = q_k - x_k is ONE v_fma_mix_f32 (x_k * -1 + q_k with
// the f16 operand widened inside the instruction: exact product, one rounding -- the bits of cvt + sub) instead of a
// conversion and a subtraction.  Round 5: the scoring phase of a query spent ~60 % of its cycles ISSUING vector
// instructions (~48 per row chunk: 8 cvt, 8 sub, 8 fma, the butterfly with its DPP wait states, a branchy store, and the
// address arithmetic of the next id), not waiting for HBM.
__device__ __forceinline__ float l2_lane_f16(const float q[8], const uint4& a) {  // this lane's 8 terms of ||q - x||^2
  const uint32_t w[4] = {a.x, a.y, a.z, a.w};
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t0, t1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(w[i]), "v"(q[2 * i]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(w[i]), "v"(q[2 * i + 1]));
    acc = __fmaf_rn(t0, t0, acc);
    acc = __fmaf_rn(t1, t1, acc);
  }
  return acc;
}
__device__ __forceinline__ float l2_lane_bf16(const float q[8], const uint4& a) {
  const uint32_t w[4] = {a.x, a.y, a.z, a.w};
  float acc = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t0 = q[2 * i] - __uint_as_float(w[i] << 16);
    const float t1 = q[2 * i + 1] - __uint_as_float(w[i] & 0xffff0000u);
    acc = __fmaf_rn(t0, t0, acc);
    acc = __fmaf_rn(t1, t1, acc);
  }
  return acc;
}
// The xor butterfly of l2_finish over the 16 lanes of a DPP row for EIGHT rows at once, as a reduce-scatter: every stage
// halves the values a lane carries (it keeps one of a pair and sends the other to the partner that keeps that one), so the
// tree costs 4 x 3 + 2 x 3 + 3 + 1 = 22 vector instructions instead of 8 x 4 adds + 7 selects, and lane l ends with the
// finished sum of row u(l) = (b2^b3, b1^b2, b0^b2) (bits of l & 15; lanes l and 15 - l hold the same row).  Every sum is
// the same tree as l2_finish's -- ((p[l] + p[l^1]) + (the pair next to it)) + (the other quad) + (the other half) -- with
// the operands of an addition possibly swapped, i.e. the same bits.  The keep rules are xors of lane bits because the
// last two stages pair lanes by MIRROR (7 - l, 15 - l: the patterns DPP has for strides 4 and 8), whose partners count
// their quads backwards.  One asm block: the selects read constant lane masks from vcc, and the DPP wait states
// (2 between a vector write and a DPP read of it) are laid out by hand.
__device__ __forceinline__ float l2_rows8_reduce_scatter(float (&a)[8]) {
  float t0, t1, t2, t3;
  asm("s_mov_b32 vcc_lo, 0x5a5a5a5a\n\ts_mov_b32 vcc_hi, 0x5a5a5a5a\n\t"  // lanes with b0 ^ b2: keep the odd row of a pair
      "v_cndmask_b32_e32 %8, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %1, %0, vcc\n\t"
      "v_cndmask_b32_e32 %9, %2, %3, vcc\n\tv_cndmask_b32_e32 %2, %3, %2, vcc\n\t"
      "v_cndmask_b32_e32 %10, %4, %5, vcc\n\tv_cndmask_b32_e32 %4, %5, %4, vcc\n\t"
      "v_cndmask_b32_e32 %11, %6, %7, vcc\n\tv_cndmask_b32_e32 %6, %7, %6, vcc\n\t"
      "v_add_f32_dpp %8, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %2, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %10, %4, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %11, %6, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b32 vcc_lo, 0x3c3c3c3c\n\ts_mov_b32 vcc_hi, 0x3c3c3c3c\n\t"  // b1 ^ b2
      "v_cndmask_b32_e32 %1, %9, %8, vcc\n\tv_cndmask_b32_e32 %3, %11, %10, vcc\n\t"   // sent
      "v_cndmask_b32_e32 %0, %8, %9, vcc\n\tv_cndmask_b32_e32 %2, %10, %11, vcc\n\t"   // kept
      "v_add_f32_dpp %0, %1, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %3, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_mov_b32 vcc_lo, 0x0ff00ff0\n\ts_mov_b32 vcc_hi, 0x0ff00ff0\n\t"  // b2 ^ b3
      "v_cndmask_b32_e32 %8, %2, %0, vcc\n\tv_cndmask_b32_e32 %9, %0, %2, vcc\n\t"     // sent, kept
      "s_nop 0\n\t"
      "v_add_f32_dpp %9, %8, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %9, %9, %9 row_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      :
      : "vcc");
  return t1;
}

// 16-byte row loads per lane in flight in the scoring phase (8 = 128 KB per 1024-thread workgroup).
// Measured on MI355X (profiles/r2a_variants.jsonl): 12 and 16 in flight, and a rolling window that
// refills each slot as soon as it is reduced, are all slower (2.62 / 2.71 / 2.56 ms vs 2.56 ms).
#ifndef NANN_SCORE_U
#define NANN_SCORE_U 8
#endif
#ifndef NANN_SCORE_NT
#define NANN_SCORE_NT 0
#endif
// wg_score_l2_part: scores[i] = -||q - table[ids[i]]||^2 for begin <= i < end, computed by
// NWAVES wavefronts of the workgroup (this one is number wave_rel among them).  No barriers
// inside, so a subset of the workgroup can run it.
// ids must be in range (the visited filter and index validation guarantee it on the fused path).
// qv: f32[d] (LDS or global).  U row loads per lane are in flight at once (U * 16 KB per
// workgroup), and the candidate ids of the next batch are fetched underneath them.
template <int LPR, int DT, int NWAVES>
__device__ __forceinline__ void wg_score_l2_part(const void* __restrict__ table, int d, const int32_t* ids,
                                                 int begin, int end, const float* qv, float* scores,
                                                 int wave_rel, bool near = false) {
  constexpr int U = (DT == DT_F32) ? NANN_SCORE_U / 2 : NANN_SCORE_U;
  constexpr int GPW = 64 / LPR;      // rows per wavefront per load
  constexpr int RPI = NWAVES * GPW;  // rows per iteration of the participating wavefronts
  if (end <= begin) return;
  const int lane = local_tid() & 63;
  const int sub = lane % LPR, grp = lane / LPR;
  const int slot = wave_rel * GPW + grp;
  float q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = qv[sub * 8 + k];
  if constexpr (LPR >= 16 && DT != DT_F32) {
    // Rows of one or more DPP rows (round 5).  A batch's ids arrive in ONE load per lane -- lane `sub` of a row's group
    // holds the id of load u = sub & (U - 1) -- and reach the group by row_newbcast (every 16-lane DPP row of a group has
    // lane u at position u); a batch's U scores leave in ONE store, lanes 0..7 of the group with the row sum the
    // reduce-scatter butterfly left them (l2_rows8_reduce_scatter): no control flow inside a batch.
    // `near` (uniform): every row starts below 4 GB and ids are below 2^24 -- the row's address is ONE v_mad_u32_u24 on
    // top of the scalar base instead of a 64-bit multiply-add and a 64-bit shift-add.
    static_assert(U == 8, "eight broadcasts below");
    constexpr uint32_t kRowBytes = LPR * 16;
    const int mine_at = (sub & (U - 1)) * RPI + slot;
    const int out_at = ((sub & 7) ^ ((sub & 4) ? 3 : 0)) * RPI + slot;  // the row whose sum the reduce-scatter leaves in this lane
    auto run = [&](auto near_c) {
      constexpr bool NEAR = decltype(near_c)::value;
      // (NEAR: the lane that holds an id holds its row's byte offset; the broadcast rides on the `or` with the lane's own
      //  16 bytes of the row -- one vector instruction per row address)
      auto row = [&](int32_t v) -> uint4 {
        if constexpr (NEAR) {
#if NANN_SCORE_NT  // measurement builds: the rows as non-temporal loads (a row is read once per query)
          typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
          const u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(static_cast<const unsigned char*>(table) + ((uint32_t)v | ((uint32_t)sub * 16u))));
          return uint4{r.x, r.y, r.z, r.w};
#else
          return *reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(table) + ((uint32_t)v | ((uint32_t)sub * 16u)));
#endif
        } else {
          return load_chunk<DT>(table, (size_t)v, d, sub).a;
        }
      };
      auto widen = [&](int32_t id) -> int32_t { return NEAR ? (int32_t)__umul24((uint32_t)id, kRowBytes) : id; };
      // Rolling refill: a slot's next row is requested as soon as its 8 terms are summed, so a batch's loads travel
      // underneath the rest of the previous batch's arithmetic, the reduce-scatter and the store instead of behind them (the
      // loop used to be load 8 - wait - compute 8; A/B on one box: +1.3 %, profiles/rd5ad_phase_repeat_and_rolling.txt).  ids are
      // fetched two batches ahead; the last batch refills nothing.
      auto ids_of = [&](int at) -> int32_t { return widen(ids[min(at + mine_at, end - 1)]); };  // past `end`: candidate end-1, dropped
      int32_t idv = ids_of(begin);
      uint4 ch[U];
      ch[0] = row(dpp_i32<0x150>(idv));
      ch[1] = row(dpp_i32<0x151>(idv));
      ch[2] = row(dpp_i32<0x152>(idv));
      ch[3] = row(dpp_i32<0x153>(idv));
      ch[4] = row(dpp_i32<0x154>(idv));
      ch[5] = row(dpp_i32<0x155>(idv));
      ch[6] = row(dpp_i32<0x156>(idv));
      ch[7] = row(dpp_i32<0x157>(idv));
      idv = ids_of(begin + RPI * U);
      auto lane_sum = [&](const uint4& c) -> float { return DT == DT_F16 ? l2_lane_f16(q, c) : l2_lane_bf16(q, c); };
      auto finish = [&](float (&s)[U], int i0) {
        float mine = l2_rows8_reduce_scatter(s);
        if constexpr (LPR >= 32) mine = mine + __shfl_xor(mine, 16);
        if constexpr (LPR >= 64) mine = mine + __shfl_xor(mine, 32);
        const int i = i0 + out_at;
        if (sub < U && i < end) scores[i] = 0.0f - mine;
      };
      int i0 = begin;
      for (; i0 + RPI * U < end; i0 += RPI * U) {
        const int32_t idn = ids_of(i0 + 2 * RPI * U);
        float s[U];
#define NANN_ROLL_SLOT(u, ctrl)                 \
        s[u] = lane_sum(ch[u]);                     \
        ch[u] = row(dpp_i32<ctrl>(idv));            \
        __builtin_amdgcn_sched_barrier(0);
        NANN_ROLL_SLOT(0, 0x150) NANN_ROLL_SLOT(1, 0x151) NANN_ROLL_SLOT(2, 0x152) NANN_ROLL_SLOT(3, 0x153)
        NANN_ROLL_SLOT(4, 0x154) NANN_ROLL_SLOT(5, 0x155) NANN_ROLL_SLOT(6, 0x156) NANN_ROLL_SLOT(7, 0x157)
#undef NANN_ROLL_SLOT
        finish(s, i0);
        idv = idn;
      }
      {
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = lane_sum(ch[u]);
        finish(s, i0);
      }
    };
    if (near) run(std::true_type{}); else run(std::false_type{});
    return;
  }
  // branch-free: positions past `end` re-read candidate end-1 and their result is dropped
  int32_t nxt[U];
#pragma unroll
  for (int u = 0; u < U; ++u) nxt[u] = ids[min(begin + u * RPI + slot, end - 1)];
  for (int i0 = begin; i0 < end; i0 += RPI * U) {
    RowChunk<DT> ch[U];
#pragma unroll
    for (int u = 0; u < U; ++u) ch[u] = load_chunk<DT>(table, (size_t)nxt[u], d, sub);
#pragma unroll
    for (int u = 0; u < U; ++u)  // ids of the next batch, underneath the row loads
      nxt[u] = ids[min(i0 + RPI * U + u * RPI + slot, end - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPI + slot;
      float x[8];
      chunk_to_float<DT>(ch[u], x);
      const float s = l2_finish<LPR>(q, x);
      if (sub == 0 && i < end) scores[i] = s;
    }
  }
}

// whole workgroup, i < n
template <int LPR, int DT, int NTHREADS>
__device__ __forceinline__ void wg_score_l2(const void* __restrict__ table, int d, const int32_t* ids,
                                            int n, const float* qv, float* scores) {
  wg_score_l2_part<LPR, DT, NTHREADS / 64>(table, d, ids, 0, n, qv, scores, local_tid() >> 6);
}

// ---------------------------------------------------------------------------
// wg_topk: TopKV2 (sorted) over one row of n scores, all kNT threads.
// Result order is the strict total order of topk_op.cc:134-142 -- value
// descending, ties by lower input position -- so any correct selection is
// bit-identical to the reference's heap.
//   1. keys = monotone u32 images of the scores, kept in registers
//      (kTopkEPT per thread; larger n re-reads them from memory);
//   2. bitwise binary search for the k-th largest key: 1 bit per step, counts
//      by ballot + scalar popcount, one LDS atomic per wavefront, early exit
//      as soon as exactly k keys are >= the probe;
//   3. selected (key, ~position) pairs are appended to LDS; keys equal to the
//      threshold are admitted in position order when not all of them fit;
//   4. rank sort of the k pairs in LDS, outputs written in rank order.
// out_pos / out_ids / out_scores / out_mapped may each be null.  ids == null
// means "ids are positions".  Returns NANN status (uniform).
template <int KCAP>  // largest k (kMaxK for the serving kernels; the evaluation traversal keeps more per level)
struct TopkScratchT {
  unsigned long long sel[KCAP < 512 ? 512 : KCAP];  // first: 16-byte aligned (the four 256-bin radix histograms alias it)
  union {
    unsigned short prank[kNT];    // partial ranks of the all-pairs rank sort: [segment][element]
    // the bin-grouped ranking (BIN, round 6): the selected pairs are GROUPED by the leading radix digit of their keys --
    // above[b] = keys of the row with a larger digit = the rank at which bin b starts, cursor[b] = selected pairs in bin b -- so
    // that a pair is ranked against its own bin only (step 4).  Dead before the all-pairs ranking (its fallback) starts.
    struct { uint32_t above[256]; uint32_t cursor[256]; } bins;
  };
  uint32_t misc[4];               // [0] nsel, [2] unordered append cursor, [3] a bin overflowed (BIN)
  uint32_t wlo[kNW], whi[kNW];    // per-wavefront min / max key (step 2a)
  uint32_t wcnt[kNW];
};
static_assert(kNT * sizeof(unsigned short) == 512 * sizeof(uint32_t), "the bin tables overlay the partial ranks");
typedef TopkScratchT<kMaxK> TopkScratch;
// candidate scores of the current round, kept in LDS behind the top-k scratch so that
// the selection does not wait on L2 (positions < kLdsScores only)
static_assert(sizeof(TopkScratch) <= kLdsScoresOff, "top-k scratch overlaps the LDS scores");
static_assert(kLdsScoresOff + kLdsScores * 4 <= kPhaseScratch, "phase scratch too small");
static_assert(sizeof(ExpandWalkScratch) <= kPhaseScratch, "phase scratch too small");

// NS = register slots per thread (n <= NS * kNT); NS == 0 re-reads keys from memory.
// SCL = the first n scores are also in LDS (lds_scores); requires NS > 0.
// BIN: rank the selected pairs bin by bin (TopkScratchT::bins)
template <int NS, bool SCL, int NT, int KCAP = kMaxK, bool BIN = false>
__device__ __forceinline__ int wg_topk_impl(const int32_t* ids, const float* scores,
                                            const float* lds_scores, int n, int k, int32_t* out_pos,
                                            int32_t* out_ids, float* out_scores, const int64_t* id_map,
                                            int64_t* out_mapped, unsigned char* scratch, SubTimer pt,
                                            int n_all = 0x7fffffff, uint32_t floor_key = 0u) {
  // (BIN only) positions >= n_all take part only with a key above floor_key: a row that is a sorted list of >= k kept results
  // followed by candidates needs only the candidates that beat the worst kept result -- TopKV2 breaks ties towards the lower
  // position, so a candidate at or below it can displace nothing; same output, a fraction of the histogram traffic
  TopkScratchT<KCAP>* S = reinterpret_cast<TopkScratchT<KCAP>*>(scratch);
  long long tsub = pt.now();
  const int tid = local_tid(), lane = tid & 63, wave = tid >> 6;
  const uint64_t lt = lanemask_lt(lane);
  constexpr bool REG = NS > 0;
  uint32_t key[REG ? NS : 1];
  if constexpr (REG) {
    // unconditional (clamped) loads: all NS of them in flight together
    float raw[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int i = min(j * NT + tid, n - 1);
      raw[j] = SCL ? lds_scores[i] : scores[i];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) key[j] = (j * NT + tid < n) ? score_key(raw[j]) : 0u;
  }
#define NANN_FOR_KEYS(...)                                                    \
  if constexpr (REG) {                                                        \
    _Pragma("unroll") for (int j = 0; j < NS; ++j) {                          \
      const int i = j * NT + tid;                                             \
      const uint32_t kj = key[j];                                             \
      const bool valid = i < n && (!BIN || i < n_all || kj > floor_key);      \
      __VA_ARGS__                                                             \
    }                                                                         \
  } else {                                                                    \
    for (int i0 = 0; i0 < n; i0 += NT) {                                      \
      const int i = i0 + tid;                                                 \
      const uint32_t kj = i < n ? score_key(scores[i]) : 0u;                  \
      const bool valid = i < n && (!BIN || i < n_all || kj > floor_key);      \
      __VA_ARGS__                                                             \
    }                                                                         \
  }

  // ---- 2a. range of the keys: the search runs on key - min(key), whose leading digit is spread
  //          over the bins (the raw keys of scores within a few binades share all but 2-3 values of
  //          their top 8 undecided bits, and same-bin LDS atomics serialise).  Round 6: per-wavefront
  //          results, reduced by everybody behind the ONE barrier that also covers the zeroing below
  //          (it was three LDS atomics per wavefront and a barrier of its own).
  {
    uint32_t lo = 0xffffffffu, hi = 0u, nv = 0u;
    NANN_FOR_KEYS({ if (valid) { lo = min(lo, kj); hi = max(hi, kj); ++nv; } })
    lo = wave_min(lo);
    hi = wave_max(hi);
    const uint32_t wv = BIN ? wave_total(wave_scan_add(nv)) : 0u;  // the keys that take part
    if (lane == 0) { S->wlo[wave] = lo; S->whi[wave] = hi; S->wcnt[wave] = wv; }
  }
  uint32_t* hist = reinterpret_cast<uint32_t*>(S->sel);  // [4][256]; sel is not in use before step 3
  for (int i = tid; i < 4 * 256; i += NT) hist[i] = 0;
  constexpr bool BINNED = BIN;  // (the serving kernels keep the all-pairs ranking: their k is 200)
  const uint32_t bin_max = (uint32_t)max(64, k >> 1);  // a fuller bin (half the keys crowded into one leading digit): the all-pairs ranking instead
  if constexpr (BINNED) {
    for (int i = tid; i < 256; i += NT) S->bins.cursor[i] = 0;
  }
  if (tid < 4) S->misc[tid] = 0;
  __syncthreads();
  pt.sub(PH_TK_LOAD, tsub);
  uint32_t kbase = 0xffffffffu, kmax = 0u, nvalid = 0u;
  for (int w = 0; w < NT / 64; ++w) {
    kbase = min(kbase, S->wlo[w]);  // smallest key
    kmax = max(kmax, S->whi[w]);
    nvalid += S->wcnt[w];
  }
  const uint32_t diff = kmax - kbase;  // largest key - smallest key
  // (wcnt is reused by the collect of keys equal to the threshold: every thread has read it before the barriers in between)
  // ---- 2b. k-th largest key: radix select over the undecided bits, 8 bits per pass
  //          (LDS histogram -> 256-bin suffix scan).  A pass ends the search early when the
  //          bin holding the k-th key is needed in full.  T is relative to kbase until the end.
  uint32_t T = 0, c_ge = BIN ? nvalid : (uint32_t)n, c_gt = 0;
  int shift0 = 0;        // the leading digit of a key: ((key - kbase) >> shift0) & mask0 (pass 0 of the search)
  uint32_t mask0 = 0u;
  if (diff != 0u) {
    const int hb = 31 - __clz((int)diff);  // highest differing bit
    {
      const int nb0 = hb + 1 < 8 ? hb + 1 : 8;
      shift0 = hb + 1 - nb0;
      mask0 = (1u << nb0) - 1u;
    }
    T = 0u;
    int top = hb + 1;          // undecided low bits
    uint32_t kk = (uint32_t)k;  // still to find among keys that match T above `top`
    bool exact = false;
    for (int pass = 0; top > 0 && !exact; ++pass) {
      const int nb = top < 8 ? top : 8;
      const int shift = top - nb;
      const uint32_t dmask = (1u << nb) - 1u;
      uint32_t* h = hist + pass * 256;
      NANN_FOR_KEYS({
        const uint32_t kq = kj - kbase;
        if (valid && (top >= 32 || (kq >> top) == (T >> top))) atomicAdd(&h[(kq >> shift) & dmask], 1u);
      })
      __syncthreads();
      // every wavefront scans the 256 bins on its own (no further barrier): lane l owns bins
      // 4l..4l+3; suffix sums over lanes give #keys with a larger digit
      const uint4 hv = reinterpret_cast<const uint4*>(h)[lane];
      const uint32_t s_l = hv.x + hv.y + hv.z + hv.w;
      const uint32_t pre = wave_scan_add(s_l);              // inclusive prefix over lanes <= l
      const uint32_t suf = wave_total(pre) - pre + s_l;      // inclusive suffix over lanes >= l
      const uint32_t ab3 = suf - s_l;   // #keys with digit > 4l+3
      const uint32_t ab2 = ab3 + hv.w;  // > 4l+2
      const uint32_t ab1 = ab2 + hv.z;  // > 4l+1
      const uint32_t ab0 = ab1 + hv.y;  // > 4l
      if constexpr (BINNED) {
        if (pass == 0 && wave == 0) { S->bins.above[4 * lane] = ab0; S->bins.above[4 * lane + 1] = ab1; S->bins.above[4 * lane + 2] = ab2; S->bins.above[4 * lane + 3] = ab3; }
      }
      int hit = -1;
      if (ab3 < kk && kk <= ab3 + hv.w) hit = 3;
      else if (ab2 < kk && kk <= ab2 + hv.z) hit = 2;
      else if (ab1 < kk && kk <= ab1 + hv.y) hit = 1;
      else if (ab0 < kk && kk <= ab0 + hv.x) hit = 0;
      const uint64_t hm = __ballot(hit >= 0);  // exactly one lane
      const int src = __ffsll((unsigned long long)hm) - 1;
      const uint32_t my_above = hit == 3 ? ab3 : hit == 2 ? ab2 : hit == 1 ? ab1 : ab0;
      const uint32_t my_inbin = hit == 3 ? hv.w : hit == 2 ? hv.z : hit == 1 ? hv.y : hv.x;
      const uint32_t sel_bin = (uint32_t)__builtin_amdgcn_readlane((int)(4 * lane + (hit < 0 ? 0 : hit)), src);
      const uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)my_above, src);
      const uint32_t inbin = (uint32_t)__builtin_amdgcn_readlane((int)my_inbin, src);
      T |= sel_bin << shift;
      c_gt += above;
      kk -= above;
      c_ge = c_gt + inbin;
      exact = (inbin == kk);
      top = shift;
    }
    T += kbase;
  } else {
    T = kbase;  // all keys equal
  }
  // c_ge = #keys >= T >= k; c_gt = #keys > T (when the search ran to the last bit).  If
  // c_ge > k, T is the exact k-th key and only some of the keys equal to T are admitted.
  __syncthreads();  // the histograms alias sel: every wavefront is done scanning them
  pt.sub(PH_TK_SEARCH, tsub);
  const bool partial_eq = c_ge > (uint32_t)k;
  // ---- 3. collect -----------------------------------------------------------
  const bool use_bins = BINNED && diff != 0u;  // (uniform)
  // a selected pair goes to the next free place of its bin: every key of a higher bin is selected too, so bin b's pairs are
  // the ranks [above[b], above[b] + cursor[b])
  auto place = [&](uint32_t kj, int i) {
    const uint32_t bin = ((kj - kbase) >> shift0) & mask0;
    const uint32_t p = atomicAdd(&S->bins.cursor[bin], 1u);
    if (p >= bin_max) S->misc[3] = 1u;
    S->sel[S->bins.above[bin] + p] = ((unsigned long long)kj << 32) | (uint32_t)(~(uint32_t)i);
  };
  if (use_bins) {
    NANN_FOR_KEYS({
      if (valid && (partial_eq ? kj > T : kj >= T)) place(kj, i);
    })
    if (partial_eq) {  // keys == T: the first (k - c_gt) in position order (as below), placed in T's bin
      const uint32_t r = (uint32_t)k - c_gt;
      uint32_t eq_base = 0;
      NANN_FOR_KEYS({
        const bool e = valid && kj == T;
        const uint64_t m = __ballot(e);
        if (lane == 0) S->wcnt[wave] = (uint32_t)popc64(m);
        __syncthreads();
        uint32_t wb = 0, tot = 0;
        for (int w = 0; w < NT / 64; ++w) {
          const uint32_t t = S->wcnt[w];
          if (w < wave) wb += t;
          tot += t;
        }
        const uint32_t rank = eq_base + wb + (uint32_t)popc64(m & lt);
        if (e && rank < r) place(kj, i);
        eq_base += tot;
        __syncthreads();
      })
    }
  } else if (!partial_eq) {
    NANN_FOR_KEYS({
      const bool s = valid && kj >= T;
      const uint64_t m = __ballot(s);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&S->misc[0], (uint32_t)popc64(m));
        b = __builtin_amdgcn_readfirstlane(b);
        if (s) S->sel[b + popc64(m & lt)] = ((unsigned long long)kj << 32) | (uint32_t)(~(uint32_t)i);
      }
    })
  } else {
    // keys > T: any slot in [0, c_gt) (unordered append)
    NANN_FOR_KEYS({
      const bool s = valid && kj > T;
      const uint64_t m = __ballot(s);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&S->misc[2], (uint32_t)popc64(m));
        b = __builtin_amdgcn_readfirstlane(b);
        if (s) S->sel[b + popc64(m & lt)] = ((unsigned long long)kj << 32) | (uint32_t)(~(uint32_t)i);
      }
    })
    // keys == T: first (k - c_gt) in position order.  Positions ascend with
    // (j, tid), so an ordered prefix count over (iteration, wave, lane) ranks them.
    const uint32_t r = (uint32_t)k - c_gt;
    uint32_t eq_base = 0;
    NANN_FOR_KEYS({
      const bool e = valid && kj == T;
      const uint64_t m = __ballot(e);
      if (lane == 0) S->wcnt[wave] = (uint32_t)popc64(m);
      __syncthreads();
      uint32_t wb = 0, tot = 0;
      for (int w = 0; w < NT / 64; ++w) {
        const uint32_t t = S->wcnt[w];
        if (w < wave) wb += t;
        tot += t;
      }
      const uint32_t rank = eq_base + wb + (uint32_t)popc64(m & lt);
      if (e && rank < r) S->sel[c_gt + rank] = ((unsigned long long)kj << 32) | (uint32_t)(~(uint32_t)i);
      eq_base += tot;
      __syncthreads();
    })
  }
#undef NANN_FOR_KEYS
  __syncthreads();
  pt.sub(PH_TK_COLLECT, tsub);
  // ---- 4. rank sort + output --------------------------------------------------
  // rank(e) = #{o : sel[o] > sel[e]} (pairs are distinct).  The k x k comparisons are
  // spread over all threads: element e = tid % K2, comparison segment = tid / K2.
  int K2 = 64;
  while (K2 < k) K2 <<= 1;
  const bool split = K2 <= NT;  // else (k > NT: NT < kMaxK, or the evaluation traversal's k up to 2048): every thread ranks its elements in full
  // the id of "my" element (e = tid) is fetched now so that its latency hides under the ranking
  int32_t my_id = 0;
  if (tid < k && ids) my_id = ids[(int)(~(uint32_t)(S->sel[tid] & 0xffffffffull))];
  const bool binned = use_bins && S->misc[3] == 0u;  // (uniform: read behind the barrier)
  if (split && !binned) {
    const int segs = NT / K2;
    const int e = tid & (K2 - 1), seg = tid / K2;
    const int len = (k + segs - 1) / segs;
    const int o0 = seg * len, o1 = min(k, o0 + len);
    int pr = 0;
    if (e < k) {
      const unsigned long long mine = S->sel[e];
      int o = o0;
      for (; o + 8 <= o1; o += 8) {  // eight broadcast reads in flight per step (a one-by-one loop pays an LDS round trip per element)
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = S->sel[o + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) pr += (v[u] > mine) ? 1 : 0;
      }
      for (; o < o1; ++o) pr += (S->sel[o] > mine) ? 1 : 0;
    }
    S->prank[tid] = (unsigned short)pr;
    __syncthreads();
  }
  for (int e = tid; e < k; e += NT) {
    const unsigned long long mine = S->sel[e];
    int rank = 0;
    if (binned) {  // against the pairs of its own bin
      const uint32_t bin = (((uint32_t)(mine >> 32) - kbase) >> shift0) & mask0;
      const int lo = (int)S->bins.above[bin], hi = lo + (int)S->bins.cursor[bin];
      rank = lo;
      int o = lo;
      for (; o + 4 <= hi; o += 4) {
        unsigned long long v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = S->sel[o + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) rank += (v[u] > mine) ? 1 : 0;
      }
      for (; o < hi; ++o) rank += (S->sel[o] > mine) ? 1 : 0;
    } else if (split) {
      for (int sg = 0; sg < NT / K2; ++sg) rank += S->prank[sg * K2 + e];
    } else {
      int o = 0;
      for (; o + 8 <= k; o += 8) {  // (eight broadcast reads in flight, as above)
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = S->sel[o + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (v[u] > mine) ? 1 : 0;
      }
      for (; o < k; ++o) rank += (S->sel[o] > mine) ? 1 : 0;
    }
    const int pos = (int)(~(uint32_t)(mine & 0xffffffffull));
    const int32_t idv = ids ? (e == tid ? my_id : ids[pos]) : pos;
    if (out_pos) out_pos[rank] = pos;
    if (out_ids) out_ids[rank] = idv;
    if (out_scores) {
      // the key is an invertible image of the score except that -0 was folded into +0: rebuild the
      // score from the key (no dependent read of scores[pos]) unless it is a zero
      const uint32_t kbits = (uint32_t)(mine >> 32);
      const uint32_t u = (kbits & 0x80000000u) ? (kbits & 0x7fffffffu) : ~kbits;
      out_scores[rank] = (u != 0u) ? __uint_as_float(u) : (SCL ? lds_scores[pos] : scores[pos]);
    }
    if (out_mapped) out_mapped[rank] = id_map[idv];
  }
  __syncthreads();
  pt.sub(PH_TK_SORT, tsub);
  return 0;
}

// lds_scores != nullptr: the first n scores are mirrored in LDS (n <= kLdsScores)
template <int NT = kNT, int KCAP = kMaxK, bool BIN = false>
__device__ __forceinline__ int wg_topk(const int32_t* ids, const float* scores, const float* lds_scores,
                                       int n, int k, int32_t* out_pos, int32_t* out_ids,
                                       float* out_scores, const int64_t* id_map, int64_t* out_mapped,
                                       unsigned char* scratch, SubTimer pt = no_timer()) {
  if (k < 0 || k > KCAP) return 7;  // NANN_ERR_BAD_ARGUMENT
  if (n < k) return 4;               // NANN_ERR_TOPK_K_GT_N, topk_op.cc:67-71
  if (k == 0) return 0;
#define NANN_TOPK_CASE(NS_, SCL_)                                                                   \
  return wg_topk_impl<NS_, SCL_, NT, KCAP, BIN>(ids, scores, lds_scores, n, k, out_pos, out_ids, out_scores, \
                                           id_map, out_mapped, scratch, pt)
  if (lds_scores != nullptr && n <= kLdsScores) {
    if (n <= 1 * NT) NANN_TOPK_CASE(1, true);
    if (n <= 2 * NT) NANN_TOPK_CASE(2, true);
    if (n <= 4 * NT) NANN_TOPK_CASE(4, true);
    if constexpr (8 * NT <= kLdsScores) NANN_TOPK_CASE(8, true);
  }
  if (n <= 1 * NT) NANN_TOPK_CASE(1, false);
  if (n <= 2 * NT) NANN_TOPK_CASE(2, false);
  if (n <= 4 * NT) NANN_TOPK_CASE(4, false);
  if (n <= 8 * NT) NANN_TOPK_CASE(8, false);
  if (n <= kTopkEPT * NT) NANN_TOPK_CASE(kTopkEPT, false);
  NANN_TOPK_CASE(0, false);
#undef NANN_TOPK_CASE
}

// The same with three register-slot cases instead of ten and the bin-grouped ranking: the evaluation traversal's LDS form (its n
// is in the thousands, its k in the hundreds; every case is a copy of the whole selection in every instance of the kernel)
// n_all / floor_key: wg_topk_impl (the caller vouches that at least k keys take part).
template <int NT, int KCAP>
__device__ __forceinline__ int wg_topk_binned(const int32_t* ids, const float* scores, int n, int k, int32_t* out_ids,
                                              float* out_scores, unsigned char* scratch, int n_all = 0x7fffffff,
                                              uint32_t floor_key = 0u) {
  if (k < 0 || k > KCAP) return 7;  // NANN_ERR_BAD_ARGUMENT
  if (n < k) return 4;               // NANN_ERR_TOPK_K_GT_N, topk_op.cc:67-71
  if (k == 0) return 0;
  if (n <= 4 * NT) return wg_topk_impl<4, false, NT, KCAP, true>(ids, scores, nullptr, n, k, nullptr, out_ids, out_scores, nullptr, nullptr, scratch, no_timer(), n_all, floor_key);
  if (n <= 8 * NT) return wg_topk_impl<8, false, NT, KCAP, true>(ids, scores, nullptr, n, k, nullptr, out_ids, out_scores, nullptr, nullptr, scratch, no_timer(), n_all, floor_key);
  return wg_topk_impl<0, false, NT, KCAP, true>(ids, scores, nullptr, n, k, nullptr, out_ids, out_scores, nullptr, nullptr, scratch, no_timer(), n_all, floor_key);
}

}  // namespace nann
