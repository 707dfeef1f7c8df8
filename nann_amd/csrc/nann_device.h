// nann_device.h -- workgroup-level building blocks of the gfx950 retrieval path.
//
// Written for CDNA4 only: 64-lane wavefronts (ballot masks are 64-bit, lane
// arithmetic is & 63), 160 KiB LDS per CU, one 1024-thread workgroup per CU.
// Each block below states which reference loop it replaces (paths relative to
// /root/reference/, UO/ = tensorflow/tensorflow/core/user_ops/).
//
//   wave_walk    BitmapRefDifference::Differ   UO/bitmap_op/bitmap_ops.cc:221-234
//   wg_expand    GroupGather::Fill             UO/beam_search_op/GroupGather_kernel.cc:137-168
//   wg_score     GatherV2 + scorer             core/kernels/gather_functor.h:96-103 + BlazeXlaOp
//   wg_topk      TopKV2 (+ Gather of ids)      core/kernels/topk_op.cc:104-205
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace nann {

constexpr int kNT = 1024;            // threads per traversal workgroup
constexpr int kNW = kNT / 64;        // 16 wavefronts
constexpr int kTopkEPT = 16;         // top-k keys held in registers per thread (n <= 16384)
constexpr int kMaxK = 1024;          // largest k / frontier a workgroup handles
constexpr int kPhaseScratch = 16384; // LDS bytes shared by the phases below
constexpr int kMaxD = 512;

enum : int { DT_F16 = 0, DT_BF16 = 1, DT_F32 = 2 };

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
__device__ __forceinline__ int popc64(uint64_t m) { return __popcll(m); }

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
  for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) p4[i] = make_uint4(0, 0, 0, 0);
}

// ---------------------------------------------------------------------------
// wave_walk: ordered first-occurrence filter against the visited bitmap.
//
// Reference semantics (bitmap_ops.cc:224-232): scan ids in order; keep an id
// iff its bit is clear, then set the bit.  Here ONE wavefront walks the list 64
// ids at a time.  Per chunk every lane reads its word (pre), then ORs its bit
// in (old); LDS executes one wave's instructions in order, so chunk c+1 sees
// every bit chunk c set -- the serial scan order is preserved across chunks
// with the atomics still pipelined.  Inside a chunk, lanes holding the same
// fresh id are resolved to the LOWEST lane with ballots (which lane the LDS
// arbiter happened to serve first does not matter).  Kept ids are compacted
// with ballot + popcount (stable), so the output order is the serial order.
//
// kLds=false walks a bitmap in global memory (shards too large for LDS): the
// pre-read is then an atomic OR of 0 so that it is served by L2 like the
// update, and the dependency on `old` keeps chunks ordered.
//
// Must be called by all lanes of exactly one wavefront.  Returns the number of
// ids kept (wave-uniform).  *err is set to 1 if an id is outside [0, n_items).
template <bool kLds>
__device__ int wave_walk(const int32_t* in, int n, uint32_t* bm, uint32_t n_items,
                         int32_t* out, int* err) {
  constexpr int U = kLds ? 8 : 2;
  const int lane = lane_id();
  const uint64_t lt = lanemask_lt();
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 64 * U) {
    int32_t idv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = c0 + u * 64 + lane;
      idv[u] = (i < n) ? in[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (c0 + u * 64 >= n) break;
      const int i = c0 + u * 64 + lane;
      const int32_t x = idv[u];
      const bool valid = i < n;
      const bool inrange = valid && (uint32_t)x < n_items;
      if (__ballot(valid && !inrange) != 0ull) {
        if (lane == 0) *err = 1;
      }
      const uint32_t word = (uint32_t)x >> 5;
      const uint32_t bit = 1u << (x & 31);
      uint32_t pre = 0xffffffffu, old = 0xffffffffu;
      if (inrange) {
        pre = kLds ? bm[word] : atomicOr(&bm[word], 0u);
        old = atomicOr(&bm[word], bit);
      }
      const bool fresh = inrange && !(pre & bit);
      bool keep = inrange && !(old & bit);
      uint64_t dupl = __ballot(fresh && !keep);
      while (dupl) {  // duplicate fresh ids inside this chunk: lowest lane wins
        const int l = __ffsll((unsigned long long)dupl) - 1;
        const int32_t xv = __shfl(x, l);
        const bool mine = fresh && x == xv;
        const uint64_t same = __ballot(mine);
        const int first = __ffsll((unsigned long long)same) - 1;
        if (mine) keep = (lane == first);
        dupl &= ~same;
      }
      const uint64_t m = __ballot(keep);
      if (keep) out[base + popc64(m & lt)] = x;
      base += popc64(m);
    }
  }
  return base;
}

// ---------------------------------------------------------------------------
// wg_expand: concatenate the CSR rows of a frontier, in frontier order,
// duplicates kept (GroupGather with one group, GroupGather_kernel.cc:137-168;
// build_opt_graph.py:39-49).  All kNT threads.  n_frontier <= kMaxK.
// Pass 1 (count, :137-145): one thread per frontier node reads its two
// row_splits and a block-wide exclusive scan turns the lengths into output
// offsets.  Pass 2 (fill, :152-168): one wavefront per row copies it with a
// single coalesced load/store per 64 neighbours.
// Returns the number of neighbours written to `raw` (uniform); -1 if a
// frontier id is out of range.
struct ExpandScratch {
  uint32_t off[kMaxK + 1];
  uint32_t rowstart[kMaxK];
  uint32_t wave_tot[kNW];
  int bad;
};

__device__ int wg_expand(const int32_t* frontier, int n_frontier, const int32_t* __restrict__ values,
                         const int64_t* __restrict__ row_splits, uint32_t n_items,
                         int32_t* raw, unsigned char* scratch) {
  ExpandScratch* S = reinterpret_cast<ExpandScratch*>(scratch);
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  if (tid == 0) S->bad = 0;
  __syncthreads();
  uint32_t len = 0, start = 0;
  if (tid < n_frontier) {
    const int32_t node = frontier[tid];
    if ((uint32_t)node < n_items) {
      const int64_t s = row_splits[node], e = row_splits[node + 1];
      start = (uint32_t)s;
      len = (uint32_t)(e - s);
    } else {
      S->bad = 1;
    }
  }
  // inclusive scan inside the wavefront
  uint32_t inc = len;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) S->wave_tot[wave] = inc;
  __syncthreads();
  uint32_t wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kNW; ++w) {
    const uint32_t t = S->wave_tot[w];
    if (w < wave) wbase += t;
    total += t;
  }
  if (tid < n_frontier) {
    S->off[tid] = wbase + inc - len;
    S->rowstart[tid] = start;
  }
  if (tid == 0) S->off[n_frontier] = total;
  __syncthreads();
  const int bad = S->bad;
  if (!bad) {
    for (int r = wave; r < n_frontier; r += kNW) {
      const uint32_t o = S->off[r], l = S->off[r + 1] - o, s = S->rowstart[r];
      for (uint32_t c = lane; c < l; c += 64) raw[o + c] = values[(size_t)s + c];
    }
  }
  __syncthreads();
  return bad ? -1 : (int)total;
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
#pragma unroll
  for (int s = 1; s < LPR; s <<= 1) acc = acc + __shfl_xor(acc, s);
  return 0.0f - acc;
}

// wg_score_l2: scores[i] = -||q - table[ids[i]]||^2 for i < n.  All threads of
// the workgroup (NTHREADS = blockDim.x).  ids must be in range (the walker and
// index validation guarantee it on the fused path).  qv: f32[d] (LDS or global).
template <int LPR, int DT, int NTHREADS>
__device__ void wg_score_l2(const void* __restrict__ table, int d, const int32_t* ids, int n,
                            const float* qv, float* scores) {
  constexpr int U = 4;
  constexpr int GPW = 64 / LPR;              // rows per wavefront per load
  constexpr int RPI = (NTHREADS / 64) * GPW;  // rows per workgroup iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, grp = lane / LPR;
  float q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = qv[sub * 8 + k];
  for (int i0 = 0; i0 < n; i0 += RPI * U) {
    int32_t idv[U];
    RowChunk<DT> ch[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPI + wave * GPW + grp;
      idv[u] = (i < n) ? ids[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (idv[u] >= 0) ch[u] = load_chunk<DT>(table, (size_t)idv[u], d, sub);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * RPI + wave * GPW + grp;
      float x[8];
      if (idv[u] >= 0) {
        chunk_to_float<DT>(ch[u], x);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = 0.0f;
      }
      const float s = l2_finish<LPR>(q, x);
      if (sub == 0 && i < n) scores[i] = s;
    }
  }
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
struct TopkScratch {
  uint32_t cnt[40];
  uint32_t nsel;
  uint32_t wcnt[kNW];
  unsigned long long sel[kMaxK];
};

template <bool REG>
__device__ int wg_topk_impl(const int32_t* ids, const float* scores, int n, int k,
                            int32_t* out_pos, int32_t* out_ids, float* out_scores,
                            const int64_t* id_map, int64_t* out_mapped, unsigned char* scratch) {
  TopkScratch* S = reinterpret_cast<TopkScratch*>(scratch);
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  const uint64_t lt = lanemask_lt();
  uint32_t key[REG ? kTopkEPT : 1];
  if constexpr (REG) {
#pragma unroll
    for (int j = 0; j < kTopkEPT; ++j) {
      const int i = j * kNT + tid;
      key[j] = (i < n) ? score_key(scores[i]) : 0u;
    }
  }
  if (tid < 40) S->cnt[tid] = 0;
  if (tid == 0) S->nsel = 0;
  __syncthreads();

#define NANN_FOR_KEYS(...)                                                    \
  if constexpr (REG) {                                                        \
    _Pragma("unroll") for (int j = 0; j < kTopkEPT; ++j) {                    \
      if (j * kNT >= n) break;                                                \
      const int i = j * kNT + tid;                                            \
      const bool valid = i < n;                                               \
      const uint32_t kj = key[j];                                             \
      __VA_ARGS__                                                             \
    }                                                                         \
  } else {                                                                    \
    for (int i0 = 0; i0 < n; i0 += kNT) {                                     \
      const int i = i0 + tid;                                                 \
      const bool valid = i < n;                                               \
      const uint32_t kj = valid ? score_key(scores[i]) : 0u;                  \
      __VA_ARGS__                                                             \
    }                                                                         \
  }

  // ---- 2. threshold search ------------------------------------------------
  uint32_t T = 0, c_ge = (uint32_t)n;
  int it = 0;
  for (int bit = 31; bit >= 0; --bit, ++it) {
    const uint32_t probe = T | (1u << bit);
    uint32_t c = 0;
    NANN_FOR_KEYS({ c += (uint32_t)popc64(__ballot(valid && kj >= probe)); })
    if (lane == 0 && c) atomicAdd(&S->cnt[it], c);
    __syncthreads();
    const uint32_t tot = S->cnt[it];
    if (tot >= (uint32_t)k) {
      T = probe;
      c_ge = tot;
      if (tot == (uint32_t)k) break;
    }
  }
  // c_ge = #keys >= T >= k.  If c_ge > k the loop ran to bit 0 and T is the
  // exact k-th key: some (not all) keys equal to T are admitted.
  uint32_t c_gt = 0;
  const bool partial_eq = c_ge > (uint32_t)k;
  if (partial_eq) {
    uint32_t c = 0;
    NANN_FOR_KEYS({ c += (uint32_t)popc64(__ballot(valid && kj > T)); })
    if (lane == 0 && c) atomicAdd(&S->cnt[33], c);
    __syncthreads();
    c_gt = S->cnt[33];
    if (tid == 0) S->nsel = c_gt;  // equal keys take slots [c_gt, k)
    __syncthreads();
  }
  // ---- 3. collect -----------------------------------------------------------
  if (!partial_eq) {
    NANN_FOR_KEYS({
      const bool s = valid && kj >= T;
      const uint64_t m = __ballot(s);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&S->nsel, (uint32_t)popc64(m));
        b = __shfl(b, 0);
        if (s) S->sel[b + popc64(m & lt)] = ((unsigned long long)kj << 32) | (uint32_t)(~(uint32_t)i);
      }
    })
  } else {
    // keys > T: any slot in [0, c_gt) (unordered append through cnt[34])
    NANN_FOR_KEYS({
      const bool s = valid && kj > T;
      const uint64_t m = __ballot(s);
      if (m) {
        uint32_t b = 0;
        if (lane == 0) b = atomicAdd(&S->cnt[34], (uint32_t)popc64(m));
        b = __shfl(b, 0);
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
      for (int w = 0; w < kNW; ++w) {
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
  // ---- 4. rank sort + output --------------------------------------------------
  for (int e = tid; e < k; e += kNT) {
    const unsigned long long mine = S->sel[e];
    int rank = 0;
    for (int o = 0; o < k; ++o) rank += (S->sel[o] > mine) ? 1 : 0;
    const int pos = (int)(~(uint32_t)(mine & 0xffffffffull));
    const int32_t idv = ids ? ids[pos] : pos;
    if (out_pos) out_pos[rank] = pos;
    if (out_ids) out_ids[rank] = idv;
    if (out_scores) out_scores[rank] = scores[pos];
    if (out_mapped) out_mapped[rank] = id_map[idv];
  }
  __syncthreads();
  return 0;
}

__device__ int wg_topk(const int32_t* ids, const float* scores, int n, int k, int32_t* out_pos,
                       int32_t* out_ids, float* out_scores, const int64_t* id_map,
                       int64_t* out_mapped, unsigned char* scratch) {
  if (k < 0 || k > kMaxK) return 7;  // NANN_ERR_BAD_ARGUMENT
  if (n < k) return 4;               // NANN_ERR_TOPK_K_GT_N, topk_op.cc:67-71
  if (k == 0) return 0;
  if (n <= kTopkEPT * kNT)
    return wg_topk_impl<true>(ids, scores, n, k, out_pos, out_ids, out_scores, id_map, out_mapped, scratch);
  return wg_topk_impl<false>(ids, scores, n, k, out_pos, out_ids, out_scores, id_map, out_mapped, scratch);
}

}  // namespace nann
