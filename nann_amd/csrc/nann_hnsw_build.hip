// nann_hnsw_build.hip -- HNSW index construction ON THE DEVICE (SURVEY.md 8 f1 follow-up).
//
// The reference builds its index with faiss.IndexHNSWFlat(d, 32).add(embeddings) on the host
// (NANN_impls/nann/delivery/build_hnsw_index.py:33-35); nann_amd/csrc/host/hnsw_build.cpp restates that algorithm
// (Malkov & Yashunin alg. 1-4 with Faiss' conventions) multi-threaded on the CPU: 16 s for 1M x 128-d on 16 cores,
// minutes for a 4M x 256-d shard -- all outside the GPU, and the longest part of every test / bench set-up.
// Same algorithm here, BATCHED: nodes go in by descending level (as Faiss adds them), in batches that grow with the
// graph (a batch never exceeds a quarter of what is already inserted, <= 16384); the nodes of a batch
//   1. descend greedily through the levels above their own (ef = 1),
//   2. on each of their levels run the ef_construction beam search on the graph of the EARLIER batches (batch
//      mates do not see each other: with random insertion order a node's true neighbours are in its own batch with
//      probability batch / inserted), pick their links with the selection heuristic (alg. 4, no
//      keepPrunedConnections) and write their own adjacency rows,
//   3. hand (dst, src) pairs to the back-link pass: pairs sorted by dst (radix sort, stable, so the result does
//      not depend on atomics or scheduling), one wavefront per dst appends the new links while there is room
//      and re-selects with the heuristic when the row is full -- one incoming link at a time, as Faiss' add_link.
// Every step is one kernel over the whole batch, one WAVEFRONT per node: the beam (up to 64 sorted (distance, id)
// candidates) lives in the lanes of the wavefront -- lane i holds the i-th best --, is merged with a node's new
// neighbours by a bitonic network on shuffles, the visited set is a per-wavefront hash table in LDS, row distances
// are computed d/8 lanes per row from 16-byte loads.  A search is a chain of dependent HBM round trips (~50 hops of
// ~2 us), so a wavefront is slow and the chip is filled by the batch: thousands of searches in flight.
//
// Output = the arrays the exporter needs, on the device: levels (Faiss convention: number of levels of the node),
// adj0 [N, 2M] (-1 = empty slot), and for nodes with more than one level their rows in adj_up [rows, M]
// (row of node i on level l >= 1: up_row[i] + l - 1).  nann_amd/index_build.py turns them into
// neighbors_level_{l}_{values,row_splits}.npy / enter_points.npy exactly as build_hnsw_index.py:41-66 does.
// Index CONTENTS are not a parity target (Faiss' own insertion order is thread-schedule dependent);
// the structural invariants and recall are (tests/test_index_build.py, -m gpu).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/nann_hip.h"
#include "nann_device.h"

namespace nann {
int fail(int code, const std::string& msg);  // nann_hip.hip
}
using namespace nann;

namespace {

#define HB_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fail(NANN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

constexpr int kHbWaves = 4;          // wavefronts (= nodes) per workgroup
constexpr int kHbVisSlots = 4096;    // visited-set slots per wavefront (16 KB): ef_construction x degree ids at most, kept < 3/4 full
constexpr uint32_t kHbEmpty = 0xffffffffu;
constexpr int kHbMaxCand = 64;       // beam slots = lanes
constexpr unsigned long long kKeyInf = 0xffffffffffffffffull;

struct HbGraph {
  const void* emb;      // [N, d] f16 / bf16 rows
  int32_t* adj0;        // [N, cap0]
  int32_t* cnt0;        // [N]
  const int32_t* up_row;  // [N] first upper row or -1
  int32_t* adj_up;      // [rows, M]
  int32_t* cnt_up;      // [rows]
  int n_items, d, M;
  int keep_pruned;      // alg. 4's keepPrunedConnections: fill a row's free slots with the nearest candidates the heuristic discarded
};

__device__ __forceinline__ int32_t* hb_row(const HbGraph& g, int node, int level, int* cap, int32_t** cnt) {
  if (level == 0) { *cap = 2 * g.M; *cnt = g.cnt0 + node; return g.adj0 + (size_t)node * 2 * g.M; }
  const size_t r = (size_t)g.up_row[node] + (size_t)(level - 1);
  *cap = g.M; *cnt = g.cnt_up + r;
  return g.adj_up + r * g.M;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}

// ascending bitonic sort of one 64-bit key per lane over the wavefront
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long key, int lane) {
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const unsigned long long other = shfl_xor_u64(key, stride);
      const bool up = (lane & size) == 0;          // ascending block
      const bool lower = (lane & stride) == 0;     // this lane keeps the smaller one in an ascending block
      const bool take_min = up == lower;
      key = take_min ? (key < other ? key : other) : (key > other ? key : other);
    }
  }
  return key;
}

// merge: `sorted` ascending over the lanes, `fresh` arbitrary -> the 64 smallest of the 128, ascending
__device__ __forceinline__ unsigned long long wave_merge64(unsigned long long sorted, unsigned long long fresh, int lane) {
  fresh = wave_sort64(fresh, lane);
  const unsigned long long rev = shfl_xor_u64(fresh, 63);  // descending
  unsigned long long key = sorted < rev ? sorted : rev;     // bitonic: the 64 smallest
#pragma unroll
  for (int stride = 32; stride > 0; stride >>= 1) {
    const unsigned long long other = shfl_xor_u64(key, stride);
    const bool lower = (lane & stride) == 0;
    key = lower ? (key < other ? key : other) : (key > other ? key : other);
  }
  return key;
}

// 8 consecutive elements of row `r` (sub-chunk `sub`) as f32; DT: 0 f16, 1 bf16
template <int DT>
__device__ __forceinline__ void hb_load8(const void* emb, size_t r, int d, int sub, float x[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(emb) + r * d + sub * 8);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (DT == 0) { x[2 * i] = half_bits_to_float(w[i] & 0xffffu); x[2 * i + 1] = half_bits_to_float(w[i] >> 16); }
    else { x[2 * i] = bf16_bits_to_float(w[i] & 0xffffu); x[2 * i + 1] = bf16_bits_to_float(w[i] >> 16); }
  }
}

// squared L2 over the LPR lanes of a row group: every lane of the group returns the total
template <int LPR>
__device__ __forceinline__ float hb_group_sum(float acc) {
#pragma unroll
  for (int s = 1; s < LPR; s <<= 1) acc += __shfl_xor(acc, s);
  return acc;
}

// distance key: squared distances are >= 0, so their f32 bit patterns order like the values
__device__ __forceinline__ unsigned long long hb_key(float dist, int id) {
  return ((unsigned long long)__float_as_uint(dist + 0.0f) << 32) | (uint32_t)id;
}

constexpr int kHbMaxEf = 40;  // ef_construction this build stages rows for (Faiss' default)
template <int D>
struct HbWaveLds {
  static constexpr int kStage = kHbMaxEf * D * 2;  // bytes of row staging for the selection among the beam
  static constexpr int kRegion = kStage > kHbVisSlots * 4 ? kStage : kHbVisSlots * 4;
  union {
    uint32_t vis[kHbVisSlots];           // visited ids (search) ...
    unsigned char stage[kRegion];        // ... reused as row staging by the selection once the search is over
  };
  int32_t list[kHbMaxCand];   // ids of a node's unvisited neighbours (compacted)
  float dist[kHbMaxCand];
};

// the selection heuristic (alg. 4 / Faiss shrink_neighbor_list) over candidates sorted ascending by distance to the
// base: keep c iff it is closer to the base than to every candidate kept so far; at most `cap`.  cand_key: lane i =
// i-th nearest (kKeyInf beyond n).  Rows are staged into `rows` (LDS, n x d halves).  Returns the keep mask
// (bit i = candidate i kept; wave-uniform).
template <int LPR, int DT>
__device__ __forceinline__ uint64_t wave_select(const HbGraph& g, unsigned long long cand_key, int n, int cap,
                                                uint16_t* rows, int lane) {
  constexpr int GPW = 64 / LPR;
  const int d = g.d, sub = lane % LPR, grp = lane / LPR;
  // stage the candidates' rows: group `grp` copies rows grp, grp + GPW, ...
  for (int i0 = 0; i0 < n; i0 += GPW) {
    const int i = i0 + grp;
    const int id = (int)(uint32_t)__shfl((int)(uint32_t)cand_key, min(i, n - 1));
    if (i < n)
      *reinterpret_cast<uint4*>(rows + (size_t)i * d + sub * 8) =
          *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(g.emb) + (size_t)id * d + sub * 8);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  uint64_t kept = 0;
  int n_kept = 0;
  for (int c = 0; c < n && n_kept < cap; ++c) {
    const float dc = __uint_as_float((uint32_t)__shfl((int)(uint32_t)(cand_key >> 32), c));  // distance of c to the base
    bool dominated = false;
    if (n_kept > 0) {
      // distances of c to the kept candidates, GPW pairs per step
      float xc[8];
      {
        const uint4 v = *reinterpret_cast<const uint4*>(rows + (size_t)c * d + sub * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (DT == 0) { xc[2 * i] = half_bits_to_float(w[i] & 0xffffu); xc[2 * i + 1] = half_bits_to_float(w[i] >> 16); }
          else { xc[2 * i] = bf16_bits_to_float(w[i] & 0xffffu); xc[2 * i + 1] = bf16_bits_to_float(w[i] >> 16); }
        }
      }
      uint64_t todo = kept;
      while (todo && !dominated) {  // uniform
        // the grp-th set bit of `todo`
        uint64_t t = todo;
        for (int k = 0; k < grp && t; ++k) t &= t - 1;
        const int s = t ? __ffsll((unsigned long long)t) - 1 : -1;
        float acc = 0.0f;
        if (s >= 0) {
          const uint4 v = *reinterpret_cast<const uint4*>(rows + (size_t)s * d + sub * 8);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float a, b;
            if (DT == 0) { a = half_bits_to_float(w[i] & 0xffffu); b = half_bits_to_float(w[i] >> 16); }
            else { a = bf16_bits_to_float(w[i] & 0xffffu); b = bf16_bits_to_float(w[i] >> 16); }
            const float t0 = xc[2 * i] - a, t1 = xc[2 * i + 1] - b;
            acc = __fmaf_rn(t0, t0, acc);
            acc = __fmaf_rn(t1, t1, acc);
          }
        }
        acc = hb_group_sum<LPR>(acc);
        dominated = __ballot(s >= 0 && acc < dc) != 0ull;
        for (int k = 0; k < GPW && todo; ++k) todo &= todo - 1;  // the GPW bits just processed
      }
    }
    if (!dominated) { kept |= 1ull << c; ++n_kept; }
  }
  if (g.keep_pruned && n_kept < cap) {
    // keepPrunedConnections (Malkov & Yashunin alg. 4, lines 15-17): the discarded candidates, nearest first, until the row
    // is full.  Faiss leaves this off (so do the reference's graphs and this builder's default); with it rows fill up to the
    // cap -- the dense-graph case of SURVEY.md 8's gather bound (L0 gathered <= ef * 64).
    const int n_all = min(n, 64);
    uint64_t rest = ~kept & (n_all >= 64 ? ~0ull : ((1ull << n_all) - 1ull));
    while (rest && n_kept < cap) {
      const int c = __ffsll((unsigned long long)rest) - 1;
      kept |= 1ull << c;
      rest &= rest - 1;
      ++n_kept;
    }
  }
  return kept;
}

struct HbBatch {
  const int32_t* nodes;   // [n] node ids of the batch
  int n;
  int level;              // the level searched
  int ef;                 // beam width (1 on the levels above the batch's own)
  int link;               // 1: select + write the node's row + emit back-link pairs
  int32_t* entry;         // [n] in: entry node of this level; out: nearest node found (entry of the next level down)
  uint32_t* pair_dst;     // [n * pair_cap] back-link targets (kHbEmpty = none)
  uint32_t* pair_src;     // [n * pair_cap]
  int pair_cap;
};

template <int LPR, int DT>
__global__ __launch_bounds__(kHbWaves * 64) void k_hb_search(HbGraph g, HbBatch b) {
  __shared__ __attribute__((aligned(16))) HbWaveLds<LPR * 8> lds_all[kHbWaves];
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bi = blockIdx.x * kHbWaves + wave;
  if (bi >= b.n) return;
  HbWaveLds<LPR * 8>& L = lds_all[wave];
  const int u = b.nodes[bi];
  const int sub = lane % LPR, grp = lane / LPR;
  float q[8];
  hb_load8<DT>(g.emb, (size_t)u, g.d, sub, q);
  for (int i = lane; i < kHbVisSlots; i += 64) L.vis[i] = kHbEmpty;
  auto row_dist = [&](int id) -> float {  // all lanes of a group pass the same id
    float x[8];
    hb_load8<DT>(g.emb, (size_t)id, g.d, sub, x);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float t = q[k] - x[k]; acc = __fmaf_rn(t, t, acc); }
    return hb_group_sum<LPR>(acc);
  };
  auto visit = [&](int id) -> bool {  // true if the id was not in the set (and is now)
    uint32_t h = ((uint32_t)id * 2654435761u) >> 20;  // 12 bits
    for (int probe = 0; probe < kHbVisSlots; ++probe) {
      const uint32_t c = atomicCAS(&L.vis[h], kHbEmpty, (uint32_t)id);
      if (c == kHbEmpty) return true;
      if (c == (uint32_t)id) return false;
      h = (h + 1) & (kHbVisSlots - 1);
    }
    return false;
  };
  __builtin_amdgcn_wave_barrier();
  const int ep = b.entry[bi];
  const float d_ep = row_dist(ep);
  if (lane == 0) visit(ep);
  int n_vis = 1;
  // the beam: lane i = i-th best (distance, id); low bit of `expanded` per lane
  unsigned long long beam = lane == 0 ? hb_key(d_ep, ep) : kKeyInf;
  bool expanded = false;
  const int ef = b.ef;
  for (;;) {
    const uint64_t open = __ballot(!expanded && beam != kKeyInf && lane < ef);
    if (!open) break;
    const int p = __ffsll((unsigned long long)open) - 1;
    if (lane == p) expanded = true;
    const int cur = (int)(uint32_t)__shfl((int)(uint32_t)beam, p);
    int cap;
    int32_t* cnt;
    const int32_t* row = hb_row(g, cur, b.level, &cap, &cnt);
    const int n_nb = min(*cnt, cap);
    // unvisited neighbours, compacted into L.list
    int nb = lane < n_nb ? row[lane] : -1;
    bool fresh = false;
    if (nb >= 0 && n_vis < (kHbVisSlots * 3) / 4) fresh = visit(nb);
    const uint64_t fm = __ballot(fresh);
    const int n_new = popc64(fm);
    if (fresh) L.list[popc64(fm & lanemask_lt(lane))] = nb;
    n_vis += n_new;
    __builtin_amdgcn_wave_barrier();
    // distances, GPW rows per step
    for (int i0 = 0; i0 < n_new; i0 += GPW) {
      const int i = i0 + grp;
      const int id = L.list[min(i, n_new - 1)];
      const float dv = row_dist(id);
      if (i < n_new && sub == 0) L.dist[i] = dv;
    }
    __builtin_amdgcn_wave_barrier();
    unsigned long long fresh_key = kKeyInf;
    if (lane < n_new) fresh_key = hb_key(L.dist[lane], L.list[lane]);
    // merge; a lane's `expanded` flag travels with its key: carry it in the key's bit 31 (ids are < 2^31)
    const unsigned long long tagged = beam == kKeyInf ? beam : (beam | (expanded ? 0x80000000ull : 0ull));
    const unsigned long long merged = wave_merge64(tagged, fresh_key, lane);
    expanded = merged != kKeyInf && (merged & 0x80000000ull) != 0ull;
    beam = merged == kKeyInf ? merged : (merged & ~0x80000000ull);
    if (lane >= ef) { beam = kKeyInf; expanded = false; }  // the beam holds ef candidates
    __builtin_amdgcn_wave_barrier();
  }
  // nearest node found: the next level's entry
  const int nearest = (int)(uint32_t)__shfl((int)(uint32_t)beam, 0);
  if (lane == 0) b.entry[bi] = nearest;
  if (!b.link) return;
  // ---- link: heuristic selection among the beam, own row, back-link pairs
  const int n_cand = popc64(__ballot(beam != kKeyInf));
  int cap;
  int32_t* cnt;
  int32_t* my_row = hb_row(g, u, b.level, &cap, &cnt);
  uint16_t* rows = reinterpret_cast<uint16_t*>(L.stage);  // the search is over: its table becomes row staging
  const uint64_t kept = wave_select<LPR, DT>(g, beam, n_cand, cap, rows, lane);
  const int n_kept = popc64(kept);
  const bool mine = (kept >> lane) & 1ull;
  const int pos = popc64(kept & lanemask_lt(lane));
  const int id = (int)(uint32_t)beam;
  if (mine) {
    my_row[pos] = id;
    b.pair_dst[(size_t)bi * b.pair_cap + pos] = (uint32_t)id;
    b.pair_src[(size_t)bi * b.pair_cap + pos] = (uint32_t)u;
  }
  for (int i = n_kept + lane; i < b.pair_cap; i += 64) b.pair_dst[(size_t)bi * b.pair_cap + i] = kHbEmpty;
  if (lane == 0) *cnt = n_kept;
}

// run starts of the sorted pair list -> (dst, first, count) records
__global__ void k_hb_runs(const uint32_t* dst, int n_pairs, int32_t* run_first, int32_t* run_len, int* n_runs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  const uint32_t k = dst[i];
  if (k == kHbEmpty) return;
  if (i > 0 && dst[i - 1] == k) return;
  int len = 1;
  while (i + len < n_pairs && dst[i + len] == k) ++len;
  const int r = atomicAdd(n_runs, 1);
  run_first[r] = i;
  run_len[r] = len;
}

// back-links of one batch: one wavefront per target node; its incoming sources in pair order (deterministic: the
// pair buffer is filled by batch position and the sort is stable).  Room left: append.  Full: re-select among the
// row + the new link (Faiss add_link), one incoming link at a time.  The row lives in the lanes (lane i = link i)
// and is written back once.
template <int LPR, int DT>
__global__ __launch_bounds__(kHbWaves * 64) void k_hb_backlink(HbGraph g, int level, const uint32_t* pair_dst,
                                                               const uint32_t* pair_src, const int32_t* run_first,
                                                               const int32_t* run_len, const int* n_runs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int GPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ri = blockIdx.x * kHbWaves + wave;
  if (ri >= *n_runs) return;
  const int sub = lane % LPR, grp = lane / LPR;
  const size_t row_bytes = (size_t)kHbMaxCand * g.d * 2;
  unsigned char* mine_lds = smem + (size_t)wave * (row_bytes + 1024);
  uint16_t* rows = reinterpret_cast<uint16_t*>(mine_lds);
  float* dist = reinterpret_cast<float*>(mine_lds + row_bytes);       // [64]
  int32_t* ids = reinterpret_cast<int32_t*>(mine_lds + row_bytes + 512);  // [64]
  const int first = run_first[ri], len = run_len[ri];
  const int v = (int)pair_dst[first];
  int cap;
  int32_t* cnt;
  int32_t* row = hb_row(g, v, level, &cap, &cnt);
  int c = min(*cnt, cap);
  int link = lane < c ? row[lane] : -1;  // the row, one link per lane
  float qv[8];
  hb_load8<DT>(g.emb, (size_t)v, g.d, sub, qv);
  for (int j = 0; j < len; ++j) {
    const int s = (int)pair_src[first + j];
    if (c < cap) {
      if (lane == c) link = s;
      ++c;
      continue;
    }
    // full: candidates = the cap links + s (at cap = 64 the 65 do not fit the lanes: the farthest of them is
    // dropped before the walk -- it could only have been kept if fewer than cap nearer ones survive).  Distances to v:
    ids[lane] = lane < cap ? link : -1;
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 <= cap; i0 += GPW) {
      const int i = i0 + grp;
      const int id = i < cap ? ids[min(i, cap - 1)] : s;
      float x[8];
      hb_load8<DT>(g.emb, (size_t)id, g.d, sub, x);
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float t = qv[k] - x[k]; acc = __fmaf_rn(t, t, acc); }
      acc = hb_group_sum<LPR>(acc);
      if (i < cap && sub == 0) dist[i] = acc;
      if (i == cap && sub == 0) dist[63] = cap == 64 ? dist[63] : acc;  // (placeholder slot; the extra's distance is re-derived below)
    }
    __builtin_amdgcn_wave_barrier();
    float d_extra;
    {  // distance of the new link, computed by group 0 and broadcast
      float x[8];
      hb_load8<DT>(g.emb, (size_t)s, g.d, sub, x);
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float t = qv[k] - x[k]; acc = __fmaf_rn(t, t, acc); }
      d_extra = hb_group_sum<LPR>(acc);
    }
    unsigned long long key = lane < cap ? hb_key(dist[lane], link) : kKeyInf;
    key = wave_sort64(key, lane);
    const unsigned long long extra = hb_key(d_extra, s);
    const int before = popc64(__ballot(key < extra));  // the extra candidate's place in the ascending order
    const int src_lane = lane <= before ? lane : lane - 1;
    const unsigned long long from = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(key >> 32), src_lane) << 32) |
                                    (uint32_t)__shfl((int)(uint32_t)key, src_lane);
    unsigned long long cand = lane == before ? extra : from;
    const int n_cand = min(cap + 1, 64);
    if (lane >= n_cand) cand = kKeyInf;
    const uint64_t kept = wave_select<LPR, DT>(g, cand, n_cand, cap, rows, lane);
    const int n_kept = popc64(kept);
    __builtin_amdgcn_wave_barrier();
    if ((kept >> lane) & 1ull) ids[popc64(kept & lanemask_lt(lane))] = (int)(uint32_t)cand;
    __builtin_amdgcn_wave_barrier();
    link = lane < n_kept ? ids[lane] : -1;
    c = n_kept;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane < cap) row[lane] = lane < c ? link : -1;
  if (lane == 0) *cnt = c;
}

template <typename T>
int dev_alloc(T** p, size_t n, std::vector<void*>* owned) {
  void* q = nullptr;
  HB_TRY(hipMalloc(&q, std::max<size_t>(n * sizeof(T), 16)));
  owned->push_back(q);
  *p = static_cast<T*>(q);
  return NANN_OK;
}

template <int LPR, int DT>
int run_batch_level(const HbGraph& g, const HbBatch& b, hipStream_t st) {
  const unsigned blocks = (unsigned)((b.n + kHbWaves - 1) / kHbWaves);
  hipLaunchKernelGGL((k_hb_search<LPR, DT>), dim3(blocks), dim3(kHbWaves * 64), 0, st, g, b);
  HB_TRY(hipGetLastError());
  return NANN_OK;
}

template <int LPR, int DT>
int run_backlink(const HbGraph& g, int level, const uint32_t* dst, const uint32_t* src, const int32_t* rf, const int32_t* rl,
                 const int* n_runs, int max_runs, hipStream_t st) {
  const size_t per_wave = (size_t)kHbMaxCand * g.d * 2 + 1024;
  const size_t lds = per_wave * kHbWaves;
  auto kern = k_hb_backlink<LPR, DT>;
  if (lds > 48 * 1024)
    HB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned blocks = (unsigned)((max_runs + kHbWaves - 1) / kHbWaves);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kHbWaves * 64), lds, st, g, level, dst, src, rf, rl, n_runs);
  HB_TRY(hipGetLastError());
  return NANN_OK;
}

}  // namespace

extern "C" {

int nann_hnsw_draw_levels(int64_t n_items, int32_t M, uint64_t seed, int32_t* levels, int64_t* n_up_rows) {
  if (n_items <= 0 || M < 2 || !levels) return fail(NANN_ERR_BAD_ARGUMENT, "nann_hnsw_draw_levels: bad argument");
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  const double mult = 1.0 / std::log((double)M);
  int64_t up = 0;
  for (int64_t i = 0; i < n_items; ++i) {
    double u = uni(rng);
    if (u < 1e-300) u = 1e-300;
    const int lv = (int)std::floor(-std::log(u) * mult) + 1;  // Faiss convention: number of levels
    levels[i] = lv;
    up += lv - 1;
  }
  if (n_up_rows) *n_up_rows = up;
  return NANN_OK;
}

int nann_hnsw_build_device(const void* item_embs, int64_t n_items, int32_t d, int32_t emb_dtype, int32_t M,
                           int32_t ef_construction, const int32_t* levels, int32_t* adj0, int32_t* up_row,
                           int32_t* adj_up, nann_stream_t stream) {
  return nann_hnsw_build_device_ex(item_embs, n_items, d, emb_dtype, M, ef_construction, 0, levels, adj0, up_row, adj_up, stream);
}

int nann_hnsw_build_device_ex(const void* item_embs, int64_t n_items, int32_t d, int32_t emb_dtype, int32_t M,
                              int32_t ef_construction, int32_t keep_pruned, const int32_t* levels, int32_t* adj0,
                              int32_t* up_row, int32_t* adj_up, nann_stream_t stream) {
  if (!item_embs || !levels || !adj0 || !up_row || n_items <= 0)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_hnsw_build_device: null argument");
  if (n_items > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "nann_hnsw_build_device: more than 2^31 - 1 items");
  if (!(d == 64 || d == 128 || d == 256)) return fail(NANN_ERR_UNSUPPORTED, "nann_hnsw_build_device: d must be 64, 128 or 256");
  if (emb_dtype != NANN_F16 && emb_dtype != NANN_BF16) return fail(NANN_ERR_UNSUPPORTED, "nann_hnsw_build_device: f16 or bf16 rows");
  if (M < 2 || 2 * M > kHbMaxCand) return fail(NANN_ERR_UNSUPPORTED, "nann_hnsw_build_device: 2 <= M <= 32");
  const int ef = ef_construction > 0 ? ef_construction : 40;
  if (ef > kHbMaxEf) return fail(NANN_ERR_UNSUPPORTED, "nann_hnsw_build_device: ef_construction <= 40 in this build");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int N = (int)n_items;
  // ---- insertion order: top level first, ascending id inside a level (Faiss adds the highest levels first)
  std::vector<int32_t> order((size_t)N), h_up((size_t)N);
  int max_lv = 1;
  int64_t up_rows = 0;
  for (int i = 0; i < N; ++i) {
    order[(size_t)i] = i;
    if (levels[i] < 1) return fail(NANN_ERR_BAD_ARGUMENT, "nann_hnsw_build_device: levels must be >= 1");
    max_lv = std::max(max_lv, levels[i]);
    h_up[(size_t)i] = levels[i] > 1 ? (int32_t)up_rows : -1;
    up_rows += levels[i] - 1;
  }
  if (up_rows > 0 && !adj_up) return fail(NANN_ERR_BAD_ARGUMENT, "nann_hnsw_build_device: adj_up is null");
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return levels[a] > levels[b]; });

  std::vector<void*> owned;
  auto cleanup = [&](int rc) { for (void* p : owned) (void)hipFree(p); return rc; };
  int32_t *d_order, *d_cnt0, *d_cnt_up, *d_entry, *d_run_first, *d_run_len;
  uint32_t *d_pair_dst, *d_pair_src, *d_pair_dst2, *d_pair_src2;
  int* d_n_runs;
  const int max_batch = 16384, pair_cap = 2 * M;
  int rc = dev_alloc(&d_order, (size_t)N, &owned);
  if (!rc) rc = dev_alloc(&d_cnt0, (size_t)N, &owned);
  if (!rc) rc = dev_alloc(&d_cnt_up, (size_t)std::max<int64_t>(up_rows, 1), &owned);
  if (!rc) rc = dev_alloc(&d_entry, (size_t)max_batch, &owned);
  if (!rc) rc = dev_alloc(&d_pair_dst, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_pair_src, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_pair_dst2, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_pair_src2, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_run_first, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_run_len, (size_t)max_batch * pair_cap, &owned);
  if (!rc) rc = dev_alloc(&d_n_runs, 1, &owned);
  if (rc) return cleanup(rc);
  size_t sort_bytes = 0;
  int key_bits = 1;
  while ((1ll << key_bits) < n_items) ++key_bits;
  hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_pair_dst, d_pair_dst2, d_pair_src, d_pair_src2, max_batch * pair_cap, 0, 32, st);
  void* d_sort = nullptr;
  {
    unsigned char* p;
    rc = dev_alloc(&p, sort_bytes, &owned);
    if (rc) return cleanup(rc);
    d_sort = p;
  }
  auto hip_rc = [&](hipError_t e, const char* what) { return e == hipSuccess ? NANN_OK : fail(NANN_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
  if ((rc = hip_rc(hipMemcpyAsync(d_order, order.data(), (size_t)N * 4, hipMemcpyHostToDevice, st), "order")) ||
      (rc = hip_rc(hipMemcpyAsync(up_row, h_up.data(), (size_t)N * 4, hipMemcpyHostToDevice, st), "up_row")) ||
      (rc = hip_rc(hipMemsetAsync(adj0, 0xff, (size_t)N * 2 * M * 4, st), "adj0")) ||
      (rc = hip_rc(hipMemsetAsync(d_cnt0, 0, (size_t)N * 4, st), "cnt0")) ||
      (rc = hip_rc(hipMemsetAsync(d_cnt_up, 0, (size_t)std::max<int64_t>(up_rows, 1) * 4, st), "cnt_up")))
    return cleanup(rc);
  if (up_rows > 0 && (rc = hip_rc(hipMemsetAsync(adj_up, 0xff, (size_t)up_rows * M * 4, st), "adj_up"))) return cleanup(rc);

  HbGraph g;
  g.emb = item_embs; g.adj0 = adj0; g.cnt0 = d_cnt0; g.up_row = up_row; g.adj_up = adj_up; g.cnt_up = d_cnt_up;
  g.n_items = N; g.d = d; g.M = M; g.keep_pruned = keep_pruned ? 1 : 0;
  const int global_entry = order[0], top = levels[order[0]] - 1;  // level index of the entry point
  std::vector<int32_t> h_entry((size_t)max_batch, global_entry);

  auto search = [&](const HbBatch& b) -> int {
#define HB_CASE(LPR_)                                                                \
  return emb_dtype == NANN_F16 ? run_batch_level<LPR_, 0>(g, b, st) : run_batch_level<LPR_, 1>(g, b, st)
    if (d == 64) HB_CASE(8);
    if (d == 128) HB_CASE(16);
    HB_CASE(32);
#undef HB_CASE
  };
  auto backlink = [&](int level, int n_pairs) -> int {
#define HB_CASE(LPR_)                                                                                                        \
  return emb_dtype == NANN_F16 ? run_backlink<LPR_, 0>(g, level, d_pair_dst2, d_pair_src2, d_run_first, d_run_len, d_n_runs, n_pairs, st) \
                               : run_backlink<LPR_, 1>(g, level, d_pair_dst2, d_pair_src2, d_run_first, d_run_len, d_n_runs, n_pairs, st)
    if (d == 64) HB_CASE(8);
    if (d == 128) HB_CASE(16);
    HB_CASE(32);
#undef HB_CASE
  };

  // ---- batches: same top level inside a batch; a batch is at most a quarter of what is already in the graph
  int pos = 1;  // order[0] is the entry point: inserted with no links
  while (pos < N) {
    const int lv = levels[order[(size_t)pos]];  // number of levels of the batch's nodes
    int end = pos;
    const int limit = std::min(max_batch, std::max(1, pos / 4));
    while (end < N && end - pos < limit && levels[order[(size_t)end]] == lv) ++end;
    const int n = end - pos;
    if ((rc = hip_rc(hipMemcpyAsync(d_entry, h_entry.data(), (size_t)n * 4, hipMemcpyHostToDevice, st), "entry"))) return cleanup(rc);
    for (int level = top; level >= 0; --level) {
      HbBatch b;
      b.nodes = d_order + pos; b.n = n; b.level = level;
      b.link = level <= lv - 1;
      b.ef = b.link ? ef : 1;
      b.entry = d_entry; b.pair_dst = d_pair_dst; b.pair_src = d_pair_src; b.pair_cap = pair_cap;
      if ((rc = search(b))) return cleanup(rc);
      if (!b.link) continue;
      const int n_pairs = n * pair_cap;
      if (hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, d_pair_dst, d_pair_dst2, d_pair_src, d_pair_src2, n_pairs, 0, 32, st) != hipSuccess)
        return cleanup(fail(NANN_ERR_HIP, "nann_hnsw_build_device: radix sort failed"));
      if ((rc = hip_rc(hipMemsetAsync(d_n_runs, 0, 4, st), "n_runs"))) return cleanup(rc);
      hipLaunchKernelGGL(k_hb_runs, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, d_pair_dst2, n_pairs, d_run_first, d_run_len, d_n_runs);
      if ((rc = backlink(level, n_pairs))) return cleanup(rc);
    }
    pos = end;
  }
  rc = hip_rc(hipStreamSynchronize(st), "nann_hnsw_build_device");
  (void)key_bits;
  return cleanup(rc);
}

}  // extern "C"
