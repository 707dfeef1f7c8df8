// nann_mlp6.h -- the MLP traversal as a PIPELINE OF PHASES (round 4): scoring in a kernel of its own.
//
// The fused traversal with the MLP scorer (nann_mlp5.h inside k_search) keeps the matrix pipe 0.41 busy: 0.54 inside the
// scoring calls, and 18 % of a query -- expand, top-k -- with the pipe idle, because the scorer's 256 registers and
// 137 KB of LDS allow ONE workgroup per CU, so nothing runs beside a query's latency-bound phases.  Both halves do better
// apart:
//   * the traversal stages (top-k of the last round, marks, expand of the next: search_one between two scoring calls) are
//     the L2 kernel's code at its occupancy -- two 512-thread workgroups per CU, 64 KB set + 11 KB scratch each -- so
//     one query's dependent HBM / LDS trips hide behind another's;
//   * the scoring of ALL queries' candidate lists of a round is one launch of k_mlp_phase_score: W2 loaded into LDS once
//     per workgroup for the whole launch (no per-call reload, no set to park), the 32-row blocks of every query laid end
//     to end and cut into 2048 equal runs, one per wavefront -- no ragged last blocks per call, no barrier anywhere.
// Per 1024-query chunk: 6 traversal launches and 5 scoring launches (each works out the blocks of every query -> their
// prefix sums for itself); state between launches lives in the query's slot (PhaseState, the candidate arrays the
// fused kernel uses anyway, the set parked in the slot's bitmap region around rounds 2 and 3).  Results are what the
// fused kernel computes: the same building blocks run on the same lists (exact precision: bit-identical to the oracle).
#pragma once
#include "nann_mlp5.h"

namespace nann {

constexpr int kPhaseChunk = 1024;     // queries per pipeline pass (one prefix workgroup, bounded workspace)
constexpr int kPhasePending = -100;   // PhaseState.status while a query waits for its scores
constexpr int kPhaseSkip = -101;      // search_one's return for a query an earlier stage finished (never stored)
constexpr int kPhaseScoreWaves = 2048;  // 256 workgroups x 8 wavefronts: runs of the scoring launch

// what a query carries from one launch to the next (in its slot)
struct PhaseState {
  int status;     // kPhasePending | final nann_status
  int r;          // the round whose scoring call is pending
  int sc_n;       // rows of that call
  int base_off;   // they are cand_ids[base_off ..] (r == 0: the index's enter points), scores to cand_scores[base_off ..]
  int nP;         // pool size so far
  int vis_count;  // ids in the parked set
  int ctr[3 * NANN_NUM_ROUNDS];
  int pad[11];
  float u[256];   // b1 + W1q^T q: the query's part of layer 1, computed once (stage 0)
};
static_assert(sizeof(PhaseState) == 128 + 1024, "PhaseState layout");

struct PhaseScoreArgs {
  unsigned char* ws;            // the search workspace: [header | slots]
  unsigned long long slot_bytes;
  unsigned long long off_cand_ids, off_cand_scores, off_state;  // within a slot
  const int32_t* enter;
  const float* proj;            // the pre-projected table
  uint32_t n_items;
  int n_queries;
  int round;
  int dry;                      // timing launches (NANN_PHASE_SHADOW, tools/): everything but the store of the scores
  MlpParams mlp;
};

// timing builds (tools/build_res_variant.py -DNANN_PHASE_VAR=bits): the split-f16 scoring launch WITHOUT 1 = its gathers,
// 2 = its W2 fragment reads, 4 = the PReLU / split arithmetic, 8 = the MFMAs; 16 = WITH the PReLU decomposition's cost (one v_max per
// element, 512 more gathered bytes per row: nann_mlp5.h) -- run as a second, dry launch behind the real
// one (NANN_PHASE_SHADOW=1), so that both see the same lists
#ifndef NANN_PHASE_VAR
#define NANN_PHASE_VAR 0
#endif
#ifndef NANN_PHASE_PACKED_EPI
#define NANN_PHASE_PACKED_EPI 0  // 1: the scoring launch's output layer as two packed dot products (2 instead of 3 instructions per unit;
                                 // measured 0.7 % slower, and other bits than the fused kernel's chain: profiles/r4x_*)
#endif

// One launch scores the pending candidate lists of every query of the chunk.  256 workgroups x 8 wavefronts; wavefront
// gw takes blocks [gw T / 2048, (gw + 1) T / 2048) of the T blocks laid end to end.  EXACT: f32 MFMA on the table
// (wg_score_mlp_xres's arithmetic), else split-f16 (wg_score_mlp_res's).
template <bool EXACT, int VAR = 0>
__global__ __launch_bounds__(512, 2) void k_mlp_phase_score(PhaseScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int H1T = 8, H2T = 4, NT = 512;
  // LDS: [W2 128 KB | beta1 b2 beta2 w3 (Mlp2Vectors without u) | per-wavefront u: 8 x 1 KB | block prefix]
  uint4* W2 = reinterpret_cast<uint4*>(smem);
  Mlp2Vectors* V = reinterpret_cast<Mlp2Vectors*>(smem + kMlpResW2Bytes);
  float* u_all = reinterpret_cast<float*>(smem + kMlpResBytes);
  int* prefix = reinterpret_cast<int*>(smem + kMlpResBytes + 8 * 1024);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cand = lane & 31, g = lane >> 5;
  // the 32-row blocks every pending query contributes to this round, as exclusive prefix sums in query order: every
  // workgroup works them out for itself from the queries' PhaseState (two queries per thread; round 4's first form had
  // a launch of its own for this -- 5 of a chunk's 17 launches, 24 us of a 32-query call's 308)
  static_assert(kPhaseChunk == 2 * NT, "two queries per thread");
  uint32_t* wave_tot = reinterpret_cast<uint32_t*>(prefix + kPhaseChunk + 4);
  uint32_t nb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q = 2 * tid + h;
    nb[h] = 0;
    if (q < a.n_queries) {
      const PhaseState* st = reinterpret_cast<const PhaseState*>(a.ws + 256 + (unsigned long long)q * a.slot_bytes + a.off_state);
      if (st->status == kPhasePending && st->r == a.round) nb[h] = (uint32_t)(st->sc_n + 31) >> 5;
    }
  }
  const uint32_t inc = wave_scan_add(nb[0] + nb[1]);
  if (lane == 63) wave_tot[wave] = inc;
  {  // once per launch: the weights and the vectors that do not depend on the query
    const uint4* src = EXACT ? reinterpret_cast<const uint4*>(a.mlp.p2x) : a.mlp.p2;
    for (int i = tid; i < kMlpResW2Vec; i += NT) W2[i] = src[i];
    if (EXACT) wg_mlp_xres_vectors<NT>(a.mlp, 0.0f, V); else wg_mlp_res_vectors<NT>(a.mlp, 0.0f, V);
    if (!EXACT && NANN_PHASE_PACKED_EPI && tid < 128) V->u[tid] = a.mlp.w3[tid] * (a.mlp.alpha2[tid] - 1.0f);  // (the same thread wrote the 0 above)
  }
  __syncthreads();
  int total = 0;
  {
    uint32_t base = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) {
      const uint32_t t = wave_tot[w];
      if (w < wave) base += t;
      tot += t;
    }
    total = (int)tot;
    const uint32_t excl = base + inc - (nb[0] + nb[1]);
    prefix[2 * tid] = (int)excl;
    prefix[2 * tid + 1] = (int)(excl + nb[0]);
    if (tid == 0) prefix[kPhaseChunk] = total;  // (queries behind n_queries contribute nothing: prefix[n_queries ..] = total)
  }
  __syncthreads();
  if (total == 0) return;
  const int gw = (int)blockIdx.x * (NT / 64) + wave;
  const int nw = (int)gridDim.x * (NT / 64);
  const int b_lo = (int)((long long)total * gw / nw), b_hi = (int)((long long)total * (gw + 1) / nw);
  if (b_lo >= b_hi) return;
  float* u_w = u_all + wave * 256;  // this wavefront's copy of the current query's u

  // the query of block b (prefix[q] <= b < prefix[q + 1]); uniform over the wavefront
  auto query_of = [&](int b) {
    int lo = 0, hi = a.n_queries - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (prefix[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  // a query's u into the wavefront's LDS copy (the split form keeps it x 2^7 like the table; exact powers of two)
  auto load_u_of = [&](const PhaseState* st) {
    float4 v = reinterpret_cast<const float4*>(st->u)[lane];
    if constexpr (!EXACT) { v.x *= kSplit2Scale; v.y *= kSplit2Scale; v.z *= kSplit2Scale; v.w *= kSplit2Scale; }
    reinterpret_cast<float4*>(u_w)[lane] = v;
  };
  struct Cur {  // the query the wavefront is in
    int q, first, n;
    const int32_t* ids;
    float* out;
  };
  auto enter_query = [&](int q, Cur& c, bool load_u) {
    unsigned char* slot = a.ws + 256 + (unsigned long long)q * a.slot_bytes;
    const PhaseState* st = reinterpret_cast<const PhaseState*>(slot + a.off_state);
    c.q = q;
    c.first = prefix[q];
    c.n = st->sc_n;
    const int off = st->base_off;
    c.ids = a.round == 0 ? a.enter : reinterpret_cast<const int32_t*>(slot + a.off_cand_ids) + off;
    c.out = reinterpret_cast<float*>(slot + a.off_cand_scores) + off;
    if (load_u) load_u_of(st);
  };
  auto row_ptr = [&](const Cur& c, int b) -> const float* {
    const int i = min((b - c.first) * 32 + cand, c.n - 1);
    const uint32_t rid = (uint32_t)c.ids[i];
    return a.proj + (size_t)(rid < a.n_items ? rid : 0u) * kMlpProjWidth + 4 * g;
  };
  auto load_tile = [&](const float* row, int t, float4 (&p)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) p[rr] = *reinterpret_cast<const float4*>(row + 32 * t + 8 * rr);
  };
  // LDS bases (byte addresses), opaque: every read is `base + immediate` (nann_mlp5.h)
  uint32_t w_lo = lds_offset_of(W2) + (uint32_t)lane * 16u;
  uint32_t w_hi = w_lo + 65536u;
  uint32_t v_at = lds_offset_of(V) + (uint32_t)g * 16u;
  uint32_t u_at = lds_offset_of(u_w) + (uint32_t)g * 16u;
  asm volatile("" : "+v"(w_lo), "+v"(w_hi), "+v"(v_at), "+v"(u_at));
  auto vec4 = [&](int float_index) -> f32x4v { return *reinterpret_cast<lds_f4_ptr>(v_at + 4 * float_index); };
  auto uvec4 = [&](int float_index) -> f32x4v { return *reinterpret_cast<lds_f4_ptr>(u_at + 4 * float_index); };
  constexpr int kBeta1 = 256, kB2 = 512, kBeta2 = 640, kW3 = 768;  // Mlp2Vectors, in floats
  constexpr int kW3b = 0;  // split-f16: w3 (alpha2 - 1), in the place of the fused kernel's per-workgroup u (here: one u per wavefront)

  if constexpr (!EXACT) {
    // ---- split-f16: one software pipeline over the wavefront's blocks (wave_mlp_split_pipeline, nann_mlp5.h)
    auto read_u_of = [&](int q) -> float4 {
      const PhaseState* st = reinterpret_cast<const PhaseState*>(a.ws + 256 + (unsigned long long)q * a.slot_bytes + a.off_state);
      float4 v = reinterpret_cast<const float4*>(st->u)[lane];
      v.x *= kSplit2Scale; v.y *= kSplit2Scale; v.z *= kSplit2Scale; v.w *= kSplit2Scale;
      return v;
    };
    Cur cur, nxt;
    enter_query(query_of(b_lo), cur, true);
    int i_cur = 0;
    SplitPipeLds L;
    L.w_lo = w_lo; L.w_hi = w_hi; L.v_at = v_at; L.u_at = u_at;
    L.u_wr = lds_offset_of(u_w) + (uint32_t)lane * 16u;
    L.seed_base = a.proj; L.seed_rows = a.n_items;  // (read by the VAR & 16 pricing build only)
    wave_mlp_split_pipeline<VAR, NANN_PHASE_PACKED_EPI ? kW3b : -1>(
        L, row_ptr(cur, b_lo), b_hi - b_lo,
        [&](int k, const float* row, const float*& next, bool& change, float4& u_next) {  // the block behind block b_lo + k
          const int b = b_lo + k;
          nxt = cur;
          if (b + 1 < b_hi && b + 1 >= prefix[cur.q + 1]) {
            int q2 = cur.q + 1;
            while (prefix[q2 + 1] <= b + 1) ++q2;  // (queries without blocks)
            enter_query(q2, nxt, false);
            change = true;
            u_next = read_u_of(nxt.q);
          }
          next = (b + 1 < b_hi) ? row_ptr(nxt, b + 1) : row;
          i_cur = (b - cur.first) * 32 + cand;
        },
        [&](int, float score) {
          if (g == 0 && i_cur < cur.n && !a.dry) cur.out[i_cur] = score;
          cur = nxt;
        });
    return;
  }
  // ---- exact f32 (wg_score_mlp_xres's arithmetic, nann_mlp5.h): tile by tile, two tiles per trip of a rolled loop
  Cur cur, nxt;
  enter_query(query_of(b_lo), cur, true);
  const float* row = row_ptr(cur, b_lo);
  float4 x[2][4];
  load_tile(row, 0, x[0]);
  load_tile(row, 1, x[1]);
  for (int b = b_lo; b < b_hi; ++b) {
    // the next block: of this query or of the next one that has blocks (its rows are gathered from tile 6 on; its u
    // replaces this one's behind the last tile)
    nxt = cur;
    bool change = false;
    if (b + 1 < b_hi && b + 1 >= prefix[cur.q + 1]) {
      int q2 = cur.q + 1;
      while (prefix[q2 + 1] <= b + 1) ++q2;  // (queries without blocks)
      enter_query(q2, nxt, false);
      change = true;
    }
    const float* next = (b + 1 < b_hi) ? row_ptr(nxt, b + 1) : row;
    const int i = (b - cur.first) * 32 + cand;
    f32x16 acc[H2T];
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4v v = vec4(kB2 + 32 * mt + 8 * rr);
        acc[mt][4 * rr] = v.x; acc[mt][4 * rr + 1] = v.y; acc[mt][4 * rr + 2] = v.z; acc[mt][4 * rr + 3] = v.w;
      }
    auto tile = [&](int t, float4 (&xt)[4]) {
      f32x4v ub[8];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { ub[rr] = uvec4(32 * t + 8 * rr); ub[4 + rr] = vec4(kBeta1 + 32 * t + 8 * rr); }
      float h[16];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const f32x4v u = ub[rr], al = ub[4 + rr];
        constexpr float kInv = 1.0f / kSplit2Scale;  // the table holds 2^7 P: exact both ways
        h[4 * rr + 0] = prelu(u.x + xt[rr].x * kInv, al.x);
        h[4 * rr + 1] = prelu(u.y + xt[rr].y * kInv, al.y);
        h[4 * rr + 2] = prelu(u.z + xt[rr].z * kInv, al.z);
        h[4 * rr + 3] = prelu(u.w + xt[rr].w * kInv, al.w);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_tile(t + 2 >= H1T ? next : row, (t + 2) & (H1T - 1), xt);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4v f[H2T];
#pragma unroll
        for (int mt = 0; mt < H2T; ++mt)  // p2x[t][mt][j][lane]: four chain steps per 16 bytes
          f[mt] = *reinterpret_cast<lds_f4_ptr>((t < 4 ? w_lo : w_hi) + (t & 3) * 16384 + (mt * 4 + j) * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < H2T; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[mt][e], h[4 * j + e], acc[mt], 0, 0, 0);
      }
    };
#pragma unroll 1
    for (int t = 0; t < H1T; t += 2) { tile(t, x[0]); tile(t + 1, x[1]); }
    // the next block's query takes over the wavefront's u (every lane has read this block's)
    if (change) load_u_of(reinterpret_cast<const PhaseState*>(a.ws + 256 + (unsigned long long)nxt.q * a.slot_bytes + a.off_state));
    // PReLU of layer 2 and the bias-free output layer (wg_score_mlp_xres's epilogue: ORDER_O)
    float part = 0.0f;
#pragma unroll
    for (int mt = 0; mt < H2T; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * g;
        part = __fmaf_rn(prelu(acc[mt][r], V->beta2[m]), V->w3[m], part);
      }
    const float other = __shfl_xor(part, 32);
    const float p0 = g == 0 ? part : other, p1 = g == 0 ? other : part;
    if (g == 0 && i < cur.n && !a.dry) cur.out[i] = p0 + p1;
    row = next;
    cur = nxt;
  }
}

constexpr size_t kPhaseScoreLds = (size_t)kMlpResBytes + 8 * 1024 + (size_t)(kPhaseChunk + 4) * 4 + 64;  // + the scan's wavefront totals

}  // namespace nann
