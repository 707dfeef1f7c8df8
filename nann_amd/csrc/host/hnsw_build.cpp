// hnsw_build.cpp -- HNSW index construction for the retrieval path's INPUT files
// (SURVEY.md 8 f1).  The reference builds its index with Faiss, which is not vendored:
//     index = faiss.IndexHNSWFlat(d, M=32); index.add(embeddings)
//     (NANN_impls/nann/delivery/build_hnsw_index.py:33-35)
// and then exports hnsw.{offsets, neighbors, cum_nneighbor_per_level, levels} as per-level CSR
// files (:41-66).  This file restates the published algorithm Faiss implements (Malkov &
// Yashunin, "Efficient and robust approximate nearest neighbor search using Hierarchical
// Navigable Small World graphs", alg. 1-4, with Faiss' parameter conventions: L2 metric,
// M links above level 0 and 2M at level 0, efConstruction = 40, level l drawn with
// P(level >= l) = M^-l, neighbour selection by the heuristic of alg. 4 without
// keepPrunedConnections) and writes the SAME arrays.  Faiss' version is not pinned by the
// reference and its insertion order is thread-schedule dependent, so index CONTENTS are not a
// parity target ("parity unpinned at the index-build boundary"); the file layout
// and the structural invariants are, and tests/test_index_build.py checks them plus recall.
//
// Host code, plain C++17 + std::thread; C ABI at the bottom.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <queue>
#include <random>
#include <thread>
#include <vector>

namespace {

struct Hnsw {
  int64_t n = 0;
  int d = 0;
  int M = 32;
  int ef_construction = 40;
  const float* x = nullptr;          // [n, d] f32
  std::vector<int> levels;           // Faiss convention: number of levels of the node (top level + 1)
  std::vector<int> cum_nb;           // cum_nneighbor_per_level: [0, 2M, 3M, 4M, ...]
  std::vector<int64_t> offsets;      // offsets[i] .. offsets[i+1]: the node's neighbour slots, all levels
  std::vector<int32_t> neighbors;    // -1 = empty slot
  int64_t entry_point = -1;
  int max_level = -1;

  int nb_at(int level) const { return level == 0 ? 2 * M : M; }
  int cum(int level) const { return level == 0 ? 0 : 2 * M + (level - 1) * M; }
  int32_t* links(int64_t i, int level) { return neighbors.data() + offsets[i] + cum(level); }
  const int32_t* links(int64_t i, int level) const { return neighbors.data() + offsets[i] + cum(level); }

  float dist(const float* a, int64_t j) const {
    const float* b = x + j * d;
    float s = 0.f;
    for (int k = 0; k < d; ++k) {
      const float t = a[k] - b[k];
      s += t * t;
    }
    return s;
  }
};

using Cand = std::pair<float, int64_t>;  // (distance, id)

// alg. 4: walk candidates nearest-first; keep c iff it is closer to the base than to every
// neighbour kept so far
void shrink(const Hnsw& h, std::vector<Cand>& cands, int max_size) {
  std::sort(cands.begin(), cands.end());
  std::vector<Cand> out;
  for (const Cand& c : cands) {
    bool good = true;
    for (const Cand& s : out) {
      if (h.dist(h.x + c.second * h.d, s.second) < c.first) { good = false; break; }
    }
    if (good) {
      out.push_back(c);
      if ((int)out.size() >= max_size) break;
    }
  }
  cands.swap(out);
}

struct Visited {
  std::vector<uint32_t> tag;
  uint32_t cur = 0;
  explicit Visited(int64_t n) : tag((size_t)n, 0) {}
  void next() { if (++cur == 0) { std::fill(tag.begin(), tag.end(), 0); cur = 1; } }
  bool test_and_set(int64_t i) { if (tag[(size_t)i] == cur) return true; tag[(size_t)i] = cur; return false; }
};

// alg. 2: beam search of width ef on one level, starting from (ep, d_ep)
void search_level(Hnsw& h, const float* q, int level, int64_t ep, float d_ep, int ef, Visited& vis,
                  std::vector<std::mutex>& locks, std::vector<Cand>& result) {
  std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> cand;  // nearest first
  std::priority_queue<Cand> best;                                        // farthest first
  vis.next();
  vis.test_and_set(ep);
  cand.emplace(d_ep, ep);
  best.emplace(d_ep, ep);
  std::vector<int32_t> nbrs;
  while (!cand.empty()) {
    const Cand c = cand.top();
    if (c.first > best.top().first && (int)best.size() >= ef) break;
    cand.pop();
    {
      std::lock_guard<std::mutex> lk(locks[(size_t)c.second]);
      const int32_t* l = h.links(c.second, level);
      nbrs.assign(l, l + h.nb_at(level));
    }
    for (int32_t v : nbrs) {
      if (v < 0) break;
      if (vis.test_and_set(v)) continue;
      const float dv = h.dist(q, v);
      if ((int)best.size() < ef || dv < best.top().first) {
        cand.emplace(dv, v);
        best.emplace(dv, v);
        if ((int)best.size() > ef) best.pop();
      }
    }
  }
  result.clear();
  while (!best.empty()) { result.push_back(best.top()); best.pop(); }
}

// link src -> dst on `level`; if src's list is full, re-select among its links + dst (alg. 1 l.13-16)
void add_link(Hnsw& h, int64_t src, int64_t dst, int level) {
  int32_t* l = h.links(src, level);
  const int cap = h.nb_at(level);
  if (l[cap - 1] < 0) {  // room left
    int i = cap - 1;
    while (i > 0 && l[i - 1] < 0) --i;
    l[i] = (int32_t)dst;
    return;
  }
  std::vector<Cand> cands;
  cands.reserve((size_t)cap + 1);
  const float* xs = h.x + src * h.d;
  cands.emplace_back(h.dist(xs, dst), dst);
  for (int i = 0; i < cap; ++i) cands.emplace_back(h.dist(xs, l[i]), (int64_t)l[i]);
  shrink(h, cands, cap);
  int i = 0;
  for (const Cand& c : cands) l[i++] = (int32_t)c.second;
  for (; i < cap; ++i) l[i] = -1;
}

void insert(Hnsw& h, int64_t id, Visited& vis, std::vector<std::mutex>& locks, std::mutex& ep_mu) {
  const float* q = h.x + id * h.d;
  const int top = h.levels[(size_t)id] - 1;
  int64_t ep;
  int ep_level;
  {
    std::lock_guard<std::mutex> lk(ep_mu);
    ep = h.entry_point;
    ep_level = h.max_level;
    if (ep < 0) {  // first node
      h.entry_point = id;
      h.max_level = top;
      return;
    }
  }
  float d_ep = h.dist(q, ep);
  // greedy descent through the levels above the node's top level (alg. 1 l.5-7)
  for (int level = ep_level; level > top; --level) {
    bool changed = true;
    std::vector<int32_t> nbrs;
    while (changed) {
      changed = false;
      {
        std::lock_guard<std::mutex> lk(locks[(size_t)ep]);
        const int32_t* l = h.links(ep, level);
        nbrs.assign(l, l + h.nb_at(level));
      }
      for (int32_t v : nbrs) {
        if (v < 0) break;
        const float dv = h.dist(q, v);
        if (dv < d_ep) { d_ep = dv; ep = v; changed = true; }
      }
    }
  }
  std::vector<Cand> w;
  for (int level = std::min(top, ep_level); level >= 0; --level) {
    search_level(h, q, level, ep, d_ep, h.ef_construction, vis, locks, w);  // alg. 1 l.9
    std::vector<Cand> sel = w;
    shrink(h, sel, h.nb_at(level));                                         // alg. 1 l.10
    {
      // A multi-threaded build makes `id` discoverable on level L before its level L-1 list is written:
      // another thread may already have back-linked into that (still empty) list.  Those entries are
      // merged with our selection instead of being overwritten (Faiss holds the node's lock for the
      // whole insertion; the links it would have seen are the ones kept here).
      std::lock_guard<std::mutex> lk(locks[(size_t)id]);
      int32_t* l = h.links(id, level);
      const int cap = h.nb_at(level);
      std::vector<Cand> merged = sel;
      for (int i = 0; i < cap && l[i] >= 0; ++i) {
        bool have = false;
        for (const Cand& c : sel) have |= c.second == (int64_t)l[i];
        if (!have) merged.emplace_back(h.dist(q, l[i]), (int64_t)l[i]);
      }
      if ((int)merged.size() > cap) shrink(h, merged, cap);
      int i = 0;
      for (const Cand& c : merged) l[i++] = (int32_t)c.second;
      for (; i < cap; ++i) l[i] = -1;
    }
    for (const Cand& c : sel) {                                             // alg. 1 l.11-16
      std::lock_guard<std::mutex> lk(locks[(size_t)c.second]);
      add_link(h, c.second, id, level);
    }
    // next level starts from the nearest element found on this one
    const Cand nearest = *std::min_element(w.begin(), w.end());
    ep = nearest.second;
    d_ep = nearest.first;
  }
  if (top > ep_level) {
    std::lock_guard<std::mutex> lk(ep_mu);
    if (top > h.max_level) { h.max_level = top; h.entry_point = id; }
  }
}

}  // namespace

extern "C" {

// Builds the graph and returns the Faiss-shaped raw arrays through caller-provided buffers:
//   levels            i32[n]          number of levels per node (top level + 1)
//   neighbors         i32[n_slots]    -1 = empty; node i's slots start at offsets[i]
//   offsets           i64[n + 1]
//   cum_nneighbor     i32[max_levels + 1]   [0, 2M, 3M, ...]
// Call once with neighbors == nullptr to obtain *n_slots and *max_levels (levels and offsets are
// filled), then again with the buffers.  seed fixes the level draw; with n_threads == 1 the
// whole build is deterministic (nodes inserted by descending level, then ascending id).
int nann_hnsw_build(const float* x, int64_t n, int32_t d, int32_t M, int32_t ef_construction,
                    uint64_t seed, int32_t n_threads, int32_t* levels, int64_t* offsets,
                    int32_t* neighbors, int64_t* n_slots, int32_t* cum_nneighbor, int32_t* max_levels) {
  if (!x || n <= 0 || d <= 0 || M < 2 || !levels || !offsets || !n_slots || !max_levels) return 7;
  Hnsw h;
  h.n = n; h.d = d; h.M = M; h.ef_construction = ef_construction > 0 ? ef_construction : 40; h.x = x;
  h.levels.resize((size_t)n);
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> uni(0.0, 1.0);
  const double mult = 1.0 / std::log((double)M);
  int maxl = 1;
  for (int64_t i = 0; i < n; ++i) {
    double u = uni(rng);
    if (u < 1e-300) u = 1e-300;
    const int lv = (int)std::floor(-std::log(u) * mult) + 1;
    h.levels[(size_t)i] = lv;
    maxl = std::max(maxl, lv);
  }
  h.offsets.resize((size_t)n + 1);
  h.offsets[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int lv = h.levels[(size_t)i];
    h.offsets[(size_t)i + 1] = h.offsets[(size_t)i] + 2 * M + (int64_t)(lv - 1) * M;
  }
  std::memcpy(levels, h.levels.data(), (size_t)n * 4);
  std::memcpy(offsets, h.offsets.data(), (size_t)(n + 1) * 8);
  *n_slots = h.offsets[(size_t)n];
  *max_levels = maxl;
  if (cum_nneighbor) {
    cum_nneighbor[0] = 0;
    for (int l = 1; l <= maxl; ++l) cum_nneighbor[l] = 2 * M + (l - 1) * M;
  }
  if (!neighbors) return 0;

  h.neighbors.assign((size_t)*n_slots, -1);
  // insertion order: top level first (Faiss adds the highest levels first), ascending id inside
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(),
                   [&](int64_t a, int64_t b) { return h.levels[(size_t)a] > h.levels[(size_t)b]; });
  std::vector<std::mutex> locks((size_t)n);
  std::mutex ep_mu;
  const int T = std::max(1, n_threads);
  // the first few nodes go in serially so that every thread starts from a connected graph
  const int64_t serial = std::min<int64_t>(n, 256);
  {
    Visited vis(n);
    for (int64_t k = 0; k < serial; ++k) insert(h, order[(size_t)k], vis, locks, ep_mu);
  }
  // level by level, so that a node's upper layers exist before lower-level nodes descend them
  int64_t pos = serial;
  while (pos < n) {
    const int lv = h.levels[(size_t)order[(size_t)pos]];
    int64_t end = pos;
    while (end < n && h.levels[(size_t)order[(size_t)end]] == lv) ++end;
    std::atomic<int64_t> next(pos);
    auto work = [&]() {
      Visited vis(n);
      for (;;) {
        const int64_t k = next.fetch_add(1);
        if (k >= end) break;
        insert(h, order[(size_t)k], vis, locks, ep_mu);
      }
    };
    if (T == 1) {
      work();
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back(work);
      for (auto& t : th) t.join();
    }
    pos = end;
  }
  std::memcpy(neighbors, h.neighbors.data(), (size_t)*n_slots * 4);
  return 0;
}

}  // extern "C"
