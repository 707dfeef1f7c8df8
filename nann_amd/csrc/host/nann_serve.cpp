// nann_serve -- a C++ serving host over the C ABI alone (include/nann_hip.h): no Python, no torch, no HIP
// headers.  What the reference does with TF-Serving sessions x virtual GPUs x MPS behind blaze-benchmark's
// closed-loop consumers (blaze-benchmark/benchmark/core/model.cc:192-235, predict_request_consumer.cc:17-53)
// is, on this design, ONE resident index and a batching front end: concurrent single requests with the serving
// signature (comm_seq f16[1, L*d] + level_topn -> top_k i64[1, k], build_opt_graph.py:151-159) are aggregated
// into one nann_search launch, by --lanes dispatcher threads, each with a stream, a workspace and page-locked staging
// of its own, so that one batch's copies run under another batch's search.  The Python twin is nann_amd/serving.py; this program is the proof that a C++
// host needs nothing but libnann_hip.so.
//
// Round 3: (1) for the l2 / mlp scorers the request is REDUCED ON THE THREAD THAT SUBMITS IT (the RPC handler of a real
// server): the mean of the non-pad history rows -- bit for bit what nann_user_seq_mean computes on the device -- so a
// dispatcher stages d floats per request instead of the L x d halves of comm_seq (512 B instead of 12.8 KB) and the
// per-batch host work drops 25x; the attention model still receives the raw sequence.  (2) The load generator keeps
// its CLOSED LOOP per logical client (one request in flight each, predict_request_consumer.cc:17-53) but multiplexes
// the clients over --client-threads OS threads, each sleeping on ONE completion counter: the dispatcher wakes a thread
// at most once per batch instead of once per request, and no condition variable lives on a client's stack (the
// round-2 version could notify a Request its client had already destroyed).  (3) Dispatcher and client threads are
// pinned to disjoint cores.
//
//   nann_serve <index_dir> <item_embs_dir> <dim> [--clients N] [--client-threads T] [--seconds S] [--max-batch B]
//              [--max-wait-us U] [--ef E] [--topk K] [--seq-len L] [--lanes N] [--model-dir DIR] [--probe-out FILE]
//              [--no-pin 1]
// index_dir / item_embs_dir: the files build_hnsw_index.py writes (nann_amd.index_build writes the same).
// Closed loop: every logical client sends a request, waits for its reply, sends the next.  Prints one JSON line.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#if defined(__F16C__) && defined(__AVX2__)
#include <immintrin.h>
#endif

#include "nann_hip.h"

namespace {

using Clock = std::chrono::steady_clock;

[[noreturn]] void die(const char* what) {
  std::fprintf(stderr, "nann_serve: %s: %s\n", what, nann_last_error());
  std::exit(1);
}
#define CHECK(expr) do { if ((expr) != NANN_OK) die(#expr); } while (0)

struct ClientThread;
// admission control, as BlazeXlaOp has it: "waiting pool is full" / "blaze wait too long" (blaze_xla_kernel.cc:229-236)
constexpr int32_t kStatusQueueFull = -2, kStatusDeadline = -3;

struct Request {
  const uint16_t* comm_seq = nullptr;  // f16 bits [L * d]: the request as the caller sent it (attention model: staged as it is)
  const float* q = nullptr;            // l2 / mlp: the query vector the submitting thread reduced it to (f32 [d])
  int64_t* top_k = nullptr;            // [level_topn[5]]
  int32_t level_topn[6] = {0, 0, 0, 0, 0, 0};  // the request's own `level_topn` feed (build_opt_graph.py:75,151-159)
  Clock::time_point t_submit;          // for the deadline (BlazeXlaOp's wait_ms, blaze_xla_kernel.cc:221-258)
  int32_t status = -1;                 // nann_status, or kStatusQueueFull / kStatusDeadline (refused before any search)
  std::atomic<int> done{0};
  ClientThread* owner = nullptr;       // whose completion counter the reply bumps
};

// one OS thread of the load generator: its logical clients' replies arrive on ONE counter
struct ClientThread {
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<long long> completed{0};
  std::atomic<int> sleeping{0};
};

struct Queue {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Request*> pending;
  int waiting = 0;  // dispatchers blocked in cv.wait (submitters notify only then)
  bool closing = false;
};

uint16_t f32_to_f16_bits(float f) {  // round to nearest even; enough for a load generator
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t m = x & 0x7fffffu;
  if (e <= 0) return (uint16_t)sign;  // flush tiny values
  if (e >= 31) return (uint16_t)(sign | 0x7c00u);
  uint32_t h = sign | ((uint32_t)e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)h;
}

float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) { x = sign; }
    else {  // subnormal
      int s = 0; uint32_t mm = m;
      while (!(mm & 0x400u)) { mm <<= 1; ++s; }
      x = sign | ((uint32_t)(127 - 15 - s + 1) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e - 15 + 127) << 23) | (m << 13);
  }
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// comm_seq f16[L, d] -> q f32[d]: mean over the rows that are not all-zero (padding), accumulated row by row in f32 and
// divided once -- the operation order of nann_user_seq_mean's kernel (k_user_seq_mean) and of the oracle, so the
// device sees the same bits whichever side reduces the request.
void seq_mean_host(const uint16_t* seq, int L, int d, float* q) {
  int count = 0;
#if defined(__F16C__) && defined(__AVX2__)
  if (d % 8 == 0 && d <= 512) {
    __m256 acc[64];
    for (int v = 0; v < d / 8; ++v) acc[v] = _mm256_setzero_ps();
    const __m128i absmask = _mm_set1_epi16(0x7fff);
    for (int r = 0; r < L; ++r) {
      const __m128i* row = reinterpret_cast<const __m128i*>(seq + (size_t)r * d);
      __m128i nz = _mm_setzero_si128();
      for (int v = 0; v < d / 8; ++v) {
        const __m128i h = _mm_loadu_si128(row + v);
        nz = _mm_or_si128(nz, _mm_and_si128(h, absmask));
        acc[v] = _mm256_add_ps(acc[v], _mm256_cvtph_ps(h));  // exact conversion, one rounded add per row: the device's order
      }
      count += !_mm_testz_si128(nz, nz);
    }
    const __m256 cnt = _mm256_set1_ps((float)count);
    for (int v = 0; v < d / 8; ++v)
      _mm256_storeu_ps(q + 8 * v, count ? _mm256_div_ps(acc[v], cnt) : _mm256_setzero_ps());
    return;
  }
#endif
  for (int r = 0; r < L; ++r) {
    uint16_t nz = 0;
    for (int k = 0; k < d; ++k) nz |= (uint16_t)(seq[(size_t)r * d + k] & 0x7fffu);
    count += nz != 0;
  }
  for (int k = 0; k < d; ++k) q[k] = 0.0f;
  for (int r = 0; r < L; ++r)
    for (int k = 0; k < d; ++k) q[k] = q[k] + f16_bits_to_f32(seq[(size_t)r * d + k]);
  for (int k = 0; k < d; ++k) q[k] = count ? q[k] / (float)count : 0.0f;
}

void pin_to_core(int core) {
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(core, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

// the cores this process may run on: the affinity mask, cut to the cgroup's CPU quota when there is one (a container
// that sees 128 cores in its mask may own 16 of them)
std::vector<int> allowed_cores() {
  cpu_set_t set;
  CPU_ZERO(&set);
  std::vector<int> out;
  if (sched_getaffinity(0, sizeof(set), &set) == 0)
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &set)) out.push_back(c);
  if (out.empty()) out.push_back(0);
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[64];
    long long period = 0;
    if (std::fscanf(f, "%63s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
      const long long n = std::max(1ll, std::atoll(quota) / period);
      if ((long long)out.size() > n) out.resize((size_t)n);
    }
    std::fclose(f);
  }
  return out;
}

void* load(const std::string& path, int dtype, int64_t* count, int64_t elem_bytes) {
  void* p = nullptr;
  int64_t bytes = 0;
  if (nann_huge_const_load(path.c_str(), dtype, nullptr, 0, /*allow_cast=*/1, &p, &bytes) != NANN_OK) die(path.c_str());
  *count = bytes / elem_bytes;
  return p;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <index_dir> <item_embs_dir> <dim> [--clients N] [--client-threads T] [--seconds S] [--max-batch B] "
                         "[--max-wait-us U] [--ef E] [--topk K] [--seq-len L] [--lanes N] [--model-dir DIR] [--probe-out FILE] [--no-pin 1] "
                         "[--mixed-topn 1] [--probe-topn a,b,c,d,e,k] [--max-queue N] [--deadline-ms MS]\n", argv[0]);
    return 2;
  }
  const std::string index_dir = argv[1], embs_dir = argv[2];
  const int d = std::atoi(argv[3]);
  int clients = 64, client_threads = 0, max_batch = 256, max_wait_us = 200, ef = 128, topk = 200, L = 50, lanes = 2, no_pin = 0;
  int mixed_topn = 0, max_queue = 0;
  double seconds = 3.0, deadline_ms = 0.0;
  std::string model_dir, probe_out, probe_topn;
  for (int i = 4; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    if (k == "--clients") clients = std::atoi(argv[i + 1]);
    else if (k == "--client-threads") client_threads = std::atoi(argv[i + 1]);
    else if (k == "--seconds") seconds = std::atof(argv[i + 1]);
    else if (k == "--max-batch") max_batch = std::atoi(argv[i + 1]);
    else if (k == "--max-wait-us") max_wait_us = std::atoi(argv[i + 1]);
    else if (k == "--ef") ef = std::atoi(argv[i + 1]);
    else if (k == "--topk") topk = std::atoi(argv[i + 1]);
    else if (k == "--seq-len") L = std::atoi(argv[i + 1]);
    else if (k == "--lanes") lanes = std::max(1, std::atoi(argv[i + 1]));
    else if (k == "--model-dir") model_dir = argv[i + 1];
    else if (k == "--probe-out") probe_out = argv[i + 1];  // after the run: one fixed request, its reply written as text
    else if (k == "--no-pin") no_pin = std::atoi(argv[i + 1]);
    else if (k == "--mixed-topn") mixed_topn = std::atoi(argv[i + 1]);  // every other logical client asks for half the beam and half the k
    else if (k == "--probe-topn") probe_topn = argv[i + 1];             // the probe request's own level_topn (within --ef / --topk)
    else if (k == "--max-queue") max_queue = std::atoi(argv[i + 1]);    // requests refused while this many wait ("waiting pool is full")
    else if (k == "--deadline-ms") deadline_ms = std::atof(argv[i + 1]); // a request that waited longer is failed, not searched ("blaze wait too long")
    else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
  }
  if (nann_device_count() < 1) { std::fprintf(stderr, "nann_serve: no HIP device\n"); return 1; }
  const std::vector<int> cores = allowed_cores();
  if (client_threads <= 0)  // the cores the dispatchers leave, at most one thread per logical client
    client_threads = std::max(1, std::min(clients, std::max(1, (int)cores.size() - lanes)));
  client_threads = std::min(client_threads, clients);

  // ---- the resident index: HugeConst loads with the casts build_model() asks for (build_opt_graph.py:70,83-90)
  nann_index_desc desc = {};
  int64_t n_rows = 0, n_ids = 0, cnt = 0;
  desc.item_embs = load(embs_dir + "/item_embs.npy", NANN_F16, &n_rows, 2);
  desc.item_ids = static_cast<const int64_t*>(load(embs_dir + "/item_ids.npy", NANN_I64, &n_ids, 8));
  desc.n_items = n_ids;
  desc.d = d;
  desc.emb_dtype = NANN_F16;
  if (n_rows != n_ids * d) { std::fprintf(stderr, "nann_serve: item_embs.npy is not [%lld, %d]\n", (long long)n_ids, d); return 1; }
  for (int l = 0; l < 2; ++l) {
    const std::string stem = index_dir + "/neighbors_level_" + std::to_string(l);
    desc.nb_values[l] = static_cast<const int32_t*>(load(stem + "_values.npy", NANN_I32, &desc.nb_nnz[l], 4));
    desc.nb_row_splits[l] = static_cast<const int64_t*>(load(stem + "_row_splits.npy", NANN_I64, &cnt, 8));
    if (cnt != n_ids + 1) { std::fprintf(stderr, "nann_serve: level %d row_splits has %lld entries\n", l, (long long)cnt); return 1; }
  }
  desc.enter_points = static_cast<const int32_t*>(load(index_dir + "/enter_points.npy", NANN_I32, &desc.n_enter, 4));
  desc.on_device = 1;
  nann_index* ix = nullptr;
  CHECK(nann_index_create(&desc, &ix));

  nann_scorer* own_scorer = nullptr;
  nann_model* model = nullptr;
  if (model_dir.empty()) {
    nann_scorer_desc sd = {};
    sd.kind = NANN_SCORER_L2;
    sd.d = d;
    sd.emb_dtype = NANN_F16;
    CHECK(nann_scorer_create(&sd, &own_scorer));
  } else {
    CHECK(nann_model_load(model_dir.c_str(), d, NANN_F16, L, &model));
  }
  const bool attention = model && nann_model_kind(model) == NANN_MODEL_ATTENTION;
  // l2 / mlp: the scorer nann_search takes; requests arrive reduced to their query vector
  const nann_scorer* scorer = attention ? nullptr : (model ? nann_model_scorer(model) : own_scorer);
  const int seq_d = attention ? 64 : d;  // the attention model's sequence is [L, 64]
  const size_t seq_elems = (size_t)L * seq_d;

  // the scorer's per-index table goes in BEFORE traffic (nann_scorer_prepare: built, waited for and pinned here, so no
  // request pays the hipMalloc + build + stream wait of a first search); without room for it the searches read the
  // embedding rows -- slower, same answers
  {
    const int rc = model ? nann_model_prepare(model, ix, nullptr) : nann_scorer_prepare(scorer, ix, nullptr);
    if (rc == NANN_ERR_CAPACITY) std::fprintf(stderr, "nann_serve: %s\n", nann_last_error());
    else if (rc != NANN_OK) die("nann_scorer_prepare");
  }
  // ---- per lane: a stream, the device buffers of one launch, page-locked staging
  // --ef / --topk are the server's MAXIMA (workspace, plan, row stride of a reply); a request carries its own level_topn
  const int32_t level_topn[6] = {ef, ef, ef, ef, ef, topk};
  int64_t ws_bytes = 0;
  if (attention) CHECK(nann_search_model_workspace_bytes(ix, model, level_topn, max_batch, &ws_bytes));
  else CHECK(nann_search_workspace_bytes(ix, level_topn, max_batch, &ws_bytes));
  const int64_t in_bytes = attention ? (int64_t)seq_elems * 2 : (int64_t)d * 4;  // staged per request
  struct Lane {
    nann_stream_t stream = nullptr;
    void *ws = nullptr, *d_in = nullptr, *d_topk = nullptr, *d_status = nullptr, *d_topn = nullptr;
    unsigned char* h_in = nullptr;
    int32_t* h_topn = nullptr;
    int64_t* h_topk = nullptr;
    int32_t* h_status = nullptr;
    std::vector<ClientThread*> touched;
  };
  std::vector<Lane> lane((size_t)lanes);
  for (Lane& ln : lane) {
    CHECK(nann_stream_create(&ln.stream));
    CHECK(nann_malloc(&ln.ws, ws_bytes));
    CHECK(nann_malloc(&ln.d_in, (int64_t)max_batch * in_bytes));
    CHECK(nann_malloc(&ln.d_topk, (int64_t)max_batch * topk * 8));
    CHECK(nann_malloc(&ln.d_status, (int64_t)max_batch * 4));
    CHECK(nann_malloc(&ln.d_topn, (int64_t)max_batch * 6 * 4));
    CHECK(nann_host_malloc(reinterpret_cast<void**>(&ln.h_topn), (int64_t)max_batch * 6 * 4));
    CHECK(nann_host_malloc(reinterpret_cast<void**>(&ln.h_in), (int64_t)max_batch * in_bytes));
    CHECK(nann_host_malloc(reinterpret_cast<void**>(&ln.h_topk), (int64_t)max_batch * topk * 8));
    CHECK(nann_host_malloc(reinterpret_cast<void**>(&ln.h_status), (int64_t)max_batch * 4));
  }

  // ---- request material: histories made of real item rows (a few thousand rows copied back once)
  const int64_t pool_rows = std::min<int64_t>(n_ids, 4096);
  std::vector<uint16_t> pool((size_t)pool_rows * d);
  CHECK(nann_memcpy(pool.data(), desc.item_embs, (int64_t)pool.size() * 2, 1, nullptr));
  CHECK(nann_stream_synchronize(nullptr));

  Queue q;
  std::atomic<long long> served{0}, failed{0}, launches{0}, batched{0}, refused{0}, expired{0};
  // a request that is answered without a search (admission control): its caller is told like any other
  auto finish_unsearched = [&](Request* r, int32_t status) {
    r->status = status;
    ClientThread* owner = r->owner;
    r->done.store(1, std::memory_order_release);
    if (owner) {
      owner->completed.fetch_add(1, std::memory_order_seq_cst);
      if (owner->sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(owner->mu);
        owner->cv.notify_one();
      }
    }
  };
  std::vector<ClientThread> cthreads((size_t)client_threads);
  std::vector<std::vector<float>> lat((size_t)client_threads);
  const auto t_end = Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(seconds));

  // ---- one launch for a batch of requests on a lane, replies handed back to their callers
  auto run_batch = [&](Lane& ln, const std::vector<Request*>& batch) {
    const int b = (int)batch.size();
    for (int i = 0; i < b; ++i)
      std::memcpy(ln.h_in + (size_t)i * in_bytes, attention ? static_cast<const void*>(batch[i]->comm_seq) : static_cast<const void*>(batch[i]->q),
                  (size_t)in_bytes);
    CHECK(nann_memcpy(ln.d_in, ln.h_in, (int64_t)b * in_bytes, 0, ln.stream));
    // level_topn is a per-request feed of the serving signature: a batch whose requests all ask for the server's values
    // takes the uniform launch, a mixed one carries its [b, 6] table to the device (the `level_topn` argument of nann_search_opt / nann_search_model_opt)
    bool uniform = true;
    for (int i = 0; i < b; ++i) {
      std::memcpy(ln.h_topn + (size_t)i * 6, batch[i]->level_topn, 6 * sizeof(int32_t));
      uniform = uniform && std::memcmp(batch[i]->level_topn, level_topn, sizeof(level_topn)) == 0;
    }
    const int32_t* d_topn = nullptr;
    if (!uniform) {
      CHECK(nann_memcpy(ln.d_topn, ln.h_topn, (int64_t)b * 6 * 4, 0, ln.stream));
      d_topn = static_cast<const int32_t*>(ln.d_topn);
    }
    if (attention) {
      CHECK(nann_search_model_opt(ix, model, ln.d_in, b, level_topn, d_topn, ln.ws, ws_bytes, static_cast<int64_t*>(ln.d_topk),
                                  nullptr, nullptr, static_cast<int32_t*>(ln.d_status), nullptr, /*options*/ nullptr,
                                  /*plan*/ nullptr, ln.stream));
    } else {
      CHECK(nann_search_opt(ix, scorer, static_cast<const float*>(ln.d_in), b, level_topn, d_topn, ln.ws, ws_bytes,
                            static_cast<int64_t*>(ln.d_topk), nullptr, nullptr, static_cast<int32_t*>(ln.d_status), nullptr,
                            /*phase_ticks*/ nullptr, /*options*/ nullptr, /*plan*/ nullptr, ln.stream));
    }
    CHECK(nann_memcpy(ln.h_topk, ln.d_topk, (int64_t)b * topk * 8, 1, ln.stream));
    CHECK(nann_memcpy(ln.h_status, ln.d_status, (int64_t)b * 4, 1, ln.stream));
    CHECK(nann_stream_synchronize(ln.stream));
    launches.fetch_add(1);
    batched.fetch_add(b);
    std::vector<ClientThread*>& touched = ln.touched;  // the client threads with a reply in this batch, each once
    touched.clear();
    for (int i = 0; i < b; ++i) {
      Request* r = batch[i];
      // the request's own k, never more than the row holds (a k outside [0, topk] fails that query in the kernel --
      // NANN_ERR_BAD_ARGUMENT, zeroed row -- and must not size a host copy: ADVICE r4)
      const int own_k = std::min(std::max(r->level_topn[5], 0), topk);
      std::memcpy(r->top_k, ln.h_topk + (size_t)i * topk, (size_t)own_k * 8);
      r->status = ln.h_status[(size_t)i];
      ClientThread* owner = r->owner;  // (read before the release: the request may be reused the moment `done` is seen)
      r->done.store(1, std::memory_order_release);
      if (owner) {
        owner->completed.fetch_add(1, std::memory_order_seq_cst);
        if (std::find(touched.begin(), touched.end(), owner) == touched.end()) touched.push_back(owner);
      }
    }
    // one wake-up per client THREAD per batch, and only if it sleeps.  The client stores `sleeping` and THEN loads
    // `completed`; this side adds to `completed` and THEN loads `sleeping`: both pairs are seq_cst, so at least one side
    // sees the other's store (with release / acquire alone the two loads may both read the old values -- store-to-load
    // reordering -- and the client sleeps through its last reply: ADVICE r3)
    for (ClientThread* t : touched)
      if (t->sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(t->mu);
        t->cv.notify_one();
      }
  };

  // ---- the dispatchers: one launch per batch of whatever arrived within max_wait_us of the first request
  std::vector<std::thread> dispatchers;
  for (int li = 0; li < lanes; ++li)
    dispatchers.emplace_back([&, li] {
      if (!no_pin) pin_to_core(cores[(size_t)li % cores.size()]);
      std::vector<Request*> batch;
      for (;;) {
        batch.clear();
        {
          std::unique_lock<std::mutex> lk(q.mu);
          ++q.waiting;
          q.cv.wait(lk, [&] { return !q.pending.empty() || q.closing; });
          if (q.pending.empty() && q.closing) { --q.waiting; return; }
          const auto deadline = Clock::now() + std::chrono::microseconds(max_wait_us);
          while ((int)q.pending.size() < max_batch && !q.closing &&
                 q.cv.wait_until(lk, deadline, [&] { return (int)q.pending.size() >= max_batch || q.closing; })) {
          }
          --q.waiting;
          const size_t take = std::min<size_t>(q.pending.size(), (size_t)max_batch);
          batch.assign(q.pending.begin(), q.pending.begin() + (long)take);
          q.pending.erase(q.pending.begin(), q.pending.begin() + (long)take);
        }
        if (deadline_ms > 0.0) {  // "blaze wait too long": a request that waited past its deadline is failed, not searched
          const auto now = Clock::now();
          size_t keep = 0;
          for (Request* r : batch) {
            if (std::chrono::duration<double, std::milli>(now - r->t_submit).count() > deadline_ms) {
              expired.fetch_add(1);
              finish_unsearched(r, kStatusDeadline);
            } else batch[keep++] = r;
          }
          batch.resize(keep);
        }
        if (!batch.empty()) run_batch(lane[(size_t)li], batch);
      }
    });

  // what a server's request handler does before queueing: reduce the history to the query vector (l2 / mlp)
  auto submit = [&](Request* const* reqs, int n) {
    const auto now = Clock::now();
    std::vector<Request*> turned_away;
    bool wake = false;
    {
      std::lock_guard<std::mutex> lk(q.mu);
      for (int i = 0; i < n; ++i) {
        reqs[i]->t_submit = now;
        if (max_queue > 0 && (int)q.pending.size() >= max_queue) turned_away.push_back(reqs[i]);  // "waiting pool is full"
        else q.pending.push_back(reqs[i]);
      }
      wake = q.waiting > 0;
    }
    if (wake) q.cv.notify_all();
    for (Request* r : turned_away) { refused.fetch_add(1); finish_unsearched(r, kStatusQueueFull); }
  };

  // ---- closed-loop clients (predict_request_consumer.cc:17-53), `clients` of them over `client_threads` OS threads
  std::vector<std::thread> workers;
  for (int c = 0; c < client_threads; ++c)
    workers.emplace_back([&, c] {
      if (!no_pin) pin_to_core(cores[(size_t)(lanes + c) % cores.size()]);
      const int mine = clients / client_threads + (c < clients % client_threads ? 1 : 0);  // logical clients of this thread
      ClientThread& me = cthreads[(size_t)c];
      std::mt19937 rng(1234u + (unsigned)c);
      std::normal_distribution<float> noise(0.0f, 0.05f);
      // a few histories per thread, made up front: the generator must not be what the test measures
      constexpr int kVariants = 16;
      std::vector<uint16_t> seqs((size_t)kVariants * seq_elems, 0);
      for (int v = 0; v < kVariants; ++v) {
        const int64_t row = (int64_t)(rng() % (uint64_t)pool_rows);
        for (int l = 0; l < L - 5; ++l)  // a user who looked at neighbours of one item (padding rows stay zero)
          for (int k = 0; k < seq_d; ++k)
            seqs[(size_t)v * seq_elems + (size_t)l * seq_d + k] =
                f32_to_f16_bits(f16_bits_to_f32(pool[(size_t)row * d + (k % d)]) + noise(rng));
      }
      std::vector<Request> reqs((size_t)mine);
      std::vector<int64_t> outs((size_t)mine * topk);
      std::vector<float> qs((size_t)mine * d);
      std::vector<Clock::time_point> t0((size_t)mine);
      std::vector<Request*> ready;
      unsigned turn = 0;
      auto arm = [&](int i) {  // logical client i sends its next request
        Request& r = reqs[(size_t)i];
        r.comm_seq = &seqs[(size_t)(turn++ % kVariants) * seq_elems];
        r.top_k = &outs[(size_t)i * topk];
        r.owner = &me;
        for (int j = 0; j < 6; ++j) r.level_topn[j] = level_topn[j];
        if (mixed_topn && (i & 1)) {  // every other logical client: half the beam, half the k (its own `level_topn` feed)
          for (int j = 0; j < 5; ++j) r.level_topn[j] = std::max(1, ef / 2);
          r.level_topn[5] = std::max(1, std::min(topk / 2, 2 * (ef / 2)));
        }
        r.status = -1;
        r.done.store(0, std::memory_order_relaxed);
        t0[(size_t)i] = Clock::now();
        if (!attention) {  // the handler's share of the request: comm_seq -> query vector
          seq_mean_host(r.comm_seq, L, d, &qs[(size_t)i * d]);
          r.q = &qs[(size_t)i * d];
        }
        ready.push_back(&r);
      };
      for (int i = 0; i < mine; ++i) arm(i);
      submit(ready.data(), (int)ready.size());
      ready.clear();
      long long seen = 0;
      int in_flight = mine;
      bool stop = false;
      while (in_flight > 0) {
        // sleep until at least one reply is in (spin briefly first: a batch takes ~0.5 ms)
        if (me.completed.load(std::memory_order_acquire) == seen) {
          std::unique_lock<std::mutex> lk(me.mu);
          me.sleeping.store(1, std::memory_order_seq_cst);
          me.cv.wait(lk, [&] { return me.completed.load(std::memory_order_seq_cst) != seen; });
          me.sleeping.store(0, std::memory_order_seq_cst);
        }
        const auto now = Clock::now();
        stop = stop || now >= t_end;
        for (int i = 0; i < mine; ++i) {
          Request& r = reqs[(size_t)i];
          if (r.owner == nullptr || !r.done.load(std::memory_order_acquire)) continue;
          ++seen;
          --in_flight;
          lat[(size_t)c].push_back(std::chrono::duration<float, std::milli>(now - t0[(size_t)i]).count());
          if (r.status == 0) served.fetch_add(1); else failed.fetch_add(1);
          r.owner = nullptr;  // retired
          if (!stop) { arm(i); ++in_flight; }
        }
        if (!ready.empty()) { submit(ready.data(), (int)ready.size()); ready.clear(); }
      }
    });
  const auto t_start = Clock::now();
  for (auto& w : workers) w.join();
  const double wall = std::chrono::duration<double>(Clock::now() - t_start).count();
  {
    std::lock_guard<std::mutex> lk(q.mu);
    q.closing = true;
  }
  q.cv.notify_all();
  for (auto& t : dispatchers) t.join();

  std::vector<float> all;
  for (auto& v : lat) all.insert(all.end(), v.begin(), v.end());
  std::sort(all.begin(), all.end());
  auto pct = [&](double p) { return all.empty() ? 0.0f : all[std::min(all.size() - 1, (size_t)(p * (double)all.size()))]; };
  const long long total = served.load() + failed.load();
  std::printf("{\"host\": \"nann_serve (C++ over the C ABI)\", \"scorer\": \"%s\", \"items\": %lld, \"dim\": %d, \"ef\": %d, "
              "\"topk\": %d, \"clients\": %d, \"client_threads\": %d, \"max_batch\": %d, \"max_wait_us\": %d, \"lanes\": %d, \"pinned\": %s, "
              "\"staged_bytes_per_request\": %lld, \"seconds\": %.2f, \"requests\": %lld, "
              "\"failed_requests\": %lld, \"refused_queue_full\": %lld, \"expired_deadline\": %lld, \"mixed_level_topn\": %s, \"qps\": %.1f, \"launches\": %lld, \"mean_batch\": %.1f, "
              "\"latency_ms\": {\"p50\": %.3f, \"p90\": %.3f, \"p99\": %.3f, \"max\": %.3f}}\n",
              model_dir.empty() ? "l2" : model_dir.c_str(), (long long)n_ids, d, ef, topk, clients, client_threads, max_batch, max_wait_us,
              lanes, no_pin ? "false" : "true", (long long)in_bytes, wall, total, failed.load(), refused.load(), expired.load(),
              mixed_topn ? "true" : "false", (double)total / wall, launches.load(),
              launches.load() ? (double)batched.load() / (double)launches.load() : 0.0, pct(0.5), pct(0.9), pct(0.99),
              all.empty() ? 0.0f : all.back());

  if (!probe_out.empty()) {  // a fixed request through the same path: item row 0 as the whole history, no noise
    std::vector<uint16_t> seq(seq_elems, 0);
    for (int l = 0; l < L - 5; ++l)
      for (int k = 0; k < seq_d; ++k) seq[(size_t)l * seq_d + k] = pool[(size_t)(k % d)];
    std::vector<int64_t> out((size_t)topk);
    std::vector<float> qv((size_t)d);
    // two requests in ONE launch: the server's level_topn, and -- with --probe-topn -- one with its own; the file holds
    // the second when there is one (status, then its k ids)
    Request r, r2;
    std::vector<int64_t> out2((size_t)topk);
    r.comm_seq = seq.data();
    r.top_k = out.data();
    for (int j = 0; j < 6; ++j) r.level_topn[j] = r2.level_topn[j] = level_topn[j];
    if (!attention) { seq_mean_host(seq.data(), L, d, qv.data()); r.q = qv.data(); }
    r2.comm_seq = r.comm_seq; r2.q = r.q; r2.top_k = out2.data();
    std::vector<Request*> probes{&r};
    if (!probe_topn.empty()) {
      if (std::sscanf(probe_topn.c_str(), "%d,%d,%d,%d,%d,%d", &r2.level_topn[0], &r2.level_topn[1], &r2.level_topn[2],
                      &r2.level_topn[3], &r2.level_topn[4], &r2.level_topn[5]) != 6) die("--probe-topn a,b,c,d,e,k");
      for (int j = 0; j < 6; ++j)  // what BatchingServer.submit checks: 0 <= t[i] <= the launch's maxima
        if (r2.level_topn[j] < 0 || r2.level_topn[j] > level_topn[j]) die("--probe-topn: every entry must lie in [0, the launch's --ef / --topk]");
      probes.push_back(&r2);
    }
    run_batch(lane[0], probes);
    const Request& shown = probes.size() > 1 ? r2 : r;
    FILE* f = std::fopen(probe_out.c_str(), "w");
    if (!f) { std::fprintf(stderr, "nann_serve: cannot write %s\n", probe_out.c_str()); return 1; }
    std::fprintf(f, "%d\n", shown.status);
    for (int i = 0; i < shown.level_topn[5]; ++i) std::fprintf(f, "%lld\n", (long long)shown.top_k[(size_t)i]);
    std::fclose(f);
  }

  for (Lane& ln : lane) {
    nann_free(ln.ws); nann_free(ln.d_in); nann_free(ln.d_topk); nann_free(ln.d_status); nann_free(ln.d_topn);
    nann_host_free(ln.h_in); nann_host_free(ln.h_topk); nann_host_free(ln.h_status); nann_host_free(ln.h_topn);
    nann_stream_destroy(ln.stream);
  }
  if (own_scorer) nann_scorer_destroy(own_scorer);
  if (model) nann_model_destroy(model);
  nann_index_destroy(ix);
  return 0;
}
