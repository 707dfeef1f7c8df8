// nann_projcache_c.cpp -- CPU-side stress hook for the pre-projected tables' lifecycle (nann_projcache.h), in
// libnann_host.so: the SAME cache template nann_hip.hip instantiates with HIP, here against a mock device that checks
// the invariants the round-3 cache broke (eviction could free a table another thread had already fetched
// for a launch it had not made yet):
//   * a table is never freed while a launch that reads it is in flight,
//   * nor while a thread holds it between acquire() and the launch,
//   * a freed table is never handed out or touched again,
//   * device memory in use never exceeds the mock's capacity.
// "Launches" complete on a device thread after a short delay; freed blocks are poisoned and kept until the end, so a
// violation is counted instead of crashing the test.  tests/test_projcache_cpu.py drives it.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "nann_projcache.h"

namespace {

struct Block {  // header in front of a mock device allocation
  std::atomic<int> inflight{0};  // launches reading the table that have not completed
  std::atomic<int> holders{0};   // threads between acquire() and their launch
  std::atomic<int> freed{0};
  size_t bytes = 0;
  float payload[4];
};

struct MockEvent { std::atomic<int> pending{0}; };

struct Device {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::pair<Block*, MockEvent*>> queue;  // launches in flight, completed in order
  std::vector<Block*> graveyard;
  std::vector<MockEvent*> events;
  std::atomic<long long> used{0}, peak{0}, violations{0}, frees{0}, builds{0};
  long long capacity = 0;
  bool stop = false;
};
Device* g_dev = nullptr;

struct MockBackend {
  typedef int Stream;
  typedef MockEvent* Event;
  static bool malloc(void** p, size_t bytes) {
    Device& d = *g_dev;
    const long long now = d.used.fetch_add((long long)bytes) + (long long)bytes;
    if (now > d.capacity) { d.used.fetch_sub((long long)bytes); return false; }
    long long pk = d.peak.load();
    while (now > pk && !d.peak.compare_exchange_weak(pk, now)) {}
    Block* b = new Block();
    b->bytes = bytes;
    *p = b->payload;
    return true;
  }
  static Block* block_of(void* p) {
    return reinterpret_cast<Block*>(static_cast<char*>(p) - offsetof(Block, payload));
  }
  static void free(void* p) {
    Device& d = *g_dev;
    Block* b = block_of(p);
    if (b->inflight.load() != 0 || b->holders.load() != 0 || b->freed.exchange(1) != 0) d.violations.fetch_add(1);
    d.used.fetch_sub((long long)b->bytes);
    d.frees.fetch_add(1);
    std::lock_guard<std::mutex> lk(d.mu);
    d.graveyard.push_back(b);  // poisoned, kept: a late toucher is counted, not a crash
  }
  static bool mem_info(size_t* free_b) {
    Device& d = *g_dev;
    const long long f = d.capacity + ((long long)1 << 30) - d.used.load();  // (the cache keeps a GiB of margin)
    *free_b = f > 0 ? (size_t)f : 0;
    return true;
  }
  static bool event_create(Event* e) {
    *e = new MockEvent();
    std::lock_guard<std::mutex> lk(g_dev->mu);
    g_dev->events.push_back(*e);
    return true;
  }
  static void event_destroy(Event) {}  // owned by the device until the end
  static void event_record(Event, Stream) {}  // the launch itself enqueues (block, event): launch_on()
  static bool event_done(Event e) { return e->pending.load() == 0; }
  static void event_wait(Event e) { while (e->pending.load() != 0) std::this_thread::yield(); }
};

typedef nann::ProjCacheT<MockBackend> Cache;

}  // namespace

extern "C" {

// n_threads workers x n_iters searches over n_indices indices of `table_bytes` each on ONE scorer's cache, with
// `capacity_tables` tables' worth of device memory; every 16th search of worker 0 is a prepare / release pair, every
// 64th an index destruction.  out[6] = violations, builds, frees, peak bytes, searches served without a table, tables
// still allocated after the cache is gone (must be 0).  Returns 0.
int nann_projcache_stress(int32_t n_threads, int32_t n_indices, int32_t n_iters, int64_t table_bytes,
                          int32_t capacity_tables, uint64_t seed, int64_t out[6]) {
  Device dev;
  dev.capacity = (long long)table_bytes * capacity_tables;
  g_dev = &dev;
  std::atomic<long long> no_table{0};
  std::thread device([&] {  // completes launches in order, a little later
    std::unique_lock<std::mutex> lk(dev.mu);
    for (;;) {
      dev.cv.wait_for(lk, std::chrono::microseconds(50), [&] { return dev.stop || !dev.queue.empty(); });
      if (dev.queue.empty()) { if (dev.stop) return; continue; }
      auto job = dev.queue.front();
      dev.queue.pop_front();
      lk.unlock();
      std::this_thread::sleep_for(std::chrono::microseconds(5));
      job.first->inflight.fetch_sub(1);   // the kernel finishes ...
      job.second->pending.fetch_sub(1);   // ... then the event behind it
      lk.lock();
    }
  });
  {
    Cache cache;
    std::atomic<uint64_t> uid_base[64];
    for (int i = 0; i < 64; ++i) uid_base[i].store((uint64_t)(i + 1));
    auto worker = [&](int w) {
      std::mt19937_64 rng(seed * 1000003ull + (uint64_t)w);
      MockEvent* last = nullptr;
      for (int it = 0; it < n_iters; ++it) {
        // a closed-loop client: at most two of its requests in flight (it waits for the one before the last)
        if (last && it % 2 == 0) MockBackend::event_wait(last);
        const int k = (int)(rng() % (uint64_t)n_indices);
        const uint64_t uid = uid_base[k].load();
        const bool pin = w == 0 && it % 16 == 7;
        Cache::Ref tab;
        const int rc = cache.acquire(uid, (size_t)table_bytes, true, pin, [&](float* t) {
          dev.builds.fetch_add(1);
          t[0] = (float)uid;  // "build"
          return 0;
        }, &tab);
        if (rc) { dev.violations.fetch_add(1); continue; }
        if (!tab) { no_table.fetch_add(1); continue; }
        Block* b = MockBackend::block_of(tab->table);
        b->holders.fetch_add(1);
        if (b->freed.load() || tab->table[0] != (float)uid) dev.violations.fetch_add(1);
        if (rng() % 4 == 0) std::this_thread::yield();  // the window the round-3 cache lost tables in
        // the launch: in flight from now until the device thread completes it; the event goes behind it
        // (the mock's record hook cannot see the block, so the pending count and the queue entry are made here)
        MockEvent* ev = cache.used(tab, w);
        if (ev) {
          b->inflight.fetch_add(1);
          ev->pending.fetch_add(1);
          std::lock_guard<std::mutex> lk(dev.mu);
          dev.queue.emplace_back(b, ev);
          dev.cv.notify_one();
          last = ev;
        }
        b->holders.fetch_sub(1);
        tab.reset();
        if (pin) cache.release(uid);
        if (w == 0 && it % 64 == 63) {  // an index is destroyed and a new one takes its place
          Cache::drop_index(uid);
          uid_base[k].fetch_add(1000);
        }
      }
    };
    std::vector<std::thread> th;
    for (int w = 0; w < n_threads; ++w) th.emplace_back(worker, w);
    for (auto& t : th) t.join();
    out[3] = dev.peak.load();
  }  // the cache goes: every table is freed (after its launches)
  {
    std::lock_guard<std::mutex> lk(dev.mu);
    dev.stop = true;
    dev.cv.notify_all();
  }
  device.join();
  out[0] = dev.violations.load();
  out[1] = dev.builds.load();
  out[2] = dev.frees.load();
  out[4] = no_table.load();
  out[5] = dev.used.load();
  for (Block* b : dev.graveyard) delete b;
  for (MockEvent* e : dev.events) delete e;
  g_dev = nullptr;
  return 0;
}

// acquire() held the cache's mutex across the allocation and the build, so the first search of a new pair
// stalled every concurrent search on the scorer.  Here one thread builds the table of index 1 for `build_ms`; meanwhile a
// second thread -- started once the build is under way -- hits the (already built) table of index 2 `n_hits` times and a
// third asks for index 1 itself.  out[4] = {hits served while the build was still running, longest single hit in
// microseconds, builds of index 1 (must be 1: the third thread waits for the first's), violations}.
int nann_projcache_slow_build(int32_t build_ms, int32_t n_hits, int64_t out[4]) {
  Device dev;
  dev.capacity = (long long)3 << 20;
  g_dev = &dev;
  std::atomic<int> building{0}, done{0}, builds1{0};
  std::atomic<long long> hits_during{0}, worst_us{0}, bad{0};
  {
    Cache cache;
    Cache::Ref t2;
    if (cache.acquire(2, 1 << 20, true, true, [&](float* t) { t[0] = 2.0f; return 0; }, &t2) || !t2) bad.fetch_add(1);
    t2.reset();
    std::thread builder([&] {
      Cache::Ref t1;
      const int rc = cache.acquire(1, 1 << 20, true, false, [&](float* t) {
        builds1.fetch_add(1);
        building.store(1);
        std::this_thread::sleep_for(std::chrono::milliseconds(build_ms));
        t[0] = 1.0f;
        done.store(1);
        return 0;
      }, &t1);
      if (rc || !t1 || t1->table[0] != 1.0f) bad.fetch_add(1);
    });
    while (!building.load()) std::this_thread::yield();
    std::thread hitter([&] {
      for (int i = 0; i < n_hits; ++i) {
        const auto a = std::chrono::steady_clock::now();
        Cache::Ref t;
        if (cache.acquire(2, 1 << 20, true, false, [&](float*) { bad.fetch_add(1); return 0; }, &t) || !t || t->table[0] != 2.0f) bad.fetch_add(1);
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - a).count();
        long long w = worst_us.load();
        while (us > w && !worst_us.compare_exchange_weak(w, us)) {}
        if (!done.load()) hits_during.fetch_add(1);
      }
    });
    std::thread same([&] {  // wants the table that is being built: waits for it, does not build a second one
      Cache::Ref t;
      if (cache.acquire(1, 1 << 20, true, false, [&](float* t_) { builds1.fetch_add(1); t_[0] = 1.0f; return 0; }, &t) || !t || t->table[0] != 1.0f)
        bad.fetch_add(1);
      if (!done.load()) bad.fetch_add(1);  // (it may only return once the build has finished)
    });
    builder.join(); hitter.join(); same.join();
    cache.release(2);
  }
  out[0] = hits_during.load(); out[1] = worst_us.load(); out[2] = builds1.load(); out[3] = bad.load() + dev.violations.load() + (dev.used.load() != 0);
  for (Block* b : dev.graveyard) delete b;
  for (MockEvent* e : dev.events) delete e;
  g_dev = nullptr;
  return 0;
}

// a call that asks for NO table (nann_search_options.preprojection = 0) must not get the one an earlier call
// cached for the pair.  A table for index 1 is built; a disabled acquire then returns none (and builds none), in either
// order with enabled ones, which keep hitting the cached table.  out[3] = {builds (must be 1), tables a disabled call got
// (must be 0), violations}.
int nann_projcache_disabled_call(int64_t out[3]) {
  Device dev;
  dev.capacity = (long long)3 << 20;
  g_dev = &dev;
  long long builds = 0, got = 0, bad = 0;
  {
    Cache cache;
    auto build = [&](float* t) { ++builds; t[0] = 1.0f; return 0; };
    Cache::Ref t;
    if (cache.acquire(1, 1 << 20, false, false, build, &t) || t) ++got;      // nothing cached yet: no table, no build
    if (cache.acquire(1, 1 << 20, true, false, build, &t) || !t) ++bad;       // the pair's table
    t.reset();
    if (cache.acquire(1, 1 << 20, false, false, build, &t)) ++bad;            // cached now: a disabled call still gets none
    if (t) ++got;
    t.reset();
    if (cache.acquire(1, 1 << 20, true, false, build, &t) || !t || t->table[0] != 1.0f) ++bad;  // and the cache still serves it
    t.reset();
  }
  out[0] = builds; out[1] = got; out[2] = bad + dev.violations.load();
  for (Block* b : dev.graveyard) delete b;
  for (MockEvent* e : dev.events) delete e;
  g_dev = nullptr;
  return 0;
}

}  // extern "C"
