// nann_comm.hip -- the exchange step of the item-id-sharded search (BASELINE
// configs 4-5), owned by the C++ host: one process per GPU, every rank searches every query on
// its shard, then ONE ncclAllGather (RCCL over xGMI) of a packed per-rank record
// [scores f32[B,k] | item ids i64[B,k]] and a merge with TopKV2's order over the shard-major
// concatenation (score descending, ties -> lower shard, then lower local rank).
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (torch ships its
// own copy) shares that one, a plain C++ host gets /opt/rocm's; libnann_hip.so itself keeps
// loading on boxes without RCCL.  The reference has no collective on this path (its only
// multi-GPU mode is replicas, blaze-benchmark/benchmark/core/model.cc:192-235).
#include "nann_search.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

using namespace nann;

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string where;
};

int load_rccl(Rccl** out) {
  static std::mutex mu;
  static Rccl R;
  std::lock_guard<std::mutex> lk(mu);
  if (!R.handle) {
    // an RCCL that is already in the process first (RTLD_NOLOAD), then the system one
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !R.handle; ++pass)
      for (const char* n : names) {
        R.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (R.handle) { R.where = n; break; }
      }
    if (!R.handle) return fail(NANN_ERR_UNSUPPORTED, std::string("RCCL not found: ") + dlerror());
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.handle, "ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.handle, "ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.handle, "ncclCommDestroy"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(dlsym(R.handle, "ncclAllGather"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.handle, "ncclGetErrorString"));
    R.CommCount = reinterpret_cast<decltype(R.CommCount)>(dlsym(R.handle, "ncclCommCount"));
    R.CommGetAsyncError = reinterpret_cast<decltype(R.CommGetAsyncError)>(dlsym(R.handle, "ncclCommGetAsyncError"));
    R.CommAbort = reinterpret_cast<decltype(R.CommAbort)>(dlsym(R.handle, "ncclCommAbort"));
    if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllGather) {
      R.handle = nullptr;
      return fail(NANN_ERR_UNSUPPORTED, "RCCL library lacks the ncclAllGather entry points");
    }
  }
  *out = &R;
  return NANN_OK;
}

#define RCCL_TRY(R, expr)                                                                       \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess)                                                                      \
      return fail(NANN_ERR_HIP, std::string(#expr) + ": " +                                     \
                                    ((R)->GetErrorString ? (R)->GetErrorString(_r) : "rccl error")); \
  } while (0)

inline size_t rec_bytes(long long B, int k) { return ((size_t)B * k * 12 + 255) & ~(size_t)255; }

}  // namespace

// per-rank record: scores (a query that failed on this shard contributes -inf and id 0: its
// slots are never selected while another shard holds real candidates) followed by the ids
__global__ void k_pack_record(const float* scores, const int64_t* ids, const int32_t* status, long long B, int k,
                              unsigned char* rec) {
  float* rs = reinterpret_cast<float*>(rec);
  int64_t* ri = reinterpret_cast<int64_t*>(rec + (size_t)B * k * 4);
  const long long n = B * k;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const bool bad = status != nullptr && status[i / k] != 0;
    rs[i] = bad ? -__builtin_inff() : scores[i];
    ri[i] = bad ? 0 : ids[i];
  }
}

// merge straight from the all-gathered records (rank-major, no transposition): one workgroup
// per query stages its world x k_in scores in LDS in shard-major order and runs TopKV2 on them
__global__ __launch_bounds__(kNT) void k_merge_records(const unsigned char* recv, unsigned long long stride, int world,
                                                       long long B, int k_in, int k_out, float* out_scores,
                                                       int64_t* out_ids) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(TopkScratch)];
  __shared__ int32_t s_pos[kMaxK];
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  float* s_sc = reinterpret_cast<float*>(dyn);  // [world * k_in]
  const long long qi = blockIdx.x;
  const int n_in = world * k_in;
  for (int i = threadIdx.x; i < n_in; i += kNT) {
    const int s = i / k_in, j = i - s * k_in;
    s_sc[i] = reinterpret_cast<const float*>(recv + (size_t)s * stride)[qi * k_in + j];
  }
  __syncthreads();
  wg_topk(nullptr, s_sc, nullptr, n_in, k_out, s_pos, nullptr, nullptr, nullptr, nullptr, scratch);
  __syncthreads();
  for (int r = threadIdx.x; r < k_out; r += kNT) {
    const int pos = s_pos[r], s = pos / k_in, j = pos - s * k_in;
    out_scores[qi * k_out + r] = s_sc[pos];
    out_ids[qi * k_out + r] =
        reinterpret_cast<const int64_t*>(recv + (size_t)s * stride + (size_t)B * k_in * 4)[qi * k_in + j];
  }
}

// Loopback stand-in for the time an all-gather over xGMI takes: what RCCL's ring kernels mostly do is WAIT for the peers'
// bytes while holding one workgroup slot per channel.  16 workgroups of 256 threads sleep until `us` microseconds have
// passed (bounded: 20 ms), so the exchange of a one-GPU measurement occupies the slots and the stream for as long as
// eight GPUs' would, without the memory traffic that repeating the device copies adds (profiles/rd5f_reserve_sweep_*).
__global__ __launch_bounds__(256) void k_loopback_wait(unsigned int us) {
  const long long t0 = wall_clock64();  // constant 100 MHz
  const long long ticks = (long long)(us > 20000u ? 20000u : us) * 100ll;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct nann_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  bool loopback = false;
  int loopback_repeat = 1;  // test facility: the loopback's device copies issued this many times
  int loopback_wait_us = 0; //   ... and a kernel of 16 waiting workgroups in front of them (an exchange as LONG as xGMI's)
  Rccl* R = nullptr;
  // nann_comm_set_timing: four events around the three parts of every nann_sharded_topk call (pack | all-gather | merge)
  bool timing = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timed_once = false;
  // round 6: a dead rank must not hang the node.  done_ev follows the merge of the LAST exchange on its stream; nann_comm_wait
  // polls it with a deadline next to ncclCommGetAsyncError and aborts the communicator (ncclCommAbort: RCCL's kernels see the
  // flag and leave) when either says the exchange will not finish
  hipEvent_t done_ev = nullptr;
  bool exchanged = false;
  bool aborted = false;
};

// an asynchronous error of the RCCL communicator (a peer died, a link went down), as NANN_ERR_HIP; aborts the communicator
static int check_async(nann_comm* c, const char* where) {
  if (!c->comm || !c->R->CommGetAsyncError) return NANN_OK;
  ncclResult_t async = ncclSuccess;
  const ncclResult_t r = c->R->CommGetAsyncError(c->comm, &async);
  if (r == ncclSuccess && (async == ncclSuccess || async == ncclInProgress)) return NANN_OK;
  const ncclResult_t bad = r != ncclSuccess ? r : async;
  const std::string msg = c->R->GetErrorString ? c->R->GetErrorString(bad) : "rccl error";
  if (c->R->CommAbort) (void)c->R->CommAbort(c->comm);
  c->comm = nullptr;
  c->aborted = true;
  return fail(NANN_ERR_HIP, std::string(where) + ": RCCL reports an asynchronous error (" + msg + "); communicator aborted");
}

extern "C" {

int nann_comm_get_unique_id(void* id) {
  if (!id) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_get_unique_id: null argument");
  Rccl* R;
  int rc = load_rccl(&R);
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == NANN_COMM_ID_BYTES, "NANN_COMM_ID_BYTES out of sync with rccl.h");
  RCCL_TRY(R, R->GetUniqueId(static_cast<ncclUniqueId*>(id)));
  return NANN_OK;
}

int nann_comm_create(int32_t world, int32_t rank, const void* id, nann_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_create: bad argument");
  nann_comm* c = new nann_comm();
  c->world = world;
  c->rank = rank;
  c->loopback = world > 1 && !id;  // single-process test facility: every "shard" returns this rank's record
  if (id) {  // (world == 1 with an id: a real one-rank communicator -- the whole RCCL path on a single GPU)
    int rc = load_rccl(&c->R);
    if (rc) { delete c; return rc; }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclResult_t r = c->R->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
      const std::string msg = c->R->GetErrorString ? c->R->GetErrorString(r) : "rccl error";
      delete c;
      return fail(NANN_ERR_HIP, "ncclCommInitRank: " + msg);
    }
  }
  *out = c;
  return NANN_OK;
}

void nann_comm_destroy(nann_comm* c) {
  if (!c) return;
  if (c->comm) (void)c->R->CommDestroy(c->comm);
  for (hipEvent_t e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->done_ev) (void)hipEventDestroy(c->done_ev);
  delete c;
}

int nann_comm_ranks(const nann_comm* c, int32_t* world, int32_t* rccl_ranks) {
  if (!c || !world || !rccl_ranks) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_ranks: null argument");
  *world = c->world;
  *rccl_ranks = 0;  // (no RCCL communicator behind this object: a loopback, or a one-shard communicator without an id)
  if (c->comm && c->R->CommCount) {
    int n = 0;
    RCCL_TRY(c->R, c->R->CommCount(c->comm, &n));
    *rccl_ranks = n;
  }
  return NANN_OK;
}

int nann_comm_set_timing(nann_comm* c, int32_t enabled, int32_t loopback_repeat, int32_t loopback_wait_us) {
  if (!c) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_set_timing: null communicator");
  if (enabled)
    for (hipEvent_t& e : c->ev)
      if (!e) NANN_HIP_TRY(hipEventCreate(&e));
  c->timing = enabled != 0;
  c->timed_once = false;
  c->loopback_repeat = std::max(1, (int)loopback_repeat);
  c->loopback_wait_us = std::max(0, (int)loopback_wait_us);
  return NANN_OK;
}

int nann_comm_last_breakdown(nann_comm* c, float ms[3]) {
  if (!c || !ms) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_last_breakdown: null argument");
  if (!c->timing || !c->timed_once) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_last_breakdown: no timed exchange yet (nann_comm_set_timing)");
  NANN_HIP_TRY(hipEventSynchronize(c->ev[3]));
  for (int i = 0; i < 3; ++i) NANN_HIP_TRY(hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
  return NANN_OK;
}

int nann_sharded_topk_workspace_bytes(int32_t world, int64_t n_queries, int32_t k_in, int64_t* nbytes) {
  if (!nbytes || world < 1 || n_queries < 0 || k_in < 0) return fail(NANN_ERR_BAD_ARGUMENT, "bad argument");
  *nbytes = (int64_t)(rec_bytes(n_queries, k_in) * (size_t)(world + 1));
  return NANN_OK;
}

int nann_sharded_topk(nann_comm* c, const float* scores, const int64_t* ids, const int32_t* status,
                      int64_t n_queries, int32_t k_in, int32_t k_out, void* workspace, int64_t workspace_bytes,
                      float* out_scores, int64_t* out_ids, nann_stream_t stream) {
  if (!c || !scores || !ids || !out_scores || !out_ids) return fail(NANN_ERR_BAD_ARGUMENT, "nann_sharded_topk: null argument");
  if (c->aborted) return fail(NANN_ERR_HIP, "nann_sharded_topk: the communicator was aborted (nann_comm_wait / nann_comm_abort); create a new one");
  {
    const int rc = check_async(c, "nann_sharded_topk");
    if (rc) return rc;
  }
  const int world = c->world;
  const long long n_in = (long long)world * k_in;
  if (k_out < 0 || k_out > kMaxK) return fail(NANN_ERR_UNSUPPORTED, "k_out must be in [0, 1024]");
  if (n_in < k_out) return fail(NANN_ERR_TOPK_K_GT_N, "fewer candidates than k_out");
  if (n_in > kTopkEPT * kNT) return fail(NANN_ERR_UNSUPPORTED, "world * k_in > 16384");
  if (n_queries <= 0 || k_out == 0) return NANN_OK;
  const size_t rb = rec_bytes(n_queries, k_in);
  if (!workspace || workspace_bytes < (int64_t)(rb * (size_t)(world + 1)))
    return fail(NANN_ERR_CAPACITY, "workspace smaller than nann_sharded_topk_workspace_bytes()");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* send = static_cast<unsigned char*>(workspace);
  unsigned char* recv = send + rb;
  const long long n = n_queries * k_in;
  if (c->timing) NANN_HIP_TRY(hipEventRecord(c->ev[0], st));
  hipLaunchKernelGGL(k_pack_record, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)), dim3(256), 0, st,
                     scores, ids, status, (long long)n_queries, (int)k_in, (world > 1 || c->comm) ? send : recv);
  NANN_HIP_TRY(hipGetLastError());
  if (c->timing) NANN_HIP_TRY(hipEventRecord(c->ev[1], st));
  if (world > 1 && c->loopback) {
    if (c->loopback_wait_us > 0) hipLaunchKernelGGL(k_loopback_wait, dim3(16), dim3(256), 0, st, (unsigned int)c->loopback_wait_us);
    for (int rep = 0; rep < c->loopback_repeat; ++rep)
      for (int r = 0; r < world; ++r)
        NANN_HIP_TRY(hipMemcpyAsync(recv + (size_t)r * rb, send, rb, hipMemcpyDeviceToDevice, st));
  } else if (c->comm) {
    RCCL_TRY(c->R, c->R->AllGather(send, recv, rb, ncclChar, c->comm, st));
  } else if (world > 1) {
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_sharded_topk: communicator without RCCL state");
  }
  if (c->timing) NANN_HIP_TRY(hipEventRecord(c->ev[2], st));
  const size_t lds = (size_t)n_in * 4;
  if (lds > 48 * 1024)
    NANN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_merge_records),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_merge_records, dim3((unsigned)n_queries), dim3(kNT), lds, st, recv, (unsigned long long)rb,
                     world, (long long)n_queries, (int)k_in, (int)k_out, out_scores, out_ids);
  NANN_HIP_TRY(hipGetLastError());
  if (c->timing) { NANN_HIP_TRY(hipEventRecord(c->ev[3], st)); c->timed_once = true; }
  if (!c->done_ev) NANN_HIP_TRY(hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming));
  NANN_HIP_TRY(hipEventRecord(c->done_ev, st));
  c->exchanged = true;
  return NANN_OK;
}

int nann_comm_abort(nann_comm* c) {
  if (!c) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_abort: null communicator");
  if (c->comm && c->R->CommAbort) (void)c->R->CommAbort(c->comm);
  else if (c->comm) (void)c->R->CommDestroy(c->comm);
  c->comm = nullptr;
  c->aborted = true;
  return NANN_OK;
}

int nann_comm_wait(nann_comm* c, int32_t timeout_ms) {
  if (!c) return fail(NANN_ERR_BAD_ARGUMENT, "nann_comm_wait: null communicator");
  if (c->aborted) return fail(NANN_ERR_HIP, "nann_comm_wait: the communicator was aborted");
  if (!c->exchanged) return NANN_OK;
  const auto t0 = std::chrono::steady_clock::now();
  int sleep_us = 20;
  for (;;) {
    const hipError_t e = hipEventQuery(c->done_ev);
    if (e == hipSuccess) return NANN_OK;
    if (e != hipErrorNotReady) return fail(NANN_ERR_HIP, std::string("nann_comm_wait: ") + hipGetErrorString(e));
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    const int rc = check_async(c, "nann_comm_wait");
    if (rc) return rc;
    const long long ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    if (timeout_ms >= 0 && ms >= timeout_ms) {
      (void)nann_comm_abort(c);
      return fail(NANN_ERR_HIP, "nann_comm_wait: the exchange did not complete within " + std::to_string(timeout_ms) +
                                    " ms (a rank that never joined the all-gather?); communicator aborted");
    }
    std::this_thread::sleep_for(std::chrono::microseconds(sleep_us));
    sleep_us = std::min(1000, sleep_us * 2);
  }
}

}  // extern "C"
