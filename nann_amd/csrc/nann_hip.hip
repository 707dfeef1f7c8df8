// nann_hip.hip -- kernels + C ABI of libnann_hip.so (gfx950 only).
// The ABI is documented in include/nann_hip.h; the workgroup building blocks
// in nann_device.h.  Reference citations are relative to /root/reference/.
#include "nann_eval.h"
#include "nann_attn.h"
#include "host/nann_graphdef_text.h"
#include "host/nann_blaze_options.h"
#include "host/nann_npy.h"
#include "host/nann_projcache.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <sys/stat.h>

using namespace nann;

// ===========================================================================
// host-side helpers
// ===========================================================================
namespace nann {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace nann

namespace {

#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return fail(NANN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
  } while (0)

struct DeviceInfo {
  bool ok = false;
  int cus = 0;
  size_t lds_max = 0;
};

// per-device facts, queried once
int device_info(DeviceInfo* out) {
  static std::mutex mu;
  static DeviceInfo cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(NANN_ERR_NO_DEVICE, "no HIP device");
  if (dev < 0 || dev >= 64) return fail(NANN_ERR_NO_DEVICE, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(mu);
  if (!cache[dev].ok) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    cache[dev].cus = p.multiProcessorCount;
    cache[dev].lds_max = p.maxSharedMemoryPerMultiProcessor ? p.maxSharedMemoryPerMultiProcessor
                                                            : p.sharedMemPerBlock;
    if (cache[dev].lds_max < p.sharedMemPerBlock) cache[dev].lds_max = p.sharedMemPerBlock;
    cache[dev].ok = true;
  }
  *out = cache[dev];
  return NANN_OK;
}

struct ResultBuf {
  OpResult* dev = nullptr;
  OpResult* host = nullptr;
  int device = -1;
};

int get_result_buf(ResultBuf** out) {
  thread_local ResultBuf rb;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (!rb.dev || rb.device != dev) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rb.dev), sizeof(OpResult)));
    if (!rb.host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&rb.host), sizeof(OpResult)));
    rb.device = dev;
  }
  *out = &rb;
  return NANN_OK;
}

int fetch_result(ResultBuf* rb, hipStream_t st) {
  HIP_TRY(hipMemcpyAsync(rb->host, rb->dev, sizeof(OpResult), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return NANN_OK;
}

inline hipStream_t as_stream(nann_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// ===========================================================================
// kernels: per-op drop-ins
// ===========================================================================

// ragged validation (GroupGather_kernel.cc:9-16, bitmap_ops.cc:12-19)
__device__ __forceinline__ int validate_ragged(long long n_values, const int64_t* rs, long long n_splits) {
  if (n_splits == 0) return 1;
  if (rs[0] != 0) return 2;
  if (rs[n_splits - 1] != n_values) return 3;
  return 0;
}

// GroupGather count pass (:137-145) for all groups at once: per frontier row j
// the output offset (exclusive scan of row lengths), then ret_row_splits.
__global__ __launch_bounds__(kNT) void k_group_gather_count(
    const int64_t* params_row_splits, long long n_params_splits, long long n_params_values,
    const int64_t* indices_values, long long n_iv, const int64_t* indices_row_splits,
    long long n_irs, int64_t* ret_row_splits, int64_t* offsets, OpResult* res) {
  __shared__ long long s_wave[kNW];
  __shared__ long long s_running;
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  if (tid == 0) {
    res->n_out = 0; res->n_out_splits = 0; res->bad_i = -1; res->code = 0; res->err = 0;
    s_running = 0; s_bad = 0;
  }
  __syncthreads();
  int code = validate_ragged(n_params_values, params_row_splits, n_params_splits);
  if (code) {
    if (tid == 0) { res->code = code; res->err = NANN_ERR_INVALID_RAGGED_PARAMS; }
    return;
  }
  // indices ragged: its values are indices_values
  code = validate_ragged(n_iv, indices_row_splits, n_irs);
  if (code) {
    if (tid == 0) { res->code = code; res->err = NANN_ERR_INVALID_RAGGED_INDICES; }
    return;
  }
  if (n_params_splits == 1 || n_irs == 1) {  // void inputs -> ([], [0])  :69-77
    if (tid == 0) { ret_row_splits[0] = 0; res->n_out = 0; res->n_out_splits = 1; }
    return;
  }
  const long long n_rows = n_params_splits - 1;
  for (long long j0 = 0; j0 < n_iv; j0 += kNT) {
    const long long j = j0 + tid;
    long long len = 0;
    if (j < n_iv) {
      const long long idx = indices_values[j];
      if (idx < 0 || idx >= n_rows) s_bad = 1;
      else len = params_row_splits[idx + 1] - params_row_splits[idx];
    }
    long long inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    long long wb = 0, tot = 0;
    for (int w = 0; w < kNW; ++w) { const long long t = s_wave[w]; if (w < wave) wb += t; tot += t; }
    const long long run = s_running;
    if (j < n_iv) offsets[j] = run + wb + inc - len;
    __syncthreads();
    if (tid == 0) s_running = run + tot;
    __syncthreads();
  }
  if (s_bad) {
    if (tid == 0) res->err = NANN_ERR_INDEX_OUT_OF_RANGE;
    return;
  }
  if (tid == 0) offsets[n_iv] = s_running;
  __syncthreads();
  for (long long i = tid; i < n_irs; i += kNT) ret_row_splits[i] = offsets[indices_row_splits[i]];
  if (tid == 0) { res->n_out = s_running; res->n_out_splits = n_irs; }
}

// GroupGather fill pass (:152-168): one wavefront per gathered row.
__global__ __launch_bounds__(256) void k_group_gather_fill(const int32_t* __restrict__ params_values,
                                                           const int64_t* __restrict__ params_row_splits,
                                                           const int64_t* __restrict__ indices_values,
                                                           long long n_iv,
                                                           const int64_t* __restrict__ offsets,
                                                           int32_t* __restrict__ ret_values) {
  const int lane = lane_id();
  const long long wave_global = (long long)blockIdx.x * (blockDim.x >> 6) + wave_id();
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long j = wave_global; j < n_iv; j += n_waves) {
    const long long g = indices_values[j];
    const long long s = params_row_splits[g], e = params_row_splits[g + 1], o = offsets[j];
    for (long long c = lane; c < e - s; c += 64) ret_values[o + c] = params_values[s + c];
  }
}

// GroupGather unique=true (GroupGather_kernel.cc:91-131): per group the SET of the gathered values.  The reference
// writes a group in its unordered_set's iteration order -- any order of the distinct values is its answer --; here:
// first-occurrence order of the unique=false list, which is the input (values / row_splits = that op's outputs).
// One open-addressing table for all groups, keyed by (group, value): a 64-bit entry is (value << 32 | position) with the
// position global, so the group of an entry is grp[position] and "the smallest position of a key" is one atomicMin
// on the entry (equal high halves).  Position p is kept iff its key's entry still says p.
constexpr unsigned long long kGguEmpty = ~0ull;
__device__ __forceinline__ uint32_t ggu_hash(uint32_t v, uint32_t g) {
  uint32_t h = v * 0x9E3779B1u ^ (g * 0x85EBCA77u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
  return h;
}
__global__ __launch_bounds__(256) void k_ggu_groups(const int64_t* row_splits, long long n_groups, uint32_t* grp) {
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x)
    for (long long p = row_splits[g] + threadIdx.x; p < row_splits[g + 1]; p += blockDim.x) grp[p] = (uint32_t)g;
}
__global__ __launch_bounds__(256) void k_ggu_insert(const int32_t* values, long long n, const uint32_t* grp,
                                                    unsigned long long* table, uint32_t mask) {
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const uint32_t v = (uint32_t)values[p], g = grp[p];
    const unsigned long long mine = ((unsigned long long)v << 32) | (unsigned long long)(uint32_t)p;
    uint32_t h = ggu_hash(v, g) & mask;
    for (;;) {
      unsigned long long e = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e == kGguEmpty) {
        e = atomicCAS(&table[h], kGguEmpty, mine);
        if (e == kGguEmpty) break;  // claimed
      }
      if ((uint32_t)(e >> 32) == v && grp[(uint32_t)e] == g) { atomicMin(&table[h], mine); break; }
      h = (h + 1) & mask;
    }
  }
}
__device__ __forceinline__ bool ggu_is_first(const int32_t* values, const uint32_t* grp, const unsigned long long* table,
                                             uint32_t mask, long long p) {
  const uint32_t v = (uint32_t)values[p], g = grp[p];
  uint32_t h = ggu_hash(v, g) & mask;
  for (;;) {
    const unsigned long long e = table[h];
    if ((uint32_t)(e >> 32) == v && e != kGguEmpty && grp[(uint32_t)e] == g) return (uint32_t)e == (uint32_t)p;
    h = (h + 1) & mask;
  }
}
// per group: distinct values -> counts[g]
__global__ __launch_bounds__(256) void k_ggu_count(const int32_t* values, const int64_t* row_splits, long long n_groups,
                                                   const uint32_t* grp, const unsigned long long* table, uint32_t mask,
                                                   long long* counts) {
  __shared__ unsigned int s_n;
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (long long p = row_splits[g] + threadIdx.x; p < row_splits[g + 1]; p += blockDim.x)
      mine += ggu_is_first(values, grp, table, mask, p) ? 1u : 0u;
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[g] = s_n;
    __syncthreads();
  }
}
__global__ __launch_bounds__(kNT) void k_ggu_scan(const long long* counts, long long n_groups, int64_t* out_row_splits,
                                                  OpResult* res) {
  __shared__ long long s_wave[kNW];
  __shared__ long long s_running;
  const int tid = threadIdx.x, lane = lane_id(), wave = wave_id();
  if (tid == 0) { s_running = 0; out_row_splits[0] = 0; }
  __syncthreads();
  for (long long g0 = 0; g0 < n_groups; g0 += kNT) {
    const long long g = g0 + tid;
    const long long c = g < n_groups ? counts[g] : 0;
    long long inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long t = __shfl_up(inc, d);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    long long wb = 0, tot = 0;
    for (int w = 0; w < kNW; ++w) { const long long t = s_wave[w]; if (w < wave) wb += t; tot += t; }
    const long long run = s_running;
    if (g < n_groups) out_row_splits[g + 1] = run + wb + inc;
    __syncthreads();
    if (tid == 0) s_running = run + tot;
    __syncthreads();
  }
  if (tid == 0) { res->n_out = s_running; res->n_out_splits = n_groups + 1; res->bad_i = -1; res->code = 0; res->err = 0; }
}
// per group: the kept positions in position order
__global__ __launch_bounds__(256) void k_ggu_emit(const int32_t* values, const int64_t* row_splits, long long n_groups,
                                                  const uint32_t* grp, const unsigned long long* table, uint32_t mask,
                                                  const int64_t* out_row_splits, int32_t* out_values) {
  __shared__ uint32_t s_wave[4];
  __shared__ long long s_base;
  const int lane = lane_id(), wave = wave_id();
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
    if (threadIdx.x == 0) s_base = out_row_splits[g];
    __syncthreads();
    const long long b = row_splits[g], e = row_splits[g + 1];
    for (long long p0 = b; p0 < e; p0 += 256) {
      const long long p = p0 + threadIdx.x;
      const bool keep = p < e && ggu_is_first(values, grp, table, mask, p);
      const unsigned long long m = __ballot(keep);
      if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t wb = 0, tot = 0;
      for (int w = 0; w < 4; ++w) { const uint32_t t = s_wave[w]; if (w < wave) wb += t; tot += t; }
      const long long base = s_base;
      if (keep) out_values[base + wb + __popcll(m & ((1ull << lane) - 1ull))] = values[p];
      __syncthreads();
      if (threadIdx.x == 0) s_base = base + tot;
      __syncthreads();
    }
  }
}

// BitmapRefDifference (bitmap_ops.cc:175-257): one workgroup; the bitmap is
// staged into LDS when it fits, walked by one wavefront, and written back.
template <bool kLds>
__global__ __launch_bounds__(kNT) void k_bitmap_ref_difference(
    const int32_t* values, long long n_values, const int64_t* row_splits, long long n_splits,
    uint32_t* flags, long long n_words, int32_t* c_values, int64_t* c_row_splits, OpResult* res) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int& s_bad = *reinterpret_cast<int*>(smem);  // first 16 bytes: flags; bitmap follows
  const int tid = threadIdx.x;
  if (tid == 0) {
    res->n_out = 0; res->n_out_splits = 0; res->bad_i = -1; res->code = 0; res->err = 0;
    s_bad = 0;
  }
  __syncthreads();
  const int code = validate_ragged(n_values, row_splits, n_splits);
  if (code) {
    if (tid == 0) { res->code = code; res->err = NANN_ERR_INVALID_RAGGED_INPUT; }
    return;
  }
  if (n_splits == 1) {  // void input :187-196; bitmap forwarded untouched
    if (tid == 0) { c_row_splits[0] = 0; res->n_out = 0; res->n_out_splits = 1; }
    return;
  }
  // bounds pre-check so that an error leaves the bitmap untouched
  const unsigned long long limit = (unsigned long long)n_words * 32ull;
  for (long long j = tid; j < n_values; j += kNT) {
    const int32_t v = values[j];
    if (v < 0 || (unsigned long long)v >= limit) s_bad = 1;
  }
  __syncthreads();
  if (s_bad) {
    if (tid == 0) res->err = NANN_ERR_INDEX_OUT_OF_RANGE;
    return;
  }
  uint32_t* bm = kLds ? reinterpret_cast<uint32_t*>(smem + 16) : flags;
  if (kLds) {
    for (long long w = tid; w < n_words; w += kNT) bm[w] = flags[w];
  }
  __syncthreads();
  if (wave_id() == 0) {
    int err = 0;
    long long base = 0;
    const long long groups = n_splits - 1;
    const uint32_t n_items = limit < 0x7fffffffull ? (uint32_t)limit : 0x7fffffffu;
    if (lane_id() == 0) c_row_splits[0] = 0;
    for (long long g = 0; g < groups; ++g) {  // ONE bitmap for all groups
      const long long s = row_splits[g], e = row_splits[g + 1];
      base = wave_walk_span<kLds>(values + s, (int)(e - s), bm, n_items, c_values, (int)base, &err);
      if (lane_id() == 0) c_row_splits[g + 1] = base;
    }
    if (lane_id() == 0) { res->n_out = base; res->n_out_splits = n_splits; }
  }
  __syncthreads();
  if (kLds) {
    for (long long w = tid; w < n_words; w += kNT) flags[w] = bm[w];
  }
}

// ---- BloomFilterDifference (bitmap_ops.cc:264-425) ------------------------------------------------
// tensorflow::Fingerprint64 = FarmHash farmhashna::Hash64, restated for the <= 11-byte decimal strings
// of int32 ids (oracle/nann_oracle.c carries the longer branches and the pinning note).
__device__ __forceinline__ uint64_t fh_rot(uint64_t v, int s) { return (v >> s) | (v << (64 - s)); }
__device__ __forceinline__ uint64_t fh_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul; a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul; b ^= (b >> 47);
  return b * mul;
}
__device__ __forceinline__ uint64_t fingerprint64_short(const unsigned char* s, int len) {  // len in [1, 16]
  constexpr uint64_t K0 = 0xc3a5c85c97cb3127ULL, K2 = 0x9ae16a3b2f90404fULL;
  auto fetch = [&](int at, int bytes) { uint64_t v = 0; for (int i = 0; i < bytes; ++i) v |= (uint64_t)s[at + i] << (8 * i); return v; };
  const uint64_t n = (uint64_t)len;
  if (len >= 8) {
    const uint64_t mul = K2 + n * 2, a = fetch(0, 8) + K2, b = fetch(len - 8, 8);
    return fh_len16(fh_rot(b, 37) * mul + a, (fh_rot(a, 25) + b) * mul, mul);
  }
  if (len >= 4) {
    const uint64_t mul = K2 + n * 2;
    return fh_len16(n + (fetch(0, 4) << 3), fetch(len - 4, 4), mul);
  }
  const uint32_t y = (uint32_t)s[0] + ((uint32_t)s[len >> 1] << 8), z = (uint32_t)n + ((uint32_t)s[len - 1] << 2);
  uint64_t v = (y * K2) ^ (z * K0);
  v ^= v >> 47;
  return v * K2;
}

struct BloomParams { long long bucket, bucket_size; unsigned long long prime[4]; };

__device__ __forceinline__ void bloom_positions(int32_t node, const BloomParams& B, uint32_t pos[4]) {
  unsigned char buf[12];  // std::to_string(node), bitmap_ops.cc:346
  int len = 0;
  uint32_t mag = node < 0 ? 0u - (uint32_t)node : (uint32_t)node;
  unsigned char rev[10];
  int nd = 0;
  do { rev[nd++] = (unsigned char)('0' + mag % 10u); mag /= 10u; } while (mag);
  if (node < 0) buf[len++] = '-';
  while (nd) buf[len++] = rev[--nd];
  uint64_t raw = fingerprint64_short(buf, len);
  if (B.bucket > 0) raw = raw % (uint64_t)B.bucket;
  const uint64_t mult[4] = {1, 3, 5, 7};
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const uint64_t tmp = ((raw * mult[l]) % B.prime[l] + B.prime[l]) % B.prime[l];
    pos[l] = (uint32_t)(tmp % (uint64_t)(B.bucket_size * 32));
  }
}

// One workgroup; the filter is staged in LDS when it fits.  Positions are hashed by all lanes of
// wavefront 0, 64 nodes per step; the test-and-set of a step runs in node order on one lane (the four
// positions of a node and of its neighbours may coincide, and which node "misses" decides what is kept).
template <bool kLds>
__global__ __launch_bounds__(kNT) void k_bloom_filter_difference(
    const int32_t* values, long long n_values, const int64_t* row_splits, long long n_splits, uint32_t* flags,
    long long n_words, BloomParams B, int32_t* c_values, int64_t* c_row_splits, OpResult* res) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = lane_id();
  if (tid == 0) { res->n_out = 0; res->n_out_splits = 0; res->bad_i = -1; res->code = 0; res->err = 0; }
  __syncthreads();
  const int code = validate_ragged(n_values, row_splits, n_splits);
  if (code) {
    if (tid == 0) { res->code = code; res->err = NANN_ERR_INVALID_RAGGED_INPUT; }
    return;
  }
  if (n_splits == 1) {  // void input :315-325; filter forwarded untouched
    if (tid == 0) { c_row_splits[0] = 0; res->n_out = 0; res->n_out_splits = 1; }
    return;
  }
  uint32_t* bm = kLds ? reinterpret_cast<uint32_t*>(smem) : flags;
  if (kLds) for (long long w = tid; w < n_words; w += kNT) bm[w] = flags[w];
  __syncthreads();
  if (wave_id() == 0) {
    long long out = 0;
    if (lane == 0) c_row_splits[0] = 0;
    for (long long g = 0; g + 1 < n_splits; ++g) {
      const long long s = row_splits[g], e = row_splits[g + 1];
      for (long long j0 = s; j0 < e; j0 += 64) {
        const int cnt = (int)((e - j0) < 64 ? (e - j0) : 64);
        uint32_t pos[4] = {0, 0, 0, 0};
        int32_t node = 0;
        if (lane < cnt) { node = values[j0 + lane]; bloom_positions(node, B, pos); }
        uint64_t keepmask = 0;
        for (int i = 0; i < cnt; ++i) {  // node order
          uint32_t p[4];
#pragma unroll
          for (int l = 0; l < 4; ++l) p[l] = (uint32_t)__builtin_amdgcn_readlane((int)pos[l], i);
          int miss = 0;
          if (lane == 0) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
              const uint32_t bit = 1u << (p[l] & 31), w = bm[p[l] >> 5];
              if (!(w & bit)) { ++miss; bm[p[l] >> 5] = w | bit; }
            }
          }
          miss = __builtin_amdgcn_readfirstlane(miss);
          if (miss > 0) keepmask |= 1ull << i;
        }
        if ((keepmask >> lane) & 1ull) c_values[out + popc64(keepmask & lanemask_lt(lane))] = node;
        out += popc64(keepmask);
      }
      if (lane == 0) c_row_splits[g + 1] = out;
    }
    if (lane == 0) { res->n_out = out; res->n_out_splits = n_splits; }
  }
  __syncthreads();
  if (kLds) for (long long w = tid; w < n_words; w += kNT) flags[w] = bm[w];
}

// GatherV2 axis 0 (gather_functor.h:96-103): word-wise coalesced row copy.
template <typename W>
__global__ __launch_bounds__(256) void k_gather_rows(const W* __restrict__ params, long long n_rows,
                                                     long long row_words,
                                                     const int32_t* __restrict__ indices,
                                                     long long n_idx, W* __restrict__ out,
                                                     OpResult* res) {
  const long long total = n_idx * row_words;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total;
       w += (long long)gridDim.x * blockDim.x) {
    const long long i = w / row_words, c = w - i * row_words;
    const long long r = indices[i];
    if (r < 0 || r >= n_rows) {  // FastBoundsCheck, gather_functor.h:85-89
      atomicMin(reinterpret_cast<unsigned long long*>(&res->bad_i), (unsigned long long)i);
      continue;
    }
    out[w] = params[r * row_words + c];
  }
}

__global__ void k_init_result(OpResult* res) {
  res->n_out = 0; res->n_out_splits = 0; res->bad_i = 0x7fffffffffffffffll; res->code = 0; res->err = 0;
}

// TopKV2 (topk_op.cc:104-205): one workgroup per row.
__global__ __launch_bounds__(kNT) void k_topk(const float* values, long long n_cols, int k,
                                              float* out_values, int32_t* out_indices) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(TopkScratch)];
  const long long row = blockIdx.x;
  wg_topk(nullptr, values + row * n_cols, nullptr, (int)n_cols, k, out_indices + row * k, nullptr,
          out_values + row * k, nullptr, nullptr, scratch);
}

// TopKV2 with k beyond the radix-select kernel's 1024 (topk_op.cc:154-173 sorts the whole row
// when k == n; the reference's own tests go to k = n = 5000): one workgroup per row, the row's
// (key, ~position) pairs bitonic-sorted in LDS (n <= 16384 -> 128 KB), first k written out.
__global__ __launch_bounds__(kNT) void k_topk_sort(const float* values, long long n_cols, int k, int n_pow2,
                                                   float* out_values, int32_t* out_indices) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* e = reinterpret_cast<unsigned long long*>(smem);
  const long long row = blockIdx.x;
  const float* v = values + row * n_cols;
  for (int i = threadIdx.x; i < n_pow2; i += kNT)  // pads sort behind every real pair (key 0 is below any score key)
    e[i] = i < n_cols ? (((unsigned long long)score_key(v[i]) << 32) | (uint32_t)(~(uint32_t)i)) : 0ull;
  __syncthreads();
  for (int size = 2; size <= n_pow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (n_pow2 >> 1); t += kNT) {
        const int lo = ((t / stride) * stride << 1) + (t % stride), hi = lo + stride;
        const bool desc = ((lo & size) == 0);  // descending overall
        const unsigned long long a = e[lo], b = e[hi];
        if ((a < b) == desc) { e[lo] = b; e[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += kNT) {
    const int pos = (int)(~(uint32_t)(e[i] & 0xffffffffull));
    out_indices[row * k + i] = pos;
    out_values[row * k + i] = v[pos];
  }
}

// merge of per-shard top-k lists: TopKV2 over the shard-major
// concatenation, ids carried along.
__global__ __launch_bounds__(kNT) void k_merge_topk(const float* scores, const int64_t* ids,
                                                    int n_in, int k_out, float* out_scores,
                                                    int64_t* out_ids) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[sizeof(TopkScratch)];
  const long long qi = blockIdx.x;
  wg_topk(nullptr, scores + qi * n_in, nullptr, n_in, k_out, nullptr, nullptr, out_scores + qi * k_out,
          ids + qi * n_in, out_ids + qi * k_out, scratch);
}

// comm_seq f16[B, L, d] -> q f32[B, d]; one workgroup per query, one thread per
// dimension.  Same order as oracle_user_seq_mean.
__global__ void k_user_seq_mean(const uint16_t* seq, int seq_len, int d, float* q) {
  // generic form (any d, any seq_len): one thread per row for the count, one per column for the sums
  __shared__ int s_count;
  const long long b = blockIdx.x;
  const uint16_t* s = seq + b * seq_len * d;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  for (int r = threadIdx.x; r < seq_len; r += blockDim.x) {
    int nz = 0;
    for (int k = 0; k < d; ++k) nz |= (s[(long long)r * d + k] & 0x7fffu) != 0;
    if (nz) atomicAdd(&s_count, 1);
  }
  __syncthreads();
  const int count = s_count;
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    float acc = 0.0f;
    for (int r = 0; r < seq_len; ++r) acc = acc + half_bits_to_float(s[(long long)r * d + k]);
    q[b * d + k] = count ? acc / (float)count : 0.0f;
  }
}

// The shape the serving path feeds (d a multiple of 8, one history <= 48 KB: 50 x 128 f16 = 12.8 KB).  Round 5: the
// generic kernel above read 52 MB at 0.66 TB/s (each of <= 50 threads scanning a whole row with 2-byte strided loads,
// then 2-byte loads again for the sums: 80 us per 4096 queries, 4.3 % of the headline step).  Here a 256-thread
// workgroup copies its query's history into LDS with 16-byte loads, all of them in flight at once (seq_len d / 8
// pieces, row-major = fully coalesced) and flags the non-zero pieces on the way.  The sums then run out of LDS in the
// oracle's order -- per column one chain over the rows ascending from +0, one divide (oracle_user_seq_mean) -- so the
// result is bitwise the generic kernel's.
constexpr int kSeqMeanThreads = 256;
constexpr int kSeqMeanMaxBytes = 48 * 1024;
__global__ __launch_bounds__(kSeqMeanThreads) void k_user_seq_mean_lds(const uint4* seq, int seq_len, int d, float* q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char seq_smem[];
  __shared__ int s_count;
  uint4* rows = reinterpret_cast<uint4*>(seq_smem);
  const int tid = threadIdx.x;
  const int ppr = d >> 3;                 // 16-byte pieces per row
  const int n_pieces = seq_len * ppr;
  const uint4* src = seq + (long long)blockIdx.x * n_pieces;
  if (tid == 0) s_count = 0;
  constexpr int kMaxPer = (kSeqMeanMaxBytes / 16 + kSeqMeanThreads - 1) / kSeqMeanThreads;  // 12
  uint4 v[kMaxPer];
#pragma unroll
  for (int j = 0; j < kMaxPer; ++j) {
    const int i = tid + j * kSeqMeanThreads;
    v[j] = (i < n_pieces) ? src[i] : uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();  // (s_count = 0 above)
  // piece i belongs to row i / ppr; the pieces of a row can straddle wavefronts and trips, so "this piece has a non-zero
  // element" goes through one flag byte per piece and thread r ORs row r's flags behind the barrier
  unsigned char* nzf = seq_smem + (size_t)n_pieces * 16;  // one byte per piece
#pragma unroll
  for (int j = 0; j < kMaxPer; ++j) {
    const int i = tid + j * kSeqMeanThreads;
    if (i < n_pieces) {
      rows[i] = v[j];
      // +-0 halves are zero: both sign bits of a word are cleared before the test
      nzf[i] = ((v[j].x | v[j].y | v[j].z | v[j].w) & 0x7fff7fffu) ? 1 : 0;
    }
  }
  __syncthreads();
  for (int r = tid; r < seq_len; r += kSeqMeanThreads) {
    int nz = 0;
    for (int k = 0; k < ppr; ++k) nz |= nzf[r * ppr + k];
    if (nz) atomicAdd(&s_count, 1);
  }
  __syncthreads();
  const int count = s_count;
  const uint16_t* h = reinterpret_cast<const uint16_t*>(seq_smem);
  for (int k = tid; k < d; k += kSeqMeanThreads) {
    float acc = 0.0f;
    for (int r = 0; r < seq_len; ++r) acc = acc + half_bits_to_float(h[r * d + k]);
    q[(long long)blockIdx.x * d + k] = count ? acc / (float)count : 0.0f;
  }
}

// stand-alone scorer (the BlazeXlaOp contract): rows scored independently.
template <int LPR, int DT>
__global__ __launch_bounds__(256) void k_score_l2(const void* table, long long n_table_rows, int d,
                                                  const int32_t* indices, long long n,
                                                  const float* qv, float* scores, OpResult* res) {
  constexpr int GPW = 64 / LPR;
  const int lane = lane_id();
  const int sub = lane % LPR, grp = lane / LPR;
  float q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) q[k] = qv[sub * 8 + k];
  const long long wave_global = (long long)blockIdx.x * (blockDim.x >> 6) + wave_id();
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long i0 = wave_global * GPW; i0 < n; i0 += n_waves * GPW) {
    const long long i = i0 + grp;
    long long row = -1;
    if (i < n) {
      row = indices ? (long long)indices[i] : i;
      if (row < 0 || row >= n_table_rows) {
        if (sub == 0) atomicMin(reinterpret_cast<unsigned long long*>(&res->bad_i), (unsigned long long)i);
        row = -1;
      }
    }
    float x[8];
    if (row >= 0) {
      const RowChunk<DT> ch = load_chunk<DT>(table, (size_t)row, d, sub);
      chunk_to_float<DT>(ch, x);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = 0.0f;
    }
    const float s = l2_finish<LPR>(q, x);
    if (sub == 0 && i < n) scores[i] = s;
  }
}

// ===========================================================================
// C ABI
// ===========================================================================
// The item-only part of a split-f16 scorer, pre-projected for the indices the scorer has searched (nann_mlp3.h: the item
// half of the MLP's layer 1, f32 [n_items, 256]; nann_attn_proj.h: q_ and the e rows of DNN layer 1, f32 [n_items, 384]):
// built at the first nann_search of the (scorer, index) pair, at most two kept per scorer (the older one goes).
// Lifecycle (round 4): host/nann_projcache.h -- ref-counted tables, LRU among the unpinned ones, retirement instead of
// freeing (an event per stream behind every launch that reads a table), pinning by nann_*_prepare.
struct HipProjBackend {
  typedef hipStream_t Stream;
  typedef hipEvent_t Event;
  static bool malloc(void** p, size_t bytes) {
    if (hipMalloc(p, bytes) == hipSuccess) return true;
    (void)hipGetLastError();  // not sticky: the search goes on without a table
    return false;
  }
  static void free(void* p) { (void)hipFree(p); }
  static bool mem_info(size_t* free_b) {
    size_t total = 0;
    if (hipMemGetInfo(free_b, &total) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
  }
  static bool event_create(Event* e) {
    if (hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
  }
  static void event_destroy(Event e) { (void)hipEventDestroy(e); }
  static void event_record(Event e, Stream s) { (void)hipEventRecord(e, s); }
  static bool event_done(Event e) { return hipEventQuery(e) != hipErrorNotReady; }
  static void event_wait(Event e) { (void)hipEventSynchronize(e); }
};
typedef nann::ProjCacheT<HipProjBackend> ProjCache;
typedef nann::ProjTableT<HipProjBackend> ProjTable;
// ---- options of a search call -------------------------------------------------------------------------------------
// Every knob of the planner is a field of nann_search_options (include/nann_hip.h, round 5); a field left at -1 takes the
// PROCESS DEFAULT: what nann_set_traversal_mode / nann_set_search_reserve / nann_set_preprojection last stored, else the
// environment read ONCE below (A/B tooling: NANN_PREPROJECT=0, NANN_SEARCH_SLOT_RESERVE=n, NANN_MLP_FORM=fused|phased,
// NANN_PHASE_SMALL32=0), else the built-in value.  plan_search and the launchers only ever see the resolved struct.
struct SearchOpt {
  int mode;        // nann_traversal_mode
  int reserve;     // workgroup slots the persistent grids leave free
  int preproject;  // 1: the MLP / attention scorers read their pre-projected tables
  int mlp_form;    // nann_mlp_form
  int small32;     // 1: at most one query per CU -> one 1024-thread workgroup per CU for the traversal (stages)
};
static std::atomic<int> g_traversal_mode{-1}, g_slot_reserve{-1}, g_preproject{-1};
static const SearchOpt& env_defaults() {
  static const SearchOpt d = [] {
    SearchOpt o{NANN_TRAVERSAL_AUTO, 0, 1, NANN_MLP_FORM_AUTO, 1};
    if (const char* e = std::getenv("NANN_PREPROJECT")) o.preproject = !(e[0] == '0' && e[1] == 0);
    if (const char* e = std::getenv("NANN_SEARCH_SLOT_RESERVE")) o.reserve = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("NANN_MLP_FORM")) o.mlp_form = std::string(e) == "fused" ? NANN_MLP_FORM_FUSED : std::string(e) == "phased" ? NANN_MLP_FORM_PHASED : NANN_MLP_FORM_AUTO;
    if (const char* e = std::getenv("NANN_PHASE_SMALL32")) o.small32 = !(e[0] == '0');
    return o;
  }();
  return d;
}
static SearchOpt resolve_options(const nann_search_options* u) {
  SearchOpt o = env_defaults();
  const int gm = g_traversal_mode.load(std::memory_order_relaxed), gr = g_slot_reserve.load(std::memory_order_relaxed),
            gp = g_preproject.load(std::memory_order_relaxed);
  if (gm >= 0) o.mode = gm;
  if (gr >= 0) o.reserve = gr;
  if (gp >= 0) o.preproject = gp;
  if (u) {
    // struct_bytes: a caller built against an older header passes a shorter struct; fields behind it keep their defaults
    const size_t nb = u->struct_bytes > 0 ? (size_t)u->struct_bytes : sizeof(*u);
    auto has = [&](const int32_t* f) { return (size_t)(reinterpret_cast<const char*>(f) - reinterpret_cast<const char*>(u)) + 4 <= nb; };
    if (has(&u->traversal_mode) && u->traversal_mode >= 0) o.mode = u->traversal_mode;
    if (has(&u->slot_reserve) && u->slot_reserve >= 0) o.reserve = u->slot_reserve;
    if (has(&u->preprojection) && u->preprojection >= 0) o.preproject = u->preprojection != 0;
    if (has(&u->mlp_form) && u->mlp_form >= 0) o.mlp_form = u->mlp_form;
  }
  return o;
}
static int check_options(const nann_search_options* u) {
  if (!u) return NANN_OK;
  if (u->struct_bytes < 0 || (u->struct_bytes > 0 && u->struct_bytes < 8)) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_options: struct_bytes");
  const SearchOpt o = resolve_options(u);
  if (o.mode < NANN_TRAVERSAL_AUTO || o.mode > NANN_TRAVERSAL_LDS_HASH32) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_options: unknown traversal_mode");
  if (o.mlp_form < NANN_MLP_FORM_AUTO || o.mlp_form > NANN_MLP_FORM_PHASED) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_options: unknown mlp_form");
  return NANN_OK;
}

struct nann_scorer {
  nann_scorer_desc desc;
  float* dev_weights = nullptr;  // MLP weights block in HBM
  uint4* dev_packed = nullptr;   // split-f16 planes in MFMA A-fragment order
  float4* dev_packed_x = nullptr;  // W2 as f32 A fragments (exact form, layer 2 resident in LDS)
  MlpParams mlp = {};
  mutable ProjCache proj;
};

struct nann_attn_scorer {
  AttnParams P = {};
  int emb_dtype = 0;
  int precision = NANN_MLP_EXACT_F32;  // NANN_MLP_SPLIT_F16: the split-f16 kernels (nann_attn_split.h, nann_attn_proj.h)
  float* dev_weights = nullptr;
  uint4* dev_packed = nullptr;         // split form: A fragments + pre-scaled vectors
  mutable ProjCache proj;
};

struct nann_index {
  nann_index_desc desc;  // device pointers
  uint64_t uid = 0;      // process-unique (a scorer keys its per-index tables by it; a pointer could be reused)
  bool owns = false;
  std::vector<void*> owned;
  int64_t max_deg[2] = {0, 0};
  uint32_t bm_words = 0;  // ceil(N/32) padded to a multiple of 4
  // the probe launch of nann_index_create (round 5): new level-0 nodes per frontier row, measured on THIS graph
  bool probe_valid = false;
  int probe_ef = 0, probe_queries = 0;
  float probe_new_per_row_mean = 0.0f, probe_new_per_row_q90 = 0.0f, probe_new_per_row_max = 0.0f;
};
static void probe_index(nann_index* ix);  // (defined behind search_impl)

extern "C" {

int nann_abi_version(void) { return NANN_ABI_VERSION; }
const char* nann_last_error(void) { return nann::g_err.c_str(); }

int nann_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int nann_malloc(void** dev_ptr, int64_t nbytes) {
  if (!dev_ptr || nbytes < 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_malloc: bad argument");
  HIP_TRY(hipMalloc(dev_ptr, (size_t)std::max<int64_t>(nbytes, 1)));
  return NANN_OK;
}

int nann_free(void* dev_ptr) {
  if (dev_ptr) HIP_TRY(hipFree(dev_ptr));
  return NANN_OK;
}

int nann_memcpy(void* dst, const void* src, int64_t nbytes, int kind, nann_stream_t stream) {
  static const hipMemcpyKind kinds[3] = {hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                                         hipMemcpyDeviceToDevice};
  if (kind < 0 || kind > 2 || nbytes < 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_memcpy: bad argument");
  if (nbytes == 0) return NANN_OK;
  HIP_TRY(hipMemcpyAsync(dst, src, (size_t)nbytes, kinds[kind], as_stream(stream)));
  return NANN_OK;
}

int nann_stream_synchronize(nann_stream_t stream) {
  HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  return NANN_OK;
}

int nann_stream_create(nann_stream_t* out) {
  if (!out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_stream_create: null argument");
  hipStream_t s = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = reinterpret_cast<nann_stream_t>(s);
  return NANN_OK;
}

int nann_stream_destroy(nann_stream_t stream) {
  if (!stream) return NANN_OK;
  HIP_TRY(hipStreamDestroy(as_stream(stream)));
  return NANN_OK;
}

int nann_host_malloc(void** host_ptr, int64_t nbytes) {
  if (!host_ptr || nbytes < 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_host_malloc: bad argument");
  HIP_TRY(hipHostMalloc(host_ptr, (size_t)std::max<int64_t>(nbytes, 1), hipHostMallocDefault));
  return NANN_OK;
}

int nann_host_free(void* host_ptr) {
  if (!host_ptr) return NANN_OK;
  HIP_TRY(hipHostFree(host_ptr));
  return NANN_OK;
}

// ---- HugeConst -------------------------------------------------------------
// The .npy decoder is a host header of its own (host/nann_npy.h: every length checked, fuzzed under ASan in the CPU suite);
// here: the file into memory, the decoder, the payload into HBM.
static uint16_t f32_to_f16_rne(float f) { return nann_npy::f32_to_f16_rne(f); }

static int read_whole_file(const char* path, std::vector<unsigned char>* out) {
  std::ifstream f(path, std::ifstream::binary | std::ifstream::ate);
  if (!f) return fail(NANN_ERR_IO, std::string("Fail to open file: ") + path);  // huge_const_op.cc:96-98
  const std::streamoff size = f.tellg();
  if (size < 0) return fail(NANN_ERR_IO, std::string("Fail to open file: ") + path);
  out->resize((size_t)size);
  f.seekg(0);
  if (size > 0) f.read(reinterpret_cast<char*>(out->data()), size);
  if (!f || f.gcount() != size) return fail(NANN_ERR_IO, std::string("short read of ") + path);
  return NANN_OK;
}

// .npy -> host bytes in `expect_dtype` (the scorer-model loader; HugeConst below copies straight from the file image)
static int read_npy_host(const char* path, int expect_dtype, const int64_t* expect_shape, int expect_rank,
                         int allow_cast, std::vector<char>* out, std::vector<int64_t>* out_shape) {
  std::vector<unsigned char> image;
  int rc = read_whole_file(path, &image);
  if (rc) return rc;
  const unsigned char* payload = nullptr;
  size_t bytes = 0;
  std::vector<char> conv;
  std::string err;
  rc = nann_npy::decode(image.data(), image.size(), expect_dtype, expect_shape, expect_rank, allow_cast != 0, &payload, &bytes, &conv,
                        out_shape, &err);
  if (rc) return fail(rc, err);
  if (!conv.empty() || bytes == 0) { conv.resize(bytes); out->swap(conv); }
  else out->assign(reinterpret_cast<const char*>(payload), reinterpret_cast<const char*>(payload) + bytes);
  return NANN_OK;
}

int nann_huge_const_load(const char* path, int expect_dtype, const int64_t* expect_shape,
                         int expect_rank, int allow_cast, void** dev_ptr, int64_t* nbytes) {
  if (!path || !dev_ptr) return fail(NANN_ERR_BAD_ARGUMENT, "nann_huge_const_load: null argument");
  std::vector<unsigned char> image;
  int rc = read_whole_file(path, &image);
  if (rc) return rc;
  const unsigned char* payload = nullptr;
  size_t bytes = 0;
  std::vector<char> conv;
  std::string err;
  rc = nann_npy::decode(image.data(), image.size(), expect_dtype, expect_shape, expect_rank, allow_cast != 0, &payload, &bytes, &conv,
                        nullptr, &err);
  if (rc) return fail(rc, err);
  const int64_t total = (int64_t)bytes;
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, (size_t)std::max<int64_t>(total, 1)));
  if (total) {
    const hipError_t e = hipMemcpy(d, payload, (size_t)total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(d);  // do not leak the allocation on a failed copy
      return fail(NANN_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
  }
  *dev_ptr = d;
  if (nbytes) *nbytes = total;
  return NANN_OK;
}

// ---- GroupGather -----------------------------------------------------------
int nann_group_gather_count(const int64_t* params_row_splits, int64_t n_params_splits,
                            int64_t n_params_values, const int64_t* indices_values,
                            int64_t n_indices_values, const int64_t* indices_row_splits,
                            int64_t n_indices_splits, int64_t* ret_row_splits,
                            int64_t* scratch_offsets, int64_t* n_ret, int64_t* n_ret_splits,
                            int32_t* ragged_code, nann_stream_t stream) {
  if (!n_ret || !n_ret_splits) return fail(NANN_ERR_BAD_ARGUMENT, "nann_group_gather_count: null out");
  ResultBuf* rb;
  int rc = get_result_buf(&rb);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_group_gather_count, dim3(1), dim3(kNT), 0, st, params_row_splits,
                     (long long)n_params_splits, (long long)n_params_values, indices_values,
                     (long long)n_indices_values, indices_row_splits, (long long)n_indices_splits,
                     ret_row_splits, scratch_offsets, rb->dev);
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (ragged_code) *ragged_code = rb->host->code;
  *n_ret = rb->host->n_out;
  *n_ret_splits = rb->host->n_out_splits;
  if (rb->host->err) return fail(rb->host->err, "GroupGather: invalid input, code " +
                                                    std::to_string(rb->host->code));
  return NANN_OK;
}

int nann_group_gather_fill(const int32_t* params_values, const int64_t* params_row_splits,
                           const int64_t* indices_values, int64_t n_indices_values,
                           const int64_t* scratch_offsets, int32_t* ret_values,
                           nann_stream_t stream) {
  if (n_indices_values <= 0) return NANN_OK;
  const int waves_per_block = 4;
  const long long blocks = std::min<long long>((n_indices_values + waves_per_block - 1) / waves_per_block, 4096);
  hipLaunchKernelGGL(k_group_gather_fill, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     params_values, params_row_splits, indices_values, (long long)n_indices_values,
                     scratch_offsets, ret_values);
  HIP_TRY(hipGetLastError());
  return NANN_OK;
}

// ---- GroupGather unique=true: the set of every group of a unique=false result, first-occurrence order ---------
static uint64_t ggu_table_slots(int64_t n_values) {
  uint64_t t = 64;
  while (t < 2 * (uint64_t)n_values) t <<= 1;
  return t;
}
int nann_group_gather_unique_scratch_bytes(int64_t n_values, int64_t n_splits, int64_t* nbytes) {
  if (!nbytes || n_values < 0 || n_splits < 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_group_gather_unique_scratch_bytes: bad argument");
  if (n_values > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "GroupGather unique: more than 2^31 values");
  *nbytes = (int64_t)(ggu_table_slots(n_values) * 8 + (((uint64_t)n_values * 4 + 255) & ~255ull) + (uint64_t)(n_splits + 1) * 8);
  return NANN_OK;
}
int nann_group_gather_unique(const int32_t* values, int64_t n_values, const int64_t* row_splits, int64_t n_splits,
                             void* scratch, int32_t* out_values, int64_t* out_row_splits, int64_t* n_out,
                             nann_stream_t stream) {
  if (!n_out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_group_gather_unique: null out");
  if (n_splits < 1) return fail(NANN_ERR_BAD_ARGUMENT, "nann_group_gather_unique: row_splits of the unique=false result expected");
  if (n_values > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "GroupGather unique: more than 2^31 values");
  hipStream_t st = as_stream(stream);
  const long long n_groups = n_splits - 1;
  if (n_values == 0 || n_groups == 0) {  // every group empty: row_splits of zeros
    HIP_TRY(hipMemsetAsync(out_row_splits, 0, (size_t)n_splits * 8, st));
    *n_out = 0;
    return NANN_OK;
  }
  if (!scratch) return fail(NANN_ERR_BAD_ARGUMENT, "nann_group_gather_unique: null scratch");
  ResultBuf* rb;
  int rc = get_result_buf(&rb);
  if (rc) return rc;
  const uint64_t slots = ggu_table_slots(n_values);
  unsigned long long* table = static_cast<unsigned long long*>(scratch);
  uint32_t* grp = reinterpret_cast<uint32_t*>(table + slots);
  long long* counts = reinterpret_cast<long long*>(reinterpret_cast<unsigned char*>(grp) + (((uint64_t)n_values * 4 + 255) & ~255ull));
  const uint32_t mask = (uint32_t)(slots - 1);
  HIP_TRY(hipMemsetAsync(table, 0xff, slots * 8, st));
  const unsigned gblocks = (unsigned)std::min<long long>(n_groups, 4096);
  const unsigned pblocks = (unsigned)std::min<long long>((n_values + 255) / 256, 4096);
  hipLaunchKernelGGL(k_ggu_groups, dim3(gblocks), dim3(256), 0, st, row_splits, n_groups, grp);
  hipLaunchKernelGGL(k_ggu_insert, dim3(pblocks), dim3(256), 0, st, values, (long long)n_values, grp, table, mask);
  hipLaunchKernelGGL(k_ggu_count, dim3(gblocks), dim3(256), 0, st, values, row_splits, n_groups, grp, table, mask, counts);
  hipLaunchKernelGGL(k_ggu_scan, dim3(1), dim3(kNT), 0, st, counts, n_groups, out_row_splits, rb->dev);
  hipLaunchKernelGGL(k_ggu_emit, dim3(gblocks), dim3(256), 0, st, values, row_splits, n_groups, grp, table, mask,
                     out_row_splits, out_values);
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  *n_out = rb->host->n_out;
  return NANN_OK;
}

// ---- BitmapRefDifference -----------------------------------------------------
int nann_bitmap_ref_difference(const int32_t* values, int64_t n_values, const int64_t* row_splits,
                               int64_t n_splits, int32_t* idx_flag, int64_t n_flag_words,
                               int32_t* c_values, int64_t* c_row_splits, int64_t* n_out,
                               int64_t* n_out_splits, int32_t* ragged_code, nann_stream_t stream) {
  if (!n_out || !n_out_splits) return fail(NANN_ERR_BAD_ARGUMENT, "nann_bitmap_ref_difference: null out");
  if (n_values > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "more than 2^31 candidates");
  DeviceInfo di;
  int rc = device_info(&di);
  if (rc) return rc;
  ResultBuf* rb;
  rc = get_result_buf(&rb);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const size_t lds_need = align_up((size_t)n_flag_words * 4, 16) + 16;
  const bool lds = lds_need <= di.lds_max;
  uint32_t* flags = reinterpret_cast<uint32_t*>(idx_flag);
  if (lds) {
    auto kern = k_bitmap_ref_difference<true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kNT), lds_need, st, values, (long long)n_values, row_splits,
                       (long long)n_splits, flags, (long long)n_flag_words, c_values, c_row_splits,
                       rb->dev);
  } else {
    hipLaunchKernelGGL(k_bitmap_ref_difference<false>, dim3(1), dim3(kNT), 16, st, values,
                       (long long)n_values, row_splits, (long long)n_splits, flags,
                       (long long)n_flag_words, c_values, c_row_splits, rb->dev);
  }
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (ragged_code) *ragged_code = rb->host->code;
  *n_out = rb->host->n_out;
  *n_out_splits = rb->host->n_out_splits;
  if (rb->host->err) return fail(rb->host->err, "BitmapRefDifference: invalid input");
  return NANN_OK;
}

// ---- BloomFilterDifference ---------------------------------------------------------
static long long bloom_prime_below(long long num) {  // bitmap_ops.cc:384-403
  for (long long n = num; n > 1; --n) {
    bool prime = true;
    for (long long i = (long long)(std::sqrt((double)n) + 1e-6); i > 1; --i)
      if (n % i == 0) { prime = false; break; }
    if (prime) return n;
  }
  return 1;
}

int nann_bloom_filter_difference(const int32_t* values, int64_t n_values, const int64_t* row_splits,
                                 int64_t n_splits, int32_t* idx_flag, int64_t n_flag_words, int64_t bucket,
                                 int64_t bucket_size, int32_t* c_values, int64_t* c_row_splits, int64_t* n_out,
                                 int64_t* n_out_splits, int32_t* ragged_code, nann_stream_t stream) {
  if (!n_out || !n_out_splits) return fail(NANN_ERR_BAD_ARGUMENT, "nann_bloom_filter_difference: null out");
  if (bucket < 0 || bucket_size < 1 || n_flag_words < bucket_size)
    return fail(NANN_ERR_BAD_ARGUMENT, "BloomFilterDifference: bucket >= 0, bucket_size >= 1, idx_flag >= bucket_size words");
  if (bucket_size > (1ll << 26)) return fail(NANN_ERR_UNSUPPORTED, "BloomFilterDifference: filter beyond 2^31 bits");
  DeviceInfo di;
  int rc = device_info(&di);
  if (rc) return rc;
  ResultBuf* rb;
  rc = get_result_buf(&rb);
  if (rc) return rc;
  BloomParams B;
  B.bucket = bucket; B.bucket_size = bucket_size;
  const int modp[4] = {29, 47, 67, 83};  // multi_hash_mod_param, bitmap_ops.cc:297
  for (int l = 0; l < 4; ++l) B.prime[l] = (unsigned long long)bloom_prime_below((long long)modp[l] * bucket_size * 32);
  hipStream_t st = as_stream(stream);
  const size_t lds_need = align_up((size_t)n_flag_words * 4, 16);
  uint32_t* flags = reinterpret_cast<uint32_t*>(idx_flag);
  if (lds_need <= di.lds_max) {
    auto kern = k_bloom_filter_difference<true>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)std::max<size_t>(lds_need, 16)));
    hipLaunchKernelGGL(kern, dim3(1), dim3(kNT), std::max<size_t>(lds_need, 16), st, values, (long long)n_values,
                       row_splits, (long long)n_splits, flags, (long long)n_flag_words, B, c_values, c_row_splits, rb->dev);
  } else {
    hipLaunchKernelGGL(k_bloom_filter_difference<false>, dim3(1), dim3(kNT), 16, st, values, (long long)n_values,
                       row_splits, (long long)n_splits, flags, (long long)n_flag_words, B, c_values, c_row_splits, rb->dev);
  }
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (ragged_code) *ragged_code = rb->host->code;
  *n_out = rb->host->n_out;
  *n_out_splits = rb->host->n_out_splits;
  if (rb->host->err) return fail(rb->host->err, "BloomFilterDifference: invalid input");
  return NANN_OK;
}

// ---- GatherV2 ------------------------------------------------------------------
int nann_gather_rows(const void* params, int64_t n_rows, int64_t row_bytes, const int32_t* indices,
                     int64_t n_indices, void* out, int64_t* bad_i, nann_stream_t stream) {
  if (row_bytes <= 0 || row_bytes % 4) return fail(NANN_ERR_BAD_ARGUMENT, "row_bytes must be a multiple of 4");
  if (bad_i) *bad_i = -1;
  if (n_indices <= 0) return NANN_OK;
  ResultBuf* rb;
  int rc = get_result_buf(&rb);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_init_result, dim3(1), dim3(1), 0, st, rb->dev);
  const bool vec = row_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(params) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const long long words = vec ? row_bytes / 16 : row_bytes / 4;
  const long long total = n_indices * words;
  const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 8192);
  if (vec)
    hipLaunchKernelGGL(k_gather_rows<uint4>, dim3(blocks), dim3(256), 0, st,
                       static_cast<const uint4*>(params), (long long)n_rows, words, indices,
                       (long long)n_indices, static_cast<uint4*>(out), rb->dev);
  else
    hipLaunchKernelGGL(k_gather_rows<uint32_t>, dim3(blocks), dim3(256), 0, st,
                       static_cast<const uint32_t*>(params), (long long)n_rows, words, indices,
                       (long long)n_indices, static_cast<uint32_t*>(out), rb->dev);
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (rb->host->bad_i != 0x7fffffffffffffffll) {
    if (bad_i) *bad_i = rb->host->bad_i;
    return fail(NANN_ERR_INDEX_OUT_OF_RANGE, "indices[" + std::to_string(rb->host->bad_i) +
                                                 "] is not in [0, " + std::to_string(n_rows) + ")");
  }
  return NANN_OK;
}

// ---- TopKV2 ----------------------------------------------------------------------
int nann_topk(const float* values, int64_t n_rows, int64_t n_cols, int32_t k, float* out_values,
              int32_t* out_indices, nann_stream_t stream) {
  if (k < 0) return fail(NANN_ERR_BAD_ARGUMENT, "Need k >= 0, got " + std::to_string(k));  // :60-61
  if (n_cols < k)
    return fail(NANN_ERR_TOPK_K_GT_N, "input must have at least k columns. Had " +
                                          std::to_string(n_cols) + ", needed " + std::to_string(k));
  if (n_cols > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "n_cols too large");
  if (k == 0 || n_rows == 0) return NANN_OK;  // :84-85
  if (k > kMaxK) {  // whole-row sort in LDS
    if (n_cols > 16384) return fail(NANN_ERR_UNSUPPORTED, "k > 1024 needs n_cols <= 16384");
    int p2 = 2;
    while (p2 < n_cols) p2 <<= 1;
    auto kern = k_topk_sort;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                p2 * 8));
    hipLaunchKernelGGL(kern, dim3((unsigned)n_rows), dim3(kNT), (size_t)p2 * 8, as_stream(stream), values,
                       (long long)n_cols, (int)k, p2, out_values, out_indices);
    HIP_TRY(hipGetLastError());
    return NANN_OK;
  }
  hipLaunchKernelGGL(k_topk, dim3((unsigned)n_rows), dim3(kNT), 0, as_stream(stream), values,
                     (long long)n_cols, (int)k, out_values, out_indices);
  HIP_TRY(hipGetLastError());
  return NANN_OK;
}

// ---- scorer ------------------------------------------------------------------------
static float f16_bits_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, out;
  if (exp == 0) {
    if (man == 0) { out = sign; }
    else {  // subnormal: normalise
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    out = sign | 0x7f800000u | (man << 13);
  } else {
    out = sign | ((exp - 15 + 127) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &out, 4);
  return f;
}

// 2^7 v = hi + lo with hi = f16(2^7 v), lo = f16(2^7 v - hi)   (nann_mlp.h, split-f16 form: the scaling
// keeps lo a normal f16 for every weight above ~1e-3 in magnitude)
static void split_f16(float v, uint16_t* hi, uint16_t* lo) {
  const float sv = v * 128.0f;
  *hi = f32_to_f16_rne(sv);
  *lo = f32_to_f16_rne(sv - f16_bits_to_f32(*hi));
}

// A fragments of v_mfma_f32_32x32x16_f16 for the item half of W1 and for W2, hi and lo planes:
//   p1[t][kc][plane][lane][i] = W1[d + 16 kc + 8 g + i][32 t + j]                       (j = lane & 31, g = lane >> 5)
//   p2[t][q][m][plane][lane][i] = W2[32 t + (r & 3) + 8 (r >> 2) + 4 g][32 m + j], r = 8 q + i
// (layer 2's k order is the C/D register order of the finished layer-1 tile: nann_mlp.h)
static void pack_split_weights(const float* w1, const float* w2, int d, int h1, int h2, std::vector<uint16_t>* out,
                               size_t* p2_off) {
  const int T = h1 / 32, KC = d / 16, M = h2 / 32;
  const size_t n1 = (size_t)T * KC * 2 * 64 * 8, n2 = (size_t)T * 2 * M * 2 * 64 * 8;
  out->assign(n1 + n2, 0);
  *p2_off = n1;
  for (int t = 0; t < T; ++t)
    for (int kc = 0; kc < KC; ++kc)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int j = lane & 31, g = lane >> 5;
          const float v = w1[(size_t)(d + 16 * kc + 8 * g + i) * h1 + 32 * t + j];
          const size_t base = ((size_t)(t * KC + kc) * 2) * 64 * 8;
          split_f16(v, &(*out)[base + (size_t)lane * 8 + i], &(*out)[base + 64 * 8 + (size_t)lane * 8 + i]);
        }
  for (int t = 0; t < T; ++t)
    for (int q = 0; q < 2; ++q)
      for (int m = 0; m < M; ++m)
        for (int lane = 0; lane < 64; ++lane)
          for (int i = 0; i < 8; ++i) {
            const int j = lane & 31, g = lane >> 5, r = 8 * q + i;
            const int k = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * g;
            const float v = w2[(size_t)k * h2 + 32 * m + j];
            const size_t base = n1 + ((size_t)((t * 2 + q) * M + m) * 2) * 64 * 8;
            split_f16(v, &(*out)[base + (size_t)lane * 8 + i], &(*out)[base + 64 * 8 + (size_t)lane * 8 + i]);
          }
}

int nann_scorer_create(const nann_scorer_desc* desc, nann_scorer** out) {
  if (!desc || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_scorer_create: null argument");
  const int d = desc->d;
  if (!(d == 64 || d == 128 || d == 256 || d == 512))
    return fail(NANN_ERR_UNSUPPORTED, "embedding dim must be 64, 128, 256 or 512");
  if (desc->emb_dtype != NANN_F16 && desc->emb_dtype != NANN_BF16 && desc->emb_dtype != NANN_F32)
    return fail(NANN_ERR_UNSUPPORTED, "embedding dtype must be f16, bf16 or f32");
  if (desc->kind != NANN_SCORER_L2 && desc->kind != NANN_SCORER_MLP)
    return fail(NANN_ERR_BAD_ARGUMENT, "unknown scorer kind");
  nann_scorer* s = new nann_scorer();
  s->desc = *desc;
  if (desc->kind == NANN_SCORER_MLP) {
    if (desc->h1 != 256 || desc->h2 != 128 || d > 256) {
      delete s;
      return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: this build has the 2d-256-128-1 shape, d <= 256");
    }
    if (desc->emb_dtype == NANN_F32) {
      delete s;
      return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: item rows must be f16 or bf16");
    }
    if (!desc->w1 || !desc->b1 || !desc->alpha1 || !desc->w2 || !desc->b2 || !desc->alpha2 || !desc->w3) {
      delete s;
      return fail(NANN_ERR_BAD_ARGUMENT, "MLP scorer: null weight pointer");
    }
    const size_t h1 = 256, h2 = 128;
    const size_t n_w1 = 2 * (size_t)d * h1, n_w2 = h1 * h2;
    const size_t total = n_w1 + h1 + h1 + n_w2 + h2 + h2 + h2;
    std::vector<float> host(total);
    size_t o = 0;
    auto put = [&](const float* src, size_t cnt) { std::memcpy(host.data() + o, src, cnt * 4); o += cnt; return o - cnt; };
    const size_t o_w1 = put(desc->w1, n_w1), o_b1 = put(desc->b1, h1), o_a1 = put(desc->alpha1, h1);
    const size_t o_w2 = put(desc->w2, n_w2), o_b2 = put(desc->b2, h2), o_a2 = put(desc->alpha2, h2);
    const size_t o_w3 = put(desc->w3, h2);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->dev_weights), total * 4);
    if (e == hipSuccess) e = hipMemcpy(s->dev_weights, host.data(), total * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      nann_scorer_destroy(s);
      return fail(NANN_ERR_HIP, std::string("scorer weights: ") + hipGetErrorString(e));
    }
    const float* w = s->dev_weights;
    s->mlp.w1 = w + o_w1; s->mlp.b1 = w + o_b1; s->mlp.alpha1 = w + o_a1;
    s->mlp.w2 = w + o_w2; s->mlp.b2 = w + o_b2; s->mlp.alpha2 = w + o_a2; s->mlp.w3 = w + o_w3;
    s->mlp.d = d; s->mlp.h1 = 256; s->mlp.h2 = 128;
    if (desc->precision != NANN_MLP_PRECISION_DEFAULT && desc->precision != NANN_MLP_EXACT_F32 &&
        desc->precision != NANN_MLP_SPLIT_F16) {
      nann_scorer_destroy(s);
      return fail(NANN_ERR_BAD_ARGUMENT, "MLP scorer: unknown precision");
    }
    if (desc->precision != NANN_MLP_EXACT_F32) {  // the pre-scaled weights must stay inside f16's range
      float wmax = 0.0f;
      for (size_t i2 = (size_t)d * h1; i2 < n_w1; ++i2) wmax = std::max(wmax, std::fabs(desc->w1[i2]));
      for (size_t i2 = 0; i2 < n_w2; ++i2) wmax = std::max(wmax, std::fabs(desc->w2[i2]));
      const bool fits = wmax <= 511.0f;
      if (!fits && desc->precision == NANN_MLP_SPLIT_F16) {
        nann_scorer_destroy(s);
        return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: split-f16 precision needs |w| <= 511; use NANN_MLP_EXACT_F32");
      }
      s->desc.precision = fits ? NANN_MLP_SPLIT_F16 : NANN_MLP_EXACT_F32;  // DEFAULT resolved
    }
    {  // the split-f16 planes are small (256 KB): always built, used when precision says so
      std::vector<uint16_t> packed;
      size_t p2_off = 0;
      pack_split_weights(desc->w1, desc->w2, d, 256, 128, &packed, &p2_off);
      e = hipMalloc(reinterpret_cast<void**>(&s->dev_packed), packed.size() * 2);
      if (e == hipSuccess) e = hipMemcpy(s->dev_packed, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        nann_scorer_destroy(s);
        return fail(NANN_ERR_HIP, std::string("scorer weights: ") + hipGetErrorString(e));
      }
      s->mlp.p1 = s->dev_packed;
      s->mlp.p2 = s->dev_packed + p2_off / 8;
    }
    {  // W2 as f32 A fragments of the exact form's resident layer 2 (nann_mlp5.h): p2x[t][mt][j][lane][i] =
       // W2[32 t + i + 8 j + 4 (lane >> 5)][32 mt + (lane & 31)] -- step r = 4 j + i of tile t in ORDER_H
      std::vector<float> px((size_t)256 * 128);
      for (int t = 0; t < 8; ++t)
        for (int mt = 0; mt < 4; ++mt)
          for (int j = 0; j < 4; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int i = 0; i < 4; ++i)
                px[((((size_t)t * 4 + mt) * 4 + j) * 64 + lane) * 4 + i] =
                    desc->w2[(size_t)(32 * t + i + 8 * j + 4 * (lane >> 5)) * 128 + 32 * mt + (lane & 31)];
      e = hipMalloc(reinterpret_cast<void**>(&s->dev_packed_x), px.size() * 4);
      if (e == hipSuccess) e = hipMemcpy(s->dev_packed_x, px.data(), px.size() * 4, hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        nann_scorer_destroy(s);
        return fail(NANN_ERR_HIP, std::string("scorer weights: ") + hipGetErrorString(e));
      }
      s->mlp.p2x = s->dev_packed_x;
    }
    // the host pointers of the descriptor are not kept
    s->desc.w1 = s->desc.b1 = s->desc.alpha1 = s->desc.w2 = s->desc.b2 = s->desc.alpha2 = s->desc.w3 = nullptr;
  }
  *out = s;
  return NANN_OK;
}

void nann_scorer_destroy(nann_scorer* s) {
  if (!s) return;
  if (s->dev_weights) (void)hipFree(s->dev_weights);
  if (s->dev_packed) (void)hipFree(s->dev_packed);
  if (s->dev_packed_x) (void)hipFree(s->dev_packed_x);
  delete s;
}

int nann_user_seq_mean(const void* comm_seq_f16, int64_t n_queries, int32_t seq_len, int32_t d,
                       float* q, nann_stream_t stream) {
  if (n_queries <= 0) return NANN_OK;
  if (seq_len <= 0 || d <= 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_user_seq_mean: bad shape");
  const long long bytes = (long long)seq_len * d * 2;
  if (d % 8 == 0 && bytes <= kSeqMeanMaxBytes && (reinterpret_cast<uintptr_t>(comm_seq_f16) & 15) == 0) {
    const size_t lds = (size_t)bytes + (size_t)seq_len * (d / 8);  // the history + one flag byte per 16-byte piece
    hipLaunchKernelGGL(k_user_seq_mean_lds, dim3((unsigned)n_queries), dim3(kSeqMeanThreads), lds, as_stream(stream),
                       static_cast<const uint4*>(comm_seq_f16), (int)seq_len, (int)d, q);
  } else {
    hipLaunchKernelGGL(k_user_seq_mean, dim3((unsigned)n_queries), dim3(std::min(256, (d + 63) / 64 * 64)),
                       0, as_stream(stream), static_cast<const uint16_t*>(comm_seq_f16), (int)seq_len,
                       (int)d, q);
  }
  HIP_TRY(hipGetLastError());
  return NANN_OK;
}

}  // extern "C"

template <int LPR>
static void launch_score(int dt, unsigned blocks, hipStream_t st, const void* table, long long nt, int d,
                         const int32_t* idx, long long n, const float* q, float* out, OpResult* res) {
  if (dt == NANN_F16)
    hipLaunchKernelGGL((k_score_l2<LPR, DT_F16>), dim3(blocks), dim3(256), 0, st, table, nt, d, idx, n, q, out, res);
  else if (dt == NANN_BF16)
    hipLaunchKernelGGL((k_score_l2<LPR, DT_BF16>), dim3(blocks), dim3(256), 0, st, table, nt, d, idx, n, q, out, res);
  else
    hipLaunchKernelGGL((k_score_l2<LPR, DT_F32>), dim3(blocks), dim3(256), 0, st, table, nt, d, idx, n, q, out, res);
}

extern "C" {

int nann_score(const nann_scorer* scorer, const float* q, const void* table, int64_t n_table_rows,
               const int32_t* indices, int64_t n, float* out_scores, int64_t* bad_i,
               nann_stream_t stream) {
  if (!scorer) return fail(NANN_ERR_BAD_ARGUMENT, "nann_score: null scorer");
  if (bad_i) *bad_i = -1;
  if (n <= 0)  // blaze_xla_predictor.cc:259-263
    return fail(NANN_ERR_EMPTY_SCORE_BATCH, "Error when getting input address or size");
  ResultBuf* rb;
  int rc = get_result_buf(&rb);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_init_result, dim3(1), dim3(1), 0, st, rb->dev);
  const int d = scorer->desc.d, lpr = d / 8, dt = scorer->desc.emb_dtype;
  if (scorer->desc.kind == NANN_SCORER_MLP) {
    const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 512);  // (each takes a contiguous run of passes)
    const MlpParams& P = scorer->mlp;
    const int split = scorer->desc.precision == NANN_MLP_SPLIT_F16;
    if (d == 64) rc = launch_score_mlp_d64(dt, split, blocks, st, P, table, n_table_rows, indices, n, q, out_scores, rb->dev);
    else if (d == 128) rc = launch_score_mlp_d128(dt, split, blocks, st, P, table, n_table_rows, indices, n, q, out_scores, rb->dev);
    else rc = launch_score_mlp_d256(dt, split, blocks, st, P, table, n_table_rows, indices, n, q, out_scores, rb->dev);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    rc = fetch_result(rb, st);
    if (rc) return rc;
    if (rb->host->bad_i != 0x7fffffffffffffffll) {
      if (bad_i) *bad_i = rb->host->bad_i;
      return fail(NANN_ERR_INDEX_OUT_OF_RANGE, "indices[" + std::to_string(rb->host->bad_i) +
                                                   "] is not in [0, " + std::to_string(n_table_rows) + ")");
    }
    return NANN_OK;
  }
  const long long rows_per_block = 4 * (64 / lpr);
  const unsigned blocks = (unsigned)std::min<long long>((n + rows_per_block - 1) / rows_per_block, 8192);
  switch (lpr) {
    case 8: launch_score<8>(dt, blocks, st, table, n_table_rows, d, indices, n, q, out_scores, rb->dev); break;
    case 16: launch_score<16>(dt, blocks, st, table, n_table_rows, d, indices, n, q, out_scores, rb->dev); break;
    case 32: launch_score<32>(dt, blocks, st, table, n_table_rows, d, indices, n, q, out_scores, rb->dev); break;
    default: launch_score<64>(dt, blocks, st, table, n_table_rows, d, indices, n, q, out_scores, rb->dev); break;
  }
  HIP_TRY(hipGetLastError());
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (rb->host->bad_i != 0x7fffffffffffffffll) {
    if (bad_i) *bad_i = rb->host->bad_i;
    return fail(NANN_ERR_INDEX_OUT_OF_RANGE, "indices[" + std::to_string(rb->host->bad_i) +
                                                 "] is not in [0, " + std::to_string(n_table_rows) + ")");
  }
  return NANN_OK;
}

// ---- 8(f2): the reference scorer model -------------------------------------------------
}  // extern "C"

// A fragments of v_mfma_f32_32x32x16_f16 for W [K][n_out] (row-major), hi and lo planes of W * 2^7:
//   out[m][kc][plane][lane][i] = W[krow(kc, lane >> 5, i)][32 m + (lane & 31)]
// natural k order (the operand is a table row): krow = 16 kc + 8 g + i
// cd order (the operand is a finished 32-unit tile in C/D register order, two chunks per tile):
//   krow = 32 (kc >> 1) + (i & 3) + 16 (kc & 1) + 8 (i >> 2) + 4 g                      (nann_attn_split.h)
static bool pack_attn_frags(const float* W, int n_out, int n_chunks, bool cd_order, std::vector<uint16_t>* out) {
  const int M = n_out / 32;
  const size_t base0 = out->size();
  out->resize(base0 + (size_t)M * n_chunks * 2 * 64 * 8);
  bool ok = true;
  for (int m = 0; m < M; ++m)
    for (int kc = 0; kc < n_chunks; ++kc)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int g = lane >> 5;
          const int k = cd_order ? 32 * (kc >> 1) + (i & 3) + 16 * (kc & 1) + 8 * (i >> 2) + 4 * g : 16 * kc + 8 * g + i;
          const float v = W[(size_t)k * n_out + 32 * m + (lane & 31)];
          if (!(std::fabs(v) <= 511.0f)) ok = false;  // v * 2^7 must stay inside f16
          const size_t base = base0 + ((size_t)(m * n_chunks + kc) * 2) * 64 * 8;
          split_f16(v, &(*out)[base + (size_t)lane * 8 + i], &(*out)[base + 64 * 8 + (size_t)lane * 8 + i]);
        }
  return ok;
}

extern "C" {

int nann_attn_scorer_create(const nann_attn_desc* desc, nann_attn_scorer** out) {
  if (!desc || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_attn_scorer_create: null argument");
  const int d = desc->d, L = desc->seq_len;
  if (d != 64 && d != 128) return fail(NANN_ERR_UNSUPPORTED, "attention scorer: d must be 64 or 128");
  if (L <= 0 || L > kAttnLP) return fail(NANN_ERR_UNSUPPORTED, "attention scorer: seq_len must be in [1, 64]");
  if (desc->emb_dtype != NANN_F16 && desc->emb_dtype != NANN_BF16)
    return fail(NANN_ERR_UNSUPPORTED, "attention scorer: item rows must be f16 or bf16");
  const float* src[] = {desc->wq1, desc->bq1, desc->aq, desc->wq2, desc->bq2, desc->wk1, desc->bk1, desc->ak,
                        desc->wk2, desc->bk2,
                        desc->w[0], desc->b[0], desc->bn_scale[0], desc->bn_shift[0], desc->alpha[0],
                        desc->w[1], desc->b[1], desc->bn_scale[1], desc->bn_shift[1], desc->alpha[1],
                        desc->w[2], desc->b[2], desc->bn_scale[2], desc->bn_shift[2], desc->alpha[2],
                        desc->w[3]};
  const size_t cnt[] = {(size_t)d * 128, 128, 128, 128 * 256, 256, 64 * 128, 128, 128, 128 * 256, 256,
                        (size_t)(64 + d) * 128, 128, 128, 128, 128,
                        128 * 64, 64, 64, 64, 64,
                        64 * 32, 32, 32, 32, 32,
                        32};
  constexpr int NV = sizeof(cnt) / sizeof(cnt[0]);
  size_t off[NV], total = 0;
  for (int i = 0; i < NV; ++i) {
    if (!src[i]) return fail(NANN_ERR_BAD_ARGUMENT, "attention scorer: null weight pointer");
    off[i] = total;
    total += (cnt[i] + 3) & ~(size_t)3;  // keep every vector 16-byte aligned (float4 staging)
  }
  std::vector<float> host(total, 0.0f);
  for (int i = 0; i < NV; ++i) std::memcpy(host.data() + off[i], src[i], cnt[i] * 4);
  nann_attn_scorer* s = new nann_attn_scorer();
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->dev_weights), total * 4);
  if (e == hipSuccess) e = hipMemcpy(s->dev_weights, host.data(), total * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    nann_attn_scorer_destroy(s);
    return fail(NANN_ERR_HIP, std::string("attention scorer weights: ") + hipGetErrorString(e));
  }
  const float* w = s->dev_weights;
  AttnParams& P = s->P;
  const float** dst[] = {&P.wq1, &P.bq1, &P.aq, &P.wq2, &P.bq2, &P.wk1, &P.bk1, &P.ak, &P.wk2, &P.bk2,
                         &P.w1, &P.b1, &P.s1, &P.t1, &P.a1, &P.w2, &P.b2, &P.s2, &P.t2, &P.a2,
                         &P.w3, &P.b3, &P.s3, &P.t3, &P.a3, &P.w4};
  for (int i = 0; i < NV; ++i) *dst[i] = w + off[i];
  P.d = d;
  P.L = L;
  s->emb_dtype = desc->emb_dtype;
  if (desc->precision != NANN_MLP_PRECISION_DEFAULT && desc->precision != NANN_MLP_EXACT_F32 &&
      desc->precision != NANN_MLP_SPLIT_F16) {
    nann_attn_scorer_destroy(s);
    return fail(NANN_ERR_BAD_ARGUMENT, "attention scorer: unknown precision");
  }
  s->precision = desc->precision;  // DEFAULT is resolved below, once the weights have been looked at
  {  // the split-f16 planes (~0.5 MB) are always built; `precision` picks the kernels
    std::vector<uint16_t> packed;
    bool ok = true;
    const size_t o_q1 = packed.size(); ok &= pack_attn_frags(desc->wq1, 128, d / 16, false, &packed);
    const size_t o_q2 = packed.size(); ok &= pack_attn_frags(desc->wq2, 256, 8, true, &packed);
    const size_t o_1a = packed.size(); ok &= pack_attn_frags(desc->w[0], 128, 4, true, &packed);                    // rows of a
    const size_t o_1e = packed.size(); ok &= pack_attn_frags(desc->w[0] + (size_t)64 * 128, 128, d / 16, false, &packed);  // rows of e
    const size_t o_2 = packed.size(); ok &= pack_attn_frags(desc->w[1], 64, 8, true, &packed);
    const size_t o_3 = packed.size(); ok &= pack_attn_frags(desc->w[2], 32, 4, true, &packed);
    const size_t o_v = (packed.size() + 7) & ~(size_t)7;  // halves; 16-byte aligned
    if (!ok && s->precision == NANN_MLP_SPLIT_F16) {
      nann_attn_scorer_destroy(s);
      return fail(NANN_ERR_UNSUPPORTED, "attention scorer, split-f16 form: |w| must be <= 511");
    }
    if (s->precision == NANN_MLP_PRECISION_DEFAULT) s->precision = ok ? NANN_MLP_SPLIT_F16 : NANN_MLP_EXACT_F32;
    std::vector<float> pv(PV_COUNT, 0.0f);
    const float WS = kAttnWS, HS = kAttnHS;
    for (int j = 0; j < 128; ++j) {
      pv[PV_BQ1 + j] = desc->bq1[j] * WS;
      pv[PV_AQ + j] = desc->aq[j] * (HS / WS);
      pv[PV_B1 + j] = desc->b[0][j] * (WS * HS);
      // bn + prelu of a hidden layer as w = fma(acc, S, T) (= 2^4 x the normalised value: power-of-two scales are exact)
      // and w + A min(w, 0): S = scale / 2^7, T = shift x 2^4, A = alpha - 1
      pv[PV_S1 + j] = desc->bn_scale[0][j] / WS;
      pv[PV_T1 + j] = desc->bn_shift[0][j] * HS;
      pv[PV_A1 + j] = desc->alpha[0][j] - 1.0f;
    }
    for (int j = 0; j < 256; ++j) pv[PV_BQ2 + j] = desc->bq2[j] * (WS * HS);
    for (int j = 0; j < 64; ++j) {
      pv[PV_B2 + j] = desc->b[1][j] * (WS * HS);
      pv[PV_S2 + j] = desc->bn_scale[1][j] / WS;
      pv[PV_T2 + j] = desc->bn_shift[1][j] * HS;
      pv[PV_A2 + j] = desc->alpha[1][j] - 1.0f;
    }
    for (int j = 0; j < 32; ++j) {
      pv[PV_B3 + j] = desc->b[2][j] * (WS * HS);
      pv[PV_S3 + j] = desc->bn_scale[2][j] / (WS * HS);
      pv[PV_T3 + j] = desc->bn_shift[2][j];
      pv[PV_A3 + j] = desc->alpha[2][j];
      pv[PV_W4 + j] = desc->w[3][j];
    }
    const size_t bytes = o_v * 2 + pv.size() * 4;
    e = hipMalloc(reinterpret_cast<void**>(&s->dev_packed), bytes);
    if (e == hipSuccess) e = hipMemcpy(s->dev_packed, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = hipMemcpy(reinterpret_cast<char*>(s->dev_packed) + o_v * 2, pv.data(), pv.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      nann_attn_scorer_destroy(s);
      return fail(NANN_ERR_HIP, std::string("attention scorer weights: ") + hipGetErrorString(e));
    }
    const uint4* base = s->dev_packed;
    P.pq1 = base + o_q1 / 8; P.pq2 = base + o_q2 / 8; P.pw1a = base + o_1a / 8; P.pw1e = base + o_1e / 8;
    P.pw2 = base + o_2 / 8; P.pw3 = base + o_3 / 8;
    P.pvec = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + o_v * 2);
  }
  *out = s;
  return NANN_OK;
}

void nann_attn_scorer_destroy(nann_attn_scorer* s) {
  if (!s) return;
  if (s->dev_weights) (void)hipFree(s->dev_weights);
  if (s->dev_packed) (void)hipFree(s->dev_packed);
  delete s;
}

int nann_attn_prepare(const nann_attn_scorer* s, const void* user_seq_f16, int64_t n_users, float* kt,
                      float* upad, nann_stream_t stream) {
  if (!s || !user_seq_f16 || !kt || !upad) return fail(NANN_ERR_BAD_ARGUMENT, "nann_attn_prepare: null argument");
  if (n_users <= 0) return NANN_OK;
  if (s->precision == NANN_MLP_SPLIT_F16)
    return launch_attn_prepare_split(as_stream(stream), s->P, user_seq_f16, n_users, kt, upad);
  return launch_attn_prepare(as_stream(stream), s->P, user_seq_f16, n_users, kt, upad);
}

int nann_attn_score(const nann_attn_scorer* s, const float* kt, const float* upad, const void* table,
                    int64_t n_table_rows, const int32_t* indices, int64_t n, float* out_scores,
                    int64_t* bad_i, nann_stream_t stream) {
  if (!s || !kt || !upad || !table) return fail(NANN_ERR_BAD_ARGUMENT, "nann_attn_score: null argument");
  if (bad_i) *bad_i = -1;
  if (n <= 0)  // blaze_xla_predictor.cc:259-263
    return fail(NANN_ERR_EMPTY_SCORE_BATCH, "Error when getting input address or size");
  ResultBuf* rb;
  int rc = get_result_buf(&rb);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(k_init_result, dim3(1), dim3(1), 0, st, rb->dev);
  const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 2048);
  if (s->precision == NANN_MLP_SPLIT_F16)
    rc = launch_score_attn_split(s->emb_dtype, std::min(blocks, 512u), st, s->P, kt, upad, table, n_table_rows, indices, n,
                                 out_scores, &rb->dev->bad_i);
  else
    rc = launch_score_attn(s->emb_dtype, blocks, st, s->P, kt, upad, table, n_table_rows, indices, n, out_scores,
                           &rb->dev->bad_i);
  if (rc) return rc;
  rc = fetch_result(rb, st);
  if (rc) return rc;
  if (rb->host->bad_i != 0x7fffffffffffffffll) {
    if (bad_i) *bad_i = rb->host->bad_i;
    return fail(NANN_ERR_INDEX_OUT_OF_RANGE, "indices[" + std::to_string(rb->host->bad_i) +
                                                 "] is not in [0, " + std::to_string(n_table_rows) + ")");
  }
  return NANN_OK;
}

// ---- a4: the scoring model behind BlazeXlaOp, loaded from a weights directory ------------
struct nann_model {
  int kind = NANN_MODEL_L2;
  int d = 0, seq_len = 0, emb_dtype = NANN_F16;
  nann_scorer* scorer = nullptr;
  nann_attn_scorer* attn = nullptr;
};

static int load_f32(const std::string& dir, const char* name, std::vector<std::vector<char>>* keep,
                    const float** out, int64_t expect_count) {
  keep->emplace_back();
  std::vector<int64_t> shape;
  const int rc = read_npy_host((dir + "/" + name + ".npy").c_str(), NANN_F32, nullptr, 0, /*allow_cast=*/1,
                               &keep->back(), &shape);
  if (rc) return rc;
  int64_t count = 1;
  for (int64_t v : shape) count *= v;
  if (expect_count >= 0 && count != expect_count)
    return fail(NANN_ERR_SHAPE_MISMATCH, std::string(name) + ".npy: " + std::to_string(count) + " values, expected " +
                                             std::to_string(expect_count));
  *out = reinterpret_cast<const float*>(keep->back().data());
  return NANN_OK;
}

}  // extern "C"

// optional precision.txt of a weights directory: "split" (split-f16 operands, the default) | "exact" (f32-input MFMA)
static int read_precision(const std::string& dir, int32_t* precision) {
  std::ifstream pf(dir + "/precision.txt");
  std::string prec;
  if (!pf || !(pf >> prec)) return NANN_OK;
  if (prec == "split") *precision = NANN_MLP_SPLIT_F16;
  else if (prec == "exact") *precision = NANN_MLP_EXACT_F32;
  else return fail(NANN_ERR_BAD_ARGUMENT, "precision.txt: expected exact or split, got '" + prec + "'");
  return NANN_OK;
}

extern "C" {

// BlazeXlaOp.graph_def as the reference writes it: a frozen GraphDef file (convert_meta.py:361-398).  The weights of
// Model.forward (model.py:189-233) are pulled out of it by host/nann_graphdef.h; nothing of the graph is executed.
static int model_from_graphdef(const std::string& path, int32_t d, int32_t emb_dtype, int32_t seq_len, nann_model* m) {
  std::ifstream f(path, std::ifstream::binary);
  if (!f) return fail(NANN_ERR_IO, "Fail to open file: " + path);
  const std::string bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  nann_gd::Graph g;
  std::string msg;
  // text first, then binary: the reference's order (blaze_xla_kernel.cc:169-175)
  if (!nann_gd::parse_graph_any(reinterpret_cast<const uint8_t*>(bytes.data()), bytes.size(), &g, &msg))
    return fail(NANN_ERR_IO, "parse proto from " + path + " failed: " + msg);  // blaze_xla_kernel.cc:169-175
  nann_gd::AttnWeights w;
  if (!nann_gd::extract_attention(g, &w, &msg))
    return fail(NANN_ERR_UNSUPPORTED, path + ": not the reference's scorer model (model.py:189-233): " + msg);
  if (w.d != d) return fail(NANN_ERR_SHAPE_MISMATCH, path + ": the model scores " + std::to_string(w.d) + "-d item rows, the request has " + std::to_string(d));
  if (w.e != kAttnE) return fail(NANN_ERR_UNSUPPORTED, path + ": user sequence embedding dim " + std::to_string(w.e) + ", this build has 64");
  const size_t hq = 2 * (size_t)kAttnE, hq2 = 4 * (size_t)kAttnE;
  const size_t expect[] = {(size_t)d * hq, hq, hq, hq * hq2, hq2, (size_t)kAttnE * hq, hq, hq, hq * hq2, hq2};
  const std::vector<float>* att[] = {&w.wq1, &w.bq1, &w.aq, &w.wq2, &w.bq2, &w.wk1, &w.bk1, &w.ak, &w.wk2, &w.bk2};
  for (int i = 0; i < 10; ++i)
    if (att[i]->size() != expect[i]) return fail(NANN_ERR_SHAPE_MISMATCH, path + ": attention tensor " + std::to_string(i) + " has an unexpected size");
  const size_t widths[4] = {128, 64, 32, 1};
  size_t n_in = (size_t)kAttnE + (size_t)d;
  for (int l = 0; l < 4; ++l) {
    if (w.w[l].size() != n_in * widths[l]) return fail(NANN_ERR_SHAPE_MISMATCH, path + ": DNN layer " + std::to_string(l + 1) + " is not the 128-64-32-1 tower this build has");
    n_in = widths[l];
  }
  nann_attn_desc ad = {};
  ad.d = d; ad.emb_dtype = emb_dtype; ad.seq_len = seq_len;
  ad.wq1 = w.wq1.data(); ad.bq1 = w.bq1.data(); ad.aq = w.aq.data(); ad.wq2 = w.wq2.data(); ad.bq2 = w.bq2.data();
  ad.wk1 = w.wk1.data(); ad.bk1 = w.bk1.data(); ad.ak = w.ak.data(); ad.wk2 = w.wk2.data(); ad.bk2 = w.bk2.data();
  for (int l = 0; l < 4; ++l) ad.w[l] = w.w[l].data();
  for (int l = 0; l < 3; ++l) {
    ad.b[l] = w.b[l].data(); ad.bn_scale[l] = w.bn_scale[l].data(); ad.bn_shift[l] = w.bn_shift[l].data();
    ad.alpha[l] = w.alpha[l].data();
  }
  ad.precision = NANN_MLP_PRECISION_DEFAULT;
  {  // an optional <file>.precision beside the graph: "split" | "exact" (as precision.txt of the directory form)
    std::ifstream pf(path + ".precision");
    std::string prec;
    if (pf && (pf >> prec)) {
      if (prec == "split") ad.precision = NANN_MLP_SPLIT_F16;
      else if (prec == "exact") ad.precision = NANN_MLP_EXACT_F32;
      else return fail(NANN_ERR_BAD_ARGUMENT, path + ".precision: expected exact or split, got '" + prec + "'");
    }
  }
  m->kind = NANN_MODEL_ATTENTION;
  return nann_attn_scorer_create(&ad, &m->attn);
}

int nann_model_load(const char* dir, int32_t d, int32_t emb_dtype, int32_t seq_len, nann_model** out) {
  if (!dir || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_load: null argument");
  const std::string D(dir);
  nann_model* m = new nann_model();
  m->d = d; m->seq_len = seq_len; m->emb_dtype = emb_dtype;
  struct stat sb;
  if (::stat(dir, &sb) == 0 && S_ISREG(sb.st_mode)) {  // a frozen GraphDef file: what the reference's attr names
    const int rc = model_from_graphdef(D, d, emb_dtype, seq_len, m);
    if (rc) { nann_model_destroy(m); return rc; }
    *out = m;
    return NANN_OK;
  }
  std::ifstream kf(D + "/scorer.txt");
  if (!kf) { nann_model_destroy(m); return fail(NANN_ERR_IO, "Fail to open file: " + D + "/scorer.txt"); }
  std::string kind;
  kf >> kind;
  std::vector<std::vector<char>> keep;
  int rc = NANN_OK;
  if (kind == "l2" || kind == "mlp") {
    nann_scorer_desc sd = {};
    sd.kind = kind == "l2" ? NANN_SCORER_L2 : NANN_SCORER_MLP;
    sd.d = d; sd.emb_dtype = emb_dtype;
    m->kind = kind == "l2" ? NANN_MODEL_L2 : NANN_MODEL_MLP;
    if (kind == "mlp") {
      sd.h1 = 256; sd.h2 = 128;
      const struct { const char* n; const float** p; int64_t cnt; } w[] = {
          {"w1", &sd.w1, 2ll * d * 256}, {"b1", &sd.b1, 256}, {"alpha1", &sd.alpha1, 256}, {"w2", &sd.w2, 256 * 128},
          {"b2", &sd.b2, 128}, {"alpha2", &sd.alpha2, 128}, {"w3", &sd.w3, 128}};
      for (const auto& e : w)
        if (!rc) rc = load_f32(D, e.n, &keep, e.p, e.cnt);
      if (!rc) rc = read_precision(D, &sd.precision);
    }
    if (!rc) rc = nann_scorer_create(&sd, &m->scorer);
  } else if (kind == "attention") {
    nann_attn_desc ad = {};
    ad.d = d; ad.emb_dtype = emb_dtype; ad.seq_len = seq_len;
    m->kind = NANN_MODEL_ATTENTION;
    // every tensor with the element count nann_attn_scorer_create indexes it by: a directory exported for another d,
    // or a truncated file, is refused instead of read out of bounds
    const int64_t E = kAttnE, hq = 2 * E, hq2 = 4 * E;
    const struct { const char* n; const float** p; int64_t cnt; } w[] = {
        {"wq1", &ad.wq1, (int64_t)d * hq}, {"bq1", &ad.bq1, hq}, {"aq", &ad.aq, hq}, {"wq2", &ad.wq2, hq * hq2}, {"bq2", &ad.bq2, hq2},
        {"wk1", &ad.wk1, E * hq}, {"bk1", &ad.bk1, hq}, {"ak", &ad.ak, hq}, {"wk2", &ad.wk2, hq * hq2}, {"bk2", &ad.bk2, hq2},
        {"w0", &ad.w[0], (E + d) * 128}, {"w1", &ad.w[1], 128 * 64}, {"w2", &ad.w[2], 64 * 32}, {"w3", &ad.w[3], 32},
        {"b0", &ad.b[0], 128}, {"b1", &ad.b[1], 64}, {"b2", &ad.b[2], 32},
        {"bn_scale0", &ad.bn_scale[0], 128}, {"bn_scale1", &ad.bn_scale[1], 64}, {"bn_scale2", &ad.bn_scale[2], 32},
        {"bn_shift0", &ad.bn_shift[0], 128}, {"bn_shift1", &ad.bn_shift[1], 64}, {"bn_shift2", &ad.bn_shift[2], 32},
        {"alpha0", &ad.alpha[0], 128}, {"alpha1", &ad.alpha[1], 64}, {"alpha2", &ad.alpha[2], 32}};
    for (const auto& e : w)
      if (!rc) rc = load_f32(D, e.n, &keep, e.p, e.cnt);
    if (!rc) rc = read_precision(D, &ad.precision);
    if (!rc) rc = nann_attn_scorer_create(&ad, &m->attn);
  } else {
    rc = fail(NANN_ERR_UNSUPPORTED, "scorer.txt: expected l2, mlp or attention, got '" + kind + "'");
  }
  if (rc) { nann_model_destroy(m); return rc; }
  *out = m;
  return NANN_OK;
}

// BlazeXlaOp.blaze_option_path as BlazeXlaOp::ParseAttr reads it (blaze_xla_kernel.cc:156-167): a text-format
// BlazeKernelOptions file, else the attr string itself as text format (host/nann_blaze_options.h).
int nann_blaze_options_parse(const char* attr, nann_blaze_options* out) {
  if (!attr || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_blaze_options_parse: null argument");
  nann_gd::BlazeOptions o;
  std::string msg;
  if (!nann_gd::parse_blaze_options_attr(attr, &o, &msg)) return fail(NANN_ERR_IO, msg);
  nann_blaze_options r = {};
  r.struct_bytes = (int32_t)sizeof(nann_blaze_options);
  r.wait_ms = o.wait_ms; r.run_mode = o.run_mode; r.xla_compilation = o.xla_compilation;
  r.auto_mixed_precision = o.auto_mixed_precision; r.disable_output_padding = o.disable_output_padding;
  r.n_warmup_batchsize = o.n_warmup_batchsize; r.max_warmup_batchsize = o.max_warmup_batchsize; r.from_file = o.from_file;
  *out = r;
  return NANN_OK;
}

void nann_model_destroy(nann_model* m) {
  if (!m) return;
  if (m->scorer) nann_scorer_destroy(m->scorer);
  if (m->attn) nann_attn_scorer_destroy(m->attn);
  delete m;
}

int nann_model_kind(const nann_model* m) { return m ? m->kind : -1; }
const nann_scorer* nann_model_scorer(const nann_model* m) { return m ? m->scorer : nullptr; }

int nann_model_workspace_bytes(const nann_model* m, int64_t* nbytes) {
  if (!m || !nbytes) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_workspace_bytes: null argument");
  *nbytes = m->kind == NANN_MODEL_ATTENTION ? (int64_t)(256 * 64 + 64 * 64) * 4 : (int64_t)kMaxD * 4;
  return NANN_OK;
}

int nann_model_forward(const nann_model* m, const void* user_seq_f16, const void* item_emb, int64_t n,
                       float* logits, void* workspace, nann_stream_t stream) {
  if (!m || !user_seq_f16 || !logits || !workspace) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_forward: null argument");
  if (n <= 0)  // blaze_xla_predictor.cc:259-263
    return fail(NANN_ERR_EMPTY_SCORE_BATCH, "Error when getting input address or size");
  if (m->kind == NANN_MODEL_ATTENTION) {
    float* kt = static_cast<float*>(workspace);
    float* upad = kt + 256 * 64;
    const int rc = nann_attn_prepare(m->attn, user_seq_f16, 1, kt, upad, stream);
    if (rc) return rc;
    return nann_attn_score(m->attn, kt, upad, item_emb, n, nullptr, n, logits, nullptr, stream);
  }
  float* q = static_cast<float*>(workspace);
  const int rc = nann_user_seq_mean(user_seq_f16, 1, m->seq_len, m->d, q, stream);
  if (rc) return rc;
  return nann_score(m->scorer, q, item_emb, n, nullptr, n, logits, nullptr, stream);
}

// ---- index ---------------------------------------------------------------------------
static int64_t dtype_bytes(int dt) { return dt == NANN_F32 ? 4 : 2; }

int nann_index_create(const nann_index_desc* desc, nann_index** out) {
  if (!desc || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_index_create: null argument");
  const nann_index_desc& h = *desc;
  if (h.n_items <= 0 || h.n_items > 0x7fffffffll) return fail(NANN_ERR_BAD_ARGUMENT, "n_items out of range");
  if (!(h.d == 64 || h.d == 128 || h.d == 256 || h.d == 512))
    return fail(NANN_ERR_UNSUPPORTED, "embedding dim must be 64, 128, 256 or 512");
  if (h.n_enter < 0 || h.n_enter > 0x7fffffffll) return fail(NANN_ERR_BAD_ARGUMENT, "n_enter out of range");
  for (int l = 0; l < 2; ++l)
    if (h.nb_nnz[l] < 0 || h.nb_nnz[l] > 0xffffffffll)
      return fail(NANN_ERR_UNSUPPORTED, "a level holds more than 2^32-1 links");
  nann_index* ix = new nann_index();
  ix->desc = h;
  {
    static std::atomic<uint64_t> next_uid{1};
    ix->uid = next_uid.fetch_add(1);
  }
  const int64_t N = h.n_items;
  // host copies of the small arrays for validation
  std::vector<int64_t> rs_host[2];
  std::vector<int32_t> ep_host((size_t)h.n_enter);
  auto fetch = [&](void* dst, const void* src, size_t bytes) -> int {
    if (!bytes) return NANN_OK;
    if (h.on_device) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); }
    else std::memcpy(dst, src, bytes);
    return NANN_OK;
  };
  int rc = NANN_OK;
  for (int l = 0; l < 2 && !rc; ++l) {
    rs_host[l].resize((size_t)N + 1);
    rc = fetch(rs_host[l].data(), h.nb_row_splits[l], (size_t)(N + 1) * 8);
  }
  if (!rc) rc = fetch(ep_host.data(), h.enter_points, (size_t)h.n_enter * 4);
  if (rc) { delete ix; return rc; }
  for (int l = 0; l < 2; ++l) {
    const auto& rs = rs_host[l];
    if (rs[0] != 0 || rs[(size_t)N] != h.nb_nnz[l]) {
      delete ix;
      return fail(NANN_ERR_INVALID_RAGGED_PARAMS, "level " + std::to_string(l) + " row_splits do not span its values");
    }
    int64_t md = 0;
    for (int64_t i = 0; i < N; ++i) {
      const int64_t len = rs[(size_t)i + 1] - rs[(size_t)i];
      if (len < 0) { delete ix; return fail(NANN_ERR_INVALID_RAGGED_PARAMS, "row_splits not monotone"); }
      md = std::max(md, len);
    }
    ix->max_deg[l] = md;
  }
  for (int64_t i = 0; i < h.n_enter; ++i) {
    if (ep_host[(size_t)i] < 0 || ep_host[(size_t)i] >= N) { delete ix; return fail(NANN_ERR_INDEX_OUT_OF_RANGE, "enter point out of range"); }
    if (i && ep_host[(size_t)i] <= ep_host[(size_t)i - 1]) { delete ix; return fail(NANN_ERR_BAD_ARGUMENT, "enter points must be ascending and unique"); }
  }
  ix->bm_words = (uint32_t)(((N + 31) / 32 + 3) / 4 * 4);
  if (!h.on_device) {  // one-time H2D, what HugeConst's GPU kernel does (huge_const_op.cc:187-218)
    auto up = [&](const void* src, size_t bytes, const void** dst) -> int {
      void* d = nullptr;
      HIP_TRY(hipMalloc(&d, std::max<size_t>(bytes, 16)));
      ix->owned.push_back(d);
      if (bytes) HIP_TRY(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
      *dst = d;
      return NANN_OK;
    };
    nann_index_desc& dd = ix->desc;
    rc = up(h.item_embs, (size_t)N * h.d * dtype_bytes(h.emb_dtype), &dd.item_embs);
    if (!rc) rc = up(h.item_ids, (size_t)N * 8, reinterpret_cast<const void**>(&dd.item_ids));
    for (int l = 0; l < 2 && !rc; ++l) {
      rc = up(h.nb_values[l], (size_t)h.nb_nnz[l] * 4, reinterpret_cast<const void**>(&dd.nb_values[l]));
      if (!rc) rc = up(h.nb_row_splits[l], (size_t)(N + 1) * 8, reinterpret_cast<const void**>(&dd.nb_row_splits[l]));
    }
    if (!rc) rc = up(h.enter_points, (size_t)h.n_enter * 4, reinterpret_cast<const void**>(&dd.enter_points));
    if (rc) { nann_index_destroy(ix); return rc; }
    ix->owns = true;
    dd.on_device = 1;
  }
  probe_index(ix);  // what a beam visits on this graph: 64 queries, one small launch (never fails the creation)
  *out = ix;
  return NANN_OK;
}

void nann_index_destroy(nann_index* ix) {
  if (!ix) return;
  ProjCache::drop_index(ix->uid);  // the scorers' pre-projected tables of this index go with it (retired, then freed)
  for (void* p : ix->owned) (void)hipFree(p);
  delete ix;
}

int nann_index_probe_info(const nann_index* ix, float out[5]) {
  if (!ix || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_index_probe_info: null argument");
  out[0] = ix->probe_valid ? (float)ix->probe_queries : 0.0f;
  out[1] = (float)ix->probe_ef;
  out[2] = ix->probe_new_per_row_mean;
  out[3] = ix->probe_new_per_row_q90;
  out[4] = ix->probe_new_per_row_max;
  return NANN_OK;
}

int nann_index_info(const nann_index* ix, int64_t out[6]) {
  if (!ix || !out) return fail(NANN_ERR_BAD_ARGUMENT, "nann_index_info: null argument");
  out[0] = ix->desc.n_items; out[1] = ix->desc.d; out[2] = ix->desc.n_enter;
  out[3] = ix->max_deg[0]; out[4] = ix->max_deg[1]; out[5] = ix->bm_words;
  return NANN_OK;
}

// ---- fused search -----------------------------------------------------------------------
constexpr int64_t kPhaseTail = 8192;  // behind the slots: reserved (round 4's first pipeline of phases kept its block prefix here)
// (SearchOpt::reserve: workgroup slots the persistent traversal grid leaves FREE -- a host that overlaps another stream's
// kernels with the search, the exchange step of a sharded search, DESIGN.md 7, keeps a few for them; the grid otherwise
// owns every CU's LDS until its first workgroups exit.)

static int bit_length(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

constexpr int kKindAttn = 2;      // plan_search: the attention model (NANN_MODEL_ATTENTION); 0 / 1 = nann_scorer_kind
constexpr int kKindMlpSplit = 3;  //   the MLP scorer in split-f16 form (two slice buffers: the scratch of an attention plan)
constexpr int kKindMlpRes = 4;    //   the MLP scorer, either precision, on the pre-projected table with layer 2 resident in LDS
                                  //   (nann_mlp5.h): 16K-slot set under the weights, or the HBM bitmap
// kind: scorer kind of the call, or -1 = "any" (workspace sizing: the largest plan)
static int plan_search(const nann_index* ix, const int32_t t[6], int64_t n_queries, int kind, SearchPlan* p, const SearchOpt& opt,
                       bool mlp_exact_hint = false) {
  for (int i = 0; i < 6; ++i)
    if (t[i] < 0 || t[i] > kMaxK) return fail(NANN_ERR_UNSUPPORTED, "level_topn entries must be in [0, 1024]");
  DeviceInfo di;
  int rc = device_info(&di);
  if (rc) return rc;
  const int64_t E = ix->desc.n_enter;
  const int64_t raw1 = (int64_t)t[0] * ix->max_deg[1];
  const int64_t raw0 = (int64_t)std::max(t[1], std::max(t[2], t[3])) * ix->max_deg[0];
  const int64_t max_raw = std::max<int64_t>(std::max(raw1, raw0), 1);
  const int64_t max_cand = std::max<int64_t>(std::max<int64_t>(E, t[0] + raw1), std::max<int64_t>(raw0, 1));
  if (max_cand > 0x3fffffffll) return fail(NANN_ERR_UNSUPPORTED, "candidate bound too large");
  p->max_cand = (int)max_cand;
  p->max_raw = (int)max_raw;
  p->pool_cap = std::max(t[1] + t[2] + t[3] + t[4], 1);
  const size_t tail = kMaxD * 4 + 256;  // q + misc behind the phase scratch
  const size_t bm_bytes = (size_t)ix->bm_words * 4;
  // scratch behind the visited set: "any" (workspace sizing) assumes the largest, so that its slots can hold the HBM
  // bitmap of whatever plan the call ends up with
  const size_t big_scratch = (size_t)std::max(kAttnScratch, kMlpSplitScratch);
  const bool res = kind == kKindMlpRes;
  const size_t bm_scratch = kind == kKindAttn ? (size_t)kAttnScratch
                            : kind == kKindMlpSplit ? (size_t)kMlpSplitScratch
                            : res ? (size_t)kMlpResBytes
                            : kind < 0 ? big_scratch : (size_t)kPhaseScratch;
  const bool bitmap_fits = !res && bm_bytes + bm_scratch + tail <= di.lds_max;  // resident layer 2 owns the LDS: HBM bitmap
  const int mode = opt.mode;
  // the bitmap plan: what MLP traversals run, what oversized shards run, and the fallback of the hash plan
  const int bm_vis = (bitmap_fits && mode != NANN_TRAVERSAL_HBM_BITMAP) ? VIS_LDS_BITMAP : VIS_HBM_BITMAP;
  const size_t bm_lds = bm_scratch + tail + (bm_vis == VIS_LDS_BITMAP ? bm_bytes : 0);
  const int bm_per_cu = (bm_vis == VIS_LDS_BITMAP || res) ? 1 : 2;
  // the hash-set plans.  Which table:
  // the visited set of a level holds its marks plus every id the level's rounds keep.  Measured on
  // HNSW(M=32) graphs (profiles/): the rows a beam walks are ~2.75x the mean degree and ~45% of the
  // gathered ids are new.  16K slots (two queries per CU) when that estimate leaves headroom, 32K
  // slots (one query per CU) for wider beams; beyond that the bitmap.  A wrong guess costs speed,
  // not correctness: overflowing queries are rerun on the bitmap kernel.
  // set entries are (remainder, probe step) tags cut from a bijection of the id space (nann_device.h, vis_key): any
  // shard below 2^27 items gets 12 position bits (rounds 1-2 stored the id: 10 position bits at 4M items)
  // bit_length(n_items), not (n_items - 1): a shard of exactly 2^20 items would otherwise take the direct form, where
  // id 2^20 - 1 at position 4095 encodes as the empty value 0xffffffff
  const int id_bits = std::max(16, bit_length((uint64_t)std::max<int64_t>(ix->desc.n_items, 1)));
  const bool tag_fits = id_bits <= 27;
  const double mean_deg0 = (double)ix->desc.nb_nnz[0] / (double)std::max<int64_t>(ix->desc.n_items, 1);
  const double walk_deg = std::min<double>((double)ix->max_deg[0], 2.75 * mean_deg0);
  // Round 5: the estimate is MEASURED per index where it can be (probe_index: 64 queries at nann_index_create, the new nodes a
  // level-0 round finds per frontier row; a high quantile of the probe's queries, so the estimate sits on the tail).  Rounds 1-4
  // guessed it from the mean degree (2.75 x mean degree walked, 45 % new: fitted to the device builder's graphs, mean degree
  // ~17) and sent a dense graph (keepPrunedConnections, mean degree 52) to the one-workgroup-per-CU 32K plan at 0.67 of the
  // roofline, where the 16K plan holds its ~8 k visited ids at 0.84 with no rerun (profiles/rd5c_dense_graph.txt).  The ratio
  // falls as the beam widens (neighbourhoods overlap more), so a probe at ef <= 64 overestimates wider beams: safe side.
  const double rows_walked = (double)t[1] + t[2] + t[3];
  const double est_guess = t[1] + 0.45 * walk_deg * rows_walked;
  // (the 90th percentile of the probe's queries x 1.15, not their maximum: on the exact k-NN graph -- every row at the cap -- one
  // of 64 probe queries found 41 new nodes per row against a mean of 13.6 and a q90 of ~20, and planning on that outlier sent
  // configs[1] to the 32K plan at 0.65 although the real sets hold ~8 k ids; a query in the tail is rerun on the bitmap kernel,
  // which is what the rerun is for: profiles/rd5u_knn_graph.txt)
  const double est_visited = ix->probe_valid ? std::min(t[1] + std::min(1.15 * (double)ix->probe_new_per_row_q90, (double)ix->probe_new_per_row_max) * rows_walked,
                                                        t[1] + (double)ix->max_deg[0] * rows_walked)
                                             : est_guess;
  // the 16K / 32K-slot set holds 16320 / 32704 ids; a measured estimate may come closer to that than a guessed one
  // (measured estimate: up to ~92 % of the capacity a set really has since a filling set cuts its pieces -- 16 256 / 32 640
  //  less the 512 below which a query is handed back; ef = 192 on the shipped graph, estimate 14.7 k: 1.72 M q/s on the 16K plan
  //  with no rerun against 1.47 M on the 32K plan, profiles/rd5ag_planner_thresholds.txt)
  // Beyond 2^20 items the entries are (tag, probe step) pairs and a probe sequence may be 62 steps long at most (longer: the
  // query is rerun); at a load of 0.92 that is 0.5 % PER INSERT, at 0.80 below 1e-6 -- such shards plan with 0.80 of the set
  // (test_a_set_that_fills_up_cuts_its_pieces: tag sets ending at 0.87 / 0.96 load rerun 1 / 12 of 12 queries).
  const bool tag_entries = id_bits > 20;
  const double fit16 = !ix->probe_valid ? 11000.0 : tag_entries ? 13000.0 : 15000.0;
  const double fit32 = !ix->probe_valid ? 24000.0 : tag_entries ? 26000.0 : 30500.0;
  const double worst_visited = t[1] + (double)ix->max_deg[0] * ((double)t[1] + t[2] + t[3]);
  const size_t hash16_lds = (size_t)vis_slots(VIS_LDS_HASH) * 4 + hash_phase_scratch<512, 16384>() + tail;
  const size_t hash32_lds = (size_t)vis_slots(VIS_LDS_HASH32) * 4 + hash_phase_scratch<kNT, 32768>() + tail;
  const bool hash_ok = (kind == NANN_SCORER_L2 || kind < 0) && tag_fits && 2 * hash16_lds <= di.lds_max &&
                       hash32_lds <= di.lds_max;
  int hash_vis = -1;
  if (mode == NANN_TRAVERSAL_LDS_HASH) hash_vis = VIS_LDS_HASH;
  else if (mode == NANN_TRAVERSAL_LDS_HASH32) hash_vis = VIS_LDS_HASH32;
  else if (mode == NANN_TRAVERSAL_AUTO && (kind == NANN_SCORER_L2 || kind < 0)) {
    if (worst_visited <= 16320.0 || est_visited <= fit16) hash_vis = VIS_LDS_HASH;
    else if (worst_visited <= 32704.0 || est_visited <= fit32) hash_vis = VIS_LDS_HASH32;
    // small batches: with at most one query per CU the second 512-thread workgroup of the 16K-slot plan has nothing to
    // overlap with, and a query is served faster by ONE 1024-thread workgroup owning the CU (measured at configs[1]:
    // B = 1 0.196 -> 0.160 ms, B = 64 0.206 -> 0.165 ms; profiles/r3d_*)
    if (hash_vis == VIS_LDS_HASH && kind == NANN_SCORER_L2 && n_queries <= (int64_t)di.cus) hash_vis = VIS_LDS_HASH32;
  }
  // attention model / split-f16 MLP: 16K slots, one workgroup per CU -- when the level's visited ids are expected to fit
  // (a beam too wide for the set would send nearly every query through both kernels)
  const bool fits16 = worst_visited <= 16320.0 || est_visited <= fit16 || mode == NANN_TRAVERSAL_LDS_HASH;
  const bool own_hash_plan = (kind == kKindAttn || kind == kKindMlpSplit || res) && tag_fits && fits16;
  if ((mode == NANN_TRAVERSAL_LDS_HASH || mode == NANN_TRAVERSAL_LDS_HASH32) && !hash_ok && kind >= 0 &&
      !(own_hash_plan && mode == NANN_TRAVERSAL_LDS_HASH))
    return fail(NANN_ERR_UNSUPPORTED, "hash-set traversal: shards below 2^27 items; the 32K-slot set: L2 scorer only");
  unsigned long long off[9];
  // the slot's last region: the HBM bitmap of a bitmap plan, and where a resident-layer-2 traversal parks its 16K-slot
  // set while it scores (nann_mlp5.h); "any" sizes for both
  uint32_t gbm_words = bm_vis == VIS_HBM_BITMAP ? ix->bm_words : 0u;
  if (res || kind < 0 || kind == kKindAttn) gbm_words = std::max<uint32_t>(std::max<uint32_t>(gbm_words, ix->bm_words), (uint32_t)vis_slots(kind < 0 ? VIS_LDS_HASH32 : VIS_LDS_HASH));
  p->slot_bytes = slot_layout(p->max_cand, p->max_raw, p->pool_cap, gbm_words, off);
  p->id_bits = id_bits;
  p->fb_vis = bm_vis;
  p->fb_lds_bytes = bm_lds;
  p->fb_slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, (int64_t)di.cus * bm_per_cu));
  if (own_hash_plan && mode != NANN_TRAVERSAL_LDS_BITMAP && mode != NANN_TRAVERSAL_HBM_BITMAP) {
    // attention model / split-f16 MLP: 16K-slot set + two weight-slice buffers, one 512-thread workgroup per CU
    p->vis = VIS_LDS_HASH;
    p->nt = 512;
    p->lds_bytes = res ? (size_t)kMlpResBytes + hash_phase_scratch<512, 16384>() + tail  // [W2 (set over its head) | vectors | phase scratch]
                       : (size_t)vis_slots(VIS_LDS_HASH) * 4 + bm_scratch + tail;
    p->slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, (int64_t)di.cus));
  } else if (hash_ok && hash_vis == VIS_LDS_HASH) {
    p->vis = VIS_LDS_HASH;
    p->nt = 512;
    p->lds_bytes = hash16_lds;
    p->slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, (int64_t)di.cus * 2));
  } else if (hash_ok && hash_vis == VIS_LDS_HASH32) {
    p->vis = VIS_LDS_HASH32;
    p->nt = kNT;
    p->lds_bytes = hash32_lds;
    p->slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, (int64_t)di.cus));
  } else {
    p->vis = bm_vis;
    p->nt = kNT;  // (the MLP kernels run kMlpNT threads; the launcher knows)
    p->lds_bytes = bm_lds;
    p->slots = p->fb_slots;
  }
  // the MLP's pipeline of phases (nann_mlp6.h): traversal stages at the 16K-slot plan's geometry, one slot per query of a chunk
  // Where it pays (profiles/r4tu_mlp_batch_sweep_phased_vs_fused.txt, r4x_mlp_loop_ab.txt; configs[2]): exact f32 at every
  // batch size (batch 32: 3.4x -- the scoring launch spreads 32 queries' rows over the chip, the fused kernel holds 32 CUs --,
  // 128: 1.7x, >= 256: +3..8 %); split-f16 below ~160 queries (batch 32: 1.64x).  Above that the fused split-f16 kernel, which
  // runs the same software-pipelined block loop, is level with it or ahead: 256-512 queries +8..19 % (one query per CU finishes
  // sooner than 12 launches), 1024: within 2 % either way box to box, 4096: +2.7 %.  nann_search_options.mlp_form forces either.
  const bool exact_form = mlp_exact_hint;
  const bool pays = exact_form || n_queries <= 160;
  const bool no_forced_bitmap = mode != NANN_TRAVERSAL_LDS_BITMAP && mode != NANN_TRAVERSAL_HBM_BITMAP;
  const bool want_phased = opt.mlp_form == NANN_MLP_FORM_PHASED || (opt.mlp_form == NANN_MLP_FORM_AUTO && pays);
  p->phased = res && own_hash_plan && no_forced_bitmap && 2 * hash16_lds <= di.lds_max && want_phased;
  p->phase_vis = VIS_LDS_HASH;
  p->phase_per_cu = 2;
  p->phase_lds_bytes = hash16_lds;
  // Wide beams (ef = 256: a level's visited ids need the 32K-slot set): the fused kernel keeps its bitmap in HBM because the
  // resident weights leave no room for a set; the pipeline's traversal stages own the LDS and run the L2 kernel's 32K-slot
  // plan (one 1024-thread workgroup per CU).  Both precisions, every batch size (profiles/r5g_*).
  const bool fits32 = worst_visited <= 32704.0 || est_visited <= fit32;
  if (res && !own_hash_plan && tag_fits && fits32 && no_forced_bitmap && mode != NANN_TRAVERSAL_LDS_HASH && hash32_lds <= di.lds_max &&
      opt.mlp_form != NANN_MLP_FORM_FUSED) {
    p->phased = true;
    p->phase_vis = VIS_LDS_HASH32;
    p->phase_per_cu = 1;
    p->phase_lds_bytes = hash32_lds;
    gbm_words = std::max<uint32_t>(gbm_words, (uint32_t)vis_slots(VIS_LDS_HASH32));  // where the 32K-slot set is parked
    p->slot_bytes = slot_layout(p->max_cand, p->max_raw, p->pool_cap, gbm_words, off);
  }
  // few queries (at most one per CU): a query's stages run faster on ONE 1024-thread workgroup per CU than on one of two
  // 512-thread workgroups that has no partner to overlap with (the L2 kernel's small-batch rule, above)
  if (p->phased && p->phase_vis == VIS_LDS_HASH && opt.small32 && n_queries <= (int64_t)di.cus && tag_fits && hash32_lds <= di.lds_max) {
    p->phase_vis = VIS_LDS_HASH32;
    p->phase_per_cu = 1;
    p->phase_lds_bytes = hash32_lds;
    gbm_words = std::max<uint32_t>(gbm_words, (uint32_t)vis_slots(VIS_LDS_HASH32));
    p->slot_bytes = slot_layout(p->max_cand, p->max_raw, p->pool_cap, gbm_words, off);
  }
  p->phase_slots = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_queries, kPhaseChunk), (int64_t)di.cus * p->phase_per_cu));
  p->phase_score_wgs = di.cus;
  p->est_visited = (float)est_visited;
  p->worst_visited = (float)worst_visited;
  if (kind < 0)  // sizing: the widest plan (two workgroups per CU, or one slot per query of a phased chunk)
    p->slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, std::max<int64_t>((int64_t)di.cus * 2, kPhaseChunk)));
  else if (const int reserve = opt.reserve) {  // leave workgroup slots to kernels of other streams (nann_search_options.slot_reserve)
    const int per_cu = p->vis == VIS_LDS_HASH && p->nt == 512 && !res && kind != kKindAttn && kind != kKindMlpSplit ? 2
                       : p->vis == VIS_HBM_BITMAP && !res ? 2 : 1;
    p->slots = std::max(1, std::min(p->slots, di.cus * per_cu - reserve));
    p->fb_slots = std::max(1, std::min(p->fb_slots, di.cus * bm_per_cu - reserve));
    // the pipeline of phases keeps the same promise (it ignored the reserve): its traversal stages leave `reserve`
    // slots, its scoring launches -- one workgroup owns a CU's LDS -- the CUs those slots stand for
    p->phase_slots = std::max(1, std::min(p->phase_slots, di.cus * p->phase_per_cu - reserve));
    p->phase_score_wgs = std::max(1, di.cus - (reserve + 1) / 2);
  }
  return NANN_OK;
}

// The three setters are PROCESS DEFAULTS of the fields of nann_search_options that a call leaves at -1 (round 5: the knobs
// themselves travel with the call -- nann_search_opt --, so two threads sharing a handle no longer share a plan).
int nann_set_search_reserve(int32_t workgroups) {
  if (workgroups < 0) return fail(NANN_ERR_BAD_ARGUMENT, "nann_set_search_reserve: negative");
  g_slot_reserve.store(workgroups, std::memory_order_relaxed);
  return NANN_OK;
}

void nann_search_options_init(nann_search_options* o) {
  if (!o) return;
  o->struct_bytes = (int32_t)sizeof(*o);
  o->traversal_mode = o->slot_reserve = o->preprojection = o->mlp_form = -1;
}

int nann_set_traversal_mode(int32_t mode) {
  if (mode < NANN_TRAVERSAL_AUTO || mode > NANN_TRAVERSAL_LDS_HASH32)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_set_traversal_mode: unknown mode");
  g_traversal_mode.store(mode, std::memory_order_relaxed);
  return NANN_OK;
}

int nann_search_workspace_bytes(const nann_index* ix, const int32_t level_topn[6], int64_t n_queries,
                                int64_t* nbytes) {
  if (!ix || !level_topn || !nbytes) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_workspace_bytes: null argument");
  SearchPlan p;
  const int rc = plan_search(ix, level_topn, n_queries, -1, &p, resolve_options(nullptr));  // (kind "any": the widest plan, whatever the options)
  if (rc) return rc;
  *nbytes = (int64_t)(256 + p.slot_bytes * (unsigned long long)std::max(p.slots, p.fb_slots) + kPhaseTail);
  return NANN_OK;
}

}  // extern "C"


// The pre-projected table of (scorer, index): found, or built on `st` by `build(table)` (a one-time wait per pair,
// ~10-20 ms per million items: nann_*_prepare moves it ahead of traffic).  *out stays null -- with NANN_OK -- when there
// is to be no table: pre-projection switched off, or no room for it in HBM (the caller then runs the kernels that
// read the embedding rows; ADVICE r3: a failed hipMalloc must not fail the search).  pin: count a prepare call.
template <typename Build>
static int projection_for(ProjCache& c, const nann_index* ix, int width, hipStream_t st, Build build, bool pin, bool enabled,
                          std::shared_ptr<ProjTable>* out) {
  const size_t bytes = (size_t)ix->desc.n_items * (size_t)width * 4;
  return c.acquire(ix->uid, bytes, enabled, pin, [&](float* t) -> int {
    const int rc = build(t);
    if (rc) return rc;
    // the table becomes visible to searches on OTHER streams when acquire() returns: it must be complete by then
    const hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(NANN_ERR_HIP, std::string("pre-projection: ") + hipGetErrorString(e));
    return NANN_OK;
  }, out);
}

static void projection_used(ProjCache& c, const std::shared_ptr<ProjTable>& tab, hipStream_t st) { c.used(tab, st); }

static int projection_release(ProjCache& c, const nann_index* ix) {
  if (c.release(ix->uid)) return NANN_OK;
  return fail(NANN_ERR_BAD_ARGUMENT, "release: no pre-projected table of this index");
}

static int mlp_projection(const nann_scorer* sc, const nann_index* ix, hipStream_t st, bool pin, bool enabled, std::shared_ptr<ProjTable>* out) {
  return projection_for(sc->proj, ix, kMlpProjWidth, st, [&](float* t) {
    return launch_mlp_preproject(ix->desc.emb_dtype, ix->desc.item_embs, (long long)ix->desc.n_items, ix->desc.d, sc->mlp.w1, t, st);
  }, pin, enabled, out);
}

static int attn_projection(const nann_attn_scorer* sc, const nann_index* ix, hipStream_t st, bool pin, bool enabled, std::shared_ptr<ProjTable>* out) {
  return projection_for(sc->proj, ix, kAttnProjWidth, st, [&](float* t) {
    return launch_attn_preproject(ix->desc.emb_dtype, sc->P, ix->desc.item_embs, (long long)ix->desc.n_items, t, st);
  }, pin, enabled, out);
}

// L2 instantiations live in nann_l2_inst.hip (one object per row dtype), MLP ones in
// nann_mlp_inst.hip (one per embedding dim): the heavy kernels compile in parallel.
static int launch_search_any(int lpr, int dt, int kind, int split, int vis, int nt, int slots, size_t lds_bytes,
                             const SearchArgs& a, hipStream_t st) {
  if (kind == NANN_SCORER_MLP) {
    if (lpr == 8) return launch_search_mlp_d64(dt, split, vis, slots, lds_bytes, a, st);
    if (lpr == 16) return launch_search_mlp_d128(dt, split, vis, slots, lds_bytes, a, st);
    if (lpr == 32) return launch_search_mlp_d256(dt, split, vis, slots, lds_bytes, a, st);
    return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: d <= 256 only");
  }
  if (dt == NANN_F16) return launch_search_l2_f16(lpr, vis, nt, slots, lds_bytes, a, st);
  if (dt == NANN_BF16) return launch_search_l2_bf16(lpr, vis, nt, slots, lds_bytes, a, st);
  return launch_search_l2_f32(lpr, vis, nt, slots, lds_bytes, a, st);
}

extern "C" {

int nann_search(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                const int32_t level_topn[6], void* workspace, int64_t workspace_bytes,
                int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                int32_t* counters, nann_stream_t stream) {
  return nann_search_opt(ix, scorer, q, n_queries, level_topn, nullptr, workspace, workspace_bytes, out_item_ids, out_scores,
                         out_index, status, counters, nullptr, nullptr, nullptr, stream);  // deprecated: thin wrapper
}

}  // extern "C"

// the traversal for (index, scorer | attention model): plan, fill the arguments, launch (+ the
// fallback launch of the hash-set plans).  tq: optional device i32[n_queries, 6], level_topn per query (level_topn then
// holds the per-launch maxima).
static int search_impl(const nann_index* ix, const nann_scorer* scorer, const nann_attn_scorer* attn,
                       const float* q, const float* kt, const float* upad, int64_t n_queries,
                       const int32_t level_topn[6], const int32_t* tq, void* workspace, int64_t workspace_bytes,
                       int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                       int32_t* counters, int64_t* phase_ticks, const nann_search_options* options, nann_search_plan* plan_out,
                       hipStream_t st) {
  if (n_queries > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "too many queries in one call");
  int rc = check_options(options);
  if (rc) return rc;
  const SearchOpt opt = resolve_options(options);
  const int kind = attn ? kKindAttn : scorer->desc.kind;
  const bool mlp = !attn && kind == NANN_SCORER_MLP;
  const bool mlp_split = mlp && scorer->desc.precision == NANN_MLP_SPLIT_F16;
  // the item-only part of the scorer, pre-projected per (scorer, index): found or built here (nann_*_prepare does it
  // ahead of traffic); without a table -- switched off, or no room in HBM -- the kernels that read the embedding rows run
  std::shared_ptr<ProjTable> tab;
  ProjCache* cache = nullptr;
  if (attn) {  // both precisions run on the table of item-only layers (nann_attn_proj.h; nann_attn_kernels.h PROJ)
    cache = &attn->proj;
    rc = attn_projection(attn, ix, st, false, opt.preproject != 0, &tab);
  } else if (mlp) {
    cache = &scorer->proj;
    rc = mlp_projection(scorer, ix, st, false, opt.preproject != 0, &tab);
  }
  if (rc) return rc;
  const bool mlp_res = mlp && tab;  // layer 2 resident in LDS, either precision (nann_mlp5.h)
  SearchPlan p;
  rc = plan_search(ix, level_topn, n_queries, mlp_res ? kKindMlpRes : mlp_split ? kKindMlpSplit : kind, &p, opt, mlp && !mlp_split);
  if (rc) return rc;
  if (plan_out) {  // what this call runs (host-side facts; the number of reruns lives in the workspace: nann_search_reruns)
    const bool ph = mlp_res && p.phased;
    const int v = ph ? p.phase_vis : p.vis;
    plan_out->visited_set = v == VIS_LDS_HASH ? NANN_TRAVERSAL_LDS_HASH : v == VIS_LDS_HASH32 ? NANN_TRAVERSAL_LDS_HASH32
                            : v == VIS_LDS_BITMAP ? NANN_TRAVERSAL_LDS_BITMAP : NANN_TRAVERSAL_HBM_BITMAP;
    plan_out->fallback_visited_set = p.fb_vis == VIS_LDS_BITMAP ? NANN_TRAVERSAL_LDS_BITMAP : NANN_TRAVERSAL_HBM_BITMAP;
    plan_out->threads = ph ? (p.phase_vis == VIS_LDS_HASH32 ? kNT : 512) : ((mlp || attn) && p.nt == kNT ? 512 : p.nt);
    plan_out->workgroups = ph ? p.phase_slots : p.slots;
    plan_out->phased = ph ? 1 : 0;
    plan_out->table = tab ? 1 : 0;
    plan_out->est_visited = p.est_visited;
    plan_out->worst_visited = p.worst_visited;
  }
  const int64_t need_slots = p.phased ? std::max<int64_t>(std::min<int64_t>(n_queries, kPhaseChunk), p.fb_slots) : std::max(p.slots, p.fb_slots);
  if (!workspace || workspace_bytes < (int64_t)(256 + p.slot_bytes * (unsigned long long)need_slots + (p.phased ? kPhaseTail : 0)))
    return fail(NANN_ERR_CAPACITY, "workspace smaller than nann_search_workspace_bytes()");
  SearchArgs a;
  a.emb = ix->desc.item_embs;
  a.item_ids = ix->desc.item_ids;
  for (int l = 0; l < 2; ++l) { a.nbv[l] = ix->desc.nb_values[l]; a.nbrs[l] = ix->desc.nb_row_splits[l]; }
  a.enter = ix->desc.enter_points;
  a.n_enter = (int)ix->desc.n_enter;
  a.n_items = (uint32_t)ix->desc.n_items;
  a.d = ix->desc.d;
  a.q = q;
  a.n_queries = (int)n_queries;
  for (int i = 0; i < 6; ++i) a.t[i] = level_topn[i];
  a.tq = tq;
  a.ws = static_cast<unsigned char*>(workspace);
  a.slot_bytes = p.slot_bytes;
  a.bm_words = ix->bm_words;
  a.max_cand = p.max_cand; a.max_raw = p.max_raw; a.pool_cap = p.pool_cap;
  a.out_ids = out_item_ids; a.out_scores = out_scores; a.out_index = out_index;
  a.status = status; a.counters = counters;
  a.phase_ticks = reinterpret_cast<long long*>(phase_ticks);
  a.id_bits = p.id_bits;
  a.redo = 0;
  a.phase = 0;
  a.proj = tab ? tab->table : nullptr;
  a.mlp = MlpParams{};
  a.attn = AttnParams{};
  a.kt = kt; a.upad = upad;
  HIP_TRY(hipMemsetAsync(workspace, 0, 256, st));  // WsHeader: query queues, hand-back counter
  const int dt = ix->desc.emb_dtype;
  const bool hashed = p.vis == VIS_LDS_HASH || p.vis == VIS_LDS_HASH32;
  // main launch, then -- hash-set plans -- the rerun of the queries whose set could have overflowed on the bitmap
  // kernel (its workgroups leave at once when there is none)
  auto both = [&](auto&& launch_on) -> int {
    int r2 = launch_on(p.vis, p.nt, p.slots, p.lds_bytes);
    if (!r2 && hashed) {
      a.redo = 1;
      r2 = launch_on(p.fb_vis, kNT, p.fb_slots, p.fb_lds_bytes);
    }
    if (cache) projection_used(*cache, tab, st);
    return r2;
  };
  if (attn) {
    a.attn = attn->P;
    if (tab) {  // the default form: q_ and the e rows of DNN layer 1 pre-projected per (model, index) (nann_attn_proj.h)
      auto launch = attn->precision == NANN_MLP_SPLIT_F16 ? launch_search_attn_proj : launch_search_attn_xproj;
      // split-f16 on the 16K-slot plan: the form with keys and weights resident for a scoring call (the bitmap plans and the
      // overflow rerun keep the slice-ring form)
      const bool resident = attn->precision == NANN_MLP_SPLIT_F16;
      return both([&](int vis, int, int slots, size_t lds) {
        if (resident && vis == VIS_LDS_HASH && !a.redo) return launch_search_attn_res(slots, lds, a, st);
        return launch(vis, slots, lds, a, st);
      });
    }
    auto launch = attn->precision == NANN_MLP_SPLIT_F16 ? launch_search_attn_split : launch_search_attn;
    return both([&](int vis, int, int slots, size_t lds) { return launch(ix->desc.d, dt, vis, slots, lds, a, st); });
  }
  a.mlp = scorer->mlp;
  a.phase = 0;
  const int exact = scorer->desc.kind == NANN_SCORER_MLP && scorer->desc.precision == NANN_MLP_EXACT_F32;
  if (mlp_res && p.phased) {
    // The default form of both precisions at beams that fit the 16K-slot set: the pipeline of phases (nann_mlp6.h).  Per
    // chunk of <= 1024 queries: traversal stage 0, then for every round its scoring launch and the
    // traversal stage behind it; last the rerun of the queries whose set could have overflowed (fused kernel, HBM bitmap).
    DeviceInfo di;
    rc = device_info(&di);
    if (rc) return rc;
    const int k5 = level_topn[5];
    for (int64_t c0 = 0; c0 < n_queries && !rc; c0 += kPhaseChunk) {
      SearchArgs c = a;
      c.n_queries = (int)std::min<int64_t>(kPhaseChunk, n_queries - c0);
      c.q = q + (size_t)c0 * a.d;
      if (tq) c.tq = tq + (size_t)c0 * 6;
      c.out_ids = out_item_ids + (size_t)c0 * k5;
      if (out_scores) c.out_scores = out_scores + (size_t)c0 * k5;
      if (out_index) c.out_index = out_index + (size_t)c0 * k5;
      c.status = status + c0;
      if (counters) c.counters = counters + (size_t)c0 * 3 * NANN_NUM_ROUNDS;
      if (phase_ticks) c.phase_ticks = reinterpret_cast<long long*>(phase_ticks) + (size_t)c0 * NANN_NUM_PHASES;
      if (c0) HIP_TRY(hipMemsetAsync(workspace, 0, 256, st));
      const int slots = (int)std::min<int64_t>(c.n_queries, (int64_t)p.phase_slots);
      for (int ph = 0; ph <= NANN_NUM_ROUNDS && !rc; ++ph) {
        c.phase = ph;
        rc = launch_search_mlp_phase(p.phase_vis, slots, p.phase_lds_bytes, c, st);
        if (!rc && ph < NANN_NUM_ROUNDS) {
          rc = launch_mlp_phase_score(exact, c, ph, p.phase_score_wgs, st);
        }
      }
      if (!rc) {
        c.redo = 1;
        c.phase = 0;
        rc = launch_search_mlp_res(exact, p.fb_vis, p.fb_slots, p.fb_lds_bytes, c, st);
      }
    }
    if (cache) projection_used(*cache, tab, st);
    return rc;
  }
  if (mlp_res)  // wide beams / large shards / forced plans: the fused kernel with layer 2 resident in LDS (nann_mlp5.h)
    return both([&](int vis, int, int slots, size_t lds) {
      return launch_search_mlp_res(exact, vis, slots, lds, a, st);
    });
  return both([&](int vis, int nt, int slots, size_t lds) {
    return launch_search_any(ix->desc.d / 8, dt, kind, mlp_split, vis, nt, slots, lds, a, st);
  });
}

// ---- the probe of nann_index_create ---------------------------------------------------------------------------------
// plan_search has to know how many ids a level's visited set will hold BEFORE it launches (16K-slot set, two workgroups per
// CU; 32K-slot set, one; bitmap).  That is a property of the graph -- degrees, and how much the neighbourhoods of a beam's
// rows overlap -- which no formula over the mean degree captures for every builder.  So the index
// measures it once: 64 of its own rows as queries, ef = min(64, #enter points), L2 scorer, on the HBM-bitmap plan (which
// cannot overflow); from the kernel's per-round counters, new nodes found per frontier row over the three level-0 rounds.
__global__ void k_probe_queries(const void* emb, int dt, int d, long long n_items, int nq, float* q) {
  const int j = blockIdx.x;
  const long long row = (long long)j * (n_items / nq);
  for (int k = threadIdx.x; k < d; k += blockDim.x) {
    float v;
    if (dt == NANN_F32) v = static_cast<const float*>(emb)[row * d + k];
    else {
      const uint32_t h = static_cast<const uint16_t*>(emb)[row * d + k];
      v = dt == NANN_F16 ? half_bits_to_float(h) : bf16_bits_to_float(h);
    }
    q[(size_t)j * d + k] = v;
  }
}

static void probe_index(nann_index* ix) {
  ix->probe_valid = false;
  // opt-out: NANN_INDEX_PROBE=0 -- the planner then falls back to its degree-based guess.  The probe allocates,
  // launches on the NULL stream and copies back with blocking copies: it SYNCHRONISES the device (include/nann_hip.h says so).
  static const bool enabled = [] { const char* e = std::getenv("NANN_INDEX_PROBE"); return !(e && e[0] == '0'); }();
  if (!enabled) return;
  const std::string saved_error = nann::g_err;  // a probe that fails must not leave ITS message behind a successful create
  struct RestoreError { const std::string& s; ~RestoreError() { nann::g_err = s; } } restore{saved_error};
  const int d = ix->desc.d;
  const int64_t E = ix->desc.n_enter, N = ix->desc.n_items;
  if (E < 8 || N < 4096) return;  // (toy indices: the degree-based guess)
  const int nq = 64, e = (int)std::min<int64_t>(64, E);
  const int32_t t[6] = {e, e, e, e, e, std::min(e, 10)};
  nann_scorer_desc sd{};
  sd.kind = NANN_SCORER_L2; sd.d = d; sd.emb_dtype = ix->desc.emb_dtype;
  nann_scorer* sc = nullptr;
  if (nann_scorer_create(&sd, &sc) != NANN_OK) return;
  nann_search_options o;
  nann_search_options_init(&o);
  o.traversal_mode = NANN_TRAVERSAL_HBM_BITMAP;
  o.slot_reserve = 0;
  int64_t ws_bytes = 0;
  const size_t n_ctr = (size_t)nq * 3 * NANN_NUM_ROUNDS;
  unsigned char* buf = nullptr;
  std::vector<int32_t> h_ctr(n_ctr), h_st((size_t)nq);
  bool ok = nann_search_workspace_bytes(ix, t, nq, &ws_bytes) == NANN_OK;
  const size_t off_q = ((size_t)ws_bytes + 255) & ~(size_t)255, off_ids = off_q + (size_t)nq * d * 4,
               off_st = off_ids + (size_t)nq * t[5] * 8, off_ctr = off_st + (size_t)nq * 4, total = off_ctr + n_ctr * 4;
  ok = ok && hipMalloc(reinterpret_cast<void**>(&buf), total) == hipSuccess;
  if (ok) {
    hipStream_t st = nullptr;
    hipLaunchKernelGGL(k_probe_queries, dim3(nq), dim3(128), 0, st, ix->desc.item_embs, ix->desc.emb_dtype, d, (long long)N, nq,
                       reinterpret_cast<float*>(buf + off_q));
    ok = search_impl(ix, sc, nullptr, reinterpret_cast<const float*>(buf + off_q), nullptr, nullptr, nq, t, nullptr, buf, ws_bytes,
                     reinterpret_cast<int64_t*>(buf + off_ids), nullptr, nullptr, reinterpret_cast<int32_t*>(buf + off_st),
                     reinterpret_cast<int32_t*>(buf + off_ctr), nullptr, &o, nullptr, st) == NANN_OK;
    ok = ok && hipMemcpy(h_ctr.data(), buf + off_ctr, n_ctr * 4, hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(h_st.data(), buf + off_st, (size_t)nq * 4, hipMemcpyDeviceToHost) == hipSuccess;
  }
  if (buf) (void)hipFree(buf);
  nann_scorer_destroy(sc);
  // only the probe's own non-sticky error (a failed hipMalloc, say) is cleared; a sticky device error stays set and the
  // caller's next HIP call reports it
  if (!ok) (void)hipGetLastError();
  if (!ok) return;
  double sum = 0.0, mx = 0.0;
  int n_valid = 0;
  std::vector<double> ratios;
  for (int j = 0; j < nq; ++j) {
    if (h_st[(size_t)j] != NANN_OK) continue;
    const int32_t* c = h_ctr.data() + (size_t)j * 3 * NANN_NUM_ROUNDS;  // [F | G | S][round]
    double rows = 0.0, fresh = 0.0;
    for (int r = 2; r <= 4; ++r) { rows += c[0 * NANN_NUM_ROUNDS + r]; fresh += c[2 * NANN_NUM_ROUNDS + r]; }
    if (rows <= 0.0) continue;
    const double ratio = fresh / rows;
    sum += ratio; mx = std::max(mx, ratio);
    ratios.push_back(ratio);
    ++n_valid;
  }
  if (n_valid < 8) return;  // (a corpus whose clusters a beam exhausts: nothing to learn from failed requests)
  ix->probe_valid = true;
  ix->probe_ef = e;
  ix->probe_queries = n_valid;
  std::sort(ratios.begin(), ratios.end());
  ix->probe_new_per_row_mean = (float)(sum / n_valid);
  ix->probe_new_per_row_q90 = (float)ratios[(size_t)(0.9 * (double)(n_valid - 1))];
  ix->probe_new_per_row_max = (float)mx;
}

extern "C" {

int nann_search_ex(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                   const int32_t level_topn[6], void* workspace, int64_t workspace_bytes,
                   int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                   int32_t* counters, int64_t* phase_ticks, nann_stream_t stream) {
  return nann_search_opt(ix, scorer, q, n_queries, level_topn, nullptr, workspace, workspace_bytes, out_item_ids, out_scores,
                         out_index, status, counters, phase_ticks, nullptr, nullptr, stream);  // deprecated: thin wrapper
}

int nann_search_opt(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                    const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace, int64_t workspace_bytes,
                    int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status, int32_t* counters,
                    int64_t* phase_ticks, const nann_search_options* options, nann_search_plan* plan, nann_stream_t stream) {
  if (!ix || !scorer || !level_topn_max || !out_item_ids || !status)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_opt: null argument");
  if (n_queries <= 0) return NANN_OK;
  if (scorer->desc.d != ix->desc.d || scorer->desc.emb_dtype != ix->desc.emb_dtype)
    return fail(NANN_ERR_BAD_ARGUMENT, "scorer and index disagree on d / dtype");
  if (level_topn && phase_ticks) return fail(NANN_ERR_UNSUPPORTED, "nann_search_opt: phase ticks with a uniform level_topn only");
  return search_impl(ix, scorer, nullptr, q, nullptr, nullptr, n_queries, level_topn_max, level_topn, workspace, workspace_bytes,
                     out_item_ids, out_scores, out_index, status, counters, phase_ticks, options, plan, as_stream(stream));
}

int nann_search_reruns(const void* workspace, int64_t* n_rerun, nann_stream_t stream) {
  if (!workspace || !n_rerun) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_reruns: null argument");
  unsigned int v = 0;
  HIP_TRY(hipMemcpyAsync(&v, static_cast<const unsigned char*>(workspace) + offsetof(WsHeader, n_redo), 4, hipMemcpyDeviceToHost, as_stream(stream)));
  HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  *n_rerun = v;
  return NANN_OK;
}

int nann_search_v(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                  const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                  int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                  int32_t* status, int32_t* counters, nann_stream_t stream) {
  return nann_search_opt(ix, scorer, q, n_queries, level_topn_max, level_topn, workspace, workspace_bytes, out_item_ids,
                         out_scores, out_index, status, counters, nullptr, nullptr, nullptr, stream);  // deprecated: thin wrapper
}

// ---- lifecycle of the pre-projected tables (ProjCache) -------------------------------------------------------
int nann_set_preprojection(int32_t enabled) {
  g_preproject.store(enabled ? 1 : 0, std::memory_order_relaxed);
  return NANN_OK;
}

static int table_width(const nann_scorer* s, const nann_attn_scorer* at) {
  if (at) return kAttnProjWidth;
  return (s && s->desc.kind == NANN_SCORER_MLP) ? kMlpProjWidth : 0;
}
static int prepare_impl(const nann_scorer* s, const nann_attn_scorer* at, const nann_index* ix, hipStream_t st) {
  if (table_width(s, at) == 0) return NANN_OK;  // L2: nothing to pre-project
  const int d = at ? at->P.d : s->desc.d, dt = at ? at->emb_dtype : s->desc.emb_dtype;
  if (d != ix->desc.d || dt != ix->desc.emb_dtype) return fail(NANN_ERR_BAD_ARGUMENT, "scorer and index disagree on d / dtype");
  std::shared_ptr<ProjTable> tab;
  const bool enabled = resolve_options(nullptr).preproject != 0;
  const int rc = at ? attn_projection(at, ix, st, true, enabled, &tab) : mlp_projection(s, ix, st, true, enabled, &tab);
  if (rc) return rc;
  if (!tab) return fail(NANN_ERR_CAPACITY, "no room in HBM for the pre-projected table (or pre-projection is switched off): "
                                           "searches of this pair will read the embedding table");
  return NANN_OK;
}
static int table_bytes_impl(const nann_scorer* s, const nann_attn_scorer* at, const nann_index* ix, int64_t* table_bytes,
                            int64_t* resident_bytes) {
  if (table_bytes) *table_bytes = ix ? (int64_t)ix->desc.n_items * table_width(s, at) * 4 : 0;
  if (resident_bytes) {
    *resident_bytes = 0;
    ProjCache* c = at ? &at->proj : s ? &s->proj : nullptr;
    if (c) *resident_bytes = (int64_t)c->resident();
  }
  return NANN_OK;
}

int nann_scorer_prepare(const nann_scorer* scorer, const nann_index* ix, nann_stream_t stream) {
  if (!scorer || !ix) return fail(NANN_ERR_BAD_ARGUMENT, "nann_scorer_prepare: null argument");
  return prepare_impl(scorer, nullptr, ix, as_stream(stream));
}
int nann_scorer_release(const nann_scorer* scorer, const nann_index* ix) {
  if (!scorer || !ix) return fail(NANN_ERR_BAD_ARGUMENT, "nann_scorer_release: null argument");
  if (table_width(scorer, nullptr) == 0) return NANN_OK;
  return projection_release(scorer->proj, ix);
}
int nann_scorer_table_bytes(const nann_scorer* scorer, const nann_index* ix, int64_t* table_bytes, int64_t* resident_bytes) {
  if (!scorer) return fail(NANN_ERR_BAD_ARGUMENT, "nann_scorer_table_bytes: null scorer");
  return table_bytes_impl(scorer, nullptr, ix, table_bytes, resident_bytes);
}

// ---- the serving signature: comm_seq + level_topn -> top_k, for whatever model the node names ----
static size_t model_query_bytes(const nann_model* m, int64_t n_queries) {
  const size_t per = m->kind == NANN_MODEL_ATTENTION ? (size_t)(256 * 64 + 64 * 64) * 4 : (size_t)m->d * 4;
  return ((size_t)n_queries * per + 255) & ~(size_t)255;
}

int nann_search_model_workspace_bytes(const nann_index* ix, const nann_model* m, const int32_t level_topn[6],
                                      int64_t n_queries, int64_t* nbytes) {
  if (!ix || !m || !level_topn || !nbytes) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_model_workspace_bytes: null argument");
  SearchPlan p;
  const int rc = plan_search(ix, level_topn, n_queries, m->kind == NANN_MODEL_ATTENTION ? kKindAttn : -1, &p, resolve_options(nullptr));
  if (rc) return rc;
  *nbytes = (int64_t)(256 + p.slot_bytes * (unsigned long long)std::max(p.slots, p.fb_slots) + kPhaseTail + 256 +
                      model_query_bytes(m, n_queries));
  return NANN_OK;
}

static int search_model_impl(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                             const int32_t level_topn[6], const int32_t* tq, void* workspace, int64_t workspace_bytes,
                             int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                             int32_t* counters, const nann_search_options* options, nann_search_plan* plan, nann_stream_t stream) {
  if (!ix || !m || !comm_seq_f16 || !level_topn || !out_item_ids || !status || !workspace)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_model: null argument");
  if (n_queries <= 0) return NANN_OK;
  if (m->d != ix->desc.d || m->emb_dtype != ix->desc.emb_dtype)
    return fail(NANN_ERR_BAD_ARGUMENT, "model and index disagree on d / dtype");
  int64_t need = 0;
  int rc = nann_search_model_workspace_bytes(ix, m, level_topn, n_queries, &need);
  if (rc) return rc;
  if (workspace_bytes < need) return fail(NANN_ERR_CAPACITY, "workspace smaller than nann_search_model_workspace_bytes()");
  const size_t qb = model_query_bytes(m, n_queries);
  const int64_t search_bytes = need - (int64_t)qb - 256;
  unsigned char* qbuf = static_cast<unsigned char*>(workspace) + ((search_bytes + 255) & ~255ll);
  hipStream_t st = as_stream(stream);
  if (m->kind == NANN_MODEL_ATTENTION) {  // per-user side once per request (build_opt_graph.py:91-107), then the traversal
    float* kt = reinterpret_cast<float*>(qbuf);
    float* upad = kt + (size_t)n_queries * 256 * 64;
    rc = nann_attn_prepare(m->attn, comm_seq_f16, n_queries, kt, upad, stream);
    if (rc) return rc;
    return search_impl(ix, nullptr, m->attn, nullptr, kt, upad, n_queries, level_topn, tq, workspace, search_bytes,
                       out_item_ids, out_scores, out_index, status, counters, nullptr, options, plan, st);
  }
  float* q = reinterpret_cast<float*>(qbuf);
  rc = nann_user_seq_mean(comm_seq_f16, n_queries, m->seq_len, m->d, q, stream);
  if (rc) return rc;
  return search_impl(ix, m->scorer, nullptr, q, nullptr, nullptr, n_queries, level_topn, tq, workspace, search_bytes,
                     out_item_ids, out_scores, out_index, status, counters, nullptr, options, plan, st);
}

int nann_search_model(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                      const int32_t level_topn[6], void* workspace, int64_t workspace_bytes,
                      int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                      int32_t* counters, nann_stream_t stream) {
  return nann_search_model_opt(ix, m, comm_seq_f16, n_queries, level_topn, nullptr, workspace, workspace_bytes, out_item_ids,
                               out_scores, out_index, status, counters, nullptr, nullptr, stream);  // deprecated: thin wrapper
}

int nann_search_model_v(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                        const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                        int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                        int32_t* status, int32_t* counters, nann_stream_t stream) {
  return nann_search_model_opt(ix, m, comm_seq_f16, n_queries, level_topn_max, level_topn, workspace, workspace_bytes,
                               out_item_ids, out_scores, out_index, status, counters, nullptr, nullptr, stream);  // deprecated: thin wrapper
}

int nann_search_model_opt(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                          const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                          int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                          int32_t* status, int32_t* counters, const nann_search_options* options, nann_search_plan* plan,
                          nann_stream_t stream) {
  return search_model_impl(ix, m, comm_seq_f16, n_queries, level_topn_max, level_topn, workspace, workspace_bytes,
                           out_item_ids, out_scores, out_index, status, counters, options, plan, stream);
}

int nann_model_prepare(const nann_model* m, const nann_index* ix, nann_stream_t stream) {
  if (!m || !ix) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_prepare: null argument");
  return prepare_impl(m->scorer, m->kind == NANN_MODEL_ATTENTION ? m->attn : nullptr, ix, as_stream(stream));
}
int nann_model_release(const nann_model* m, const nann_index* ix) {
  if (!m || !ix) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_release: null argument");
  const nann_attn_scorer* at = m->kind == NANN_MODEL_ATTENTION ? m->attn : nullptr;
  if (table_width(m->scorer, at) == 0) return NANN_OK;
  return projection_release(at ? at->proj : m->scorer->proj, ix);
}
int nann_model_table_bytes(const nann_model* m, const nann_index* ix, int64_t* table_bytes, int64_t* resident_bytes) {
  if (!m) return fail(NANN_ERR_BAD_ARGUMENT, "nann_model_table_bytes: null model");
  return table_bytes_impl(m->scorer, m->kind == NANN_MODEL_ATTENTION ? m->attn : nullptr, ix, table_bytes, resident_bytes);
}

// ---- the evaluation graph's traversal (nann_eval.h) ----------------------------------------
}  // extern "C"

// l2: the L2 scorer's instances may keep `seen` in LDS (nann_eval.h) -- a window of the index's bitmap beside their scratch.
// seen_lds: 0 = the slot form, 1 = the LDS form, 2 = the LDS form with more than one window.  l2 == false also sizes the
// workspace: the most slots any plan of this index takes.
struct EvalPlan {
  int cat_cap = 0, slots = 0, seen_lds = 0, n_windows = 1;
  unsigned long long slot_bytes = 0;
  uint32_t vis_words = 0, lds_words = 0, win_owners = 0;
};
static int eval_plan(const nann_index* ix, int64_t n_queries, bool l2, EvalPlan* p) {
  DeviceInfo di;
  const int rc = device_info(&di);
  if (rc) return rc;
  // result || next: a frontier holds at most kEvalMaxK rows, a set at most every item
  const int64_t deg = std::max<int64_t>(std::max(ix->max_deg[0], ix->max_deg[1]), 1);
  const int64_t nxt = std::max<int64_t>(std::min<int64_t>(ix->desc.n_items, (int64_t)kEvalMaxK * deg), ix->desc.n_enter);
  if (kEvalMaxK + nxt > 0x3fffffffll) return fail(NANN_ERR_UNSUPPORTED, "candidate bound too large");
  p->cat_cap = (int)(kEvalMaxK + nxt);
  const size_t lds = eval_l2_lds_base() + eval_seen_lds_bytes(ix->bm_words);
  static const bool force_hbm = [] { const char* e = std::getenv("NANN_EVAL_SEEN"); return e && std::string(e) == "hbm"; }();
  // (a thread of the LDS form owns 32 words of the bitmaps' current window; a round sweeps the id space window by window:
  //  nann_eval.h, kEvalWinOwners / kEvalMaxWindows)
  const bool lds_ok = lds <= di.lds_max && !force_hbm && eval_windows(ix->bm_words) <= kEvalMaxWindows && deg < 65536;  // (a row's length rides in 16 bits of its packed bounds)
  p->win_owners = eval_win_owners(ix->bm_words);
  p->n_windows = (int)eval_windows(ix->bm_words);
  p->seen_lds = (l2 && lds_ok) ? (p->n_windows > 1 ? 2 : 1) : 0;
  p->lds_words = (uint32_t)(eval_seen_lds_bytes(ix->bm_words) / 4);
  p->vis_words = eval_vis_words(ix->bm_words);
  unsigned long long off[7];
  p->slot_bytes = eval_slot_layout(ix->bm_words, p->vis_words, p->cat_cap, off);
  // workgroups per CU: two (the slot form: 2048 threads) unless the LDS bitmap leaves room for one
  const int per_cu = p->seen_lds ? 1 : 2;
  p->slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_queries, (int64_t)di.cus * per_cu));
  return NANN_OK;
}

static int eval_impl(const nann_index* ix, const nann_scorer* scorer, const nann_attn_scorer* attn, const float* q,
                     const float* kt, const float* upad, int64_t n_queries, const int32_t num_scoring[3],
                     const int32_t top_k[3], int32_t topk_eval, void* workspace, int64_t workspace_bytes,
                     int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* n_out, int32_t* status,
                     hipStream_t st, int32_t* counters = nullptr) {
  if (n_queries > 0x7fffffffll) return fail(NANN_ERR_UNSUPPORTED, "too many queries in one call");
  if (num_scoring[2] != 1) return fail(NANN_ERR_BAD_ARGUMENT, "num_scoring_per_level[2] must be 1 (model.py:347)");
  for (int l = 0; l < 3; ++l)
    if (top_k[l] < 1 || top_k[l] > kEvalMaxK || num_scoring[l] < 0)
      return fail(NANN_ERR_UNSUPPORTED, "top_k_per_level entries must be in [1, 2048]");
  if (topk_eval < 1 || topk_eval > kEvalMaxK) return fail(NANN_ERR_UNSUPPORTED, "topk_eval must be in [1, 2048]");
  EvalArgs a;
  const bool l2 = !attn && scorer->desc.kind != NANN_SCORER_MLP;
  EvalPlan pl, sizing;
  int rc = eval_plan(ix, n_queries, l2, &pl);
  if (!rc) rc = eval_plan(ix, n_queries, false, &sizing);  // (the workspace the caller sized: nann_search_eval_workspace_bytes)
  if (rc) return rc;
  const int slots = pl.slots, seen_lds = pl.seen_lds;
  a.cat_cap = pl.cat_cap;
  a.slot_bytes = sizing.slot_bytes;  // every plan lays its slots out at the sizing stride (its own arrays sit at the front)
  if (!workspace || workspace_bytes < (int64_t)(256 + a.slot_bytes * (unsigned long long)slots))
    return fail(NANN_ERR_CAPACITY, "workspace smaller than nann_search_eval_workspace_bytes()");
  a.emb = ix->desc.item_embs;
  a.item_ids = ix->desc.item_ids;
  for (int l = 0; l < 2; ++l) { a.nbv[l] = ix->desc.nb_values[l]; a.nbrs[l] = ix->desc.nb_row_splits[l]; }
  a.enter = ix->desc.enter_points;
  a.n_enter = (int)ix->desc.n_enter;
  a.n_items = (uint32_t)ix->desc.n_items;
  a.d = ix->desc.d;
  a.q = q;
  a.n_queries = (int)n_queries;
  for (int l = 0; l < 3; ++l) { a.num_scoring[l] = num_scoring[l]; a.top_k[l] = top_k[l]; }
  a.topk_eval = topk_eval;
  a.ws = static_cast<unsigned char*>(workspace);
  a.bm_words = ix->bm_words;
  // the second-level bitmap (nann_eval.h, round 6): one bit per word of `seen`, overlaid on the phase scratch behind the scan
  // scratch -- it fits shards of up to ~7 M items; beyond, the slot form keeps round 4's full scans
  a.use_dirty = 256 + (size_t)((ix->bm_words + 31u) >> 5) * 4 <= eval_dirty_room() ? 1 : 0;
  a.vis_words = pl.vis_words;
  a.lds_words = pl.lds_words;
  a.win_owners = pl.win_owners;
  a.n_windows = pl.n_windows;
  a.out_ids = out_item_ids; a.out_scores = out_scores; a.out_index = out_index; a.n_out = n_out; a.status = status;
  a.counters = counters;
  a.mlp = MlpParams{};
  a.attn = AttnParams{};
  a.kt = kt; a.upad = upad;
  HIP_TRY(hipMemsetAsync(workspace, 0, 256, st));
  const int dt = ix->desc.emb_dtype, d = ix->desc.d;
  // measurement builds of nann_eval.h (-DNANN_EVAL_TICKS=1, tools/build_res_variant.py): per-phase ticks of the launch on stderr
  static const bool want_ticks = [] { const char* e = std::getenv("NANN_EVAL_TICKS"); return e && e[0] == '1'; }();
  a.ticks = nullptr;
  if (want_ticks && !attn && scorer->desc.kind != NANN_SCORER_MLP) {
    static unsigned long long* g_ticks = nullptr;
    if (!g_ticks) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&g_ticks), 16 * 8));
    HIP_TRY(hipMemsetAsync(g_ticks, 0, 16 * 8, st));
    a.ticks = g_ticks;
    rc = launch_eval_l2(d / 8, dt, seen_lds, slots, a, st);
    unsigned long long h[16];
    HIP_TRY(hipMemcpyAsync(h, g_ticks, sizeof h, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::fprintf(stderr, "EVALTICKS users %lld (100 MHz ticks, sum over users) entry_score %llu entry_topk %llu level_start %llu gather %llu scan %llu score %llu topk %llu select %llu | scan: pass1 %llu wgscan %llu (emit = scan)\n",
                 (long long)n_queries, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
    return rc;
  }
  if (attn) {
    a.attn = attn->P;
    return launch_eval_attn(d, dt, slots, a, st);
  }
  if (scorer->desc.kind == NANN_SCORER_MLP) {
    a.mlp = scorer->mlp;  // (always the f32 MFMA form: the evaluation job is the accuracy reference)
    if (d == 64) return launch_eval_mlp_d64(dt, slots, a, st);
    if (d == 128) return launch_eval_mlp_d128(dt, slots, a, st);
    if (d == 256) return launch_eval_mlp_d256(dt, slots, a, st);
    return fail(NANN_ERR_UNSUPPORTED, "MLP scorer: d in {64, 128, 256}");
  }
  return launch_eval_l2(d / 8, dt, seen_lds, slots, a, st);
}

extern "C" {

int nann_search_eval_workspace_bytes(const nann_index* ix, const nann_model* m, int64_t n_queries, int64_t* nbytes) {
  if (!ix || !nbytes) return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_eval_workspace_bytes: null argument");
  EvalPlan sizing;
  const int rc = eval_plan(ix, n_queries, false, &sizing);  // (the most slots, the widest `visited` any scorer's plan takes)
  if (rc) return rc;
  *nbytes = (int64_t)(256 + sizing.slot_bytes * (unsigned long long)sizing.slots + 256 + (m ? model_query_bytes(m, n_queries) : 0));
  return NANN_OK;
}

int nann_search_eval(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                     const int32_t num_scoring_per_level[3], const int32_t top_k_per_level[3], int32_t topk_eval,
                     void* workspace, int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                     int32_t* out_index, int32_t* n_out, int32_t* status, nann_stream_t stream) {
  if (!ix || !scorer || !q || !num_scoring_per_level || !top_k_per_level || !out_item_ids || !n_out || !status)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_eval: null argument");
  if (n_queries <= 0) return NANN_OK;
  if (scorer->desc.d != ix->desc.d || scorer->desc.emb_dtype != ix->desc.emb_dtype)
    return fail(NANN_ERR_BAD_ARGUMENT, "scorer and index disagree on d / dtype");
  return eval_impl(ix, scorer, nullptr, q, nullptr, nullptr, n_queries, num_scoring_per_level, top_k_per_level, topk_eval,
                   workspace, workspace_bytes, out_item_ids, out_scores, out_index, n_out, status, as_stream(stream));
}

int nann_search_eval_ex(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                        const int32_t num_scoring_per_level[3], const int32_t top_k_per_level[3], int32_t topk_eval,
                        void* workspace, int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                        int32_t* out_index, int32_t* n_out, int32_t* status, int32_t* counters, nann_stream_t stream) {
  if (!ix || !scorer || !q || !num_scoring_per_level || !top_k_per_level || !out_item_ids || !n_out || !status)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_eval_ex: null argument");
  if (n_queries <= 0) return NANN_OK;
  if (scorer->desc.d != ix->desc.d || scorer->desc.emb_dtype != ix->desc.emb_dtype)
    return fail(NANN_ERR_BAD_ARGUMENT, "scorer and index disagree on d / dtype");
  return eval_impl(ix, scorer, nullptr, q, nullptr, nullptr, n_queries, num_scoring_per_level, top_k_per_level, topk_eval,
                   workspace, workspace_bytes, out_item_ids, out_scores, out_index, n_out, status, as_stream(stream), counters);
}

int nann_search_eval_model(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                           const int32_t num_scoring_per_level[3], const int32_t top_k_per_level[3],
                           int32_t topk_eval, void* workspace, int64_t workspace_bytes, int64_t* out_item_ids,
                           float* out_scores, int32_t* out_index, int32_t* n_out, int32_t* status,
                           nann_stream_t stream) {
  if (!ix || !m || !comm_seq_f16 || !num_scoring_per_level || !top_k_per_level || !out_item_ids || !n_out || !status ||
      !workspace)
    return fail(NANN_ERR_BAD_ARGUMENT, "nann_search_eval_model: null argument");
  if (n_queries <= 0) return NANN_OK;
  if (m->d != ix->desc.d || m->emb_dtype != ix->desc.emb_dtype)
    return fail(NANN_ERR_BAD_ARGUMENT, "model and index disagree on d / dtype");
  int64_t need = 0;
  int rc = nann_search_eval_workspace_bytes(ix, m, n_queries, &need);
  if (rc) return rc;
  if (workspace_bytes < need) return fail(NANN_ERR_CAPACITY, "workspace smaller than nann_search_eval_workspace_bytes()");
  const size_t qb = model_query_bytes(m, n_queries);
  const int64_t search_bytes = need - (int64_t)qb - 256;
  unsigned char* qbuf = static_cast<unsigned char*>(workspace) + ((search_bytes + 255) & ~255ll);
  hipStream_t st = as_stream(stream);
  if (m->kind == NANN_MODEL_ATTENTION) {
    float* kt = reinterpret_cast<float*>(qbuf);
    float* upad = kt + (size_t)n_queries * 256 * 64;
    // (the evaluation job always runs the f32 form of the model, whatever precision serving uses)
    rc = launch_attn_prepare(st, m->attn->P, comm_seq_f16, n_queries, kt, upad);
    if (rc) return rc;
    return eval_impl(ix, nullptr, m->attn, nullptr, kt, upad, n_queries, num_scoring_per_level, top_k_per_level,
                     topk_eval, workspace, search_bytes, out_item_ids, out_scores, out_index, n_out, status, st);
  }
  float* q = reinterpret_cast<float*>(qbuf);
  rc = nann_user_seq_mean(comm_seq_f16, n_queries, m->seq_len, m->d, q, stream);
  if (rc) return rc;
  return eval_impl(ix, m->scorer, nullptr, q, nullptr, nullptr, n_queries, num_scoring_per_level, top_k_per_level,
                   topk_eval, workspace, search_bytes, out_item_ids, out_scores, out_index, n_out, status, st);
}

// ---- merge ------------------------------------------------------------------------------
int nann_merge_topk(const float* scores, const int64_t* ids, int64_t n_queries, int32_t n_shards,
                    int32_t k_in, int32_t k_out, float* out_scores, int64_t* out_ids,
                    nann_stream_t stream) {
  const int64_t n_in = (int64_t)n_shards * k_in;
  if (k_out < 0 || k_out > kMaxK) return fail(NANN_ERR_UNSUPPORTED, "k_out must be in [0, 1024]");
  if (n_in < k_out) return fail(NANN_ERR_TOPK_K_GT_N, "fewer candidates than k_out");
  if (n_queries <= 0 || k_out == 0) return NANN_OK;
  hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)n_queries), dim3(kNT), 0, as_stream(stream), scores,
                     ids, (int)n_in, (int)k_out, out_scores, out_ids);
  HIP_TRY(hipGetLastError());
  return NANN_OK;
}

int nann_merge_topk_host(const float* scores, const int64_t* ids, int64_t n_queries, int32_t n_shards,
                         int32_t k_in, int32_t k_out, float* out_scores, int64_t* out_ids) {
  const int64_t n_in = (int64_t)n_shards * k_in;
  if (k_out < 0) return fail(NANN_ERR_BAD_ARGUMENT, "k_out < 0");
  if (n_in < k_out) return fail(NANN_ERR_TOPK_K_GT_N, "fewer candidates than k_out");
  std::vector<int32_t> idx((size_t)n_in);
  for (int64_t qi = 0; qi < n_queries; ++qi) {
    const float* s = scores + qi * n_in;
    for (int64_t i = 0; i < n_in; ++i) idx[(size_t)i] = (int32_t)i;
    std::partial_sort(idx.begin(), idx.begin() + k_out, idx.end(), [s](int32_t x, int32_t y) {
      if (s[y] < s[x]) return true;   // value descending
      if (s[y] > s[x]) return false;
      return x < y;                   // ties -> lower position (shard-major)
    });
    for (int32_t i = 0; i < k_out; ++i) {
      out_scores[qi * k_out + i] = s[idx[(size_t)i]];
      out_ids[qi * k_out + i] = ids[qi * n_in + idx[(size_t)i]];
    }
  }
  return NANN_OK;
}

}  // extern "C"
