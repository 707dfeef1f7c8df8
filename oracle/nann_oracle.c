/*
 * nann_oracle.c -- CPU restatement of the NANN retrieval hot path.
 * TEST INFRASTRUCTURE ONLY (see nann_oracle.h for the rules and the parity
 * status of each function).  Plain C11, no dependencies beyond libc/libm/
 * pthreads.  Citations are relative to /root/reference/;
 * UO/ = tensorflow/tensorflow/core/user_ops/.
 *
 * Canonical floating-point orders (shared with the HIP kernels):
 *
 *  L2 scorer, d = 8*L, L a power of two:
 *      p[l] = 0; for k in 0..7: t = q[8l+k] - x[8l+k]; p[l] = fmaf(t,t,p[l])
 *      for s = 1,2,4,..,L/2: p[l] = p[l] + p[l^s]      (xor butterfly)
 *      score = 0.0f - p[0]
 *  MLP scorer (x=[q;e], W1 [2d,H1], W2 [H1,H2], w3 [H2]):
 *      u[j]  = b1[j];  for k in 0..d-1:      u[j]  = fmaf(q[k], W1[k][j], u[j])
 *      P[j]  = 0;      for k in ORDER_E(d):  P[j]  = fmaf(e[k], W1[d+k][j], P[j])
 *      a1[j] = u[j] + P[j]        (round 4: the item part is a chain of its own, so
 *                                  that a table of P per item holds the same bits)
 *      h1[j] = prelu(a1[j], alpha1[j])
 *      a2[m] = b2[m];  for k in ORDER_H(H1): a2[m] = fmaf(h1[k], W2[k][m], a2[m])
 *      h2[m] = prelu(a2[m], alpha2[m])
 *      p_s   = 0;      for m in ORDER_O(H2, s): p_s = fmaf(h2[m], w3[m], p_s)   s=0,1
 *      score = p_0 + p_1
 *    ORDER_E(d)  = 0, d/2, 1, d/2+1, ..., d/2-1, d-1
 *    ORDER_H(H)  = for t in 0..H/32-1, r in 0..15: k0 = 32t + (r&3) + 8(r>>2); k0, k0+4
 *    ORDER_O(H,s)= for t in 0..H/32-1, r in 0..15:      32t + (r&3) + 8(r>>2) + 4s
 *    (these are the k orders in which v_mfma_f32_32x32x2_f32 consumes its
 *     operands when the layer-1 accumulators feed layer 2 in place)
 *    prelu(x,a)  = (x > 0 ? x : 0) + a * (x < 0 ? x : 0)     (model_util.py:9-11)
 */
#define _GNU_SOURCE
#include "nann_oracle.h"

#include <math.h>
#include <pthread.h>
#ifdef __F16C__
#include <immintrin.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* ragged validation: GroupGather_kernel.cc:9-16, bitmap_ops.cc:12-19        */
int oracle_validate_ragged(int64_t n_values, const int64_t* row_splits,
                           int64_t n_splits) {
  if (n_splits == 0) return 1;
  if (row_splits[0] != 0) return 2;
  if (row_splits[n_splits - 1] != n_values) return 3;
  return 0;
}

/* ------------------------------------------------------------------------ */
/* GroupGather, unique=true: GroupGather_kernel.cc:91-131.  Per group, the SET of the values of the gathered rows
 * (`std::unordered_set<T>` filled row by row, :98-106), its size the group's row length (:110-114), its members
 * written in the set's iteration order (:121-124) -- which the C++ standard leaves to the library, so any order of
 * a group's distinct values is an answer of the reference.  This restatement (and the HIP op) emits FIRST-OCCURRENCE
 * order: value v precedes w iff v's first copy in the unique=false list precedes w's.  Parity = ret_row_splits equal
 * and every group's values equal as a set; first-occurrence order is the build's own, tested against this function.
 * Validation and the void-input path are the shared head of Compute (:62-77).  */
int oracle_group_gather_unique_i32(const int32_t* pv, int64_t n_pv, const int64_t* prs,
                                   int64_t n_prs, const int64_t* iv, int64_t n_iv,
                                   const int64_t* irs, int64_t n_irs, int32_t* out_values,
                                   int64_t out_cap, int64_t* out_rs, int64_t* n_out,
                                   int64_t* n_out_splits, int* ragged_code) {
  int code = oracle_validate_ragged(n_pv, prs, n_prs);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_PARAMS; /* :62-64 */
  code = oracle_validate_ragged(n_iv, irs, n_irs);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_INDICES; /* :65-67 */
  if (n_prs == 1 || n_irs == 1) { /* :69-77 */
    out_rs[0] = 0;
    *n_out = 0;
    *n_out_splits = 1;
    return ORACLE_OK;
  }
  const int64_t num_groups = n_irs - 1, n_rows = n_prs - 1;
  out_rs[0] = 0;
  *n_out_splits = n_irs;
  int64_t sum = 0;
  for (int64_t i = 0; i < num_groups; ++i) {
    /* the group's list as unique=false would emit it, then an open-addressing set over it */
    int64_t len = 0;
    for (int64_t j = irs[i]; j < irs[i + 1]; ++j) {
      const int64_t idx = iv[j];
      if (idx < 0 || idx >= n_rows) return ORACLE_ERR_INDEX_OUT_OF_RANGE; /* UB in the reference */
      len += prs[idx + 1] - prs[idx];
    }
    uint64_t cap = 16;
    while (cap < 2 * (uint64_t)len) cap <<= 1;
    int32_t* keys = (int32_t*)malloc(cap * 4);
    unsigned char* used = (unsigned char*)calloc(cap, 1);
    if (!keys || !used) { free(keys); free(used); return ORACLE_ERR_BAD_ARGUMENT; }
    int64_t w = sum;
    for (int64_t j = irs[i]; j < irs[i + 1]; ++j) {
      const int64_t g = iv[j];
      for (int64_t k = prs[g]; k < prs[g + 1]; ++k) {
        const int32_t v = pv[k];
        uint64_t h = ((uint64_t)(uint32_t)v * 0x9E3779B97F4A7C15ull) >> 20 & (cap - 1);
        while (used[h] && keys[h] != v) h = (h + 1) & (cap - 1);
        if (used[h]) continue; /* a later copy: the set already holds v (:103-104) */
        used[h] = 1;
        keys[h] = v;
        if (out_values) {
          if (w >= out_cap) { free(keys); free(used); return ORACLE_ERR_BAD_ARGUMENT; }
          out_values[w] = v;
        }
        ++w;
      }
    }
    free(keys);
    free(used);
    sum = w;
    out_rs[i + 1] = sum; /* :110-114 */
  }
  *n_out = sum;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* GroupGather, unique=false: GroupGather_kernel.cc:136-170                  */
int oracle_group_gather_i32(const int32_t* pv, int64_t n_pv, const int64_t* prs,
                            int64_t n_prs, const int64_t* iv, int64_t n_iv,
                            const int64_t* irs, int64_t n_irs, int32_t* out_values,
                            int64_t out_cap, int64_t* out_rs, int64_t* n_out,
                            int64_t* n_out_splits, int* ragged_code) {
  int code = oracle_validate_ragged(n_pv, prs, n_prs);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_PARAMS; /* :62-64 */
  code = oracle_validate_ragged(n_iv, irs, n_irs);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_INDICES; /* :65-67 */

  if (n_prs == 1 || n_irs == 1) { /* void inputs -> ([], [0])  :69-77 */
    out_rs[0] = 0;
    *n_out = 0;
    *n_out_splits = 1;
    return ORACLE_OK;
  }
  const int64_t num_groups = n_irs - 1;
  const int64_t n_rows = n_prs - 1;
  out_rs[0] = 0;
  *n_out_splits = n_irs;
  /* count pass :137-145 */
  int64_t sum = 0;
  for (int64_t i = 0; i < num_groups; ++i) {
    for (int64_t j = irs[i]; j < irs[i + 1]; ++j) {
      const int64_t idx = iv[j];
      if (idx < 0 || idx >= n_rows) return ORACLE_ERR_INDEX_OUT_OF_RANGE; /* UB in the reference */
      sum += prs[idx + 1] - prs[idx];
    }
    out_rs[i + 1] = sum;
  }
  *n_out = sum;
  if (!out_values) return ORACLE_OK;
  if (out_cap < sum) return ORACLE_ERR_BAD_ARGUMENT;
  /* fill pass :152-168 */
  for (int64_t i = 0; i < num_groups; ++i) {
    int64_t w = out_rs[i];
    for (int64_t j = irs[i]; j < irs[i + 1]; ++j) {
      const int64_t g = iv[j];
      for (int64_t k = prs[g]; k < prs[g + 1]; ++k) out_values[w++] = pv[k];
    }
  }
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* BitmapRefDifference: bitmap_ops.cc:175-257 (hot loop :224-232)            */
int oracle_bitmap_ref_difference_i32(const int32_t* values, int64_t n_values,
                                     const int64_t* row_splits, int64_t n_splits,
                                     int32_t* bitmap, int64_t n_words,
                                     int32_t* out_values, int64_t* out_rs,
                                     int64_t* n_out, int64_t* n_out_splits,
                                     int* ragged_code) {
  const int code = oracle_validate_ragged(n_values, row_splits, n_splits);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_INPUT; /* :182-184 */
  if (n_splits == 1) { /* void input :187-196, bitmap forwarded untouched */
    out_rs[0] = 0;
    *n_out = 0;
    *n_out_splits = 1;
    return ORACLE_OK;
  }
  const int64_t num_groups = n_splits - 1;
  /* bounds pre-check (the reference has none: Appendix C) so that an error
   * leaves the bitmap untouched */
  for (int64_t j = 0; j < n_values; ++j) {
    const int64_t v = values[j];
    if (v < 0 || (v >> 5) >= n_words) return ORACLE_ERR_INDEX_OUT_OF_RANGE;
  }
  uint32_t* bm = (uint32_t*)bitmap;
  int64_t w = 0;
  out_rs[0] = 0;
  for (int64_t i = 0; i < num_groups; ++i) { /* ONE bitmap for all groups */
    for (int64_t j = row_splits[i]; j < row_splits[i + 1]; ++j) {
      const int32_t node = values[j];
      const int32_t flag_index = node >> 5;
      const uint32_t bit = 1u << (node & 31);
      if (!(bm[flag_index] & bit)) {
        out_values[w++] = node;
        bm[flag_index] |= bit;
      }
    }
    out_rs[i + 1] = w;
  }
  *n_out = w;
  *n_out_splits = n_splits;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* sibling ops                                              */
int oracle_bitmap_init_i32(const int32_t* idx, int64_t n_idx, int32_t length, int32_t* bitmap) {
  if (length < 0 || n_idx > length) return ORACLE_ERR_BAD_ARGUMENT; /* bitmap_ops.cc:56-57 */
  memset(bitmap, 0, (size_t)length * 4);
  for (int64_t i = 0; i < n_idx; ++i) { /* :65-72 */
    const int32_t node = idx[i];
    if (node < 0 || (node >> 5) >= length) return ORACLE_ERR_INDEX_OUT_OF_RANGE; /* UB in the reference */
    ((uint32_t*)bitmap)[node >> 5] |= 1u << (node & 31);
  }
  return ORACLE_OK;
}

int oracle_bitmap_difference_i32(const int32_t* idx_next, int64_t n, const int32_t* idx_flag,
                                 int64_t n_words, int32_t* idx_next_new, int64_t* n_out,
                                 int32_t* idx_flag_new) {
  for (int64_t i = 0; i < n; ++i)
    if (idx_next[i] < 0 || (idx_next[i] >> 5) >= n_words) return ORACLE_ERR_INDEX_OUT_OF_RANGE;
  memcpy(idx_flag_new, idx_flag, (size_t)n_words * 4); /* :117-121 */
  uint32_t* bm = (uint32_t*)idx_flag_new;
  int64_t w = 0;
  for (int64_t i = 0; i < n; ++i) { /* :124-132 */
    const int32_t node = idx_next[i];
    const uint32_t bit = 1u << (node & 31);
    if (!(bm[node >> 5] & bit)) {
      idx_next_new[w++] = node;
      bm[node >> 5] |= bit;
    }
  }
  *n_out = w;
  return ORACLE_OK;
}

typedef struct { const float* v; int asc; } rt_cmp_t;
static int rt_cmp(const void* pa, const void* pb, void* ctx) {
  const rt_cmp_t* c = (const rt_cmp_t*)ctx;
  const int64_t a = *(const int64_t*)pa, b = *(const int64_t*)pb;
  const float va = c->v[a], vb = c->v[b];
  if (c->asc ? va < vb : va > vb) return -1;
  if (c->asc ? va > vb : va < vb) return 1;
  return a < b ? -1 : (a > b ? 1 : 0); /* ties: lower position (unspecified in the reference) */
}

/* ------------------------------------------------------------------------ */
/* FarmHash Fingerprint64 (farmhashna::Hash64) for len <= 32                  */
static const uint64_t FH_K0 = 0xc3a5c85c97cb3127ULL, FH_K1 = 0xb492b66fbe98f273ULL, FH_K2 = 0x9ae16a3b2f90404fULL;
static uint64_t fh_fetch64(const char* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint64_t fh_fetch32(const char* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t fh_rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
static uint64_t fh_len16(uint64_t u, uint64_t v, uint64_t mul) {
  uint64_t a = (u ^ v) * mul; a ^= (a >> 47);
  uint64_t b = (v ^ a) * mul; b ^= (b >> 47);
  return b * mul;
}
uint64_t oracle_fingerprint64(const char* s, int64_t len) {
  const uint64_t n = (uint64_t)len;
  if (len <= 16) {
    if (len >= 8) {
      const uint64_t mul = FH_K2 + n * 2, a = fh_fetch64(s) + FH_K2, b = fh_fetch64(s + len - 8);
      const uint64_t c = fh_rot(b, 37) * mul + a, d = (fh_rot(a, 25) + b) * mul;
      return fh_len16(c, d, mul);
    }
    if (len >= 4) {
      const uint64_t mul = FH_K2 + n * 2, a = fh_fetch32(s);
      return fh_len16(n + (a << 3), fh_fetch32(s + len - 4), mul);
    }
    if (len > 0) {
      const uint8_t a = (uint8_t)s[0], b = (uint8_t)s[len >> 1], c = (uint8_t)s[len - 1];
      const uint32_t y = (uint32_t)a + ((uint32_t)b << 8), z = (uint32_t)n + ((uint32_t)c << 2);
      uint64_t v = (y * FH_K2) ^ (z * FH_K0);
      v ^= v >> 47;
      return v * FH_K2;
    }
    return FH_K2;
  }
  if (len <= 32) {
    const uint64_t mul = FH_K2 + n * 2, a = fh_fetch64(s) * FH_K1, b = fh_fetch64(s + 8);
    const uint64_t c = fh_fetch64(s + len - 8) * mul, d = fh_fetch64(s + len - 16) * FH_K2;
    return fh_len16(fh_rot(a + b, 43) + fh_rot(c, 30) + d, a + fh_rot(b + FH_K2, 18) + c, mul);
  }
  return 0;
}

/* bitmap_ops.cc:384-411: the largest prime <= multi_hash_mod_param[l] * bucket_size * 32 */
static int64_t bloom_prime_below(int64_t num) {
  for (int64_t n = num; n > 1; --n) {
    int prime = 1;
    for (int64_t i = (int64_t)(sqrt((double)n) + 1e-6); i > 1; --i)
      if (n % i == 0) { prime = 0; break; }
    if (prime) return n;
  }
  return 1;
}

void oracle_bloom_positions(int32_t node, int64_t bucket, int64_t bucket_size, int64_t pos[4]) {
  static const int mult[4] = {1, 3, 5, 7}, modp[4] = {29, 47, 67, 83}; /* :296-297 */
  char buf[24];
  const int len = snprintf(buf, sizeof buf, "%d", node); /* std::to_string(node) :346 */
  uint64_t raw = oracle_fingerprint64(buf, len);
  if (bucket > 0) raw = raw % (uint64_t)bucket; /* :348 */
  for (int l = 0; l < 4; ++l) {
    const int64_t prime = bloom_prime_below((int64_t)modp[l] * bucket_size * 32);
    /* uint64 arithmetic as written at :352 (rawHash * mult wraps mod 2^64) */
    const uint64_t tmp = ((raw * (uint64_t)mult[l]) % (uint64_t)prime + (uint64_t)prime) % (uint64_t)prime;
    pos[l] = (int64_t)(tmp % (uint64_t)(bucket_size * 32)); /* :353 */
  }
}

int oracle_bloom_filter_difference_i32(const int32_t* values, int64_t n_values, const int64_t* row_splits,
                                       int64_t n_splits, int32_t* idx_flag, int64_t n_flag_words,
                                       int64_t bucket, int64_t bucket_size, int32_t* out_values,
                                       int64_t* out_rs, int64_t* n_out, int64_t* n_out_splits,
                                       int* ragged_code) {
  const int code = oracle_validate_ragged(n_values, row_splits, n_splits);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_INPUT; /* :310-312 */
  if (bucket < 0 || bucket_size < 1 || n_flag_words < bucket_size) return ORACLE_ERR_BAD_ARGUMENT;
  if (n_splits == 1) { /* void input :315-325 */
    out_rs[0] = 0;
    *n_out = 0;
    *n_out_splits = 1;
    return ORACLE_OK;
  }
  uint32_t* bm = (uint32_t*)idx_flag;
  int64_t w = 0;
  out_rs[0] = 0;
  for (int64_t i = 0; i < n_splits - 1; ++i) {
    for (int64_t j = row_splits[i]; j < row_splits[i + 1]; ++j) {
      int64_t pos[4];
      oracle_bloom_positions(values[j], bucket, bucket_size, pos);
      int miss = 0;
      for (int l = 0; l < 4; ++l) { /* :350-361: test and set, one position after the other */
        const uint32_t bit = 1u << (pos[l] & 31);
        if (!(bm[pos[l] >> 5] & bit)) { ++miss; bm[pos[l] >> 5] |= bit; }
      }
      if (miss > 0) out_values[w++] = values[j];
    }
    out_rs[i + 1] = w;
  }
  *n_out = w;
  *n_out_splits = n_splits;
  return ORACLE_OK;
}

int oracle_batch_topk_on_rt_f32(const float* values, int64_t n_values, const int64_t* row_splits,
                                int64_t n_splits, const int64_t* k, int k_is_scalar, int ascending,
                                float* values_out, int64_t* idx_out, int64_t* row_splits_out,
                                int64_t* n_out, int64_t* n_out_splits, int* ragged_code) {
  const int code = oracle_validate_ragged(n_values, row_splits, n_splits);
  if (ragged_code) *ragged_code = code;
  if (code) return ORACLE_ERR_INVALID_RAGGED_INPUT; /* BatchTopKOnRT_kernel.cc:73-75 */
  const int64_t groups = n_splits - 1;
  row_splits_out[0] = 0;
  if (groups == 0) { /* void input :84-92 */
    *n_out = 0; *n_out_splits = 1;
    return ORACLE_OK;
  }
  int64_t* tmp = (int64_t*)malloc((size_t)(n_values ? n_values : 1) * 8);
  if (!tmp) return ORACLE_ERR_BAD_ARGUMENT;
  rt_cmp_t ctx = {values, ascending};
  int64_t w = 0;
  for (int64_t g = 0; g < groups; ++g) {
    const int64_t s = row_splits[g], e = row_splits[g + 1], len = e - s;
    int64_t kk = k_is_scalar ? k[0] : k[g];
    if (kk < 0) kk = 0;
    if (kk > len) kk = len; /* min(len, k) :113-117 */
    for (int64_t i = 0; i < len; ++i) tmp[i] = s + i;
    qsort_r(tmp, (size_t)len, 8, rt_cmp, &ctx);
    for (int64_t i = 0; i < kk; ++i) {
      values_out[w] = values[tmp[i]];
      idx_out[w] = tmp[i] - s; /* row-local :145-146 */
      ++w;
    }
    row_splits_out[g + 1] = w;
  }
  free(tmp);
  *n_out = w;
  *n_out_splits = n_splits;
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* GatherV2, axis 0: gather_functor.h:38-116 (memcpy per row :96-103)        */
int oracle_gather_rows(const void* params, int64_t n_rows, int64_t row_bytes,
                       const int32_t* idx, int64_t n_idx, void* out, int64_t* bad_i) {
  const char* p = (const char*)params;
  char* o = (char*)out;
  for (int64_t i = 0; i < n_idx; ++i) {
    const int64_t r = idx[i];
    if (r < 0 || r >= n_rows) { /* FastBoundsCheck, gather_functor.h:85-89 */
      if (bad_i) *bad_i = i;
      return ORACLE_ERR_INDEX_OUT_OF_RANGE;
    }
    memcpy(o + i * row_bytes, p + r * row_bytes, (size_t)row_bytes);
  }
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* TopKV2: topk_op.cc:104-205.  stable_comp (:134-142): a before b iff
 * v[a] > v[b] or (v[a] == v[b] and a < b).                                  */
static inline int topk_better(const float* v, int32_t a, int32_t b) {
  if (v[b] < v[a]) return 1;
  if (v[b] > v[a]) return 0;
  return a < b;
}

static int topk_qsort_cmp(const void* pa, const void* pb, void* ctx) {
  const float* v = (const float*)ctx;
  const int32_t a = *(const int32_t*)pa, b = *(const int32_t*)pb;
  if (a == b) return 0;
  return topk_better(v, a, b) ? -1 : 1;
}

int oracle_topk_f32(const float* values, int64_t n, int32_t k, float* out_values,
                    int32_t* out_indices) {
  if (k < 0) return ORACLE_ERR_BAD_ARGUMENT;
  if (n < k) return ORACLE_ERR_TOPK_K_GT_N; /* :67-71 */
  if (k == 0) return ORACLE_OK;
  /* bounded heap whose root is the WORST kept element (gtl::TopN, :176-193) */
  int32_t* heap = out_indices;
  int32_t size = 0;
  for (int32_t c = 0; c < (int32_t)n; ++c) {
    if (size < k) {
      int32_t i = size++;
      heap[i] = c;
      while (i > 0) { /* sift up: parent must be worse-or-equal than child */
        const int32_t p = (i - 1) >> 1;
        if (topk_better(values, heap[p], heap[i])) {
          const int32_t t = heap[p]; heap[p] = heap[i]; heap[i] = t;
          i = p;
        } else break;
      }
    } else if (topk_better(values, c, heap[0])) {
      heap[0] = c;
      int32_t i = 0;
      for (;;) {
        const int32_t l = 2 * i + 1, r = l + 1;
        int32_t w = i; /* w = worst among i, l, r */
        if (l < size && topk_better(values, heap[w], heap[l])) w = l;
        if (r < size && topk_better(values, heap[w], heap[r])) w = r;
        if (w == i) break;
        const int32_t t = heap[w]; heap[w] = heap[i]; heap[i] = t;
        i = w;
      }
    }
  }
  qsort_r(heap, (size_t)k, sizeof(int32_t), topk_qsort_cmp, (void*)values);
  for (int32_t i = 0; i < k; ++i) out_values[i] = values[out_indices[i]];
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* exact scalar conversions                                                  */
float oracle_half_to_float(uint16_t h) {
#ifdef __F16C__
  return _cvtsh_ss(h); /* exact, same value as the bit-level path below */
#endif
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t exp = (h >> 10) & 0x1f;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else { /* subnormal: normalise */
      int e = -1;
      do { man <<= 1; ++e; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

float oracle_bf16_to_float(uint16_t h) {
  const uint32_t bits = (uint32_t)h << 16;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

uint16_t oracle_float_to_half(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    man |= 0x800000u;
    const int shift = 14 - exp; /* 14..24 */
    uint32_t hm = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) ++hm;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13;
  const uint32_t rem = man & 0x1fffu;
  uint32_t out = sign | ((uint32_t)exp << 10) | hm;
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) ++out; /* carries into exp correctly */
  return (uint16_t)out;
}

static inline float load_elem(const void* row, int dtype, int k) {
  switch (dtype) {
    case ORACLE_EMB_F16: return oracle_half_to_float(((const uint16_t*)row)[k]);
    case ORACLE_EMB_BF16: return oracle_bf16_to_float(((const uint16_t*)row)[k]);
    default: return ((const float*)row)[k];
  }
}

static inline int64_t elem_bytes(int dtype) { return dtype == ORACLE_EMB_F32 ? 4 : 2; }

void oracle_user_seq_mean(const uint16_t* seq, int seq_len, int d, float* q) {
  int count = 0;
  for (int r = 0; r < seq_len; ++r) {
    int nz = 0;
    for (int k = 0; k < d; ++k) nz |= (seq[(int64_t)r * d + k] & 0x7fffu) != 0;
    count += nz;
  }
  for (int k = 0; k < d; ++k) {
    float s = 0.0f;
    for (int r = 0; r < seq_len; ++r) s = s + oracle_half_to_float(seq[(int64_t)r * d + k]);
    q[k] = count ? s / (float)count : 0.0f;
  }
}

/* ------------------------------------------------------------------------ */
/* scorers                                                                   */
static inline float prelu(float x, float a) {
  const float pos = x > 0.0f ? x : 0.0f;
  const float neg = x < 0.0f ? x : 0.0f;
  return pos + a * neg;
}

static int is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

static float score_l2_row(const float* q, const void* row, int dtype, int d) {
  float p[64], t2[64];
  const int L = d / 8;
  for (int l = 0; l < L; ++l) {
    float acc = 0.0f;
    for (int k = 0; k < 8; ++k) {
      const float t = q[8 * l + k] - load_elem(row, dtype, 8 * l + k);
      acc = fmaf(t, t, acc);
    }
    p[l] = acc;
  }
  for (int s = 1; s < L; s <<= 1) {
    for (int l = 0; l < L; ++l) t2[l] = p[l] + p[l ^ s];
    memcpy(p, t2, sizeof(float) * (size_t)L);
  }
  return 0.0f - p[0];
}

/* per-query part of the MLP: u[j] = b1[j] + sum_k q[k] W1[k][j] */
static void mlp_query_part(const oracle_scorer_t* sc, const float* q, float* u) {
  for (int j = 0; j < sc->h1; ++j) {
    float acc = sc->b1[j];
    for (int k = 0; k < sc->d; ++k) acc = fmaf(q[k], sc->w1[(int64_t)k * sc->h1 + j], acc);
    u[j] = acc;
  }
}

static float score_mlp_row(const oracle_scorer_t* sc, const float* u, const void* row,
                           float* h1, float* h2) {
  const int d = sc->d, H1 = sc->h1, H2 = sc->h2;
  float e[512];
  for (int k = 0; k < d; ++k) e[k] = load_elem(row, sc->emb_dtype, k);
  for (int j = 0; j < H1; ++j) {
    float acc = 0.0f; /* the item part is its own chain: the same value whoever scores the item */
    for (int kk = 0; kk < d / 2; ++kk) { /* ORDER_E */
      acc = fmaf(e[kk], sc->w1[(int64_t)(d + kk) * H1 + j], acc);
      acc = fmaf(e[d / 2 + kk], sc->w1[(int64_t)(d + d / 2 + kk) * H1 + j], acc);
    }
    h1[j] = prelu(u[j] + acc, sc->alpha1[j]);
  }
  for (int m = 0; m < H2; ++m) {
    float acc = sc->b2[m];
    for (int t = 0; t < H1 / 32; ++t)
      for (int r = 0; r < 16; ++r) { /* ORDER_H */
        const int k0 = 32 * t + (r & 3) + 8 * (r >> 2);
        acc = fmaf(h1[k0], sc->w2[(int64_t)k0 * H2 + m], acc);
        acc = fmaf(h1[k0 + 4], sc->w2[(int64_t)(k0 + 4) * H2 + m], acc);
      }
    h2[m] = prelu(acc, sc->alpha2[m]);
  }
  float p[2];
  for (int s = 0; s < 2; ++s) {
    float acc = 0.0f;
    for (int t = 0; t < H2 / 32; ++t)
      for (int r = 0; r < 16; ++r) { /* ORDER_O */
        const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * s;
        acc = fmaf(h2[m], sc->w3[m], acc);
      }
    p[s] = acc;
  }
  return p[0] + p[1];
}

static int query_floats(const oracle_scorer_t* sc) { /* floats per query of a batch */
  if (sc->kind == ORACLE_SCORER_ATTN && sc->attn) {
    const oracle_attn_model_t* m = (const oracle_attn_model_t*)sc->attn;
    return m->L * m->E;
  }
  return sc->d;
}

static int scorer_ok(const oracle_scorer_t* sc) {
  if (sc && sc->kind == ORACLE_SCORER_ATTN) {
    const oracle_attn_model_t* m = (const oracle_attn_model_t*)sc->attn;
    return m && m->d == sc->d && m->emb_dtype == sc->emb_dtype;
  }
  if (sc->d <= 0 || sc->d % 8 || sc->d > 512 || !is_pow2(sc->d / 8)) return 0;
  if (sc->kind == ORACLE_SCORER_MLP) {
    if (sc->h1 <= 0 || sc->h1 % 32 || sc->h2 <= 0 || sc->h2 % 32) return 0;
    if (sc->h1 > 1024 || sc->h2 > 1024 || sc->d % 2) return 0;
  }
  return 1;
}

int oracle_score_rows(const oracle_scorer_t* sc, const float* q, const void* rows,
                      int64_t n, float* out) {
  if (!scorer_ok(sc)) return ORACLE_ERR_BAD_ARGUMENT;
  if (n <= 0) return ORACLE_ERR_EMPTY_SCORE_BATCH; /* blaze_xla_predictor.cc:259-263 */
  const int64_t rb = (int64_t)sc->d * elem_bytes(sc->emb_dtype);
  if (sc->kind == ORACLE_SCORER_L2) {
    for (int64_t i = 0; i < n; ++i)
      out[i] = score_l2_row(q, (const char*)rows + i * rb, sc->emb_dtype, sc->d);
    return ORACLE_OK;
  }
  if (sc->kind == ORACLE_SCORER_ATTN) { /* q = the user sequence; the per-user projection is redone per call */
    const oracle_attn_model_t* m = (const oracle_attn_model_t*)sc->attn;
    float* kproj = (float*)malloc((size_t)m->L * 4 * m->E * sizeof(float));
    if (!kproj) return ORACLE_ERR_BAD_ARGUMENT;
    int rc = oracle_attn_prepare(m, q, kproj);
    if (!rc) rc = oracle_attn_score_rows(m, q, kproj, rows, n, out);
    free(kproj);
    return rc;
  }
  float u[1024], h1[1024], h2[1024];
  mlp_query_part(sc, q, u);
  for (int64_t i = 0; i < n; ++i)
    out[i] = score_mlp_row(sc, u, (const char*)rows + i * rb, h1, h2);
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* the traversal schedule: build_opt_graph.py:109-149     */
typedef struct {
  int32_t* ids; float* scores; int64_t cap;
} buf_t;

static int ensure(buf_t* b, int64_t n) {
  if (n <= b->cap) return 1;
  int64_t c = b->cap ? b->cap : 1024;
  while (c < n) c *= 2;
  int32_t* ni = (int32_t*)realloc(b->ids, (size_t)c * sizeof(int32_t));
  if (!ni) return 0;
  b->ids = ni;
  float* ns = (float*)realloc(b->scores, (size_t)c * sizeof(float));
  if (!ns) return 0;
  b->scores = ns;
  b->cap = c;
  return 1;
}

/* forward(): GatherV2 + BlazeXlaOp + Squeeze, build_opt_graph.py:91-107 */
static int forward(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                   const int32_t* ids, int64_t n, float* scores, char** tmp,
                   int64_t* tmp_cap) {
  if (n <= 0) return ORACLE_ERR_EMPTY_SCORE_BATCH;
  const int64_t rb = (int64_t)ix->d * elem_bytes(ix->emb_dtype);
  if (n * rb > *tmp_cap) {
    char* t = (char*)realloc(*tmp, (size_t)(n * rb));
    if (!t) return ORACLE_ERR_BAD_ARGUMENT;
    *tmp = t;
    *tmp_cap = n * rb;
  }
  int rc = oracle_gather_rows(ix->item_embs, ix->n_items, rb, ids, n, *tmp, NULL);
  if (rc) return rc;
  rc = oracle_score_rows(sc, q, *tmp, n, scores);
  if (rc) return rc;
  /* tf.squeeze of a [1,1] logits tensor yields a scalar: TopKV2 (:63-65) and
   * ConcatV2 then reject it */
  if (n == 1) return ORACLE_ERR_TOPK_SCALAR_INPUT;
  return ORACLE_OK;
}

/* top_k(): TopKV2 + Gather(ids, indices), build_opt_graph.py:52-66 */
static int topk_ids(const int32_t* ids, const float* scores, int64_t n, int32_t k,
                    int32_t* out_ids, float* out_scores, int32_t* tmp_idx) {
  const int rc = oracle_topk_f32(scores, n, k, out_scores, tmp_idx);
  if (rc) return rc;
  for (int32_t i = 0; i < k; ++i) out_ids[i] = ids[tmp_idx[i]];
  return ORACLE_OK;
}

/* per-thread reusable buffers: the reference keeps its tensors in TF's
 * allocator pools; re-malloc'ing MBs per query would serialise threads on
 * the kernel's mmap lock and misprice the CPU baseline */
typedef struct {
  int32_t* bm; int64_t bm_words;
  buf_t cand, pool, beam;
  char* tmp; int64_t tmp_cap;
  int32_t* tmp_idx; int64_t tmp_idx_cap;
  int32_t* raw; int64_t raw_cap;
  int64_t* iv; int64_t iv_cap;
} search_ws_t;

static void ws_free(search_ws_t* w) {
  free(w->bm); free(w->cand.ids); free(w->cand.scores); free(w->pool.ids); free(w->pool.scores);
  free(w->beam.ids); free(w->beam.scores); free(w->tmp); free(w->tmp_idx); free(w->raw); free(w->iv);
  memset(w, 0, sizeof *w);
}

static int oracle_search_ws(search_ws_t* W, const oracle_index_t* ix, const oracle_scorer_t* sc,
                            const float* q, const int32_t t[6], int64_t* out_item_ids,
                            float* out_scores, int32_t* out_index, oracle_counters_t* ctr) {
  if (!ix || !sc || !scorer_ok(sc) || sc->d != ix->d || sc->emb_dtype != ix->emb_dtype)
    return ORACLE_ERR_BAD_ARGUMENT;
  for (int i = 0; i < 6; ++i)
    if (t[i] < 0) return ORACLE_ERR_BAD_ARGUMENT;
  oracle_counters_t c;
  memset(&c, 0, sizeof c);
  int rc = ORACLE_OK;
  const int64_t n_words = (ix->n_items + 31) / 32; /* build_opt_graph.py:114 */
  if (W->bm_words < n_words) {
    free(W->bm);
    W->bm = (int32_t*)malloc((size_t)(n_words > 0 ? n_words : 1) * 4);
    W->bm_words = W->bm ? n_words : 0;
  }
  int32_t* bm = W->bm;
#define cand (W->cand)
#define pool (W->pool)
#define beam (W->beam)
#define tmp (W->tmp)
#define tmp_cap (W->tmp_cap)
#define tmp_idx (W->tmp_idx)
#define raw (W->raw)
#define raw_cap (W->raw_cap)
  int64_t rs2[2], ors[2], n_out, n_os;

#define FAIL(code) do { rc = (code); goto done; } while (0)
#define CHECK(expr) do { rc = (expr); if (rc) goto done; } while (0)

  /* ---- level 2 (entry layer): lines :111-112 ---- */
  const int64_t E = ix->n_enter;
  if (!ensure(&cand, E + 1) || !bm) FAIL(ORACLE_ERR_BAD_ARGUMENT);
  c.scored[0] = E;
  CHECK(forward(ix, sc, q, ix->enter_points, E, cand.scores, &tmp, &tmp_cap));
  {
    int64_t maxk = E;
    for (int i = 0; i < 6; ++i) if (t[i] > maxk) maxk = t[i];
    const int64_t poolcap = (int64_t)t[1] + t[2] + t[3] + t[4] + 1;
    if (poolcap > maxk) maxk = poolcap;
    if (W->tmp_idx_cap < maxk + 1) {
      free(tmp_idx);
      tmp_idx = (int32_t*)malloc((size_t)(maxk + 1) * 4);
      W->tmp_idx_cap = tmp_idx ? maxk + 1 : 0;
    }
    if (!tmp_idx || !ensure(&beam, maxk + 1) || !ensure(&pool, poolcap))
      FAIL(ORACLE_ERR_BAD_ARGUMENT);
  }
  /* R, sR = topk(EP, s, t0) */
  CHECK(topk_ids(ix->enter_points, cand.scores, E, t[0], beam.ids, beam.scores, tmp_idx));
  int64_t nR = t[0];

  /* ---- level 1: lines :114-127 ---- */
  {
    /* C = ragged_gather(NB1, R)  :116 */
    int64_t irs[2] = {0, nR};
    if (W->iv_cap < nR + 1) {
      free(W->iv);
      W->iv = (int64_t*)malloc((size_t)(nR + 1) * 8);
      W->iv_cap = W->iv ? nR + 1 : 0;
    }
    int64_t* iv = W->iv;
    if (!iv) FAIL(ORACLE_ERR_BAD_ARGUMENT);
    for (int64_t i = 0; i < nR; ++i) iv[i] = beam.ids[i];
    rc = oracle_group_gather_i32(ix->nb_values[1], ix->nb_nnz[1], ix->nb_row_splits[1],
                                 ix->n_items + 1, iv, nR, irs, 2, NULL, 0, ors, &n_out,
                                 &n_os, NULL);
    if (!rc) {
      if (n_out > raw_cap) {
        free(raw);
        raw = (int32_t*)malloc((size_t)(n_out ? n_out : 1) * 4);
        raw_cap = n_out;
      }
      rc = oracle_group_gather_i32(ix->nb_values[1], ix->nb_nnz[1], ix->nb_row_splits[1],
                                   ix->n_items + 1, iv, nR, irs, 2, raw, raw_cap, ors,
                                   &n_out, &n_os, NULL);
    }
    if (rc) goto done;
    const int64_t nG = n_out;
    c.frontier[1] = nR;
    c.gathered[1] = nG;
    /* flags = zeros  :115-118 */
    memset(bm, 0, (size_t)n_words * 4);
    /* R = diff(R, bm)  :119-120 -- marks the entry winners */
    if (!ensure(&cand, nR + nG + 1)) FAIL(ORACLE_ERR_BAD_ARGUMENT);
    rs2[0] = 0; rs2[1] = nR;
    CHECK(oracle_bitmap_ref_difference_i32(beam.ids, nR, rs2, 2, bm, n_words, cand.ids, ors,
                                           &n_out, &n_os, NULL));
    if (n_out != nR) FAIL(ORACLE_ERR_BAD_ARGUMENT); /* duplicate enter points: the
        reference graph would concat ids and scores of different lengths */
    memcpy(cand.scores, beam.scores, (size_t)nR * 4);
    /* C = diff(C, bm)  :121-122 */
    rs2[1] = nG;
    CHECK(oracle_bitmap_ref_difference_i32(raw, nG, rs2, 2, bm, n_words, cand.ids + nR, ors,
                                           &n_out, &n_os, NULL));
    const int64_t nC = n_out;
    c.scored[1] = nC;
    /* sC = forward(C)  :124 */
    CHECK(forward(ix, sc, q, cand.ids + nR, nC, cand.scores + nR, &tmp, &tmp_cap));
    /* P, sP = topk(R || C, sR || sC, t1)  :125-127 */
    CHECK(topk_ids(cand.ids, cand.scores, nR + nC, t[1], pool.ids, pool.scores, tmp_idx));
  }
  int64_t nP = t[1];

  /* ---- level 0: lines :129-141 ---- */
  memset(bm, 0, (size_t)n_words * 4); /* re-Assign zeros :131 */
  rs2[0] = 0; rs2[1] = nP;
  CHECK(oracle_bitmap_ref_difference_i32(pool.ids, nP, rs2, 2, bm, n_words, beam.ids, ors,
                                         &n_out, &n_os, NULL)); /* :132-133 */
  int64_t nB = n_out;
  for (int i = 0; i < 3; ++i) {
    int64_t irs[2] = {0, nB};
    if (W->iv_cap < nB + 1) {
      free(W->iv);
      W->iv = (int64_t*)malloc((size_t)(nB + 1) * 8);
      W->iv_cap = W->iv ? nB + 1 : 0;
    }
    int64_t* iv = W->iv;
    if (!iv) FAIL(ORACLE_ERR_BAD_ARGUMENT);
    for (int64_t j = 0; j < nB; ++j) iv[j] = beam.ids[j];
    rc = oracle_group_gather_i32(ix->nb_values[0], ix->nb_nnz[0], ix->nb_row_splits[0],
                                 ix->n_items + 1, iv, nB, irs, 2, NULL, 0, ors, &n_out,
                                 &n_os, NULL);
    if (!rc) {
      if (n_out > raw_cap) {
        free(raw);
        raw = (int32_t*)malloc((size_t)(n_out ? n_out : 1) * 4);
        raw_cap = n_out;
      }
      rc = oracle_group_gather_i32(ix->nb_values[0], ix->nb_nnz[0], ix->nb_row_splits[0],
                                   ix->n_items + 1, iv, nB, irs, 2, raw, raw_cap, ors,
                                   &n_out, &n_os, NULL); /* :136 */
    }
    if (rc) goto done;
    const int64_t nG = n_out;
    c.frontier[2 + i] = nB;
    c.gathered[2 + i] = nG;
    if (!ensure(&cand, nG + 1)) FAIL(ORACLE_ERR_BAD_ARGUMENT);
    rs2[1] = nG;
    CHECK(oracle_bitmap_ref_difference_i32(raw, nG, rs2, 2, bm, n_words, cand.ids, ors,
                                           &n_out, &n_os, NULL)); /* :137 */
    const int64_t nC = n_out;
    c.scored[2 + i] = nC;
    CHECK(forward(ix, sc, q, cand.ids, nC, cand.scores, &tmp, &tmp_cap)); /* :138 */
    /* B, sB = topk(C, sC, t[2+i]) -- beam = best NEW nodes only  :139 */
    CHECK(topk_ids(cand.ids, cand.scores, nC, t[2 + i], beam.ids, beam.scores, tmp_idx));
    nB = t[2 + i];
    memcpy(pool.ids + nP, beam.ids, (size_t)nB * 4);       /* :140 */
    memcpy(pool.scores + nP, beam.scores, (size_t)nB * 4); /* :141 */
    nP += nB;
  }
  /* ---- final: lines :143-149 ---- */
  CHECK(oracle_topk_f32(pool.scores, nP, t[5], out_scores, tmp_idx));
  for (int32_t i = 0; i < t[5]; ++i) {
    const int32_t ii = pool.ids[tmp_idx[i]];
    if (out_index) out_index[i] = ii;
    out_item_ids[i] = ix->item_ids[ii]; /* Gather(item_ids, P) :144 */
  }
done:
  if (ctr) *ctr = c;
  return rc;
#undef FAIL
#undef CHECK
#undef cand
#undef pool
#undef beam
#undef tmp
#undef tmp_cap
#undef tmp_idx
#undef raw
#undef raw_cap
}

int oracle_search(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                  const int32_t t[6], int64_t* out_item_ids, float* out_scores,
                  int32_t* out_index, oracle_counters_t* ctr) {
  search_ws_t W;
  memset(&W, 0, sizeof W);
  const int rc = oracle_search_ws(&W, ix, sc, q, t, out_item_ids, out_scores, out_index, ctr);
  ws_free(&W);
  return rc;
}


/* ------------------------------------------------------------------------ */
typedef struct {
  const oracle_index_t* ix; const oracle_scorer_t* sc; const float* q;
  int64_t nq; const int32_t* t; int64_t* ids; float* scores; int32_t* index;
  oracle_counters_t* ctr; int32_t* status; int64_t next; pthread_mutex_t mu;
} batch_t;

static void* batch_worker(void* arg) {
  batch_t* b = (batch_t*)arg;
  const int32_t k = b->t[5];
  search_ws_t W;
  memset(&W, 0, sizeof W);
  for (;;) {
    const int64_t i = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
    if (i >= b->nq) break;
    b->status[i] = oracle_search_ws(&W, b->ix, b->sc, b->q + i * query_floats(b->sc), b->t, b->ids + i * k,
                                    b->scores + i * k, b->index ? b->index + i * k : NULL,
                                    b->ctr ? b->ctr + i : NULL);
  }
  ws_free(&W);
  return NULL;
}

int oracle_search_batch(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                        int64_t nq, const int32_t t[6], int64_t* ids, float* scores,
                        int32_t* index, oracle_counters_t* ctr, int32_t* status,
                        int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  batch_t b = {ix, sc, q, nq, t, ids, scores, index, ctr, status, 0, PTHREAD_MUTEX_INITIALIZER};
  pthread_t th[256];
  int started = 0;
  for (int i = 1; i < n_threads; ++i)
    if (pthread_create(&th[started], NULL, batch_worker, &b) == 0) ++started;
  batch_worker(&b);
  for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* eval-graph variant: model.py:299-362                                      */
static int cmp_i32(const void* a, const void* b) {
  const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

int oracle_search_eval(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                       const int32_t num_scoring[3], const int32_t top_k_per_level[3],
                       int32_t topk_eval, int64_t* out_item_ids, float* out_scores,
                       int32_t* out_index, int32_t* n_out) {
  if (!scorer_ok(sc) || sc->d != ix->d || sc->emb_dtype != ix->emb_dtype) return ORACLE_ERR_BAD_ARGUMENT;
  if (num_scoring[2] != 1) return ORACLE_ERR_BAD_ARGUMENT; /* model.py:347 */
  const int64_t N = ix->n_items;
  int rc = ORACLE_OK;
  char* tmp = NULL; int64_t tmp_cap = 0;
  uint8_t* visited = (uint8_t*)calloc((size_t)N, 1);
  uint8_t* seen = (uint8_t*)calloc((size_t)N, 1);
  int64_t cap = ix->n_enter + 1;
  for (int l = 0; l < 2; ++l) if (ix->nb_nnz[l] + 1 > cap) cap = ix->nb_nnz[l] + 1;
  if (cap < N + 1) cap = N + 1;
  int32_t* res_ids = (int32_t*)malloc((size_t)cap * 4);
  float* res_sc = (float*)malloc((size_t)cap * 4);
  int32_t* cat_ids = (int32_t*)malloc((size_t)cap * 2 * 4);
  float* cat_sc = (float*)malloc((size_t)cap * 2 * 4);
  int32_t* nxt = (int32_t*)malloc((size_t)cap * 4);
  float* nxt_sc = (float*)malloc((size_t)cap * 4);
  int32_t* cand = (int32_t*)malloc((size_t)cap * 4);
  int32_t* tidx = (int32_t*)malloc((size_t)cap * 2 * 4);
  if (!visited || !seen || !res_ids || !res_sc || !cat_ids || !cat_sc || !nxt || !nxt_sc || !cand || !tidx) {
    rc = ORACLE_ERR_BAD_ARGUMENT;
    goto done;
  }
  /* start level: score all enter points, keep min(k, n) (:349-352, :268) */
  int64_t n_res = ix->n_enter;
  if (n_res <= 0) { rc = ORACLE_ERR_EMPTY_SCORE_BATCH; goto done; }
  rc = forward(ix, sc, q, ix->enter_points, n_res, cat_sc, &tmp, &tmp_cap);
  if (rc == ORACLE_ERR_TOPK_SCALAR_INPUT) rc = ORACLE_OK; /* a single candidate is fine here: k = min(k, n) */
  if (rc) goto done;
  {
    int32_t k = top_k_per_level[2] < n_res ? top_k_per_level[2] : (int32_t)n_res;
    rc = topk_ids(ix->enter_points, cat_sc, n_res, k, res_ids, res_sc, tidx);
    if (rc) goto done;
    n_res = k;
  }
  for (int level = 1; level >= 0; --level) { /* search_level, :299-337 */
    memset(visited, 0, (size_t)N);
    for (int64_t i = 0; i < n_res; ++i) visited[res_ids[i]] = 1; /* visited_idx = idx_ep */
    int64_t n_cand = n_res;
    memcpy(cand, res_ids, (size_t)n_res * 4);
    for (int it = 0; it < num_scoring[level]; ++it) {
      /* neighbours of the candidates, unique, minus visited, ascending (:316-319) */
      int64_t n_next = 0;
      for (int64_t i = 0; i < n_cand; ++i) {
        const int64_t s = ix->nb_row_splits[level][cand[i]], e = ix->nb_row_splits[level][cand[i] + 1];
        for (int64_t j = s; j < e; ++j) {
          const int32_t v = ix->nb_values[level][j];
          if (v < 0 || v >= N) { rc = ORACLE_ERR_INDEX_OUT_OF_RANGE; goto done; }
          if (!visited[v] && !seen[v]) { seen[v] = 1; nxt[n_next++] = v; }
        }
      }
      qsort(nxt, (size_t)n_next, 4, cmp_i32);
      for (int64_t i = 0; i < n_next; ++i) { seen[nxt[i]] = 0; visited[nxt[i]] = 1; } /* :321 */
      if (n_next == 0) {
        /* exhausted frontier: the eval graph scores with plain TF ops (get_scores, :240-262), which return
         * an empty tensor for an empty batch -- unlike the serving graph's BlazeXlaOp, nothing fails.  The
         * concat adds nothing, top_k keeps min(k, n) of the (sorted) result, no candidate passes the mask. */
        if (top_k_per_level[level] < n_res) n_res = top_k_per_level[level];
        n_cand = 0;
        continue;
      }
      rc = forward(ix, sc, q, nxt, n_next, nxt_sc, &tmp, &tmp_cap); /* :323 */
      if (rc == ORACLE_ERR_TOPK_SCALAR_INPUT) rc = ORACLE_OK;
      if (rc) goto done;
      /* top-k of result || next (:326-328) */
      memcpy(cat_ids, res_ids, (size_t)n_res * 4);
      memcpy(cat_sc, res_sc, (size_t)n_res * 4);
      memcpy(cat_ids + n_res, nxt, (size_t)n_next * 4);
      memcpy(cat_sc + n_res, nxt_sc, (size_t)n_next * 4);
      const int64_t n_cat = n_res + n_next;
      const int32_t k = top_k_per_level[level] < n_cat ? top_k_per_level[level] : (int32_t)n_cat;
      rc = topk_ids(cat_ids, cat_sc, n_cat, k, res_ids, res_sc, tidx);
      if (rc) goto done;
      n_res = k;
      /* next frontier: new nodes scoring at least the worst kept result (:333-334) */
      const float worst = res_sc[n_res - 1];
      n_cand = 0;
      for (int64_t i = 0; i < n_next; ++i)
        if (nxt_sc[i] >= worst) cand[n_cand++] = nxt[i];
    }
  }
  {
    const int32_t k = topk_eval < n_res ? topk_eval : (int32_t)n_res; /* results[:topk_eval] :358 */
    for (int32_t i = 0; i < k; ++i) {
      if (out_index) out_index[i] = res_ids[i];
      out_scores[i] = res_sc[i];
      out_item_ids[i] = ix->item_ids[res_ids[i]];
    }
    *n_out = k;
  }
done:
  free(tmp); free(visited); free(seen); free(res_ids); free(res_sc); free(cat_ids); free(cat_sc);
  free(nxt); free(nxt_sc); free(cand); free(tidx);
  return rc;
}

int oracle_brute_force(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                       int32_t k, int32_t* out_index, float* out_scores) {
  if (!scorer_ok(sc)) return ORACLE_ERR_BAD_ARGUMENT;
  float* s = (float*)malloc((size_t)(ix->n_items ? ix->n_items : 1) * 4);
  if (!s) return ORACLE_ERR_BAD_ARGUMENT;
  int rc = oracle_score_rows(sc, q, ix->item_embs, ix->n_items, s);
  if (!rc) rc = oracle_topk_f32(s, ix->n_items, k, out_scores, out_index);
  free(s);
  return rc;
}

int oracle_merge_topk(const float* scores, const int64_t* ids, int n_shards, int32_t k_in,
                      int32_t k_out, float* out_scores, int64_t* out_ids) {
  const int64_t n = (int64_t)n_shards * k_in;
  int32_t* idx = (int32_t*)malloc((size_t)(k_out ? k_out : 1) * 4);
  if (!idx) return ORACLE_ERR_BAD_ARGUMENT;
  const int rc = oracle_topk_f32(scores, n, k_out, out_scores, idx);
  if (!rc)
    for (int32_t i = 0; i < k_out; ++i) out_ids[i] = ids[idx[i]];
  free(idx);
  return rc;
}

/* ------------------------------------------------------------------------ */
/* the reference's scorer model (see nann_oracle.h)         */
static void dense(const float* x, int n_in, const float* w, const float* b, int n_out, float* y) {
  for (int j = 0; j < n_out; ++j) {
    float acc = b ? b[j] : 0.0f;
    for (int k = 0; k < n_in; ++k) acc = fmaf(x[k], w[(int64_t)k * n_out + j], acc);
    y[j] = acc;
  }
}

static int attn_ok(const oracle_attn_model_t* m) {
  return m && m->d > 0 && m->d <= 512 && m->E > 0 && m->E <= 256 && m->L > 0 && m->L <= 1024 &&
         m->h[0] > 0 && m->h[0] <= 1024 && m->h[1] > 0 && m->h[1] <= 1024 && m->h[2] > 0 && m->h[2] <= 1024;
}

int oracle_attn_prepare(const oracle_attn_model_t* m, const float* user_seq, float* kproj) {
  if (!attn_ok(m)) return ORACLE_ERR_BAD_ARGUMENT;
  const int E = m->E;
  float k1[512];
  for (int l = 0; l < m->L; ++l) { /* model_util.py:84-85 */
    dense(user_seq + (int64_t)l * E, E, m->wk1, m->bk1, 2 * E, k1);
    for (int j = 0; j < 2 * E; ++j) k1[j] = prelu(k1[j], m->ak[j]);
    dense(k1, 2 * E, m->wk2, m->bk2, 4 * E, kproj + (int64_t)l * 4 * E);
  }
  return ORACLE_OK;
}

int oracle_attn_score_rows(const oracle_attn_model_t* m, const float* user_seq, const float* kproj,
                           const void* rows, int64_t n, float* out_scores) {
  if (!attn_ok(m)) return ORACLE_ERR_BAD_ARGUMENT;
  if (n <= 0) return ORACLE_ERR_EMPTY_SCORE_BATCH; /* blaze_xla_predictor.cc:259-263 */
  const int d = m->d, E = m->E, L = m->L;
  float e[512], q1[512], q_[1024], att[1024], x[1024], y[1024];
  const float inv = 1.0f / sqrtf((float)(4 * E)); /* model_util.py:89-91 */
  for (int64_t i = 0; i < n; ++i) {
    const char* row = (const char*)rows + i * (int64_t)d * elem_bytes(m->emb_dtype);
    for (int k = 0; k < d; ++k) e[k] = load_elem(row, m->emb_dtype, k);
    dense(e, d, m->wq1, m->bq1, 2 * E, q1); /* :81-82 */
    for (int j = 0; j < 2 * E; ++j) q1[j] = prelu(q1[j], m->aq[j]);
    dense(q1, 2 * E, m->wq2, m->bq2, 4 * E, q_);
    float mx = -INFINITY;
    for (int l = 0; l < L; ++l) { /* einsum knd,kld->knl, / sqrt(d_k) (:90-91) */
      float acc = 0.0f;
      for (int k = 0; k < 4 * E; ++k) acc = fmaf(q_[k], kproj[(int64_t)l * 4 * E + k], acc);
      att[l] = acc * inv;
      if (att[l] > mx) mx = att[l];
    }
    float sum = 0.0f;
    for (int l = 0; l < L; ++l) { att[l] = expf(att[l] - mx); sum += att[l]; } /* softmax (:93) */
    for (int k = 0; k < E; ++k) { /* att_out summed over the sequence (:95, model.py:204-206) */
      float acc = 0.0f;
      for (int l = 0; l < L; ++l) acc = fmaf(att[l] / sum, user_seq[(int64_t)l * E + k], acc);
      x[k] = acc;
    }
    for (int k = 0; k < d; ++k) x[E + k] = e[k]; /* concat (model.py:211) */
    int n_in = E + d;
    for (int layer = 0; layer < 3; ++layer) { /* DNN + bn + prelu (model.py:213-216) */
      dense(x, n_in, m->w[layer], m->b[layer], m->h[layer], y);
      for (int j = 0; j < m->h[layer]; ++j)
        x[j] = prelu(fmaf(y[j], m->bn_scale[layer][j], m->bn_shift[layer][j]), m->alpha[layer][j]);
      n_in = m->h[layer];
    }
    float logit = 0.0f; /* last layer: no bias, no activation (:218-219) */
    for (int k = 0; k < n_in; ++k) logit = fmaf(x[k], m->w[3][k], logit);
    out_scores[i] = logit;
  }
  return ORACLE_OK;
}
