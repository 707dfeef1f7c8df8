/*
 * nann_oracle.h -- CPU restatement of the NANN HNSW-with-model-scoring retrieval
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  It is the checker, never the product: nothing under
 * nann_amd/ imports, links or calls it, and the product path fails loudly
 * when the HIP extension is missing.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - oracle_group_gather / oracle_bitmap_ref_difference: PINNED against the
 *     reference's own test inputs (group_gather_test.py:18-26,
 *     bitmap_ref_difference.py:16-29) with the outputs the reference kernels
 *     produced for them, committed in tests/golden/.
 *   - oracle_topk / oracle_gather_rows / oracle_search: restated from
 *     topk_op.cc:104-205, gather_functor.h:38-116 and
 *     build_opt_graph.py:69-149.  The reference holds no expected outputs
 *     for these (its tests are print-only) -> "parity unpinned" beyond the
 *     algorithm text; the total order (value desc, index asc) fully
 *     determines TopKV2's result.
 *   - scorers: the BASELINE scorers (L2, 3-layer MLP) behind the BlazeXlaOp
 *     contract (rows scored independently, f32 logits); the reference's own
 *     trained model has no checkpoint in the tree -> unpinned by construction.
 *
 * All file:line citations are relative to /root/reference/.
 * UO/ = tensorflow/tensorflow/core/user_ops/.
 */
#ifndef NANN_ORACLE_H_
#define NANN_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes (shared numbering with include/nann_hip.h) */
enum {
  ORACLE_OK = 0,
  ORACLE_ERR_INVALID_RAGGED_PARAMS = 1,  /* GroupGather_kernel.cc:62-64  */
  ORACLE_ERR_INVALID_RAGGED_INDICES = 2, /* GroupGather_kernel.cc:65-67  */
  ORACLE_ERR_INVALID_RAGGED_INPUT = 3,   /* bitmap_ops.cc:182-184        */
  ORACLE_ERR_TOPK_K_GT_N = 4,            /* topk_op.cc:67-71             */
  ORACLE_ERR_INDEX_OUT_OF_RANGE = 5,     /* gather_op.cc:173 (and the bounds
                                            checks the reference omits)  */
  ORACLE_ERR_EMPTY_SCORE_BATCH = 6,      /* blaze_xla_predictor.cc:259-263 */
  ORACLE_ERR_BAD_ARGUMENT = 7,
  ORACLE_ERR_TOPK_SCALAR_INPUT = 8       /* topk_op.cc:63-65 after Squeeze of
                                            a single candidate            */
};

enum { ORACLE_EMB_F16 = 0, ORACLE_EMB_BF16 = 1, ORACLE_EMB_F32 = 2 };
enum { ORACLE_SCORER_L2 = 0, ORACLE_SCORER_MLP = 1,
       ORACLE_SCORER_ATTN = 2 /* the reference's own model (oracle_attn_model_t); the "query" of a search is then the
                                  user sequence f32[L, E] instead of a vector f32[d] */ };

/* ragged validation, GroupGather_kernel.cc:9-16 / bitmap_ops.cc:12-19:
 * returns 0 or the reference's code 1/2/3. */
int oracle_validate_ragged(int64_t n_values, const int64_t* row_splits,
                           int64_t n_splits);

/* GroupGather<int32>, unique=false path (GroupGather_kernel.cc:55-173).
 * out_values may be NULL to query *n_out only.  out_row_splits has
 * n_indices_splits entries (or 1 entry on the void-input path).
 * *n_out_splits receives the number of row_splits written. */
int oracle_group_gather_i32(const int32_t* params_values, int64_t n_params_values,
                            const int64_t* params_row_splits, int64_t n_params_splits,
                            const int64_t* indices_values, int64_t n_indices_values,
                            const int64_t* indices_row_splits, int64_t n_indices_splits,
                            int32_t* out_values, int64_t out_cap,
                            int64_t* out_row_splits, int64_t* n_out,
                            int64_t* n_out_splits, int* ragged_code);

/* GroupGather<int32>, unique=true path (GroupGather_kernel.cc:91-131): per group the set of the gathered values,
 * emitted in first-occurrence order (the reference's order is the unordered_set's: any order is its answer).
 * Same argument meaning as oracle_group_gather_i32. */
int oracle_group_gather_unique_i32(const int32_t* params_values, int64_t n_params_values,
                                   const int64_t* params_row_splits, int64_t n_params_splits,
                                   const int64_t* indices_values, int64_t n_indices_values,
                                   const int64_t* indices_row_splits, int64_t n_indices_splits,
                                   int32_t* out_values, int64_t out_cap,
                                   int64_t* out_row_splits, int64_t* n_out,
                                   int64_t* n_out_splits, int* ragged_code);

/* BitmapRefDifference<int32> (bitmap_ops.cc:175-257).  bitmap is mutated in
 * place (Ref semantics).  n_bitmap_words is used for the bounds check the
 * reference omits (Appendix C) -> ORACLE_ERR_INDEX_OUT_OF_RANGE. */
int oracle_bitmap_ref_difference_i32(const int32_t* values, int64_t n_values,
                                     const int64_t* row_splits, int64_t n_splits,
                                     int32_t* bitmap, int64_t n_bitmap_words,
                                     int32_t* out_values, int64_t* out_row_splits,
                                     int64_t* n_out, int64_t* n_out_splits,
                                     int* ragged_code);

/* ---- sibling ops of SURVEY.md 8(a8): registered by the reference but not wired into the
 * serving graph.  Covered: BitmapInit, BitmapDifference, BatchTopKOnRT.  Not covered:
 * BloomFilterDifference (needs FarmHash Fingerprint64, absent from the tree), BlazeTopK
 * (unstable ties and an out-of-bounds loop). */

/* BitmapInit<int32> (bitmap_ops.cc:28-75): length-word bitmap with the bits of idx set.
 * Requires 0 <= length and n_idx <= length (:56-57, the reference's own odd check). */
int oracle_bitmap_init_i32(const int32_t* idx, int64_t n_idx, int32_t length, int32_t* bitmap);

/* BitmapDifference<int32> (bitmap_ops.cc:83-143): value-semantics variant -- copies the
 * bitmap, then the same first-occurrence filter over ONE flat list. */
int oracle_bitmap_difference_i32(const int32_t* idx_next, int64_t n, const int32_t* idx_flag,
                                 int64_t n_words, int32_t* idx_next_new, int64_t* n_out,
                                 int32_t* idx_flag_new);

/* BatchTopKOnRT<float> (UO/topk_op/BatchTopKOnRT_kernel.cc:62-155): per ragged row the
 * min(k[i], len) best values (descending, or ascending), row-local indices.  The reference
 * uses std::partial_sort_copy, whose order among EQUAL values is unspecified; this
 * restatement breaks ties by lower position (one of the valid outcomes).
 * k_is_scalar: k[0] applies to every row (:99-101). */
/* tensorflow::Fingerprint64 (core/platform/fingerprint.h) = farmhash::Fingerprint64, an un-vendored
 * dependency of the fork (FarmHash, farmhashna::Hash64): restated from the published algorithm for
 * strings of up to 32 bytes (decimal ids never exceed 20).  Pinned by the fork's own known answers,
 * core/platform/fingerprint_test.cc:27-28 ("Hello", "World": the 4..7-byte branch); the other length
 * branches follow the published text only -> parity unpinned there.  Longer input returns 0. */
uint64_t oracle_fingerprint64(const char* s, int64_t len);

/* BloomFilterDifference<int32> (UO/bitmap_op/bitmap_ops.cc:264-425): per node, in input order, four
 * positions of a 32 * bucket_size-bit filter derived from Fingerprint64 of its DECIMAL STRING (:346-357,
 * primes from init_prime_array :405-411); the node is kept iff at least one of the four bits was clear,
 * and all four are set.  Approximate by design (false "visited"), exact in its arithmetic.  One filter
 * for all groups; void input -> ([], [0]).  n_flag_words >= bucket_size is the caller's contract (the
 * reference indexes without a check); here a shorter array is ORACLE_ERR_BAD_ARGUMENT. */
int oracle_bloom_filter_difference_i32(const int32_t* values, int64_t n_values, const int64_t* row_splits,
                                       int64_t n_splits, int32_t* idx_flag, int64_t n_flag_words,
                                       int64_t bucket, int64_t bucket_size, int32_t* out_values,
                                       int64_t* out_rs, int64_t* n_out, int64_t* n_out_splits,
                                       int* ragged_code);
/* the four filter positions of one node (exposed for the device kernel's parity test) */
void oracle_bloom_positions(int32_t node, int64_t bucket, int64_t bucket_size, int64_t pos[4]);

int oracle_batch_topk_on_rt_f32(const float* values, int64_t n_values, const int64_t* row_splits,
                                int64_t n_splits, const int64_t* k, int k_is_scalar, int ascending,
                                float* values_out, int64_t* idx_out, int64_t* row_splits_out,
                                int64_t* n_out, int64_t* n_out_splits, int* ragged_code);

/* GatherV2 axis 0 (gather_functor.h:38-116): out[i,:] = params[idx[i],:].
 * row_bytes = slice bytes.  bad index -> ORACLE_ERR_INDEX_OUT_OF_RANGE and
 * *bad_i = first bad position (gather_op.cc:170-175). */
int oracle_gather_rows(const void* params, int64_t n_rows, int64_t row_bytes,
                       const int32_t* idx, int64_t n_idx, void* out,
                       int64_t* bad_i);

/* TopKV2 sorted=true (topk_op.cc:104-205): values descending, ties broken by
 * lower input index.  n < k -> ORACLE_ERR_TOPK_K_GT_N. */
int oracle_topk_f32(const float* values, int64_t n, int32_t k,
                    float* out_values, int32_t* out_indices);

/* exact conversions used by every scorer */
float oracle_half_to_float(uint16_t h);
float oracle_bf16_to_float(uint16_t h);
uint16_t oracle_float_to_half(float f); /* round-to-nearest-even */

/* user sequence -> query vector (mean of non-pad rows, f32).
 * seq: f16[seq_len, d]; a row is "pad" iff all its elements are zero.
 * q[k] = (sum_r seq[r][k], r ascending, plain f32 adds) / count. */
void oracle_user_seq_mean(const uint16_t* seq_f16, int seq_len, int d, float* q);

typedef struct {
  int kind;          /* ORACLE_SCORER_* */
  int d;             /* embedding dim */
  int emb_dtype;     /* ORACLE_EMB_* of item rows */
  /* MLP only: x=[q;e] in R^{2d} -> H1 -> PReLU -> H2 -> PReLU -> 1 (no bias)
   * (model.py:218-219 last layer bias-free; model_util.py:9-11
   * PReLU = max(0,x)+alpha*min(0,x), per-channel alpha) */
  int h1, h2;
  const float* w1;     /* [2d, h1] row-major */
  const float* b1;     /* [h1] */
  const float* alpha1; /* [h1] */
  const float* w2;     /* [h1, h2] row-major */
  const float* b2;     /* [h2] */
  const float* alpha2; /* [h2] */
  const float* w3;     /* [h2] */
  const void* attn;    /* ORACLE_SCORER_ATTN only: const oracle_attn_model_t* */
} oracle_scorer_t;

/* Score n item rows (already gathered, contiguous [n, d] in emb_dtype)
 * against one query vector q f32[d]: the BlazeXlaOp contract
 * (blaze_xla_predictor.cc:360-459: pad -> rows scored independently ->
 * slice; f32 logits).  n == 0 -> ORACLE_ERR_EMPTY_SCORE_BATCH.
 * Canonical summation orders are documented in nann_oracle.c and DESIGN.md;
 * the HIP kernels use the same orders so L2 scores are bit-identical. */
int oracle_score_rows(const oracle_scorer_t* sc, const float* q,
                      const void* rows, int64_t n, float* out_scores);

typedef struct {
  int64_t n_items;
  int d;
  int emb_dtype;
  const void* item_embs;      /* [n_items, d] */
  const int64_t* item_ids;    /* [n_items] */
  /* level 0 and level 1 adjacency in the reference's on-disk layout
   * (build_hnsw_index.py:49-66): values int32, row_splits int64[n_items+1] */
  const int32_t* nb_values[2];
  const int64_t* nb_row_splits[2];
  int64_t nb_nnz[2];
  const int32_t* enter_points; /* [n_enter] ascending internal indices */
  int64_t n_enter;
} oracle_index_t;

#define ORACLE_NUM_ROUNDS 5 /* entry, level-1, level-0 x3 */
typedef struct {
  int64_t frontier[ORACLE_NUM_ROUNDS]; /* F_r: rows walked          */
  int64_t gathered[ORACLE_NUM_ROUNDS]; /* G_r: neighbours gathered  */
  int64_t scored[ORACLE_NUM_ROUNDS];   /* S_r: unique rows scored   */
} oracle_counters_t;

/* The traversal schedule of build_opt_graph.py:109-149
 * for one query.  out_* have level_topn[5] entries.  out_index = internal
 * indices before the item_ids gather (not a reference output; for tests). */
int oracle_search(const oracle_index_t* ix, const oracle_scorer_t* sc,
                  const float* q, const int32_t level_topn[6],
                  int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                  oracle_counters_t* ctr);

/* SURVEY.md 8(f3), oracle side only so far: the EVAL-graph search `Model.retrieval` /
 * `search_level` (NANN_impls/nann/model/model.py:299-362), a different schedule from
 * build_model(): per level `num_scoring[level]` rounds; candidates of a round =
 * sorted(unique(neighbours) - visited) (tf.unique + tf.sets.set_difference: ascending id
 * order); the result set is re-top-k'ed every round to `top_k_per_level[level]` with
 * k = min(k, n) (:268); next frontier = new nodes with score >= the worst kept score
 * (:333-334); one visited set per level seeded with the level's entry set; results
 * truncated to topk_eval (:358).  Defaults of the reference: start level 2,
 * num_scoring = [3,1,1], top_k_per_level = [400,200,100], topk_eval = 200 (config.py:50-58).
 * Arrays are indexed by level (0,1,2); num_scoring[2] must be 1 (:347).
 * out_* have topk_eval entries; *n_out = entries actually produced. */
int oracle_search_eval(const oracle_index_t* ix, const oracle_scorer_t* sc, const float* q,
                       const int32_t num_scoring[3], const int32_t top_k_per_level[3],
                       int32_t topk_eval, int64_t* out_item_ids, float* out_scores,
                       int32_t* out_index, int32_t* n_out);

/* n_queries independent searches, one query per thread (mirrors the
 * reference's sessions x threads concurrency, gen_benchmark_conf.py:22-30).
 * status[i] per query.  Returns ORACLE_OK if the batch ran. */
int oracle_search_batch(const oracle_index_t* ix, const oracle_scorer_t* sc,
                        const float* q /*[n_queries,d]*/, int64_t n_queries,
                        const int32_t level_topn[6], int64_t* out_item_ids,
                        float* out_scores, int32_t* out_index,
                        oracle_counters_t* ctr /*[n_queries] or NULL*/,
                        int32_t* status, int n_threads);

/* brute force: score all items, TopKV2 top-k (recall ground truth; mirrors
 * test_all, main.py:194-237). */
int oracle_brute_force(const oracle_index_t* ix, const oracle_scorer_t* sc,
                       const float* q, int32_t k, int32_t* out_index,
                       float* out_scores);

/* merge per-shard top-k lists: concat in shard order then TopKV2 semantics
 * (SURVEY.md 8e).  scores/ids: [n_shards, k_in]. */
int oracle_merge_topk(const float* scores, const int64_t* ids, int n_shards,
                      int32_t k_in, int32_t k_out, float* out_scores,
                      int64_t* out_ids);

/* SURVEY.md 8(f2), oracle side only so far -- PARITY UNPINNED: TensorFlow is not in the image,
 * so this restatement of the reference's scorer model has not been checked against a run of the
 * frozen graph; it is cross-checked against an independent float64 numpy restatement
 * (tests/test_attn_model_cpu.py).
 *
 * The model behind BlazeXlaOp in the reference (NANN_impls/nann/model/model.py:189-233,
 * model_util.py:9-11,32-67,70-97): for one user sequence u [L, E] and a candidate row e [d]
 *   q1 = prelu(e Wq1 + bq1; aq)      [2E]      k1_l = prelu(u_l Wk1 + bk1; ak)  [2E]
 *   q_ = q1 Wq2 + bq2                [4E]      k_l  = k1_l Wk2 + bk2            [4E]
 *   att_l = <q_, k_l> / sqrt(4E);  p = softmax_l(att);  a = sum_l p_l u_l        [E]
 *   x = [a ; e];  three times x = prelu(bn(x W + b)); logit = x W4 (no bias, model.py:218-219)
 * Inference-mode batch norm is folded to y = x * bn_scale + bn_shift
 * (scale = gamma / sqrt(moving_var + 1e-3), shift = beta - moving_mean * scale).
 * No attention mask: zero-padded sequence positions take part in the softmax, as in
 * nonlinear_attention().  All dense layers: acc = bias; acc = fmaf(x[k], W[k][j], acc), k ascending.
 *
 * The conventions this restatement ASSUMES and nothing the reference holds pins (no TensorFlow-written frozen graph of
 * Model.forward, no checkpoint under NANN_impls): the day such a file appears, the diff against it is this table.
 *   convention                     assumed here                                          where the reference states it
 *   -----------------------------  ----------------------------------------------------  -------------------------------------
 *   batch-norm epsilon             1e-3 (tf.layers.batch_normalization's default)        model_util.py:53 passes none
 *   batch-norm at inference        moving_mean / moving_variance, folded into scale /    model_util.py:53 (training=training);
 *                                  shift per channel: y = x * scale + shift              convert_meta.py:361-398 freezes them
 *   batch-norm position            dense -> bn -> prelu, bias inside the dense layer     model_util.py:42-67
 *   prelu                          max(x, 0) + alpha * min(x, 0), alpha per channel,     model_util.py:9-11
 *                                  initial value as trained (no constraint on its sign)
 *   Tensordot / dense layout       weights [in, out] row-major, y_j = sum_k x_k W[k][j]  model_util.py:44-49,81-85 (tf.layers.dense
 *                                  (kernel of tf.layers.dense; no transpose)             = Tensordot on the last axis in the graph)
 *   attention scale                1 / sqrt(4E) on the logits, softmax over all L        model_util.py:87-93
 *                                  positions, padded ones included (no mask)
 *   attention pooling              weights applied to the RAW sequence u_l, not to k_l   model_util.py:95
 *   concat order of the DNN input  [attended user vector ; item row]                     model.py:213
 *   output layer                   no bias, no activation: the logit                     model.py:218-219
 *   item row dtype                 the index's storage type widened to f32 exactly       build_hnsw_index.py:41-66 (f32 on disk)
 *   accumulation order             k ascending, one fmaf per term (XLA's CPU dot may     blaze_xla_predictor.cc:360-459 leaves it
 *                                  tile differently: tolerance 1e-5, north_star)         to the compiled graph
 */
typedef struct {
  int d, E, L, emb_dtype;
  const float *wq1, *bq1, *aq; /* [d,2E] [2E] [2E] */
  const float *wq2, *bq2;      /* [2E,4E] [4E] */
  const float *wk1, *bk1, *ak; /* [E,2E] [2E] [2E] */
  const float *wk2, *bk2;      /* [2E,4E] [4E] */
  int h[3];                    /* DNN widths, 128-64-32 in the reference */
  const float* w[4];           /* [E+d,h0] [h0,h1] [h1,h2] [h2] */
  const float* b[3];
  const float* bn_scale[3];
  const float* bn_shift[3];
  const float* alpha[3];
} oracle_attn_model_t;

/* per-user part: kproj[l] = k_l, f32 [L, 4E] */
int oracle_attn_prepare(const oracle_attn_model_t* m, const float* user_seq, float* kproj);
/* logits of n candidate rows ([n, d] in m->emb_dtype) for that user */
int oracle_attn_score_rows(const oracle_attn_model_t* m, const float* user_seq, const float* kproj,
                           const void* rows, int64_t n, float* out_scores);

#ifdef __cplusplus
}
#endif
#endif /* NANN_ORACLE_H_ */
