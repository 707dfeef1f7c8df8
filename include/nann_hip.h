/*
 * nann_hip.h -- C ABI of libnann_hip.so: the MI355X (gfx950) implementation of
 * NANN's HNSW-with-model-scoring retrieval hot path.
 *
 * Every entry point replaces one piece of the reference's TensorFlow custom-op
 * path (citations relative to /root/reference/,
 * UO/ = tensorflow/tensorflow/core/user_ops/).  extern "C", POD arguments,
 * int status returns, no exceptions/STL/torch types across the boundary.
 * The op-kernel shims in nann_amd/tf_ops/ (REGISTER_OP / REGISTER_KERNEL_BUILDER
 * with the reference's op names) and the Python mirror in nann_amd/ops.py are
 * thin callers of this file.  INTEGRATION.md shows the binding a reference
 * maintainer would add.
 *
 * Memory convention: unless a parameter is marked [host], data pointers are
 * DEVICE pointers (HBM) valid on the library's current HIP device, and the
 * call is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 * default stream).  Calls that return a data-dependent count synchronise the
 * stream before returning (exactly where the reference must know the size to
 * allocate_output: GroupGather_kernel.cc:147-148, bitmap_ops.cc:245).
 * All entry points are thread-safe and re-entrant on shared immutable handles.
 */
#ifndef NANN_HIP_H_
#define NANN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NANN_ABI_VERSION 6

/* status codes; 1..8 share the oracle's numbering (oracle/nann_oracle.h) and
 * map to the TF errors the reference raises at the cited lines */
enum nann_status {
  NANN_OK = 0,
  NANN_ERR_INVALID_RAGGED_PARAMS = 1,  /* InvalidArgument, GroupGather_kernel.cc:62-64 */
  NANN_ERR_INVALID_RAGGED_INDICES = 2, /* InvalidArgument, GroupGather_kernel.cc:65-67 */
  NANN_ERR_INVALID_RAGGED_INPUT = 3,   /* InvalidArgument, bitmap_ops.cc:182-184 */
  NANN_ERR_TOPK_K_GT_N = 4,            /* InvalidArgument, topk_op.cc:67-71 */
  NANN_ERR_INDEX_OUT_OF_RANGE = 5,     /* InvalidArgument, gather_op.cc:170-175; also the
                                          bounds checks the reference omits (UB there) */
  NANN_ERR_EMPTY_SCORE_BATCH = 6,      /* blaze_xla_predictor.cc:259-263 */
  NANN_ERR_BAD_ARGUMENT = 7,
  NANN_ERR_TOPK_SCALAR_INPUT = 8,      /* topk_op.cc:63-65 after Squeeze of one candidate */
  NANN_ERR_HIP = 100,                  /* a HIP runtime call failed (nann_last_error) */
  NANN_ERR_NO_DEVICE = 101,
  NANN_ERR_UNSUPPORTED = 102,
  NANN_ERR_CAPACITY = 103,             /* caller-provided output too small; count returned */
  NANN_ERR_IO = 104,                   /* HugeConst: file/npy header problems */
  NANN_ERR_DTYPE_MISMATCH = 105,       /* HugeConst: huge_const_op.cc:117-147 */
  NANN_ERR_SHAPE_MISMATCH = 106        /* HugeConst: huge_const_op.cc:111-115 */
};

enum nann_dtype { NANN_F16 = 0, NANN_BF16 = 1, NANN_F32 = 2, NANN_I32 = 3, NANN_I64 = 4,
                  NANN_F64 = 5 };
enum nann_scorer_kind { NANN_SCORER_L2 = 0, NANN_SCORER_MLP = 1 };

typedef void* nann_stream_t; /* hipStream_t */

int nann_abi_version(void);
/* [host] NUL-terminated description of the calling thread's last failure */
const char* nann_last_error(void);
/* number of visible HIP devices (0 when there is no GPU; never fails) */
int nann_device_count(void);

/* ---- device memory plumbing for hosts without torch (the TF shims) -------- */
int nann_malloc(void** dev_ptr, int64_t nbytes);
int nann_free(void* dev_ptr);
/* kind: 0 host->device, 1 device->host, 2 device->device; async on stream */
int nann_memcpy(void* dst, const void* src, int64_t nbytes, int kind, nann_stream_t stream);
int nann_stream_synchronize(nann_stream_t stream);
/* a stream of the host's own (non-blocking with respect to the null stream) and page-locked staging memory: what a
 * multi-lane C++ host needs to overlap one batch's copies with another's search (csrc/host/nann_serve.cpp) */
int nann_stream_create(nann_stream_t* out);
int nann_stream_destroy(nann_stream_t stream);
int nann_host_malloc(void** host_ptr, int64_t nbytes);
int nann_host_free(void* host_ptr);

/* ---- a6: HugeConst (UO/huge_const_op/huge_const_op.cc:58-226) --------------
 * Loads a .npy (format 1.0/2.0, C order) into HBM once; the GPU kernel of the
 * reference op does the same one-time H2D copy (:187-218).  expect_dtype /
 * expect_shape mirror the op's dtype/shape attrs and are validated against
 * the header (:108-147); expect_shape may be NULL to accept the file's shape.
 * allow_cast != 0 reproduces the Python wrapper huge_constant()
 * (NANN_impls/nann/model/model_util.py:107-121), which casts the array to the
 * requested dtype (f32->f16, i64->i32, ...) -- here at load time, without
 * rewriting the file as the wrapper does.  *nbytes = bytes resident in HBM.
 * path [host]; expect_shape [host]. */
int nann_huge_const_load(const char* path, int expect_dtype, const int64_t* expect_shape,
                         int expect_rank, int allow_cast, void** dev_ptr, int64_t* nbytes);

/* ---- a1: GroupGather<int32>, unique=false (UO/beam_search_op/
 *          GroupGather_kernel.cc:55-173) -------------------------------------
 * Step 1 (count): validates both ragged inputs (codes 1/2/3 in *ragged_code
 * [host]), writes ret_row_splits (device, n_indices_splits entries, or [0] on
 * the void-input path), returns the number of values in *n_ret [host] and the
 * number of row_splits written in *n_ret_splits [host].
 * Step 2 (fill): writes ret_values (device, n_ret entries).  scratch_offsets
 * (device, int64[n_indices_values + 1]) carries the per-row output offsets
 * from step 1 to step 2. */
int nann_group_gather_count(const int64_t* params_row_splits, int64_t n_params_splits,
                            int64_t n_params_values, const int64_t* indices_values,
                            int64_t n_indices_values, const int64_t* indices_row_splits,
                            int64_t n_indices_splits, int64_t* ret_row_splits,
                            int64_t* scratch_offsets, int64_t* n_ret, int64_t* n_ret_splits,
                            int32_t* ragged_code, nann_stream_t stream);
int nann_group_gather_fill(const int32_t* params_values, const int64_t* params_row_splits,
                           const int64_t* indices_values, int64_t n_indices_values,
                           const int64_t* scratch_offsets, int32_t* ret_values,
                           nann_stream_t stream);

/* GroupGather unique=true (UO/beam_search_op/GroupGather_kernel.cc:91-131): per group the SET of the gathered
 * values -- the reference inserts the rows into a std::unordered_set per group and writes it in the set's iteration
 * order, i.e. any order of a group's distinct values is its answer; this library emits first-occurrence order.
 * Input = the unique=false result (values int32[n_values], row_splits int64[n_splits], both device: count + fill
 * above, whose validation and void-input path are the shared head of Compute).  out_values (device) must hold
 * n_values entries, out_row_splits n_splits; scratch: nann_group_gather_unique_scratch_bytes().  *n_out [host]. */
int nann_group_gather_unique_scratch_bytes(int64_t n_values, int64_t n_splits, int64_t* nbytes);
int nann_group_gather_unique(const int32_t* values, int64_t n_values, const int64_t* row_splits, int64_t n_splits,
                             void* scratch, int32_t* out_values, int64_t* out_row_splits, int64_t* n_out,
                             nann_stream_t stream);

/* ---- a2: BitmapRefDifference<int32> (UO/bitmap_op/bitmap_ops.cc:175-257) ---
 * Serial-scan semantics: first occurrence of every id whose bit is clear is
 * kept, in input order, and its bit is set; ONE bitmap shared by all groups.
 * idx_flag (device, int32[n_flag_words]) is mutated in place (Ref input).
 * c_values (device) must hold n_values entries; c_row_splits n_splits entries.
 * *n_out / *n_out_splits [host].  ids outside [0, 32*n_flag_words) ->
 * NANN_ERR_INDEX_OUT_OF_RANGE with the bitmap untouched (UB in the reference). */
int nann_bitmap_ref_difference(const int32_t* idx_next_values, int64_t n_values,
                               const int64_t* idx_next_row_splits, int64_t n_splits,
                               int32_t* idx_flag, int64_t n_flag_words, int32_t* c_values,
                               int64_t* c_row_splits, int64_t* n_out, int64_t* n_out_splits,
                               int32_t* ragged_code, nann_stream_t stream);

/* ---- a8: BloomFilterDifference<int32> (UO/bitmap_op/bitmap_ops.cc:264-425) ----------------
 * The approximate sibling of BitmapRefDifference (registered by the reference, not wired into the
 * serving graph): per node, in input order, four positions of a 32 * bucket_size-bit filter from
 * tensorflow::Fingerprint64 (FarmHash) of the node's decimal string; kept iff one of them was clear;
 * all four set.  idx_flag (device, int32[n_flag_words >= bucket_size]) mutated in place.  Buffers,
 * counts and errors as nann_bitmap_ref_difference. */
int nann_bloom_filter_difference(const int32_t* idx_next_values, int64_t n_values,
                                 const int64_t* idx_next_row_splits, int64_t n_splits, int32_t* idx_flag,
                                 int64_t n_flag_words, int64_t bucket, int64_t bucket_size, int32_t* c_values,
                                 int64_t* c_row_splits, int64_t* n_out, int64_t* n_out_splits,
                                 int32_t* ragged_code, nann_stream_t stream);

/* ---- a3: GatherV2 axis 0 (core/kernels/gather_functor.h:38-116) ------------
 * out[i,:] = params[indices[i],:]; row_bytes must be a multiple of 4.
 * A bad index returns NANN_ERR_INDEX_OUT_OF_RANGE and its position in *bad_i
 * [host] (gather_op.cc:170-175). */
int nann_gather_rows(const void* params, int64_t n_rows, int64_t row_bytes,
                     const int32_t* indices, int64_t n_indices, void* out, int64_t* bad_i,
                     nann_stream_t stream);

/* ---- a5: TopKV2 sorted=true (core/kernels/topk_op.cc:40-205) ---------------
 * values f32[n_rows, n_cols] -> out_values f32[n_rows,k], out_indices
 * i32[n_rows,k]; descending, ties -> lower index.  n_cols < k ->
 * NANN_ERR_TOPK_K_GT_N (checked on the host, nothing launched).  k <= 1024: radix select;
 * larger k (the reference's tests go to k = n = 5000): whole-row sort in LDS, n_cols <= 16384. */
int nann_topk(const float* values, int64_t n_rows, int64_t n_cols, int32_t k,
              float* out_values, int32_t* out_indices, nann_stream_t stream);

/* ---- a4: the scorer behind the BlazeXlaOp contract (UO/blaze_op/
 *          blaze_xla_kernel.cc:24-33, blaze_xla_predictor.cc:360-459) --------
 * A scorer holds what the frozen scoring GraphDef holds in the reference.
 * L2: s = -||q - x||^2.  MLP: x=[q;e] -> h1 -> PReLU -> h2 -> PReLU -> 1
 * (weights f32; [host] pointers, copied to HBM at creation). */
typedef struct nann_scorer nann_scorer;
typedef struct nann_index nann_index;
typedef struct {
  int32_t kind;      /* nann_scorer_kind */
  int32_t d;         /* embedding dim: 64, 128, 256 or 512 */
  int32_t emb_dtype; /* NANN_F16 / NANN_BF16 / NANN_F32 of item rows */
  int32_t h1, h2;    /* MLP only; multiples of 32 */
  const float* w1;     /* [2d, h1] row-major */
  const float* b1;     /* [h1] */
  const float* alpha1; /* [h1] PReLU slope (model_util.py:9-11) */
  const float* w2;     /* [h1, h2] */
  const float* b2;     /* [h2] */
  const float* alpha2; /* [h2] */
  const float* w3;     /* [h2]; last layer has no bias (model.py:218-219) */
  int32_t precision;   /* MLP only: nann_mlp_precision */
} nann_scorer_desc;
/* How the MLP's contractions run on the matrix cores.
 *   SPLIT_F16  every f32 operand as two f16 values (22 significant bits), products on
 *              v_mfma_f32_32x32x16_f16 with f32 accumulation; scores within 1e-5 relative of the
 *              fp32 chain (north_star's bound; ~3e-7 measured), ids equal up to near-ties.
 *              Preconditions: 16-bit item rows; |w| <= 511 for the item half of W1 and for W2
 *              (checked at creation: NANN_ERR_UNSUPPORTED); hidden activations |h1| <= 511
 *              (f16 range after the x2^7 operand scale).  A larger activation SATURATES (the
 *              hi plane is cut with round-toward-zero, which never produces inf): the score is
 *              finite and wrong, so a model with such activations must use EXACT_F32.  The
 *              reference's models are batch-normalised / PReLU nets with O(1) activations.
 *   EXACT_F32  v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, scores BIT-identical to the oracle's
 *              fp32 chain (and therefore identical top-k ids); 1/16 of the 16-bit MFMA rate.
 *   In nann_search BOTH forms run with the ITEM HALF OF LAYER 1 PRE-PROJECTED: P_i = W1e^T e_i is the same vector
 *   whoever scores item i (the canonical order of layer 1 is a1 = u + P with P a chain of its own, oracle/
 *   nann_oracle.c), so it is computed for every item once per (scorer, index) pair -- a resident f32 [n_items, 256]
 *   table owned by the scorer, 1 GB per million items, ~10 ms per million to build: at the pair's first search, or
 *   ahead of traffic by nann_scorer_prepare (below) -- and the traversal gathers that row instead of running layer 1,
 *   with all of layer 2 resident in LDS (csrc/nann_mlp5.h).  Without room for the table (or with pre-projection
 *   switched off) the kernels that read the embedding rows run instead: same results for EXACT_F32, same tolerance for
 *   SPLIT_F16.  nann_score (stand-alone rows, no index) runs all three layers on the matrix cores.
 *   DEFAULT    (0, what a zero-initialised descriptor asks for) SPLIT_F16 when the weights meet its
 *              precondition, else EXACT_F32: 1e-5 is the contract, bit-exactness is opt-in. */
enum nann_mlp_precision { NANN_MLP_PRECISION_DEFAULT = 0, NANN_MLP_SPLIT_F16 = 1, NANN_MLP_EXACT_F32 = 2 };
int nann_scorer_create(const nann_scorer_desc* desc /*[host]*/, nann_scorer** out);
void nann_scorer_destroy(nann_scorer* s);

/* Lifecycle of the pre-projected tables (MLP scorers; attention models, both precisions since round 4: nann_model_* below).
 *   nann_scorer_prepare      builds the table of (scorer, index) on `stream` NOW (or finds it), waits for it, and PINS
 *                            it: a pinned table is never evicted.  A serving host calls this at start-up, so that no
 *                            request pays the build (a hipMalloc + ~10-20 ms per million items + one stream wait).
 *                            NANN_ERR_CAPACITY when HBM has no room for it (searches of the pair still work: they read
 *                            the embedding rows).  Counts: n prepare calls need n releases.
 *   nann_scorer_release      drops one pin; at zero the table is retired at once.
 *   nann_scorer_table_bytes  *table_bytes = HBM bytes the table of this pair takes (ix may be NULL: 0), *resident_bytes
 *                            = bytes the scorer holds right now, all indices (either may be NULL).  This memory is NOT
 *                            part of nann_search_workspace_bytes.
 * Without prepare, the first nann_search of a pair builds the table inside the call (one stream wait) and the scorer
 * keeps the unpinned tables of the two indices it searched last (least recently used goes).  Eviction, release and
 * nann_index_destroy only RETIRE a table: it is freed by a later call once every launch that reads it has completed
 * (an event per stream behind each search), so no call synchronises the device and a concurrent search on another
 * thread never loses its table.  Thread-safe.
 * nann_search_options.preprojection = 0 runs a call without tables; nann_set_preprojection(0) makes that the process
 * default (NANN_PREPROJECT=0 in the environment: the same). */
int nann_scorer_prepare(const nann_scorer* scorer, const nann_index* ix, nann_stream_t stream);
int nann_scorer_release(const nann_scorer* scorer, const nann_index* ix);
int nann_scorer_table_bytes(const nann_scorer* scorer, const nann_index* ix, int64_t* table_bytes,
                            int64_t* resident_bytes);
int nann_set_preprojection(int32_t enabled);

/* comm_seq f16[n_queries, seq_len, d] -> q f32[n_queries, d]: mean over
 * non-pad (not all-zero) rows; the user side of forward()
 * (build_opt_graph.py:76-79, 91-107). */
int nann_user_seq_mean(const void* comm_seq_f16, int64_t n_queries, int32_t seq_len, int32_t d,
                       float* q, nann_stream_t stream);

/* Score n rows against ONE query vector q f32[d] (rows are scored
 * independently, f32 logits -- the contract PadToStatic/SliceToDynamic rely
 * on).  indices == NULL: rows = item_emb[n, d] as BlazeXlaOp receives them
 * (already gathered).  indices != NULL: fused GatherV2 + score over
 * table[n_table_rows, d] (out-of-range -> NANN_ERR_INDEX_OUT_OF_RANGE,
 * *bad_i [host]).  n == 0 -> NANN_ERR_EMPTY_SCORE_BATCH. */
int nann_score(const nann_scorer* scorer, const float* q, const void* table,
               int64_t n_table_rows, const int32_t* indices, int64_t n, float* out_scores,
               int64_t* bad_i, nann_stream_t stream);

/* ---- a4 (b): the scoring model a BlazeXlaOp node names ----------------------------------
 * The reference's op takes the path of a frozen TensorFlow GraphDef in its `graph_def` attr and runs it in a
 * nested session (blaze_xla_kernel.cc:156-180, blaze_xla_predictor.cc:360-459).  nann_model_load takes the
 * same attr value:
 *   a FILE       the frozen GraphDef itself as convert_meta.py:361-398 writes it (`frozen_graph.pb`:
 *                Model.forward(training=False), model.py:189-233, frozen + fold_constants'ed or merely frozen),
 *                text or binary: read as text first, then as binary, the reference's order (blaze_xla_kernel.cc:
 *                169-175; csrc/host/nann_graphdef_text.h, nann_graphdef.h).
 *                No TensorFlow here and nothing of the graph is executed: the weights are pulled out of the
 *                Const nodes by the names the reference's Python gives their consumers (csrc/host/
 *                nann_graphdef.h), batch norm folded to scale / shift, and handed to the hand-written kernels
 *                (nann_attn_desc).  A graph that is not that model -> NANN_ERR_UNSUPPORTED naming what is
 *                missing; a file that parses as neither -> NANN_ERR_IO.  An optional `<file>.precision` beside it
 *                holds "split" | "exact".
 *   a DIRECTORY  of .npy weight files, for scorers that have no frozen graph in the reference (BASELINE's L2
 *                and MLP) and for hosts that hold the model as arrays:
 *                  scorer.txt   one word: l2 | mlp | attention
 *                  mlp          w1 [2d,256]  b1  alpha1  w2 [256,128]  b2  alpha2  w3 [128]     (nann_scorer_desc)
 *                  attention    wq1 bq1 aq wq2 bq2 wk1 bk1 ak wk2 bk2  w0..w3  b0..b2  bn_scale0..2  bn_shift0..2
 *                               alpha0..2                                                    (nann_attn_desc)
 *                  precision.txt (optional, mlp / attention): "split" (default: split-f16 operands on the 16-bit
 *                               MFMA) | "exact" (f32-input MFMA)
 *                Every tensor is checked against the element count (d, seq_len) imply: NANN_ERR_SHAPE_MISMATCH.
 * nann_model_forward is forward() of build_opt_graph.py:91-107 for ONE user: user_seq f16
 * [seq_len, d] (l2 / mlp: its non-pad mean is the query vector; attention: [seq_len, 64]),
 * item_emb [n, d] rows as BlazeXlaOp receives them (already gathered) -> f32 logits[n].
 * All pointers device; workspace = nann_model_workspace_bytes() bytes; asynchronous on stream. */
typedef struct nann_model nann_model;
enum nann_model_kind { NANN_MODEL_L2 = 0, NANN_MODEL_MLP = 1, NANN_MODEL_ATTENTION = 2 };
int nann_model_load(const char* dir /*[host]*/, int32_t d, int32_t emb_dtype, int32_t seq_len, nann_model** out);
void nann_model_destroy(nann_model* m);
int nann_model_kind(const nann_model* m);
/* the l2 / mlp model as the scorer nann_search takes (borrowed; NULL for attention) */
const nann_scorer* nann_model_scorer(const nann_model* m);
int nann_model_workspace_bytes(const nann_model* m, int64_t* nbytes);
int nann_model_forward(const nann_model* m, const void* user_seq_f16, const void* item_emb, int64_t n,
                       float* logits, void* workspace, nann_stream_t stream);

/* BlazeXlaOp's `blaze_option_path` attr as BlazeXlaOp::ParseAttr reads it (UO/blaze_op/blaze_xla_kernel.cc:156-167): first
 * as the PATH of a text-format BlazeKernelOptions file (core/protobuf/config.proto:805-841; NANN_impls/nann/delivery/
 * opt_default.conf is the one build_opt_graph.py:104 passes), then the attr STRING ITSELF as text format; neither ->
 * NANN_ERR_IO "parse proto from ... failed" (the reference: errors::Internal).  A top-level field the message does not
 * have fails the parse, as protobuf's TextFormat does.  What the MI355X op acts on: wait_ms (admission deadline,
 * blaze_xla_kernel.cc:221-258) and run_mode (SKIP, :183-205); the rest is reported for the host's log.  attr [host]. */
typedef struct {
  int32_t struct_bytes;
  int32_t wait_ms;
  int32_t run_mode;               /* 0 DEFAULT, 1 BENCHMARK, 2 SKIP */
  int32_t xla_compilation;
  int32_t auto_mixed_precision;
  int32_t disable_output_padding;
  int32_t n_warmup_batchsize;     /* entries of warmup_batchsize (no warm-up here: rows are scored as they come) */
  int32_t max_warmup_batchsize;
  int32_t from_file;              /* 1: the attr named a readable file; 0: the attr string was the message */
} nann_blaze_options;
int nann_blaze_options_parse(const char* attr, nann_blaze_options* out);

/* ---- a6 + a7: resident index and the fused traversal -----------------------
 * The index is what the serving graph's HugeConst nodes hold
 * (build_opt_graph.py:83-90, 70): item_embs [N,d], item_ids i64[N], per level
 * CSR (values i32, row_splits i64[N+1]) for levels 0 and 1, enter_points
 * i32[E] (ascending, unique). */
typedef struct {
  int64_t n_items;
  int32_t d;
  int32_t emb_dtype;
  const void* item_embs;
  const int64_t* item_ids;
  const int32_t* nb_values[2];
  const int64_t* nb_row_splits[2];
  int64_t nb_nnz[2];
  const int32_t* enter_points;
  int64_t n_enter;
  int32_t on_device; /* 0: [host] pointers, copied to HBM once (HugeConst's one-time
                        H2D); 1: device pointers, borrowed for the handle's lifetime */
} nann_index_desc;
int nann_index_create(const nann_index_desc* desc /*[host]*/, nann_index** out);
void nann_index_destroy(nann_index* ix);
/* [host] out: n_items, d, n_enter, max row length at level 0 / 1, bitmap words */
int nann_index_info(const nann_index* ix, int64_t out[6]);
/* nann_index_create note: the probe below makes the call SYNCHRONISE the device (a hipMalloc, one small launch on the NULL
 * stream, two blocking copies, a hipFree): create indices at start-up, not next to traffic.  NANN_INDEX_PROBE=0 in the
 * environment skips it (the planner falls back to its estimate from the mean degree).  A probe that fails never fails the
 * creation and leaves nann_last_error as it found it.
 * The probe nann_index_create runs on every index of >= 4096 items (round 5): 64 of its own rows as queries, ef = min(64,
 * #enter points), L2 scorer, one small launch -- how many NEW nodes a level-0 round finds per frontier row on THIS graph,
 * which is what sizes a query's visited set (16K-slot hash set, two workgroups per CU; 32K slots, one; bitmap).  The planner
 * uses the measurement in place of rounds 1-4's guess from the mean degree: 1.15 x the 90th percentile of the probe's queries
 * (a query in the tail beyond it is rerun on the bitmap kernel like any query whose set would overflow).  out[5] = {valid
 * queries of the probe (0: none, the planner falls back to the guess), ef, mean, 90th percentile, max}. */
int nann_index_probe_info(const nann_index* ix, float out[5]);

#define NANN_NUM_ROUNDS 5
/* Where a query's visited set lives (the reference: a TemporaryVariable bitmap of N/32 words,
 * build_opt_graph.py:115-118).  AUTO picks per call: shards below 2^27 items -> an exact hash
 * set of visited ids in LDS -- LDS_HASH (16K slots, 64 KB: two queries per CU with the L2 scorer,
 * one with the matrix-core scorers) or, L2 scorer only, for beams whose visited set is expected to
 * outgrow it and for batches of at most one query per CU, LDS_HASH32 (32K slots, one query per CU);
 * a query whose set could overflow is rerun on a bitmap kernel inside the same call.  Otherwise
 * (matrix-core scorers with wide beams, larger shards) LDS_BITMAP when ceil(N/32) words fit the CU's
 * LDS next to the phase buffers, else HBM_BITMAP.  Results are identical in every mode (tested);
 * the knob exists for tests and tuning.  Per call: nann_search_options.traversal_mode; nann_set_traversal_mode stores the
 * process default of that field.  Thread-safe. */
enum nann_traversal_mode { NANN_TRAVERSAL_AUTO = 0, NANN_TRAVERSAL_LDS_BITMAP = 1,
                           NANN_TRAVERSAL_HBM_BITMAP = 2, NANN_TRAVERSAL_LDS_HASH = 3,
                           NANN_TRAVERSAL_LDS_HASH32 = 4 };
int nann_set_traversal_mode(int32_t mode);
/* Workgroup slots the persistent traversal grid leaves free on the device (default 0: it takes every CU, two
 * workgroups each for the L2 plan).  A host that runs other kernels NEXT TO a search -- the exchange of batch i on its
 * own stream while batch i + 1 is searched (8(e), nann_sharded_topk) -- reserves a few: without them those kernels
 * only start when the first traversal workgroups exit (measured, profiles/r4_overlap_*.json).  Per call:
 * nann_search_options.slot_reserve; this setter stores the process default of that field. */
int nann_set_search_reserve(int32_t workgroups);

/* Workspace bytes nann_search needs for (index, level_topn, n_queries), any scorer. */
int nann_search_workspace_bytes(const nann_index* ix, const int32_t level_topn[6] /*[host]*/,
                                int64_t n_queries, int64_t* nbytes);

/* ONE search operation, ONE entry point per query form (ABI v6):
 *     nann_search_opt        queries as vectors   q f32[n_queries, d]                 (l2 / mlp scorers)
 *     nann_search_model_opt  queries as comm_seq  f16[n_queries, seq_len, E]          (any model a BlazeXlaOp node names)
 * Both take per-launch maxima `level_topn_max`, an optional per-query `level_topn` table, per-call options and return the
 * planner's choice.  The five older spellings below (nann_search, nann_search_v, nann_search_ex, nann_search_model,
 * nann_search_model_v: rounds 1-5) are DEPRECATED THIN WRAPPERS -- each is one `return` of the canonical call with NULLs in
 * the places it does not have (tests/test_abi.py asserts that) -- kept so that hosts built against v5 keep linking; new
 * hosts (csrc/host/nann_serve.cpp, tf_ops/nann_tf_ops.cc, nann_amd/retrieval.py) call the canonical pair only.
 *
 * The whole schedule of build_model() (NANN_impls/nann/delivery/
 * build_opt_graph.py:109-149) for n_queries independent
 * queries, persistent workgroups that pull queries from a device-wide queue, visited set in LDS
 * (nann_traversal_mode).
 *   q            f32[n_queries, d]      (nann_user_seq_mean of comm_seq)
 *   level_topn   [host] i32[6]          (the `level_topn` feed)
 *   workspace    device, nann_search_workspace_bytes(...) bytes
 *   out_item_ids i64[n_queries, level_topn[5]]   ('top_k' fetch)
 *   out_scores   f32[n_queries, level_topn[5]]   (not a reference output;
 *                                                 for the 1e-5 check) or NULL
 *   out_index    i32[n_queries, level_topn[5]]   internal indices or NULL
 *   status       i32[n_queries]: per-query nann_status -- a query the
 *                reference would fail (k > n, empty score batch, ...) gets
 *                its code here and zeroed outputs
 *   counters     i32[n_queries, 3, NANN_NUM_ROUNDS] (F_r, G_r, S_r per round:
 *                rows walked, neighbours gathered, rows scored) or NULL
 * Asynchronous on `stream`. */
/* deprecated: thin wrapper of nann_search_opt */
int nann_search(const nann_index* ix, const nann_scorer* scorer, const float* q,
                int64_t n_queries, const int32_t level_topn[6], void* workspace,
                int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                int32_t* out_index, int32_t* status, int32_t* counters, nann_stream_t stream);

/* The same with level_topn PER QUERY, as the reference feeds it per request (build_opt_graph.py:75,151-159: `level_topn`
 * is a placeholder of the serving signature, so two requests of one batch may ask for different beams).
 *   level_topn_max [host] i32[6]  per-launch maxima: they size the workspace (nann_search_workspace_bytes) and the
 *                                 plan, and level_topn_max[5] is the ROW STRIDE of out_item_ids / out_scores / out_index
 *   level_topn     device i32[n_queries, 6] (NULL: every query takes level_topn_max -- nann_search's fast path)
 * Query i returns its level_topn[i][5] results at the head of row i, zeros behind.  An entry outside
 * [0, level_topn_max[j]] fails THAT query with NANN_ERR_BAD_ARGUMENT in status[i]; k > n and the other failures of
 * the reference are per query as before.  Each query's result is bit-identical to a uniform launch with its values. */
/* deprecated: thin wrapper of nann_search_opt */
int nann_search_v(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                  const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                  int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                  int32_t* status, int32_t* counters, nann_stream_t stream);

/* Same as nann_search, plus per-query time attribution for tuning: phase_ticks
 * i64[n_queries, NANN_NUM_PHASES] receives shader-clock ticks spent in
 * {bitmap zeroing, mark walks, CSR expand+walk, gather+score, top-k, other} followed by
 * sub-phases {top-k: load, search, collect, sort; expand: pass 1, pipeline loop, filter; hash-set
 * expand per piece: lookup + prefetch, id load wait, insert, barrier, check, rank + store}; NULL
 * disables the instrumentation (nann_search passes NULL). */
#define NANN_NUM_PHASES 19
/* deprecated: thin wrapper of nann_search_opt */
int nann_search_ex(const nann_index* ix, const nann_scorer* scorer, const float* q,
                   int64_t n_queries, const int32_t level_topn[6], void* workspace,
                   int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                   int32_t* out_index, int32_t* status, int32_t* counters, int64_t* phase_ticks,
                   nann_stream_t stream);

/* ---- options of a search call (round 5; ABI v5) -----------------------------------------------------------------
 * Every knob of the planner travels WITH THE CALL, so threads that share an index / scorer handle do not share a plan
 * (rounds 1-4 had three process-global setters beside an ABI that promises re-entrancy).  A field left at -1 takes the
 * process default: what the matching nann_set_* call stored, else the built-in value (AUTO, 0, 1, AUTO).
 *   traversal_mode  enum nann_traversal_mode: where a query's visited set lives
 *   slot_reserve    workgroup slots the persistent grids leave free for kernels of other streams (the exchange of a
 *                   sharded search that overlaps the next batch); the MLP's pipeline of phases honours it too
 *   preprojection   1: MLP / attention scorers read their per-(scorer, index) tables; 0: the kernels that read the
 *                   embedding rows (a host with no HBM to spare)
 *   mlp_form        enum nann_mlp_form: the fused kernel or the pipeline of phases for an MLP traversal on its table;
 *                   AUTO = the planner's measured rule (exact f32: phased; split-f16: phased up to 160 queries and for
 *                   wide beams)
 * nann_search_plan ([host], optional) receives what the planner chose; the number of queries the hash-set kernel handed
 * back to the bitmap kernel is read from the workspace afterwards (nann_search_reruns: synchronises `stream`). */
enum nann_mlp_form { NANN_MLP_FORM_AUTO = 0, NANN_MLP_FORM_FUSED = 1, NANN_MLP_FORM_PHASED = 2 };
typedef struct {
  int32_t struct_bytes;   /* sizeof(nann_search_options) of the caller's header (nann_search_options_init sets it) */
  int32_t traversal_mode;
  int32_t slot_reserve;
  int32_t preprojection;
  int32_t mlp_form;
} nann_search_options;
typedef struct {
  int32_t visited_set;           /* enum nann_traversal_mode of the main launch (of the traversal stages when phased) */
  int32_t fallback_visited_set;  /* ... of the rerun of handed-back queries */
  int32_t threads;               /* per workgroup */
  int32_t workgroups;            /* resident workgroups (= queries in flight) */
  int32_t phased;                /* 1: the MLP's pipeline of phases */
  int32_t table;                 /* 1: a pre-projected table is read */
  float est_visited;             /* the planner's estimate of a level's visited ids (hash-set capacity: 16K / 32K slots) */
  float worst_visited;           /* the bound from the index's maximum degrees */
} nann_search_plan;
void nann_search_options_init(nann_search_options* o);
/* nann_search_v + phase_ticks (uniform level_topn only) + options + plan.  level_topn == NULL: level_topn_max for all. */
int nann_search_opt(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                    const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace, int64_t workspace_bytes,
                    int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status, int32_t* counters,
                    int64_t* phase_ticks, const nann_search_options* options, nann_search_plan* plan, nann_stream_t stream);
/* queries of the LAST search on `workspace` that were rerun on the bitmap kernel (phased MLP: of its last chunk) */
int nann_search_reruns(const void* workspace, int64_t* n_rerun, nann_stream_t stream);

/* The serving signature in one call (build_opt_graph.py:151-159): comm_seq f16[n_queries, seq_len, E]
 * + level_topn -> top_k, scored by whatever model the BlazeXlaOp nodes of the graph name.  l2 / mlp:
 * nann_user_seq_mean + nann_search.  attention: the per-user projection once per request
 * (nann_attn_prepare), then the fused traversal with the attention + DNN scorer on the matrix cores
 * (scores within 1e-5 of the oracle restatement: device expf / MFMA order; ids tie-aware).
 * workspace: nann_search_model_workspace_bytes(); other arguments as nann_search. */
int nann_search_model_workspace_bytes(const nann_index* ix, const nann_model* m, const int32_t level_topn[6],
                                      int64_t n_queries, int64_t* nbytes);
/* deprecated: thin wrapper of nann_search_model_opt */
int nann_search_model(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                      const int32_t level_topn[6], void* workspace, int64_t workspace_bytes,
                      int64_t* out_item_ids, float* out_scores, int32_t* out_index, int32_t* status,
                      int32_t* counters, nann_stream_t stream);

/* per-query level_topn (nann_search_v) and the table lifecycle (nann_scorer_prepare ...) for a model */
/* deprecated: thin wrapper of nann_search_model_opt */
int nann_search_model_v(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                        const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                        int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                        int32_t* status, int32_t* counters, nann_stream_t stream);
/* ... with per-call options and the planner's choice (see nann_search_opt) */
int nann_search_model_opt(const nann_index* ix, const nann_model* m, const void* comm_seq_f16, int64_t n_queries,
                          const int32_t level_topn_max[6], const int32_t* level_topn, void* workspace,
                          int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores, int32_t* out_index,
                          int32_t* status, int32_t* counters, const nann_search_options* options, nann_search_plan* plan,
                          nann_stream_t stream);
int nann_model_prepare(const nann_model* m, const nann_index* ix, nann_stream_t stream);
int nann_model_release(const nann_model* m, const nann_index* ix);
int nann_model_table_bytes(const nann_model* m, const nann_index* ix, int64_t* table_bytes, int64_t* resident_bytes);

/* ---- 8(f3): the evaluation graph's traversal, one kernel per batch of users ---------------
 * Model.retrieval + search_level (NANN_impls/nann/model.py:299-362), the traversal behind
 * main.py --job-type test: start level scored whole, then levels 1 and 0 with
 * num_scoring_per_level[level] rounds each; neighbours taken as an ascending SET minus visited
 * (tf.unique + tf.sets.difference), top_k = min(k, n), next frontier = new nodes scoring at least
 * the worst kept result, an exhausted frontier is not an error.  Arrays are indexed by level
 * (config.py:50-58); num_scoring_per_level[2] must be 1, top_k_per_level and topk_eval in [1, 2048].
 * Outputs [n_queries, topk_eval] (out_scores / out_index may be NULL); n_out[q] = valid rows of
 * query q (the rest is zero); status[q] as nann_search (NANN_ERR_CAPACITY: more than 1024 new nodes
 * tie at the threshold of one round).  Bit-identical to oracle_search_eval for the l2 and exact mlp
 * scorers.  workspace: nann_search_eval_workspace_bytes(ix, model or NULL, n_queries).
 * nann_search_eval_model: comm_seq f16[n_queries, seq_len, E] in, as nann_search_model. */
int nann_search_eval_workspace_bytes(const nann_index* ix, const nann_model* m, int64_t n_queries,
                                     int64_t* nbytes);
int nann_search_eval(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                     const int32_t num_scoring_per_level[3], const int32_t top_k_per_level[3],
                     int32_t topk_eval, void* workspace, int64_t workspace_bytes, int64_t* out_item_ids,
                     float* out_scores, int32_t* out_index, int32_t* n_out, int32_t* status,
                     nann_stream_t stream);
/* nann_search_eval + counters i32[n_queries, 3] (device, or NULL): rows walked F, neighbours gathered G, rows scored S (the
 * enter points included), summed over a user's rounds -- what SURVEY.md 8(d)'s byte formula is evaluated on (bench.py's
 * roofline of the f3 line); a user whose request failed keeps zeros. */
int nann_search_eval_ex(const nann_index* ix, const nann_scorer* scorer, const float* q, int64_t n_queries,
                        const int32_t num_scoring_per_level[3], const int32_t top_k_per_level[3], int32_t topk_eval,
                        void* workspace, int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                        int32_t* out_index, int32_t* n_out, int32_t* status, int32_t* counters, nann_stream_t stream);
int nann_search_eval_model(const nann_index* ix, const nann_model* m, const void* comm_seq_f16,
                           int64_t n_queries, const int32_t num_scoring_per_level[3],
                           const int32_t top_k_per_level[3], int32_t topk_eval, void* workspace,
                           int64_t workspace_bytes, int64_t* out_item_ids, float* out_scores,
                           int32_t* out_index, int32_t* n_out, int32_t* status, nann_stream_t stream);

/* ---- 8(e): merge of per-shard top-k lists ----------------------------------
 * scores f32[n_queries, n_shards, k_in], ids i64[n_queries, n_shards, k_in]
 * (shard-major as all-gathered); concat in shard order, then TopKV2 order
 * (score desc, ties -> lower shard then lower local rank).
 * nann_merge_topk: device pointers; nann_merge_topk_host: [host] pointers. */
int nann_merge_topk(const float* scores, const int64_t* ids, int64_t n_queries, int32_t n_shards,
                    int32_t k_in, int32_t k_out, float* out_scores, int64_t* out_ids,
                    nann_stream_t stream);
int nann_merge_topk_host(const float* scores, const int64_t* ids, int64_t n_queries,
                         int32_t n_shards, int32_t k_in, int32_t k_out, float* out_scores,
                         int64_t* out_ids);

/* ---- 8(e): the exchange step owned by the C++ host (one process per GPU) -----------------
 * Rank g searches every query on shard g (nann_search), then nann_sharded_topk packs its
 * [scores | item ids] lists into one record, exchanges the records with ONE ncclAllGather
 * (RCCL over xGMI, k_in x 12 B per query per shard) and merges them on the device with
 * nann_merge_topk's order.  The reference has no collective on this path (replicas only,
 * blaze-benchmark/benchmark/core/model.cc:192-235); BASELINE configs 4-5 define this one.
 * RCCL is bound at run time (an RCCL already loaded in the process is shared, else
 * /opt/rocm/lib/librccl.so.1); without it these calls return NANN_ERR_UNSUPPORTED.
 *   nann_comm_get_unique_id  one rank draws the 128-byte id (ncclGetUniqueId) and the host
 *                            hands it to the others by its own means (file, socket, MPI ...)
 *   nann_comm_create         collective over all ranks (ncclCommInitRank) on the calling
 *                            thread's current HIP device; world == 1 needs no id and no RCCL;
 *                            world > 1 with id == NULL makes a LOOPBACK communicator for
 *                            single-process tests (every shard returns this rank's record)
 *   status (may be NULL)     a query with status != 0 on this shard contributes scores -inf /
 *                            ids 0: it is never selected while another shard holds real
 *                            candidates, and surfaces as (-inf, 0) entries otherwise
 *   workspace                device, nann_sharded_topk_workspace_bytes(...) bytes
 * Asynchronous on `stream`; every rank must make the same sequence of calls. */
typedef struct nann_comm nann_comm;
#define NANN_COMM_ID_BYTES 128
int nann_comm_get_unique_id(void* id /*[host] NANN_COMM_ID_BYTES*/);
int nann_comm_create(int32_t world, int32_t rank, const void* id /*[host]*/, nann_comm** out);
void nann_comm_destroy(nann_comm* c);
/* *world = shards of this communicator; *rccl_ranks = ncclCommCount of the RCCL communicator behind it (0: none -- a
 * loopback, or a one-shard communicator created without an id): what a multi-GPU bench line states so that "N ranks
 * exchanged over RCCL" is a measured fact. */
int nann_comm_ranks(const nann_comm* c, int32_t* world, int32_t* rccl_ranks);
/* enabled: HIP events around the three parts of every later nann_sharded_topk call on its stream;
 * nann_comm_last_breakdown -> ms[3] = {pack, all-gather, merge} of the LAST call (waits for it).  Loopback communicators
 * only (one-GPU measurements of the overlap and the slot reserve, tools/overlap_bench.py): loopback_wait_us > 0 puts a
 * kernel of 16 waiting workgroups in front of the stand-in copies -- an exchange that holds its slots and its stream as long
 * as 8 GPUs' all-gather over xGMI would (~1.5 ms at configs[3]) --, loopback_repeat >= 1 issues the copies that many times. */
int nann_comm_set_timing(nann_comm* c, int32_t enabled, int32_t loopback_repeat, int32_t loopback_wait_us);
int nann_comm_last_breakdown(nann_comm* c, float ms[3]);
/* A dead rank must not hang the node (round 6).  nann_sharded_topk only ENQUEUES; a host that then waits on the stream waits
 * for ever when a peer never joins the all-gather.  Instead:
 *   nann_comm_wait   bounded wait for the LAST exchange enqueued on this communicator: polls its completion event next to
 *                    ncclCommGetAsyncError.  Done -> NANN_OK.  RCCL reports an asynchronous error, or timeout_ms (>= 0; < 0: no
 *                    deadline) elapse -> the communicator is ABORTED (ncclCommAbort: RCCL's kernels see the flag and leave, so
 *                    the stream drains) and NANN_ERR_HIP is returned with the reason in nann_last_error.
 *   nann_comm_abort  the same abort on the host's own decision (a watchdog, a failed health check of a peer).
 * An aborted communicator fails every later call with NANN_ERR_HIP; the host destroys it and creates a new one with the ranks
 * that are left.  nann_sharded_topk itself refuses to enqueue on a communicator whose RCCL state already reports an error. */
int nann_comm_wait(nann_comm* c, int32_t timeout_ms);
int nann_comm_abort(nann_comm* c);
int nann_sharded_topk_workspace_bytes(int32_t world, int64_t n_queries, int32_t k_in, int64_t* nbytes);
int nann_sharded_topk(nann_comm* c, const float* scores, const int64_t* ids, const int32_t* status,
                      int64_t n_queries, int32_t k_in, int32_t k_out, void* workspace,
                      int64_t workspace_bytes, float* out_scores, int64_t* out_ids, nann_stream_t stream);

/* ---- 8(f1): HNSW index construction on the device -------------------------------------------------
 * What the reference does on the host with faiss.IndexHNSWFlat(d, M).add(embeddings)
 * (NANN_impls/nann/delivery/build_hnsw_index.py:33-35): the HNSW insertion algorithm (Malkov & Yashunin alg. 1-4
 * with Faiss' conventions: L2, M links above level 0 and 2M at level 0, efConstruction = 40, selection heuristic
 * without keepPrunedConnections), batched -- nodes go in by descending level in batches of at most a quarter of
 * the graph built so far (<= 16384), every node of a batch searches the graph of the earlier batches with one
 * wavefront, back-links are applied per target in a deterministic order (csrc/nann_hnsw_build.hip).  Index
 * contents are not a parity target (Faiss' own are thread-schedule dependent); layout, invariants and recall are.
 *   nann_hnsw_draw_levels   [host] levels[i] = number of levels of node i (Faiss convention, P(levels > l) = M^-l),
 *                           same draw as the CPU builder's; *n_up_rows = sum(levels - 1) = rows of adj_up
 *   nann_hnsw_build_device  item_embs device [n, d] f16 | bf16 (d in 64 | 128 | 256), levels [host];
 *                           outputs (device, caller-allocated): adj0 i32[n, 2M] (-1 = empty slot), up_row i32[n]
 *                           (first row of node i in adj_up, -1 if it has one level), adj_up i32[n_up_rows, M]
 *                           (level l >= 1 of node i: row up_row[i] + l - 1).  Synchronous (returns when built).
 * nann_amd/index_build.py exports these as build_hnsw_index.py:41-66 does (enter points = levels > start_level,
 * per-level CSR over all items with the -1 slots dropped). */
int nann_hnsw_draw_levels(int64_t n_items, int32_t M, uint64_t seed, int32_t* levels /*[host]*/, int64_t* n_up_rows);
int nann_hnsw_build_device(const void* item_embs, int64_t n_items, int32_t d, int32_t emb_dtype, int32_t M,
                           int32_t ef_construction, const int32_t* levels /*[host]*/, int32_t* adj0, int32_t* up_row,
                           int32_t* adj_up, nann_stream_t stream);
/* The same with alg. 4's keepPrunedConnections switch (keep_pruned != 0: a row's free slots are filled with the nearest
 * candidates the heuristic discarded -- Faiss and hence the reference leave it off): rows fill up to their cap, mean level-0
 * degree ~55 of 64 instead of ~17.  The dense-graph family SURVEY.md 8's gather bound (L0 gathered <= ef * 64) is about;
 * bench.py --graph hnsw_dense and the planner tests run on it. */
int nann_hnsw_build_device_ex(const void* item_embs, int64_t n_items, int32_t d, int32_t emb_dtype, int32_t M,
                              int32_t ef_construction, int32_t keep_pruned, const int32_t* levels /*[host]*/, int32_t* adj0,
                              int32_t* up_row, int32_t* adj_up, nann_stream_t stream);

/* ---- 8(f2): the reference's own scorer model behind the BlazeXlaOp contract -------------
 * NANN_impls/nann/model/model.py:189-233 + model_util.py:70-97: softmax attention of the candidate
 * over the user's behaviour sequence u f16[L, 64] (comm_seq, build_opt_graph.py:76-79), then a DNN
 * 128-64-32-1 with batch norm (folded to scale/shift) and PReLU, last layer bias-free.  f32 logits,
 * rows scored independently.  This build: E = 64, L <= 64, d in {64, 128}, rows f16 or bf16.
 * All descriptor pointers are [host] f32; the scorer owns device copies. */
typedef struct nann_attn_scorer nann_attn_scorer;
typedef struct {
  int32_t d;         /* item embedding dim */
  int32_t emb_dtype; /* NANN_F16 | NANN_BF16 */
  int32_t seq_len;   /* L */
  const float *wq1, *bq1, *aq; /* [d,128] [128] [128] */
  const float *wq2, *bq2;      /* [128,256] [256] */
  const float *wk1, *bk1, *ak; /* [64,128] [128] [128] */
  const float *wk2, *bk2;      /* [128,256] [256] */
  const float* w[4];           /* [64+d,128] [128,64] [64,32] [32] */
  const float* b[3];
  const float* bn_scale[3];
  const float* bn_shift[3];
  const float* alpha[3];
  int32_t precision;  /* enum nann_mlp_precision: NANN_MLP_SPLIT_F16 (every f32 operand as hi + lo f16 on the
                       * 16-bit MFMA, 3x fewer matrix cycles; logits within ~1e-6 of the exact form; the default),
                       * NANN_MLP_EXACT_F32 (f32-input MFMA) */
} nann_attn_desc;
int nann_attn_scorer_create(const nann_attn_desc* desc /*[host]*/, nann_attn_scorer** out);
void nann_attn_scorer_destroy(nann_attn_scorer* s);
/* Per-user part, once per request: user_seq f16[n_users, L, 64] -> kt f32[n_users, 256, 64]
 * (the projected keys, transposed, zero beyond L) and upad f32[n_users, 64, 64] (the sequence,
 * zero-padded); both caller-owned device buffers. */
int nann_attn_prepare(const nann_attn_scorer* s, const void* user_seq_f16, int64_t n_users, float* kt,
                      float* upad, nann_stream_t stream);
/* Logits of n candidate rows for ONE user (kt / upad of that user); table / indices / errors as
 * nann_score. */
int nann_attn_score(const nann_attn_scorer* s, const float* kt, const float* upad, const void* table,
                    int64_t n_table_rows, const int32_t* indices, int64_t n, float* out_scores,
                    int64_t* bad_i, nann_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NANN_HIP_H_ */
