// nann_attn.h -- the reference's own scorer model on the matrix cores.
//
// What sits behind BlazeXlaOp in the reference (NANN_impls/nann/model/model.py:189-233,
// model_util.py:9-11,32-67,70-97), for one user sequence u [L, E] and a candidate row e [d]:
//   q1 = prelu(e Wq1 + bq1)  [2E]      k1_l = prelu(u_l Wk1 + bk1)  [2E]
//   q_ = q1 Wq2 + bq2        [4E]      k_l  = k1_l Wk2 + bk2        [4E]
//   att_l = <q_, k_l> / sqrt(4E);  p = softmax_l(att);  a = sum_l p_l u_l   [E]
//   x = [a ; e];  three times x = prelu(bn(x W + b));  logit = x W4
// with E = 64, L <= 64 (50 in the reference), DNN widths 128-64-32, inference batch norm folded
// to scale/shift.  Verified on hardware against the oracle restatement (oracle_attn_*): the per-user
// projection bit for bit, logits within 1e-5 (MFMA order, device expf).  Parity with the reference's
// frozen graph itself is unpinned (no TensorFlow in the image, no checkpoint in the tree).
//
// Mapping.  One wavefront = 32 candidates, every dense layer is the same step on
// v_mfma_f32_32x32x2_f32: the activations of a layer stay in the 32x32 C/D register layout
// (lane (c, s) holds, for candidate c, the units (r&3) + 8(r>>2) + 4s of each 32-unit tile) and
// are consumed IN PLACE as the B operand of the next layer, the A operand being the rows of W in
// that same order, staged through LDS (nann_mlp.h uses the identical trick for its layer 2).
// The candidate row is loaded straight into that layout, so even the first layers are the same
// step.  The user side (k_l for all l, transposed, and the zero-padded sequence) is computed
// once per user by k_attn_prepare.  The attention logits are accumulated q_-tile by q_-tile, so
// q_ never exists in full; the softmax runs in registers (16 positions per lane and tile, the
// other 16 in lane ^ 32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nann {

constexpr int kAttnE = 64;    // user-sequence embedding dim
constexpr int kAttnLP = 64;   // sequence positions, padded
constexpr int kAttnNT = 512;  // 8 wavefronts x 32 candidates per pass
constexpr int kAttnSlice = 8192;  // floats staged per step (32 KB)
constexpr int kAttnXResFloats = 64 * 64 + 64 * 128 + 128 * 64 + 64 * 32;  // f32 form, resident: padded sequence, W1a, W2, W3 (88 KB)
constexpr int kAttnVecFloats = 1536;  // split form: the pre-scaled small vectors, kept in LDS behind the slices (6 KB)

struct AttnParams {  // device pointers, f32
  const float *wq1, *bq1, *aq;  // [d,128] [128] [128]
  const float *wq2, *bq2;       // [128,256] [256]
  const float *wk1, *bk1, *ak;  // [64,128] [128] [128]
  const float *wk2, *bk2;       // [128,256] [256]
  const float *w1, *b1, *s1, *t1, *a1;  // [64+d,128] ...
  const float *w2, *b2, *s2, *t2, *a2;  // [128,64]
  const float *w3, *b3, *s3, *t3, *a3;  // [64,32]
  const float* w4;                       // [32]
  int d, L;
  // split-f16 form (nann_attn_split.h): A fragments of every weight matrix, hi / lo f16 planes x 2^7, packed on
  // the host in MFMA lane order, [output tile][16-deep chunk][plane][64 lanes] x 16 B; and the pre-scaled vectors
  const uint4 *pq1, *pq2, *pw1a, *pw1e, *pw2, *pw3;
  const float* pvec;
};

// defined in nann_attn_inst.hip
int launch_attn_prepare(hipStream_t st, const AttnParams& P, const void* user_seq_f16, long long n_users,
                        float* kt, float* upad);
int launch_score_attn(int dt, unsigned blocks, hipStream_t st, const AttnParams& P, const float* kt,
                      const float* upad, const void* table, long long n_table_rows, const int32_t* indices,
                      long long n, float* scores, long long* bad_i);

// split-f16 form, defined in nann_attn_split_inst.hip
int launch_attn_prepare_split(hipStream_t st, const AttnParams& P, const void* user_seq_f16, long long n_users,
                              float* kt, float* upad);
int launch_score_attn_split(int dt, unsigned blocks, hipStream_t st, const AttnParams& P, const float* kt,
                            const float* upad, const void* table, long long n_table_rows, const int32_t* indices,
                            long long n, float* scores, long long* bad_i);

}  // namespace nann
