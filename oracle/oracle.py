"""ctypes binding of the CPU oracle (oracle/nann_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under nann_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

OK = 0
ERR_INVALID_RAGGED_PARAMS = 1
ERR_INVALID_RAGGED_INDICES = 2
ERR_INVALID_RAGGED_INPUT = 3
ERR_TOPK_K_GT_N = 4
ERR_INDEX_OUT_OF_RANGE = 5
ERR_EMPTY_SCORE_BATCH = 6
ERR_BAD_ARGUMENT = 7
ERR_TOPK_SCALAR_INPUT = 8

EMB_F16, EMB_BF16, EMB_F32 = 0, 1, 2
SCORER_L2, SCORER_MLP, SCORER_ATTN = 0, 1, 2
NUM_ROUNDS = 5


def _cpu_has(*flags):
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(x in have for x in flags)
    except OSError:
        pass
    return False


def build(force=False):
    """Compile the oracle with gcc (seconds).  Returns the path of the .so to load."""
    fast = os.path.join(_BUILD, "liboracle.so")
    gen = os.path.join(_BUILD, "liboracle_generic.so")
    src = os.path.join(_HERE, "nann_oracle.c")
    stale = (not os.path.exists(fast) or not os.path.exists(gen)
             or os.path.getmtime(src) > os.path.getmtime(fast))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return fast if _cpu_has("fma", "avx2", "f16c") else gen


class ScorerStruct(C.Structure):
    _fields_ = [("kind", C.c_int), ("d", C.c_int), ("emb_dtype", C.c_int),
                ("h1", C.c_int), ("h2", C.c_int),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("alpha1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("alpha2", C.c_void_p),
                ("w3", C.c_void_p), ("attn", C.c_void_p)]


class IndexStruct(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("d", C.c_int), ("emb_dtype", C.c_int),
                ("item_embs", C.c_void_p), ("item_ids", C.c_void_p),
                ("nb_values", C.c_void_p * 2), ("nb_row_splits", C.c_void_p * 2),
                ("nb_nnz", C.c_int64 * 2),
                ("enter_points", C.c_void_p), ("n_enter", C.c_int64)]


class CountersStruct(C.Structure):
    _fields_ = [("frontier", C.c_int64 * NUM_ROUNDS), ("gathered", C.c_int64 * NUM_ROUNDS),
                ("scored", C.c_int64 * NUM_ROUNDS)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_half_to_float.restype = C.c_float
        _lib.oracle_half_to_float.argtypes = [C.c_uint16]
        _lib.oracle_float_to_half.restype = C.c_uint16
        _lib.oracle_float_to_half.argtypes = [C.c_float]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def group_gather(params_values, params_row_splits, indices_values, indices_row_splits, unique=False):
    """-> (status, ragged_code, ret_values int32, ret_row_splits int64); unique=True: per group the set of the
    gathered values in first-occurrence order (GroupGather_kernel.cc:91-131)"""
    pv = _c(params_values, np.int32); prs = _c(params_row_splits, np.int64)
    iv = _c(indices_values, np.int64); irs = _c(indices_row_splits, np.int64)
    ors = np.zeros(max(len(irs), 1), np.int64)
    n_out = C.c_int64(0); n_os = C.c_int64(0); code = C.c_int(0)
    f = lib().oracle_group_gather_unique_i32 if unique else lib().oracle_group_gather_i32
    args = [_p(pv), C.c_int64(len(pv)), _p(prs), C.c_int64(len(prs)), _p(iv), C.c_int64(len(iv)),
            _p(irs), C.c_int64(len(irs))]
    rc = f(*args, None, C.c_int64(0), _p(ors), C.byref(n_out), C.byref(n_os), C.byref(code))
    if rc:
        return rc, code.value, np.zeros(0, np.int32), np.zeros(0, np.int64)
    out = np.zeros(max(n_out.value, 1), np.int32)
    rc = f(*args, _p(out), C.c_int64(len(out)), _p(ors), C.byref(n_out), C.byref(n_os),
           C.byref(code))
    return rc, code.value, out[:n_out.value].copy(), ors[:n_os.value].copy()


def bitmap_ref_difference(values, row_splits, bitmap):
    """bitmap (np.int32 array) is mutated in place. -> (status, code, c_values, c_row_splits)"""
    v = _c(values, np.int32); rs = _c(row_splits, np.int64)
    assert bitmap.dtype == np.int32 and bitmap.flags.c_contiguous
    out = np.zeros(max(len(v), 1), np.int32)
    ors = np.zeros(max(len(rs), 1), np.int64)
    n_out = C.c_int64(0); n_os = C.c_int64(0); code = C.c_int(0)
    rc = lib().oracle_bitmap_ref_difference_i32(
        _p(v), C.c_int64(len(v)), _p(rs), C.c_int64(len(rs)), _p(bitmap), C.c_int64(len(bitmap)),
        _p(out), _p(ors), C.byref(n_out), C.byref(n_os), C.byref(code))
    if rc:
        return rc, code.value, np.zeros(0, np.int32), np.zeros(0, np.int64)
    return rc, code.value, out[:n_out.value].copy(), ors[:n_os.value].copy()


def bitmap_init(idx, length):
    idx = _c(idx, np.int32)
    bm = np.zeros(max(length, 1), np.int32)
    rc = lib().oracle_bitmap_init_i32(_p(idx), C.c_int64(len(idx)), C.c_int32(length), _p(bm))
    return rc, bm[:max(length, 0)]


def bitmap_difference(idx_next, idx_flag):
    v = _c(idx_next, np.int32); f = _c(idx_flag, np.int32)
    out = np.zeros(max(len(v), 1), np.int32); fnew = np.zeros(max(len(f), 1), np.int32)
    n_out = C.c_int64(0)
    rc = lib().oracle_bitmap_difference_i32(_p(v), C.c_int64(len(v)), _p(f), C.c_int64(len(f)), _p(out),
                                            C.byref(n_out), _p(fnew))
    return rc, out[:n_out.value].copy(), fnew[:len(f)]


def fingerprint64(s):
    b = s.encode() if isinstance(s, str) else bytes(s)
    lib().oracle_fingerprint64.restype = C.c_uint64
    return int(lib().oracle_fingerprint64(C.c_char_p(b), C.c_int64(len(b))))


def bloom_filter_difference(values, row_splits, idx_flag, bucket=0, bucket_size=1):
    """idx_flag: int32 array, mutated in place.  -> (status, ragged_code, c_values, c_row_splits)"""
    v = _c(values, np.int32); rs = _c(row_splits, np.int64)
    assert idx_flag.dtype == np.int32 and idx_flag.flags["C_CONTIGUOUS"]
    out = np.zeros(max(len(v), 1), np.int32); out_rs = np.zeros(max(len(rs), 1), np.int64)
    n_out, n_rs, code = C.c_int64(0), C.c_int64(0), C.c_int(0)
    rc = lib().oracle_bloom_filter_difference_i32(_p(v), C.c_int64(len(v)), _p(rs), C.c_int64(len(rs)), _p(idx_flag),
                                                  C.c_int64(len(idx_flag)), C.c_int64(bucket), C.c_int64(bucket_size),
                                                  _p(out), _p(out_rs), C.byref(n_out), C.byref(n_rs), C.byref(code))
    return rc, code.value, out[: n_out.value], out_rs[: n_rs.value]


def blaze_topk(values, k):
    """BlazeTopK (UO/topk_op/BlazeTopK_kernel.cc:64-101): the k largest values, sorted by value; the order of
    equal values is unspecified there (std::partial_sort), so any top_k answer is one of its answers."""
    v = _c(values, np.float32)
    if k < 0 or k > len(v):
        return ERR_BAD_ARGUMENT, np.zeros(0, np.float32), np.zeros(0, np.int32)
    return topk(v, k)


def batch_topk_on_rt(values, row_splits, k, ascending=False):
    v = _c(values, np.float32); rs = _c(row_splits, np.int64)
    scalar = np.ndim(k) == 0
    kk = _c(np.atleast_1d(k), np.int64)
    ov = np.zeros(max(len(v), 1), np.float32); oi = np.zeros(max(len(v), 1), np.int64)
    ors = np.zeros(max(len(rs), 1), np.int64)
    n_out = C.c_int64(0); n_os = C.c_int64(0); code = C.c_int(0)
    rc = lib().oracle_batch_topk_on_rt_f32(_p(v), C.c_int64(len(v)), _p(rs), C.c_int64(len(rs)), _p(kk),
                                           C.c_int(1 if scalar else 0), C.c_int(1 if ascending else 0),
                                           _p(ov), _p(oi), _p(ors), C.byref(n_out), C.byref(n_os),
                                           C.byref(code))
    return rc, ov[:n_out.value].copy(), oi[:n_out.value].copy(), ors[:n_os.value].copy()


def gather_rows(params, idx):
    params = np.ascontiguousarray(params)
    idx = _c(idx, np.int32)
    n_rows = params.shape[0]
    row_bytes = params.strides[0] if params.ndim > 1 else params.itemsize
    out = np.zeros((len(idx),) + params.shape[1:], params.dtype)
    bad = C.c_int64(-1)
    rc = lib().oracle_gather_rows(_p(params), C.c_int64(n_rows), C.c_int64(row_bytes), _p(idx),
                                  C.c_int64(len(idx)), _p(out), C.byref(bad))
    return rc, out, bad.value


def topk(values, k):
    v = _c(values, np.float32)
    ov = np.zeros(max(k, 1), np.float32); oi = np.zeros(max(k, 1), np.int32)
    rc = lib().oracle_topk_f32(_p(v), C.c_int64(len(v)), C.c_int32(k), _p(ov), _p(oi))
    return rc, ov[:max(k, 0)], oi[:max(k, 0)]


def user_seq_mean(seq_f16):
    seq = _c(seq_f16, np.float16)
    L, d = seq.shape
    q = np.zeros(d, np.float32)
    lib().oracle_user_seq_mean(_p(seq), C.c_int(L), C.c_int(d), _p(q))
    return q


def _emb_dtype(a):
    if a.dtype == np.float16:
        return EMB_F16
    if a.dtype == np.float32:
        return EMB_F32
    if a.dtype == np.uint16:  # bf16 bit patterns
        return EMB_BF16
    raise TypeError(f"unsupported embedding dtype {a.dtype}")


class Scorer:
    """kind 'l2' or 'mlp' (weights: dict w1,b1,alpha1,w2,b2,alpha2,w3 as f32 arrays)."""

    def __init__(self, kind, d, emb_dtype, weights=None, attn_model=None):
        """kind "attention": attn_model = AttnModel; the queries of search/search_batch are then the user
        sequences f32[L * E] instead of vectors f32[d]."""
        self.s = ScorerStruct()
        self.s.kind = {"l2": SCORER_L2, "mlp": SCORER_MLP, "attention": SCORER_ATTN}[kind]
        if kind == "attention":
            self._attn = attn_model
            self.s.attn = C.cast(C.pointer(attn_model.s), C.c_void_p)
        self.s.d = d
        self.s.emb_dtype = emb_dtype
        self._keep = {}
        if kind == "mlp":
            for name in ("w1", "b1", "alpha1", "w2", "b2", "alpha2", "w3"):
                a = _c(weights[name], np.float32)
                self._keep[name] = a
                setattr(self.s, name, a.ctypes.data)
            self.s.h1 = self._keep["w1"].shape[1]
            self.s.h2 = self._keep["w2"].shape[1]
            assert self._keep["w1"].shape[0] == 2 * d


def score_rows(scorer, q, rows):
    rows = np.ascontiguousarray(rows)
    q = _c(q, np.float32)
    n = rows.shape[0]
    out = np.zeros(max(n, 1), np.float32)
    rc = lib().oracle_score_rows(C.byref(scorer.s), _p(q), _p(rows), C.c_int64(n), _p(out))
    return rc, out[:n]


class Index:
    """Holds the reference's arrays (build_hnsw_index.py layout) for the oracle."""

    def __init__(self, item_embs, item_ids, nb_values, nb_row_splits, enter_points):
        self.embs = np.ascontiguousarray(item_embs)
        self.ids = _c(item_ids, np.int64)
        self.nbv = [_c(nb_values[l], np.int32) for l in (0, 1)]
        self.nbrs = [_c(nb_row_splits[l], np.int64) for l in (0, 1)]
        self.ep = _c(enter_points, np.int32)
        s = IndexStruct()
        s.n_items, s.d = self.embs.shape
        s.emb_dtype = _emb_dtype(self.embs)
        s.item_embs = self.embs.ctypes.data
        s.item_ids = self.ids.ctypes.data
        for l in (0, 1):
            s.nb_values[l] = self.nbv[l].ctypes.data
            s.nb_row_splits[l] = self.nbrs[l].ctypes.data
            s.nb_nnz[l] = len(self.nbv[l])
        s.enter_points = self.ep.ctypes.data
        s.n_enter = len(self.ep)
        self.s = s


def search(index, scorer, q, level_topn):
    """One query. -> (status, item_ids i64[k], scores f32[k], idx i32[k], counters dict)"""
    q = _c(q, np.float32)
    t = _c(level_topn, np.int32)
    k = int(t[5])
    ids = np.zeros(max(k, 1), np.int64); sc = np.zeros(max(k, 1), np.float32)
    ix = np.zeros(max(k, 1), np.int32)
    ctr = CountersStruct()
    rc = lib().oracle_search(C.byref(index.s), C.byref(scorer.s), _p(q), _p(t), _p(ids), _p(sc),
                             _p(ix), C.byref(ctr))
    counters = {"frontier": list(ctr.frontier), "gathered": list(ctr.gathered),
                "scored": list(ctr.scored)}
    return rc, ids[:k], sc[:k], ix[:k], counters


def search_eval(index, scorer, q, num_scoring=(3, 1, 1), top_k_per_level=(400, 200, 100), topk_eval=200):
    """Eval-graph variant (model.py:299-362). -> (status, item_ids, scores, idx)"""
    q = _c(q, np.float32)
    ns = _c(num_scoring, np.int32); tk = _c(top_k_per_level, np.int32)
    ids = np.zeros(max(topk_eval, 1), np.int64); sc = np.zeros(max(topk_eval, 1), np.float32)
    ix = np.zeros(max(topk_eval, 1), np.int32)
    n = C.c_int32(0)
    rc = lib().oracle_search_eval(C.byref(index.s), C.byref(scorer.s), _p(q), _p(ns), _p(tk),
                                  C.c_int32(topk_eval), _p(ids), _p(sc), _p(ix), C.byref(n))
    return rc, ids[:n.value], sc[:n.value], ix[:n.value]


def search_batch(index, scorer, q, level_topn, n_threads=1):
    """-> (status[nq], item_ids[nq,k], scores[nq,k], idx[nq,k], counters[nq,3,5] int64)"""
    q = _c(q, np.float32)
    nq = q.shape[0]
    t = _c(level_topn, np.int32)
    k = int(t[5])
    ids = np.zeros((nq, max(k, 1)), np.int64); sc = np.zeros((nq, max(k, 1)), np.float32)
    ix = np.zeros((nq, max(k, 1)), np.int32)
    ctr = (CountersStruct * nq)()
    st = np.zeros(nq, np.int32)
    lib().oracle_search_batch(C.byref(index.s), C.byref(scorer.s), _p(q), C.c_int64(nq), _p(t),
                              _p(ids), _p(sc), _p(ix), ctr, _p(st), C.c_int(n_threads))
    counters = np.frombuffer(ctr, dtype=np.int64).reshape(nq, 3, NUM_ROUNDS).copy()
    return st, ids[:, :k], sc[:, :k], ix[:, :k], counters


def brute_force(index, scorer, q, k):
    q = _c(q, np.float32)
    oi = np.zeros(k, np.int32); ov = np.zeros(k, np.float32)
    rc = lib().oracle_brute_force(C.byref(index.s), C.byref(scorer.s), _p(q), C.c_int32(k), _p(oi),
                                  _p(ov))
    return rc, oi, ov


def merge_topk(scores, ids, k_out):
    s = _c(scores, np.float32); i = _c(ids, np.int64)
    n_shards, k_in = s.shape
    os_ = np.zeros(k_out, np.float32); oi = np.zeros(k_out, np.int64)
    rc = lib().oracle_merge_topk(_p(s), _p(i), C.c_int(n_shards), C.c_int32(k_in), C.c_int32(k_out),
                                 _p(os_), _p(oi))
    return rc, os_, oi


def algorithmic_bytes(counters, d, emb_bytes, n_enter, k_out=200):
    """BASELINE.md section 4 / SURVEY.md 8(d): bytes one query must move.
    counters: [..., 3, 5] (frontier, gathered, scored)."""
    c = np.asarray(counters, dtype=np.int64)
    F, G, S = c[..., 0, :], c[..., 1, :], c[..., 2, :]
    per_round = S * d * emb_bytes + G * 4 + F * 16 + G * 8
    return per_round.sum(axis=-1) + n_enter * 4 + k_out * 12


# ---- f2: the reference's scorer model (attention over the user sequence + DNN); PARITY UNPINNED
class AttnModelStruct(C.Structure):
    _fields_ = ([("d", C.c_int), ("E", C.c_int), ("L", C.c_int), ("emb_dtype", C.c_int)] +
                [(n, C.c_void_p) for n in ("wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2")] +
                [("h", C.c_int * 3), ("w", C.c_void_p * 4), ("b", C.c_void_p * 3),
                 ("bn_scale", C.c_void_p * 3), ("bn_shift", C.c_void_p * 3), ("alpha", C.c_void_p * 3)])


class AttnModel:
    """weights: dict with wq1,bq1,aq,wq2,bq2,wk1,bk1,ak,wk2,bk2 and lists w[4], b[3], bn_scale[3],
    bn_shift[3], alpha[3] (f32 arrays; nann_amd.synth.make_attn_weights makes a seeded set)."""

    def __init__(self, d, E, L, emb_dtype, weights):
        s = AttnModelStruct()
        s.d, s.E, s.L, s.emb_dtype = d, E, L, emb_dtype
        self._keep = []

        def hold(a):
            a = _c(a, np.float32)
            self._keep.append(a)
            return a.ctypes.data

        for n in ("wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2"):
            setattr(s, n, hold(weights[n]))
        for i in range(3):
            s.h[i] = int(np.asarray(weights["w"][i]).shape[1])
            s.b[i], s.bn_scale[i] = hold(weights["b"][i]), hold(weights["bn_scale"][i])
            s.bn_shift[i], s.alpha[i] = hold(weights["bn_shift"][i]), hold(weights["alpha"][i])
        for i in range(4):
            s.w[i] = hold(weights["w"][i])
        self.s = s


def attn_score_rows(model, user_seq, rows):
    """user_seq f32[L, E]; rows [n, d] in the model's emb dtype -> (status, logits f32[n])"""
    u = _c(user_seq, np.float32)
    rows = np.ascontiguousarray(rows)
    n = rows.shape[0]
    kproj = np.zeros((model.s.L, 4 * model.s.E), np.float32)
    rc = lib().oracle_attn_prepare(C.byref(model.s), _p(u), _p(kproj))
    if rc:
        return rc, np.zeros(0, np.float32)
    out = np.zeros(max(n, 1), np.float32)
    rc = lib().oracle_attn_score_rows(C.byref(model.s), _p(u), _p(kproj), _p(rows), C.c_int64(n), _p(out))
    return rc, out[:n]


def tolerant_parity(got_idx, got_scores, exp_idx, exp_scores, rtol=1e-5):
    """Tie-aware comparison of one query's sorted top-k lists when scores are only equal within a
    tolerance (split-f16 MLP): identical ids, or -- where a near-tie flipped an order or a boundary --
    the two sorted score lists agree element by element within rtol * max(1, |score|) and the ids that
    differ sit within that tolerance of the last kept score.  Returns "exact" | "near-tie" | "diverged"."""
    if (got_idx == exp_idx).all():
        ok = np.abs(got_scores - exp_scores) <= rtol * np.maximum(1.0, np.abs(exp_scores))
        return "exact" if ok.all() else "diverged"
    tol = rtol * np.maximum(1.0, np.abs(exp_scores))
    if not (np.abs(got_scores - exp_scores) <= tol).all():
        return "diverged"
    if set(got_idx.tolist()) == set(exp_idx.tolist()):
        return "near-tie"
    kth = exp_scores[-1]
    odd = set(got_idx.tolist()) ^ set(exp_idx.tolist())
    pos = {int(i): float(s) for i, s in zip(exp_idx, exp_scores)}
    pos.update({int(i): float(s) for i, s in zip(got_idx, got_scores)})
    return "near-tie" if all(abs(pos[i] - kth) <= 2 * rtol * max(1.0, abs(kth)) for i in odd) else "diverged"
