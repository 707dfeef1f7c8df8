"""Host-side mirror of the reference's custom-op surface for the retrieval path.

Same op names, argument order/meaning and error behaviour as the TensorFlow
wrappers the reference generates from its REGISTER_OP blocks
(`tf.group_gather`, `tf.bitmap_ref_difference`, `tf.huge_const`,
`tf.blaze_xla_op`; tensorflow/python/user_ops/user_ops.py) plus the two stock
ops the serving graph chains between them (`tf.gather`, `tf.math.top_k`).
Every function is a thin call into the C ABI (include/nann_hip.h) on torch
CUDA tensors -- torch only provides device memory and the stream.

Citations are relative to /root/reference/,
UO/ = tensorflow/tensorflow/core/user_ops/.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, last_error


# ---- errors: the classes the reference raises through tf.errors --------------
class NannError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"[{_lib.STATUS_NAMES.get(status, status)}] {message}")
        self.status = status


class InvalidArgumentError(NannError):
    """errors::InvalidArgument (OP_REQUIRES in the reference kernels)."""


class InternalError(NannError):
    """errors::Internal"""


class NotFoundError(NannError):
    """errors::NotFound (huge_const_op.cc:96-98)"""


class UnimplementedError(NannError):
    """errors::Unimplemented"""


_INVALID = {1, 2, 3, 4, 5, 7, 8}


def _raise(status, what=""):
    msg = last_error() or what
    if status in _INVALID:
        raise InvalidArgumentError(status, msg)
    if status == 104:
        raise NotFoundError(status, msg)
    if status == 102:
        raise UnimplementedError(status, msg)
    raise InternalError(status, msg)


def _check(status, what=""):
    if status != _lib.OK:
        _raise(status, what)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(t, dtype):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t), dtype=dtype)
    if not t.is_cuda:
        t = t.cuda()
    return t.to(dtype).contiguous()


# ---- a1: GroupGather ----------------------------------------------------------
def group_gather(params_values, params_row_splits, indices_values, indices_row_splits, unique=False):
    """tf.group_gather (UO/beam_search_op/GroupGather_kernel.cc:18-42): for each
    group of `indices`, concatenate the CSR rows params[i] in order, duplicates
    kept.  Returns (ret_values int32, ret_row_splits int64).

    unique=True is the reference's per-group unordered_set path (:91-131): a group's
    row holds its distinct values.  The reference writes them in the set's iteration
    order (implementation-defined: any order is its answer); this op emits them in
    first-occurrence order.  The serving graph never sets it (build_opt_graph.py:48)."""
    pv = _dev(params_values, torch.int32)
    prs = _dev(params_row_splits, torch.int64)
    iv = _dev(indices_values, torch.int64)
    irs = _dev(indices_row_splits, torch.int64)
    ret_rs = torch.zeros(max(irs.numel(), 1), dtype=torch.int64, device=pv.device)
    offsets = torch.empty(iv.numel() + 1, dtype=torch.int64, device=pv.device)
    n_ret, n_rs, code = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    st = lib().nann_group_gather_count(
        _ptr(prs), C.c_int64(prs.numel()), C.c_int64(pv.numel()), _ptr(iv), C.c_int64(iv.numel()),
        _ptr(irs), C.c_int64(irs.numel()), _ptr(ret_rs), _ptr(offsets), C.byref(n_ret),
        C.byref(n_rs), C.byref(code), _stream())
    _check(st, "GroupGather")
    ret_values = torch.empty(n_ret.value, dtype=torch.int32, device=pv.device)
    if n_ret.value:
        st = lib().nann_group_gather_fill(_ptr(pv), _ptr(prs), _ptr(iv), C.c_int64(iv.numel()),
                                          _ptr(offsets), _ptr(ret_values), _stream())
        _check(st, "GroupGather")
    ret_rs = ret_rs[: n_rs.value]
    if not unique or n_rs.value <= 1:
        return ret_values, ret_rs
    nbytes = C.c_int64(0)
    _check(lib().nann_group_gather_unique_scratch_bytes(C.c_int64(n_ret.value), C.c_int64(n_rs.value), C.byref(nbytes)), "GroupGather")
    scratch = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=pv.device)
    out = torch.empty(max(n_ret.value, 1), dtype=torch.int32, device=pv.device)
    out_rs = torch.empty(n_rs.value, dtype=torch.int64, device=pv.device)
    n_out = C.c_int64(0)
    st = lib().nann_group_gather_unique(_ptr(ret_values), C.c_int64(n_ret.value), _ptr(ret_rs), C.c_int64(n_rs.value),
                                        _ptr(scratch), _ptr(out), _ptr(out_rs), C.byref(n_out), _stream())
    _check(st, "GroupGather")
    return out[: n_out.value], out_rs


# ---- a2: BitmapRefDifference ---------------------------------------------------
def bitmap_ref_difference(idx_next_values, idx_next_row_splits, idx_flag):
    """tf.bitmap_ref_difference (UO/bitmap_op/bitmap_ops.cc:150-257): ordered
    first-occurrence filter of ids whose bit is clear; `idx_flag` (int32 CUDA
    tensor, ceil(N/32) words) is the Ref input and is mutated IN PLACE.
    Returns (c_values, c_row_splits, idx_flag)."""
    if not (isinstance(idx_flag, torch.Tensor) and idx_flag.is_cuda and idx_flag.dtype == torch.int32
            and idx_flag.is_contiguous()):
        raise InvalidArgumentError(7, "idx_flag must be a contiguous int32 CUDA tensor (Ref input)")
    v = _dev(idx_next_values, torch.int32)
    rs = _dev(idx_next_row_splits, torch.int64)
    out = torch.empty(max(v.numel(), 1), dtype=torch.int32, device=v.device)
    out_rs = torch.zeros(max(rs.numel(), 1), dtype=torch.int64, device=v.device)
    n_out, n_rs, code = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    st = lib().nann_bitmap_ref_difference(
        _ptr(v), C.c_int64(v.numel()), _ptr(rs), C.c_int64(rs.numel()), _ptr(idx_flag),
        C.c_int64(idx_flag.numel()), _ptr(out), _ptr(out_rs), C.byref(n_out), C.byref(n_rs),
        C.byref(code), _stream())
    _check(st, "BitmapRefDifference")
    return out[: n_out.value], out_rs[: n_rs.value], idx_flag


# ---- a8: sibling ops the reference registers next to these (not wired into the serving
# graph), expressed with the kernels above ------------------------------
def bitmap_init(idx, length):
    """tf.bitmap_init (UO/bitmap_op/bitmap_ops.cc:28-75): int32[length] bitmap with the bits
    of `idx` set."""
    idx = _dev(idx, torch.int32)
    if length < 0 or idx.numel() > length:  # the reference's own check (:56-57)
        raise InvalidArgumentError(7, f"require: length >= idx.size() and length >=0 but length:{length}"
                                      f"idx.size():{idx.numel()}")
    flags = torch.zeros(max(length, 0), dtype=torch.int32, device=idx.device)
    if idx.numel():
        bitmap_ref_difference(idx, [0, idx.numel()], flags)
    return flags


def bitmap_difference(idx_next, idx_flag):
    """tf.bitmap_difference (bitmap_ops.cc:83-143): value-semantics variant of
    bitmap_ref_difference over one flat list; returns (idx_next_new, idx_flag_new) and
    leaves `idx_flag` untouched."""
    idx = _dev(idx_next, torch.int32)
    flags = _dev(idx_flag, torch.int32).clone()
    if idx.numel() == 0:
        return idx, flags
    out, _, flags = bitmap_ref_difference(idx, [0, idx.numel()], flags)
    return out, flags


def bloom_filter_difference(idx_next_values, idx_next_row_splits, idx_flag, bucket=0, bucket_size=1):
    """tf.bloom_filter_difference (UO/bitmap_op/bitmap_ops.cc:264-425): BitmapRefDifference's approximate
    sibling -- four Fingerprint64-derived positions per node in a 32 * bucket_size-bit filter; a node is kept
    iff one of them was clear.  `idx_flag` (int32 CUDA tensor) is the Ref input, mutated IN PLACE."""
    if not (isinstance(idx_flag, torch.Tensor) and idx_flag.is_cuda and idx_flag.dtype == torch.int32
            and idx_flag.is_contiguous()):
        raise InvalidArgumentError(7, "idx_flag must be a contiguous int32 CUDA tensor (Ref input)")
    v = _dev(idx_next_values, torch.int32)
    rs = _dev(idx_next_row_splits, torch.int64)
    out = torch.empty(max(v.numel(), 1), dtype=torch.int32, device=v.device)
    out_rs = torch.zeros(max(rs.numel(), 1), dtype=torch.int64, device=v.device)
    n_out, n_rs, code = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    st = lib().nann_bloom_filter_difference(
        _ptr(v), C.c_int64(v.numel()), _ptr(rs), C.c_int64(rs.numel()), _ptr(idx_flag), C.c_int64(idx_flag.numel()),
        C.c_int64(bucket), C.c_int64(bucket_size), _ptr(out), _ptr(out_rs), C.byref(n_out), C.byref(n_rs),
        C.byref(code), _stream())
    _check(st, "BloomFilterDifference")
    return out[: n_out.value], out_rs[: n_rs.value], idx_flag


def blaze_top_k(values, k):
    """tf.blaze_top_k (UO/topk_op/BlazeTopK_kernel.cc:13-101): the k largest values of each row, sorted by
    value, and their int32 indices.  The reference finds them by a sampled threshold + std::partial_sort and
    leaves the order of EQUAL values unspecified; TopKV2's answer (ties -> lower index) is one of its
    answers, so the same kernel serves.  0 <= k <= input_len is required (:47-48)."""
    v = _dev(values, torch.float32)
    if k < 0 or k > v.shape[-1]:
        raise InvalidArgumentError(7, f"require: 0 <= k <= input_len, but{k} > {v.shape[-1]}")
    return top_k(v, k)


def batch_top_k_on_rt(values, row_splits, k, ascending=False):
    """tf.batch_top_k_on_rt (UO/topk_op/BatchTopKOnRT_kernel.cc:24-155): per ragged row the
    min(k, len) best values, row-local int64 indices and the output row_splits.  Equal values
    keep their input order (the reference's std::partial_sort_copy leaves it unspecified)."""
    v = _dev(values, torch.float32)
    rs = torch.as_tensor(np.asarray(row_splits.cpu() if isinstance(row_splits, torch.Tensor) else row_splits),
                         dtype=torch.int64)
    if rs.numel() == 0 or int(rs[0]) != 0 or int(rs[-1]) != v.numel():
        code = 1 if rs.numel() == 0 else (2 if int(rs[0]) != 0 else 3)
        raise InvalidArgumentError(3, f"Invalid RaggedTensor input, code: {code}")
    groups = rs.numel() - 1
    ks = [int(k)] * groups if np.ndim(k) == 0 else [int(x) for x in k]
    if len(ks) != groups:
        raise InvalidArgumentError(7, f"Size of k vector does NOT match number of groups: {len(ks)}!={groups}")
    vals, idxs, out_rs = [], [], [0]
    for g in range(groups):
        s, e = int(rs[g]), int(rs[g + 1])
        kk = max(0, min(ks[g], e - s))
        if kk:
            row = v[s:e]
            tv, ti = top_k(-row if ascending else row, kk)
            vals.append(row[ti.long()])
            idxs.append(ti.to(torch.int64))
        out_rs.append(out_rs[-1] + kk)
    cat = lambda xs, dt: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=v.device)
    return cat(vals, torch.float32), cat(idxs, torch.int64), torch.tensor(out_rs, dtype=torch.int64)


# ---- a3: GatherV2 ---------------------------------------------------------------
def gather(params, indices):
    """tf.gather axis 0 (core/kernels/gather_op.cc, gather_functor.h:38-116)."""
    assert params.is_cuda and params.is_contiguous()
    idx = _dev(indices, torch.int32)
    row_bytes = params[0].numel() * params.element_size() if params.dim() > 1 else params.element_size()
    if row_bytes % 4:
        raise UnimplementedError(102, "row size must be a multiple of 4 bytes")
    out = torch.empty((idx.numel(),) + tuple(params.shape[1:]), dtype=params.dtype, device=params.device)
    bad = C.c_int64(-1)
    st = lib().nann_gather_rows(_ptr(params), C.c_int64(params.shape[0]), C.c_int64(row_bytes), _ptr(idx),
                                C.c_int64(idx.numel()), _ptr(out), C.byref(bad), _stream())
    _check(st, "GatherV2")
    return out


# ---- a5: TopKV2 -------------------------------------------------------------------
def top_k(values, k):
    """tf.math.top_k(sorted=True) (core/kernels/topk_op.cc:40-205): returns
    (values, indices int32); descending, ties -> lower index."""
    v = _dev(values, torch.float32)
    squeeze = v.dim() == 1
    if v.dim() == 0:
        raise InvalidArgumentError(8, "input must be >= 1-D, got shape []")  # topk_op.cc:63-65
    v2 = v.reshape(-1, v.shape[-1])
    ov = torch.empty((v2.shape[0], max(k, 0)), dtype=torch.float32, device=v.device)
    oi = torch.empty((v2.shape[0], max(k, 0)), dtype=torch.int32, device=v.device)
    st = lib().nann_topk(_ptr(v2), C.c_int64(v2.shape[0]), C.c_int64(v2.shape[1]), C.c_int32(k), _ptr(ov),
                         _ptr(oi), _stream())
    _check(st, "TopKV2")
    shape = tuple(v.shape[:-1]) + (k,)
    return (ov.reshape(shape), oi.reshape(shape)) if not squeeze else (ov[0], oi[0])


# ---- a4: the scorer behind BlazeXlaOp ----------------------------------------------
_DT = {torch.float16: _lib.F16, torch.bfloat16: _lib.BF16, torch.float32: _lib.F32}


_PRECISION = {"split": _lib.MLP_SPLIT_F16, "exact": _lib.MLP_EXACT_F32}


def _precision_code(precision):
    """'split' | 'exact' -> nann_mlp_precision; anything else is an error (a typo must not silently select
    the slower kernel)."""
    try:
        return _PRECISION[precision]
    except KeyError:
        raise ValueError(f"precision must be 'split' or 'exact', got {precision!r}") from None


class Scorer:
    """What the frozen scoring GraphDef is to BlazeXlaOp (blaze_xla_kernel.cc:24-33):
    kind 'l2' (s = -||q - x||^2) or 'mlp' (weights dict as synth.make_mlp_weights)."""

    def __init__(self, kind, d, emb_dtype=torch.float16, weights=None, precision="split"):
        """precision (mlp): "split" (the default: north_star's contract is 1e-5) = split-f16 operands on the
        16-bit MFMA, scores within 1e-5 of the fp32 chain; "exact" = f32-input MFMA, scores bit-identical to
        the oracle, ~3x slower (nann_mlp_precision)."""
        if kind not in ("l2", "mlp"):
            raise ValueError(f"scorer kind must be 'l2' or 'mlp', got {kind!r}")
        self.kind, self.d, self.emb_dtype, self.precision = kind, d, emb_dtype, precision
        desc = _lib.ScorerDesc()
        desc.precision = _precision_code(precision)
        desc.kind = _lib.SCORER_L2 if kind == "l2" else _lib.SCORER_MLP
        desc.d = d
        desc.emb_dtype = _DT[emb_dtype]
        self._keep = {}
        if kind == "mlp":
            for name in ("w1", "b1", "alpha1", "w2", "b2", "alpha2", "w3"):
                a = np.ascontiguousarray(weights[name], dtype=np.float32)
                self._keep[name] = a
                setattr(desc, name, a.ctypes.data)
            desc.h1, desc.h2 = self._keep["w1"].shape[1], self._keep["w2"].shape[1]
        self.handle = C.c_void_p(0)
        _check(lib().nann_scorer_create(C.byref(desc), C.byref(self.handle)), "scorer")

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "handle", None) and self.handle.value:
                lib().nann_scorer_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass


class AttnScorer:
    """The reference's own scorer model behind the BlazeXlaOp contract (SURVEY.md 8 f2;
    model.py:189-233, model_util.py:70-97): weights dict as synth.make_attn_weights.
    precision: "split" (default; split-f16 operands on the 16-bit MFMA, nann_attn_split.h) | "exact" (f32-input MFMA)."""

    def __init__(self, d, seq_len, emb_dtype=torch.float16, weights=None, precision="split"):
        self.d, self.seq_len, self.emb_dtype, self.precision = d, seq_len, emb_dtype, precision
        desc = _lib.AttnDesc()
        desc.d, desc.seq_len, desc.emb_dtype = d, seq_len, _DT[emb_dtype]
        desc.precision = _precision_code(precision)
        self._keep = []

        def hold(a):
            a = np.ascontiguousarray(a, dtype=np.float32)
            self._keep.append(a)
            return a.ctypes.data

        for name in ("wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2"):
            setattr(desc, name, hold(weights[name]))
        for i in range(4):
            desc.w[i] = hold(weights["w"][i])
        for i in range(3):
            desc.b[i], desc.bn_scale[i] = hold(weights["b"][i]), hold(weights["bn_scale"][i])
            desc.bn_shift[i], desc.alpha[i] = hold(weights["bn_shift"][i]), hold(weights["alpha"][i])
        self.handle = C.c_void_p(0)
        _check(lib().nann_attn_scorer_create(C.byref(desc), C.byref(self.handle)), "attention scorer")

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "handle", None) and self.handle.value:
                lib().nann_attn_scorer_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass

    def prepare(self, comm_seq):
        """comm_seq f16[B, L, 64] (the `comm_seq` feed reshaped) -> per-user (kt f32[B,256,64],
        upad f32[B,64,64]), the user side of forward() computed once per request."""
        u = _dev(comm_seq, torch.float16).reshape(-1, self.seq_len, 64)
        kt = torch.empty((u.shape[0], 256, 64), dtype=torch.float32, device=u.device)
        upad = torch.empty((u.shape[0], 64, 64), dtype=torch.float32, device=u.device)
        _check(lib().nann_attn_prepare(self.handle, _ptr(u), C.c_int64(u.shape[0]), _ptr(kt), _ptr(upad),
                                       _stream()), "attention prepare")
        return kt, upad

    def score(self, kt, upad, item_emb=None, table=None, indices=None):
        """f32 logits of the candidate rows for ONE user (kt[b], upad[b]); arguments as blaze_score."""
        bad = C.c_int64(-1)
        if item_emb is not None:
            rows, n_table, idx, n = item_emb.contiguous(), item_emb.shape[0], None, item_emb.shape[0]
        else:
            rows, n_table = table, table.shape[0]
            idx = _dev(indices, torch.int32)
            n = idx.numel()
        out = torch.empty(max(n, 1), dtype=torch.float32, device=kt.device)
        st = lib().nann_attn_score(self.handle, _ptr(kt), _ptr(upad), _ptr(rows), C.c_int64(n_table), _ptr(idx),
                                   C.c_int64(n), _ptr(out), C.byref(bad), _stream())
        _check(st, "BlazeXlaOp")
        return out[:n]


def save_scorer_dir(path, kind, weights=None, precision=None):
    """Write a scoring model as the weights directory BlazeXlaOp's `graph_def` attr names on this
    build (include/nann_hip.h, nann_model_load): scorer.txt + one .npy per weight tensor.
    kind: "l2" | "mlp" (dict as synth.make_mlp_weights) | "attention" (dict as synth.make_attn_weights)."""
    import os
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "scorer.txt"), "w") as f:
        f.write(kind + "\n")
    if precision is not None:  # mlp / attention: "exact" | "split" (precision.txt, read by nann_model_load)
        with open(os.path.join(path, "precision.txt"), "w") as f:
            f.write(precision + "\n")
    if kind == "mlp":
        for name in ("w1", "b1", "alpha1", "w2", "b2", "alpha2", "w3"):
            np.save(os.path.join(path, name + ".npy"), np.asarray(weights[name], np.float32))
    elif kind == "attention":
        for name in ("wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2"):
            np.save(os.path.join(path, name + ".npy"), np.asarray(weights[name], np.float32))
        for i in range(4):
            np.save(os.path.join(path, f"w{i}.npy"), np.asarray(weights["w"][i], np.float32))
        for i in range(3):
            for key, stem in (("b", "b"), ("bn_scale", "bn_scale"), ("bn_shift", "bn_shift"), ("alpha", "alpha")):
                np.save(os.path.join(path, f"{stem}{i}.npy"), np.asarray(weights[key][i], np.float32))


class Model:
    """The scoring model a BlazeXlaOp node names (nann_model_load): `path` is what the op's `graph_def` attr
    holds -- the reference's frozen GraphDef FILE (convert_meta.py:361-398; blaze_xla_kernel.cc:156-180), read
    without TensorFlow, or a weights DIRECTORY (save_scorer_dir) for the l2 / mlp scorers."""

    def __init__(self, path, d, seq_len=50, emb_dtype=torch.float16):
        self.d, self.seq_len = d, seq_len
        self.handle = C.c_void_p(0)
        _check(lib().nann_model_load(path.encode(), C.c_int32(d), C.c_int32(_DT[emb_dtype]), C.c_int32(seq_len),
                                     C.byref(self.handle)), "BlazeXlaOp model")
        self.kind = ("l2", "mlp", "attention")[lib().nann_model_kind(self.handle)]
        nb = C.c_int64(0)
        _check(lib().nann_model_workspace_bytes(self.handle, C.byref(nb)))
        self._ws_bytes = nb.value

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "handle", None) and self.handle.value:
                lib().nann_model_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass

    def forward(self, user_seq_emb, item_emb):
        """forward() of build_opt_graph.py:91-107: user_seq_emb f16 [1, L, E], item_emb [n, d] -> logits f32 [n, 1]
        (model.py:226-227)."""
        u = _dev(user_seq_emb, torch.float16)
        rows = item_emb.contiguous()
        n = rows.shape[0]
        out = torch.empty(max(n, 1), dtype=torch.float32, device=u.device)
        ws = torch.empty(self._ws_bytes, dtype=torch.uint8, device=u.device)
        _check(lib().nann_model_forward(self.handle, _ptr(u), _ptr(rows), C.c_int64(n), _ptr(out), _ptr(ws),
                                        _stream()), "BlazeXlaOp")
        return out[:n].reshape(-1, 1)


def user_seq_mean(comm_seq):
    """comm_seq f16[B, L, d] (the `comm_seq` feed reshaped, build_opt_graph.py:76-79)
    -> q f32[B, d]: mean of the non-pad history rows."""
    s = _dev(comm_seq, torch.float16)
    b, l, d = s.shape
    q = torch.empty((b, d), dtype=torch.float32, device=s.device)
    _check(lib().nann_user_seq_mean(_ptr(s), C.c_int64(b), C.c_int32(l), C.c_int32(d), _ptr(q), _stream()))
    return q


def blaze_score(scorer, q, item_emb=None, table=None, indices=None):
    """The BlazeXlaOp contract for one user (forward(), build_opt_graph.py:91-107):
    f32 logits, one per candidate row, rows scored independently.
    Either item_emb [n, d] (already gathered, as BlazeXlaOp receives it) or
    table + indices (fused GatherV2 + score)."""
    q = _dev(q, torch.float32).reshape(-1)
    bad = C.c_int64(-1)
    if item_emb is not None:
        rows, n_table, idx, n = item_emb.contiguous(), item_emb.shape[0], None, item_emb.shape[0]
    else:
        rows, n_table = table, table.shape[0]
        idx = _dev(indices, torch.int32)
        n = idx.numel()
    out = torch.empty(max(n, 1), dtype=torch.float32, device=q.device)
    st = lib().nann_score(scorer.handle, _ptr(q), _ptr(rows), C.c_int64(n_table), _ptr(idx), C.c_int64(n),
                          _ptr(out), C.byref(bad), _stream())
    _check(st, "BlazeXlaOp")
    return out[:n]


# ---- a6: HugeConst ---------------------------------------------------------------------
_NPY = {np.dtype(np.float16): (_lib.F16, torch.float16), np.dtype(np.float32): (_lib.F32, torch.float32),
        np.dtype(np.float64): (_lib.F64, torch.float64), np.dtype(np.int32): (_lib.I32, torch.int32),
        np.dtype(np.int64): (_lib.I64, torch.int64)}


class HugeConst:
    """tf.huge_const (UO/huge_const_op/huge_const_op.cc:58-226): a .npy file
    loaded into HBM once and held for the lifetime of the object.
    dtype/shape play the role of the op's attrs and are validated against the
    npy header (:108-147).  `.tensor` is a zero-copy torch view."""

    def __init__(self, path, dtype, shape, allow_cast=False):
        code, tdt = _NPY[np.dtype(dtype)]
        shp = (C.c_int64 * len(shape))(*shape)
        ptr, nbytes = C.c_void_p(0), C.c_int64(0)
        _check(lib().nann_huge_const_load(path.encode(), C.c_int(code), shp, C.c_int(len(shape)),
                                          C.c_int(1 if allow_cast else 0), C.byref(ptr),
                                          C.byref(nbytes)), "HugeConst")
        self._ptr, self.nbytes = ptr, nbytes.value
        self.shape, self.dtype = tuple(shape), tdt
        self.tensor = _wrap_device_pointer(ptr.value, self.shape, tdt, self)

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "_ptr", None) and self._ptr.value:
                lib().nann_free(self._ptr)
                self._ptr = C.c_void_p(0)
        except Exception:
            pass


def _wrap_device_pointer(ptr, shape, dtype, owner):
    """torch view of library-owned HBM via __cuda_array_interface__."""
    typestr = {torch.float16: "<f2", torch.float32: "<f4", torch.float64: "<f8", torch.int32: "<i4",
               torch.int64: "<i8"}[dtype]

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                  "version": 3, "strides": None}
    h._owner = owner
    n = int(np.prod(shape)) if len(shape) else 1
    if n == 0:
        return torch.empty(shape, dtype=dtype, device="cuda")
    return torch.as_tensor(h, device="cuda")


def huge_const(path, dtype=None):
    """model_util.huge_constant (NANN_impls/nann/model/model_util.py:107-121):
    the header supplies the shape; a dtype different from the file's is cast
    (np.load(path).astype(dtype)) -- at load time here, WITHOUT the wrapper's
    in-place rewrite of the file."""
    with open(path, "rb") as f:
        np.lib.format.read_magic(f)
        shape, fortran, file_dtype = np.lib.format.read_array_header_1_0(f)
    return HugeConst(path, dtype or file_dtype, shape, allow_cast=dtype is not None)
