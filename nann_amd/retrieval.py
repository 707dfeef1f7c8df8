"""The traversal schedule of the reference's serving graph on the MI355X.

`build_model()` in NANN_impls/nann/delivery/build_opt_graph.py:69-149 chains
GroupGather / BitmapRefDifference / GatherV2 / BlazeXlaOp / TopKV2 ~40 times
per request.  Two equivalent executions of that schedule live here:

  * `search()`        one fused launch for a whole batch of queries
                      (nann_search in include/nann_hip.h) -- the product path;
  * `search_per_op()` the same schedule spelled op by op with the drop-in ops
                      of nann_amd.ops, line for line against build_model(), so
                      that each op is exercised in the composition the
                      reference uses it in (plumbing/parity, not performance).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib
from .ops import _check, _ptr, _stream


class Index:
    """Device-resident corpus + graph: what the serving graph's HugeConst nodes
    hold (build_opt_graph.py:83-90) plus the baked enter_points Const (:70)."""

    def __init__(self, item_embs, item_ids, nb_values, nb_row_splits, enter_points, device=None):
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")

        def put(a, dt):
            if isinstance(a, torch.Tensor):
                return a.to(device=dev, dtype=dt).contiguous()
            return torch.as_tensor(np.ascontiguousarray(a)).to(device=dev, dtype=dt).contiguous()

        if isinstance(item_embs, np.ndarray) and item_embs.dtype == np.uint16:  # bf16 bit patterns
            self.item_embs = torch.as_tensor(item_embs.view(np.int16)).to(dev).view(torch.bfloat16)
        elif isinstance(item_embs, torch.Tensor):
            self.item_embs = item_embs.to(dev).contiguous()
        else:
            self.item_embs = torch.as_tensor(np.ascontiguousarray(item_embs)).to(dev)
        self.item_ids = put(item_ids, torch.int64)
        self.nb_values = [put(nb_values[l], torch.int32) for l in (0, 1)]
        self.nb_row_splits = [put(nb_row_splits[l], torch.int64) for l in (0, 1)]
        self.enter_points = put(enter_points, torch.int32)
        self.n_items, self.d = self.item_embs.shape
        desc = _lib.IndexDesc()
        desc.n_items, desc.d = self.n_items, self.d
        desc.emb_dtype = ops._DT[self.item_embs.dtype]
        desc.item_embs = self.item_embs.data_ptr()
        desc.item_ids = self.item_ids.data_ptr()
        for l in (0, 1):
            desc.nb_values[l] = self.nb_values[l].data_ptr()
            desc.nb_row_splits[l] = self.nb_row_splits[l].data_ptr()
            desc.nb_nnz[l] = self.nb_values[l].numel()
        desc.enter_points = self.enter_points.data_ptr()
        desc.n_enter = self.enter_points.numel()
        desc.on_device = 1
        self.handle = C.c_void_p(0)
        with torch.cuda.device(dev):
            _check(lib().nann_index_create(C.byref(desc), C.byref(self.handle)), "index")
        info = (C.c_int64 * 6)()
        _check(lib().nann_index_info(self.handle, info))
        self.max_deg = (int(info[3]), int(info[4]))
        pr = (C.c_float * 5)()
        _check(lib().nann_index_probe_info(self.handle, pr))
        # the probe launch of nann_index_create: new level-0 nodes per frontier row on this graph (what the planner sizes
        # a query's visited set with); None for toy indices / corpora whose probe requests all failed
        self.probe = ({"queries": int(pr[0]), "ef": int(pr[1]), "new_per_row_mean": round(float(pr[2]), 3),
                       "new_per_row_q90": round(float(pr[3]), 3), "new_per_row_max": round(float(pr[4]), 3)} if pr[0] > 0 else None)
        self.bitmap_words = int(math.ceil(self.n_items / 32))  # build_opt_graph.py:114
        self.device = dev
        self._ws = None

    @classmethod
    def from_files(cls, index_dir, item_embs_dir, start_level=2):
        """Load the reference's on-disk artefacts straight into HBM through HugeConst, with the
        dtype casts build_model() requests (build_opt_graph.py:70,83-90): `item_embs.npy` -> f16,
        `item_ids.npy` i64, `neighbors_level_{l}_values.npy` -> i32, `..._row_splits.npy` i64,
        `enter_points.npy` -> i32 (files as written by build_hnsw_index.py:41-67)."""
        import os
        assert start_level == 2, "the serving graph walks levels 1 and 0 below the entry layer"
        hc = {
            "embs": ops.huge_const(os.path.join(item_embs_dir, "item_embs.npy"), np.float16),
            "ids": ops.huge_const(os.path.join(item_embs_dir, "item_ids.npy"), np.int64),
            "ep": ops.huge_const(os.path.join(index_dir, "enter_points.npy"), np.int32),
        }
        for l in (0, 1):
            hc[f"v{l}"] = ops.huge_const(os.path.join(index_dir, f"neighbors_level_{l}_values.npy"), np.int32)
            hc[f"rs{l}"] = ops.huge_const(os.path.join(index_dir, f"neighbors_level_{l}_row_splits.npy"), np.int64)
        ix = cls(hc["embs"].tensor, hc["ids"].tensor, [hc["v0"].tensor, hc["v1"].tensor],
                 [hc["rs0"].tensor, hc["rs1"].tensor], hc["ep"].tensor)
        ix._huge_consts = hc  # the HugeConst objects own the HBM the tensors view
        return ix

    @classmethod
    def from_dict(cls, g, device=None):
        return cls(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"],
                   device=device)

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "handle", None) and self.handle.value:
                lib().nann_index_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass

    def workspace(self, level_topn, n_queries):
        t = (C.c_int32 * 6)(*[int(x) for x in level_topn])
        nbytes = C.c_int64(0)
        _check(lib().nann_search_workspace_bytes(self.handle, t, C.c_int64(n_queries), C.byref(nbytes)))
        # a fresh buffer per call (torch's caching allocator makes it cheap and stream-ordered):
        # concurrent searches on one Index from several threads/streams must not share scratch
        return torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=self.device)


TRAVERSAL_MODES = {"auto": 0, "lds_bitmap": 1, "hbm_bitmap": 2, "lds_hash": 3, "lds_hash32": 4}


def set_traversal_mode(mode="auto"):
    """Where nann_search keeps a query's visited set (include/nann_hip.h, nann_traversal_mode):
    "auto" | "lds_bitmap" | "hbm_bitmap" | "lds_hash" | "lds_hash32".  Results are identical in every mode; the
    knob exists for the parity tests and for tuning.  The process default of search_options(traversal=...)."""
    _check(lib().nann_set_traversal_mode(C.c_int32(TRAVERSAL_MODES[mode])), "traversal mode")


MLP_FORMS = {"auto": 0, "fused": 1, "phased": 2}


def search_options(traversal=None, slot_reserve=None, preprojection=None, mlp_form=None):
    """nann_search_options for one call; None = the process default of the field (nann_set_* / built-in)."""
    o = _lib.SearchOptions()
    lib().nann_search_options_init(C.byref(o))
    if traversal is not None:
        o.traversal_mode = TRAVERSAL_MODES[traversal]
    if slot_reserve is not None:
        o.slot_reserve = int(slot_reserve)
    if preprojection is not None:
        o.preprojection = 1 if preprojection else 0
    if mlp_form is not None:
        o.mlp_form = MLP_FORMS[mlp_form]
    return o


class SearchResult:
    __slots__ = ("item_ids", "scores", "index", "status", "counters", "phase_ticks", "plan", "_ws", "slot_reserve")

    def __init__(self, item_ids, scores, index, status, counters, phase_ticks=None, plan=None, ws=None, slot_reserve=None):
        self.item_ids, self.scores, self.index, self.status, self.counters = (
            item_ids, scores, index, status, counters)
        self.phase_ticks = phase_ticks
        self.plan = plan  # dict: what the planner chose (nann_search_plan)
        self._ws = ws
        self.slot_reserve = slot_reserve  # nann_search_options.slot_reserve of the call (None: no options were passed)

    def reruns(self):
        """queries of this call that the hash-set kernel handed back to the bitmap kernel (synchronises the stream;
        read it before the next search on the same index reuses the workspace)"""
        n = C.c_int64(0)
        _check(lib().nann_search_reruns(_ptr(self._ws), C.byref(n), _stream()), "reruns")
        return n.value


def _level_topn_args(level_topn, b, dev):
    """level_topn as the C ABI takes it: uniform i32[6] -> (maxima, NULL); per query [B, 6] (the reference feeds
    `level_topn` per request, build_opt_graph.py:75,151-159) -> (column maxima [host], device i32[B, 6])."""
    lt = np.asarray(level_topn.cpu() if isinstance(level_topn, torch.Tensor) else level_topn, dtype=np.int64)
    if lt.ndim == 1:
        assert lt.shape[0] == 6
        return (C.c_int32 * 6)(*[int(x) for x in lt]), None, int(lt[5])
    assert lt.shape == (b, 6), "per-query level_topn: [n_queries, 6]"
    mx = np.maximum(lt.max(axis=0), 0)
    tq = torch.as_tensor(lt.astype(np.int32)).to(dev).contiguous()
    return (C.c_int32 * 6)(*[int(x) for x in mx]), tq, int(mx[5])


_VIS_NAMES = {v: k for k, v in TRAVERSAL_MODES.items()}


def _plan_dict(p):
    return {"visited_set": _VIS_NAMES.get(p.visited_set, p.visited_set), "fallback_visited_set": _VIS_NAMES.get(p.fallback_visited_set),
            "threads": p.threads, "workgroups": p.workgroups, "phased": bool(p.phased), "table": bool(p.table),
            "est_visited": round(float(p.est_visited), 1), "worst_visited": round(float(p.worst_visited), 1)}


def search(index, scorer, q, level_topn, want_counters=True, want_phase_ticks=False, options=None):
    """Fused execution of build_model()'s schedule for a batch of queries.
    q: f32[B, d] CUDA tensor (ops.user_seq_mean of `comm_seq`).  level_topn: i32[6] for the whole batch, or
    [B, 6] per query (nann_search_v; rows of the outputs are level_topn.max(0)[5] wide, zero behind a query's
    own k).  options: search_options(...) for this call (nann_search_opt).  Asynchronous: the returned tensors are valid
    once the current stream reaches them.  status[b] != 0 marks a request the reference would have failed."""
    q = q.to(device=index.device, dtype=torch.float32).contiguous()
    b = q.shape[0]
    dev = index.device
    t, tq, k = _level_topn_args(level_topn, b, dev)
    out_ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    out_scores = torch.empty((b, k), dtype=torch.float32, device=dev)
    out_index = torch.empty((b, k), dtype=torch.int32, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    counters = torch.zeros((b, 3, _lib.NUM_ROUNDS), dtype=torch.int32, device=dev) if want_counters else None
    ticks = (torch.zeros((b, _lib.NUM_PHASES), dtype=torch.int64, device=dev)
             if want_phase_ticks else None)
    ws = index.workspace(list(t), b)
    plan = _lib.SearchPlan()
    assert not (want_phase_ticks and tq is not None), "phase ticks: uniform level_topn only"
    with torch.cuda.device(dev):
        _check(lib().nann_search_opt(index.handle, scorer.handle, _ptr(q), C.c_int64(b), t, _ptr(tq), _ptr(ws),
                                     C.c_int64(ws.numel()), _ptr(out_ids), _ptr(out_scores), _ptr(out_index),
                                     _ptr(status), _ptr(counters), _ptr(ticks),
                                     C.byref(options) if options is not None else None, C.byref(plan), _stream()), "search")
    return SearchResult(out_ids, out_scores, out_index, status, counters, ticks, _plan_dict(plan), ws,
                        slot_reserve=int(options.slot_reserve) if options is not None else None)


def search_model(index, model, comm_seq, level_topn, want_counters=True, options=None):
    """The serving signature (build_opt_graph.py:151-159) for a batch: comm_seq f16[B, seq_len, E] +
    level_topn (i32[6], or [B, 6] per request) -> SearchResult, scored by `model` (ops.Model: l2 / mlp / the
    reference's attention + DNN model -- the per-user projection runs once per request, then the fused traversal;
    nann_search_model_opt)."""
    seq = comm_seq.to(device=index.device, dtype=torch.float16).contiguous()
    b = seq.shape[0]
    dev = index.device
    t, tq, k = _level_topn_args(level_topn, b, dev)
    out_ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    out_scores = torch.empty((b, k), dtype=torch.float32, device=dev)
    out_index = torch.empty((b, k), dtype=torch.int32, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    counters = torch.zeros((b, 3, _lib.NUM_ROUNDS), dtype=torch.int32, device=dev) if want_counters else None
    nbytes = C.c_int64(0)
    _check(lib().nann_search_model_workspace_bytes(index.handle, model.handle, t, C.c_int64(b), C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=dev)
    plan = _lib.SearchPlan()
    with torch.cuda.device(dev):
        _check(lib().nann_search_model_opt(index.handle, model.handle, _ptr(seq), C.c_int64(b), t, _ptr(tq), _ptr(ws),
                                           C.c_int64(ws.numel()), _ptr(out_ids), _ptr(out_scores), _ptr(out_index),
                                           _ptr(status), _ptr(counters),
                                           C.byref(options) if options is not None else None, C.byref(plan), _stream()), "search")
    return SearchResult(out_ids, out_scores, out_index, status, counters, None, _plan_dict(plan), ws,
                        slot_reserve=int(options.slot_reserve) if options is not None else None)


def prepare(index, scorer):
    """Build the pre-projected table of (scorer | model, index) now and pin it (nann_scorer_prepare /
    nann_model_prepare): no request pays for it.  Returns (table_bytes, resident_bytes)."""
    is_model = isinstance(scorer, ops.Model)
    L = lib()
    with torch.cuda.device(index.device):
        _check((L.nann_model_prepare if is_model else L.nann_scorer_prepare)(scorer.handle, index.handle, _stream()),
               "prepare")
    return table_bytes(index, scorer)


def release(index, scorer):
    is_model = isinstance(scorer, ops.Model)
    L = lib()
    _check((L.nann_model_release if is_model else L.nann_scorer_release)(scorer.handle, index.handle), "release")


def table_bytes(index, scorer):
    is_model = isinstance(scorer, ops.Model)
    L = lib()
    tb, rb = C.c_int64(0), C.c_int64(0)
    _check((L.nann_model_table_bytes if is_model else L.nann_scorer_table_bytes)(
        scorer.handle, index.handle if index is not None else None, C.byref(tb), C.byref(rb)), "table_bytes")
    return tb.value, rb.value


class EvalResult:
    """Outputs of search_eval: rows [b, :n_out[b]] are valid (the rest is zero)."""

    def __init__(self, item_ids, scores, index, n_out, status):
        self.item_ids, self.scores, self.index, self.n_out, self.status = item_ids, scores, index, n_out, status


def search_eval(index, scorer, q, num_scoring=(3, 1, 1), top_k_per_level=(400, 200, 100), topk_eval=200, want_counters=False):
    """Model.retrieval() (model.py:299-362) for a batch of users in ONE kernel (nann_search_eval /
    nann_search_eval_model).  `scorer`: ops.Scorer with q f32[B, d], or ops.Model with q = comm_seq
    f16[B, seq_len, E].  Same results as search_eval_per_op, user by user."""
    is_model = isinstance(scorer, ops.Model)
    dev = index.device
    q = q.to(device=dev, dtype=torch.float16 if is_model else torch.float32).contiguous()
    b, k = q.shape[0], int(topk_eval)
    ns = (C.c_int32 * 3)(*[int(x) for x in num_scoring])
    tk = (C.c_int32 * 3)(*[int(x) for x in top_k_per_level])
    out_ids = torch.empty((b, k), dtype=torch.int64, device=dev)
    out_scores = torch.empty((b, k), dtype=torch.float32, device=dev)
    out_index = torch.empty((b, k), dtype=torch.int32, device=dev)
    n_out = torch.empty(b, dtype=torch.int32, device=dev)
    status = torch.empty(b, dtype=torch.int32, device=dev)
    nbytes = C.c_int64(0)
    _check(lib().nann_search_eval_workspace_bytes(index.handle, scorer.handle if is_model else None, C.c_int64(b),
                                                  C.byref(nbytes)))
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=dev)
    counters = None
    with torch.cuda.device(dev):
        if want_counters and not is_model:  # F, G, S per user (nann_search_eval_ex)
            counters = torch.zeros((b, 3), dtype=torch.int32, device=dev)
            _check(lib().nann_search_eval_ex(index.handle, scorer.handle, _ptr(q), C.c_int64(b), ns, tk, C.c_int32(k), _ptr(ws),
                                             C.c_int64(ws.numel()), _ptr(out_ids), _ptr(out_scores), _ptr(out_index),
                                             _ptr(n_out), _ptr(status), _ptr(counters), _stream()), "search_eval")
        else:
            fn = lib().nann_search_eval_model if is_model else lib().nann_search_eval
            _check(fn(index.handle, scorer.handle, _ptr(q), C.c_int64(b), ns, tk, C.c_int32(k), _ptr(ws),
                      C.c_int64(ws.numel()), _ptr(out_ids), _ptr(out_scores), _ptr(out_index), _ptr(n_out), _ptr(status),
                      _stream()), "search_eval")
    res = EvalResult(out_ids, out_scores, out_index, n_out, status)
    res.counters = counters
    return res


# -----------------------------------------------------------------------------
# build_model() spelled with the per-op drop-ins (one query)
def _fake_row_splits(x):
    """build_opt_graph.py:29-30"""
    return torch.tensor([0, x.numel()], dtype=torch.int64, device=x.device)


def _set_difference(a, flags):
    """build_opt_graph.py:33-36"""
    values, _, flags = ops.bitmap_ref_difference(a, _fake_row_splits(a), flags)
    return values, flags


def _ragged_gather(values, row_splits, idx):
    """build_opt_graph.py:39-49"""
    out, _ = ops.group_gather(values, row_splits, idx.to(torch.int64), _fake_row_splits(idx), unique=False)
    return out


def _top_k(ids, scores, k):
    """build_opt_graph.py:52-66"""
    scores, indices = ops.top_k(scores, k)
    return ops.gather(ids, indices), scores


def search_per_op(index, scorer, q, level_topn):
    """One query through the op-by-op schedule (build_opt_graph.py:109-149).
    Raises the error the reference graph would raise.  Returns
    (item_ids i64[k], scores f32[k], internal index i32[k])."""
    q = q.reshape(-1)
    t = [int(x) for x in level_topn]

    def forward(idx):  # :91-107
        if idx.numel() == 0:
            raise ops.InternalError(6, "Error when getting input address or size")
        s = ops.blaze_score(scorer, q, table=index.item_embs, indices=idx)
        if idx.numel() == 1:  # tf.squeeze -> scalar; TopKV2/ConcatV2 reject it
            raise ops.InvalidArgumentError(8, "input must be >= 1-D, got shape []")
        return s

    enter_points = index.enter_points
    # level 2
    scores = forward(enter_points)                                          # :111
    idx_results, scores_result = _top_k(enter_points, scores, t[0])         # :112
    # level 1
    idx_next = _ragged_gather(index.nb_values[1], index.nb_row_splits[1], idx_results)   # :116
    flags = torch.zeros(index.bitmap_words, dtype=torch.int32, device=index.device)      # :115-118
    idx_results, _ = _set_difference(idx_results, flags)                    # :119-120
    idx_next, _ = _set_difference(idx_next, flags)                          # :121-122
    scores_next = forward(idx_next)                                         # :124
    idx_result, scores_result = _top_k(torch.cat([idx_results, idx_next]),
                                       torch.cat([scores_result, scores_next]), t[1])    # :125-127
    # level 0
    idx_candidate = idx_result
    flags.zero_()                                                           # :131
    idx_candidate, _ = _set_difference(idx_candidate, flags)                # :132-133
    for i in range(3):                                                      # :135
        idx_next = _ragged_gather(index.nb_values[0], index.nb_row_splits[0], idx_candidate)   # :136
        idx_next, _ = _set_difference(idx_next, flags)                      # :137
        scores_next = forward(idx_next)                                     # :138
        idx_candidate, scores_candidate = _top_k(idx_next, scores_next, t[i + 2])   # :139
        idx_result = torch.cat([idx_result, idx_candidate])                 # :140
        scores_result = torch.cat([scores_result, scores_candidate])        # :141
    idx_result, scores_result = _top_k(idx_result, scores_result, t[5])     # :143
    item_ids = ops.gather(index.item_ids.view(torch.int32).reshape(-1, 2), idx_result)   # :144
    return item_ids.reshape(-1).view(torch.int64), scores_result, idx_result


# -----------------------------------------------------------------------------
# f3: the eval-graph traversal (Model.retrieval / search_level, model.py:299-362) spelled with
# the same drop-in ops.  It differs from the serving schedule in its frontier rule (new nodes
# that score at least the worst kept result), in the min(k, n) guard of its top_k (model.py:268)
# and in visiting neighbours as an ascending set (tf.unique + tf.sets, :316-321).
def search_eval_per_op(index, scorer, q, num_scoring=(3, 1, 1), top_k_per_level=(400, 200, 100),
                       topk_eval=200, backend=None, stats=None):
    """One query through Model.retrieval().  num_scoring / top_k_per_level are indexed by level
    (0, 1, start level 2), as config.py:50-58 lists them.  Returns
    (item_ids i64[<=topk_eval], scores f32, internal index i32).  `backend`: the module providing
    group_gather / bitmap_ref_difference / blaze_score / top_k / gather (default: nann_amd.ops, the
    HIP kernels; the CPU tests pass a stand-in to check this host logic against the oracle).  `stats`: a dict that
    receives F / G / S (rows walked, neighbours gathered, rows scored incl. the enter points: the fused kernel's counters)."""
    B = ops if backend is None else backend
    assert int(num_scoring[2]) == 1                                          # model.py:347
    n_f = n_g = 0
    n_s = int(index.enter_points.numel())
    q = q.reshape(-1)

    def get_scores(idx):                                                     # :240-262
        if idx.numel() == 0:  # plain TF scoring of an empty batch: an empty tensor, the level loop goes on
            return torch.empty(0, dtype=torch.float32, device=idx.device)   # (only the serving graph's BlazeXlaOp fails here)
        return B.blaze_score(scorer, q, table=index.item_embs, indices=idx)

    def top_k(ids, scores, k):                                               # :264-283, k = min(k, n)
        k = min(int(k), int(ids.numel()))
        scores, indices = B.top_k(scores, k)
        return B.gather(ids, indices), scores

    results = index.enter_points
    scores = get_scores(results)                                             # :350-351
    results, scores = top_k(results, scores, top_k_per_level[2])             # :353
    for level in (1, 0):                                                     # :355-356
        flags = torch.zeros(index.bitmap_words, dtype=torch.int32, device=index.enter_points.device)
        idx_result, scores_result = results, scores
        _, _, flags = B.bitmap_ref_difference(results, _fake_row_splits(results), flags)   # visited = idx_ep (:311)
        idx_candidate = results
        for _ in range(int(num_scoring[level])):
            nxt, _ = B.group_gather(index.nb_values[level], index.nb_row_splits[level],
                                    idx_candidate.to(torch.int64), _fake_row_splits(idx_candidate),
                                    unique=False)                            # :316
            n_f += int(idx_candidate.numel())
            n_g += int(nxt.numel())
            nxt, _, flags = B.bitmap_ref_difference(nxt, _fake_row_splits(nxt), flags)   # unique, minus visited, visited |= (:317-321)
            n_s += int(nxt.numel())
            idx_next = torch.sort(nxt).values                                # tf.sets results are ascending
            scores_next = get_scores(idx_next)                               # :323
            idx_result, scores_result = top_k(torch.cat([idx_result, idx_next]),
                                              torch.cat([scores_result, scores_next]),
                                              top_k_per_level[level])        # :326-328
            mask = scores_next >= scores_result[-1]                          # :330
            idx_candidate = idx_next[mask]                                   # :331
        results, scores = idx_result, scores_result
    if stats is not None:
        stats.update(F=n_f, G=n_g, S=n_s)
    results, scores = results[:topk_eval], scores[:topk_eval]                # :358
    item_ids = B.gather(index.item_ids.view(torch.int32).reshape(-1, 2), results)   # :360
    return item_ids.reshape(-1).view(torch.int64), scores, results
