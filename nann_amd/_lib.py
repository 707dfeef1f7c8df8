"""ctypes view of libnann_hip.so (the C ABI in include/nann_hip.h).

There is no fallback: if the HIP extension is missing this module raises, and
every op in nann_amd goes through it.
"""
import ctypes as C
import os

from . import build as _build

_LIB = None

OK = 0
STATUS_NAMES = {
    0: "OK", 1: "INVALID_RAGGED_PARAMS", 2: "INVALID_RAGGED_INDICES", 3: "INVALID_RAGGED_INPUT",
    4: "TOPK_K_GT_N", 5: "INDEX_OUT_OF_RANGE", 6: "EMPTY_SCORE_BATCH", 7: "BAD_ARGUMENT",
    8: "TOPK_SCALAR_INPUT", 100: "HIP", 101: "NO_DEVICE", 102: "UNSUPPORTED", 103: "CAPACITY",
    104: "IO", 105: "DTYPE_MISMATCH", 106: "SHAPE_MISMATCH",
}
F16, BF16, F32, I32, I64, F64 = 0, 1, 2, 3, 4, 5
SCORER_L2, SCORER_MLP = 0, 1
MLP_DEFAULT, MLP_SPLIT_F16, MLP_EXACT_F32 = 0, 1, 2
NUM_ROUNDS = 5
NUM_PHASES = 19
PHASE_NAMES = ("zero", "walk", "expand", "score", "topk", "other", "tk_load", "tk_search",
               "tk_collect", "tk_sort", "ex_pass1", "ex_loop", "ex_walkbusy",
               "ex_lookup", "ex_load", "ex_insert", "ex_bar_a", "ex_check", "ex_rank")

# every symbol include/nann_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "nann_abi_version", "nann_last_error", "nann_device_count", "nann_malloc", "nann_free",
    "nann_memcpy", "nann_stream_synchronize", "nann_stream_create", "nann_stream_destroy", "nann_host_malloc",
    "nann_host_free", "nann_huge_const_load", "nann_group_gather_count",
    "nann_group_gather_fill", "nann_group_gather_unique_scratch_bytes", "nann_group_gather_unique", "nann_bitmap_ref_difference", "nann_bloom_filter_difference", "nann_gather_rows", "nann_topk",
    "nann_scorer_create", "nann_scorer_destroy", "nann_user_seq_mean", "nann_score",
    "nann_index_create", "nann_index_destroy", "nann_index_info", "nann_index_probe_info", "nann_search_workspace_bytes",
    "nann_search", "nann_search_v", "nann_search_ex", "nann_search_opt", "nann_search_options_init", "nann_search_reruns", "nann_search_model_opt", "nann_set_traversal_mode", "nann_set_search_reserve", "nann_search_model_workspace_bytes",
    "nann_search_model", "nann_search_model_v",
    "nann_scorer_prepare", "nann_scorer_release", "nann_scorer_table_bytes", "nann_set_preprojection",
    "nann_model_prepare", "nann_model_release", "nann_model_table_bytes", "nann_search_eval_workspace_bytes", "nann_search_eval", "nann_search_eval_ex", "nann_search_eval_model",
    "nann_merge_topk", "nann_merge_topk_host",
    "nann_attn_scorer_create", "nann_attn_scorer_destroy", "nann_attn_prepare", "nann_attn_score",
    "nann_blaze_options_parse", "nann_model_load", "nann_model_destroy", "nann_model_kind", "nann_model_scorer", "nann_model_workspace_bytes", "nann_model_forward",
    "nann_comm_get_unique_id", "nann_comm_create", "nann_comm_destroy", "nann_comm_ranks", "nann_comm_set_timing", "nann_comm_last_breakdown", "nann_comm_wait", "nann_comm_abort", "nann_sharded_topk_workspace_bytes",
    "nann_sharded_topk", "nann_hnsw_draw_levels", "nann_hnsw_build_device", "nann_hnsw_build_device_ex",
]


class SearchOptions(C.Structure):
    """nann_search_options (include/nann_hip.h): -1 = the process default of the field"""
    _fields_ = [("struct_bytes", C.c_int32), ("traversal_mode", C.c_int32), ("slot_reserve", C.c_int32),
                ("preprojection", C.c_int32), ("mlp_form", C.c_int32)]


class SearchPlan(C.Structure):
    """nann_search_plan: what the planner chose for a call"""
    _fields_ = [("visited_set", C.c_int32), ("fallback_visited_set", C.c_int32), ("threads", C.c_int32),
                ("workgroups", C.c_int32), ("phased", C.c_int32), ("table", C.c_int32),
                ("est_visited", C.c_float), ("worst_visited", C.c_float)]


class ScorerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("emb_dtype", C.c_int32),
                ("h1", C.c_int32), ("h2", C.c_int32),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("alpha1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("alpha2", C.c_void_p),
                ("w3", C.c_void_p), ("precision", C.c_int32)]


class AttnDesc(C.Structure):
    _fields_ = ([("d", C.c_int32), ("emb_dtype", C.c_int32), ("seq_len", C.c_int32)] +
                [(n, C.c_void_p) for n in ("wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2")] +
                [("w", C.c_void_p * 4), ("b", C.c_void_p * 3), ("bn_scale", C.c_void_p * 3),
                 ("bn_shift", C.c_void_p * 3), ("alpha", C.c_void_p * 3), ("precision", C.c_int32)])


class IndexDesc(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("d", C.c_int32), ("emb_dtype", C.c_int32),
                ("item_embs", C.c_void_p), ("item_ids", C.c_void_p),
                ("nb_values", C.c_void_p * 2), ("nb_row_splits", C.c_void_p * 2),
                ("nb_nnz", C.c_int64 * 2),
                ("enter_points", C.c_void_p), ("n_enter", C.c_int64),
                ("on_device", C.c_int32)]


def lib_path():
    # NANN_HIP_LIB: load another build of the same ABI (kernel-variant experiments, tools/)
    return os.environ.get("NANN_HIP_LIB") or _build.LIB


def lib():
    """Load (building first if the sources are newer) the HIP extension."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        # the default library is rebuilt when its sources changed since it was linked (content
        # hash, nann_amd/build.py); a library named by NANN_HIP_LIB is loaded as it is
        if not os.path.exists(path) or (not os.environ.get("NANN_HIP_LIB") and _build.is_stale()):
            try:
                _build.build()
            except Exception as e:  # no hipcc and no prebuilt library: fail loudly
                raise ImportError(
                    f"nann_amd: HIP extension {path} is missing and could not be built ({e}); "
                    "there is no CPU fallback") from e
        L = C.CDLL(path)
        L.nann_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            getattr(L, name)  # AttributeError if the ABI is incomplete
        L.nann_scorer_destroy.restype = None
        L.nann_attn_scorer_destroy.restype = None
        L.nann_index_destroy.restype = None
        L.nann_comm_destroy.restype = None
        L.nann_model_destroy.restype = None
        _LIB = L
    return _LIB


def last_error():
    return lib().nann_last_error().decode("utf-8", "replace")
