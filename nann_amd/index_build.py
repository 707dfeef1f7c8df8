"""HNSW index construction + export in the reference's on-disk layout.

Mirrors NANN_impls/nann/delivery/build_hnsw_index.py: `build_and_save_index(embeddings,
start_level, num_neighbors, output_dir)` writes `enter_points.npy`,
`neighbors_level_{l}_values.npy` (int64, -1 slots dropped) and
`neighbors_level_{l}_row_splits.npy` (int64[N+1], a row for every item, empty when the node is
absent at that level).  The graph itself comes from nann_amd/csrc/host/hnsw_build.cpp (C++,
multi-threaded) instead of Faiss, which is not available; its raw arrays have Faiss' shapes
(`levels`, `offsets`, `neighbors`, `cum_nneighbor_per_level`) so the export code below is the
reference's, vectorised.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "host", "hnsw_build.cpp")
# every source of libnann_host.so (the HNSW builder + the CPU hooks onto the frozen-GraphDef reader and the
# pre-projected tables' cache) and what they include
_SRCS = [_SRC, os.path.join(_HERE, "csrc", "host", "nann_graphdef_c.cpp"),
         os.path.join(_HERE, "csrc", "host", "nann_projcache_c.cpp")]
_DEPS = _SRCS + [os.path.join(_HERE, "csrc", "host", "nann_graphdef.h"), os.path.join(_HERE, "csrc", "host", "nann_graphdef_text.h"),
                 os.path.join(_HERE, "csrc", "host", "nann_blaze_options.h"), os.path.join(_HERE, "csrc", "host", "nann_npy.h"), os.path.join(_HERE, "csrc", "host", "nann_projcache.h")]
_LIB_PATH = os.path.join(_HERE, "_build", "libnann_host.so")
_LIB = None


def build_host_lib(force=False, sanitize=False):
    """g++ -O3 the host-side builder into nann_amd/_build/libnann_host.so.  sanitize: the same sources under AddressSanitizer +
    UBSan into libnann_host_asan.so -- what the parser fuzzers of the CPU suite load (tests/fuzz/; the process needs
    LD_PRELOAD=libasan.so)."""
    path = _LIB_PATH.replace(".so", "_asan.so") if sanitize else _LIB_PATH
    if force or not os.path.exists(path) or max(os.path.getmtime(p) for p in _DEPS) > os.path.getmtime(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        flags = (["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
                 if sanitize else ["-O3"])
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-pthread", "-mavx2", "-mfma", "-Wno-invalid-offsetof",
                                                 "-o", path] + _SRCS)
    return path


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build_host_lib())
    return _LIB


def build_hnsw(embeddings, num_neighbors=32, ef_construction=40, seed=0, n_threads=None):
    """-> dict(levels i32[N], offsets i64[N+1], neighbors i32[slots], cum_nneighbor_per_level),
    the arrays build_hnsw_index.py:36-39 reads from faiss' `index.hnsw`."""
    x = np.ascontiguousarray(embeddings, dtype=np.float32)
    n, d = x.shape
    if n_threads is None:
        n_threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1
    levels = np.zeros(n, np.int32)
    offsets = np.zeros(n + 1, np.int64)
    cum = np.zeros(64, np.int32)
    n_slots, max_levels = C.c_int64(0), C.c_int32(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    args = [p(x), C.c_int64(n), C.c_int32(d), C.c_int32(num_neighbors), C.c_int32(ef_construction),
            C.c_uint64(seed), C.c_int32(n_threads), p(levels), p(offsets)]
    rc = _lib().nann_hnsw_build(*args, None, C.byref(n_slots), p(cum), C.byref(max_levels))
    if rc:
        raise ValueError(f"nann_hnsw_build: bad argument ({rc})")
    neighbors = np.empty(n_slots.value, np.int32)
    rc = _lib().nann_hnsw_build(*args, p(neighbors), C.byref(n_slots), p(cum), C.byref(max_levels))
    if rc:
        raise ValueError(f"nann_hnsw_build: bad argument ({rc})")
    return {"levels": levels, "offsets": offsets, "neighbors": neighbors,
            "cum_nneighbor_per_level": cum[: max_levels.value + 1].copy()}


def export_levels(raw, start_level=2):
    """build_hnsw_index.py:41-66 on the raw arrays: enter points = nodes with
    `levels > start_level`; per level below it a CSR over ALL items."""
    levels, offsets, neighbors, cum = (raw["levels"], raw["offsets"], raw["neighbors"],
                                       raw["cum_nneighbor_per_level"])
    n = len(levels)
    enter_points = np.nonzero(levels > start_level)[0]                      # :45
    nb_values, nb_row_splits = [], []
    for level in range(start_level):                                         # :49
        width = int(cum[level + 1] - cum[level])
        present = levels > level                                             # :53 `level >= levels[idx]` -> empty
        slots = offsets[:-1, None] + int(cum[level]) + np.arange(width)[None, :]
        vals = neighbors[np.minimum(slots, len(neighbors) - 1)]
        keep = present[:, None] & (vals >= 0)                                # :59 drop -1 slots
        row_len = keep.sum(1)
        rs = np.zeros(n + 1, np.int64)
        np.cumsum(row_len, out=rs[1:])
        nb_values.append(vals[keep].astype(np.int64))                        # :66 int64 on disk
        nb_row_splits.append(rs)
    return {"enter_points": enter_points, "nb_values": nb_values, "nb_row_splits": nb_row_splits}


def build_and_save_index(embeddings, start_level, num_neighbors, output_dir, seed=0, n_threads=None):
    """Same name and arguments as the reference's function (build_hnsw_index.py:33)."""
    raw = build_hnsw(embeddings, num_neighbors=num_neighbors, seed=seed, n_threads=n_threads)
    ex = export_levels(raw, start_level)
    os.makedirs(output_dir, exist_ok=True)
    np.save(os.path.join(output_dir, "enter_points.npy"), ex["enter_points"])
    for level in range(start_level):
        np.save(os.path.join(output_dir, f"neighbors_level_{level}_values.npy"), ex["nb_values"][level])
        np.save(os.path.join(output_dir, f"neighbors_level_{level}_row_splits.npy"), ex["nb_row_splits"][level])
    return raw, ex


# ---- construction on the device (csrc/nann_hnsw_build.hip) -----------------------------------------------------
def build_hnsw_gpu(item_embs, num_neighbors=32, ef_construction=40, seed=0, start_level=2, want_raw=False,
                   keep_pruned=False):
    """HNSW(M) over the rows of `item_embs` (CUDA tensor f16 | bf16 [N, d], or a numpy f16 array) built ON THE GPU
    (nann_hnsw_build_device).  Returns the export of build_hnsw_index.py:41-66 -- {"enter_points", "nb_values"
    [start_level], "nb_row_splits"[start_level], "levels"} as numpy arrays (values int64 on disk) -- assembled with
    torch on the device; with want_raw also the Faiss-shaped raw arrays of build_hnsw().  keep_pruned: the selection
    heuristic's keepPrunedConnections switch (off in Faiss, hence in the reference's graphs): rows fill up to their cap --
    the dense-graph family (mean level-0 degree ~55 of 64 instead of ~17)."""
    import torch
    from . import _lib
    from .ops import _check, _ptr, _stream, _DT
    L = _lib.lib()
    x = item_embs if isinstance(item_embs, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(item_embs))
    x = x.cuda().contiguous()
    n, d = x.shape
    m = int(num_neighbors)
    levels = np.zeros(n, np.int32)
    n_up = C.c_int64(0)
    _check(L.nann_hnsw_draw_levels(C.c_int64(n), C.c_int32(m), C.c_uint64(seed), levels.ctypes.data_as(C.c_void_p),
                                   C.byref(n_up)), "hnsw levels")
    adj0 = torch.empty((n, 2 * m), dtype=torch.int32, device=x.device)
    up_row = torch.empty(n, dtype=torch.int32, device=x.device)
    adj_up = torch.empty((max(n_up.value, 1), m), dtype=torch.int32, device=x.device)
    torch.cuda.synchronize()
    _check(L.nann_hnsw_build_device_ex(_ptr(x), C.c_int64(n), C.c_int32(d), C.c_int32(_DT[x.dtype]), C.c_int32(m),
                                       C.c_int32(ef_construction), C.c_int32(1 if keep_pruned else 0),
                                       levels.ctypes.data_as(C.c_void_p), _ptr(adj0), _ptr(up_row), _ptr(adj_up), _stream()),
           "hnsw build")
    lev = torch.as_tensor(levels, device=x.device)
    out = {"levels": levels, "enter_points": np.nonzero(levels > start_level)[0],  # build_hnsw_index.py:45
           "nb_values": [], "nb_row_splits": []}
    for level in range(start_level):                                              # :49
        if level == 0:
            rows = adj0
        else:  # row of node i on level l >= 1: up_row[i] + l - 1 (absent: an empty row, :53)
            rows = torch.full((n, m), -1, dtype=torch.int32, device=x.device)
            has = lev > level
            rows[has] = adj_up[(up_row[has] + (level - 1)).long()]
        keep = rows >= 0                                                          # :59 drop the -1 slots
        rs = torch.zeros(n + 1, dtype=torch.int64, device=x.device)
        torch.cumsum(keep.sum(1), 0, out=rs[1:])
        out["nb_values"].append(rows[keep].to(torch.int64).cpu().numpy())         # :66 int64 on disk
        out["nb_row_splits"].append(rs.cpu().numpy())
    if want_raw:
        cum = np.concatenate([[0], 2 * m + m * np.arange(int(levels.max()))]).astype(np.int32)
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(2 * m + (levels.astype(np.int64) - 1) * m, out=offsets[1:])
        nb = np.full(int(offsets[-1]), -1, np.int32)
        a0, au, ur = adj0.cpu().numpy(), adj_up.cpu().numpy(), up_row.cpu().numpy()
        nb[(offsets[:-1, None] + np.arange(2 * m)[None, :]).ravel()] = a0.ravel()
        for i in np.nonzero(levels > 1)[0]:
            for l in range(1, levels[i]):
                nb[offsets[i] + cum[l]: offsets[i] + cum[l] + m] = au[ur[i] + l - 1]
        out["raw"] = {"levels": levels, "offsets": offsets, "neighbors": nb, "cum_nneighbor_per_level": cum}
    return out
