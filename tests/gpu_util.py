"""Shared helpers of the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np
import torch


def require_gpu():
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    from nann_amd import _lib
    assert _lib.lib().nann_device_count() > 0, "libnann_hip.so sees no HIP device"


def cuda(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


import contextlib


@contextlib.contextmanager
def traversal_mode(mode):
    """Run a block with the visited set forced to one residence (lds_hash / lds_bitmap / hbm_bitmap)."""
    from nann_amd import retrieval
    retrieval.set_traversal_mode(mode)
    try:
        yield
    finally:
        retrieval.set_traversal_mode("auto")


MODES = ["lds_hash", "lds_hash32", "lds_bitmap", "hbm_bitmap"]

_CACHE = {}


def synth_index(n, d, ef, seed=1234, noise=1.0, n_clusters=64, mode="hnsw"):
    """(graph dict, oracle Index, device Index) -- cached per test session."""
    from nann_amd import retrieval, synth
    from oracle import oracle as O
    key = (n, d, ef, seed, noise, n_clusters, mode)
    if key not in _CACHE:
        g = synth.make_index(n, d, ef=ef, seed=seed, noise=noise, n_clusters=n_clusters, mode=mode,
                             device="cuda")
        oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        dix = retrieval.Index.from_dict(g)
        _CACHE[key] = (g, oix, dix)
    return _CACHE[key]


def queries_for(g, nq, seed=4321):
    from nann_amd import synth
    return synth.make_queries(g["item_embs"], g["assign"], nq, seed=seed)


def tolerant_parity(*args, **kw):
    from oracle import oracle as O
    return O.tolerant_parity(*args, **kw)
