"""Item-id sharding across the GPUs of one node (BASELINE configs 4-5).

One process per GPU.  Rank g holds shard g of the corpus -- its own item rows, item ids
and an independently built graph over them -- and runs the full traversal for EVERY
query.  The only exchange step is one all-gather of the per-shard top-k lists
(k x (f32 score + i64 id) = 2.4 KB per query per shard at k = 200) followed by a merge
with TopKV2's order over the shard-major concatenation: score descending, ties -> lower
shard, then lower local rank.

torch.distributed is the transport (backend "nccl" is RCCL over xGMI on the MI355X box,
"gloo" in the CPU tests); the merge is nann_merge_topk (device) or
nann_merge_topk_host (the host-side merge north_star names).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from ._lib import lib
from .ops import _check, _ptr, _stream


def merge_host(scores, ids, k_out):
    """scores f32[nq, shards, k], ids i64[nq, shards, k] (numpy) -> ([nq,k_out], [nq,k_out])."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    nq, shards, k = scores.shape
    out_s = np.empty((nq, k_out), np.float32)
    out_i = np.empty((nq, k_out), np.int64)
    _check(lib().nann_merge_topk_host(
        scores.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.c_int64(nq),
        C.c_int32(shards), C.c_int32(k), C.c_int32(k_out), out_s.ctypes.data_as(C.c_void_p),
        out_i.ctypes.data_as(C.c_void_p)), "merge")
    return out_s, out_i


def merge_device(scores, ids, k_out):
    """CUDA tensors, same shapes as merge_host."""
    nq, shards, k = scores.shape
    out_s = torch.empty((nq, k_out), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((nq, k_out), dtype=torch.int64, device=scores.device)
    _check(lib().nann_merge_topk(_ptr(scores), _ptr(ids), C.c_int64(nq), C.c_int32(shards), C.c_int32(k),
                                 C.c_int32(k_out), _ptr(out_s), _ptr(out_i), _stream()), "merge")
    return out_s, out_i


def all_gather_topk(scores, ids, world, group=None):
    """[nq, k] per rank -> [nq, world, k] on every rank (shard-major per query)."""
    nq, k = scores.shape
    gs = torch.empty((world * nq, k), dtype=scores.dtype, device=scores.device)
    gi = torch.empty((world * nq, k), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)  # rank-major concatenation
    dist.all_gather_into_tensor(gi, ids.contiguous(), group=group)
    return (gs.view(world, nq, k).permute(1, 0, 2).contiguous(),
            gi.view(world, nq, k).permute(1, 0, 2).contiguous())


class ShardedSearch:
    """Exchange + merge for one rank of a sharded search."""

    def __init__(self, index, scorer, level_topn, world, merge="device", group=None):
        self.world, self.k, self.merge_kind, self.group = world, int(level_topn[5]), merge, group

    def merge(self, result):
        """result: this rank's retrieval.SearchResult.  A query that failed on a shard
        contributes -inf scores (its slots are never selected while another shard has
        real candidates)."""
        scores = result.scores
        bad = (result.status != 0)[:, None]
        scores = torch.where(bad, torch.full_like(scores, float("-inf")), scores)
        gs, gi = all_gather_topk(scores, result.item_ids, self.world, self.group)
        if self.merge_kind == "host":
            s, i = merge_host(gs.cpu().numpy(), gi.cpu().numpy(), self.k)
            return torch.as_tensor(i), torch.as_tensor(s)
        s, i = merge_device(gs, gi, self.k)
        return i, s
