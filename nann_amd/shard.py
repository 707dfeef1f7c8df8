"""Item-id sharding across the GPUs of one node (BASELINE configs 4-5).

One process per GPU.  Rank g holds shard g of the corpus -- its own item rows, item ids
and an independently built graph over them -- and runs the full traversal for EVERY
query.  The only exchange step is one all-gather of the per-shard top-k lists
(k x (f32 score + i64 id) = 2.4 KB per query per shard at k = 200) followed by a merge
with TopKV2's order over the shard-major concatenation: score descending, ties -> lower
shard, then lower local rank.

Two transports:
  * "rccl"  -- the product path: the C ABI's own communicator (nann_comm_*, ncclAllGather of
               ONE packed record per rank straight from the library, merge on the device);
               torch.distributed only carries the 128-byte communicator id to the ranks once.
  * "torch" -- torch.distributed collectives ("gloo" in the CPU tests of this logic, where no
               GPU exists) + nann_merge_topk / nann_merge_topk_host.
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
    dev = scores.device
    if scores.is_cuda and dist.get_backend(group) == "gloo":  # gloo gathers host tensors only (dry runs on one GPU)
        scores, ids = scores.cpu(), ids.cpu()
    gs = torch.empty((world * nq, k), dtype=scores.dtype, device=scores.device)
    gi = torch.empty((world * nq, k), dtype=ids.dtype, device=ids.device)
    dist.all_gather_into_tensor(gs, scores.contiguous(), group=group)  # rank-major concatenation
    dist.all_gather_into_tensor(gi, ids.contiguous(), group=group)
    return (gs.view(world, nq, k).permute(1, 0, 2).contiguous().to(dev),
            gi.view(world, nq, k).permute(1, 0, 2).contiguous().to(dev))


def record_bytes(n_queries, k):
    """bytes of one rank's packed record (nann_comm.hip rec_bytes): f32[B, k] scores, then i64[B, k] ids, padded to 256"""
    return (n_queries * k * 12 + 255) & ~255


def pack_record_host(scores, ids, status):
    """Host stand-in of k_pack_record (nann_comm.hip): the record a rank contributes to the all-gather -- scores f32[B, k]
    followed by item ids i64[B, k]; a query that failed on this shard (status != 0) contributes (-inf, 0)."""
    scores = np.ascontiguousarray(scores, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    b, k = scores.shape
    bad = np.asarray(status) != 0 if status is not None else np.zeros(b, bool)
    rec = np.zeros(record_bytes(b, k), np.uint8)
    rec[: b * k * 4] = np.where(bad[:, None], np.float32(-np.inf), scores).astype(np.float32).view(np.uint8).ravel()
    rec[b * k * 4: b * k * 12] = np.where(bad[:, None], 0, ids).astype(np.int64).view(np.uint8).ravel()
    return rec


def merge_records_host(recv, n_queries, k_in, k_out):
    """Host stand-in of k_merge_records: recv uint8[world, record_bytes] (rank-major, as ncclAllGather leaves it) ->
    (scores f32[B, k_out], ids i64[B, k_out]) in TopKV2's order over the shard-major concatenation (score descending,
    ties -> lower shard, then lower local rank), read straight from the records without a transposition."""
    world = recv.shape[0]
    b, k = n_queries, k_in
    sc = np.stack([recv[s, : b * k * 4].view(np.float32).reshape(b, k) for s in range(world)], 1)   # [B, world, k]
    ii = np.stack([recv[s, b * k * 4: b * k * 12].view(np.int64).reshape(b, k) for s in range(world)], 1)
    return merge_host(sc, ii, k_out)


class Comm:
    """nann_comm: the library's own RCCL communicator.  Collective: every rank constructs it.
    The 128-byte id travels from rank 0 through the existing torch.distributed group (any
    backend) -- a C++ host would hand it over by its own means."""

    def __init__(self, world, rank, group=None):
        self.world, self.rank = world, rank
        idb = np.zeros(128, np.uint8)
        if world > 1:
            box = [None]
            if rank == 0:  # a failure here must reach every rank, or the others would wait in the broadcast forever
                st = lib().nann_comm_get_unique_id(idb.ctypes.data_as(C.c_void_p))
                box = [idb.tobytes() if st == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            if box[0] is None:
                raise RuntimeError("nann_comm_get_unique_id failed on rank 0 (RCCL not loadable?)")
            idb = np.frombuffer(box[0], np.uint8).copy()
        self.handle = C.c_void_p(0)
        _check(lib().nann_comm_create(C.c_int32(world), C.c_int32(rank),
                                      idb.ctypes.data_as(C.c_void_p) if world > 1 else None,
                                      C.byref(self.handle)), "comm create")

    @classmethod
    def single_rank_rccl(cls):
        """A REAL one-rank RCCL communicator (ncclGetUniqueId, ncclCommInitRank, and ncclAllGather in
        nann_sharded_topk): the library's whole RCCL binding exercised on a single GPU."""
        self = cls.__new__(cls)
        self.world, self.rank, self.handle = 1, 0, C.c_void_p(0)
        idb = np.zeros(128, np.uint8)
        _check(lib().nann_comm_get_unique_id(idb.ctypes.data_as(C.c_void_p)), "comm id")
        _check(lib().nann_comm_create(C.c_int32(1), C.c_int32(0), idb.ctypes.data_as(C.c_void_p),
                                      C.byref(self.handle)), "comm create")
        return self

    @classmethod
    def loopback(cls, world):
        """Single-process test communicator: every one of `world` shards returns this rank's record."""
        self = cls.__new__(cls)
        self.world, self.rank, self.handle = world, 0, C.c_void_p(0)
        self.is_loopback = True
        _check(lib().nann_comm_create(C.c_int32(world), C.c_int32(0), None, C.byref(self.handle)), "comm create")
        return self

    def ranks(self):
        """(shards, ranks of the RCCL communicator behind this object: ncclCommCount; 0 = none, e.g. a loopback)"""
        w, r = C.c_int32(0), C.c_int32(0)
        _check(lib().nann_comm_ranks(self.handle, C.byref(w), C.byref(r)), "comm ranks")
        return w.value, r.value

    def set_timing(self, enabled=True, loopback_repeat=1, loopback_wait_us=0):
        """HIP events around pack | all-gather | merge of every later exchange (nann_comm_set_timing).  Loopback only:
        loopback_wait_us puts 16 waiting workgroups in front of the stand-in copies (an exchange as long as xGMI's, on one
        GPU), loopback_repeat issues the copies that many times"""
        _check(lib().nann_comm_set_timing(self.handle, C.c_int32(1 if enabled else 0), C.c_int32(loopback_repeat),
                                          C.c_int32(loopback_wait_us)), "comm timing")

    def wait(self, timeout_ms=-1):
        """bounded wait for the last exchange enqueued on this communicator (nann_comm_wait): raises NannError and ABORTS the
        communicator when RCCL reports an asynchronous error or the deadline passes -- a dead rank does not hang the node"""
        _check(lib().nann_comm_wait(self.handle, C.c_int32(int(timeout_ms))), "comm wait")

    def abort(self):
        _check(lib().nann_comm_abort(self.handle), "comm abort")

    def last_breakdown(self):
        """{pack, all_gather, merge} ms of the last exchange (waits for it)"""
        ms = (C.c_float * 3)()
        _check(lib().nann_comm_last_breakdown(self.handle, ms), "comm breakdown")
        return {"pack": round(ms[0], 4), "all_gather": round(ms[1], 4), "merge": round(ms[2], 4)}

    def __del__(self):
        try:  # at interpreter shutdown the module globals may already be gone
            if getattr(self, "handle", None) and self.handle.value:
                lib().nann_comm_destroy(self.handle)
                self.handle = C.c_void_p(0)
        except Exception:
            pass


class ShardedSearch:
    """Exchange + merge for one rank of a sharded search."""
    RESERVED_SLOTS = 32  # workgroup slots kept free for the exchange's kernels when it overlaps the next search (profiles/rd5e_reserve_sweep.txt)

    def __init__(self, level_topn, world, rank=0, merge="device", group=None, transport="torch", comm=None):
        self.world, self.k, self.merge_kind, self.group = world, int(level_topn[5]), merge, group
        self.transport = transport
        self.comm = comm if comm is not None else (Comm(world, rank, group) if transport == "rccl" else None)
        self._ws = None
        self._comm_stream = None
        self._warned_reserve = False

    def search_options(self, overlap=True, reserve=None):
        """nann_search_options for the searches whose exchanges this object overlaps (pass to retrieval.search).  The
        persistent traversal grid owns every CU's LDS until its first workgroups exit; RCCL's kernels need a few
        workgroup slots of their own to run NEXT TO it (measured on one GPU, profiles/r4g_overlap.txt: 16 slots of 512
        cost the search ~3 %, an exchange that only starts when the grid drains costs the whole overlap).  Per call
        since round 5 (the process-wide setter was never restored and taxed unrelated searches)."""
        from . import retrieval
        real = self._real_rccl()
        if reserve is None:
            reserve = self.RESERVED_SLOTS if (overlap and real) else 0
        return retrieval.search_options(slot_reserve=reserve)

    def search(self, index, scorer, q, level_topn, overlap=True, **kw):
        """retrieval.search with this object's options applied (the slot reserve an overlapped exchange needs): the search
        whose result goes to merge(overlap=...)."""
        from . import retrieval
        return retrieval.search(index, scorer, q, level_topn, options=self.search_options(overlap=overlap), **kw)

    def _real_rccl(self):
        return self.transport == "rccl" and self.world > 1 and not getattr(self.comm, "is_loopback", False)

    def _exchange(self, result):
        """pack + ncclAllGather + merge on the CURRENT stream (nann_sharded_topk)"""
        nq, k = result.scores.shape
        dev = result.scores.device
        nbytes = C.c_int64(0)
        _check(lib().nann_sharded_topk_workspace_bytes(C.c_int32(self.world), C.c_int64(nq), C.c_int32(k),
                                                      C.byref(nbytes)))
        if self._ws is None or self._ws.numel() < nbytes.value:
            self._ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=dev)
        out_s = torch.empty((nq, self.k), dtype=torch.float32, device=dev)
        out_i = torch.empty((nq, self.k), dtype=torch.int64, device=dev)
        _check(lib().nann_sharded_topk(self.comm.handle, _ptr(result.scores), _ptr(result.item_ids),
                                       _ptr(result.status), C.c_int64(nq), C.c_int32(k), C.c_int32(self.k),
                                       _ptr(self._ws), C.c_int64(self._ws.numel()), _ptr(out_s), _ptr(out_i),
                                       _stream()), "sharded top-k")
        return out_i, out_s

    def merge(self, result, overlap=False):
        """result: this rank's retrieval.SearchResult -> (item_ids i64[nq,k], scores f32[nq,k]) of
        the whole corpus, identical on every rank.  A query that failed on a shard (status != 0)
        contributes -inf scores / id 0 from that shard: never selected while another shard holds
        real candidates.

        overlap (rccl transport): the exchange of this batch runs on a stream of its own behind the search that
        produced it, so the NEXT batch's search (enqueued on the caller's stream right after this call) overlaps it --
        per step the device then costs max(search, exchange + merge) instead of their sum.  Exchanges still execute
        in call order (one communicator, one stream, one workspace).  The returned tensors belong to that stream:
        call wait() before reading them on the caller's stream."""
        if self.transport == "rccl":
            if not overlap:
                return self._exchange(result)
            if self._real_rccl() and not (getattr(result, "slot_reserve", None) or 0) > 0 and not self._warned_reserve:
                # without reserved workgroup slots RCCL's kernels only start when the persistent traversal grid of
                # the NEXT search drains -- the overlap is lost silently
                import warnings
                warnings.warn("ShardedSearch.merge(overlap=True): the search that produced this result ran without a slot reserve "
                              "(pass sharded.search_options() to retrieval.search, or call sharded.search()); the exchange will "
                              "not overlap the next search", RuntimeWarning, stacklevel=2)
                self._warned_reserve = True
            dev = result.scores.device
            cur = torch.cuda.current_stream(dev)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=dev)
            cs = self._comm_stream
            cs.wait_stream(cur)  # the search that wrote `result`
            with torch.cuda.stream(cs):
                out = self._exchange(result)
            for t in (result.scores, result.item_ids, result.status):
                t.record_stream(cs)  # the caching allocator must not hand these out again before the exchange read them
            return out
        if self.transport == "records":
            # the device path's record layout and merge with host stand-ins for its kernels and a torch.distributed
            # all-gather of the raw bytes (CPU tests of the N-rank flow: tests/test_shard_cpu.py)
            nq, k = result.scores.shape
            rec = torch.as_tensor(pack_record_host(result.scores.cpu().numpy(), result.item_ids.cpu().numpy(),
                                                   result.status.cpu().numpy() if result.status is not None else None))
            recv = torch.empty(self.world * rec.numel(), dtype=torch.uint8)
            dist.all_gather_into_tensor(recv, rec, group=self.group)
            s, i = merge_records_host(recv.view(self.world, rec.numel()).numpy(), nq, k, self.k)
            return torch.as_tensor(i), torch.as_tensor(s)
        scores = result.scores
        bad = (result.status != 0)[:, None]
        scores = torch.where(bad, torch.full_like(scores, float("-inf")), scores)
        ids = torch.where(bad, torch.zeros_like(result.item_ids), result.item_ids)
        gs, gi = all_gather_topk(scores, ids, self.world, self.group)
        if self.merge_kind == "host":
            s, i = merge_host(gs.cpu().numpy(), gi.cpu().numpy(), self.k)
            return torch.as_tensor(i), torch.as_tensor(s)
        s, i = merge_device(gs, gi, self.k)
        return i, s

    def wait(self, timeout_ms=None):
        """make the caller's current stream wait for every overlapped exchange issued so far.  timeout_ms (rccl transport): first
        a BOUNDED host wait for the last exchange (Comm.wait: raises and aborts the communicator instead of hanging on a dead rank)"""
        if timeout_ms is not None and self.transport == "rccl" and self.comm is not None:
            self.comm.wait(timeout_ms)
        if self._comm_stream is not None:
            torch.cuda.current_stream(self._comm_stream.device).wait_stream(self._comm_stream)
