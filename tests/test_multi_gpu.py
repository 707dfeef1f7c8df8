"""-m gpu, N > 1 devices: the exchange of the item-id-sharded search over REAL RCCL (BASELINE configs 3-4).

One process per GPU, N = min(8, devices): every rank builds its own shard (own items, own graph), searches every query on
it with the fused kernel, and nann_sharded_topk exchanges the per-shard lists with ONE ncclAllGather over xGMI and merges
them on the device.  Rank 0 re-derives every shard's lists and their merge with the CPU oracle: merged ids and scores are
bit-identical, with one shard failing some queries and the exchange overlapped with the next search or not;
nann_comm_ranks reports N RCCL ranks; the bounded wait passes.

The pool's boxes have ONE GPU, where this test skips: it is here so that the first multi-GPU box the suite meets is a
correctness run.  `test_multi_gpu_harness_on_one_gpu` runs the same worker with every rank on device 0 and gloo carrying
the records (RCCL refuses two ranks on one device), so the harness itself is exercised every round."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp, shared_gpu):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nann_amd import ops, retrieval, shard, synth
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0 if shared_gpu else rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # carries the 128-byte RCCL id and the checker's gathers only
    d, ef, k, nq, n_items = 64, 32, 200, 96, 20000
    topn = [ef] * 5 + [k]
    g = synth.make_index(n_items, d, ef=ef, seed=7, noise=1.0, n_clusters=32, shard=rank, device="cuda")
    index = retrieval.Index.from_dict(g, device=dev)
    scorer = ops.Scorer("l2", d, torch.float16)
    transport = "records" if shared_gpu else "rccl"
    ss = shard.ShardedSearch(topn, world, rank, transport=transport)
    if not shared_gpu:
        assert ss.comm.ranks() == (world, world), ss.comm.ranks()  # N shards, N ranks in the RCCL communicator
    fail_rank = min(3, world - 1)
    merged = []
    for batch in range(3):
        seq = synth.make_queries(g["item_embs"], g["assign"], nq, seed=100 + batch)  # (every rank: the same seed, its own shard's rows ...)
        seq = [seq]
        dist.broadcast_object_list(seq, src=0)                                        # ... so rank 0's queries go to all
        q = ops.user_seq_mean(torch.as_tensor(seq[0]).to(dev))
        overlap = bool(batch & 1) and not shared_gpu
        r = ss.search(index, scorer, q, topn, overlap=overlap)
        if rank == fail_rank:  # this shard fails every fifth query (TopKV2's k > n code): contributes (-inf, 0) for those
            r.status[::5] = 4
        mi, ms = ss.merge(r, overlap=overlap)
        ss.wait(timeout_ms=60000)
        torch.cuda.synchronize()
        merged.append((mi.cpu().numpy(), ms.cpu().numpy(), r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
                       q.cpu().numpy()))
    box = [None] * world
    dist.all_gather_object(box, [(m[2], m[3], m[4]) for m in merged])
    ok = True
    if rank == 0:
        for batch in range(3):
            mi, ms, _, _, _, q = merged[batch]
            parts = []
            for r_ in range(world):  # every shard's lists again, from the CPU oracle on that shard's index
                g_ = synth.make_index(n_items, d, ef=ef, seed=7, noise=1.0, n_clusters=32, shard=r_, device="cuda")
                oix = O.Index(g_["item_embs"], g_["item_ids"], g_["nb_values"], g_["nb_row_splits"], g_["enter_points"])
                st, ids, sc, _, _ = O.search_batch(oix, O.Scorer("l2", d, O.EMB_F16), q, topn)
                st = st.copy()
                if r_ == fail_rank:
                    st[::5] = 4
                ok = ok and (box[r_][batch][0] == st).all() and (box[r_][batch][1][st == 0] == ids[st == 0]).all() \
                    and (box[r_][batch][2][st == 0].view(np.uint32) == sc[st == 0].view(np.uint32)).all()
                parts.append((st, ids, sc))
            for b in range(nq):
                s_in = np.stack([np.where(p[0][b] == 0, p[2][b], -np.inf).astype(np.float32) for p in parts])
                i_in = np.stack([np.where(p[0][b] == 0, p[1][b], 0) for p in parts])
                rc, es, ei = O.merge_topk(s_in, i_in, k)
                ok = ok and rc == 0 and (mi[b] == ei).all() and (ms[b].view(np.uint32) == es.view(np.uint32)).all()
    res = [None] * world
    dist.all_gather_object(res, [m[0].tobytes() for m in merged])
    ok = ok and all(x == res[0] for x in res)  # identical on every rank
    if rank == 0:
        open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def _device_count():
    from nann_amd import _lib
    return _lib.lib().nann_device_count()


def test_sharded_search_over_rccl_on_every_gpu_of_the_node(tmp_path):
    n = min(8, _device_count())
    if n < 2:
        pytest.skip("one GPU: the RCCL exchange needs N > 1 devices (the harness runs in test_multi_gpu_harness_on_one_gpu)")
    mp.spawn(_worker, args=(n, _free_port(), str(tmp_path), False), nprocs=n, join=True)
    assert open(tmp_path / "ok").read() == "1"


def test_multi_gpu_harness_on_one_gpu(tmp_path):
    assert torch.cuda.is_available()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"
