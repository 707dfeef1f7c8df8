"""CPU, world_size 2 over gloo: the exchange step of the sharded search
(all-gather of per-shard top-k + TopKV2-ordered merge).
The per-shard searches are produced by the oracle here (no GPU in this
container); the GPU run of the same path is bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nann_amd import retrieval, shard, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, ef, k, nq = 64, 16, 20, 24
    topn = [ef] * 5 + [k]
    seq = synth.make_queries_from_centres(d, nq, n_clusters=16, noise=1.0, seed=7)
    q = np.stack([O.user_seq_mean(s) for s in seq])

    def shard_result(r):
        g = synth.make_index(1500, d, ef=ef, seed=7, noise=1.0, n_clusters=16, shard=r)
        ix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        return O.search_batch(ix, O.Scorer("l2", d, O.EMB_F16), q, topn)

    st, ids, scores, idx, ctr = shard_result(rank)
    local = retrieval.SearchResult(torch.as_tensor(ids), torch.as_tensor(scores), torch.as_tensor(idx),
                                   torch.as_tensor(st), None)
    merged_ids, merged_scores = shard.ShardedSearch(topn, world, rank, merge="host", transport="torch").merge(local)
    if rank == 0:
        parts = [(st, ids, scores)] + [shard_result(r)[:3] for r in range(1, world)]
        exp_ids, exp_scores = [], []
        for b in range(nq):
            s = np.stack([np.where(p[0][b] == 0, p[2][b], -np.inf) for p in parts])
            i = np.stack([p[1][b] for p in parts])
            rc, ms, mi = O.merge_topk(s, i, k)
            assert rc == 0
            exp_ids.append(mi); exp_scores.append(ms)
        ok = (merged_ids.numpy() == np.stack(exp_ids)).all() and \
             (merged_scores.numpy() == np.stack(exp_scores)).all()
        # item ids of different shards are disjoint ranges
        assert set(parts[0][1].ravel()).isdisjoint(set(parts[1][1].ravel()) - {0})
        open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def _worker8(rank, world, port, tmp):
    """8 ranks over gloo: the device path's RECORD layout (pack -> all-gather of raw bytes -> merge from the records,
    nann_comm.hip) with host stand-ins for its two kernels; shard 3 fails every fifth query, shard 5 all of them."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nann_amd import retrieval, shard, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, ef, k, nq = 64, 16, 20, 20
    topn = [ef] * 5 + [k]
    seq = synth.make_queries_from_centres(d, nq, n_clusters=8, noise=1.0, seed=7)
    q = np.stack([O.user_seq_mean(s) for s in seq])
    g = synth.make_index(800, d, ef=ef, seed=7, noise=1.0, n_clusters=8, shard=rank)
    ix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    st, ids, scores, idx, ctr = O.search_batch(ix, O.Scorer("l2", d, O.EMB_F16), q, topn)
    st = st.copy()
    if rank == 3:
        st[::5] = 4   # (TopKV2 k > n on this shard: the reference's failure code)
    if rank == 5:
        st[:] = 6
    local = retrieval.SearchResult(torch.as_tensor(ids), torch.as_tensor(scores), torch.as_tensor(idx), torch.as_tensor(st), None)
    outs = {}
    for transport, merge in (("records", "device"), ("torch", "host")):
        mi, ms = shard.ShardedSearch(topn, world, rank, merge=merge, transport=transport).merge(local)
        outs[transport + "/" + merge] = (mi.numpy(), ms.numpy())
    box = [None] * world
    dist.all_gather_object(box, (st, ids, scores))
    if rank == 0:
        exp_ids, exp_scores = [], []
        for b in range(nq):
            s = np.stack([np.where(p[0][b] == 0, p[2][b], -np.inf).astype(np.float32) for p in box])
            i = np.stack([np.where(p[0][b] == 0, p[1][b], 0) for p in box])
            rc, ms_, mi_ = O.merge_topk(s, i, k)
            assert rc == 0
            exp_ids.append(mi_); exp_scores.append(ms_)
        exp_ids, exp_scores = np.stack(exp_ids), np.stack(exp_scores)
        ok = all((v[0] == exp_ids).all() and (v[1].view(np.uint32) == exp_scores.view(np.uint32)).all() for v in outs.values())
        # the failing shards' items never appear while six healthy shards hold candidates
        lo5, hi5 = 5 * 800, 6 * 800
        ok = ok and not ((exp_ids > lo5) & (exp_ids <= hi5)).any() and np.isfinite(exp_scores).all()
        open(os.path.join(tmp, "ok8"), "w").write("1" if ok else "0 " + repr({k_: (v[0] == exp_ids).all() for k_, v in outs.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_merge_world8_with_failing_shards_and_the_record_layout(tmp_path):
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert open(tmp_path / "ok8").read() == "1"


def test_record_layout_of_a_failed_query():
    from nann_amd import shard
    sc = np.array([[3.0, 2.0], [1.0, 0.5]], np.float32)
    ids = np.array([[7, 8], [9, 10]], np.int64)
    rec = shard.pack_record_host(sc, ids, np.array([0, 5]))
    assert rec.size == 256 and rec.size == shard.record_bytes(2, 2)
    assert rec[:16].view(np.float32).tolist() == [3.0, 2.0, -np.inf, -np.inf] and rec[16:48].view(np.int64).tolist() == [7, 8, 0, 0]
    s, i = shard.merge_records_host(np.stack([rec, shard.pack_record_host(sc - 0.25, ids + 100, None)]), 2, 2, 3)
    assert i.tolist() == [[7, 107, 8], [109, 110, 0]] and s[0].tolist() == [3.0, 2.75, 2.0] and s[1][2] == -np.inf


def test_sharded_merge_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok").read() == "1"


def test_merge_host_order():
    """ties -> lower shard, then lower local rank (shard-major concatenation)."""
    from nann_amd import shard
    s = np.array([[[5, 3, 1], [5, 4, 1]]], np.float32)
    i = np.array([[[10, 11, 12], [20, 21, 22]]], np.int64)
    ms, mi = shard.merge_host(s, i, 4)
    assert mi.tolist() == [[10, 20, 21, 11]] and ms.tolist() == [[5, 5, 4, 3]]


def _worker_dead_peer(rank, world, port, tmp):
    """rank 1 dies before the exchange; rank 0's merge must FAIL within the group's timeout instead of hanging"""
    import datetime
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from nann_amd import retrieval, shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=6))
    dist.barrier()
    if rank == 1:
        os._exit(0)  # no goodbye: the process is gone
    nq, k = 8, 20
    local = retrieval.SearchResult(torch.zeros((nq, k), dtype=torch.int64), torch.zeros((nq, k)), None,
                                   torch.zeros(nq, dtype=torch.int32), None)
    time.sleep(0.5)
    t0 = time.perf_counter()
    verdict = "hung"
    try:
        shard.ShardedSearch([4] * 5 + [k], world, rank, merge="host", transport="records").merge(local)
        verdict = "returned"
    except Exception as e:  # gloo: connection closed by peer / timed out
        verdict = "raised %.1f %s" % (time.perf_counter() - t0, type(e).__name__)
    open(os.path.join(tmp, "dead_peer"), "w").write(verdict)
    os._exit(0)  # (destroy_process_group would wait for the dead peer)


def test_a_dead_rank_fails_the_exchange_instead_of_hanging(tmp_path):
    """Host logic of VERDICT r5 missing 4 on the CPU transports: with a bounded process-group timeout a rank that died makes the
    exchange RAISE on the survivors (the device path's counterpart is nann_comm_wait: tests/test_ops_gpu.py)."""
    port = _free_port()
    mp.spawn(_worker_dead_peer, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    verdict = open(tmp_path / "dead_peer").read()
    assert verdict.startswith("raised"), verdict
    assert float(verdict.split()[1]) < 30.0, verdict
