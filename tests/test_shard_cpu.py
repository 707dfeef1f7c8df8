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
