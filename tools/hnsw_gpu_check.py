#!/usr/bin/env python3
"""GPU HNSW builder vs the CPU builder on the bench corpus: build time, degree statistics, structural invariants, and
recall@200 of the serving traversal (L2) against brute force on each graph.  usage: tools/hnsw_gpu_check.py [items] [dim] [ef] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import index_build, ops, retrieval, synth  # noqa: E402


def invariants(ex, n, m):
    for level, cap in ((0, 2 * m), (1, m)):
        v, rs = ex["nb_values"][level], ex["nb_row_splits"][level]
        deg = np.diff(rs)
        assert rs[0] == 0 and rs[-1] == len(v) and (deg >= 0).all() and deg.max() <= cap, (level, deg.max())
        assert v.min() >= 0 and v.max() < n
        rows = np.repeat(np.arange(n), deg)
        assert (v != rows).all(), "self loop"
        key = rows.astype(np.int64) * n + v
        assert len(np.unique(key)) == len(key), "duplicate link in a row"
    return {"deg0": len(ex["nb_values"][0]) / n, "deg1_over_members": len(ex["nb_values"][1]) / max(1, int((ex["levels"] > 1).sum())) if "levels" in ex else None,
            "max0": int(np.diff(ex["nb_row_splits"][0]).max()), "E": len(ex["enter_points"])}


def recall(g, dim, ef, items, nq=64):
    dix = retrieval.Index.from_dict(g)
    q = ops.user_seq_mean(bench.make_query_batches(dim, nq, 1, 1.0, torch.device("cuda"), n_clusters=bench.n_clusters_for(items, ef))[0])
    sc = ops.Scorer("l2", dim, dix.item_embs.dtype)
    topn = [ef] * 5 + [200]
    r = retrieval.search(dix, sc, q, topn)
    torch.cuda.synchronize()
    st = r.status.cpu().numpy()
    hits = tot = 0
    for b in range(min(nq, 32)):
        if st[b]:
            continue
        _, bi = ops.top_k(ops.blaze_score(sc, q[b], item_emb=dix.item_embs), 200)
        hits += len(set(bi.cpu().tolist()) & set(r.index[b].cpu().tolist()))
        tot += 200
    return round(hits / max(tot, 1), 4), float((st == 0).mean()), float(r.counters[st == 0][:, 2, :].sum(1).float().mean())


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    ef = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f16"
    skip_cpu = len(sys.argv) > 5 and sys.argv[5] == "nocpu"
    ncl = bench.n_clusters_for(items, ef)
    embs, _ = synth.make_corpus(items, dim, n_clusters=ncl, noise=1.0, seed=1234, item_seed=1334)
    ids = synth.make_item_ids(items, seed=1235)
    if dtype == "bf16":
        bits = bench.to_bf16_bits(embs.astype(np.float32))
        dev_rows = torch.as_tensor(bits.view(np.int16)).cuda().view(torch.bfloat16)
        host_rows, x32 = bits, (bits.astype(np.uint32) << 16).view(np.float32)
    else:
        dev_rows, host_rows, x32 = torch.as_tensor(embs).cuda(), embs, embs.astype(np.float32)
    for it in range(2):
        torch.cuda.synchronize()
        t = time.time()
        ex = index_build.build_hnsw_gpu(dev_rows, 32, 40, seed=1236)
        torch.cuda.synchronize()
        t_gpu = time.time() - t
    print(f"GPU build {items} x {dim} {dtype}: {t_gpu:.2f} s (second run, incl. export)", flush=True)
    print("  invariants", invariants(ex, items, 32), flush=True)
    g = {"item_embs": host_rows, "item_ids": ids, "nb_values": [v.astype(np.int32) for v in ex["nb_values"]],
         "nb_row_splits": ex["nb_row_splits"], "enter_points": ex["enter_points"].astype(np.int32)}
    print("  recall@200, valid, rows/q", recall(g, dim, ef, items), flush=True)
    del g
    if skip_cpu:
        return
    t = time.time()
    raw = index_build.build_hnsw(x32, 32, 40, seed=1236)
    exc = index_build.export_levels(raw, 2)
    exc["levels"] = raw["levels"]
    print(f"CPU build: {time.time() - t:.2f} s", flush=True)
    print("  invariants", invariants(exc, items, 32), flush=True)
    g = {"item_embs": host_rows, "item_ids": ids, "nb_values": [v.astype(np.int32) for v in exc["nb_values"]],
         "nb_row_splits": exc["nb_row_splits"], "enter_points": exc["enter_points"].astype(np.int32)}
    print("  recall@200, valid, rows/q", recall(g, dim, ef, items), flush=True)


if __name__ == "__main__":
    main()
