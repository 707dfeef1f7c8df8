#!/usr/bin/env python3
"""configs[2] at full size (1M x 128-d f16, ef=128, top-200, MLP 256-128-1): the answers must not depend on how the
queries are batched, whichever form of the traversal a batch size selects (pipeline of phases: exact always, split-f16
at <= 160 queries; fused kernel otherwise; chunks of 1024).  2500 queries in one call against the same queries in calls
of 100 / 300 / 1024: status, ids, scores (bitwise) and counters equal; plus the size-independent properties of every
valid answer (ids in range and unique, scores descending).  usage: tools/mlp_batch_independence.py [index cache dir]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import ops, retrieval, synth  # noqa: E402


def main():
    cache = sys.argv[1] if len(sys.argv) > 1 else None
    dev = torch.device("cuda")
    items, dim, ef = 1_000_000, 128, 128
    g = bench.make_index(items, dim, ef, "hnsw", 1.0, "f16", 0, dev, bench.usable_cores(), cache_dir=cache)
    index = retrieval.Index.from_dict(g, device=dev)
    rows = g["item_embs"][:: max(1, items // 65536)]
    w = synth.make_mlp_weights_metric(dim, rows)
    topn = [ef] * 5 + [200]
    n = 2500
    seqs = bench.make_query_batches(dim, n, 1, 1.0, dev, n_clusters=bench.n_clusters_for(items, ef))[0]
    q = ops.user_seq_mean(seqs)
    out = {"workload": "1M x 128-d f16, ef=128, top-200, MLP 256-128-1, %d queries" % n}
    for prec in ("exact", "split"):
        sc = ops.Scorer("mlp", dim, torch.float16, weights=w, precision=prec)

        def run(lo, hi):
            r = retrieval.search(index, sc, q[lo:hi], topn)
            torch.cuda.synchronize()
            return [x.cpu().numpy() for x in (r.status, r.item_ids, r.scores, r.index, r.counters)]

        whole = run(0, n)
        res = {"valid": int((whole[0] == 0).sum())}
        ok = whole[0] == 0
        idx = whole[3][ok]
        res["ids_in_range"] = bool(((idx >= 0) & (idx < items)).all())
        res["ids_unique_per_query"] = bool(all(len(np.unique(r_)) == r_.size for r_ in idx))
        res["scores_descending"] = bool((np.diff(whole[2][ok], axis=1) <= 0).all())
        for step in (100, 300, 1024):
            parts = [run(i, min(n, i + step)) for i in range(0, n, step)]
            cat = [np.concatenate([p[j] for p in parts]) for j in range(5)]
            res["calls_of_%d_bitwise_equal" % step] = bool(
                (whole[0] == cat[0]).all() and (whole[1] == cat[1]).all() and (whole[3] == cat[3]).all() and
                (whole[2].view(np.uint32) == cat[2].view(np.uint32)).all() and (whole[4] == cat[4]).all())
        out[prec] = res
    print(json.dumps(out))
    bad = [k for p in ("exact", "split") for k, v in out[p].items() if v is False]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
