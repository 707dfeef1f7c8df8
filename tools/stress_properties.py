#!/usr/bin/env python3
"""Many query batches through the fused traversal on one index, checking size-independent properties of every
valid answer (ids in range and unique, scores descending, item ids = the index's); the first offenders are
re-run on the oracle and printed.  usage: tools/stress_properties.py items dim ef dtype n_batches batch [mode]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import ops, retrieval  # noqa: E402


def main():
    items, dim, ef = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    dtype, n_batches, batch = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    mode = sys.argv[7] if len(sys.argv) > 7 else "auto"
    dev = torch.device("cuda")
    g = bench.make_index(items, dim, ef, "hnsw", 1.0, dtype, 0, dev, bench.usable_cores())
    dix = retrieval.Index.from_dict(g)
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    sc = ops.Scorer("l2", dim, tdt)
    topn = [ef] * 5 + [200]
    retrieval.set_traversal_mode(mode)
    bad_total = 0
    for bi in range(n_batches):
        seq = bench.make_query_batches(dim, batch, 1, 1.0, dev, seed=1000 + bi)[0]
        q = ops.user_seq_mean(seq)
        r = retrieval.search(dix, sc, q, topn)
        torch.cuda.synchronize()
        st = r.status.cpu().numpy()
        idx = r.index.cpu().numpy()
        scs = r.scores.cpu().numpy()
        ok = st == 0
        srt = np.sort(idx, axis=1)
        bad = ok & ((idx.min(1) < 0) | (idx.max(1) >= items) | (srt[:, 1:] == srt[:, :-1]).any(1) |
                    (scs[:, 1:] > scs[:, :-1]).any(1))
        print(f"batch {bi}: valid {int(ok.sum())}/{batch}, status histogram {np.bincount(st).tolist()[:8]}, "
              f"bad {int(bad.sum())}", flush=True)
        if bad.any():
            bad_total += int(bad.sum())
            from oracle import oracle as O
            oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
            osc = O.Scorer("l2", dim, O.EMB_F16 if dtype == "f16" else O.EMB_BF16)
            for b in np.nonzero(bad)[0][:4]:
                est, eids, esc, eidx, ectr = O.search_batch(oix, osc, q[b:b + 1].cpu().numpy(), topn, n_threads=1)
                row = idx[b]
                where = np.nonzero((row < 0) | (row >= items))[0]
                print(f"  query {b}: oracle status {est[0]}; device counters {r.counters[b].cpu().numpy().tolist()}; "
                      f"oracle counters {ectr[0].tolist()}; out-of-range ranks {where.tolist()[:10]} values "
                      f"{row[where].tolist()[:10]}; first mismatch rank "
                      f"{int(np.argmax(row != eidx[0])) if (row != eidx[0]).any() else -1}", flush=True)
    print("TOTAL bad", bad_total)


if __name__ == "__main__":
    main()
