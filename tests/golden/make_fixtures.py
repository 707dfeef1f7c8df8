"""Regenerates tests/golden/small_*.npz: a small seeded index (reference on-disk
layout), queries, and the CPU oracle's answers for them.

These are REGRESSION vectors for the restated schedule (the reference holds no
end-to-end expected outputs: its tests are print-only); the known answers that
pin the two custom ops against the reference itself are in reference_ops.json.

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nann_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def make(name, n, d, ef, k, nq, noise, seed):
    g = synth.make_index(n, d, ef=ef, seed=seed, noise=noise, n_clusters=32)
    seq = synth.make_queries(g["item_embs"], g["assign"], nq, seed=seed + 7)
    q = np.stack([O.user_seq_mean(s) for s in seq])
    ix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    sc = O.Scorer("l2", d, O.EMB_F16)
    topn = np.array([ef] * 5 + [k], np.int32)
    st, ids, scores, idx, ctr = O.search_batch(ix, sc, q, topn)
    np.savez_compressed(
        os.path.join(HERE, name), item_embs=g["item_embs"], item_ids=g["item_ids"],
        nb_values_0=g["nb_values"][0], nb_row_splits_0=g["nb_row_splits"][0],
        nb_values_1=g["nb_values"][1], nb_row_splits_1=g["nb_row_splits"][1],
        enter_points=g["enter_points"], comm_seq=seq, q=q, level_topn=topn,
        status=st, out_item_ids=ids, out_scores=scores, out_index=idx, counters=ctr)
    print(name, "status", np.bincount(st), "scored/query", ctr[:, 2, :].sum(1).mean())


if __name__ == "__main__":
    make("small_l2_d64.npz", 3000, 64, 16, 20, 12, 1.0, 11)
    make("small_l2_d128.npz", 2500, 128, 24, 30, 8, 1.0, 23)
