"""CPU: the C++ HNSW builder produces the reference's file layout and a graph
that the restated search recalls well on."""
import os

import numpy as np


def test_layout_invariants_and_recall(oracle, tmp_path):
    from nann_amd import index_build, synth
    n, d, ef = 6000, 32, 16
    embs, _ = synth.make_corpus(n, 64, n_clusters=16, noise=1.0, seed=3)
    embs = embs[:, :64]
    raw, ex = index_build.build_and_save_index(embs.astype(np.float32), 2, 32, str(tmp_path), seed=1,
                                               n_threads=1)
    levels = raw["levels"]
    # level law: P(level >= l) = 32^-l (Faiss), levels are 1-based counts
    assert levels.min() == 1 and abs((levels >= 2).mean() - 1 / 32) < 0.02
    assert raw["cum_nneighbor_per_level"][:3].tolist() == [0, 64, 96]
    # files exactly as build_hnsw_index.py leaves them
    ep = np.load(tmp_path / "enter_points.npy")
    assert ep.dtype == np.int64 and (ep == np.nonzero(levels > 2)[0]).all()
    for level, cap in ((0, 64), (1, 32)):
        v = np.load(tmp_path / f"neighbors_level_{level}_values.npy")
        rs = np.load(tmp_path / f"neighbors_level_{level}_row_splits.npy")
        assert v.dtype == np.int64 and rs.dtype == np.int64 and len(rs) == n + 1
        lens = np.diff(rs)
        assert rs[0] == 0 and rs[-1] == len(v) and lens.max() <= cap
        assert (lens[levels <= level] == 0).all()          # absent nodes have empty rows
        assert (lens[levels > level] > 0).mean() > 0.99    # present nodes are linked
        assert v.min() >= 0 and v.max() < n
        assert (levels[v] > level).all()                   # links stay inside the level
        for i in np.nonzero(lens)[0][:200]:                # no self links, no repeats in a row
            row = v[rs[i]:rs[i + 1]]
            assert i not in row and len(set(row.tolist())) == len(row)
    # deterministic with one thread
    raw2 = index_build.build_hnsw(embs.astype(np.float32), 32, seed=1, n_threads=1)
    assert (raw2["neighbors"] == raw["neighbors"]).all()
    # the restated search over this graph finds most of the true top-k
    n_enter = max(len(ep), ef)
    ep32 = np.nonzero(levels > 2)[0].astype(np.int32)
    if len(ep32) < ef:  # small corpus: widen the entry layer as the generator does (E >= ef)
        extra = np.setdiff1d(np.nonzero(levels > 1)[0], ep32)[: ef - len(ep32)]
        ep32 = np.sort(np.concatenate([ep32, extra])).astype(np.int32)
        if len(ep32) < ef:
            more = np.setdiff1d(np.arange(n), ep32)[: ef - len(ep32)]
            ep32 = np.sort(np.concatenate([ep32, more])).astype(np.int32)
    ids = synth.make_item_ids(n)
    oix = oracle.Index(embs, ids, [x.astype(np.int32) for x in ex["nb_values"]], ex["nb_row_splits"], ep32)
    sc = oracle.Scorer("l2", 64, oracle.EMB_F16)
    rng = np.random.default_rng(0)
    hits = tot = 0
    for b in range(24):
        q = embs[rng.integers(0, n)].astype(np.float32) + 0.05 * rng.standard_normal(64).astype(np.float32)
        rc, _, _, idx, _ = oracle.search(oix, sc, q, [ef] * 5 + [10])
        if rc:
            continue
        _, bi, _ = oracle.brute_force(oix, sc, q, 10)
        hits += len(set(bi.tolist()) & set(idx.tolist())); tot += 10
    assert tot >= 10 * 8 and hits / tot > 0.8, (hits, tot)  # requests whose rounds run dry fail like the reference


def test_multithreaded_build_is_valid():
    from nann_amd import index_build, synth
    embs, _ = synth.make_corpus(4000, 64, n_clusters=8, noise=1.0, seed=5)
    raw = index_build.build_hnsw(embs.astype(np.float32), 32, seed=2, n_threads=4)
    ex = index_build.export_levels(raw, 2)
    lens = np.diff(ex["nb_row_splits"][0])
    assert lens.max() <= 64 and lens.min() >= 1 and ex["nb_values"][0].max() < 4000
    # rows written under contention (a node's own list merged with the back-links that arrived first) keep the
    # single-threaded invariants: ids in range, inside the level, no self link, no repeat
    levels = raw["levels"]
    for level, cap in ((0, 64), (1, 32)):
        v, rs = ex["nb_values"][level], ex["nb_row_splits"][level]
        rl = np.diff(rs)
        assert rl.max() <= cap and v.min() >= 0 and v.max() < 4000 and (levels[v] > level).all()
        assert (rl[levels <= level] == 0).all()
        for i in np.nonzero(rl)[0]:
            row = v[rs[i]:rs[i + 1]]
            assert i not in row and len(set(row.tolist())) == len(row), (level, i)
