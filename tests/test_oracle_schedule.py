"""CPU: the C oracle's schedule against an independent pure-Python/numpy
restatement of build_opt_graph.py:109-149 written with sets and sorted()."""
import os

import numpy as np
import pytest


def py_diff(values, visited):
    out = []
    for v in values:
        if v not in visited:
            visited.add(v)
            out.append(v)
    return out


def py_topk(ids, scores, k):
    assert len(ids) == len(scores) and len(ids) >= k
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))[:k]
    return [ids[i] for i in order], [scores[i] for i in order]


def py_search(z, oracle, q, t):
    embs = z["item_embs"]
    d = embs.shape[1]
    sc = oracle.Scorer("l2", d, oracle.EMB_F16)

    def score(ids):
        rc, s = oracle.score_rows(sc, q, embs[np.asarray(ids, np.int64)])
        assert rc == 0
        return s.tolist()

    def nbrs(level, frontier):
        v, rs = z[f"nb_values_{level}"], z[f"nb_row_splits_{level}"]
        out = []
        for f in frontier:
            out.extend(v[rs[f]:rs[f + 1]].tolist())
        return out

    ep = z["enter_points"].tolist()
    R, sR = py_topk(ep, score(ep), t[0])
    C = nbrs(1, R)
    vis = set()
    R = py_diff(R, vis)
    C = py_diff(C, vis)
    P, sP = py_topk(R + C, sR + score(C), t[1])
    vis = set()
    B = py_diff(P, vis)
    for i in range(3):
        C = py_diff(nbrs(0, B), vis)
        B, sB = py_topk(C, score(C), t[2 + i])
        P, sP = P + B, sP + sB
    P, sP = py_topk(P, sP, t[5])
    return z["item_ids"][np.asarray(P)], np.asarray(sP, np.float32), np.asarray(P)


@pytest.mark.parametrize("name", ["small_l2_d64.npz", "small_l2_d128.npz"])
def test_schedule_matches_python_restatement(oracle, golden_dir, name):
    z = dict(np.load(os.path.join(golden_dir, name)))
    ix = oracle.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                      [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    sc = oracle.Scorer("l2", z["item_embs"].shape[1], oracle.EMB_F16)
    t = z["level_topn"].tolist()
    for b in range(len(z["q"])):
        rc, ids, scores, idx, _ = oracle.search(ix, sc, z["q"][b], t)
        assert rc == 0
        pids, pscores, pidx = py_search(z, oracle, z["q"][b], t)
        assert (ids == pids).all() and (idx == pidx).all()
        assert (scores.view(np.uint32) == pscores.view(np.uint32)).all()


def test_failing_requests(oracle, golden_dir):
    """Steps with fewer candidates than k fail the request (topk_op.cc:67-71)."""
    z = dict(np.load(os.path.join(golden_dir, "small_l2_d64.npz")))
    ix = oracle.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                      [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    sc = oracle.Scorer("l2", 64, oracle.EMB_F16)
    E = len(z["enter_points"])
    rc, *_ = oracle.search(ix, sc, z["q"][0], [E + 1, 4, 4, 4, 4, 4])
    assert rc == oracle.ERR_TOPK_K_GT_N
    rc, *_ = oracle.search(ix, sc, z["q"][0], [4, 4, 4, 4, 4, 17])  # pool of 16 < 17
    assert rc == oracle.ERR_TOPK_K_GT_N
    rc, *_ = oracle.search(ix, sc, z["q"][0], [0, 4, 4, 4, 4, 4])  # empty frontier -> empty batch
    assert rc == oracle.ERR_EMPTY_SCORE_BATCH


def test_merge_topk(oracle):
    rng = np.random.default_rng(5)
    s = np.sort(rng.integers(0, 20, size=(4, 10)).astype(np.float32), axis=1)[:, ::-1].copy()
    ids = rng.integers(1, 10**6, size=(4, 10)).astype(np.int64)
    rc, ms, mi = oracle.merge_topk(s, ids, 10)
    flat_s, flat_i = s.reshape(-1), ids.reshape(-1)
    order = sorted(range(40), key=lambda i: (-flat_s[i], i))[:10]
    assert rc == 0 and mi.tolist() == flat_i[order].tolist() and ms.tolist() == flat_s[order].tolist()


def test_eval_graph_variant_matches_python_restatement(oracle, golden_dir):
    """SURVEY.md 8(f3), oracle side: Model.retrieval / search_level (model.py:299-362) written
    with Python sets exactly as the TF graph does (tf.unique -> tf.sets.set_difference ->
    tf.sets.set_union -> top_k(min(k, n)) -> score >= worst frontier rule)."""
    z = dict(np.load(os.path.join(golden_dir, "small_l2_d64.npz")))
    ix = oracle.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                      [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    sc = oracle.Scorer("l2", 64, oracle.EMB_F16)
    num_scoring, tk, topk_eval = [3, 1, 1], [40, 20, 10], 25

    def score(q, ids):
        rc, s = oracle.score_rows(sc, q, z["item_embs"][np.asarray(ids, np.int64)])
        assert rc == 0
        return s.tolist()

    def py_eval(q):
        ep = z["enter_points"].tolist()
        res, sres = py_topk(ep, score(q, ep), min(tk[2], len(ep)))
        for level in (1, 0):
            v, rs = z[f"nb_values_{level}"], z[f"nb_row_splits_{level}"]
            visited = set(res)
            cand = list(res)
            for _ in range(num_scoring[level]):
                nxt = []
                for c in cand:
                    nxt.extend(v[rs[c]:rs[c + 1]].tolist())
                nxt = sorted(set(nxt) - visited)           # tf.unique + tf.sets.set_difference
                visited |= set(nxt)                        # tf.sets.set_union
                snxt = score(q, nxt)
                res, sres = py_topk(res + nxt, sres + snxt, min(tk[level], len(res) + len(nxt)))
                cand = [i for i, s in zip(nxt, snxt) if s >= sres[-1]]
        return res[:topk_eval], sres[:topk_eval]

    for b in range(len(z["q"])):
        rc, ids, scores, idx = oracle.search_eval(ix, sc, z["q"][b], num_scoring, tk, topk_eval)
        assert rc == 0
        pres, psc = py_eval(z["q"][b])
        assert idx.tolist() == pres
        assert (scores.view(np.uint32) == np.asarray(psc, np.float32).view(np.uint32)).all()
        assert (ids == z["item_ids"][np.asarray(pres)]).all()
