#!/usr/bin/env python3
"""CPU study for the next kernel design step: how many pages of a paged visited
bitmap does one query touch, as a function of how the internal node ids are ordered?
The visited bitmap (1 bit per item, 125 KB at 1M items) is what pins the fused kernel to one
workgroup per CU; if node ids are renumbered by locality at index-load time (the ids are internal:
item_ids maps them back, list orders and therefore tie-breaking are unchanged), a query's ~10 k
visited nodes fall into few pages and the LDS footprint shrinks by an order of magnitude.
usage: tools/study_visited_locality.py [n_items] [n_queries]"""
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import breadth_first_order, reverse_cuthill_mckee

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import index_build, synth  # noqa: E402


def traverse(g, q, t):
    """The serving schedule (build_opt_graph.py:109-149) in numpy; returns every id whose bit the
    query sets (the visited set)."""
    x = g["item_embs"].astype(np.float32)
    score = lambda ids: -((x[ids] - q) ** 2).sum(1)
    nbv, nbrs = g["nb_values"], g["nb_row_splits"]

    def expand(level, frontier, vis):
        out = []
        for f in frontier:
            for v in nbv[level][nbrs[level][f]:nbrs[level][f + 1]]:
                if v not in vis:
                    vis.add(int(v))
                    out.append(int(v))
        return np.asarray(out, np.int64)

    def topk(ids, s, k):
        o = np.lexsort((np.arange(len(s)), -s))[:k]
        return ids[o], s[o]

    touched = set()
    ep = g["enter_points"].astype(np.int64)
    R, sR = topk(ep, score(ep), t[0])
    vis = set(int(v) for v in R)
    C = expand(1, R, vis)
    touched |= vis
    P, sP = topk(np.concatenate([R, C]), np.concatenate([sR, score(C)]), t[1])
    vis = set(int(v) for v in P)
    B = P
    for i in range(3):
        C = expand(0, B, vis)
        if len(C) < t[2 + i]:
            break
        B, sB = topk(C, score(C), t[2 + i])
        P = np.concatenate([P, B])
    touched |= vis
    return np.fromiter(touched, np.int64)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    d, ef = 64, 128
    t0 = time.time()
    x, assign = synth.make_corpus(n, d, n_clusters=256, noise=1.0)
    raw = index_build.build_hnsw(x.astype(np.float32), num_neighbors=32)
    ex = index_build.export_levels(raw, 2)
    g = {"item_embs": x, "nb_values": [v.astype(np.int64) for v in ex["nb_values"]],
         "nb_row_splits": ex["nb_row_splits"], "enter_points": ex["enter_points"]}
    print(f"index: {n} items, {len(g['enter_points'])} enter points, built in {time.time() - t0:.0f} s", flush=True)
    seqs = synth.make_queries(x, assign, nq, seed=99).astype(np.float32)
    qs = np.stack([s[np.abs(s).sum(1) > 0].mean(0) for s in seqs])
    t = [ef] * 5 + [200]
    visited = [traverse(g, q, t) for q in qs]
    print(f"visited ids per query: mean {np.mean([len(v) for v in visited]):.0f}, max {max(len(v) for v in visited)}")
    # ---- orderings: new_id = rank[old_id]
    rs0, v0 = g["nb_row_splits"][0], g["nb_values"][0]
    A = csr_matrix((np.ones(len(v0), np.int8), v0, rs0), shape=(n, n))
    A = ((A + A.T) > 0).astype(np.int8).tocsr()
    orders = {"as stored": np.arange(n)}
    o = np.argsort(assign, kind="stable")
    r = np.empty(n, np.int64); r[o] = np.arange(n); orders["by generating cluster"] = r
    bfs = breadth_first_order(A, int(g["enter_points"][0]), directed=False, return_predecessors=False)
    rest = np.setdiff1d(np.arange(n), bfs, assume_unique=False)
    o = np.concatenate([bfs, rest]); r = np.empty(n, np.int64); r[o] = np.arange(n); orders["BFS over level 0"] = r
    o = reverse_cuthill_mckee(A, symmetric_mode=True); r = np.empty(n, np.int64); r[o] = np.arange(n)
    orders["reverse Cuthill-McKee"] = r
    # 1-d projection order is a cheap stand-in for a space-filling curve / k-means order on real data
    from numpy.linalg import svd
    xs = x[np.random.default_rng(0).choice(n, 20000, replace=False)].astype(np.float32)
    u, s, vt = svd(xs - xs.mean(0), full_matrices=False)
    km = _kmeans_order(x.astype(np.float32), 1024)
    orders["k-means(1024) order"] = km
    for page in (1024, 4096):
        print(f"--- page = {page} ids ({page // 8} B); flat bitmap = {n // 8} B")
        for name, rank in orders.items():
            pages = np.array([len(np.unique(rank[v] // page)) for v in visited])
            print(f"{name:>24}: pages/query mean {pages.mean():7.1f}  p95 {np.percentile(pages, 95):6.0f}  max {pages.max():5d}"
                  f"  -> {int(pages.max()) * page // 8 / 1024:7.1f} KiB worst case")


def _kmeans_order(x, k, iters=8):
    rng = np.random.default_rng(1)
    c = x[rng.choice(len(x), k, replace=False)].copy()
    for _ in range(iters):
        a = _assign(x, c)
        for j in range(k):
            m = a == j
            if m.any():
                c[j] = x[m].mean(0)
    a = _assign(x, c)
    # order the clusters themselves along a greedy nearest-neighbour chain so that neighbouring clusters get neighbouring id ranges
    left = set(range(k)); cur = 0; chain = [0]; left.discard(0)
    while left:
        cand = np.fromiter(left, int)
        nxt = int(cand[np.argmin(((c[cand] - c[cur]) ** 2).sum(1))])
        chain.append(nxt); left.discard(nxt); cur = nxt
    pos = np.empty(k, np.int64); pos[np.asarray(chain)] = np.arange(k)
    o = np.argsort(pos[a], kind="stable")
    r = np.empty(len(x), np.int64); r[o] = np.arange(len(x))
    return r


def _assign(x, c, chunk=20000):
    out = np.empty(len(x), np.int64)
    cn = (c ** 2).sum(1)
    for i in range(0, len(x), chunk):
        xx = x[i:i + chunk]
        out[i:i + chunk] = np.argmin(cn[None, :] - 2.0 * xx @ c.T, axis=1)
    return out


if __name__ == "__main__":
    main()
