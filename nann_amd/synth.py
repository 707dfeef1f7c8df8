"""Seeded synthetic inputs in the reference's on-disk layout.

The hot path takes its index as INPUT: `item_embs.npy`, `item_ids.npy`,
`neighbors_level_{0,1}_{values,row_splits}.npy`, `enter_points.npy` exactly as
`NANN_impls/nann/delivery/build_hnsw_index.py:33-67` writes them (a CSR row for
EVERY item at every level, empty when the node is absent there; -1 slots
dropped; enter points = nodes whose Faiss `levels` exceed the start level).
Faiss is not available, so this module produces arrays with the same layout and
the same level law (P(level >= l) = M^-l); each layer is built like an HNSW
layer inserted sequentially with exact search and Faiss' neighbour-selection
heuristic (ragged rows, long-range links from early nodes), or optionally as
plain exact k-NN rows at the degree cap.  Search parity is pinned at this CSR
boundary regardless of who built the graph.

numpy does the random draws (bit-reproducible on every box); torch is used only
for the chunked distance GEMM + top-k (CPU here, cuda on the MI355X box).
"""
import math

import numpy as np

M_DEFAULT = 32  # --hnsw-num-neighbors, build_hnsw_index.py:24
START_LEVEL = 2  # --hnsw-start-level, build_hnsw_index.py:22


def make_centres(d, n_clusters=256, seed=1234):
    """Cluster centres f32[C, d]; shared by every shard of a sharded corpus."""
    return np.random.default_rng(seed).standard_normal((n_clusters, d), dtype=np.float32)


def make_corpus(n_items, d, n_clusters=256, noise=0.5, seed=1234, item_seed=None):
    """-> (item_embs f16[N,d], cluster assignment i32[N]).  Centres come from
    `seed`, the items from `item_seed` (default seed + 100) so that shards of
    one corpus share centres but not items."""
    centres = make_centres(d, n_clusters, seed)
    rng = np.random.default_rng(seed + 100 if item_seed is None else item_seed)
    assign = rng.integers(0, n_clusters, size=n_items, dtype=np.int32)
    x = centres[assign] + noise * rng.standard_normal((n_items, d), dtype=np.float32)
    x *= np.float32(1.0 / math.sqrt(d))
    return x.astype(np.float16), assign


def make_queries_from_centres(d, n_queries, n_clusters=256, noise=0.5, seq_len=50, min_len=7,
                              seed=1234, query_seed=4321):
    """UserBehavior-shaped `comm_seq` f16[B, seq_len, d] that does not depend on
    any shard's items: each history is 7..50 fresh draws around one centre,
    zero-padded tail (convert_UB_to_tfrecord.py:121-137)."""
    centres = make_centres(d, n_clusters, seed)
    rng = np.random.default_rng(query_seed)
    seq = np.zeros((n_queries, seq_len, d), np.float32)
    lens = rng.integers(min_len, seq_len + 1, size=n_queries)
    cl = rng.integers(0, n_clusters, size=n_queries)
    x = centres[cl][:, None, :] + noise * rng.standard_normal((n_queries, seq_len, d), dtype=np.float32)
    x *= np.float32(1.0 / math.sqrt(d))
    mask = np.arange(seq_len)[None, :] < lens[:, None]
    seq[mask] = x[mask]
    return seq.astype(np.float16)


def make_item_ids(n_items, seed=1235):
    """Permutation of 1..N (0 means "missing", convert_UB_to_tfrecord.py:112)."""
    rng = np.random.default_rng(seed)
    return (rng.permutation(n_items).astype(np.int64) + 1)


def assign_levels(n_items, m=M_DEFAULT, seed=1236, min_enter=0):
    """Faiss level law: level = floor(-ln(U)/ln(M)).  Returns the per-node top
    level (0-based).  Promotes extra nodes until >= min_enter reach START_LEVEL
    (the generator must guarantee E >= ef)."""
    rng = np.random.default_rng(seed)
    u = rng.random(n_items)
    lev = np.floor(-np.log(np.maximum(u, 1e-300)) / math.log(m)).astype(np.int32)
    need = min_enter - int((lev >= START_LEVEL).sum())
    if need > 0:
        cand = np.nonzero(lev < START_LEVEL)[0]
        pick = rng.choice(cand, size=need, replace=False)
        lev[pick] = START_LEVEL
    return lev


def _heuristic_select(sub, cand, cand_d, cand_valid, width, dev):
    """HNSW neighbour-selection heuristic (Malkov & Yashunin alg. 4; Faiss
    shrink_neighbor_list): walk candidates nearest-first, keep c iff it is
    closer to the base node than to every neighbour kept so far.
    cand: [c, k] local ids sorted by cand_d ascending; cand_valid: [c, k] bool.
    Returns (sel [c, width] local ids, kept-first; lens [c])."""
    import torch
    c, k = cand.shape
    sel = torch.zeros((c, width), dtype=torch.int64, device=dev)
    lens = torch.zeros(c, dtype=torch.int64, device=dev)
    step = max(1, min(c, (1 << 26) // (k * k)))
    for a in range(0, c, step):
        b = min(a + step, c)
        cv = sub[cand[a:b]]                                       # [b-a, k, d]
        csq = (cv * cv).sum(2)
        pd = csq[:, :, None] - 2.0 * torch.bmm(cv, cv.transpose(1, 2)) + csq[:, None, :]
        base_d = cand_d[a:b]
        kept = torch.zeros((b - a, k), dtype=torch.bool, device=dev)
        nkept = torch.zeros(b - a, dtype=torch.int64, device=dev)
        for j in range(k):
            dom = ((pd[:, j, :] <= base_d[:, j, None]) & kept).any(1)
            take = (~dom) & (nkept < width) & cand_valid[a:b, j]
            kept[:, j] = take
            nkept += take.long()
        order = torch.argsort((~kept).to(torch.int8), dim=1, stable=True)[:, :width]
        sel[a:b] = torch.gather(cand[a:b], 1, order)
        lens[a:b] = nkept
    return sel, lens


def _hnsw_like(x_f16, members, deg, device=None, chunk=4096):
    """Graph over `members` shaped like an HNSW layer built by sequential
    insertion with exact search: node i first links to heuristic-selected
    nearest neighbours among the nodes inserted BEFORE it (so early nodes keep
    long-range links), every link gets a back-link, and over-full rows are
    shrunk by the same heuristic to `deg` links -- the structure of
    Faiss' add_links_starting_from / shrink_neighbor_list without the
    approximate search.  Returns (ids int32 [n, deg] GLOBAL ids, lens [n])."""
    import torch
    dev = torch.device(device or "cpu")
    sub = torch.from_numpy(np.ascontiguousarray(x_f16[members])).to(dev).float()
    n = sub.shape[0]
    sq = (sub * sub).sum(1)
    k = min(2 * deg, n - 1)
    # ---- forward links: nearest among earlier nodes -------------------------
    src_l, dst_l, d_l = [], [], []
    for s in range(0, n, chunk):
        e = min(s + chunk, n)
        c = e - s
        dist = sq[s:e, None] - 2.0 * (sub[s:e] @ sub[:e].T) + sq[None, :e]
        rows = torch.arange(s, e, device=dev)
        # only earlier nodes are candidates: columns < s are all earlier, the diagonal block needs the mask
        dist[:, s:e].masked_fill_(rows[None, :] >= rows[:, None], float("inf"))
        kk = min(k, e)
        dv, idx = torch.topk(dist, kk, dim=1, largest=False, sorted=True)
        del dist
        valid = torch.isfinite(dv)
        sel, lens = _heuristic_select(sub, idx, dv, valid, min(deg, kk), dev)
        m = torch.arange(sel.shape[1], device=dev)[None, :] < lens[:, None]
        src = rows[:, None].expand_as(sel)[m]
        dst = sel[m]
        src_l.append(src); dst_l.append(dst)
    src = torch.cat(src_l); dst = torch.cat(dst_l)
    # ---- add back-links, then shrink every row to <= deg ---------------------
    a = torch.cat([src, dst]); b = torch.cat([dst, src])
    dd = torch.empty(len(a), dtype=torch.float32, device=dev)
    for s0 in range(0, len(a), 1 << 22):  # edge distances in pieces: [edges, d] temporaries stay small
        s1 = min(s0 + (1 << 22), len(a))
        dd[s0:s1] = ((sub[a[s0:s1]] - sub[b[s0:s1]]) ** 2).sum(1)
    o1 = torch.argsort(dd, stable=True)                 # by (row, dist): distance first, then a stable sort by row
    order = o1[torch.argsort(a[o1], stable=True)]
    del o1
    a, b, dd = a[order], b[order], dd[order]
    counts = torch.bincount(a, minlength=n)
    starts = torch.cumsum(counts, 0) - counts
    rank = torch.arange(len(a), device=dev) - starts[a]
    keep = rank < k
    cand = torch.zeros((n, k), dtype=torch.int64, device=dev)
    cand_d = torch.full((n, k), float("inf"), device=dev)
    cand[a[keep], rank[keep]] = b[keep]
    cand_d[a[keep], rank[keep]] = dd[keep]
    out = np.zeros((n, deg), np.int64)
    lens_out = np.zeros(n, np.int64)
    mem_t = torch.from_numpy(np.asarray(members, dtype=np.int64)).to(dev)
    for s in range(0, n, chunk):
        e = min(s + chunk, n)
        sel, lens = _heuristic_select(sub, cand[s:e], cand_d[s:e], torch.isfinite(cand_d[s:e]),
                                      min(deg, k), dev)
        out[s:e, :sel.shape[1]] = mem_t[sel].cpu().numpy()
        lens_out[s:e] = lens.cpu().numpy()
    return out.astype(np.int32), lens_out


def _knn(x_f16, members, k, device=None, chunk=4096):
    """Exact k nearest neighbours (squared L2, self excluded), nearest first.
    Returns (ids int32 [n, k] GLOBAL ids, lens [n])."""
    import torch
    dev = torch.device(device or "cpu")
    sub = torch.from_numpy(np.ascontiguousarray(x_f16[members])).to(dev).float()
    n = sub.shape[0]
    k = min(k, n - 1)
    sq = (sub * sub).sum(1)
    out = np.zeros((n, k), np.int64)
    mem_t = torch.from_numpy(np.asarray(members, dtype=np.int64)).to(dev)
    for s in range(0, n, chunk):
        e = min(s + chunk, n)
        dist = sq[s:e, None] - 2.0 * (sub[s:e] @ sub.T) + sq[None, :]
        dist[torch.arange(e - s, device=dev), torch.arange(s, e, device=dev)] = float("inf")
        idx = torch.topk(dist, k, dim=1, largest=False, sorted=True).indices
        out[s:e] = mem_t[idx].cpu().numpy()
    return out.astype(np.int32), np.full(n, k, np.int64)


def build_graph(item_embs, levels, m=M_DEFAULT, device=None, mode="hnsw"):
    """-> dict with nb_values[2] (int32), nb_row_splits[2] (int64[N+1]),
    enter_points (int32, ascending) in build_hnsw_index.py's layout.
    mode="hnsw": insertion-order graph with heuristic pruning (ragged rows,
                 <= 2M links at level 0, <= M above) -- the bench/test default;
    mode="knn":  exact k-NN rows, every row at the degree cap (the worst-case
                 gather sizes of SURVEY.md section 8's table)."""
    n = item_embs.shape[0]
    nb_values, nb_row_splits = [], []
    for level in range(START_LEVEL):
        members = np.nonzero(levels >= level)[0]
        deg = 2 * m if level == 0 else m  # Faiss: 2M links at level 0, M above
        if mode == "hnsw":
            nbrs, lens = _hnsw_like(item_embs, members, deg, device=device)
        else:
            nbrs, lens = _knn(item_embs, members, deg, device=device)
        k = nbrs.shape[1]
        row_len = np.zeros(n, np.int64)
        row_len[members] = lens
        rs = np.zeros(n + 1, np.int64)
        np.cumsum(row_len, out=rs[1:])
        keep = np.arange(k)[None, :] < lens[:, None]
        nb_values.append(np.ascontiguousarray(nbrs[keep], dtype=np.int32))
        nb_row_splits.append(rs)
    enter = np.nonzero(levels >= START_LEVEL)[0].astype(np.int32)
    return {"nb_values": nb_values, "nb_row_splits": nb_row_splits, "enter_points": enter}


def make_queries(item_embs, assign, n_queries, seq_len=50, min_len=7, seed=4321):
    """UserBehavior-shaped `comm_seq`: f16[B, seq_len, d]; each history is 7..50
    items of one random cluster, zero-padded tail
    (convert_UB_to_tfrecord.py:121-137; reference feed comm_seq f16[1,50*64])."""
    rng = np.random.default_rng(seed)
    n, d = item_embs.shape
    order = np.argsort(assign, kind="stable")
    starts = np.searchsorted(assign[order], np.arange(assign.max() + 2))
    seq = np.zeros((n_queries, seq_len, d), np.float16)
    for b in range(n_queries):
        c = int(rng.integers(0, assign.max() + 1))
        lo, hi = starts[c], starts[c + 1]
        if hi <= lo:
            lo, hi = 0, n
        length = int(rng.integers(min_len, seq_len + 1))
        picks = order[rng.integers(lo, hi, size=length)]
        seq[b, :length] = item_embs[picks]
    return seq


def make_mlp_weights(d, h1=256, h2=128, seed=777):
    """x=[q;e] (2d) -> h1 -> PReLU -> h2 -> PReLU -> 1.  Init as the reference:
    N(0, 1/fan_in) kernels (model_util.py:48), bias 0.1 (:49), PReLU alpha 0.25
    (:10), last layer bias-free (model.py:218-219)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    return {
        "w1": (rng.standard_normal((2 * d, h1)) / math.sqrt(2 * d)).astype(f32),
        "b1": np.full(h1, 0.1, f32), "alpha1": np.full(h1, 0.25, f32),
        "w2": (rng.standard_normal((h1, h2)) / math.sqrt(h1)).astype(f32),
        "b2": np.full(h2, 0.1, f32), "alpha2": np.full(h2, 0.25, f32),
        "w3": (rng.standard_normal(h2) / math.sqrt(h2)).astype(f32),
    }


def mlp_forward(w, q, rows):
    """The 2d-256-128-1 PReLU MLP on f32 arrays (numpy, BLAS order -- a workload generator's view of the scorer, not a
    parity reference): q f32[d], rows [n, d] -> scores f32[n]."""
    q = np.asarray(q, np.float32)
    x = np.asarray(rows, np.float32)
    d = q.shape[0]
    h = x @ w["w1"][d:] + (q @ w["w1"][:d] + w["b1"])
    h = np.maximum(h, 0) + w["alpha1"] * np.minimum(h, 0)
    o = h @ w["w2"] + w["b2"]
    o = np.maximum(o, 0) + w["alpha2"] * np.minimum(o, 0)
    return (o @ w["w3"]).astype(np.float32)


def make_mlp_weights_metric(d, item_embs=None, h1=256, h2=128, seed=779, n_fit=4096):
    """An MLP scorer that RANKS LIKE THE INDEX METRIC -- what the reference's training produces (model.py:94-149 trains
    the scorer on the corpus the HNSW graph is later built over, so that graph neighbourhoods are score
    neighbourhoods; with random-init weights the traversal's recall against brute force under the same scorer is
    ~0.4 and the "QPS @ recall parity" metric is vacuous).  No training here: the first two layers are CONSTRUCTED,
    the last one FITTED in closed form:
      layer 1   unit j < 128: +r_j . (q - e), unit j + 128: -r_j . (q - e)  (r_j random directions, b1 = 0):
                PReLU(z) + PReLU(-z) = (1 - alpha) |z|
      layer 2   o = M (h[:128] + h[128:]) with M = I + a small positive dense matrix: every o_k >= 0 (PReLU = identity),
                a dense mixture of the |r_j . (q - e)|
      layer 3   w3 = least squares of -||q - e|| on o over (query, item) pairs sampled from the corpus
    so score(q, e) ~ -sum_j |r_j . (q - e)|, a norm of the projected difference that orders items nearly as L2 does.
    item_embs: rows to sample the fitting pairs from (f16 / f32 [n, d]); None: unit-scale Gaussian rows."""
    assert h1 == 256 and h2 == 128
    rng = np.random.default_rng(seed)
    f32 = np.float32
    r = rng.standard_normal((d, h2)).astype(f32) / f32(math.sqrt(d))            # 128 directions in R^d
    w1 = np.zeros((2 * d, h1), f32)
    w1[:d, :h2], w1[:d, h2:] = r, -r                                             # rows of q
    w1[d:, :h2], w1[d:, h2:] = -r, r                                             # rows of e
    m = np.eye(h2, dtype=f32) + (rng.random((h2, h2)).astype(f32) * f32(0.2 / h2))
    w2 = np.concatenate([m.T, m.T], axis=0).astype(f32)                          # o = (h_plus + h_minus) @ m.T
    w = {"w1": w1, "b1": np.zeros(h1, f32), "alpha1": np.full(h1, 0.25, f32), "w2": w2, "b2": np.zeros(h2, f32),
         "alpha2": np.full(h2, 0.25, f32), "w3": np.zeros(h2, f32)}
    if item_embs is None:
        x = rng.standard_normal((n_fit, d)).astype(f32) / f32(math.sqrt(d))
    else:
        x = np.asarray(item_embs[rng.integers(0, len(item_embs), size=n_fit)], f32)
    qs = x[rng.permutation(n_fit)] + (rng.standard_normal((n_fit, d)).astype(f32) * f32(0.05))
    z = qs - x
    h = z @ r
    feats = (f32(0.75) * np.abs(h)) @ m.T                                        # what layer 2 outputs for these pairs
    target = -np.sqrt((z * z).sum(1))
    w["w3"] = np.linalg.lstsq(feats.astype(np.float64), target.astype(np.float64), rcond=None)[0].astype(f32)
    return w


def make_attn_weights(d, E=64, h=(128, 64, 32), seed=778):
    """Seeded weights of the reference's scorer model (model.py:189-233, model_util.py:70-97) with
    the reference's initialisers' scales: attention dense layers glorot-uniform with zero bias
    (tf.layers.dense defaults), DNN layers variance-scaling(fan_in) normal with bias 0.1
    (model_util.py:47-48), PReLU slopes around 0.25 (:10), and non-trivial batch-norm statistics,
    folded (scale = gamma / sqrt(var + 1e-3), shift = beta - mean * scale)."""
    rng = np.random.default_rng(seed)

    def glorot(n_in, n_out):
        lim = math.sqrt(6.0 / (n_in + n_out))
        return rng.uniform(-lim, lim, size=(n_in, n_out)).astype(np.float32)

    def small(n, centre=0.0, spread=0.05):
        return (centre + spread * rng.standard_normal(n)).astype(np.float32)

    w = {"wq1": glorot(d, 2 * E), "bq1": small(2 * E), "aq": small(2 * E, 0.25),
         "wq2": glorot(2 * E, 4 * E), "bq2": small(4 * E),
         "wk1": glorot(E, 2 * E), "bk1": small(2 * E), "ak": small(2 * E, 0.25),
         "wk2": glorot(2 * E, 4 * E), "bk2": small(4 * E),
         "w": [], "b": [], "bn_scale": [], "bn_shift": [], "alpha": []}
    n_in = E + d
    for n_out in h:
        w["w"].append((rng.standard_normal((n_in, n_out)) / math.sqrt(n_in)).astype(np.float32))
        w["b"].append(small(n_out, 0.1, 0.02))
        gamma, beta = small(n_out, 1.0, 0.1), small(n_out, 0.0, 0.1)
        mean, var = small(n_out, 0.0, 0.3), (0.5 + rng.random(n_out)).astype(np.float32)
        scale = (gamma / np.sqrt(var + np.float32(1e-3))).astype(np.float32)
        w["bn_scale"].append(scale)
        w["bn_shift"].append((beta - mean * scale).astype(np.float32))
        w["alpha"].append(small(n_out, 0.25))
        n_in = n_out
    w["w"].append((rng.standard_normal(n_in) / math.sqrt(n_in)).astype(np.float32))
    return w


def make_index(n_items, d, ef, m=M_DEFAULT, device=None, mode="hnsw", seed=1234,
               n_clusters=256, noise=0.5, shard=0):
    """One call: corpus + ids + graph.  Guarantees E >= ef."""
    embs, assign = make_corpus(n_items, d, n_clusters=n_clusters, noise=noise, seed=seed,
                               item_seed=seed + 100 + 1000 * shard)
    ids = make_item_ids(n_items, seed=seed + 1 + 1000 * shard) + shard * n_items
    levels = assign_levels(n_items, m=m, seed=seed + 2 + 1000 * shard, min_enter=ef)
    g = build_graph(embs, levels, m=m, device=device, mode=mode)
    g.update({"item_embs": embs, "item_ids": ids, "assign": assign, "levels": levels})
    return g


def save_index(g, out_dir):
    """Write the reference's .npy files (npy format 1.0, C order)."""
    import os
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "item_embs.npy"), g["item_embs"])
    np.save(os.path.join(out_dir, "item_ids.npy"), g["item_ids"])
    np.save(os.path.join(out_dir, "enter_points.npy"), g["enter_points"].astype(np.int64))
    for l in range(START_LEVEL):
        np.save(os.path.join(out_dir, f"neighbors_level_{l}_values.npy"), g["nb_values"][l])
        np.save(os.path.join(out_dir, f"neighbors_level_{l}_row_splits.npy"),
                g["nb_row_splits"][l])
