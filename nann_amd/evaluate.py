"""Recall harness of the reference's evaluation jobs on the HIP ops.

Mirrors NANN_impls/main.py `--job-type test` (:144-188: HNSW retrieval with the eval-graph traversal
`Model.retrieval`, model.py:299-362) and `test_all` (:194-237: brute force over every item), with the
reference's metric `calc_pr` (nann/util.py:14-25: ONE ground-truth item per user; precision, recall, F1 of
a retrieved id list).  Inputs the reference reads from TFRecords / checkpoints are arguments here:
`user_seqs` (the `comm_seq` feed, f16 [U, L, E]) and `ground_truths` (i64 [U], item ids).
"""
from collections import defaultdict

import numpy as np
import torch

from . import ops, retrieval


def calc_pr(ground_truth, retrievals):
    """nann/util.py:14-25."""
    ground_truths = {int(ground_truth)}
    retrievals = set(int(x) for x in retrievals)
    hit_num = len(ground_truths & retrievals)
    p = hit_num * 1.0 / len(retrievals)
    r = hit_num * 1.0 / len(ground_truths)
    f1 = 2 * p * r / (p + r) if p + r > 0 else 0.0
    return p, r, f1


class AverageMeter:
    """nann/util.py:28-59, the part the evaluation uses."""

    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n

    @property
    def avg(self):
        return self.sum / max(self.count, 1)


def _query(scorer, seq):
    """the per-user input of an ops.Scorer (l2 / mlp): mean of the non-pad history rows"""
    if isinstance(scorer, ops.Model):
        raise NotImplementedError("an ops.Model takes the raw sequence (Model.forward); only ops.Scorer has a query vector")
    return ops.user_seq_mean(seq[None])[0]


def _score_all(index, scorer, seq):
    """logits of EVERY item for one user: ops.Model -> forward() on the raw sequence (the attention model's
    input is [L, 64], not a mean); ops.Scorer -> blaze_score on the mean of the history rows"""
    if isinstance(scorer, ops.Model):
        return scorer.forward(seq[None], index.item_embs).reshape(-1)
    return ops.blaze_score(scorer, _query(scorer, seq), item_emb=index.item_embs)


def test(index, scorer, user_seqs, ground_truths, topk_eval=(200,), num_scoring_per_level=(3, 1, 1),
         top_k_per_level=(400, 200, 100), num_test_batch=None, fused=True):
    """main.py:144-188 -> {"precision" | "recall" | "f1": {topk: meter}}.  Retrieval = Model.retrieval's
    traversal (threshold frontier, min(k, n) guard, per-level round counts): all users in ONE kernel
    (retrieval.search_eval, the default) or op by op (retrieval.search_eval_per_op, fused=False).  `scorer`:
    ops.Scorer (the query is the mean of the user's history rows) or ops.Model (user_seqs go in as they are)."""
    n = len(ground_truths) if num_test_batch is None else min(num_test_batch, len(ground_truths))
    prec, rec, f1m = defaultdict(AverageMeter), defaultdict(AverageMeter), defaultdict(AverageMeter)
    seqs = torch.as_tensor(np.asarray(user_seqs)).to(index.device)[:n]
    kmax = max(topk_eval)
    if not fused and isinstance(scorer, ops.Model):
        raise NotImplementedError("the op-by-op spelling scores through blaze_score (ops.Scorer); "
                                  "an ops.Model runs on the fused kernel (fused=True)")
    if fused:
        q = seqs if isinstance(scorer, ops.Model) else ops.user_seq_mean(seqs)
        r = retrieval.search_eval(index, scorer, q, num_scoring_per_level, top_k_per_level, kmax)
        status, n_out, all_ids = r.status.cpu().numpy(), r.n_out.cpu().numpy(), r.item_ids.cpu().numpy()
        if status.any():
            raise RuntimeError(f"eval traversal failed for users {np.nonzero(status)[0][:8].tolist()}: "
                               f"status {status[status != 0][:8].tolist()}")
    for u in range(n):
        if fused:
            ids = all_ids[u, :n_out[u]]
        else:
            ids, _, _ = retrieval.search_eval_per_op(index, scorer, _query(scorer, seqs[u]), num_scoring_per_level,
                                                     top_k_per_level, kmax)
            ids = ids.cpu().numpy()
        for k in topk_eval:
            assert ids.shape[0] >= k  # main.py:169
            p, r_, f = calc_pr(ground_truths[u], ids[:k])
            prec[k].update(p); rec[k].update(r_); f1m[k].update(f)
    return {"precision": prec, "recall": rec, "f1": f1m}


def test_all(index, scorer, user_seqs, ground_truths, topk_eval=(200,), num_test_batch=None):
    """main.py:194-237: score EVERY item for each user, take the top max(topk_eval) (fast_argtopk,
    util.py:9-11) -> the recall ceiling of the scorer itself."""
    n = len(ground_truths) if num_test_batch is None else min(num_test_batch, len(ground_truths))
    prec, rec, f1m = defaultdict(AverageMeter), defaultdict(AverageMeter), defaultdict(AverageMeter)
    seqs = torch.as_tensor(np.asarray(user_seqs)).to(index.device)
    for u in range(n):
        scores_all = _score_all(index, scorer, seqs[u])
        _, idx = ops.top_k(scores_all, max(topk_eval))
        ids = index.item_ids[idx.long()].cpu().numpy()
        for k in topk_eval:
            p, r, f = calc_pr(ground_truths[u], ids[:k])
            prec[k].update(p); rec[k].update(r); f1m[k].update(f)
    return {"precision": prec, "recall": rec, "f1": f1m}


def recall_vs_bruteforce(index, scorer, q, level_topn, n_queries=None):
    """How much of the brute-force top-k (same scorer) the serving-graph traversal (nann_search) returns:
    the recall@k figure bench.py reports."""
    r = retrieval.search(index, scorer, q, level_topn, want_counters=False)
    k = int(level_topn[5])
    st, got = r.status.cpu().numpy(), r.index.cpu().numpy()
    hits = total = 0
    for b in range(q.shape[0] if n_queries is None else min(n_queries, q.shape[0])):
        if st[b]:
            continue
        _, bi = ops.top_k(ops.blaze_score(scorer, q[b], item_emb=index.item_embs), k)
        hits += len(set(bi.cpu().tolist()) & set(got[b].tolist()))
        total += k
    return hits / max(total, 1)
