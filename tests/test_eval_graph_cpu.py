"""CPU: the host logic of retrieval.search_eval_per_op (the eval-graph traversal, SURVEY.md 8 f3:
model.py:299-362) against the oracle's restatement of the same schedule.  The HIP ops are
replaced by a stand-in built from the oracle's per-op functions, so what is checked here is the
composition (frontier rule, min(k, n) guard, ascending neighbour sets, visited handling); the ops
themselves are checked against the oracle one by one in the -m gpu tests, and
test_search_gpu.py::test_eval_graph_matches_oracle runs the same function on the HIP ops."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch


class OracleOps:
    """group_gather / bitmap_ref_difference / blaze_score / top_k / gather of nann_amd.ops with CPU
    tensors, computed by the oracle."""

    def __init__(self, oracle):
        self.O = oracle

    def group_gather(self, pv, prs, iv, irs, unique=False):
        rc, _, out, ors = self.O.group_gather(pv.numpy(), prs.numpy(), iv.numpy(), irs.numpy())
        assert rc == 0
        return torch.as_tensor(out), torch.as_tensor(ors)

    def bitmap_ref_difference(self, v, rs, flags):
        rc, _, out, ors = self.O.bitmap_ref_difference(v.numpy(), rs.numpy(), flags.numpy())  # in place
        assert rc == 0
        return torch.as_tensor(out), torch.as_tensor(ors), flags

    def blaze_score(self, scorer, q, table=None, indices=None):
        rc, out = self.O.score_rows(scorer, q.numpy(), table.numpy()[indices.numpy().astype(np.int64)])
        assert rc == 0
        return torch.as_tensor(out)

    def top_k(self, values, k):
        rc, ov, oi = self.O.topk(values.numpy(), k)
        assert rc == 0
        return torch.as_tensor(ov), torch.as_tensor(oi)

    def gather(self, params, indices):
        rc, out, _ = self.O.gather_rows(params.numpy(), indices.numpy())
        assert rc == 0
        return torch.as_tensor(out)


def cpu_index(z):
    n = z["item_embs"].shape[0]
    return SimpleNamespace(
        item_embs=torch.as_tensor(z["item_embs"]), item_ids=torch.as_tensor(z["item_ids"].astype(np.int64)),
        nb_values=[torch.as_tensor(z[f"nb_values_{l}"].astype(np.int32)) for l in (0, 1)],
        nb_row_splits=[torch.as_tensor(z[f"nb_row_splits_{l}"].astype(np.int64)) for l in (0, 1)],
        enter_points=torch.as_tensor(z["enter_points"].astype(np.int32)), bitmap_words=(n + 31) // 32)


@pytest.mark.parametrize("name", ["small_l2_d64.npz", "small_l2_d128.npz"])
@pytest.mark.parametrize("num_scoring,top_k_per_level,topk_eval", [
    ((3, 1, 1), (400, 200, 100), 200),   # config.py:50-58
    ((2, 2, 1), (50, 30, 10), 20),
    ((1, 1, 1), (5000, 5000, 5000), 64),  # every k above what exists: the min(k, n) guard
    ((3, 1, 1), (16, 8, 4), 200),         # topk_eval above the result size
    ((40, 12, 1), (5000, 5000, 5000), 200),  # more rounds than the graph has nodes to offer: the frontier runs
                                             # dry and the eval graph carries on with empty tensors (no error)
])
def test_eval_graph_composition_matches_oracle(oracle, golden_dir, name, num_scoring, top_k_per_level, topk_eval):
    from nann_amd import retrieval
    z = dict(np.load(os.path.join(golden_dir, name)))
    d = z["item_embs"].shape[1]
    oix = oracle.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                       [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    osc = oracle.Scorer("l2", d, oracle.EMB_F16)
    ix = cpu_index(z)
    backend = OracleOps(oracle)
    n_ok = 0
    for q in z["q"][:12]:
        rc, eids, esc, eidx = oracle.search_eval(oix, osc, q, num_scoring, top_k_per_level, topk_eval)
        if rc:
            with pytest.raises(Exception):
                retrieval.search_eval_per_op(ix, osc, torch.as_tensor(q), num_scoring, top_k_per_level,
                                             topk_eval, backend=backend)
            continue
        ids, sc, idx = retrieval.search_eval_per_op(ix, osc, torch.as_tensor(q), num_scoring, top_k_per_level,
                                                    topk_eval, backend=backend)
        assert (idx.numpy() == eidx).all() and (ids.numpy() == eids).all()
        assert (sc.numpy().view(np.uint32) == esc.view(np.uint32)).all()
        n_ok += 1
    assert n_ok > 0
