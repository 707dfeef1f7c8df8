"""CPU: the oracle against the known answers of the reference's own op tests
(tests/golden/reference_ops.json) and against the committed regression vectors."""
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref_ops(golden_dir):
    with open(os.path.join(golden_dir, "reference_ops.json")) as f:
        return json.load(f)


def test_group_gather_known_answers(oracle, ref_ops):
    for case in ref_ops["group_gather"]:
        rc, code, vals, rs = oracle.group_gather(case["params_values"], case["params_row_splits"],
                                                 case["indices_values"], case["indices_row_splits"])
        assert rc == case["status"], case["name"]
        if rc:
            assert code == case["ragged_code"], case["name"]
            continue
        assert vals.tolist() == case["ret_values"], case["name"]
        assert rs.tolist() == case["ret_row_splits"], case["name"]


def test_bitmap_ref_difference_known_answers(oracle, ref_ops):
    for case in ref_ops["bitmap_ref_difference"]:
        bm = np.zeros(case["bitmap_words"], np.int32)
        for call in case["calls"]:
            rc, _, vals, rs = oracle.bitmap_ref_difference(call["values"], call["row_splits"], bm)
            assert rc == 0, case["name"]
            assert vals.tolist() == call["c_values"], case["name"]
            assert rs.tolist() == call["c_row_splits"], case["name"]
        assert bm.tolist() == case["final_flags"], case["name"]


def test_ragged_validation_codes(oracle):
    # GroupGather_kernel.cc:9-16
    rc, code, _, _ = oracle.group_gather([1, 2], [0, 1], [0], [0, 1])  # last split != n_values
    assert (rc, code) == (oracle.ERR_INVALID_RAGGED_PARAMS, 3)
    rc, code, _, _ = oracle.group_gather([1, 2], [0, 2], [0], [1, 1])  # indices: first split != 0
    assert (rc, code) == (oracle.ERR_INVALID_RAGGED_INDICES, 2)
    bm = np.zeros(2, np.int32)
    rc, code, _, _ = oracle.bitmap_ref_difference([1, 2, 3], [0, 2], bm)
    assert (rc, code) == (oracle.ERR_INVALID_RAGGED_INPUT, 3)
    assert not bm.any()


def test_bitmap_out_of_range_leaves_bitmap_untouched(oracle):
    bm = np.zeros(2, np.int32)
    rc, _, _, _ = oracle.bitmap_ref_difference([1, 64], [0, 2], bm)
    assert rc == oracle.ERR_INDEX_OUT_OF_RANGE
    assert not bm.any()


def test_topk_order_and_errors(oracle):
    # value desc, ties -> lower index (topk_op.cc:134-142)
    rc, v, i = oracle.topk([1, 3, 3, 2, 5, 5], 3)
    assert rc == 0 and i.tolist() == [4, 5, 1] and v.tolist() == [5, 5, 3]
    rc, v, i = oracle.topk([0.0, -0.0, 0.0], 3)  # -0 == +0 for the comparator
    assert i.tolist() == [0, 1, 2]
    rc, v, i = oracle.topk([2, 7, 7], 1)  # k == 1: first maximum (:112-130)
    assert i.tolist() == [1]
    rc, v, i = oracle.topk([4, 1, 4, 4], 4)  # k == n (:154-173)
    assert i.tolist() == [0, 2, 3, 1]
    rc, _, _ = oracle.topk([1, 2], 3)
    assert rc == oracle.ERR_TOPK_K_GT_N
    rng = np.random.default_rng(0)
    x = rng.integers(0, 50, size=5000).astype(np.float32)  # many ties
    rc, v, i = oracle.topk(x, 300)
    order = np.lexsort((np.arange(len(x)), -x))[:300]
    assert i.tolist() == order.tolist()


def test_gather_rows(oracle):
    p = np.arange(40, dtype=np.float16).reshape(10, 4)
    rc, out, bad = oracle.gather_rows(p, [3, 3, 0, 9])
    assert rc == 0 and (out == p[[3, 3, 0, 9]]).all()
    rc, out, bad = oracle.gather_rows(p, [3, 10, 11])
    assert rc == oracle.ERR_INDEX_OUT_OF_RANGE and bad == 1


def test_half_conversion_exact(oracle):
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([L.oracle_half_to_float(int(h)) for h in allh[::7]], np.float32)
    exp = ref[::7]
    ok = (got == exp) | (np.isnan(got) & np.isnan(exp))
    assert ok.all()
    vals = np.random.default_rng(1).standard_normal(2000).astype(np.float32) * 3
    back = np.array([L.oracle_float_to_half(float(v)) for v in vals], np.uint16)
    assert (back == vals.astype(np.float16).view(np.uint16)).all()


def test_l2_scorer_matches_float64(oracle):
    rng = np.random.default_rng(2)
    for d in (64, 128, 256):
        rows = (rng.standard_normal((50, d)) / np.sqrt(d)).astype(np.float16)
        q = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
        rc, s = oracle.score_rows(oracle.Scorer("l2", d, oracle.EMB_F16), q, rows)
        ref = -((q.astype(np.float64) - rows.astype(np.float64)) ** 2).sum(1)
        assert rc == 0
        np.testing.assert_allclose(s, ref, rtol=1e-5)
    rc, _ = oracle.score_rows(oracle.Scorer("l2", 64, oracle.EMB_F16), np.zeros(64, np.float32),
                              np.zeros((0, 64), np.float16))
    assert rc == oracle.ERR_EMPTY_SCORE_BATCH  # blaze_xla_predictor.cc:259-263


def test_mlp_scorer_matches_float64(oracle):
    from nann_amd import synth
    rng = np.random.default_rng(3)
    d = 128
    w = synth.make_mlp_weights(d)
    w["alpha1"] = np.full(256, 0.25, np.float32); w["alpha2"] = np.full(128, 0.1, np.float32)
    rows = (rng.standard_normal((40, d)) / np.sqrt(d)).astype(np.float16)
    q = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
    rc, s = oracle.score_rows(oracle.Scorer("mlp", d, oracle.EMB_F16, w), q, rows)
    assert rc == 0
    x = np.concatenate([np.tile(q, (40, 1)), rows.astype(np.float32)], 1).astype(np.float64)
    prelu = lambda z, a: np.maximum(z, 0) + a * np.minimum(z, 0)  # model_util.py:9-11
    h1 = prelu(x @ w["w1"] + w["b1"], w["alpha1"])
    h2 = prelu(h1 @ w["w2"] + w["b2"], w["alpha2"])
    np.testing.assert_allclose(s, h2 @ w["w3"], rtol=1e-5, atol=1e-6)


def test_user_seq_mean(oracle):
    rng = np.random.default_rng(4)
    seq = np.zeros((50, 64), np.float16)
    seq[:13] = rng.standard_normal((13, 64)).astype(np.float16)
    q = oracle.user_seq_mean(seq)
    np.testing.assert_allclose(q, seq[:13].astype(np.float64).mean(0), rtol=1e-5, atol=1e-6)
    assert (oracle.user_seq_mean(np.zeros((50, 64), np.float16)) == 0).all()


@pytest.mark.parametrize("name", ["small_l2_d64.npz", "small_l2_d128.npz"])
def test_regression_vectors(oracle, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    ix = oracle.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                      [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    d = z["item_embs"].shape[1]
    sc = oracle.Scorer("l2", d, oracle.EMB_F16)
    q = np.stack([oracle.user_seq_mean(s) for s in z["comm_seq"]])
    assert (q.view(np.uint32) == z["q"].view(np.uint32)).all()
    st, ids, scores, idx, ctr = oracle.search_batch(ix, sc, q, z["level_topn"], n_threads=2)
    assert (st == z["status"]).all()
    assert (ids == z["out_item_ids"]).all()
    assert (scores.view(np.uint32) == z["out_scores"].view(np.uint32)).all()
    assert (ctr == z["counters"]).all()
    # single-query entry point agrees with the batch one
    rc, ids0, sc0, idx0, c0 = oracle.search(ix, sc, q[0], z["level_topn"])
    assert rc == 0 and (ids0 == ids[0]).all()


def test_sibling_ops(oracle, ref_ops):
    """BatchTopKOnRT against the reference test's known answers; BitmapInit /
    BitmapDifference against the Ref variant they are defined by."""
    for case in ref_ops["batch_topk_on_rt"]:
        rc, v, i, rs = oracle.batch_topk_on_rt(case["values"], case["row_splits"], case["k"], case["ascending"])
        assert rc == 0, case["name"]
        assert v.tolist() == case["values_out"] and i.tolist() == case["idx_out"], case["name"]
        assert rs.tolist() == case["row_splits_out"], case["name"]
    rc, bm = oracle.bitmap_init([1, 1, 2, 33, 63], 5)
    assert rc == 0 and bm.tolist() == [6, -2147483646, 0, 0, 0]
    rc, _ = oracle.bitmap_init([1, 2, 3], 2)  # n_idx > length (bitmap_ops.cc:56-57)
    assert rc == oracle.ERR_BAD_ARGUMENT
    flags = np.array([32254, 0, 0, 0], np.int32)
    rc, out, fnew = oracle.bitmap_difference([4, 40, 40, 6, 41], flags)
    assert rc == 0 and out.tolist() == [40, 41] and flags.tolist() == [32254, 0, 0, 0]
    assert fnew.tolist() == [32254, 768, 0, 0]


def test_group_gather_unique_is_the_per_group_set(oracle, ref_ops):
    """GroupGather_kernel.cc:91-131: unique=true = per group the set of the gathered values (any order is the
    reference's answer: unordered_set); the restatement emits first-occurrence order."""
    for case in ref_ops["group_gather"]:
        if case["status"]:
            continue
        args = (case["params_values"], case["params_row_splits"], case["indices_values"], case["indices_row_splits"])
        rc, _, v, rs = oracle.group_gather(*args, unique=True)
        assert rc == 0
        plain, prs = case["ret_values"], case["ret_row_splits"]
        assert len(rs) == len(prs)
        for g in range(len(prs) - 1):
            mine = v[rs[g]:rs[g + 1]].tolist()
            ref = plain[prs[g]:prs[g + 1]]
            assert set(mine) == set(ref) and len(set(mine)) == len(mine)
            assert mine == list(dict.fromkeys(ref))  # first-occurrence order
    # the docstring example (group_gather_test.py:7-10): rows [[0,1,1,2,3,4],[3,4,5,5,6],[7,8,8,9],[10,11,12]], groups [[0,1],[3]]
    rc, _, v, rs = oracle.group_gather([0, 1, 1, 2, 3, 4, 3, 4, 5, 5, 6, 7, 8, 8, 9, 10, 11, 12], [0, 6, 11, 15, 18], [0, 1, 3], [0, 2, 3],
                                       unique=True)
    assert rc == 0 and v.tolist() == [0, 1, 2, 3, 4, 5, 6, 10, 11, 12] and rs.tolist() == [0, 7, 10]
