"""-m gpu: the fused traversal (nann_search) and the op-by-op schedule against the
CPU oracle: bit-exact neighbour indices, top-k item ids, scores and per-round
counters; per-query failure codes; independence from batch size."""
import os

import numpy as np
import pytest
import torch

from gpu_util import MODES, bits, cuda, queries_for, require_gpu, synth_index, tolerant_parity, traversal_mode

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()


def _run(dix, q, topn, mode="auto"):
    from nann_amd import ops, retrieval
    sc = ops.Scorer("l2", dix.d, dix.item_embs.dtype)
    with traversal_mode(mode):
        r = retrieval.search(dix, sc, cuda(q), topn)
        torch.cuda.synchronize()
    return (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
            r.index.cpu().numpy(), r.counters.cpu().numpy())


def _assert_same(got, exp):
    st, ids, scores, idx, ctr = got
    est, eids, escores, eidx, ectr = exp
    assert (st == est).all(), (st, est)
    ok = est == 0
    assert (idx[ok] == eidx[ok]).all()
    assert (ids[ok] == eids[ok]).all()
    assert (bits(scores[ok]) == bits(escores[ok])).all()
    assert (ctr[ok] == ectr[ok]).all()
    assert (ids[~ok] == 0).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["small_l2_d64.npz", "small_l2_d128.npz"])
def test_committed_vectors(golden_dir, name, mode):
    from nann_amd import ops, retrieval
    z = np.load(os.path.join(golden_dir, name))
    dix = retrieval.Index(z["item_embs"], z["item_ids"], [z["nb_values_0"], z["nb_values_1"]],
                          [z["nb_row_splits_0"], z["nb_row_splits_1"]], z["enter_points"])
    q = ops.user_seq_mean(cuda(z["comm_seq"])).cpu().numpy()
    assert (bits(q) == bits(z["q"])).all()
    got = _run(dix, q, z["level_topn"], mode)
    _assert_same(got, (z["status"], z["out_item_ids"], z["out_scores"], z["out_index"], z["counters"]))


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,d,ef,k,nq", [(20000, 64, 32, 20, 300), (60000, 128, 64, 50, 64)])
def test_search_matches_oracle(oracle, n, d, ef, k, nq, mode):
    g, oix, dix = synth_index(n, d, ef)
    seq = queries_for(g, nq)
    q = np.stack([oracle.user_seq_mean(s) for s in seq])
    topn = [ef] * 5 + [k]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", d, oracle.EMB_F16), q, topn, n_threads=8)
    assert (exp[0] == 0).mean() > 0.5, "workload should be mostly valid requests"
    got = _run(dix, q, topn, mode)
    _assert_same(got, exp)
    # uneven level_topn, as in the reference's own benchmark feed (gen_runmeta.py:23)
    topn2 = [ef // 2, ef, 2 * ef, 2 * ef, 2 * ef, k]
    exp2 = oracle.search_batch(oix, oracle.Scorer("l2", d, oracle.EMB_F16), q[:40], topn2, n_threads=8)
    _assert_same(_run(dix, q[:40], topn2, mode), exp2)


@pytest.mark.parametrize("mode", ["lds_hash", "lds_bitmap"])
def test_batch_size_independence(oracle, mode):
    g, oix, dix = synth_index(20000, 64, 32)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 1400, seed=99)])
    topn = [32] * 5 + [20]
    full = _run(dix, q, topn, mode)  # more queries than workgroup slots: slots are reused
    for b in (0, 255, 256, 511, 512, 1399):
        one = _run(dix, q[b:b + 1], topn, mode)
        assert (one[1][0] == full[1][b]).all() and (bits(one[2][0]) == bits(full[2][b])).all()


@pytest.mark.parametrize("mode", MODES)
def test_failing_requests_get_reference_codes(oracle, mode):
    g, oix, dix = synth_index(20000, 64, 32)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 8, seed=5)])
    E = len(g["enter_points"])
    sc = oracle.Scorer("l2", 64, oracle.EMB_F16)
    for topn in ([E + 1, 8, 8, 8, 8, 8], [8, 8, 8, 8, 8, 33], [0, 8, 8, 8, 8, 8], [8, 1000, 8, 8, 8, 8]):
        exp = oracle.search_batch(oix, sc, q, topn)
        assert (exp[0] != 0).all()
        got = _run(dix, q, topn, mode)
        assert (got[0] == exp[0]).all(), (topn, got[0], exp[0])
        assert (got[1] == 0).all()


def test_per_op_schedule_matches_oracle(oracle):
    """build_model() spelled with the drop-in ops, one query at a time."""
    from nann_amd import ops, retrieval
    g, oix, dix = synth_index(20000, 64, 32)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 6, seed=77)])
    topn = [32] * 5 + [20]
    sc = ops.Scorer("l2", 64)
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn)
    for b in range(6):
        if exp[0][b]:
            with pytest.raises(ops.NannError) as e:
                retrieval.search_per_op(dix, sc, cuda(q[b]), topn)
            assert e.value.status == exp[0][b]
            continue
        ids, scores, idx = retrieval.search_per_op(dix, sc, cuda(q[b]), topn)
        assert (ids.cpu().numpy() == exp[1][b]).all()
        assert (idx.cpu().numpy() == exp[3][b]).all()
        assert (bits(scores.cpu().numpy()) == bits(exp[2][b])).all()


@pytest.mark.parametrize("mode", MODES)
def test_exact_ties_follow_position_order(oracle, mode):
    """Duplicate embeddings give exactly equal scores: TopKV2's lower-position rule
    (topk_op.cc:134-142) must decide, which depends on the serial first-occurrence
    order of BitmapRefDifference."""
    from nann_amd import retrieval, synth
    g = synth.make_index(6000, 64, ef=16, seed=5, noise=1.0, n_clusters=8, device="cuda")
    embs = g["item_embs"].copy()
    embs[1::2] = embs[0::2]  # every item has an identical twin
    g["item_embs"] = embs
    oix = oracle.Index(embs, g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    dix = retrieval.Index.from_dict(g)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 40, seed=3)])
    topn = [16] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn)
    _assert_same(_run(dix, q, topn, mode), exp)


def test_mlp_search_matches_oracle(oracle):
    """BASELINE configs[2] shape at test size: the traversal with the 256-128-1 MLP scorer."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 128, 32)
    w = synth.make_mlp_weights(128)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 48, seed=11)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("mlp", 128, oracle.EMB_F16, w), q, topn, n_threads=8)
    sc = ops.Scorer("mlp", 128, torch.float16, w, precision="exact")
    r = retrieval.search(dix, sc, cuda(q), topn)
    torch.cuda.synchronize()
    got = (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
           r.index.cpu().numpy(), r.counters.cpu().numpy())
    assert (exp[0] == 0).mean() > 0.5
    _assert_same(got, exp)


@pytest.mark.parametrize("precision", ["exact", "split"])
def test_mlp_pipeline_of_phases_chunks_and_fused_agree(oracle, precision):
    """The MLP traversal as a pipeline of phases (nann_mlp6.h) against the fused kernel (nann_mlp5.h) on the same
    queries.  1100 queries in one call: exact f32 = two chunks of the pipeline (1024 + 76), split-f16 = the fused
    kernel (above 160 queries); the same queries in calls of 100: the pipeline for both.  Results do not depend on
    the batch size or the form bit for bit (both forms run the same block loop), and a sample equals the oracle
    (exact: bit for bit, counters included)."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 128, 32)
    w = synth.make_mlp_weights(128)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 1100, seed=21)])
    topn = [32] * 5 + [20]
    sc = ops.Scorer("mlp", 128, torch.float16, w, precision=precision)

    def run(qq):
        r = retrieval.search(dix, sc, cuda(qq), topn)
        torch.cuda.synchronize()
        return (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(), r.index.cpu().numpy(),
                r.counters.cpu().numpy())

    whole = run(q)
    parts = [run(q[i:i + 100]) for i in range(0, 1100, 100)]
    cat = [np.concatenate([p_[j] for p_ in parts]) for j in range(5)]
    assert (whole[0] == cat[0]).all() and (whole[0] == 0).mean() > 0.5
    assert (whole[1] == cat[1]).all() and (bits(whole[2]) == bits(cat[2])).all() and (whole[4] == cat[4]).all()
    sample = np.r_[0:24, 1020:1044, 1090:1100]  # both chunks and the seam
    exp = oracle.search_batch(oix, oracle.Scorer("mlp", 128, oracle.EMB_F16, w), q[sample], topn, n_threads=8)
    if precision == "exact":
        _assert_same(tuple(a[sample] for a in whole), exp)
    else:
        assert (whole[0][sample] == exp[0]).all()
        kinds = [tolerant_parity(whole[3][b], whole[2][b], exp[3][i], exp[2][i]) for i, b in enumerate(sample) if exp[0][i] == 0]
        assert kinds.count("diverged") <= max(1, len(kinds) // 16), kinds


@pytest.mark.parametrize("mode", ["auto", "lds_hash", "lds_bitmap", "hbm_bitmap"])
def test_mlp_split_f16_search_within_tolerance(oracle, mode):
    """(auto = the 16K-slot hash set with one 512-thread workgroup per CU; the bitmap kernels are its overflow
    fallback and what a forced mode runs.)  The traversal with the split-f16 MLP (scores within 1e-5 of fp32, not bit-identical): status codes
    equal, every query's result either identical to the oracle's or different only where scores tie within
    the tolerance; a traversal that went another way because a near-tie fell differently at a beam
    boundary is tolerated for a small fraction of queries and must still overlap the oracle's answer."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(60000, 128, 64)
    w = synth.make_mlp_weights(128)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 96, seed=13)])
    topn = [64] * 5 + [50]
    est, eids, esc, eidx, ectr = oracle.search_batch(oix, oracle.Scorer("mlp", 128, oracle.EMB_F16, w), q, topn, n_threads=8)
    with traversal_mode(mode):
        r = retrieval.search(dix, ops.Scorer("mlp", 128, torch.float16, w, precision="split"), cuda(q), topn)
        torch.cuda.synchronize()
    st, idx, sc = r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()
    ok = est == 0
    assert ok.mean() > 0.5
    kinds = [tolerant_parity(idx[b], sc[b], eidx[b], esc[b]) for b in np.nonzero(ok & (st == 0))[0]]
    n_div = kinds.count("diverged")
    print("split-f16 MLP traversal vs oracle:", {k: kinds.count(k) for k in ("exact", "near-tie", "diverged")})
    assert (st == est).sum() >= len(st) - n_div
    assert kinds.count("exact") >= 0.9 * len(kinds), kinds
    assert n_div <= 1, kinds  # (round 2 allowed 5 %; measured on every plan: none)
    for b in np.nonzero(ok & (st == 0))[0]:
        assert len(set(idx[b].tolist()) & set(eidx[b].tolist())) >= 0.9 * topn[5]


@pytest.mark.parametrize("mode", MODES)
def test_long_rows_with_repeated_ids(oracle, mode):
    """A CSR whose rows exceed 64 ids and repeat ids inside a row (legal input for
    GroupGather / BitmapRefDifference, cf. group_gather_test.py's [0,1,1,2,...]): takes the
    flat 64-ids-per-step walker with its in-step duplicate resolution instead of the
    row-per-step one."""
    from nann_amd import retrieval
    g, _, _ = synth_index(20000, 64, 32)
    rng = np.random.default_rng(8)
    g2 = dict(g)
    nbv, nbrs = [], []
    for level in (0, 1):
        v, rs = g["nb_values"][level], g["nb_row_splits"][level]
        rows = []
        for i in range(len(rs) - 1):
            row = v[rs[i]:rs[i + 1]]
            if len(row) and level == 0:
                hop2 = np.concatenate([v[rs[j]:rs[j + 1]] for j in row[:6]])  # neighbours of neighbours
                row = np.concatenate([row, row[:3], hop2])[:150]               # > 64 ids, with repeats
            rows.append(row)
        nbv.append(np.concatenate(rows).astype(np.int32))
        nbrs.append(np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64))
    g2["nb_values"], g2["nb_row_splits"] = nbv, nbrs
    assert np.diff(nbrs[0]).max() > 64
    oix = oracle.Index(g2["item_embs"], g2["item_ids"], nbv, nbrs, g2["enter_points"])
    dix = retrieval.Index.from_dict(g2)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 64, seed=21)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn, n_threads=8)
    assert (exp[0] == 0).mean() > 0.5
    _assert_same(_run(dix, q, topn, mode), exp)


def test_index_from_reference_files(oracle, tmp_path):
    """The index as build_hnsw_index.py leaves it on disk (f32 embeddings, int64 neighbour
    values and enter points) -> HugeConst loads with build_model()'s casts -> same results as
    the oracle on the cast arrays."""
    from nann_amd import retrieval, synth
    g, _, _ = synth_index(20000, 64, 32)
    d = str(tmp_path)
    disk = dict(g)
    disk["item_embs"] = g["item_embs"].astype(np.float32)      # extract_feature writes f32
    disk["nb_values"] = [v.astype(np.int64) for v in g["nb_values"]]  # build_hnsw_index.py:66
    synth.save_index(disk, d)
    dix = retrieval.Index.from_files(d, d)
    assert dix.item_embs.dtype == torch.float16 and dix.nb_values[0].dtype == torch.int32
    oix = oracle.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 32, seed=4)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn, n_threads=8)
    _assert_same(_run(dix, q, topn), exp)


def test_cpp_serving_host_over_the_c_abi(oracle, tmp_path):
    """f4 in C++: csrc/host/nann_serve.cpp sees only include/nann_hip.h and libnann_hip.so -- it loads the index
    files through nann_huge_const_load, batches the single requests of closed-loop client threads into
    nann_search launches (the job of blaze-benchmark's consumers, predict_request_consumer.cc:17-53) and reports
    throughput and latency.  Its reply to one fixed request equals the oracle's answer for that request."""
    from nann_amd import serving, synth
    g, oix, _ = synth_index(20000, 64, 32)
    d = str(tmp_path)
    disk = dict(g)
    disk["item_embs"] = g["item_embs"].astype(np.float32)
    disk["nb_values"] = [v.astype(np.int64) for v in g["nb_values"]]
    synth.save_index(disk, d)
    L, topn = 50, [32] * 5 + [20]
    probe = os.path.join(d, "probe.txt")
    stats = serving.run_serve_host(d, d, 64, clients=48, seconds=1.5, max_batch=64, max_wait_us=300, ef=32, topk=20,
                                   seq_len=L, probe_out=probe)
    assert stats["requests"] > 200 and stats["launches"] > 0 and stats["mean_batch"] > 1.5, stats
    assert stats["latency_ms"]["p50"] > 0 and stats["latency_ms"]["p99"] >= stats["latency_ms"]["p50"]
    seq = np.zeros((L, 64), np.float16)
    seq[:L - 5] = g["item_embs"][0]
    rc, eids, _, _, _ = oracle.search(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), oracle.user_seq_mean(seq), topn)
    got = [int(x) for x in open(probe).read().split()]
    assert got[0] == rc
    if rc == 0:
        assert got[1:] == eids.tolist()
    # level_topn is a per-request feed (build_opt_graph.py:75,151-159): clients with two different level_topn share the
    # launches (--mixed-topn), and a probe with ITS OWN level_topn, launched next to one with the server's, gets the
    # oracle's answer for its values
    own = [16, 24, 32, 8, 12, 10]
    stats = serving.run_serve_host(d, d, 64, clients=48, seconds=1.0, max_batch=64, max_wait_us=300, ef=32, topk=20,
                                   seq_len=L, probe_out=probe, mixed_topn=True, probe_topn=own)
    assert stats["requests"] > 100 and stats["mixed_level_topn"] is True and stats["failed_requests"] <= stats["requests"] // 2, stats
    rc, eids, _, _, _ = oracle.search(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), oracle.user_seq_mean(seq), own)
    got = [int(x) for x in open(probe).read().split()]
    assert got[0] == rc and (rc != 0 or got[1:] == eids.tolist()), (got[:4], rc)
    # admission control (BlazeXlaOp's waiting pool / wait_ms, blaze_xla_kernel.cc:221-258): with a queue bound far below
    # the offered load, requests are refused instead of queueing without limit, and the rest are still served
    stats = serving.run_serve_host(d, d, 64, clients=256, seconds=1.0, max_batch=16, max_wait_us=50, ef=32, topk=20,
                                   seq_len=L, max_queue=8, lanes=1)
    assert stats["refused_queue_full"] > 0 and stats["requests"] > stats["refused_queue_full"], stats


@pytest.mark.parametrize("kind", ["mlp", "attention"])
def test_cpp_serving_host_with_a_model(oracle, tmp_path, kind):
    """The C++ host with `--model-dir`: the 256-128-1 MLP from a weights directory (default = split-f16 precision) and the
    reference's attention + DNN model from a frozen GraphDef file -- both run the pre-projected traversals.  The reply
    to the fixed probe request matches the oracle's traversal under the fp32 model (ids tie-aware: the split forms
    are a 1e-5 contract)."""
    from nann_amd import frozen_graph, ops, serving, synth
    d_emb = 64
    g, oix, _ = synth_index(20000, d_emb, 32)
    d = str(tmp_path)
    disk = dict(g)
    disk["item_embs"] = g["item_embs"].astype(np.float32)
    disk["nb_values"] = [v.astype(np.int64) for v in g["nb_values"]]
    synth.save_index(disk, d)
    L, topn = 50, [32] * 5 + [20]
    if kind == "mlp":
        w = synth.make_mlp_weights(d_emb)
        model = os.path.join(d, "mlp_model")
        ops.save_scorer_dir(model, "mlp", w)
        osc = oracle.Scorer("mlp", d_emb, oracle.EMB_F16, w)
        seq = np.zeros((L, d_emb), np.float16)
        seq[:L - 5] = g["item_embs"][0]
        q = oracle.user_seq_mean(seq)
    else:
        w = synth.make_attn_weights(d_emb, 64)
        model = os.path.join(d, "frozen_graph.pb")
        frozen_graph.write_attention_graph(model, w, seq_len=L)
        osc = oracle.Scorer("attention", d_emb, oracle.EMB_F16, attn_model=oracle.AttnModel(d_emb, 64, L, oracle.EMB_F16, w))
        seq = np.zeros((L, 64), np.float16)  # the attention model's sequence is [L, 64]
        seq[:L - 5] = g["item_embs"][0][:64]
        q = seq.astype(np.float32).reshape(-1)
    probe = os.path.join(d, "probe.txt")
    stats = serving.run_serve_host(d, d, d_emb, clients=32, seconds=1.0, max_batch=64, max_wait_us=300, ef=32, topk=20,
                                   seq_len=L, probe_out=probe, model_dir=model, lanes=1)
    assert stats["requests"] > 50 and stats["failed_requests"] >= 0, stats
    rc, eids, esc, eidx, _ = oracle.search(oix, osc, q, topn)
    got = [int(x) for x in open(probe).read().split()]
    assert got[0] == rc
    if rc == 0:  # ids equal, or different only inside groups of scores that tie within the tolerance
        exp = eids.tolist()
        if got[1:] != exp:
            assert sorted(got[1:]) == sorted(exp) or len(set(got[1:]) & set(exp)) >= len(exp) - 1, (got[1:], exp)


@pytest.mark.parametrize("mode", ["lds_hash", "lds_bitmap"])
@pytest.mark.parametrize("d,dtype,ef,k", [(256, "bf16", 64, 40), (64, "f32", 32, 20), (256, "f16", 48, 30),
                                          (128, "bf16", 256, 200), (512, "f16", 32, 20), (512, "f32", 32, 20), (512, "bf16", 32, 20)])
def test_search_row_dtypes_and_dims(oracle, d, dtype, ef, k, mode):
    """bf16 / f32 rows, 256-d (BASELINE configs[4] shape: 256-d bf16, ef_search=256) and the widest rows the library takes
    (512-d: 64 lanes per row) through the fused traversal."""
    from nann_amd import retrieval
    g, _, _ = synth_index(20000, d, min(ef, 64))
    x = g["item_embs"].astype(np.float32)
    if dtype == "bf16":
        dev = cuda(x).to(torch.bfloat16)
        host = dev.view(torch.int16).cpu().numpy().view(np.uint16)
        code = oracle.EMB_BF16
    elif dtype == "f32":
        host, dev, code = x, cuda(x), oracle.EMB_F32
    else:
        host, dev, code = g["item_embs"], cuda(g["item_embs"]), oracle.EMB_F16
    oix = oracle.Index(host, g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    dix = retrieval.Index(dev, g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 40, seed=5)])
    topn = [min(ef, len(g["enter_points"]))] + [ef] * 4 + [k]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", d, code), q, topn, n_threads=8)
    assert (exp[0] == 0).mean() > 0.5
    _assert_same(_run(dix, q, topn, mode), exp)


@pytest.mark.parametrize("mode,cap", [("lds_hash", 16320), ("lds_hash32", 32704)])
def test_hash_set_overflow_falls_back_to_the_bitmap_kernel(oracle, mode, cap):
    """A query whose visited set could outgrow the hash set (16 320 / 32 704 ids) is handed back with
    NANN_ERR_CAPACITY and rerun on the bitmap kernel inside the same nann_search call: full-degree
    exact-kNN rows walked by wide beams visit more than that, and the results still equal the oracle's."""
    g, oix, dix = synth_index(120000, 64, 256, n_clusters=4, mode="knn")
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 48, seed=31)])
    wide = 512 if cap < 20000 else 1024
    topn = [256, wide, wide, wide, wide, 200]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn, n_threads=8)
    ok = exp[0] == 0
    assert ok.mean() > 0.5
    visited_l0 = topn[1] + exp[4][:, 2, 2:5].sum(1)  # marks + ids kept in the three level-0 rounds
    assert (visited_l0[ok] > cap).any(), "workload must overflow the set for some query"
    _assert_same(_run(dix, q, topn, mode), exp)


def test_mlp_pipeline_hands_overflowing_queries_to_the_fused_bitmap_kernel(oracle):
    """The MLP's pipeline of phases on a forced 16K-slot plan, one launch mixing (per-request level_topn) beams that
    fit the set with beams that outgrow it: a traversal stage hands an overflowing query back (NANN_ERR_CAPACITY),
    the later stages and scoring launches skip it, the fused HBM-bitmap kernel reruns it at the end of the call --
    every reply bitwise equal to the oracle's."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(120000, 64, 256, n_clusters=4, mode="knn")
    w = synth.make_mlp_weights(64)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 16, seed=33)])
    variants = [[256, 512, 512, 512, 512, 200], [128, 128, 128, 128, 128, 200]]
    rows = np.asarray([variants[b % 2] for b in range(len(q))], np.int32)
    osc = oracle.Scorer("mlp", 64, oracle.EMB_F16, w)
    with traversal_mode("lds_hash"):
        r = retrieval.search(dix, ops.Scorer("mlp", 64, torch.float16, w, precision="exact"), cuda(q), rows)
        torch.cuda.synchronize()
    got = (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
           r.index.cpu().numpy(), r.counters.cpu().numpy())
    for v, topn in enumerate(variants):
        sel = np.arange(v, len(q), 2)
        exp = oracle.search_batch(oix, osc, q[sel], topn, n_threads=16)
        ok = exp[0] == 0
        assert ok.mean() > 0.5
        visited_l0 = topn[1] + exp[4][:, 2, 2:5].sum(1)  # marks + ids kept in the three level-0 rounds
        assert ((visited_l0[ok] > 16320).all() if v == 0 else (visited_l0[ok] < 14000).all()), visited_l0
        _assert_same(tuple(a[sel] for a in got), exp)


def test_mlp_traversal_in_hbm_bitmap_mode(oracle):
    """The MLP traversal with the visited bitmap in HBM (what MLP shards beyond ~1.07M items run)."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    w = synth.make_mlp_weights(64)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=23)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("mlp", 64, oracle.EMB_F16, w), q, topn, n_threads=8)
    with traversal_mode("hbm_bitmap"):
        r = retrieval.search(dix, ops.Scorer("mlp", 64, torch.float16, w, precision="exact"), cuda(q), topn)
        torch.cuda.synchronize()
    got = (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
           r.index.cpu().numpy(), r.counters.cpu().numpy())
    _assert_same(got, exp)


@pytest.mark.parametrize("kind", ["l2", "mlp"])
def test_eval_graph_matches_oracle(oracle, kind):
    """f3: Model.retrieval / search_level (model.py:299-362) on the HIP ops vs the oracle's
    restatement: threshold frontier rule, min(k, n) guard, ascending neighbour sets."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    w = synth.make_mlp_weights(64) if kind == "mlp" else None
    osc = oracle.Scorer(kind, 64, oracle.EMB_F16, w)
    sc = ops.Scorer(kind, 64, torch.float16, w, precision="exact")
    qs = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 6, seed=9)])
    n_ok = 0
    for cfg in [((3, 1, 1), (400, 200, 100), 200), ((2, 2, 1), (60, 40, 16), 30)]:
        for q in qs:
            rc, eids, esc, eidx = oracle.search_eval(oix, osc, q, *cfg)
            if rc:
                continue
            ids, s, idx = retrieval.search_eval_per_op(dix, sc, cuda(q), *cfg)
            assert (idx.cpu().numpy() == eidx).all() and (ids.cpu().numpy() == eids).all()
            assert (bits(s.cpu().numpy()) == bits(esc)).all()
            n_ok += 1
    assert n_ok >= 6


@pytest.mark.parametrize("kind", ["l2", "mlp"])
def test_fused_eval_graph_matches_oracle(oracle, kind):
    """f3 in one kernel (nann_search_eval): every user of a batch equals oracle_search_eval bit for bit --
    ids, scores, internal indices, number of rows -- for the reference's defaults (config.py:50-58), narrow
    levels, and a shape whose frontier runs dry (min(k, n) guard, empty score batch is not an error)."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    w = synth.make_mlp_weights(64) if kind == "mlp" else None
    osc = oracle.Scorer(kind, 64, oracle.EMB_F16, w)
    sc = ops.Scorer(kind, 64, torch.float16, w, precision="exact")
    qs = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 40 if kind == "l2" else 12, seed=9)])
    # (the last shape: top_k_per_level above the serving kernels' 1024 -- the reference's values are defaults, config.py:50-58)
    for cfg in [((3, 1, 1), (400, 200, 100), 200), ((2, 2, 1), (60, 40, 16), 30), ((1, 0, 1), (50, 50, 8), 64),
                ((3, 2, 1), (1024, 700, 300), 1024), ((3, 1, 1), (2000, 1000, 500), 1500)]:
        r = retrieval.search_eval(dix, sc, cuda(qs), *cfg)
        torch.cuda.synchronize()
        st, n_out = r.status.cpu().numpy(), r.n_out.cpu().numpy()
        ids, scs, idx = r.item_ids.cpu().numpy(), r.scores.cpu().numpy(), r.index.cpu().numpy()
        for b, q in enumerate(qs):
            rc, eids, esc, eidx = oracle.search_eval(oix, osc, q, *cfg)
            assert st[b] == rc, (cfg, b, st[b], rc)
            if rc:
                assert n_out[b] == 0 and not ids[b].any()
                continue
            n = len(eids)
            assert n_out[b] == n
            assert (idx[b, :n] == eidx).all() and (ids[b, :n] == eids).all()
            assert (bits(scs[b, :n]) == bits(esc)).all()
            assert not ids[b, n:].any() and not scs[b, n:].any()


def test_fused_eval_graph_tiny_graph_and_per_op_agree(oracle):
    """A graph small enough that every level's frontier exhausts (n_out < topk_eval), and the fused kernel
    against the per-op spelling on the same users."""
    from nann_amd import ops, retrieval
    g, oix, dix = synth_index(300, 64, 8)
    sc, osc = ops.Scorer("l2", 64), oracle.Scorer("l2", 64, oracle.EMB_F16)
    qs = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 8, seed=2)])
    cfg = ((4, 3, 1), (400, 400, 400), 350)
    r = retrieval.search_eval(dix, sc, cuda(qs), *cfg)
    torch.cuda.synchronize()
    for b, q in enumerate(qs):
        rc, eids, esc, eidx = oracle.search_eval(oix, osc, q, *cfg)
        assert rc == 0 and int(r.status[b]) == 0
        n = int(r.n_out[b])
        assert n == len(eids) <= 300 and (r.index[b, :n].cpu().numpy() == eidx).all()
        ids, s, idx = retrieval.search_eval_per_op(dix, sc, cuda(q), *cfg)
        assert (idx.cpu().numpy() == eidx).all() and (bits(s.cpu().numpy()) == bits(r.scores[b, :n].cpu().numpy())).all()


def test_fused_eval_graph_counters_equal_the_per_op_spelling(oracle):
    """nann_search_eval_ex's counters (rows walked F, neighbours gathered G, rows scored S: what bench.py's f3 roofline is
    priced on) against the same quantities counted in the op-by-op spelling of Model.retrieval, whose every op is pinned
    to the oracle; results with and without counters are the same bits."""
    from nann_amd import ops, retrieval
    g, oix, dix = synth_index(20000, 64, 32)
    sc = ops.Scorer("l2", 64)
    qs = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=6)])
    for cfg in (((3, 1, 1), (400, 200, 100), 200), ((2, 2, 1), (64, 300, 50), 40)):
        r = retrieval.search_eval(dix, sc, cuda(qs), *cfg, want_counters=True)
        r0 = retrieval.search_eval(dix, sc, cuda(qs), *cfg)
        torch.cuda.synchronize()
        assert (r.index.cpu().numpy() == r0.index.cpu().numpy()).all() and (bits(r.scores.cpu().numpy()) == bits(r0.scores.cpu().numpy())).all()
        ctr = r.counters.cpu().numpy()
        for b in range(0, 24, 3):
            st = {}
            retrieval.search_eval_per_op(dix, sc, cuda(qs[b]), *cfg, stats=st)
            assert int(r.status[b]) == 0 and ctr[b].tolist() == [st["F"], st["G"], st["S"]], (b, ctr[b], st)


def test_fused_eval_graph_with_the_attention_model(oracle, tmp_path):
    """nann_search_eval_model: comm_seq in, the reference's attention + DNN model as the scorer; scores within
    1e-5 of the oracle's, ids tie-aware (the tolerance the attention scorer is held to everywhere)."""
    from nann_amd import ops, retrieval, synth
    d, L = 64, 50
    g, oix, dix = synth_index(20000, d, 32)
    w = synth.make_attn_weights(d, 64)
    ops.save_scorer_dir(str(tmp_path), "attention", w, precision="exact")
    model = ops.Model(str(tmp_path), d, L)
    seqs = queries_for(g, 8, seed=5)
    cfg = ((2, 1, 1), (100, 60, 30), 50)
    r = retrieval.search_eval(dix, model, cuda(seqs), *cfg)
    torch.cuda.synchronize()
    osc = oracle.Scorer("attention", d, oracle.EMB_F16, attn_model=oracle.AttnModel(d, 64, L, oracle.EMB_F16, w))
    kinds = []
    for b, s in enumerate(seqs):
        rc, eids, esc, eidx = oracle.search_eval(oix, osc, s.astype(np.float32).reshape(-1), *cfg)
        assert rc == 0 and int(r.status[b]) == 0 and int(r.n_out[b]) == len(eids)
        n = len(eids)
        kinds.append(tolerant_parity(r.index[b, :n].cpu().numpy(), r.scores[b, :n].cpu().numpy(), eidx, esc))
    assert kinds.count("diverged") <= 1 and kinds.count("exact") >= 5, kinds
    # the evaluation job runs the f32 form of the model whatever precision the weights directory asks serving for
    ops.save_scorer_dir(str(tmp_path / "split"), "attention", w, precision="split")
    r2 = retrieval.search_eval(dix, ops.Model(str(tmp_path / "split"), d, L), cuda(seqs), *cfg)
    torch.cuda.synchronize()
    assert (r2.index.cpu().numpy() == r.index.cpu().numpy()).all() and (r2.n_out.cpu().numpy() == r.n_out.cpu().numpy()).all()
    assert (bits(r2.scores.cpu().numpy()) == bits(r.scores.cpu().numpy())).all()


def test_serving_front_end_on_the_device(oracle):
    """f4: the batching front end (nann_amd/serving.py) over the real backend -- single requests with the
    reference's signature (comm_seq f16[1, L*d], level_topn -> top_k i64[1, k], build_opt_graph.py:151-159)
    are aggregated into nann_search launches; every reply equals the oracle's answer for that request,
    failing requests raise in their caller only; closed-loop load reports throughput and latency."""
    from nann_amd import ops, serving
    g, oix, dix = synth_index(20000, 64, 32)
    seqs = queries_for(g, 300, seed=41)
    topn = [32] * 5 + [20]
    q = np.stack([oracle.user_seq_mean(s) for s in seqs])
    est, eids, _, _, _ = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn, n_threads=8)
    srv = serving.BatchingServer(serving.device_backend(dix, ops.Scorer("l2", 64)), 50, 64, topn, max_batch=128,
                                 max_wait_us=500)
    try:
        futs = [srv.submit(s.reshape(1, -1)) for s in seqs]
        for b, f in enumerate(futs):
            if est[b]:
                with pytest.raises(serving.RequestFailed) as e:
                    f.result(30)
                assert e.value.status == est[b]
            else:
                assert (f.result(30) == eids[b][None]).all()
        assert srv.batches < len(seqs)  # requests were aggregated
        stats = serving.closed_loop(srv, lambda cid: seqs[cid % len(seqs)], n_clients=32, duration_s=1.0)
        assert stats["requests"] > 100 and stats["latency_us"]["p50"] > 0
    finally:
        srv.close()


@pytest.mark.parametrize("kind", ["l2", "mlp", "attention"])
def test_search_model_serving_signature(oracle, tmp_path, kind):
    """nann_search_model: comm_seq + level_topn -> top_k for the model a BlazeXlaOp node names (a weights
    directory).  l2 / mlp: bit-identical to the oracle.  attention (f2): the reference's own scorer FUSED into
    the traversal -- per-user projection once per request, candidates scored on the matrix cores inside
    k_search; logits are equal to the oracle restatement within 1e-5 (device expf, MFMA order), so ids are
    compared tie-aware."""
    from nann_amd import ops, retrieval, synth
    d, L, nq = 64, 50, 40
    g, oix, dix = synth_index(20000, d, 32)
    seqs = queries_for(g, nq, seed=17)                      # f16 [nq, 50, 64]
    topn = [32] * 5 + [20]
    w = {"l2": None, "mlp": synth.make_mlp_weights(d), "attention": synth.make_attn_weights(d, 64)}[kind]
    ops.save_scorer_dir(str(tmp_path), kind, w, precision=None if kind == "l2" else "exact")
    m = ops.Model(str(tmp_path), d, L)
    r = retrieval.search_model(dix, m, cuda(seqs), topn)
    torch.cuda.synchronize()
    st, idx, sc = r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()
    if kind == "attention":
        osc = oracle.Scorer("attention", d, oracle.EMB_F16, attn_model=oracle.AttnModel(d, 64, L, oracle.EMB_F16, w))
        q = seqs.astype(np.float32).reshape(nq, -1)
    else:
        osc = oracle.Scorer(kind, d, oracle.EMB_F16, w)
        q = np.stack([oracle.user_seq_mean(s) for s in seqs])
    est, eids, esc, eidx, ectr = oracle.search_batch(oix, osc, q, topn, n_threads=8)
    ok = est == 0
    assert ok.mean() > 0.5
    if kind != "attention":
        assert (st == est).all() and (idx[ok] == eidx[ok]).all() and (bits(sc[ok]) == bits(esc[ok])).all()
        assert (r.item_ids.cpu().numpy()[ok] == eids[ok]).all() and (r.counters.cpu().numpy()[ok] == ectr[ok]).all()
        return
    kinds = [tolerant_parity(idx[b], sc[b], eidx[b], esc[b]) for b in np.nonzero(ok & (st == 0))[0]]
    assert (st == est).sum() >= nq - 2
    assert kinds.count("exact") >= 0.8 * len(kinds), kinds
    assert kinds.count("diverged") <= 2, kinds


@pytest.mark.parametrize("d,mode,rows", [(64, "auto", "f16"), (128, "auto", "f16"), (128, "hbm_bitmap", "f16"),
                                         (64, "lds_bitmap", "f16"), (64, "auto", "bf16")])
def test_search_model_attention_split_f16(oracle, tmp_path, d, mode, rows):
    """The serving signature with the attention model in its split-f16 form (precision.txt = split; since round 3 the
    traversal runs it with the item-only layers pre-projected per (model, index), nann_attn_proj.h, on every plan):
    the fused traversal against the oracle's traversal with the fp32 model, tie-aware at 1e-5; and against the
    device's own f32 form."""
    from nann_amd import ops, retrieval, synth
    L, nq = 50, 40
    g, oix, dix = synth_index(20000, d, 32)
    code, tdt = oracle.EMB_F16, torch.float16
    if rows == "bf16":  # the same graph over bf16 rows (the pre-projection reads them through its own conversion)
        dev = cuda(g["item_embs"].astype(np.float32)).to(torch.bfloat16)
        host = dev.view(torch.int16).cpu().numpy().view(np.uint16)
        oix = oracle.Index(host, g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        dix = retrieval.Index(dev, g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        code, tdt = oracle.EMB_BF16, torch.bfloat16
    seqs = queries_for(g, nq, seed=17)
    topn = [32] * 5 + [20]
    w = synth.make_attn_weights(d, 64)
    ops.save_scorer_dir(str(tmp_path / "split"), "attention", w, precision="split")
    ops.save_scorer_dir(str(tmp_path / "exact"), "attention", w, precision="exact")
    with traversal_mode(mode):
        r = retrieval.search_model(dix, ops.Model(str(tmp_path / "split"), d, L, emb_dtype=tdt), cuda(seqs), topn)
        r2 = retrieval.search_model(dix, ops.Model(str(tmp_path / "exact"), d, L, emb_dtype=tdt), cuda(seqs), topn)
        torch.cuda.synchronize()
    st, idx, sc = r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()
    osc = oracle.Scorer("attention", d, code, attn_model=oracle.AttnModel(d, 64, L, code, w))
    est, eids, esc, eidx, _ = oracle.search_batch(oix, osc, seqs.astype(np.float32).reshape(nq, -1), topn, n_threads=8)
    ok = (est == 0) & (st == 0)
    assert (st == est).sum() >= nq - 2 and ok.mean() > 0.5
    kinds = [tolerant_parity(idx[b], sc[b], eidx[b], esc[b]) for b in np.nonzero(ok)[0]]
    assert kinds.count("exact") >= 0.8 * len(kinds) and kinds.count("diverged") <= 2, kinds
    for b in np.nonzero(ok)[0]:  # where the lists agree the logits are the fp32 model's within 1e-5
        same = idx[b] == eidx[b]
        assert (np.abs(sc[b][same] - esc[b][same]) <= 1e-5 * np.maximum(1.0, np.abs(esc[b][same]))).all()
    st2, idx2, sc2 = r2.status.cpu().numpy(), r2.index.cpu().numpy(), r2.scores.cpu().numpy()
    both = (st == 0) & (st2 == 0)
    kinds2 = [tolerant_parity(idx[b], sc[b], idx2[b], sc2[b]) for b in np.nonzero(both)[0]]
    assert kinds2.count("diverged") <= 2, kinds2


def test_recall_harness_test_and_test_all(oracle):
    """f3: the reference's evaluation jobs (main.py:144-237) on the HIP ops -- `test` (eval-graph retrieval)
    and `test_all` (brute force), scored with calc_pr (util.py:14-25): one ground-truth item per user.
    Ground truth here = the item the brute force ranks first, so test_all's recall@k is 1 by construction and
    test's recall says how often the traversal keeps the best item."""
    from nann_amd import evaluate, ops
    assert evaluate.calc_pr(5, [1, 5, 9, 9]) == (1 / 3, 1.0, 0.5)   # a set of 3 retrieved ids, one hit
    assert evaluate.calc_pr(4, [1, 5]) == (0.0, 0.0, 0.0)
    g, oix, dix = synth_index(20000, 64, 32)
    seqs = queries_for(g, 24, seed=3)
    sc = ops.Scorer("l2", 64)
    truths = []
    for s in seqs:
        q = ops.user_seq_mean(cuda(s)[None])[0]
        _, bi = ops.top_k(ops.blaze_score(sc, q, item_emb=dix.item_embs), 1)
        truths.append(int(dix.item_ids[bi.long()].cpu()[0]))
    all_ = evaluate.test_all(dix, sc, seqs, truths, topk_eval=(10, 50))
    assert all_["recall"][10].avg == 1.0 and abs(all_["precision"][50].avg - 1 / 50) < 1e-12
    got = evaluate.test(dix, sc, seqs, truths, topk_eval=(10, 50), num_scoring_per_level=(2, 1, 1),
                        top_k_per_level=(100, 60, 30))
    assert got["recall"][50].avg >= 0.7 and got["recall"][10].avg <= got["recall"][50].avg
    per_op = evaluate.test(dix, sc, seqs, truths, topk_eval=(10, 50), num_scoring_per_level=(2, 1, 1),
                           top_k_per_level=(100, 60, 30), fused=False)
    assert all(per_op[m][k].avg == got[m][k].avg for m in ("precision", "recall", "f1") for k in (10, 50))
    # the traversal's first hit is the oracle's eval-graph answer for the same user
    q0 = oracle.user_seq_mean(seqs[0])
    rc, eids, _, _ = oracle.search_eval(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q0, (2, 1, 1), (100, 60, 30), 50)
    assert rc == 0 and (truths[0] in eids[:50].tolist()) == (evaluate.calc_pr(truths[0], eids[:50])[1] == 1.0)


def test_projection_tables_follow_the_index(oracle):
    """A split-f16 MLP scorer keeps the pre-projected tables of the two indices it searched last: searched against
    three indices in turn (A, B, C, A -- C evicts A's table, A's comes back rebuilt) it must answer each one exactly
    as a scorer that has only ever seen that index."""
    from nann_amd import ops, retrieval, synth
    w = synth.make_mlp_weights(64)
    topn = [32] * 5 + [20]
    shared = ops.Scorer("mlp", 64, torch.float16, w, precision="split")
    cases = []
    for seed in (1234, 77, 901):
        g, _, dix = synth_index(20000, 64, 32, seed=seed)
        q = cuda(np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=seed + 1)]))
        fresh = retrieval.search(dix, ops.Scorer("mlp", 64, torch.float16, w, precision="split"), q, topn)
        torch.cuda.synchronize()
        cases.append((dix, q, fresh.status.cpu().numpy(), fresh.index.cpu().numpy(), fresh.scores.cpu().numpy()))
    for k in (0, 1, 2, 0, 2, 1):
        dix, q, st, idx, sc = cases[k]
        r = retrieval.search(dix, shared, q, topn)
        torch.cuda.synchronize()
        assert (r.status.cpu().numpy() == st).all()
        ok = st == 0
        assert (r.index.cpu().numpy()[ok] == idx[ok]).all() and (bits(r.scores.cpu().numpy()[ok]) == bits(sc[ok])).all(), k


@pytest.mark.parametrize("kind,mode", [("l2", "lds_hash"), ("l2", "lds_bitmap"), ("l2", "hbm_bitmap"), ("mlp", "auto")])
def test_level_topn_per_query(oracle, kind, mode):
    """The reference feeds `level_topn` PER REQUEST (build_opt_graph.py:75,151-159): one launch (nann_search_v) mixing
    four different level_topn -- beams of different widths, a different k, one the reference fails in TopKV2 (k > n:
    more entry winners asked for than there are enter points) and one outside the launch's maxima -- must give every
    request the oracle's answer for ITS values, bit for bit, with zeros behind its own k."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    nq = 64
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, nq, seed=29)])
    E = len(g["enter_points"])
    variants = [[32] * 5 + [20], [16, 24, 32, 8, 12, 10], [8] * 5 + [5], [32, 32, 20, 20, 20, 33], [E + 1] + [32] * 4 + [20]]
    rows = np.asarray([variants[b % len(variants)] for b in range(nq)], np.int32)
    w = synth.make_mlp_weights(64) if kind == "mlp" else None
    sc = ops.Scorer(kind, 64, torch.float16, w, precision="exact")
    osc = oracle.Scorer(kind, 64, oracle.EMB_F16, w)
    with traversal_mode(mode):
        r = retrieval.search(dix, sc, cuda(q), rows)
        torch.cuda.synchronize()
    kmax = int(rows[:, 5].max())
    assert r.item_ids.shape == (nq, kmax)
    st, ids, scs, idx, ctr = (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
                              r.index.cpu().numpy(), r.counters.cpu().numpy())
    seen_ok = 0
    for v, topn in enumerate(variants):
        sel = np.arange(v, nq, len(variants))
        est, eids, esc, eidx, ectr = oracle.search_batch(oix, osc, q[sel], topn, n_threads=8)
        k = topn[5]
        assert (st[sel] == est).all(), (v, st[sel], est)
        ok = est == 0
        seen_ok += int(ok.sum())
        assert (idx[sel][ok][:, :k] == eidx[ok]).all() and (ids[sel][ok][:, :k] == eids[ok]).all(), v
        assert (bits(scs[sel][ok][:, :k]) == bits(esc[ok])).all() and (ctr[sel][ok] == ectr[ok]).all(), v
        assert (ids[sel][:, k:] == 0).all() and (scs[sel][:, k:] == 0).all(), v   # zeros behind a query's own k
        assert (ids[sel][~ok] == 0).all()
        if topn[0] > E:
            assert (est == 4).all()  # TopKV2: k > n (topk_op.cc:67-71), per request
    assert seen_ok >= nq // 2
    # a request whose level_topn exceeds the launch's maxima fails alone with BAD_ARGUMENT
    from nann_amd import _lib
    import ctypes as C
    mx = (C.c_int32 * 6)(*[32] * 5 + [20])
    tq = torch.as_tensor(np.asarray([[32] * 5 + [20], [33] + [32] * 4 + [20], [32] * 5 + [20]], np.int32)).cuda()
    out_ids = torch.empty((3, 20), dtype=torch.int64, device="cuda")
    status = torch.empty(3, dtype=torch.int32, device="cuda")
    ws = dix.workspace([32] * 5 + [20], 3)
    rc = _lib.lib().nann_search_v(dix.handle, sc.handle, C.c_void_p(cuda(q[:3]).data_ptr()), C.c_int64(3), mx,
                                  C.c_void_p(tq.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel()),
                                  C.c_void_p(out_ids.data_ptr()), None, None, C.c_void_p(status.data_ptr()), None, None)
    torch.cuda.synchronize()
    assert rc == 0 and status.cpu().tolist()[1] == 7 and (out_ids[1] == 0).all()
    assert status.cpu().tolist()[0] == st[0] and (out_ids[0].cpu().numpy() == ids[0][:20]).all()


def test_level_topn_per_query_serving_signature(oracle, tmp_path):
    """nann_search_model_v: comm_seq + per-request level_topn -> top_k, as the serving graph's signature has it."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    seqs = queries_for(g, 24, seed=31)
    ops.save_scorer_dir(str(tmp_path), "l2")
    model = ops.Model(str(tmp_path), 64, seqs[0].shape[0])
    variants = [[32] * 5 + [20], [16] * 5 + [10]]
    rows = np.asarray([variants[b % 2] for b in range(24)], np.int32)
    r = retrieval.search_model(dix, model, cuda(np.stack(seqs)), rows)
    torch.cuda.synchronize()
    q = np.stack([oracle.user_seq_mean(s) for s in seqs])
    for v, topn in enumerate(variants):
        sel = np.arange(v, 24, 2)
        est, eids, esc, eidx, ectr = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q[sel], topn)
        ok = est == 0
        assert (r.status.cpu().numpy()[sel] == est).all()
        assert (r.item_ids.cpu().numpy()[sel][ok][:, :topn[5]] == eids[ok]).all()


def test_prepared_tables_first_request_latency_and_release(oracle):
    """nann_scorer_prepare builds (and pins) the pre-projected table ahead of traffic: the first search of the pair then
    costs what every later one costs (no hipMalloc, no build, no stream wait inside the request), the bytes are
    reported, pinned tables survive searches of other indices, release returns the memory, and a search after the
    release still answers identically (it rebuilds)."""
    import time
    from nann_amd import ops, retrieval, synth
    w = synth.make_mlp_weights(64)
    topn = [32] * 5 + [20]
    idxs = [synth_index(20000, 64, 32, seed=s) for s in (1234, 77, 901)]
    qs = [cuda(np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=5)])) for g, _, _ in idxs]

    def timed(sc, dix, q):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = retrieval.search(dix, sc, q, topn)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    # cold scorer, no prepare: the first request pays the build
    cold = ops.Scorer("mlp", 64, torch.float16, w, precision="split")
    t_cold, r_ref = timed(cold, idxs[0][2], qs[0])
    sc = ops.Scorer("mlp", 64, torch.float16, w, precision="split")
    tb, resident = retrieval.table_bytes(idxs[0][2], sc)
    assert tb == 20000 * 256 * 4 and resident == 0
    tb, resident = retrieval.prepare(idxs[0][2], sc)
    assert resident == tb
    t_first, r = timed(sc, idxs[0][2], qs[0])
    steady = sorted(timed(sc, idxs[0][2], qs[0])[0] for _ in range(7))[3]
    print(f"first request: cold {t_cold * 1e3:.2f} ms, after prepare {t_first * 1e3:.2f} ms, steady {steady * 1e3:.2f} ms")
    assert t_first < 3.0 * steady + 1e-3, (t_first, steady)   # no build in the request (cold: tens of ms more)
    assert (r.index.cpu().numpy() == r_ref.index.cpu().numpy()).all()
    assert (bits(r.scores.cpu().numpy()) == bits(r_ref.scores.cpu().numpy())).all()
    # the pinned table survives the scorer searching two other indices (which fill the two unpinned places) and a third
    for k in (1, 2):
        retrieval.search(idxs[k][2], sc, qs[k], topn)
    torch.cuda.synchronize()
    _, resident = retrieval.table_bytes(idxs[0][2], sc)
    assert resident >= 3 * tb
    t_again, _ = timed(sc, idxs[0][2], qs[0])
    assert t_again < 3.0 * steady + 1e-3
    retrieval.release(idxs[0][2], sc)
    torch.cuda.synchronize()
    _, resident = retrieval.table_bytes(idxs[0][2], sc)
    assert resident <= 2 * tb   # the released table is gone (retired, then freed once its launches were done)
    _, r2 = timed(sc, idxs[0][2], qs[0])
    assert (r2.index.cpu().numpy() == r_ref.index.cpu().numpy()).all()


def test_tables_under_concurrent_threads(oracle):
    """3 indices x 4 host threads on ONE scorer, each thread on its own stream, searching the indices in turn so that
    the two unpinned places keep turning over: every reply equals the single-threaded one (round 3 could free a table
    another thread was about to launch with -- ADVICE r3)."""
    import threading
    from nann_amd import ops, retrieval, synth
    w = synth.make_mlp_weights(64)
    topn = [32] * 5 + [20]
    idxs = [synth_index(20000, 64, 32, seed=s) for s in (1234, 77, 901)]
    qs = [cuda(np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=5)])) for g, _, _ in idxs]
    ref = []
    for (g, _, dix), q in zip(idxs, qs):
        r = retrieval.search(dix, ops.Scorer("mlp", 64, torch.float16, w, precision="split"), q, topn)
        torch.cuda.synchronize()
        ref.append((r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()))
    shared = ops.Scorer("mlp", 64, torch.float16, w, precision="split")
    errors = []

    def worker(tid):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for it in range(12):
                    k = (tid + it) % 3
                    r = retrieval.search(idxs[k][2], shared, qs[k], topn)
                    stream.synchronize()
                    st, ix, sc_ = r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()
                    ok = ref[k][0] == 0
                    if not ((st == ref[k][0]).all() and (ix[ok] == ref[k][1][ok]).all()
                            and (bits(sc_[ok]) == bits(ref[k][2][ok])).all()):
                        errors.append((tid, it, k))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    tb, resident = retrieval.table_bytes(idxs[0][2], shared)
    torch.cuda.synchronize()
    assert retrieval.table_bytes(idxs[0][2], shared)[1] <= 3 * tb   # two kept + at most one retired not yet reaped


def test_search_without_preprojection_matches(oracle):
    """nann_set_preprojection(0) (what a host without HBM to spare asks for, and what a failed table allocation falls
    back to): the kernels that read the embedding rows -- exact form bit-identical to the oracle, i.e. to the
    pre-projected form; split form within tolerance."""
    from nann_amd import _lib, ops, retrieval, synth
    g, oix, dix = synth_index(20000, 64, 32)
    w = synth.make_mlp_weights(64)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 24, seed=23)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("mlp", 64, oracle.EMB_F16, w), q, topn, n_threads=8)
    try:
        _lib.lib().nann_set_preprojection(0)
        sc = ops.Scorer("mlp", 64, torch.float16, w, precision="exact")
        r = retrieval.search(dix, sc, cuda(q), topn)
        torch.cuda.synchronize()
        assert retrieval.table_bytes(dix, sc)[1] == 0
        _assert_same((r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
                      r.index.cpu().numpy(), r.counters.cpu().numpy()), exp)
        r = retrieval.search(dix, ops.Scorer("mlp", 64, torch.float16, w, precision="split"), cuda(q), topn)
        torch.cuda.synchronize()
        ok = exp[0] == 0
        kinds = [tolerant_parity(r.index.cpu().numpy()[b], r.scores.cpu().numpy()[b], exp[3][b], exp[2][b]) for b in np.nonzero(ok)[0]]
        assert kinds.count("diverged") <= 1, kinds
    finally:
        _lib.lib().nann_set_preprojection(1)
    # round 5: the same per CALL (nann_search_options.preprojection), next to a call on the same scorer that does use a table
    sc = ops.Scorer("mlp", 64, torch.float16, w, precision="exact")
    r0 = retrieval.search(dix, sc, cuda(q), topn, options=retrieval.search_options(preprojection=False))
    torch.cuda.synchronize()
    assert not r0.plan["table"] and retrieval.table_bytes(dix, sc)[1] == 0
    r1 = retrieval.search(dix, sc, cuda(q), topn)
    torch.cuda.synchronize()
    assert r1.plan["table"] and retrieval.table_bytes(dix, sc)[1] > 0
    # and in the REVERSE order -- with the pair's table cached by r1, a preprojection=0 call still runs without it
    r2 = retrieval.search(dix, sc, cuda(q), topn, options=retrieval.search_options(preprojection=False))
    torch.cuda.synchronize()
    assert not r2.plan["table"] and retrieval.table_bytes(dix, sc)[1] > 0  # (the cached table stays for the calls that want it)
    for r in (r0, r1, r2):
        _assert_same((r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(),
                      r.index.cpu().numpy(), r.counters.cpu().numpy()), exp)


def test_search_options_travel_with_the_call(oracle):
    """nann_search_options (round 5): the planner's knobs are fields of the call, not process state.  Every plan forced
    per call answers like the oracle and reports itself in nann_search_plan; the process default (set_traversal_mode)
    is untouched by a call's options; two threads sharing the index and the scorer run DIFFERENT plans concurrently."""
    import threading
    from nann_amd import ops, retrieval
    g, oix, dix = synth_index(20000, 64, 32)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 200, seed=31)])
    topn = [32] * 5 + [20]
    exp = oracle.search_batch(oix, oracle.Scorer("l2", 64, oracle.EMB_F16), q, topn, n_threads=8)
    sc = ops.Scorer("l2", 64, torch.float16)

    def run(mode, qq=q):
        r = retrieval.search(dix, sc, cuda(qq), topn, options=retrieval.search_options(traversal=mode) if mode else None)
        torch.cuda.synchronize()
        return r, (r.status.cpu().numpy(), r.item_ids.cpu().numpy(), r.scores.cpu().numpy(), r.index.cpu().numpy(),
                   r.counters.cpu().numpy())

    for mode in MODES:
        r, got = run(mode)
        assert r.plan["visited_set"] == mode, r.plan
        _assert_same(got, exp)
    r, got = run(None)  # defaults: the planner's own choice, and no trace of the forced calls above
    assert r.plan["visited_set"] in ("lds_hash", "lds_hash32") and r.reruns() == 0
    _assert_same(got, exp)
    r1, _ = run(None, q[:1])
    assert r1.plan["visited_set"] == "lds_hash32" and r1.plan["workgroups"] == 1  # at most one query per CU: one 1024-thread workgroup
    out, errs = {}, []

    def worker(mode):
        try:
            for _ in range(6):
                torch.cuda.set_device(0)
                with torch.cuda.stream(torch.cuda.Stream()):
                    r, got = run(mode)
                    assert r.plan["visited_set"] == mode
                    _assert_same(got, exp)
            out[mode] = True
        except Exception as e:  # noqa: BLE001
            errs.append((mode, repr(e)))

    ths = [threading.Thread(target=worker, args=(m,)) for m in ("lds_hash", "hbm_bitmap", "lds_hash32")]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs and len(out) == 3, errs
    # a reserve leaves workgroup slots free without changing an answer (600 queries: every slot of the plan is taken)
    q3 = np.concatenate([q, q, q])
    r, got = run(None, q3)
    full = r.plan["workgroups"]
    r2 = retrieval.search(dix, sc, cuda(q3), topn, options=retrieval.search_options(slot_reserve=16))
    torch.cuda.synchronize()
    assert full > 16 and r2.plan["workgroups"] == full - 16, (r.plan, r2.plan)
    assert (r2.item_ids.cpu().numpy() == got[1]).all()


@pytest.mark.parametrize("precision,form", [("exact", "auto"), ("split", "fused"), ("split", "phased")])
def test_mlp_traversal_scores_against_float64_numpy(oracle, precision, form):
    """the oracle's MLP order changed together with the kernels (a1 = u + P), so "bit-identical to the oracle"
    cannot catch a mistake the two share.  This check does not touch the oracle's scorer: every score the traversal
    returns -- fused kernel and pipeline of phases, both precisions, on the pre-projected table -- against a float64 numpy
    evaluation of the reference's MLP, concat([q ; e]) W1 + b1 -> PReLU -> W2 + b2 -> PReLU -> w3 (model_util.py:9-11,
    model.py:218-219), with per-unit PReLU slopes of both signs."""
    from nann_amd import ops, retrieval, synth
    g, oix, dix = synth_index(20000, 128, 32)
    rng = np.random.default_rng(17)
    w = synth.make_mlp_weights(128)
    w["alpha1"] = rng.uniform(-0.3, 1.2, 256).astype(np.float32)
    w["alpha2"] = rng.uniform(-0.3, 1.2, 128).astype(np.float32)
    w["b1"] = (rng.standard_normal(256) * 0.1).astype(np.float32)
    w["b2"] = (rng.standard_normal(128) * 0.1).astype(np.float32)
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, 300, seed=41)])
    topn = [32] * 5 + [20]
    sc = ops.Scorer("mlp", 128, torch.float16, w, precision=precision)
    r = retrieval.search(dix, sc, cuda(q), topn, options=retrieval.search_options(mlp_form=form))
    torch.cuda.synchronize()
    assert r.plan["table"] and r.plan["phased"] == (form == "phased" or precision == "exact"), r.plan
    st, idx, got = r.status.cpu().numpy(), r.index.cpu().numpy(), r.scores.cpu().numpy()
    ok = np.nonzero(st == 0)[0]
    assert len(ok) > 150
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    prelu = lambda z, a: np.maximum(z, 0) + a * np.minimum(z, 0)
    worst = 0.0
    for b in ok[:120]:
        e = g["item_embs"][idx[b]].astype(np.float64)
        x = np.concatenate([np.tile(q[b].astype(np.float64), (len(e), 1)), e], 1)
        ref = prelu(prelu(x @ w64["w1"] + w64["b1"], w64["alpha1"]) @ w64["w2"] + w64["b2"], w64["alpha2"]) @ w64["w3"]
        worst = max(worst, float(np.max(np.abs(got[b] - ref) / np.maximum(1.0, np.abs(ref)))))
        assert (np.diff(got[b]) <= 0).all()  # sorted descending
    # exact f32: fmaf chains of 256 + 256 + 128 terms; split-f16: north_star's 1e-5
    assert worst <= (3e-6 if precision == "exact" else 1e-5), worst


@pytest.mark.parametrize("mode", ["auto", "lds_bitmap"])
def test_search_random_level_topn(oracle, mode):
    """Thirty random level_topn vectors (beams of 1..600 per stage, top-k 1..400) on graphs of 20 000 and 60 000 items, batches of
    3 and 300 queries (one and two workgroups per CU): every selection size the bin-grouped ranking of top-k sees, requests the
    reference fails included -- status, ids, scores and counters equal to the oracle's."""
    rng = np.random.default_rng(61)
    sc = oracle.Scorer("l2", 64, oracle.EMB_F16)
    n_valid = 0
    for trial in range(30):
        g, oix, dix = synth_index(20000 if trial % 2 else 60000, 64, 32)
        E = len(g["enter_points"])
        nq = 3 if trial % 3 == 0 else 300
        q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, nq, seed=200 + trial)])
        t = [int(rng.integers(1, min(E, 600) + 1))] + [int(x) for x in rng.integers(1, 601, size=4)]
        t.append(int(rng.integers(1, min(400, t[1] + t[2] + t[3] + t[4]) + 1)))
        exp = oracle.search_batch(oix, sc, q, t, n_threads=8)
        _assert_same(_run(dix, q, t, mode), exp)
        n_valid += int((exp[0] == 0).sum())
    assert n_valid > 1000
