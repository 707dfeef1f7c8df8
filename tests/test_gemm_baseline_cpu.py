"""The GEMM-backed CPU port of the schedule (oracle/gemm_baseline.py: bench.py's second CPU baseline for the MLP
workloads) against the parity oracle: same status codes, id lists identical up to near-ties (GEMM summation order
differs from the canonical chains), scores within 1e-5."""
import numpy as np

from nann_amd import synth
from oracle import gemm_baseline as G
from oracle import oracle as O


def test_gemm_port_agrees_with_the_oracle():
    g = synth.make_index(6000, 64, ef=32, seed=5, n_clusters=8, mode="hnsw_cpu")
    w = synth.make_mlp_weights(64)
    seqs = synth.make_queries(g["item_embs"], g["assign"], 24, seed=3)
    q = np.stack([O.user_seq_mean(s) for s in seqs])
    topn = [32] * 5 + [20]
    oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    est, eids, esc, eidx, _ = O.search_batch(oix, O.Scorer("mlp", 64, O.EMB_F16, w), q, topn, n_threads=4)
    st, ids, sc, _ = G.search_batch(g, w, q, topn, n_threads=4)
    assert (st == est).all(), (st, est)
    ok = np.nonzero(est == 0)[0]
    assert len(ok) >= 12
    inv = {int(v): k for k, v in enumerate(g["item_ids"])}
    kinds = []
    for b in ok:
        gidx = np.asarray([inv[int(v)] for v in ids[b]], np.int32)
        kinds.append(O.tolerant_parity(gidx, sc[b], eidx[b], esc[b]))
    assert kinds.count("diverged") <= 1 and kinds.count("exact") >= 0.8 * len(kinds), kinds
    # failure codes of the reference are reproduced too: k > n at the entry layer
    st2, _, _, _ = G.search_batch(g, w, q[:2], [len(g["enter_points"]) + 1] + [32] * 4 + [20])
    assert (st2 == 4).all()
