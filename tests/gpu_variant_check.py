"""Run by test_search_gpu.py::test_global_bitmap_variants in a fresh process (the kernel
variant knob NANN_L2_VARIANT is read once per process): fused traversal vs the oracle on a
small index, L2 and MLP scorers.  Exit code 0 = identical."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from gpu_util import bits, cuda, queries_for, synth_index  # noqa: E402
from nann_amd import ops, retrieval, synth  # noqa: E402
from oracle import oracle  # noqa: E402


def check(kind, n, d, ef, k, nq):
    g, oix, dix = synth_index(n, d, ef)
    w = synth.make_mlp_weights(d) if kind == "mlp" else None
    q = np.stack([oracle.user_seq_mean(s) for s in queries_for(g, nq, seed=23)])
    topn = [ef] * 5 + [k]
    est, eids, esc, eidx, ectr = oracle.search_batch(oix, oracle.Scorer(kind, d, oracle.EMB_F16, w), q, topn,
                                                     n_threads=8)
    r = retrieval.search(dix, ops.Scorer(kind, d, torch.float16, w), cuda(q), topn)
    torch.cuda.synchronize()
    st = r.status.cpu().numpy()
    ok = est == 0
    same = ((st == est).all() and (r.index.cpu().numpy()[ok] == eidx[ok]).all()
            and (r.item_ids.cpu().numpy()[ok] == eids[ok]).all()
            and (bits(r.scores.cpu().numpy()[ok]) == bits(esc[ok])).all()
            and (r.counters.cpu().numpy()[ok] == ectr[ok]).all())
    print(kind, n, d, ef, "variant", os.environ.get("NANN_L2_VARIANT"), "ok" if same else "MISMATCH", flush=True)
    return bool(same) and bool(ok.mean() > 0.5)


if __name__ == "__main__":
    good = check("l2", 20000, 64, 32, 20, 96) and check("l2", 60000, 128, 64, 50, 48) and \
        check("mlp", 20000, 64, 32, 20, 24)
    sys.exit(0 if good else 1)
