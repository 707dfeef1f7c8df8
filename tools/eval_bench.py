#!/usr/bin/env python3
"""The evaluation graph's traversal (k_search_eval) on configs[1]'s index: users/s of one launch over
1024 users with the reference's defaults (config.py:50-58: num_scoring (3, 1, 1), top_k_per_level (400, 200, 100),
topk_eval 200) and with (2000, 1000, 500) / 1000, each checked against oracle_search_eval on a sample of users
(bit for bit: ids, scores, internal indices, row counts).  NANN_EVAL_SEEN=hbm in the environment keeps `seen` in the
slot instead of LDS (the form of shards whose bitmap does not fit).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import ops, retrieval  # noqa: E402


def main():
    cache = sys.argv[1] if len(sys.argv) > 1 else None
    users = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    n_check = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    # [items dim dtype ef]: another shard shape, e.g. `4000000 256 bf16 256` = configs[4]'s shard (the LDS form in windows)
    items = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    dim = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    dtype = sys.argv[6] if len(sys.argv) > 6 else "f16"
    ef = int(sys.argv[7]) if len(sys.argv) > 7 else 128
    dev = torch.device("cuda")
    g = bench.make_index(items, dim, ef, "hnsw", 1.0, dtype, 0, dev, bench.usable_cores(), cache_dir=cache)
    index = retrieval.Index.from_dict(g, device=dev)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    scorer = ops.Scorer("l2", dim, tdt)
    seqs = bench.make_query_batches(dim, users, 1, 1.0, dev, n_clusters=bench.n_clusters_for(items, ef))[0]
    q = ops.user_seq_mean(seqs)
    torch.cuda.synchronize()
    from oracle import oracle as O
    oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
    osc = O.Scorer("l2", dim, {"f16": O.EMB_F16, "bf16": O.EMB_BF16, "f32": O.EMB_F32}[dtype])
    qh = q.cpu().numpy()
    out = {"workload": "%d x %d-d %s index, L2 scorer, %d users per launch" % (items, dim, dtype, users),
           "seen": os.environ.get("NANN_EVAL_SEEN", "lds")}
    for name, cfg in (("defaults_400_200_100", ((3, 1, 1), (400, 200, 100), 200)),
                      ("wide_2000_1000_500", ((3, 1, 1), (2000, 1000, 500), 1000))):
        for _ in range(3):
            r = retrieval.search_eval(index, scorer, q, *cfg)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        ev0.record()
        for _ in range(reps):
            r = retrieval.search_eval(index, scorer, q, *cfg)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / reps
        st, n_out = r.status.cpu().numpy(), r.n_out.cpu().numpy()
        ids, scs, idx = r.item_ids.cpu().numpy(), r.scores.cpu().numpy(), r.index.cpu().numpy()
        same = 0
        t0 = time.time()
        for b in range(min(n_check, users)):
            rc, eids, esc, eidx = O.search_eval(oix, osc, qh[b], *cfg)
            n = len(eids) if rc == 0 else 0
            ok = (st[b] == rc and n_out[b] == n and (idx[b, :n] == eidx[:n]).all() and (ids[b, :n] == eids[:n]).all()
                  and (scs[b, :n].view(np.uint32) == np.asarray(esc[:n], np.float32).view(np.uint32)).all())
            same += bool(ok)
        out[name] = {"ms_per_launch": round(ms, 3), "users_per_s": round(users / ms * 1e3, 1),
                     "status_ok": int((st == 0).sum()), "mean_rows_out": float(n_out.mean()),
                     "oracle_users_checked": min(n_check, users), "bit_identical": same,
                     "oracle_s_per_user": round((time.time() - t0) / max(1, min(n_check, users)), 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
