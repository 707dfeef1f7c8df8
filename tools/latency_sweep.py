#!/usr/bin/env python3
"""Latency of one launch at small request batches, per visited-set plan (a batch below the CU count leaves
every workgroup a CU of its own, so the plan that wins on a full chip -- 512 threads, two workgroups per CU -- need not be the
one with the shortest chain for ONE query).  configs[1]'s index, L2 scorer; prints one JSON line per (batch, plan):
median / max of 20 launches by HIP events, after 5 untimed ones.

usage: tools/latency_sweep.py [index cache dir] [--batches 1,16,64,128,256,512] [--plans auto,lds_hash,lds_hash32,lds_bitmap]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import ops, retrieval  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cache", nargs="?", default=None)
    ap.add_argument("--batches", default="1,16,64,128,256,512")
    ap.add_argument("--plans", default="auto,lds_hash,lds_hash32,lds_bitmap")
    a = ap.parse_args()
    dev = torch.device("cuda")
    ef, k, d = 128, 200, 128
    g = bench.make_index(1_000_000, d, ef, "hnsw", 1.0, "f16", 0, dev, bench.usable_cores(), cache_dir=a.cache)
    index = retrieval.Index.from_dict(g, device=dev)
    scorer = ops.Scorer("l2", d, torch.float16)
    topn = [ef] * 5 + [k]
    seqs = bench.make_query_batches(d, 1024, 2, 1.0, dev, n_clusters=bench.n_clusters_for(1_000_000, ef))
    ref = {}
    for bsz in [int(x) for x in a.batches.split(",")]:
        q = ops.user_seq_mean(seqs[0][:bsz])
        for plan in a.plans.split(","):
            opt = retrieval.search_options(traversal=plan)
            ts = []
            for it in range(25):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = retrieval.search(index, scorer, q, topn, want_counters=False, options=opt)
                e1.record()
                torch.cuda.synchronize()
                if it >= 5:
                    ts.append(e0.elapsed_time(e1))
            ids = r.item_ids.cpu().numpy()
            if bsz not in ref:
                ref[bsz] = ids
            assert (ids == ref[bsz]).all(), "plans disagree"
            ts = np.asarray(ts)
            print(json.dumps({"batch": bsz, "requested": plan, "plan": r.plan, "ms_p50": round(float(np.median(ts)), 4), "ms_max": round(float(ts.max()), 4),
                              "qps": round(bsz / (float(np.median(ts)) * 1e-3), 1)}), flush=True)


if __name__ == "__main__":
    main()
