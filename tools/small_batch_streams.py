#!/usr/bin/env python3
"""A batch below the CU count leaves CUs idle (B = 64: 64 of 256).  VERDICT r5 next 9 proposed to split a batch's rounds over two
stream-ordered launches so that rounds 3-5 of batch i overlap rounds 1-2 of batch i + 1.  A query's rounds are sequentially
dependent (round r + 1 walks the frontier round r selected), so a split cannot shorten a batch; what it can do -- fill the idle
CUs with ANOTHER batch -- whole launches on separate streams do as well, without parking a traversal's state between launches.
This tool measures that: S streams, each issuing launches of B queries back to back on configs[1]'s index (L2 scorer), wall clock
over all of them and the latency of a launch by HIP events on its stream.

usage: tools/small_batch_streams.py [index cache dir] [--batch 64] [--streams 1,2,4] [--launches 200]"""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("cache", nargs="?", default=None)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--streams", default="1,2,4")
    ap.add_argument("--launches", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda")
    ef, k, d = 128, 200, 128
    g = bench.make_index(1_000_000, d, ef, "hnsw", 1.0, "f16", 0, dev, bench.usable_cores(), cache_dir=a.cache)
    index = retrieval.Index.from_dict(g, device=dev)
    scorer = ops.Scorer("l2", d, torch.float16)
    topn = [ef] * 5 + [k]
    seqs = bench.make_query_batches(d, 1024, 2, 1.0, dev, n_clusters=bench.n_clusters_for(1_000_000, ef))
    qs = [ops.user_seq_mean(seqs[0][i * a.batch:(i + 1) * a.batch]) for i in range(4)]
    ref = retrieval.search(index, scorer, qs[0], topn)
    torch.cuda.synchronize()
    for ns in [int(x) for x in a.streams.split(",")]:
        streams = [torch.cuda.Stream() for _ in range(ns)]
        for s in streams:  # warm-up
            with torch.cuda.stream(s):
                for _ in range(10):
                    retrieval.search(index, scorer, qs[0], topn, want_counters=False)
        torch.cuda.synchronize()
        evs = [[] for _ in range(ns)]
        keep = []
        t0 = time.perf_counter()
        for i in range(a.launches):
            for j, s in enumerate(streams):
                with torch.cuda.stream(s):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    r = retrieval.search(index, scorer, qs[j % 4], topn, want_counters=False)
                    e1.record(s)
                    evs[j].append((e0, e1))
                    if i == a.launches - 1 and j == 0:
                        keep.append(r)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        lat = np.array([e0.elapsed_time(e1) for per in evs for e0, e1 in per])
        same = bool((keep[0].item_ids == ref.item_ids).all()) and bool((keep[0].scores == ref.scores).all())
        print(json.dumps({"batch": a.batch, "streams": ns, "launches_per_stream": a.launches,
                          "queries_per_s": round(ns * a.launches * a.batch / wall, 1),
                          "launch_ms_p50": round(float(np.percentile(lat, 50)), 4), "launch_ms_p99": round(float(np.percentile(lat, 99)), 4),
                          "wall_ms_per_round_of_launches": round(wall / a.launches * 1e3, 4),
                          "stream_0_reply_equals_the_single_stream_reply": same}))


if __name__ == "__main__":
    main()
