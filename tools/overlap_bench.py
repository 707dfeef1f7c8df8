#!/usr/bin/env python3
"""Does the exchange step of a sharded search overlap the NEXT batch's traversal on one GPU?  

configs[3]'s per-rank pipeline on ONE device with an 8-shard LOOPBACK communicator (nann_comm_create(world=8, id=NULL):
pack -> 8 device-to-device copies of the 9.8 MB record in place of the ncclAllGather -> k_merge_records over 8 shards):
batch 4096 on the 1M x 128-d index, a different query batch per step.  Three orders, same work:

  serial      search and exchange on one stream: a step costs search + exchange
  overlapped  the exchange of batch i on a stream of its own behind the search that produced it, the caller's stream
              goes straight on to batch i + 1 (ShardedSearch.merge(overlap=True), what bench.py --gpus N runs):
              a step costs max(search, exchange) IF the exchange's kernels get compute units while the persistent
              k_search grid is resident
  search only the floor

and the same with the traversal grid leaving n workgroup slots free for the exchange's kernels (per call:
nann_search_options.slot_reserve).  Round 5: `--repeat R` issues the loopback's device copies R times
(nann_comm_set_timing's loopback_repeat), so that the stand-in exchange lasts as long as 8 GPUs' all-gather over xGMI
-- but ALSO 50 x its memory traffic, which is what then slows the search: profiles/rd5f_reserve_sweep_copies_x50.jsonl --,
`--wait-us T` instead puts 16 workgroups in front of the copies that WAIT for T microseconds (what RCCL's ring kernels do
while the peers' bytes cross xGMI: hold a slot per channel), and `--reserves 0,8,16,32` sweeps the reserve in one run.  Prints one JSON line per reserve.  Under
`rocprofv3 --kernel-trace` the start / end stamps of k_merge_records against k_search show the same thing kernel by
kernel (tools/gpu_r4.sh overlap)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import ops, retrieval, shard  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("cache", nargs="?", default=None)
    ap.add_argument("steps", nargs="?", type=int, default=30)
    ap.add_argument("--repeat", type=int, default=1, help="loopback copies issued this many times (15 ~ xGMI's 1.5 ms)")
    ap.add_argument("--wait-us", type=int, default=0, help="16 waiting workgroups in front of the copies: the exchange lasts this long")
    ap.add_argument("--reserves", default=os.environ.get("NANN_SEARCH_SLOT_RESERVE", "0"))
    a = ap.parse_args()
    cache, steps = a.cache, a.steps
    batch, shards, ef, k = 4096, 8, 128, 200
    dev = torch.device("cuda")
    g = bench.make_index(1_000_000, 128, ef, "hnsw", 1.0, "f16", 0, dev, bench.usable_cores(), cache_dir=cache)
    index = retrieval.Index.from_dict(g, device=dev)
    scorer = ops.Scorer("l2", 128, torch.float16)
    topn = [ef] * 5 + [k]
    seqs = bench.make_query_batches(128, batch, 8, 1.0, dev, n_clusters=bench.n_clusters_for(1_000_000, ef))
    qs = [ops.user_seq_mean(s) for s in seqs]
    torch.cuda.synchronize()

    def make_sharded():
        ss = shard.ShardedSearch(topn, shards, 0, transport="rccl", comm=shard.Comm.loopback(shards))
        ss.comm.set_timing(True, loopback_repeat=a.repeat, loopback_wait_us=a.wait_us)
        return ss

    def run(mode, reserve):
        ss = make_sharded()
        outs = []
        opt = retrieval.search_options(slot_reserve=reserve)

        def step(j):
            r = retrieval.search(index, scorer, qs[j % len(qs)], topn, want_counters=False, options=opt)
            if mode == "search_only":
                return r.item_ids
            return ss.merge(r, overlap=(mode == "overlapped"))[0]

        for j in range(6):
            outs.append(step(j))
        ss.wait()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(steps):
            outs.append(step(j))
        ss.wait()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        return dt, outs[-1].cpu().numpy()

    for reserve in [int(x) for x in a.reserves.split(",")]:
        res = {}
        ref = None
        for mode in ("search_only", "serial", "overlapped", "serial", "overlapped"):
            dt, last = run(mode, reserve)
            res.setdefault(mode, []).append(round(dt, 4))
            if mode != "search_only":
                if ref is None:
                    ref = last
                assert (last == ref).all(), "overlapped and serial orders disagree"
        # the exchange alone (pack + copies + merge), serial on the stream, and its parts by the communicator's own events
        r = retrieval.search(index, scorer, qs[0], topn, want_counters=False)
        ss = make_sharded()
        for _ in range(3):
            ss.merge(r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ss.merge(r)
        torch.cuda.synchronize()
        res["exchange_only"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        res["exchange_parts_ms"] = ss.comm.last_breakdown()
        s_, e = min(res["search_only"]), res["exchange_only"]
        res["summary"] = {"search_ms": s_, "exchange_ms": e, "serial_ms": min(res["serial"]), "overlapped_ms": min(res["overlapped"]),
                          "ideal_overlap_ms": round(max(s_, e), 4), "sum_ms": round(s_ + e, 4),
                          "hidden_fraction_of_exchange": round((min(res["serial"]) - min(res["overlapped"])) / e, 3) if e else None,
                          "slot_reserve": reserve, "loopback_repeat": a.repeat, "loopback_wait_us": a.wait_us}
        print(json.dumps({"workload": "1M x 128-d f16, ef=128, L2, batch 4096, 8-shard loopback exchange (9.8 MB record x 8 x %d)" % a.repeat, **res}), flush=True)


if __name__ == "__main__":
    main()
