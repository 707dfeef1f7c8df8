#!/usr/bin/env python
"""bench.py -- retrieval QPS of the fused HNSW-with-model-scoring traversal on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic UserBehavior-shaped
queries: comm_seq -> query vectors -> layered traversal (neighbour gather, visited-bitmap
dedup, embedding gather + L2 scoring, top-k) -> top-200 item ids.  Inputs are resident
in HBM when the timed region starts.

Workload (BASELINE.json configs[1]): 1M items x 128-d f16, M=32, ef_search=128
(level_topn = [128]*5 + [200]), L2 scoring, on each GPU.  With N > 1 ranks the corpus
is N shards of 1M items (configs[3] shape: item-id sharding); every rank searches every
query on its shard, per-shard top-200 lists are all-gathered over RCCL and merged.

Prints ONE JSON line (rank 0).  `roofline` prices the traversal kernel's ALGORITHMIC
bytes (BASELINE.md section 4 formula over the kernel's own per-round counters, which the
parity tests pin to the oracle's) against 8 TB/s HBM; `cpu_baseline` is the oracle
(oracle/nann_oracle.c, a port of the reference's CPU op loops) timed on this box's cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--items", type=int, default=1_000_000, help="items per GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--topk", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096, help="queries per step")
    ap.add_argument("--graph", default="hnsw", choices=["hnsw", "knn"])
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--merge", default="device", choices=["device", "host"])
    ap.add_argument("--scorer", default="l2", choices=["l2", "mlp"],
                    help="l2 = BASELINE configs[1] (the headline metric); mlp = configs[2]: 256-128-1 MLP on MFMA")
    ap.add_argument("--index-cache", default=None, help="directory to cache the synthetic index in")
    ap.add_argument("--batch-sweep", default="", help="comma-separated smaller batch sizes to time as well, e.g. 1,64,1024")
    ap.add_argument("--phase-ticks", action="store_true",
                    help="one extra instrumented launch: per-phase time attribution")
    return ap.parse_args()


def algorithmic_bytes(counters, d, emb_bytes, n_enter, k_out=200):
    """bytes one query must move, from the per-round counters the kernel
    emits (frontier F, gathered G, scored S): S*d*sizeof(emb) + G*4 (adjacency) + F*16 (two
    row_splits) + G*8 (visited word read+write), + entry ids + the result."""
    c = np.asarray(counters, dtype=np.int64)
    F, G, S = c[..., 0, :], c[..., 1, :], c[..., 2, :]
    return (S * d * emb_bytes + G * 12 + F * 16).sum(axis=-1) + n_enter * 4 + k_out * 12


def load_pmc_traffic(args):
    """HBM bytes per k_search launch from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json, written from tools/gpu_round.sh output): PMC counters cannot
    be read from inside the timed process.  Only reported when the workload matches."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            p = json.load(f)
    except (OSError, ValueError):
        return None
    w = p.get("workload", {})
    same = (w.get("items") == args.items and w.get("dim") == args.dim and w.get("ef") == args.ef and
            w.get("batch") == args.batch and w.get("scorer") == args.scorer and w.get("topk") == args.topk)
    if not same:
        return None
    # MI355X_MICROARCH.md (HBM / rocprofv3): KiB units; gfx950 FETCH_SIZE counts wide reads at half size
    b = (p["FETCH_SIZE_KiB"] * p.get("fetch_correction", 2.0) + p["WRITE_SIZE_KiB"]) * 1024.0
    return {"bytes_per_launch": b, "source": "profiles/pmc_latest.json (%s)" % p.get("kernel_version", "?")}


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from nann_amd import ops, retrieval, shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    topn = [args.ef] * 5 + [args.topk]
    t0 = time.time()
    g = None
    cache = None
    if args.index_cache:
        os.makedirs(args.index_cache, exist_ok=True)
        cache = os.path.join(args.index_cache,
                             f"idx_{args.items}_{args.dim}_{args.ef}_{args.graph}_{args.noise}_{rank}.npz")
        if os.path.exists(cache):
            z = np.load(cache)
            g = {"item_embs": z["item_embs"], "item_ids": z["item_ids"],
                 "nb_values": [z["nb_values_0"], z["nb_values_1"]],
                 "nb_row_splits": [z["nb_row_splits_0"], z["nb_row_splits_1"]],
                 "enter_points": z["enter_points"]}
    if g is None:
        g = synth.make_index(args.items, args.dim, ef=args.ef, mode=args.graph, noise=args.noise,
                             device=str(dev), shard=rank)
        if cache:
            np.savez(cache, item_embs=g["item_embs"], item_ids=g["item_ids"],
                     nb_values_0=g["nb_values"][0], nb_values_1=g["nb_values"][1],
                     nb_row_splits_0=g["nb_row_splits"][0], nb_row_splits_1=g["nb_row_splits"][1],
                     enter_points=g["enter_points"])
    index = retrieval.Index.from_dict(g, device=dev)
    mlp_w = synth.make_mlp_weights(args.dim) if args.scorer == "mlp" else None
    scorer = ops.Scorer(args.scorer, args.dim, weights=mlp_w)
    seq_host = synth.make_queries_from_centres(args.dim, args.batch, noise=args.noise)
    comm_seq = torch.as_tensor(seq_host).to(dev)
    setup_s = time.time() - t0

    sharded = shard.ShardedSearch(index, scorer, topn, world, merge=args.merge) if world > 1 else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    def step(i=None):
        q = ops.user_seq_mean(comm_seq)
        if i is not None:
            ev[i][0].record()
        r = retrieval.search(index, scorer, q, topn, want_counters=True)
        if i is not None:
            ev[i][1].record()
        if sharded is not None:
            return sharded.merge(r), r
        return (r.item_ids, r.scores), r

    for _ in range(args.warmup):
        out, r = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for i in range(args.steps):
        out, r = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the traversal kernel (rank-local launch, HIP events on its stream)
    kern_all = np.asarray([a.elapsed_time(b) for a, b in ev], np.float64)
    kern_ms = float(kern_all.mean())
    status = r.status.cpu().numpy()
    counters = r.counters.cpu().numpy().astype(np.int64)
    n_valid = int((status == 0).sum())
    bytes_per_launch = float(algorithmic_bytes(counters, args.dim, 2, len(g["enter_points"]),
                                               args.topk).sum())
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": "k_search", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "kernel_ms": round(kern_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_query": round(bytes_per_launch / args.batch, 1),
                "rows_scored_per_query": round(float(counters[:, 2, :].sum(1).mean()), 1)}
    pmc = load_pmc_traffic(args)
    if pmc is not None:
        roofline["traffic"] = pmc["bytes_per_launch"]
        roofline["traffic_source"] = pmc["source"]
    if args.scorer == "mlp":
        # 2*(256*256 + 256*128 + 128) flop per scored row; f32-input MFMA dense
        # peak 157.3 TFLOP/s (MI355X_MICROARCH.md).  The HBM figure is kept alongside.
        # The kernel computes the query half of layer 1 (W1q.q) once per query instead of once
        # per row, so the MFMA work it issues is d*256 + 256*128 MACs per row: `issued` below.
        rows = float(counters[:, 2, :].sum())
        flops = rows * 2.0 * (2 * args.dim * 256 + 256 * 128 + 128)
        issued = rows * 2.0 * (args.dim * 256 + 256 * 128)
        tf = flops / (kern_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "k_search (MLP scorer)", "achieved": round(tf, 2), "peak": 157.3,
                    "unit": "TFLOP/s", "frac": round(tf / 157.3, 4), "traffic": None,
                    "kernel_ms": round(kern_ms, 4),
                    "mfma_issued_TFLOPs": round(issued / (kern_ms * 1e-3) / 1e12, 2),
                    "mfma_issued_frac": round(issued / (kern_ms * 1e-3) / 1e12 / 157.3, 4),
                    "hbm_algorithmic_GBps": round(achieved, 1),
                    "rows_scored_per_query": roofline["rows_scored_per_query"]}

    # ---- whole-job throughput: every rank searched every query on its 1M-item shard
    qps = args.batch * args.steps / elapsed
    value = qps * world
    result = {
        "metric": "retrieval QPS @ recall@200 parity, 1M items/128-d",
        "value": round(value, 1), "unit": "queries/s x 1M-item shards searched",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f16 rows / f32 L2" if args.scorer == "l2" else "f16 rows / f32 MFMA MLP",
        "data": "synthetic",
        "config": {"workload": f"{args.items} items/GPU x {args.dim}-d f16, M=32 {args.graph} graph, "
                               f"ef_search={args.ef}, top-{args.topk}, "
                               + ("L2 scoring (BASELINE configs[1]" if args.scorer == "l2"
                                  else "3-layer MLP 256-128-1 scorer on MFMA (BASELINE configs[2]")
                               + (", sharded as configs[3]" if world > 1 else "") + ")",
                   "level_topn": topn, "batch": args.batch, "items_total": args.items * world,
                   "parallelism": f"item-id shards x{world}" if world > 1 else "single GPU",
                   "merge": args.merge if world > 1 else None},
        "qps_end_to_end": round(qps, 1), "valid_queries": n_valid, "setup_s": round(setup_s, 1),
        "roofline": roofline,
        # per-launch latency of the traversal for one batch (HIP events, this rank)
        "batch_latency_ms": {"batch": args.batch, "p50": round(float(np.percentile(kern_all, 50)), 4),
                             "p99": round(float(np.percentile(kern_all, 99)), 4),
                             "max": round(float(kern_all.max()), 4)},
    }
    if args.batch_sweep:
        # smaller request batches (B in {1, 64, 1024}): latency and QPS of one launch
        sweep = []
        for bsz in [int(x) for x in args.batch_sweep.split(",") if x]:
            bsz = max(1, min(bsz, args.batch))
            qb = ops.user_seq_mean(comm_seq[:bsz])
            ts = []
            for it in range(3 + 10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                retrieval.search(index, scorer, qb, topn, want_counters=False)
                e1.record()
                torch.cuda.synchronize()
                if it >= 3:
                    ts.append(e0.elapsed_time(e1))
            ts = np.asarray(ts)
            sweep.append({"batch": bsz, "ms_p50": round(float(np.percentile(ts, 50)), 4),
                          "ms_max": round(float(ts.max()), 4),
                          "qps": round(bsz / (float(np.percentile(ts, 50)) * 1e-3), 1)})
        result["batch_sweep"] = sweep

    if args.phase_ticks:
        from nann_amd import _lib
        rr = retrieval.search(index, scorer, ops.user_seq_mean(comm_seq), topn, want_phase_ticks=True)
        torch.cuda.synchronize()
        tk = rr.phase_ticks.cpu().numpy().astype(np.float64)
        tot = tk[:, :6].sum()  # the first six phases partition the query; the rest are sub-phases
        result["phase_breakdown"] = {
            "ticks_per_query": {n: round(float(tk[:, i].mean()), 1) for i, n in enumerate(_lib.PHASE_NAMES)},
            "fraction": {n: round(float(tk[:, i].sum() / tot), 4) for i, n in enumerate(_lib.PHASE_NAMES)}}

    if rank == 0 and world == 1:
        from oracle import oracle as O  # checker / CPU baseline only
        oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        osc = O.Scorer(args.scorer, args.dim, O.EMB_F16, mlp_w)
        qh = ops.user_seq_mean(comm_seq).cpu().numpy()
        cores = usable_cores()
        if not args.no_cpu_baseline:
            # bounded sample: grow the chunk until the time budget is used
            n_done, t_cpu, chunk = 0, 0.0, min(args.batch, 4 * cores)
            first = None
            while t_cpu < args.cpu_seconds and n_done < 50 * args.batch:
                sel = np.arange(n_done, n_done + chunk) % args.batch
                t1 = time.perf_counter()
                res = O.search_batch(oix, osc, qh[sel], topn, n_threads=cores)
                t_cpu += time.perf_counter() - t1
                if first is None:
                    first = (sel, res)
                n_done += chunk
                chunk = min(args.batch, chunk * 2)
            result["cpu_baseline"] = {"value": round(n_done / t_cpu, 1), "unit": "queries/s", "cores": cores,
                                      "kind": "port",
                                      "sample": f"{n_done} queries of the same batch, one query per thread, "
                                                f"{t_cpu:.1f} s"}
        else:
            sel = np.arange(min(args.batch, 4 * cores))
            first = (sel, O.search_batch(oix, osc, qh[sel], topn, n_threads=cores))
        # parity on the first CPU chunk: identical ids/scores => identical recall
        sel, (st, ids, scores, idx, ctr) = first
        gi = out[0].cpu().numpy()[sel]
        gs = out[1].cpu().numpy()[sel]
        ok = st == 0
        result["parity"] = {"queries_checked": int(len(sel)),
                            "status_equal": bool((status[sel] == st).all()),
                            "ids_equal": bool((gi[ok] == ids[ok]).all()),
                            "scores_bitwise_equal": bool((gs[ok].view(np.uint32) == scores[ok].view(np.uint32)).all())}
    if rank == 0 and world == 1:
        # recall@k of the traversal vs brute force under the same scorer (test_all,
        # main.py:194-237).  Brute force = score ALL items with the device scorer (parity-tested
        # against the oracle) + TopKV2 on the device; 16 queries.
        hits, nrec = 0, 0
        gidx = r.index.cpu().numpy()
        qd = ops.user_seq_mean(comm_seq)
        for b in range(min(16, args.batch)):
            if status[b]:
                continue
            sc_all = ops.blaze_score(scorer, qd[b], item_emb=index.item_embs)
            _, bi = ops.top_k(sc_all, args.topk)
            hits += len(set(bi.cpu().tolist()) & set(gidx[b].tolist()))
            nrec += args.topk
        result["recall_at_k_vs_bruteforce"] = round(hits / max(nrec, 1), 4)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
