#!/usr/bin/env python
"""bench.py -- retrieval QPS of the fused HNSW-with-model-scoring traversal on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...            (spawns one rank per GPU itself), or
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic UserBehavior-shaped
queries: comm_seq -> query vectors -> layered traversal (neighbour gather, visited-set
dedup, embedding gather + scoring, top-k) -> top-200 item ids.  Inputs are resident in HBM
when the timed region starts; every step sees a DIFFERENT batch of queries.

Primary workload (BASELINE.json configs[1]): 1M items x 128-d f16, M=32, ef_search=128
(level_topn = [128]*5 + [200]), L2 scoring, on each GPU; the graph comes from the shipped
index builder (nann_amd/csrc/nann_hnsw_build.hip, on the device: HNSW M=32, efConstruction=40 -- what
the reference gets from faiss.IndexHNSWFlat, build_hnsw_index.py:33-35; --graph hnsw_cpu: the same
algorithm by the host-side builder of rounds 1-2).  With N > 1 ranks the
corpus is N shards of 1M items (configs[3] shape: item-id sharding); every rank searches
every query on its shard, the per-shard top-200 lists are exchanged with one ncclAllGather
issued by the C ABI (nann_sharded_topk) and merged on the device.

Prints ONE JSON line (rank 0).  `roofline` prices the traversal kernel's ALGORITHMIC bytes
(SURVEY.md 8d formula over the kernel's own per-round counters, which the parity tests pin
to the oracle's) against 8 TB/s HBM; `cpu_baseline` is the oracle (oracle/nann_oracle.c, a
port of the reference's CPU op loops) timed on this box's cores.  At N = 1 the line also
carries `secondary`: configs[2] (MLP scorer), an HBM-honest stress run (2M x 256-d bf16,
ef=256: config 5's shard shape at half its size, table >> Infinity Cache) and the
B in {1, 64, 1024} latency sweep.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3  # f32-input MFMA dense peak
F16_MFMA_PEAK_TF = 2500.0  # f16/bf16 MFMA dense peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--items", type=int, default=1_000_000, help="items per GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="row dtype of item_embs")
    ap.add_argument("--ef", type=int, default=128)
    ap.add_argument("--topk", type=int, default=200)
    ap.add_argument("--batch", type=int, default=4096, help="queries per step")
    ap.add_argument("--graph", default="hnsw", choices=["hnsw", "hnsw_dense", "hnsw_cpu", "synth", "knn"],
                    help="hnsw = the shipped HNSW builder ON THE DEVICE (default since round 3: 1M x 128-d in ~0.6 s); "
                         "hnsw_dense = the same with the heuristic's keepPrunedConnections (rows filled to their cap: mean level-0 degree ~55); "
                         "hnsw_cpu = the same algorithm by the host-side C++ builder (rounds 1-2: 16-30 s); synth = exact-search "
                         "insertion graph built with torch (round-1 default, ~2 min at 1M); knn = full-degree exact k-NN rows")
    ap.add_argument("--noise", type=float, default=1.0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="primary workload only")
    ap.add_argument("--merge", default="device", choices=["device", "host"])
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the N>1 run (nccl = RCCL; gloo + NANN_BENCH_SHARED_GPU=1: "
                         "dry run of the multi-rank flow with every rank on cuda:0, exchange staged through the host)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: rccl = ncclAllGather issued by the C ABI; torch = torch.distributed collectives")
    ap.add_argument("--replicas", action="store_true",
                    help="--gpus N as N independent replicas of the single-GPU workload: every rank holds the whole index and "
                         "answers its own query stream, no collective on the data path -- the reference's only multi-GPU mode "
                         "(blaze-benchmark/benchmark/core/model.cc:192-235; SURVEY.md 8e 'Replicas')")
    ap.add_argument("--no-overlap-exchange", action="store_true",
                    help="N > 1, rccl transport: run each batch's all-gather + merge on the search stream instead of "
                         "a stream of its own underneath the next batch's search")
    ap.add_argument("--scorer", default="l2", choices=["l2", "mlp"],
                    help="l2 = BASELINE configs[1] (the headline metric); mlp = configs[2]: 256-128-1 MLP on MFMA")
    ap.add_argument("--mlp-precision", default="split", choices=["split", "exact"],
                    help="--scorer mlp: split = split-f16 operands on the 16-bit MFMA (scores within 1e-5); exact = f32 MFMA")
    ap.add_argument("--mlp-weights", default="metric", choices=["metric", "random"],
                    help="--scorer mlp: metric = layers constructed / fitted to rank like the index metric (recall means something); "
                         "random = random-init weights of the architecture (rounds 1-2)")
    ap.add_argument("--traversal", default="auto", choices=["auto", "lds_bitmap", "hbm_bitmap", "lds_hash", "lds_hash32"])
    ap.add_argument("--index-cache", default=None, help="directory to cache built indices in")
    ap.add_argument("--stress-items", type=int, default=4_000_000,
                    help="items of the HBM-honest secondary workload: configs[4]'s shard at its own size (4M x 256-d bf16, ef=256)")
    ap.add_argument("--phase-ticks", action="store_true",
                    help="one extra instrumented launch: per-phase time attribution")
    return ap.parse_args()


def algorithmic_bytes(counters, d, emb_bytes, n_enter, k_out=200):
    """bytes one query must move, from the per-round counters the kernel
    emits (frontier F, gathered G, scored S): S*d*sizeof(emb) + G*4 (adjacency) + F*16 (two
    row_splits) + G*8 (visited word read+write), + entry ids + the result.  Returns
    (total, hbm_only): the visited-set term never leaves LDS in the LDS modes, so `hbm_only`
    drops it."""
    c = np.asarray(counters, dtype=np.int64)
    F, G, S = c[..., 0, :], c[..., 1, :], c[..., 2, :]
    base = (S * d * emb_bytes + G * 4 + F * 16).sum(axis=-1) + n_enter * 4 + k_out * 12
    return base + (G * 8).sum(axis=-1), base


INFINITY_CACHE_BYTES = 256 << 20
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md "HBM": 8 TB/s spec, ~6.3 TB/s achievable (float4 copy)


def what_binds(table_bytes, row_bytes):
    """Which resource a gather of random `row_bytes` rows out of a `table_bytes` table runs against.
    Own measurement of that access pattern alone, tools/ubench_gather.hip (profiles/r2a_ubench_gather.txt: random 256-byte
    rows, nothing else to do): 7.34 TB/s on a 256 MiB table (it lives in the 256 MiB Infinity Cache), 7.19 TB/s on 1 GiB,
    6.40 TB/s on 4 GiB (HBM's stream rate, the guide's 6.3)."""
    mib = table_bytes / float(1 << 20)
    gather = 7339.0 if mib <= 256 else 7190.0 if mib <= 1024 else 6400.0
    if table_bytes <= INFINITY_CACHE_BYTES:
        binds = ("vector issue (the probe / scan / select phases and the DPP reductions of the scoring loop); the %.0f MiB row table is "
                 "resident in the 256 MiB Infinity Cache and the entry rows hit L2, so the algorithmic bytes are served faster than "
                 "HBM streams: `frac` is a fraction of the 8 TB/s table value, not HBM utilisation" % mib)
    else:
        binds = "HBM stream rate (the %.0f MiB row table is %.1fx the Infinity Cache)" % (mib, mib / 256.0)
    return {"achievable_hbm_GBs": HBM_ACHIEVABLE_GBS,
            "achievable_gather_GBs": gather,
            "achievable_source": "MI355X_MICROARCH.md (6.3 TB/s float4 copy); tools/ubench_gather.hip, profiles/r2a_ubench_gather.txt "
                                 "(random %d-byte rows alone on a table of this size class)" % 256,
            "table_MiB": round(mib, 1), "table_in_infinity_cache": bool(table_bytes <= INFINITY_CACHE_BYTES), "binds": binds}


def _pmc_entry(tag):
    """profiles/pmc_latest.json: {"workloads": {tag: {...}}} (one entry per workload the rocprofv3 PMC passes were run on,
    written by tools/pmc_to_json.py from the same bench command under rocprofv3; older files hold ONE entry at top level).
    PMC counters cannot be read from inside the timed process, so the line carries the committed measurement -- only
    for the workload (tag) it was collected on."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            p = json.load(f)
    except (OSError, ValueError):
        return None
    if "workloads" in p:
        return p["workloads"].get(tag)
    return p if p.get("workload_tag") == tag else None


def load_pmc_traffic(tag):
    """HBM bytes per k_search launch (FETCH_SIZE / WRITE_SIZE passes)."""
    p = _pmc_entry(tag)
    if p is None or "FETCH_SIZE_KiB" not in p:
        return None
    # MI355X_MICROARCH.md (HBM / rocprofv3): KiB units; gfx950 FETCH_SIZE counts wide reads at half size
    b = (p["FETCH_SIZE_KiB"] * p.get("fetch_correction", 2.0) + p["WRITE_SIZE_KiB"]) * 1024.0
    return {"bytes_per_launch": b, "source": "profiles/pmc_latest.json (%s)" % p.get("kernel_version", "?")}


def load_pmc_counters(tag):
    """matrix-pipe counters of the fused MLP traversal (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES pass)."""
    p = _pmc_entry(tag)
    if p is None or "SQ_VALU_MFMA_BUSY_CYCLES" not in p:
        return None
    # (PMC counters cannot be read from inside the timed process: these are the COMMITTED rocprofv3 passes of the same bench
    # command -- hence "_committed" in every name; `source` names the file and the kernel version they were taken on)
    out = {"source": "profiles/pmc_latest.json (%s)" % p.get("kernel_version", "?")}
    # SQ_VALU_MFMA_BUSY_CYCLES sums, over the SIMDs, the cycles their matrix pipe was busy; SQ_BUSY_CYCLES the cycles
    # an SQ (one per XCD ... summed over SEs) had waves: busy fraction = MFMA cycles / (4 SIMDs x CU-cycles the kernel ran)
    if p.get("GRBM_GUI_ACTIVE"):
        cu_cycles = p["GRBM_GUI_ACTIVE"] / 8.0 * 256  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        out["mfma_busy_frac_committed"] = round(p["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * cu_cycles), 4)
    for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE",
              "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in p:
            out[k + "_committed"] = p[k]
    sc = p.get("by_kernel", {}).get("k_mlp_phase_score")
    if sc:  # the pipeline of phases: the scoring launches on their own (their share of the call: profiles/r4*_phase_trace_*.txt)
        out["scoring_launches_committed"] = {k: sc[k] for k in ("mfma_busy_frac", "shader_clock_GHz_in_pass", "SQ_INSTS_MFMA", "FETCH_SIZE_KiB") if k in sc}
    return out


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


def to_bf16_bits(x_f32):
    """f32 -> bf16 bit patterns (uint16), round to nearest even (numpy has no bfloat16)."""
    u = np.ascontiguousarray(x_f32, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def n_clusters_for(items, ef):
    """Cluster count of the synthetic corpus.  SURVEY.md 8: "the generator must guarantee E >= ef and each round yields
    >= ef new nodes, else the reference returns InvalidArgument" (TopKV2 k > n, topk_op.cc:67-71).  A beam of width ef
    walks ~25-30 ef nodes of its neighbourhood; on HNSW(M=32, efConstruction=40) graphs -- mean level-0 degree ~17:
    each node keeps the ~8.7 candidates the selection heuristic lets through plus as many back-links -- a cluster
    smaller than that is exhausted before the third level-0 round (measured: 100k items in 256 clusters of 390,
    ef = 64: 20 % of the requests valid; clusters of >= 30 ef items: 100 %).  256 clusters wherever they stay that
    large (1M items / ef = 128 and up: the round-1/2 corpora unchanged), fewer for small corpora / wide beams."""
    return int(min(256, max(1, items // (30 * ef))))


def make_index(items, dim, ef, graph, noise, dtype, rank, dev, n_threads, cache_dir=None):
    """Seeded synthetic corpus + graph in the reference's array layout."""
    from nann_amd import index_build, synth
    cache = None
    ncl = n_clusters_for(items, ef)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        cache = os.path.join(cache_dir, f"idx_{items}_{dim}_{dtype}_{ef}_{graph}_{noise}_{rank}_c{ncl}.npz")
        if os.path.exists(cache):
            z = np.load(cache)
            return {"item_embs": z["item_embs"], "item_ids": z["item_ids"],
                    "nb_values": [z["nb_values_0"], z["nb_values_1"]],
                    "nb_row_splits": [z["nb_row_splits_0"], z["nb_row_splits_1"]],
                    "enter_points": z["enter_points"]}
    if graph in ("hnsw", "hnsw_dense", "hnsw_cpu"):
        import torch
        embs, _ = synth.make_corpus(items, dim, n_clusters=ncl, noise=noise, seed=1234, item_seed=1234 + 100 + 1000 * rank)
        x32 = embs.astype(np.float32)  # the builder sees exactly the values the index stores (f16-exact)
        if dtype == "bf16":
            bits = to_bf16_bits(x32)
            x32 = (bits.astype(np.uint32) << 16).view(np.float32)
        if graph in ("hnsw", "hnsw_dense"):  # HNSW construction on the device (csrc/nann_hnsw_build.hip): 1M x 128-d in ~0.6 s
            # hnsw_dense: the same builder with the heuristic's keepPrunedConnections on -- rows fill up to their cap
            with torch.cuda.device(dev):
                rows = (torch.as_tensor(bits.view(np.int16)).to(dev).view(torch.bfloat16) if dtype == "bf16"
                        else torch.as_tensor(embs).to(dev))
                ex = index_build.build_hnsw_gpu(rows, num_neighbors=32, ef_construction=40, seed=1236 + 1000 * rank,
                                                keep_pruned=graph == "hnsw_dense")
                del rows
        else:  # the host-side builder (csrc/host/hnsw_build.cpp), multi-threaded
            raw = index_build.build_hnsw(x32, num_neighbors=32, ef_construction=40, seed=1236 + 1000 * rank,
                                         n_threads=n_threads)
            ex = index_build.export_levels(raw, start_level=2)
        del x32
        if len(ex["enter_points"]) < ef:
            raise RuntimeError(f"only {len(ex['enter_points'])} enter points for ef={ef}: corpus too small")
        g = {"item_embs": bits if dtype == "bf16" else embs,
             "item_ids": synth.make_item_ids(items, seed=1235 + 1000 * rank) + rank * items,
             "nb_values": [v.astype(np.int32) for v in ex["nb_values"]],
             "nb_row_splits": ex["nb_row_splits"], "enter_points": ex["enter_points"].astype(np.int32)}
    else:
        g = synth.make_index(items, dim, ef=ef, mode="hnsw" if graph == "synth" else "knn", noise=noise,
                             device=str(dev), shard=rank, n_clusters=ncl)
        if dtype == "bf16":
            g["item_embs"] = to_bf16_bits(g["item_embs"].astype(np.float32))
    if cache:
        np.savez(cache, item_embs=g["item_embs"], item_ids=g["item_ids"],
                 nb_values_0=g["nb_values"][0], nb_values_1=g["nb_values"][1],
                 nb_row_splits_0=g["nb_row_splits"][0], nb_row_splits_1=g["nb_row_splits"][1],
                 enter_points=g["enter_points"])
    return g


def make_query_batches(dim, batch, n_batches, noise, dev, seed=4321, n_clusters=256):
    """comm_seq f16[n_batches, B, 50, d] generated on the device (seeded): each history is 7..50 draws
    around one of the corpus' cluster centres, zero-padded tail (convert_UB_to_tfrecord.py:121-137)."""
    import torch
    from nann_amd import synth
    centres = torch.as_tensor(synth.make_centres(dim, n_clusters, 1234)).to(dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = []
    for _ in range(n_batches):
        lens = torch.randint(7, 51, (batch,), generator=gen, device=dev)
        cl = torch.randint(0, n_clusters, (batch,), generator=gen, device=dev)
        x = centres[cl][:, None, :] + noise * torch.randn((batch, 50, dim), generator=gen, device=dev)
        x = x * (1.0 / np.sqrt(dim))
        mask = torch.arange(50, device=dev)[None, :] < lens[:, None]
        out.append((x * mask[:, :, None]).to(torch.float16))
    return out


def run_workload(name, args, dev, rank, world, cfg, sharded=None, want_cpu=False, want_parity=True, want_recall=True):
    """Build/load the index, time `steps` passes, return the result dict for this workload."""
    import torch
    import torch.distributed as dist
    from nann_amd import ops, retrieval, synth
    items, dim, ef, topk, batch = cfg["items"], cfg["dim"], cfg["ef"], cfg["topk"], cfg["batch"]
    steps, warmup, scorer_kind, dtype = cfg["steps"], cfg["warmup"], cfg["scorer"], cfg["dtype"]
    topn = [ef] * 5 + [topk]
    cores = usable_cores()
    t0 = time.time()
    g = cfg.get("_index") or make_index(items, dim, ef, cfg["graph"], args.noise, dtype, rank, dev,
                                        max(1, cores // world), args.index_cache)
    index = retrieval.Index.from_dict(g, device=dev)
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    # the MLP's weights: constructed + last layer fitted so that it ranks like the index metric (what the reference's
    # training produces; synth.make_mlp_weights_metric) -- with random-init weights (--mlp-weights random) recall against
    # brute force under the same scorer is ~0.4 and "QPS @ recall parity" says nothing
    mlp_w = None
    if scorer_kind == "mlp":
        rows = g["item_embs"][:: max(1, items // 65536)]
        if dtype == "bf16":
            rows = (rows.astype(np.uint32) << 16).view(np.float32)
        mlp_w = (synth.make_mlp_weights(dim) if args.mlp_weights == "random"
                 else synth.make_mlp_weights_metric(dim, rows))
    precision = cfg.get("mlp_precision", "exact") if scorer_kind == "mlp" else "exact"
    scorer = ops.Scorer(scorer_kind, dim, tdt, weights=mlp_w, precision=precision)
    n_batches = min(steps + warmup, 40)  # (the driver's 3 + 20 steps: every step its own batch; long matrix-core warm-ups wrap around)
    seqs = make_query_batches(dim, batch, n_batches, args.noise, dev, seed=cfg.get("query_seed", 4321),
                              n_clusters=n_clusters_for(items, ef))
    setup_s = time.time() - t0
    retrieval.set_traversal_mode(cfg.get("traversal", "auto"))

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    # per-call options (nann_search_options): a sharded run that overlaps its exchanges leaves RCCL a few workgroup slots
    sopt = sharded.search_options(overlap=cfg.get("overlap_exchange", False)) if sharded is not None else None
    timed_comm = sharded is not None and sharded.transport == "rccl" and sharded.comm is not None
    if timed_comm:
        sharded.comm.set_timing(True)  # HIP events around pack | all-gather | merge of every exchange, on the exchange's stream
    if cfg.get("mlp_form"):  # on the options the sharded path made (its slot reserve stays)
        if sopt is None:
            sopt = retrieval.search_options(mlp_form=cfg["mlp_form"])
        else:
            sopt.mlp_form = retrieval.MLP_FORMS[cfg["mlp_form"]]

    def step(j, i=None):
        q = ops.user_seq_mean(seqs[j % n_batches])
        if i is not None:
            ev[i][0].record()
        r = retrieval.search(index, scorer, q, topn, want_counters=True, options=sopt)
        if i is not None:
            ev[i][1].record()
        if sharded is not None:
            return sharded.merge(r, overlap=cfg.get("overlap_exchange", False)), r, q
        return (r.item_ids, r.scores), r, q

    import gc
    gc.collect()  # handles of an earlier workload (hipFree of GBs synchronises the device) go now, not inside the timed loop
    for j in range(warmup):
        out, r, q = step(j)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    host_marks = []
    for i in range(steps):
        out, r, q = step(warmup + i, i)
        host_marks.append(time.perf_counter())
    if sharded is not None:
        sharded.wait()  # overlapped exchanges: all of them belong to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    host_enqueue_ms = [round((b - a) * 1e3, 3) for a, b in zip([t_start] + host_marks[:-1], host_marks)]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the traversal kernel (rank-local launch, HIP events on its stream)
    kern_all = np.asarray([a.elapsed_time(b) for a, b in ev], np.float64)
    kern_ms = float(kern_all.mean())
    breakdown = None
    if sharded is not None:
        # what one step costs on THIS rank, part by part (HIP events on the streams the parts run on): the search launch,
        # and -- last exchange of the run -- pack, all-gather, merge.  Gathered from every rank below: an efficiency
        # number from an 8-GPU node then explains itself (step = max(search, exchange) when the exchange overlaps the
        # next search, their sum otherwise).
        mine = {"rank": rank, "search": round(kern_ms, 4)}
        if timed_comm:
            try:
                mine.update(sharded.comm.last_breakdown())
                mine["shards"], mine["rccl_ranks"] = sharded.comm.ranks()
            except Exception as e:  # noqa: BLE001
                mine["error"] = repr(e)
        else:
            mine["note"] = "torch.distributed transport: exchange parts are not timed separately"
        box = [None] * world
        dist.all_gather_object(box, mine)
        breakdown = box
    status = r.status.cpu().numpy()
    counters = r.counters.cpu().numpy().astype(np.int64)
    ok = status == 0
    n_valid = int(ok.sum())
    tot_b, hbm_b = algorithmic_bytes(counters[ok], dim, 2, len(g["enter_points"]), topk)
    bytes_per_launch = float(tot_b.sum())
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    rows = float(counters[ok][:, 2, :].sum())
    roofline = {"bound": "hbm", "kernel": "k_search", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "kernel_ms": round(kern_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_query": round(bytes_per_launch / max(n_valid, 1), 1),
                # the same without the visited-set term, which never leaves LDS in the LDS modes
                "frac_without_visited_set_bytes": round(float(hbm_b.sum()) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "rows_scored_per_query": round(rows / max(n_valid, 1), 1),
                "gathered_per_query": round(float(counters[ok][:, 1, :].sum()) / max(n_valid, 1), 1)}
    roofline.update(what_binds(float(items) * dim * 2, dim * 2))
    roofline["frac_of_achievable_hbm"] = round(achieved / HBM_ACHIEVABLE_GBS, 4)
    pmc = load_pmc_traffic(name)
    if pmc is not None:
        roofline["traffic"] = pmc["bytes_per_launch"]
        roofline["traffic_source"] = pmc["source"]
        # fabric-side rate of the COMMITTED counter pass over THIS run's kernel time (the counters cannot be read in-process)
        roofline["counter_traffic_GBs"] = round(pmc["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, 1)
    else:
        roofline["counter_traffic_GBs"] = None
    if scorer_kind == "mlp":
        # What the traversal EXECUTES on the matrix cores per scored row (round 4: both precisions run on the table of
        # pre-projected item halves with layer 2 resident in LDS, nann_mlp5.h; without a table -- NANN_PREPROJECT=0, no room in
        # HBM -- the forms that read the embedding rows run, priced by their own counts):
        #   split-f16  layer 2 only, 3 f16 products per f32 MAC: 3 * 2*256*128 = 196 608 flop = 6 MFMAs of 32x32x16 per
        #              (32 rows x 32 outputs x 16 k); forms that still run layer 1 add 2 * 2*d*256
        #   exact f32  layer 2 only: 2*256*128 = 65 536 flop = 16 MFMAs of 32x32x2 per (32 rows x 32 outputs x 32 k); the
        #              form that runs layer 1 adds 2*d*256
        # `frac` prices THOSE (never more than the pipe can do); SURVEY.md 8(d)'s nominal 2*(2d*256 + 256*128 + 128) per
        # row -- which counts the hoisted query half and the looked-up item half of layer 1 as if they were computed per
        # candidate -- is the second figure.  The issued count is cross-checked against SQ_INSTS_MFMA of the committed
        # rocprofv3 pass (profiles/pmc_latest.json).
        # which form ran: the planner's own answer for the timed calls (nann_search_plan: table read? pipeline of phases?)
        table_form = bool(r.plan["table"])
        phased = bool(r.plan["phased"])
        nominal = rows * 2.0 * (2 * dim * 256 + 256 * 128 + 128)
        if precision == "split":
            per_row = 3 * 2.0 * 256 * 128 + (0 if table_form else 2 * 2.0 * dim * 256)
            peak, flop_per_mfma = F16_MFMA_PEAK_TF, 32 * 32 * 16 * 2
            flops_note = "v_mfma_f32_32x32x16_f16 on split-f16 operands (3 products per layer-2 MAC%s)" % (
                "; layer 1: item half from the pre-projected table, query half hoisted" if table_form else ", 2 per layer-1 MAC")
        else:
            per_row = 2.0 * 256 * 128 + (0 if table_form else 2.0 * dim * 256)
            peak, flop_per_mfma = F32_MFMA_PEAK_TF, 32 * 32 * 2 * 2
            flops_note = "v_mfma_f32_32x32x2_f32 (%s)" % ("layer 2 only: item half of layer 1 from the pre-projected table, query half hoisted"
                                                        if table_form else "query half of layer 1 hoisted")
        executed = rows * per_row
        tf = executed / (kern_ms * 1e-3) / 1e12
        tf_nominal = nominal / (kern_ms * 1e-3) / 1e12
        # bytes the scorer gathers per row: the 1 KB row of the table (f32 x 256) instead of the d x 2 B embedding row
        row_bytes = 1024 if table_form else dim * 2
        tot_b2, _ = algorithmic_bytes(counters[ok], row_bytes // 2, 2, len(g["enter_points"]), topk)
        # (why the MLP lines run batch 1024 where the headline runs 4096: the rate is flat from 1024 up -- 396 k at 4096 against
        # 383-388 k at 1024, profiles/r4g_bench_mlp_split_b*.json -- and a chunk of the pipeline of phases is 1024 queries)
        kernel_label = ("pipeline of phases (nann_mlp6.h): k_mlp_phase_score<%s> x 5 rounds [dominant] + k_search<phase> x 6; "
                        "kernel_ms = the whole call" % precision) if phased else (
                        "k_search (MLP scorer, %s%s)" % (precision, ", layer 2 resident in LDS" if table_form else ""))
        # BOTH rooflines: executed matrix-core flops against the MFMA peak of the operand type, and the bytes
        # the scorer gathers (1 KB table rows) + the traversal's adjacency / id traffic against HBM; `bound` = the one the
        # kernel sits closer to, and achieved / peak / unit / frac are that one's
        hbm_gbps = float(tot_b2.sum()) / (kern_ms * 1e-3) / 1e9
        frac_mfma, frac_hbm = tf / peak, hbm_gbps / HBM_PEAK_GBS
        bound = "hbm" if frac_hbm > frac_mfma else "mfma"
        roofline = {"bound": bound, "kernel": kernel_label,
                    "achieved": round(hbm_gbps, 1) if bound == "hbm" else round(tf, 2),
                    "peak": HBM_PEAK_GBS if bound == "hbm" else peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                    "frac": round(max(frac_mfma, frac_hbm), 4),
                    "frac_mfma": round(frac_mfma, 4), "frac_hbm": round(frac_hbm, 4),
                    "mfma_achieved_TFLOPs": round(tf, 2), "mfma_peak_TFLOPs": peak,
                    "algorithmic_bytes_per_launch": float(tot_b2.sum()),
                    "traffic": None, "kernel_ms": round(kern_ms, 4),
                    "flops": "EXECUTED matrix-core flops per scored row x rows scored: " + flops_note,
                    "executed_flops_per_row": per_row, "executed_flops_per_launch": executed,
                    "mfma_instructions_per_launch": executed / flop_per_mfma,
                    # SURVEY.md 8(d)'s figure beside it (includes work that is hoisted / looked up, not executed):
                    # (no ratio to the peak for this one: on the exact line it counts 3x the work the pipe executes)
                    "nominal_8d_flops_per_launch": nominal, "nominal_8d_TFLOPs": round(tf_nominal, 2),
                    # tools/ubench_mfma.hip (profiles/r2_ubench_mfma.txt): a bare loop of this MFMA on every SIMD
                    # holds 1.6-1.9 PFLOP/s (f16) -- the chip clocks ~1.75 GHz under it, not 2.4
                    "frac_of_measured_mfma_loop": (round(tf / 1830.0, 4) if precision == "split" else None),
                    "hbm_algorithmic_GBps": round(float(tot_b2.sum()) / (kern_ms * 1e-3) / 1e9, 1),
                    "hbm_bytes_gathered_per_row": row_bytes,
                    "rows_scored_per_query": roofline["rows_scored_per_query"]}
        roofline.update(what_binds(float(items) * row_bytes, row_bytes))
        roofline["binds"] = ("neither roofline: vector issue beside the MFMAs at the board's power limit (PReLU + hi/lo split of every "
                             "activation: ~6 vector instructions per MFMA, DESIGN.md 4.2); " + roofline["binds"])
        pmc = load_pmc_counters(name)
        if pmc is not None:
            roofline["pmc_committed"] = pmc
            if pmc.get("SQ_INSTS_MFMA_committed"):
                roofline["executed_over_committed_pmc_instructions"] = round(executed / flop_per_mfma / pmc["SQ_INSTS_MFMA_committed"], 4)
        pmc = load_pmc_traffic(name)
        if pmc is not None:
            roofline["traffic"] = pmc["bytes_per_launch"]
            roofline["traffic_source"] = pmc["source"]
            roofline["counter_traffic_GBs"] = round(pmc["bytes_per_launch"] / (kern_ms * 1e-3) / 1e9, 1)

    qps = batch * steps / elapsed
    try:  # the planner's choice for the timed calls and how many queries of the last one were rerun on the bitmap kernel
        plan_info, reruns = dict(r.plan), int(r.reruns())
    except Exception:
        plan_info, reruns = None, None
    res = {"workload": name, "qps_end_to_end": round(qps, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
           **({"exchange_breakdown_ms": breakdown,
               "rccl_ranks_seen": max([b.get("rccl_ranks", 0) for b in breakdown] + [0])} if breakdown else {}),
           "plan": plan_info, "reruns_last_step": reruns, "index_probe": index.probe,
           "batch": batch, "steps": steps, "valid_queries": n_valid, "setup_s": round(setup_s, 1),
           # host time to ENQUEUE each step (no sync inside the loop): a value near ms_per_step = the host blocked
           "host_enqueue_ms": host_enqueue_ms[:8],
           "n_enter": int(len(g["enter_points"])),
           "mean_degree_l0": round(float(len(g["nb_values"][0])) / items, 2),
           "traversal": cfg.get("traversal", "auto"), "roofline": roofline,
           **({"mlp_weights": args.mlp_weights} if scorer_kind == "mlp" else {}),
           "batch_latency_ms": {"batch": batch, "p50": round(float(np.percentile(kern_all, 50)), 4),
                                "p99": round(float(np.percentile(kern_all, 99)), 4),
                                "max": round(float(kern_all.max()), 4)}}

    if args.phase_ticks:
        from nann_amd import _lib
        rr = retrieval.search(index, scorer, q, topn, want_phase_ticks=True)
        torch.cuda.synchronize()
        tk = rr.phase_ticks.cpu().numpy().astype(np.float64)
        tot = tk[:, :6].sum()  # the first six phases partition the query; the rest are sub-phases
        res["phase_breakdown"] = {
            "ticks_per_query": {n: round(float(tk[:, i].mean()), 1) for i, n in enumerate(_lib.PHASE_NAMES)},
            "fraction": {n: round(float(tk[:, i].sum() / tot), 4) for i, n in enumerate(_lib.PHASE_NAMES)}}

    # ---- checker legs: oracle = test infrastructure, never inside the timed region
    n_check = min(cfg.get("parity_queries", 64), batch)
    if want_parity or want_cpu:
        from oracle import oracle as O
        code = O.EMB_F16 if dtype == "f16" else O.EMB_BF16
        oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
        osc = O.Scorer(scorer_kind, dim, code, mlp_w)
        qh = q.cpu().numpy()
        threads = max(1, cores // world)
        first = None
        if want_cpu and rank == 0 and world == 1:
            # bounded sample: grow the chunk until the time budget is used
            n_done, t_cpu, chunk = 0, 0.0, min(batch, max(n_check, 4 * cores))
            while t_cpu < cfg.get("cpu_seconds", args.cpu_seconds) and n_done < 50 * batch:
                sel = np.arange(n_done, n_done + chunk) % batch
                t1 = time.perf_counter()
                ores = O.search_batch(oix, osc, qh[sel], topn, n_threads=cores)
                t_cpu += time.perf_counter() - t1
                if first is None:
                    first = (sel, ores)
                n_done += chunk
                chunk = min(batch, chunk * 2)
            res["cpu_baseline"] = {"value": round(n_done / t_cpu, 1), "unit": "queries/s", "cores": cores,
                                   "kind": "port",
                                   "sample": f"{n_done} queries of the last batch, one query per thread, "
                                             f"{t_cpu:.1f} s"}
            if scorer_kind == "mlp":
                # the oracle scores with scalar fmaf chains in the canonical order (what parity needs); the reference's
                # CPU path scores through XLA / Eigen GEMMs (blaze_xla_predictor.cc:360-459).  The rate THAT would have
                # on these cores: the same schedule with each round's candidates scored as one f32 GEMM batch
                # (oracle/gemm_baseline.py, numpy on OpenBLAS, one query per thread)
                res["cpu_baseline"]["scorer"] = "scalar fmaf chains in the canonical order (the parity oracle's rate)"
                from oracle import gemm_baseline as GB
                n_g, t_g, chunk = 0, 0.0, max(2 * cores, 16)
                while t_g < 0.6 * cfg.get("cpu_seconds", args.cpu_seconds) and n_g < 20 * batch:
                    sel_g = np.arange(n_g, n_g + chunk) % batch
                    gst, gids, gsc, dt_g = GB.search_batch(g, mlp_w, qh[sel_g], topn, n_threads=cores)
                    if n_g == 0:  # the port answers like the oracle (status equal, id lists up to near-ties)
                        nn = min(len(first[0]), chunk)
                        same = sum(int((gids[b] == first[1][1][b]).all()) for b in range(nn) if first[1][0][b] == 0)
                        g_check = {"queries": nn, "status_equal": bool((gst[:nn] == first[1][0][:nn]).all()),
                                   "id_lists_identical": same}
                    t_g += dt_g
                    n_g += chunk
                    chunk = min(batch, chunk * 2)
                res["cpu_baseline_gemm"] = {"value": round(n_g / t_g, 1), "unit": "queries/s", "cores": cores, "kind": "port",
                                            "scorer": "f32 GEMMs per scoring round (numpy / OpenBLAS), one query per thread: "
                                                      "how the reference's CPU path scores (XLA / Eigen GEMMs)",
                                            "sample": f"{n_g} queries of the last batch, {t_g:.1f} s",
                                            "agrees_with_oracle": g_check}
        if first is None:
            sel = np.arange(n_check)
            first = (sel, O.search_batch(oix, osc, qh[sel], topn, n_threads=threads))
        sel, (st, ids, scores, idx, ctr) = first
        sel, st, ids, scores = sel[:n_check], st[:n_check], ids[:n_check], scores[:n_check]
        if world == 1 and precision == "split":
            # scores are equal within 1e-5, not bitwise: tie-aware comparison of the sorted lists
            gx, gs = r.index.cpu().numpy()[sel], r.scores.cpu().numpy()[sel]
            okc = (st == 0) & (status[sel] == 0)
            kinds = [O.tolerant_parity(gx[b], gs[b], idx[:n_check][b], scores[b]) for b in np.nonzero(okc)[0]]
            errs = [float(np.max(np.abs(gs[b] - scores[b]) / np.maximum(1.0, np.abs(scores[b]))))
                    for b in np.nonzero(okc)[0] if (gx[b] == idx[:n_check][b]).all()]
            res["parity"] = {"queries_checked": int(len(sel)), "tolerance": "1e-5 * max(1, |score|), tie-aware ids",
                             "status_equal": bool((status[sel] == st).all()),
                             "ids_identical": kinds.count("exact"), "near_tie_only": kinds.count("near-tie"),
                             "diverged": kinds.count("diverged"),
                             "max_rel_score_err_on_identical": max(errs) if errs else None}
        elif world == 1 or sharded is None:  # one GPU, or a replica (its own whole index): this rank's answers against the oracle
            gi, gs = out[0].cpu().numpy()[sel], out[1].cpu().numpy()[sel]
            okc = st == 0
            res["parity"] = {"queries_checked": int(len(sel)),
                             "status_equal": bool((status[sel] == st).all()),
                             "ids_equal": bool((gi[okc] == ids[okc]).all()),
                             "scores_bitwise_equal": bool((gs[okc].view(np.uint32) == scores[okc].view(np.uint32)).all()),
                             "counters_equal": bool((counters[sel][okc] == ctr[:n_check][okc]).all())}
        else:
            # merged result vs the oracle's per-shard searches merged with the same rule
            box = [None] * world
            dist.all_gather_object(box, (st, ids, scores))
            if rank == 0:
                s_all = np.stack([np.where((b[0] == 0)[:, None], b[2], -np.inf) for b in box], 1)
                i_all = np.stack([np.where((b[0] == 0)[:, None], b[1], 0) for b in box], 1)
                exp_i, exp_s = [], []
                for b in range(len(sel)):
                    rc, ms, mi = O.merge_topk(s_all[b], i_all[b], topk)
                    exp_i.append(mi); exp_s.append(ms)
                gi, gs = out[0].cpu().numpy()[sel], out[1].cpu().numpy()[sel]
                res["parity"] = {"queries_checked": int(len(sel)), "shards": world,
                                 "merged_ids_equal": bool((gi == np.stack(exp_i)).all()),
                                 "merged_scores_bitwise_equal": bool(
                                     (gs.view(np.uint32) == np.stack(exp_s).view(np.uint32)).all())}
    if want_recall and (world == 1 or sharded is None):
        # recall@k of the traversal vs brute force under the same scorer (test_all, main.py:194-237):
        # score ALL items with the device scorer (parity-tested against the oracle) + TopKV2; 16 queries
        hits, nrec = 0, 0
        gidx = r.index.cpu().numpy()
        for b in range(min(16, batch)):
            if status[b]:
                continue
            sc_all = ops.blaze_score(scorer, q[b], item_emb=index.item_embs)
            _, bi = ops.top_k(sc_all, topk)
            hits += len(set(bi.cpu().tolist()) & set(gidx[b].tolist()))
            nrec += topk
        res["recall_at_k_vs_bruteforce"] = round(hits / max(nrec, 1), 4)
    retrieval.set_traversal_mode("auto")
    res["_index"], res["_handles"] = g, (index, scorer, seqs)
    return res


def hbm_roofline(bytes_per_launch, ms, kernel, **extra):
    """the `roofline` block of an HBM-bound line: algorithmic bytes of ONE launch / its duration against 8 TB/s"""
    ach = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "kernel_ms": round(ms, 4),
            "algorithmic_bytes_per_launch": float(bytes_per_launch), **extra}


def batch_sweep(handles, topn, sizes, dim, n_enter):
    """smaller request batches (B in {1, 64, 1024}): latency and QPS of one launch, each with its roofline
    (the launch's own counters through 8(d)'s byte formula / its median duration: a batch that fills 1 or 64 of 256 CUs is
    priced against the whole chip's 8 TB/s all the same -- that is what the line is there to show)"""
    import torch
    from nann_amd import ops, retrieval
    index, scorer, seqs = handles
    sweep = []
    for bsz in sizes:
        qb = ops.user_seq_mean(seqs[0][:bsz])
        ts = []
        for it in range(3 + 10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = retrieval.search(index, scorer, qb, topn, want_counters=it == 0)
            e1.record()
            torch.cuda.synchronize()
            if it == 0:
                okb = r.status.cpu().numpy() == 0
                tot_b, _ = algorithmic_bytes(r.counters.cpu().numpy().astype(np.int64)[okb], dim, 2, n_enter, topn[5])
                plan = r.plan
            if it >= 3:
                ts.append(e0.elapsed_time(e1))
        ts = np.asarray(ts)
        p50 = float(np.percentile(ts, 50))
        sweep.append({"batch": bsz, "ms_p50": round(p50, 4),
                      "ms_max": round(float(ts.max()), 4),
                      "qps": round(bsz / (p50 * 1e-3), 1),
                      "plan": "%s, %d threads x %d workgroups" % (plan["visited_set"], plan["threads"], plan["workgroups"]),
                      "roofline": hbm_roofline(float(tot_b.sum()), p50, "k_search (one launch of %d queries)" % bsz)})
    return sweep


def eval_graph_rate(handles, dim, n_enter, n_users=1024, pmc_prefix="eval_graph_f3_", g=None, dtype="f16", n_check=4):
    """f3: users/s of the evaluation graph's traversal in one kernel (nann_search_eval): the reference's defaults
    (config.py:50-58: 3/1/1 rounds, top 400/200/100, 200 returned) and a wide setting above the serving kernels' 1024.
    roofline: SURVEY.md 8(d)'s byte formula over the kernel's OWN counters (nann_search_eval_ex: rows walked F, neighbours
    gathered G, rows scored S per user; pinned to the op-by-op spelling by test_fused_eval_graph_counters_...)."""
    import torch
    from nann_amd import ops, retrieval
    index, scorer, seqs = handles
    q = ops.user_seq_mean(seqs[0][:n_users])
    out = {}
    for name, cfg in (("defaults", ((3, 1, 1), (400, 200, 100), 200)), ("top_k_2000_1000_500", ((3, 1, 1), (2000, 1000, 500), 1000))):
        ts = []
        for it in range(1 + 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = retrieval.search_eval(index, scorer, q, *cfg, want_counters=it == 0)
            e1.record()
            torch.cuda.synchronize()
            if it == 0:
                c = r.counters.cpu().numpy().astype(np.int64)[r.status.cpu().numpy() == 0]
                F, G, S = c[:, 0], c[:, 1], c[:, 2]
                bytes_launch = float((S * dim * 2 + G * 4 + F * 16 + G * 8).sum() + len(c) * (n_enter * 4 + cfg[2] * 12))
            if it >= 1:
                ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts))
        res = {"users": int(q.shape[0]), "ms": round(ms, 3), "users_per_s": round(q.shape[0] / (ms * 1e-3), 1),
               "failed": int((r.status != 0).sum()), "mean_rows": round(float(r.n_out.float().mean()), 1),
               "rows_scored_per_user": round(float(S.mean()), 1), "gathered_per_user": round(float(G.mean()), 1),
               "roofline": hbm_roofline(bytes_launch, ms, "k_search_eval")}
        pmc = load_pmc_traffic(pmc_prefix + name) if pmc_prefix else None  # (the committed rocprofv3 passes of tools/eval_bench.py: the same two launches)
        if pmc is not None:
            res["roofline"]["traffic"] = pmc["bytes_per_launch"]
            res["roofline"]["traffic_source"] = pmc["source"]
            res["roofline"]["counter_traffic_GBs"] = round(pmc["bytes_per_launch"] / (ms * 1e-3) / 1e9, 1)
        res["roofline"]["achievable_hbm_GBs"] = HBM_ACHIEVABLE_GBS
        res["roofline"]["binds"] = ("one 1 024-thread workgroup per CU (the bitmap window owns the LDS): scoring at what a lone workgroup draws with 32 registers of "
                                    "rows in flight (~46 GB/s per CU), the other phases -- walk, owners' pass, scan / emit, top-k -- chains of dependent "
                                    "trips and barriers with nothing to overlap them (DESIGN.md 4.4)")
        if g is not None and n_check > 0:  # checker leg, outside the timed region: a few users against oracle_search_eval, bit for bit
            from oracle import oracle as O
            oix = O.Index(g["item_embs"], g["item_ids"], g["nb_values"], g["nb_row_splits"], g["enter_points"])
            osc = O.Scorer("l2", dim, {"f16": O.EMB_F16, "bf16": O.EMB_BF16, "f32": O.EMB_F32}[dtype])
            qh, st, n_out = q.cpu().numpy(), r.status.cpu().numpy(), r.n_out.cpu().numpy()
            ids, scs, idx = r.item_ids.cpu().numpy(), r.scores.cpu().numpy(), r.index.cpu().numpy()
            same = 0
            for b in range(0, n_users, max(1, n_users // n_check))[:n_check]:
                rc, eids, esc, eidx = O.search_eval(oix, osc, qh[b], *cfg)
                k = len(eids) if rc == 0 else 0
                same += bool(st[b] == rc and n_out[b] == k and (idx[b, :k] == eidx[:k]).all() and (ids[b, :k] == eids[:k]).all()
                             and (scs[b, :k].view(np.uint32) == np.asarray(esc[:k], np.float32).view(np.uint32)).all())
            res["parity"] = {"users_checked": n_check, "bit_identical_to_oracle_search_eval": same}
        if name == "defaults":
            out.update(res)
        else:
            out[name] = res
    out["kernel"] = ("k_search_eval, LDS form (search_eval_lds, nann_eval.h): seen bitmap in LDS doubling as the round's staging area, owner-thread scan "
                     "against visited, marks as a list, bin-ranked top-k; parity: tests/test_search_gpu.py, tools/eval_bench.py")
    return out


def attention_model_rate(handles, dim, topn, precision, n_users=512):
    """f2: the serving signature with the reference's attention + DNN model as the scorer (nann_search_model:
    per-user projection once per request, then the fused traversal), random-init weights of that architecture"""
    import tempfile
    import torch
    from nann_amd import ops, retrieval, synth
    index = handles[0]
    with tempfile.TemporaryDirectory() as tmp:
        ops.save_scorer_dir(tmp, "attention", synth.make_attn_weights(dim, 64), precision=precision)
        model = ops.Model(tmp, dim, 50)
    g = torch.Generator(device=index.device).manual_seed(77)
    seq = (torch.randn((n_users, 50, 64), generator=g, device=index.device) * 0.5).to(torch.float16)
    ts = []
    n_warm = 40 if precision == "split" else 8  # steady-state clocks (see the MLP lines)
    for it in range(n_warm + 8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = retrieval.search_model(index, model, seq, topn, want_counters=it == 0)
        e1.record()
        if it == 0:
            torch.cuda.synchronize()
            okq = r.status.cpu().numpy() == 0
            ctr = r.counters.cpu().numpy().astype(np.int64)[okq]
            plan = r.plan
        if it >= n_warm:
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    out = {"users": n_users, "precision": precision, "ms": round(ms, 3),
           "queries_per_s": round(n_users / (ms * 1e-3), 1), "failed": int((r.status != 0).sum())}
    # roofline (VERDICT r4 missing 5): what the traversal EXECUTES per scored row on the pre-projected table
    #   split-f16  220 x v_mfma_f32_32x32x16_f16 per 32 rows (attention logits, softmax-weighted sum, DNN 128-64-32-1 attention half)
    #   f32 form   608 x v_mfma_f32_32x32x2_f32 per 32 rows
    # and the bytes it gathers: the table row is 384 f32 = 1 536 B (q_ 256 + e W1e 128) instead of the 2 d-byte embedding row
    rows = float(ctr[:, 2, :].sum())
    mfma_per_32, flop_per_mfma, peak = (220, 32 * 32 * 16 * 2, F16_MFMA_PEAK_TF) if precision == "split" else (608, 32 * 32 * 2 * 2, F32_MFMA_PEAK_TF)
    row_bytes = 1536 if plan["table"] else dim * 2
    executed = rows / 32.0 * mfma_per_32 * flop_per_mfma
    tf = executed / (ms * 1e-3) / 1e12
    tot_b, _ = algorithmic_bytes(ctr, row_bytes // 2, 2, int(index.enter_points.numel()), topn[5])
    gbps = float(tot_b.sum()) / (ms * 1e-3) / 1e9
    frac_mfma, frac_hbm = tf / peak, gbps / HBM_PEAK_GBS
    bound = "hbm" if frac_hbm > frac_mfma else "mfma"
    out["roofline"] = {"bound": bound, "kernel": "k_search<attention model, %s> + the per-user projection (kernel_ms = the whole nann_search_model call)" % precision,
                       "achieved": round(gbps, 1) if bound == "hbm" else round(tf, 2), "peak": HBM_PEAK_GBS if bound == "hbm" else peak,
                       "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(max(frac_mfma, frac_hbm), 4),
                       "frac_mfma": round(frac_mfma, 4), "frac_hbm": round(frac_hbm, 4), "traffic": None, "kernel_ms": round(ms, 4),
                       "executed_flops_per_launch": executed, "mfma_instructions_per_32_rows": mfma_per_32,
                       "algorithmic_bytes_per_launch": float(tot_b.sum()), "hbm_bytes_gathered_per_row": row_bytes,
                       "rows_scored_per_query": round(rows / max(len(ctr), 1), 1),
                       "note": "only valid when the table form ran (plan.table): %s" % bool(plan["table"])}
    out["form"] = ("item-only layers pre-projected per (model, index), keys and weights resident in LDS per scoring call: "
                   "nann_attn_proj.h wg_score_attn_res" if precision == "split" else
                   "f32 MFMA on the same pre-projected table: nann_attn_kernels.h wg_score_attn<PROJ>")
    if precision == "split":
        pmc = load_pmc_counters("attention_model_f2_split")
        if pmc is not None:
            out["roofline"]["pmc_committed"] = pmc
    return out


def strip(res):
    return {k: v for k, v in res.items() if not k.startswith("_")}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU through
    torch.distributed.run and pass its output through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    from nann_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("NANN_BENCH_SHARED_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = None
    replicas = bool(args.replicas) and world > 1
    if replicas:  # the process group only carries the barriers and the max-over-ranks of the timed region
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if args.dist_backend == "gloo" else "nccl", rank=rank, world_size=world,
                                **({} if args.dist_backend == "gloo" else {"device_id": dev}))
    elif world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
            args.transport = "torch"
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        exchange_note = None
        try:
            sharded = shard.ShardedSearch([args.ef] * 5 + [args.topk], world, rank, merge=args.merge,
                                          transport=args.transport)
            ok = torch.ones(1, device=dev)
        except Exception as e:  # the library's own RCCL communicator could not be made: say so, keep going
            exchange_note = f"{args.transport} transport failed ({e!r}); fell back to torch.distributed collectives"
            ok = torch.zeros(1, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            if exchange_note is None:
                exchange_note = "another rank could not create the C-ABI communicator; fell back to torch.distributed"
            args.transport = "torch"
            sharded = shard.ShardedSearch([args.ef] * 5 + [args.topk], world, rank, merge=args.merge,
                                          transport="torch")

    primary_cfg = {"items": args.items, "dim": args.dim, "ef": args.ef, "topk": args.topk, "batch": args.batch,
                   "steps": args.steps, "warmup": args.warmup, "scorer": args.scorer, "dtype": args.dtype,
                   "graph": args.graph, "traversal": args.traversal, "mlp_precision": args.mlp_precision,
                   "overlap_exchange": world > 1 and not replicas and args.transport == "rccl" and not args.no_overlap_exchange,
                   "query_seed": 4321 + (1000 * rank if replicas else 0)}  # replicas: every rank its own query stream
    is_headline = (args.items == 1_000_000 and args.dim == 128 and args.ef == 128 and args.topk == 200
                   and args.dtype == "f16")
    tag = f"{args.items}x{args.dim}{args.dtype}_ef{args.ef}_k{args.topk}_b{args.batch}_{args.scorer}_{args.graph}"
    if args.scorer == "mlp" and args.mlp_precision == "exact":
        tag += "_exact"  # (its own counters in profiles/pmc_latest.json)
    prim = run_workload(tag, args, dev, rank, world, primary_cfg, sharded=sharded,
                        want_cpu=not args.no_cpu_baseline, want_parity=True, want_recall=True)

    qps = prim["qps_end_to_end"]
    desc = (f"{args.items} items/GPU x {args.dim}-d {args.dtype}, M=32 graph from "
            + {"hnsw": "the shipped HNSW builder on the device (efConstruction=40)",
               "hnsw_dense": "the shipped HNSW builder on the device (efConstruction=40, keepPrunedConnections: rows filled to their cap)",
               "hnsw_cpu": "the shipped host-side HNSW builder (efConstruction=40)", "synth": "synth.py (exact-search insertion)",
               "knn": "exact k-NN rows"}[args.graph]
            + f", ef_search={args.ef}, top-{args.topk}, "
            + ("L2 scoring" if args.scorer == "l2" else "3-layer MLP 256-128-1 scorer on MFMA")
            + (" (BASELINE configs[1])" if is_headline and args.scorer == "l2" else
               " (BASELINE configs[2])" if is_headline else "")
            + (f", {world} independent replicas (no collective)" if replicas else
               f", sharded {world}-way as configs[3]" if world > 1 else ""))
    result = {
        "metric": "retrieval QPS @ recall@200 parity, 1M items/128-d",
        # whole-job throughput = the units ALL ranks processed per second.  The metric's unit is one query searched over a
        # 1M-item / 128-d index; with N ranks the corpus is N x 1M items (weak scaling: per-GPU work fixed, the corpus grows
        # with N), every rank searches every query on its own 1M-item shard, so a step of B queries is N x B units, and a
        # unit only counts once its shard's list has been exchanged and merged into the complete answer (the timed region
        # ends behind the last merge).  Ideal weak scaling: value(N) = N x value(1); the complete answers per second over
        # the N x larger corpus (= value / N, flat under ideal scaling) are under `complete_answers_per_s`.
        # Replicas (--replicas): every rank answers B complete queries per step over the whole index: N x B queries/s, plainly.
        "value": round(qps * world, 1),
        "unit": "queries/s" if world == 1 or replicas else "shard-queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": prim["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": f"{args.dtype} rows / f32 " + ("L2" if args.scorer == "l2" else "MFMA MLP"),
        "data": "synthetic",
        "config": {"workload": desc, "level_topn": [args.ef] * 5 + [args.topk], "batch": args.batch,
                   "items_total": args.items * world,
                   "parallelism": (f"replicas x{world} (independent query streams, no collective: the reference's mode, "
                                   "blaze-benchmark/benchmark/core/model.cc:192-235)" if replicas else
                                   f"item-id shards x{world}" if world > 1 else "single GPU"),
                   **({"exchange_overlapped_with_next_search": primary_cfg["overlap_exchange"]} if world > 1 and not replicas else {}),
                   "exchange": (f"{args.transport} all-gather + {args.merge} merge" if world > 1 and not replicas else None)},
        "qps_end_to_end": qps,
        "complete_answers_per_s": qps * world if replicas else qps,
        "weak_scaling": ({"units_per_step": args.batch * world, "unit": f"one query answered over the whole {args.items}-item index by one replica",
                          "complete_answers_per_s": qps * world, "efficiency_definition": "value(N) / (N x value(1))"} if replicas else
                         {"units_per_step": args.batch * world, "unit": f"one query searched over one {args.items}-item shard (merged into its complete answer)",
                          "complete_answers_per_s": qps,
                          "efficiency_definition": "value(N) / (N x value(1)) = complete_answers_per_s(N) / value(1)"}),
    }
    if world > 1 and not replicas and exchange_note:
        result["exchange_note"] = exchange_note
    for k in ("valid_queries", "setup_s", "n_enter", "mean_degree_l0", "traversal", "plan", "reruns_last_step", "index_probe",
              "exchange_breakdown_ms", "rccl_ranks_seen", "roofline", "batch_latency_ms",
              "cpu_baseline", "parity", "recall_at_k_vs_bruteforce", "phase_breakdown", "host_enqueue_ms"):
        if k in prim:
            result[k] = prim[k]

    if rank == 0 and world == 1 and not args.no_secondary and args.scorer == "l2":
        sec = {}
        try:
            sec["batch_sweep"] = batch_sweep(prim["_handles"], [args.ef] * 5 + [args.topk], [1, 64, 1024], args.dim, prim["n_enter"])
        except Exception as e:  # a failing extra must not take the headline line with it
            sec["batch_sweep"] = {"error": repr(e)}
        try:
            sec["eval_graph_f3"] = eval_graph_rate(prim["_handles"], args.dim, prim["n_enter"], g=prim["_index"], dtype=args.dtype)
        except Exception as e:
            sec["eval_graph_f3"] = {"error": repr(e)}
        for prec in ("split", "exact"):
            try:
                sec["attention_model_f2_" + prec] = attention_model_rate(prim["_handles"], args.dim,
                                                                         [args.ef] * 5 + [args.topk], prec)
            except Exception as e:
                sec["attention_model_f2_" + prec] = {"error": repr(e)}
        for prec, key in (("split", "mlp_configs2_split_f16"), ("exact", "mlp_configs2_exact_f32")):
            try:  # BASELINE configs[2]: same index, MLP scorer on the matrix cores
                # steady state: the chip ramps its clocks over the first ~0.2 s of a matrix-core workload (a 5-step run of
                # this traversal measures 2.98 ms per launch, the same launch after 100 warm-up steps 2.62 ms:
                # profiles/r4e_power.txt), so these lines warm up for a few hundred ms before the timed steps
                cfg = dict(primary_cfg, scorer="mlp", mlp_precision=prec, batch=min(args.batch, 1024),
                           steps=40 if prec == "split" else 20, warmup=100 if prec == "split" else 40, _index=prim["_index"])
                # the tag a direct `--scorer mlp --batch 1024` run has: real batch, real scorer (the PMC entries key on it)
                mlp_tag = (f"{args.items}x{args.dim}{args.dtype}_ef{args.ef}_k{args.topk}_b{cfg['batch']}_mlp_{args.graph}"
                           + ("_exact" if prec == "exact" else ""))
                sec[key] = strip(run_workload(mlp_tag, args, dev, rank, world, cfg,
                                              want_cpu=prec == "split" and not args.no_cpu_baseline,
                                              want_parity=True, want_recall=prec == "split"))
            except Exception as e:
                sec[key] = {"error": repr(e)}
        prim.pop("_handles", None)
        prim.pop("_index", None)
        torch.cuda.empty_cache()
        try:  # degree sensitivity: configs[1] on the DENSE graph family -- the same corpus, the same builder with
            # the heuristic's keepPrunedConnections on (rows filled to their cap: mean level-0 degree ~52 of 64 instead of ~17)
            cfg = dict(primary_cfg, graph="hnsw_dense", steps=10, warmup=3, cpu_seconds=min(args.cpu_seconds, 4.0))
            dense = run_workload(tag.replace("_" + args.graph, "_hnsw_dense"), args, dev, rank, world, cfg,
                                 want_cpu=not args.no_cpu_baseline, want_parity=True, want_recall=True)
            dense.pop("_handles", None)
            dense.pop("_index", None)
            sec["dense_graph"] = strip(dense)
            del dense
            torch.cuda.empty_cache()
        except Exception as e:
            sec["dense_graph"] = {"error": repr(e)}
        try:  # HBM-honest: config 5's shard shape (256-d bf16, ef=256) at a size whose table is 4x the Infinity Cache
            cfg = {"items": args.stress_items, "dim": 256, "ef": 256, "topk": 200, "batch": 2048, "steps": 5,
                   "warmup": 2, "scorer": "l2", "dtype": "bf16", "graph": "hnsw", "traversal": "auto"}
            cfg["cpu_seconds"] = min(args.cpu_seconds, 5.0)
            stress = run_workload(f"{args.stress_items}x256bf16_ef256", args, dev, rank, world, cfg,
                                  want_cpu=not args.no_cpu_baseline, want_parity=True, want_recall=False)
            sec["hbm_stress_config5_shape"] = strip(stress)
            try:  # f3 on the same shard: the LDS form sweeping the id space in windows (round 6; this shard: 4 of them)
                ev = eval_graph_rate(stress["_handles"], 256, stress["n_enter"], pmc_prefix=f"eval_graph_f3_{args.stress_items}x256bf16_",
                                     g=stress["_index"], dtype="bf16", n_check=2)
                ev["kernel"] = "k_search_eval, LDS form in windows (search_eval_win, nann_eval.h): shards beyond ~1 M items; parity: tests/test_eval_edges_gpu.py, tools/eval_bench.py"
                sec["eval_graph_f3_config5_shape"] = ev
            except Exception as e:
                sec["eval_graph_f3_config5_shape"] = {"error": repr(e)}
            try:  # the same rows and beam under the MLP scorer: wide beams take the bitmap plan + pre-projected scorer
                cfg = dict(cfg, scorer="mlp", mlp_precision="split", batch=1024, steps=20, warmup=40,
                           parity_queries=32, _index=stress["_index"])
                stress.pop("_handles", None)
                sec["mlp_config5_shape_split_f16"] = strip(run_workload(
                    f"{args.stress_items}x256bf16_ef256_mlp_split", args, dev, rank, world, cfg, want_cpu=False,
                    want_parity=True, want_recall=False))
            except Exception as e:
                sec["mlp_config5_shape_split_f16"] = {"error": repr(e)}
        except Exception as e:
            sec["hbm_stress_config5_shape"] = {"error": repr(e)}
        result["secondary"] = sec
        # the headline's table lives in the Infinity Cache; the line that IS bound by HBM's stream rate is named
        stress_line = sec.get("hbm_stress_config5_shape") or {}
        if isinstance(result.get("roofline"), dict) and isinstance(stress_line.get("roofline"), dict):
            sr = stress_line["roofline"]
            result["roofline"]["hbm_bound_evidence"] = {
                "line": "secondary.hbm_stress_config5_shape", "workload": stress_line.get("workload"),
                "table_MiB": sr.get("table_MiB"), "achieved_GBs": sr.get("achieved"), "frac_of_8TBs": sr.get("frac"),
                "frac_of_achievable_hbm": sr.get("frac_of_achievable_hbm"), "counter_traffic_GBs": sr.get("counter_traffic_GBs")}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
