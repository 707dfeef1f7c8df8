#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>.txt (tools/gpu_r3.sh: per-kernel means of the rocprofv3 --pmc passes) -> profiles/pmc_latest.json,
the per-workload counter record bench.py attaches to its roofline objects (`traffic`, `mfma_busy_frac`).
usage: tools/pmc_to_json.py <gpurun tag> <kernel version note>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, note = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    rows = {}
    for line in open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}.txt")):
        m = re.match(r"(\S+) (\S+) (.*) dispatches (\d+) mean (\S+) min", line)
        if not m:
            continue
        group, counter, kernel, n, mean = m.group(1), m.group(2), m.group(3), int(m.group(4)), float(m.group(5))
        rows.setdefault(group, {})[(counter, kernel)] = (mean, n)

    def pick(groups, counter, kernel_sub):
        for g in groups:
            for (c, k), (mean, n) in rows.get(g, {}).items():
                if c == counter and kernel_sub in k:
                    return mean, k, n
        return None

    out = {"workloads": {}}

    def hbm(groups, kernel_sub, tags, workload):
        f, w = pick(groups, "FETCH_SIZE", kernel_sub), pick(groups, "WRITE_SIZE", kernel_sub)
        if not f or not w:
            return
        e = {"kernel": f[1], "kernel_version": note, "workload": workload, "FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0],
             "fetch_correction": 2.0,
             "note": f"mean over {f[2]} k_search dispatches, separate --pmc passes (profiles/{tag}_pmc.txt)"}
        for t in tags:
            out["workloads"].setdefault(t, {}).update(e)

    hbm(["l2_fetch", "l2_write"], "k_search<16, 0, 2, 0, 512>", ["1000000x128f16_ef128_k200_b4096_l2_hnsw"],
        "BASELINE configs[1]: 1M x 128-d f16, ef=128, top-200, L2, batch 4096")
    hbm(["stress_fetch", "stress_write"], "k_search<32, 1, 3, 0, 1024>", ["2000000x256bf16_ef256", "2000000x256bf16_ef256_k200_b2048_l2_hnsw"],
        "config 5's shard shape at half size: 2M x 256-d bf16, ef=256, batch 2048")
    hbm(["shard4m_fetch", "shard4m_write"], "k_search<32, 1, 3, 0, 1024>", ["4000000x256bf16_ef256_k200_b2048_l2_hnsw"],
        "config 5's shard: 4M x 256-d bf16, ef=256, batch 2048")
    # the fused MLP traversals = the k_search instance that issued the most MFMA instructions in the pass
    def mlp(groups_a, fetch_g, write_g, tags, workload):
        kern, most = None, 0.0
        for (c, k), (mean, n) in rows.get(groups_a[0], {}).items():
            if c == "SQ_INSTS_MFMA" and "k_search" in k and mean > most:
                kern, most = k, mean
        if not kern:
            return
        e = {"kernel": kern, "kernel_version": note, "workload": workload}
        for g in groups_a:
            for (c, k), (mean, n) in rows.get(g, {}).items():
                if k == kern:
                    e[c] = mean
        f, w = pick([fetch_g], "FETCH_SIZE", kern[:60]), pick([write_g], "WRITE_SIZE", kern[:60])
        if f and w:  # HBM traffic of the same kernel (separate --pmc passes; KiB, gfx950 FETCH_SIZE x 2: MI355X_MICROARCH.md)
            e.update({"FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0], "fetch_correction": 2.0})
        for t in tags:
            out["workloads"].setdefault(t, {}).update(e)

    mlp(["mlp_a", "mlp_b"], "mlp_fetch", "mlp_write",
        ["1000000x128f16_ef128_k200_b1024_mlp_hnsw", "1000000x128f16_ef128_k200_b4096_l2_hnsw_mlp_split"],
        "BASELINE configs[2]: 1M x 128-d f16, ef=128, top-200, MLP 256-128-1 split-f16, batch 1024")
    mlp(["mlpx_a", "mlpx_b"], "mlpx_fetch", "mlpx_write",
        ["1000000x128f16_ef128_k200_b1024_mlp_hnsw_exact", "1000000x128f16_ef128_k200_b4096_l2_hnsw_mlp_exact"],
        "BASELINE configs[2]: 1M x 128-d f16, ef=128, top-200, MLP 256-128-1 exact f32, batch 1024")
    # the attention model's fused traversal (tools/attn_bench.py): split-f16 form = the instance with the most MFMAs
    attn_kernel, most = None, 0.0
    for (c, k), (mean, n) in rows.get("attn_a", {}).items():
        if c == "SQ_INSTS_MFMA" and "k_search" in k and ", 6, 512>" in k and mean > most:
            attn_kernel, most = k, mean
    if attn_kernel:
        e = {"kernel": attn_kernel, "kernel_version": note,
             "workload": "f2: attention + DNN model, split-f16, pre-projected, 512 users on configs[1]'s index"}
        for g in ("attn_a", "attn_b"):
            for (c, k), (mean, n) in rows.get(g, {}).items():
                if k == attn_kernel:
                    e[c] = mean
        f, w = pick(["attn_fetch"], "FETCH_SIZE", attn_kernel[:60]), pick(["attn_write"], "WRITE_SIZE", attn_kernel[:60])
        if f and w:
            e.update({"FETCH_SIZE_KiB": f[0], "WRITE_SIZE_KiB": w[0], "fetch_correction": 2.0})
        out["workloads"].setdefault("attention_model_f2_split", {}).update(e)
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        old = json.load(open(path))
        if "workloads" in old:
            for t, e in old["workloads"].items():
                out["workloads"].setdefault(t, e)
    except (OSError, ValueError):
        pass
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({t: sorted(e) for t, e in out["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
