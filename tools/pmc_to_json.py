#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>.txt (tools/gpu_r3.sh: per-kernel means of the rocprofv3 --pmc passes) -> profiles/pmc_latest.json,
the per-workload counter record bench.py attaches to its roofline objects (`traffic`, `mfma_busy_frac`).
usage: tools/pmc_to_json.py <gpurun tag> <kernel version note>
       tools/pmc_to_json.py --phased <gpurun tag> <note>   gpurun_out/pmc_phased_{split,exact}_<tag>.txt (tools/gpu_r4.sh
           pmc_phased: per-kernel SUMS over a run of the MLP pipeline of phases) -> the two MLP entries, per search call"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def phased(tag, note):
    """One search call of the pipeline = 6 traversal launches + 5 prefix + 5 scoring launches + the fallback launch:
    counters summed over all of them per call (so that mfma_busy_frac is the call's), the scoring launches' own beside."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    out = json.load(open(path))
    for prec, tags, wl in (("split", ["1000000x128f16_ef128_k200_b1024_mlp_hnsw", "1000000x128f16_ef128_k200_b4096_l2_hnsw_mlp_split"],
                            "BASELINE configs[2]: 1M x 128-d f16, ef=128, top-200, MLP 256-128-1 split-f16, batch 1024"),
                           ("exact", ["1000000x128f16_ef128_k200_b1024_mlp_hnsw_exact", "1000000x128f16_ef128_k200_b4096_l2_hnsw_mlp_exact"],
                            "BASELINE configs[2]: 1M x 128-d f16, ef=128, top-200, MLP 256-128-1 exact f32, batch 1024")):
        f = os.path.join(ROOT, "gpurun_out", f"pmc_phased_{prec}_{tag}.txt")
        if not os.path.exists(f):
            continue
        by = {}
        for line in open(f):
            m = re.match(r"PMCPH \S+ (\S+) (\S+) (\S+) dispatches (\d+) sum (\S+)(?: dur_ns (\S+))?", line)
            if not m:
                continue
            _, kern, ctr, n, v, dur = m.groups()
            by.setdefault(kern, {})[ctr] = float(v)
            by[kern]["dispatches"] = int(n)
            if dur and ctr == "GRBM_GUI_ACTIVE":
                by[kern]["dur_ns"] = float(dur)
        calls = by["k_mlp_phase_score"]["dispatches"] / 5.0
        e = {"kernel": "pipeline of phases (nann_mlp6.h): k_search<phase> x 6, k_mlp_phase_score x 5, fallback launch",
             "kernel_version": note, "workload": wl, "search_calls_in_pass": calls, "fetch_correction": 2.0, "by_kernel": {}}
        for kern, c in by.items():
            per = {k: v / calls for k, v in c.items() if k not in ("dispatches", "dur_ns")}
            if "FETCH_SIZE" in per:
                per["FETCH_SIZE_KiB"] = per.pop("FETCH_SIZE")
            if "WRITE_SIZE" in per:
                per["WRITE_SIZE_KiB"] = per.pop("WRITE_SIZE")
            if c.get("dur_ns") and "GRBM_GUI_ACTIVE" in c:
                per["shader_clock_GHz_in_pass"] = round(c["GRBM_GUI_ACTIVE"] / 8.0 / c["dur_ns"], 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "GRBM_GUI_ACTIVE" in per:
                per["mfma_busy_frac"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * per["GRBM_GUI_ACTIVE"] / 8.0 * 256), 4)
            e["by_kernel"][kern] = per
            for k, v in per.items():
                if k not in ("shader_clock_GHz_in_pass", "mfma_busy_frac"):
                    e[k] = e.get(k, 0.0) + v
        for t in tags:
            out["workloads"][t] = e
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({t: sorted(e) for t, e in out["workloads"].items()}, indent=1)[:1500])


def main():
    if sys.argv[1] == "--phased":
        return phased(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
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
        if c == "SQ_INSTS_MFMA" and "k_search" in k and (", 6, 512>" in k or ", 11, 512>" in k) and mean > most:
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
