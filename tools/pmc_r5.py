#!/usr/bin/env python3
"""Round 5: rocprofv3 --pmc output of ONE pass (a directory of *counter_collection.csv) -> per-search-call counter totals of
the traversal's kernels, merged into profiles/pmc_latest.json under a workload tag (the tags bench.py looks its committed
`traffic` / `pmc_committed` fields up by).  A search call may be one k_search launch (L2, fused MLP) or the 12 launches of
the MLP's pipeline of phases: counters are SUMMED over every dispatch whose kernel name matches --kernels and divided by
the number of search calls the profiled command made (--calls = its --warmup + --steps).
usage: tools/pmc_r5.py <csv dir> --tag <workload tag> [--tag ...] --calls N --workload "<description>" --note "<kernel version>"
                       [--kernels 'k_search|k_mlp_phase']"""
import argparse
import collections
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tag", action="append", required=True)
    ap.add_argument("--calls", type=int, required=True)
    ap.add_argument("--workload", default="")
    ap.add_argument("--note", default="")
    ap.add_argument("--kernels", default="k_search|k_mlp_phase")
    a = ap.parse_args()
    pat = re.compile(a.kernels)
    tot = collections.defaultdict(float)
    n_disp = collections.defaultdict(int)
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if not pat.search(k):
                continue
            c = r["Counter_Name"]
            tot[c] += float(r["Counter_Value"])
            by_kernel[k[:72]][c] += float(r["Counter_Value"])
            n_disp[c] += 1
    if not tot:
        print("pmc_r5: no matching dispatches in", a.dir)
        return
    per = {c: v / a.calls for c, v in tot.items()}
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        out = json.load(open(path))
    except (OSError, ValueError):
        out = {"workloads": {}}
    for t in a.tag:
        e = out["workloads"].setdefault(t, {})
        if e.get("kernel_version") != a.note:  # a new kernel version: drop what the older passes left
            e.clear()
        e.update({"kernel": a.kernels + " (summed over the launches of a search call)", "kernel_version": a.note, "workload": a.workload,
                  "search_calls_in_pass": a.calls, "fetch_correction": 2.0})
        for c, v in per.items():
            e[{"FETCH_SIZE": "FETCH_SIZE_KiB", "WRITE_SIZE": "WRITE_SIZE_KiB"}.get(c, c)] = v
        e.setdefault("by_kernel_per_call", {}).update({k: {c: v / a.calls for c, v in d.items()} for k, d in by_kernel.items()})
    json.dump(out, open(path, "w"), indent=1)
    print("pmc_r5", a.tag, {c: round(v, 1) for c, v in per.items()}, "dispatches", dict(n_disp))


if __name__ == "__main__":
    main()
