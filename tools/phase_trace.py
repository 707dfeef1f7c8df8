#!/usr/bin/env python3
"""Per-launch durations of the MLP pipeline of phases (nann_mlp6.h) from a rocprofv3 --kernel-trace csv.

A batch is the same sequence of launches every step (traversal stage, block prefix, scoring launch, ...): the trace's
kernels in start order, cut at every k_search launch that follows a fallback launch, give one sequence per step;
prints, per position, the kernel, its mean duration and the mean gap to the launch in front of it -- over the LAST
`keep` steps (steady-state clock).  usage: phase_trace.py <kernel_trace.csv> [keep]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    for key in ("k_mlp_phase_score", "k_mlp_phase_prefix", "k_search", "k_user_seq_mean", "k_mlp_preproject"):
        if key in name:
            if key == "k_search":
                return "k_search<phase>" if ", 9," in name or ",9," in name else "k_search<fallback>"
            if key == "k_mlp_phase_score":
                m = re.search(r"k_mlp_phase_score<(\w+), (\d+)>", name)
                return key + ("<exact>" if m and m.group(1) == "true" else "") + ("/var%s" % m.group(2) if m and m.group(2) != "0" else "")
            return key
    return name[:40]


def main():
    path = sys.argv[1]
    keep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    steps, cur = [], []
    for s, e, n in rows:
        if not n.startswith(("k_search<", "k_mlp_phase_")):
            continue
        if n == "k_search<phase>" and cur and cur[-1][2] == "k_search<fallback>":
            steps.append(cur)
            cur = []
        cur.append((s, e, n))
    if cur:
        steps.append(cur)
    length = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
    steps = [s for s in steps if len(s) == length][-keep:]
    dur, gap = defaultdict(list), defaultdict(list)
    for st in steps:
        for i, (s, e, n) in enumerate(st):
            dur[i].append(e - s)
            if i:
                gap[i].append(s - st[i - 1][1])
    tot_d = tot_g = 0.0
    print(f"{len(steps)} steps of {length} launches")
    for i in range(length):
        d = sum(dur[i]) / len(dur[i]) / 1e3
        g = sum(gap[i]) / len(gap[i]) / 1e3 if i else 0.0
        tot_d += d
        tot_g += g
        print(f"{i:3d} {steps[0][i][2]:28s} {d:9.1f} us   gap {g:7.1f} us")
    span = sum(st[-1][1] - st[0][0] for st in steps) / len(steps) / 1e3
    print(f"sum of durations {tot_d:.1f} us, of gaps {tot_g:.1f} us, first start -> last end {span:.1f} us")


if __name__ == "__main__":
    main()
