#!/usr/bin/env python3
"""Scoring rate of the stand-alone attention + DNN scorer (nann_attn_score), both precisions, without a
traversal around it.  usage: tools/attn_rate.py [d] [rows]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import ops, synth  # noqa: E402


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 256 * 16
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    table = (torch.randn((1 << 20, d), generator=g, device=dev) * 0.3).to(torch.float16)
    idx = torch.randint(0, 1 << 20, (n,), generator=g, device=dev, dtype=torch.int32)
    u = (torch.randn((1, 50, 64), generator=g, device=dev) * 0.3).to(torch.float16)
    w = synth.make_attn_weights(d, 64)
    ref = None
    for prec in ("split", "exact"):
        sc = ops.AttnScorer(d, 50, torch.float16, w, precision=prec)
        kt, upad = sc.prepare(u)
        ts = []
        for it in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = sc.score(kt[0], upad[0], table=table, indices=idx)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[1:]))
        o = out.float().cpu().numpy()
        err = "" if ref is None else f"; max |split - exact| / max(1, |exact|) = {np.abs(ref - o).max() / max(1.0, np.abs(o).max()):.2e}"
        ref = o if ref is None else ref
        if os.environ.get("NANN_ATTN_TIMING") and prec == "split":  # a timing build wrote per-step shader cycles there
            print("cycles of the last pass of block 0 [q1 x4 | q_/att x16 | softmax + a | DNN1 x8 | DNN2, 3]:",
                  [int(x) for x in o[:5]], flush=True)
        print(f"d={d} {prec}: {ms:.3f} ms for {n} rows = {n / ms / 1e3:.1f} M rows/s; "
              f"{ms * 1e3 / (n / 256 / 256):.2f} us per 256-row pass per CU{err}", flush=True)


if __name__ == "__main__":
    main()
