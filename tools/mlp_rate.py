#!/usr/bin/env python3
"""Scoring rate of the stand-alone MLP scorer (nann_score -> k_score_mlp): rows/s and shader cycles per
256-row pass, without a traversal around it.  usage: tools/mlp_rate.py [d] [rows] [precision ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import ops, synth  # noqa: E402


def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 256 * 48
    precs = sys.argv[3:] or ["split", "exact"]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    table = (torch.randn((1 << 20, d), generator=g, device=dev) * 0.3).to(torch.float16)
    idx = torch.randint(0, 1 << 20, (n,), generator=g, device=dev, dtype=torch.int32)
    q = torch.randn(d, generator=g, device=dev)
    w = synth.make_mlp_weights(d)
    for prec in precs:
        sc = ops.Scorer("mlp", d, torch.float16, w, precision=prec)
        ts = []
        for it in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = ops.blaze_score(sc, q, table=table, indices=idx)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[1:]))
        if os.environ.get("NANN_MLP_TIMING") and prec == "split":  # a timing build wrote shader cycles of a pass there
            names = ["L1: burst reads landed", "L1: MFMAs issued", "L1: chain drained + PReLU/split", "L2: burst reads landed",
                     "L2: step total before hand-over", "hand-over: LDS write issued (incl. wait for the fetched slice)",
                     "barrier", "whole pass"]
            print("cycles of the last pass of block 0:", dict(zip(names, [int(x) for x in out[:8].float().cpu()])), flush=True)
        passes_per_cu = n / 256 / 256
        print(f"d={d} {prec}: {ms:.3f} ms for {n} rows = {n / ms / 1e3:.1f} M rows/s; "
              f"{ms * 1e3 / passes_per_cu:.2f} us per 256-row pass per CU; checksum {float(out.float().sum()):.6g}",
              flush=True)


if __name__ == "__main__":
    main()
