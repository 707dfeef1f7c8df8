#!/usr/bin/env python3
"""The serving signature with the reference's attention + DNN model (f2) on BASELINE configs[1]'s index: queries/s of
nann_search_model for the split-f16 form (NANN_MLP_MAPPING=1 in the environment: the round-2 form without the
pre-projection) and the f32 form.  usage: tools/attn_bench.py [index cache dir] [users ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cache = sys.argv[1] if len(sys.argv) > 1 else None
    users = [int(x) for x in sys.argv[2:]] or [512, 1024]
    from nann_amd import retrieval
    dev = torch.device("cuda:0")
    g = bench.make_index(1_000_000, 128, 128, "hnsw", 1.0, "f16", 0, dev, bench.usable_cores(), cache)
    index = retrieval.Index.from_dict(g, device=dev)
    topn = [128] * 5 + [200]
    for n in users:
        for prec in ("split", "exact"):
            if prec == "exact" and n != users[0]:
                continue
            r = bench.attention_model_rate((index, None, None), 128, topn, prec, n_users=n)
            r["mapping"] = os.environ.get("NANN_MLP_MAPPING", "default")
            print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
