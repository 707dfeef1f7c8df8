#!/usr/bin/env python3
"""The C++ serving host (nann_amd/csrc/host/nann_serve.cpp) under closed-loop load on BASELINE configs[1]'s index:
build the 1M x 128-d graph with the shipped builder, write it in the reference's file layout, run the host.
usage: tools/serve_bench.py [items] [clients ...]"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nann_amd import serving, synth  # noqa: E402


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    clients = [int(x) for x in sys.argv[2:]] or [64, 512, 2048]
    g = bench.make_index(items, 128, 128, "hnsw", 1.0, "f16", 0, torch.device("cuda"), bench.usable_cores())
    with tempfile.TemporaryDirectory() as d:
        disk = dict(g)
        disk["item_embs"] = np.asarray(g["item_embs"]).astype(np.float32)
        disk["nb_values"] = [np.asarray(v).astype(np.int64) for v in g["nb_values"]]
        synth.save_index(disk, d)
        del g, disk
        torch.cuda.empty_cache()
        for lanes in (1, 2, 4):
            for c in clients:
                r = serving.run_serve_host(d, d, 128, clients=c, seconds=3.0, max_batch=1024, max_wait_us=200, ef=128,
                                           topk=200, lanes=lanes)
                print(json.dumps(r), flush=True)
        # the batch size a lane settles at follows the client count; a larger cap for the many-client points
        for lanes, c in ((1, 4096), (2, 4096)):
            r = serving.run_serve_host(d, d, 128, clients=c, seconds=3.0, max_batch=2048, max_wait_us=200, ef=128,
                                       topk=200, lanes=lanes)
            print(json.dumps(r), flush=True)
        # the model-scoring configs behind the same front end: the 256-128-1 MLP (weights directory, split-f16) and the
        # reference's attention + DNN model read from a frozen GraphDef, as BlazeXlaOp.graph_def names it
        from nann_amd import frozen_graph, ops
        mlp_dir = os.path.join(d, "mlp_model")
        ops.save_scorer_dir(mlp_dir, "mlp", synth.make_mlp_weights(128), precision="split")
        pb = os.path.join(d, "frozen_graph.pb")
        frozen_graph.write_attention_graph(pb, synth.make_attn_weights(128, 64), seq_len=50)
        for model in (mlp_dir, pb):
            for lanes, c in ((1, 512), (1, 2048), (2, 2048)):
                r = serving.run_serve_host(d, d, 128, clients=c, seconds=3.0, max_batch=1024, max_wait_us=200, ef=128,
                                           topk=200, lanes=lanes, model_dir=model)
                print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
