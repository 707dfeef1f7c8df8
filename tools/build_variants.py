#!/usr/bin/env python3
"""Build kernel-experiment variants of libnann_hip.so side by side (nann_amd/_build/var_<name>/),
for A/B runs on the GPU box in ONE call: tools/gpu_round.sh benches every var_* it finds
against the same cached index (NANN_HIP_LIB selects the library).
usage: tools/build_variants.py [name=flag,flag ...]   (no arguments: the default set)"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import build  # noqa: E402

DEFAULT = {
    "su12": ["-DNANN_SCORE_U=12"],   # scoring phase: 12 row loads per lane in flight
    "su16": ["-DNANN_SCORE_U=16"],   # ... 16 (256 KB per CU)
    "roll": ["-DNANN_SCORE_ROLL=1"],   # scoring phase: rolling window of row loads
    "roll_su12": ["-DNANN_SCORE_ROLL=1", "-DNANN_SCORE_U=12"],
    "roll_su16": ["-DNANN_SCORE_ROLL=1", "-DNANN_SCORE_U=16"],   # 32 VGPRs spilled at 1024 threads: expect no gain
    "tkms": ["-DNANN_TOPK_MINSUB=1"],  # top-k: radix search on key - min(key)
    "fp": ["-DNANN_FILTER_PACKED=1"],  # LDS filter without the pre-read (3 barriers per piece)
    "su16_tkms_fp": ["-DNANN_SCORE_U=16", "-DNANN_TOPK_MINSUB=1", "-DNANN_FILTER_PACKED=1"],
    # L2 only: exact hash set of visited ids (64 KB) + small phase buffers -> two 512-thread workgroups per CU
    "compact": ["-DNANN_COMPACT=1"],
    "compact_tkms": ["-DNANN_COMPACT=1", "-DNANN_TOPK_MINSUB=1"],
}


def main():
    todo = dict(DEFAULT)
    if len(sys.argv) > 1:
        todo = {}
        for a in sys.argv[1:]:
            name, _, flags = a.partition("=")
            todo[name] = [f for f in flags.split(",") if f]
    with ThreadPoolExecutor(max(1, min(3, len(todo)))) as ex:
        for path in ex.map(lambda kv: build.build(variant=kv[0], extra_flags=kv[1]), todo.items()):
            print(path)


if __name__ == "__main__":
    main()
