#!/usr/bin/env python3
"""Build kernel-experiment variants of libnann_hip.so side by side (nann_amd/_build/var_<name>/), for
A/B runs on the GPU box in ONE call (tools/gpu_round.sh benches every var_* it finds against the
same cached index; NANN_HIP_LIB selects the library).  Only the units the L2 traversal needs are
rebuilt per variant; results go to profiles/*_variants.jsonl and winners become the default.
usage: tools/build_variants.py name=flag,flag [...]"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import build  # noqa: E402


def main():
    todo = {}
    for a in sys.argv[1:]:
        name, _, flags = a.partition("=")
        todo[name] = [f for f in flags.split(",") if f]
    with ThreadPoolExecutor(max(1, min(2, len(todo)))) as ex:
        for path in ex.map(lambda kv: build.build(variant=kv[0], extra_flags=kv[1]), todo.items()):
            print(path)


if __name__ == "__main__":
    main()
