#!/usr/bin/env python3
"""Timing / tuning builds of the resident-layer-2 MLP traversal (nann_amd/csrc/nann_mlp5.h): recompiles ONLY
the translation unit nann_mlp_res_inst.hip lives in with extra -D flags and links it with the objects of the shipped build into
nann_amd/_build/var_<name>/libnann_hip.so (load with NANN_HIP_LIB=...).  ~20 s per variant instead of a full rebuild.
usage: tools/build_res_variant.py <name> [--unit nann_l2_inst.hip] [-DNANN_RES_PF=2] [-DNANN_RES_VAR=1] ...
(--unit: the FIRST translation unit that holds that source instead -- e.g. the f16 L2 traversal for the phase-repeat builds
of nann_search.h, NANN_REPEAT_SCORE / NANN_REPEAT_TOPK)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nann_amd import build as B  # noqa: E402


def main():
    name, flags = sys.argv[1], sys.argv[2:]
    source = "nann_mlp_res_inst.hip"
    if flags and flags[0] == "--unit":
        source, flags = flags[1], flags[2:]
    if not os.environ.get("NANN_VARIANT_NO_BASE_BUILD"):
        B.build()  # the shipped objects must exist (NANN_VARIANT_NO_BASE_BUILD=1: link against them as they lie)
    out = os.path.join(B.OUT_DIR, "var_" + name)
    os.makedirs(out, exist_ok=True)
    unit = next(u for u in B.UNITS if any(src == source for src, _ in u[1]))  # the object the kernels live in
    obj = os.path.join(out, unit[0])
    log = os.path.join(out, "compile.log")
    cmd = B.unit_command(unit[0], unit[1], out, flags, save_temps=False)
    with open(log, "w") as f:
        subprocess.check_call(cmd, stderr=f, stdout=f)
    objs = [obj if o == unit[0] else os.path.join(B.OUT_DIR, o[:-2] + ".d", o) for o, _ in B.UNITS]
    lib = os.path.join(out, "libnann_hip.so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
    import re
    txt = open(log, errors="replace").read()
    for b in txt.split("Function Name: ")[1:]:
        g = lambda k: re.search(k + r": (\d+)", b)  # noqa: E731
        print(b.split()[0][:64], "vgpr", g("VGPRs").group(1), "scratch", g(r"ScratchSize \[bytes/lane\]").group(1))
    print(lib)


if __name__ == "__main__":
    main()
