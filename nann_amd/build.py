"""Builds libnann_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnann_hip.so")
SOURCES = ["nann_hip.hip"]
DEPS = ["nann_hip.hip", "nann_device.h", os.path.join("..", "..", "include", "nann_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(SRC_DIR, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    """Compile every HIP source into nann_amd/_build/libnann_hip.so.  Returns its path."""
    if not force and not is_stale():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-fno-fast-math", "-ffp-contract=off",
           "-o", LIB] + [os.path.join(SRC_DIR, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
