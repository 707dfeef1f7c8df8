"""Builds nann_amd/tf_ops/nann_tf_ops.cc (OUR op shim) against tests/tf_mock -- a functional model of the TensorFlow
op-kernel API, TensorFlow itself is not in the image -- plus the executor-side driver into
tests/tf_mock/_build/libnann_tf_ops_mock.so, linked against the in-tree libnann_hip.so.  g++ only; the .so travels to
the GPU box with the snapshot (and is rebuilt there if its sources changed)."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SHIM = os.path.join(ROOT, "nann_amd", "tf_ops", "nann_tf_ops.cc")
DRIVER = os.path.join(HERE, "tfm_driver.cc")
MOCK = os.path.join(HERE, "tensorflow", "core", "framework", "op_kernel.h")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnann_tf_ops_mock.so")
FLAGS = ["-std=c++14", "-O2", "-g", "-fPIC", "-shared", "-pthread", "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _hash():
    h = hashlib.sha256()
    for p in (SHIM, DRIVER, MOCK, os.path.join(ROOT, "include", "nann_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False):
    from nann_amd import build as nbuild
    hip_lib = nbuild.build()
    stamp = LIB + ".srchash"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == _hash():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["g++"] + FLAGS + ["-I", HERE, "-I", os.path.join(ROOT, "include"), SHIM, DRIVER, "-o", LIB,
                             "-L", os.path.dirname(hip_lib), "-lnann_hip", "-Wl,-rpath," + os.path.dirname(hip_lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("tf_mock build failed:\n" + r.stderr[-6000:])
    if r.stderr.strip():
        raise RuntimeError("tf_mock build has warnings (the shim is kept -Wall -Wextra clean):\n" + r.stderr[-6000:])
    with open(stamp, "w") as f:
        f.write(_hash())
    return LIB


if __name__ == "__main__":
    print(build(force=True))
