#!/usr/bin/env python3
"""sha256 of the device assembly of every object of the DEFAULT build (kernels only: comment and
directive lines that can carry paths are dropped).  Used to show that an edit guarded by a
build-time knob leaves the shipped kernels byte-for-byte unchanged:
    tools/isa_fingerprint.py > before.txt; <edit>; tools/isa_fingerprint.py | diff before.txt -"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nann_amd import build  # noqa: E402


def one(unit):
    obj, parts = unit
    d = tempfile.mkdtemp(prefix="isa_")
    subprocess.check_call(build.unit_command(obj, parts, d), stderr=subprocess.DEVNULL)
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith(".s") and "amdgcn" in f:
            for line in open(os.path.join(d, f), errors="replace"):
                s = re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", line.split(";")[0].rstrip())  # per-compile id
                if not s or re.match(r"\s*\.(file|ident|section|loc|amdgpu_metadata|end_amdgpu_metadata)\b", s):
                    continue
                h.update(s.encode() + b"\n")
    subprocess.call(["rm", "-rf", d])
    return obj, h.hexdigest()


if __name__ == "__main__":
    with ThreadPoolExecutor(4) as ex:
        for obj, digest in ex.map(one, build.UNITS):
            print(obj, digest)
