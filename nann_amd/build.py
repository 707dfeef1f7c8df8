"""Builds libnann_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnann_hip.so")
# translation units: (object name, [(source, {macro: value} defined around its #include), ...]) -- compiled in parallel, then
# linked.  A unit of several sources is compiled through a generated wrapper (_build/<obj>.d/unit.hip) that includes them
# one after the other.  Round 5: 13 units -> 9: the three small host-facing files share one object, the
# resident-layer-2 MLP kernels ride with the d = 64 MLP instances.  Round 6: 12 units -- the evaluation traversal's LDS forms are
# units of their own, and the bf16 and f32 L2 instances are two again (their shared unit had become the longest: it alone bounded
# a build from scratch at ~6 min on 8 cores).
UNITS = [("nann_core.o", [("nann_hip.hip", {}), ("nann_comm.hip", {}), ("nann_hnsw_build.hip", {})]),
         ("nann_l2_f16.o", [("nann_l2_inst.hip", {"NANN_L2_DT": "0", "NANN_L2_NAME": "f16"})]),
         ("nann_l2_bf16.o", [("nann_l2_inst.hip", {"NANN_L2_DT": "1", "NANN_L2_NAME": "bf16"})]),
         ("nann_l2_f32.o", [("nann_l2_inst.hip", {"NANN_L2_DT": "2", "NANN_L2_NAME": "f32"})]),
         ("nann_mlp_d64_res.o", [("nann_mlp_inst.hip", {"NANN_MLP_D": "64"}), ("nann_mlp_res_inst.hip", {})]),
         ("nann_mlp_d128.o", [("nann_mlp_inst.hip", {"NANN_MLP_D": "128"})]),
         ("nann_mlp_d256.o", [("nann_mlp_inst.hip", {"NANN_MLP_D": "256"})]),
         ("nann_attn.o", [("nann_attn_inst.hip", {})]),
         ("nann_attn_split.o", [("nann_attn_split_inst.hip", {})]),
         ("nann_eval.o", [("nann_eval_inst.hip", {})]),
         ("nann_eval_lds.o", [("nann_eval_lds_inst.hip", {})]),
         ("nann_eval_win.o", [("nann_eval_win_inst.hip", {})])]
DEPS = ["nann_hip.hip", "nann_mlp_inst.hip", "nann_mlp_res_inst.hip", "nann_mlp5.h", "nann_mlp6.h", "nann_l2_inst.hip", "nann_attn_inst.hip", "nann_attn_split_inst.hip", "nann_attn_split.h", "nann_attn_proj.h", "nann_eval_inst.hip", "nann_eval_lds_inst.hip", "nann_eval_win_inst.hip", "nann_eval.h", "nann_comm.hip", "nann_hnsw_build.hip", "nann_device.h", "nann_mlp.h", "nann_mlp2.h", "nann_mlp3.h",
        "nann_attn.h", "nann_attn_kernels.h", "nann_search.h", os.path.join("host", "nann_graphdef.h"), os.path.join("host", "nann_graphdef_text.h"), os.path.join("host", "nann_blaze_options.h"), os.path.join("host", "nann_npy.h"), os.path.join("host", "nann_projcache.h"),
        os.path.join("..", "..", "include", "nann_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fno-fast-math", "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash():
    """Content hash of every source the library is built from (mtimes do not survive the copy to
    the GPU box; contents do)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        with open(os.path.join(SRC_DIR, d), "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_stale():
    """True when there is no library or it was built from other sources (hash recorded next to it)."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(LIB + ".srchash") as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def build(force=False, verbose=False, variant=None, extra_flags=()):
    """Compile every HIP source for gfx950 into nann_amd/_build/libnann_hip.so (objects in
    parallel, then one link).  Returns the library path.  `variant`: build into
    nann_amd/_build/var_<variant>/ with `extra_flags` instead (kernel experiments; load it
    with NANN_HIP_LIB)."""
    if variant is None:
        if not force and not is_stale():
            return LIB
        # one builder at a time: the ranks of a multi-process launch that all find the library stale (a snapshot whose
        # sources are newer than its .so) must not compile into the same directory together -- the first one builds, the
        # others wait on the lock and find a fresh library
        import fcntl
        os.makedirs(OUT_DIR, exist_ok=True)
        with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not force and not is_stale():
                    return LIB
                return _build_into(OUT_DIR, LIB, (), verbose)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    out = os.path.join(OUT_DIR, "var_" + variant)
    return _build_into(out, os.path.join(out, "libnann_hip.so"), tuple(extra_flags), verbose)


def _check_occupancy(log_path):
    """The 16K-slot hash-set traversal with the L2 scorer (k_search<.., VIS=2, SC=0, 512>) only pays off with TWO workgroups per
    CU = 4 waves per SIMD; one VGPR over 128 halves its occupancy without any other symptom (seen: 130
    VGPRs -> 2.82 ms instead of 1.99 ms).  Refuse to link such an object."""
    import re
    name = None
    for line in open(log_path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"Occupancy \[waves/SIMD\]: (\d+)", line)
        if m and name and re.search(r"k_searchILi\d+ELi\d+ELi2ELi0ELi512E", name) and int(m.group(1)) < 4:
            raise RuntimeError(f"{name}: occupancy {m.group(1)} waves/SIMD, the hash-set kernel needs 4")


def unit_command(obj, parts, odir, extra_flags=(), save_temps=True):
    """Write the wrapper of a translation unit into `odir` and return the hipcc command that compiles it to odir/obj."""
    unit = os.path.join(odir, "unit.hip")
    with open(unit, "w") as f:
        for src, macros in parts:
            for k, v in macros.items():
                f.write("#define %s %s\n" % (k, v))
            f.write('#include "%s"\n' % os.path.join(SRC_DIR, src))
            for k in macros:
                f.write("#undef %s\n" % k)
    return [_hipcc()] + FLAGS + list(extra_flags) + ["-I", SRC_DIR] + (["--save-temps=obj"] if save_temps else []) + \
           ["-Rpass-analysis=kernel-resource-usage", "-c", unit, "-o", os.path.join(odir, obj)]


def _build_into(OUT_DIR, LIB, extra_flags, verbose, only=None):
    """only: names of the objects to recompile (kernel iteration: `python -m nann_amd.build --only nann_eval.o`); the others are
    linked as they lie -- the caller vouches that their sources did not change."""
    os.makedirs(OUT_DIR, exist_ok=True)
    procs = []
    for obj, parts in UNITS:
        if only is not None and obj not in only:
            continue
        # one directory per object: --save-temps keeps the device assembly for the audit below
        odir = os.path.join(OUT_DIR, obj[:-2] + ".d")
        os.makedirs(odir, exist_ok=True)
        cmd = unit_command(obj, parts, odir, extra_flags)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        log = open(os.path.join(odir, "compile.log"), "w")
        procs.append((cmd, odir, subprocess.Popen(cmd, stderr=log, stdout=log)))
    for cmd, odir, p in procs:
        if p.wait() != 0:
            sys.stderr.write(open(os.path.join(odir, "compile.log")).read()[-6000:])
            raise subprocess.CalledProcessError(p.returncode, cmd)
        _check_occupancy(os.path.join(odir, "compile.log"))
    # hipcc (ROCm 7.2) can place VGPR spill code ahead of the exec restore of a join block; the
    # lanes that were masked off then reload garbage (seen: top-k positions all -1).  Refuse
    # to ship an object with that pattern.
    from . import isa_audit
    for cmd, odir, _ in procs:
        for f in os.listdir(odir):
            if f.endswith(".s") and "amdgcn" in f:
                hits = isa_audit.flow_hits(os.path.join(odir, f))
                if hits:
                    raise RuntimeError("miscompiled spill placement in %s: %s" % (f, hits[:3]))
            if not f.endswith(".o") and f != "compile.log" and not os.environ.get("NANN_KEEP_TEMPS"):
                os.remove(os.path.join(odir, f))  # preprocessed sources, bitcode, assembly: ~15 MB per object
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + \
           [os.path.join(OUT_DIR, obj[:-2] + ".d", obj) for obj, _ in UNITS] + ["-ldl"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    with open(LIB + ".srchash", "w") as f:
        f.write(source_hash())
    return LIB


if __name__ == "__main__":
    if "--only" in sys.argv:
        names = sys.argv[sys.argv.index("--only") + 1].split(",")
        assert all(any(n == o for o, _ in UNITS) for n in names), names
        print(_build_into(OUT_DIR, LIB, (), True, only=names))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
