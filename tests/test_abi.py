"""CPU: libnann_hip.so builds for gfx950, loads without a GPU, and exports every
symbol include/nann_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

from nann_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "nann_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nann_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_the_abi():
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/nann_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms


def test_abi_version_and_error_string():
    L = _lib.lib()
    assert L.nann_abi_version() == 6
    assert isinstance(_lib.last_error(), str)
    assert L.nann_device_count() >= 0


def test_status_codes_match_oracle():
    from oracle import oracle as O
    hdr = open(os.path.join(ROOT, "include", "nann_hip.h")).read()
    for name, val in [("INVALID_RAGGED_PARAMS", O.ERR_INVALID_RAGGED_PARAMS),
                      ("INVALID_RAGGED_INDICES", O.ERR_INVALID_RAGGED_INDICES),
                      ("INVALID_RAGGED_INPUT", O.ERR_INVALID_RAGGED_INPUT),
                      ("TOPK_K_GT_N", O.ERR_TOPK_K_GT_N), ("INDEX_OUT_OF_RANGE", O.ERR_INDEX_OUT_OF_RANGE),
                      ("EMPTY_SCORE_BATCH", O.ERR_EMPTY_SCORE_BATCH), ("BAD_ARGUMENT", O.ERR_BAD_ARGUMENT),
                      ("TOPK_SCALAR_INPUT", O.ERR_TOPK_SCALAR_INPUT)]:
        m = re.search(rf"NANN_ERR_{name}\s*=\s*(\d+)", hdr)
        assert m and int(m.group(1)) == val


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under nann_amd/ may import, include, link
    or load it (comments may cite its canonical summation orders)."""
    pkg = os.path.join(ROOT, "nann_amd")
    banned = [r"from\s+oracle", r"import\s+oracle", r"#\s*include\s*[\"<][^\">]*oracle", r"liboracle",
              r"oracle\.py", r"oracle/_build"]
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for pat in banned:
                    assert not re.search(pat, text), (f, pat)


def test_cpp_serving_host_builds_against_the_header_alone():
    """nann_serve.cpp includes nothing but include/nann_hip.h, the C++ standard library and three system headers (no HIP,
    no torch), links against the in-tree library, and fails loudly without a GPU."""
    import subprocess
    from nann_amd import serving
    includes = [line.split()[1] for line in open(serving._SERVE_SRC) if line.startswith("#include")]
    assert "\"nann_hip.h\"" in includes
    # besides the ABI header: the C++ standard library, and the system headers for thread pinning / F16C conversion
    system = {"<pthread.h>", "<sched.h>", "<immintrin.h>"}
    assert all(i == "\"nann_hip.h\"" or i in system or (i.startswith("<") and "/" not in i and "." not in i)
               for i in includes), includes
    assert not any(w in i for i in includes for w in ("hip/", "torch", "rocm")), includes
    exe = serving.build_serve_host()
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "/nonexistent", "/nonexistent", "64"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_deprecated_search_entry_points_only_forward():
    """ABI v6: ONE search operation = nann_search_opt / nann_search_model_opt.  The five spellings of rounds 1-5 stay exported
    for hosts built against v5, each as a single `return` of the canonical call; the hosts of this repo call the pair only."""
    src = open(os.path.join(ROOT, "nann_amd", "csrc", "nann_hip.hip")).read()
    hdr = open(os.path.join(ROOT, "include", "nann_hip.h")).read()
    for name, target in (("nann_search", "nann_search_opt"), ("nann_search_v", "nann_search_opt"), ("nann_search_ex", "nann_search_opt"),
                         ("nann_search_model", "nann_search_model_opt"), ("nann_search_model_v", "nann_search_model_opt")):
        m = re.search(r"\nint %s\(([^)]*)\)\s*\{(.*?)\n\}\n" % name, src, flags=re.S)
        assert m, name
        body = re.sub(r"//[^\n]*", "", m.group(2)).strip()
        assert re.fullmatch(r"return %s\((.|\n)*\);" % target, body), (name, body)
        assert body.count(";") == 1, (name, body)
        assert re.search(r"/\* deprecated: thin wrapper of %s \*/\s*int %s\(" % (target, name), hdr), name
    deprecated = r"\bnann_search(_v|_ex)?\(|\bnann_search_model(_v)?\("
    for rel in ("nann_amd/csrc/host/nann_serve.cpp", "nann_amd/tf_ops/nann_tf_ops.cc", "nann_amd/retrieval.py", "nann_amd/serving.py",
                "nann_amd/shard.py", "nann_amd/evaluate.py"):
        text = re.sub(r"//[^\n]*|#[^\n]*", "", open(os.path.join(ROOT, rel)).read())
        text = re.sub(r'""".*?"""', "", text, flags=re.S)
        assert not re.search(deprecated, text), rel
