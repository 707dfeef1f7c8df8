"""CPU: the three parsers of external bytes -- protobuf wire (nann_graphdef.h), protobuf text (nann_graphdef_text.h, also
BlazeXlaOp's blaze_option_path) and .npy (nann_npy.h) -- fuzzed with hypothesis under AddressSanitizer + UBSan
(tests/fuzz/fuzz_parsers.py in a child process with libasan preloaded, against libnann_host_asan.so), plus the example
cases the loaders' contracts name: npy formats 1.0 / 2.0 / 3.0, Fortran order refused (huge_const_op.cc:108-109), shape
overflow, truncation, dtype mismatch.  The suite runs a small budget per parser; the 10^5-example run of the same script is
profiles/r6_parser_fuzz_1e5.txt."""
import ctypes as C
import io
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))


def test_parsers_survive_fuzzing_under_asan_and_ubsan():
    import run_fuzz
    examples = int(os.environ.get("NANN_FUZZ_EXAMPLES", "800"))
    r = run_fuzz.run(examples=examples, timeout=1500)
    report = [l for l in r.stdout.splitlines() if l.startswith("FUZZ")]
    assert r.returncode == 0, (report, r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert report[-1] == "FUZZ DONE"
    seen = {l.split()[1]: (int(l.split()[3]), int(l.split()[5])) for l in report[:-1]}
    for name in ("wire", "text", "blaze_options", "npy"):
        n, accepted = seen[name]
        assert n == examples and 0 < accepted < n, (name, seen)  # both outcomes are exercised: decoded AND rejected inputs
    assert seen["npy_roundtrip"][0] == seen["npy_roundtrip"][1] > 0 and seen["valid_graph"][1] > 0


def _npy_info(path, dtype, allow_cast=0):
    from nann_amd import index_build
    L = C.CDLL(index_build.build_host_lib())
    shape = (C.c_int64 * 32)()
    rank, nbytes = C.c_int32(0), C.c_int64(0)
    err = C.create_string_buffer(256)
    rc = L.nann_host_npy_info(str(path).encode(), C.c_int32(dtype), C.c_int32(allow_cast), shape, C.byref(rank), C.byref(nbytes), err, C.c_int32(256))
    return rc, list(shape[: rank.value]), nbytes.value, err.value.decode()


F16, F32, I32, I64, F64 = 0, 2, 3, 4, 5


def test_huge_const_reads_npy_1_2_and_3_and_refuses_fortran_order(tmp_path):
    """npy.h:541-571 accepts formats 1.0 and 2.0 (3.0 = 2.0's layout, utf-8 header); huge_const_op.cc:108-109 refuses Fortran order."""
    a = np.arange(24, dtype=np.int64).reshape(4, 6)
    for version in ((1, 0), (2, 0), (3, 0)):
        p = tmp_path / ("v%d.npy" % version[0])
        with open(p, "wb") as f:
            np.lib.format.write_array(f, a, version=version)
        assert open(p, "rb").read()[6] == version[0]
        assert _npy_info(p, I64) == (0, [4, 6], 192, "")
    p = tmp_path / "fortran.npy"
    with open(p, "wb") as f:
        np.lib.format.write_array(f, np.asfortranarray(a), version=(1, 0))
    rc, _, _, err = _npy_info(p, I64)
    assert rc == 102 and err == "Fortran order NOT supported."
    rc, _, _, err = _npy_info(tmp_path / "v1.npy", I32)  # huge_const_op.cc:117-121: the op never casts
    assert rc == 105 and err == "DataType mismatch: <i8!=<i4"
    assert _npy_info(tmp_path / "v1.npy", I32, allow_cast=1) == (0, [4, 6], 96, "")  # the Python wrapper's astype (model_util.py:116-119)
    rc, _, _, err = _npy_info(tmp_path / "missing.npy", I64)
    assert rc == 104 and err.startswith("Fail to open file")


def _image(header, payload=b"", major=1, declared=None):
    hdr = header.encode() + b"\n"
    n = len(hdr) if declared is None else declared
    pre = b"\x93NUMPY" + bytes([major, 0]) + (struct.pack("<H", n) if major == 1 else struct.pack("<I", n))
    return pre + hdr + payload


@pytest.mark.parametrize("image,status,needle", [
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (4294967296, 4294967296), }"), 104, "overflows int64"),
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (1000000,), }", b"\0" * 64), 104, "truncated npy payload"),
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (2,), }", b"\0" * 16, declared=60000), 104, "truncated npy header"),
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (2,), }", b"\0" * 16, major=2, declared=1 << 30), 104, "longer than 1 MiB"),
    (_image("{'descr': '<i8', 'shape': (2,), }", b"\0" * 16), 104, "without fortran_order"),
    (_image("{'descr': '<i8', 'fortran_order': Maybe, 'shape': (2,), }", b"\0" * 16), 104, "neither True nor False"),
    (_image("{'descr': [('a', '<f4')], 'fortran_order': False, 'shape': (2,), }", b"\0" * 16), 102, "structured"),
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (2, }", b"\0" * 16), 104, "bad npy shape"),
    (_image("{'descr': '<i8', 'fortran_order': False, 'shape': (2,), }", b"\0" * 16, major=7), 104, "unsupported npy version"),
    (b"\x93NUMPX\x01\x00\x00\x00", 104, "not an npy file"),
    (b"", 104, "not an npy file"),
])
def test_malformed_npy_files_are_refused_with_a_message(tmp_path, image, status, needle):
    """every length of an external file is checked before it is used: the old reader threw std::out_of_range through the C ABI
    on a header without a colon and multiplied dims into a negative allocation size (found while writing the fuzzers)"""
    p = tmp_path / "bad.npy"
    p.write_bytes(image)
    rc, _, _, err = _npy_info(p, I64)
    assert rc == status and needle in err, (rc, err)


def test_wire_reader_refuses_overflowing_and_oversized_splat_tensors(tmp_path):
    """the wire fuzzer's first finding (round 6): TensorShapeProto dims whose product leaves int64 (UB), and a 20-byte message
    whose float_val list would be repeated into gigabytes"""
    sys.path.insert(0, ROOT)
    from nann_amd import frozen_graph as fg, index_build
    L = C.CDLL(index_build.build_host_lib())

    def parses(tensor_bytes):
        node = fg._ld(1, b"x") + fg._ld(2, b"Const") + fg._ld(5, fg._ld(1, b"value") + fg._ld(2, fg._ld(8, tensor_bytes)))
        data = fg._ld(1, node)
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        return L.nann_fuzz_graphdef(buf, C.c_int64(len(data)), C.c_int32(1)) == 0

    def tensor(dims, payload):
        shape = b"".join(fg._ld(2, fg._key(1, 0) + fg._varint(d)) for d in dims)
        return fg._key(1, 0) + fg._varint(1) + fg._ld(2, shape) + payload

    one = fg._key(5, 5) + struct.pack("<f", 1.0)
    assert parses(tensor([2, 3], one))                                     # a splat of 6 floats
    assert not parses(tensor([103721674595110, 1582972], one))             # the fuzzer's case: the product overflows
    assert not parses(tensor([1 << 30], one))                              # 4 GiB from 20 bytes
    assert not parses(tensor([2, 3], fg._ld(4, b"\0" * 20)))                # tensor_content of the wrong size
    assert parses(tensor([2, 3], fg._ld(4, b"\0" * 24)))
