"""Fuzzers of the three parsers of EXTERNAL bytes: the protobuf wire reader (nann_graphdef.h), the
protobuf text reader (nann_graphdef_text.h; also BlazeXlaOp's blaze_option_path) and the .npy decoder (nann_npy.h).

Run as a script in a process that has AddressSanitizer preloaded, against libnann_host_asan.so (the host sources under
-fsanitize=address,undefined -fno-sanitize-recover): tests/test_parser_fuzz.py does that with a small example budget in
the CPU suite; `FUZZ_EXAMPLES=100000 python tests/fuzz/run_fuzz.py` is the long run (profiles/r6_parser_fuzz_1e5.txt).
hypothesis strategies generate STRUCTURED inputs -- protobuf messages with the field numbers GraphDef uses, then damaged
(truncated varints, length fields past the end, deep nesting, DT_* with tensor_content of the wrong size); text-format token
soup and damaged TensorFlow-written pbtxt; npy headers of every version with lying lengths, Fortran order, overflowing
shapes, dtype mismatches -- and every example must be either decoded or rejected with a message.  A crash or a sanitizer
report ends the process (ASAN_OPTIONS=abort_on_error=1).  Properties checked on accepted inputs: a numpy-written file
decodes to its payload (checksum), a valid frozen graph still parses."""
import ctypes as C
import io
import os
import struct
import sys

import numpy as np
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
N = int(os.environ.get("FUZZ_EXAMPLES", "2000"))
LIB = C.CDLL(os.environ["NANN_FUZZ_LIB"])
COUNTS = {}
SET = settings(max_examples=N, deadline=None, database=None, derandomize=True,
               suppress_health_check=list(HealthCheck))


def count(name, accepted):
    c = COUNTS.setdefault(name, [0, 0])
    c[0] += 1
    c[1] += 1 if accepted else 0


# ---------------------------------------------------------------- protobuf wire format
def varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def key(field, wire):
    return varint((field << 3) | wire)


def ld(field, payload):
    return key(field, 2) + varint(len(payload)) + payload


DT = st.sampled_from([0, 1, 2, 3, 9, 19, 14, 7, 101, 119, 255, 1 << 40])
small_bytes = st.binary(max_size=24)
names = st.sampled_from([b"nonlinear_attention/query/0/Tensordot/MatMul", b"a", b"dnn/0_dnn/bn/batchnorm/mul_1", b"x/read", b"",
                         b"Const", b"MatMul", b"BiasAdd", b"Identity", b"Reshape", b"Transpose", b"value", b"T", b"dtype", b"shape",
                         b"^ctl", b"a:1", b"_7__cf__7"]) | small_bytes


@st.composite
def tensor_shape(draw):
    dims = draw(st.lists(st.integers(-2, 1 << 33) | st.sampled_from([0, 1, 2, 64, 128, 256]), max_size=5))
    body = b"".join(ld(2, key(1, 0) + varint(d) + draw(st.sampled_from([b"", ld(2, b"n")]))) for d in dims)
    if draw(st.booleans()) and draw(st.integers(0, 9)) == 0:
        body += key(3, 0) + varint(1)  # unknown_rank
    return body


@st.composite
def tensor_proto(draw):
    parts = [key(1, 0) + varint(draw(DT))]
    if draw(st.booleans()):
        parts.append(ld(2, draw(tensor_shape())))
    kind = draw(st.integers(0, 6))
    if kind == 0:
        parts.append(ld(4, draw(st.binary(max_size=64))))                       # tensor_content of any size
    elif kind == 1:
        parts.append(ld(5, b"".join(struct.pack("<f", f) for f in draw(st.lists(st.floats(width=32), max_size=8)))))  # packed float_val
    elif kind == 2:
        parts.append(b"".join(key(5, 5) + struct.pack("<f", f) for f in draw(st.lists(st.floats(width=32), max_size=4))))
    elif kind == 3:
        parts.append(ld(13, b"".join(varint(draw(st.integers(0, 70000))) for _ in range(draw(st.integers(0, 6))))))  # half_val
    elif kind == 4:
        parts.append(ld(7, b"".join(varint(draw(st.integers(-5, 1 << 35))) for _ in range(draw(st.integers(0, 6))))))  # int_val
    elif kind == 5:
        parts.append(ld(10, b"".join(varint(draw(st.integers(-5, 1 << 62))) for _ in range(draw(st.integers(0, 4))))))  # int64_val
    draw(st.randoms(use_true_random=False)).shuffle(parts)
    return b"".join(parts)


@st.composite
def attr_value(draw):
    kind = draw(st.integers(0, 7))
    if kind == 0:
        return ld(8, draw(tensor_proto()))
    if kind == 1:
        return key(6, 0) + varint(draw(DT))
    if kind == 2:
        return ld(7, draw(tensor_shape()))
    if kind == 3:
        return ld(2, draw(small_bytes))
    if kind == 4:
        return key(3, 0) + varint(draw(st.integers(-3, 1 << 40)))
    if kind == 5:
        return key(4, 5) + struct.pack("<f", draw(st.floats(width=32)))
    if kind == 6:
        return key(5, 0) + varint(draw(st.integers(0, 2)))
    return ld(1, b"".join(draw(st.lists(st.one_of(small_bytes.map(lambda b: ld(2, b)), st.integers(0, 300).map(lambda v: key(3, 0) + varint(v))),
                                        max_size=4))))  # list


@st.composite
def node_def(draw):
    parts = [ld(1, draw(names)), ld(2, draw(names))]
    parts += [ld(3, draw(names)) for _ in range(draw(st.integers(0, 3)))]
    if draw(st.booleans()):
        parts.append(ld(4, b"/device:CPU:0"))
    for _ in range(draw(st.integers(0, 3))):
        parts.append(ld(5, ld(1, draw(names)) + ld(2, draw(attr_value()))))  # map<string, AttrValue> entry
    if draw(st.integers(0, 5)) == 0:
        parts.append(key(draw(st.integers(6, 40)), 0) + varint(7))  # an unknown field
    return b"".join(parts)


@st.composite
def graph_def(draw):
    body = b"".join(ld(1, draw(node_def())) for _ in range(draw(st.integers(0, 5))))
    if draw(st.booleans()):
        body += ld(4, key(1, 0) + varint(draw(st.integers(0, 2000))))  # VersionDef
    return body


@st.composite
def damaged(draw, base):
    """truncations, byte flips, overlong varints, length fields past the end, deep nesting"""
    b = bytearray(draw(base))
    for _ in range(draw(st.integers(0, 3))):
        op = draw(st.integers(0, 5))
        if op == 0 and b:
            del b[draw(st.integers(0, len(b) - 1)):]
        elif op == 1 and b:
            i = draw(st.integers(0, len(b) - 1))
            b[i] ^= 1 << draw(st.integers(0, 7))
        elif op == 2:
            i = draw(st.integers(0, len(b)))
            b[i:i] = b"\xff" * draw(st.integers(1, 12))        # a varint that never ends / runs past 10 bytes
        elif op == 3:
            i = draw(st.integers(0, len(b)))
            b[i:i] = key(draw(st.integers(1, 8)), 2) + varint(draw(st.sampled_from([1 << 31, 1 << 40, (1 << 64) - 1, len(b) + 1])))
        elif op == 4:
            depth = draw(st.integers(1, 200))
            inner = bytes(b)
            for _ in range(depth):
                inner = ld(draw(st.sampled_from([1, 2, 5, 8])), inner)
                if len(inner) > 1 << 16:
                    break
            b = bytearray(inner)
        elif op == 5 and b:
            i = draw(st.integers(0, len(b) - 1))
            b[i:i] = draw(st.binary(max_size=8))
    return bytes(b)


def _call_graph(data, fmt):
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data + (b"\0" if not data else b""))
    return LIB.nann_fuzz_graphdef(buf, C.c_int64(len(data)), C.c_int32(fmt))


@seed(1)
@SET
@given(damaged(graph_def()) | st.binary(max_size=96), st.sampled_from([0, 1, 1, 1, 2]))
def fuzz_wire(data, fmt):
    if fmt == 2:
        data = ld(2, ld(2, data))  # SavedModel{meta_graphs{graph_def}}
    count("wire", _call_graph(data, fmt) == 0)


# ---------------------------------------------------------------- protobuf text format
TOK = st.sampled_from(["node", "{", "}", "<", ">", "name:", "op:", "input:", "device:", "attr", "key:", "value", "tensor", "dtype:",
                       "DT_FLOAT", "DT_HALF", "DT_INT32", "DT_BOGUS", "19", "tensor_shape", "dim", "size:", "tensor_content:", "float_val:",
                       "half_val:", "int_val:", "type:", "shape", "list", "s:", "i:", "f:", "b:", "true", "false", "[", "]", ",", ";", ":",
                       "1.5", "-3", "1e40", "inf", "nan", "0x1f", "9223372036854775808", "# c\n", "\n", '"a"', "'b'", '"\\000\\001"',
                       '"\\x4"', '"\\', '"unterminated', '"a" "b"', "versions", "producer:", "meta_graphs", "graph_def", "wait_ms:",
                       "run_mode:", "SKIP", "warmup_batchsize:", "config_proto", "xla_compilation:", "unknown_field:"])
text_soup = st.lists(TOK | st.text(alphabet="abc_.:{}<>[]\"'\\#\n 0123456789-", max_size=6), max_size=40).map(" ".join)
with open(os.path.join(ROOT, "tests", "golden", "tf_written", "half_plus_two_saved_model.pbtxt"), "rb") as f:
    PBTXT = f.read()


@st.composite
def damaged_pbtxt(draw):
    lo = draw(st.integers(0, len(PBTXT) - 1))
    b = bytearray(PBTXT[lo: lo + draw(st.integers(0, 1500))])
    for _ in range(draw(st.integers(0, 3))):
        if b:
            i = draw(st.integers(0, len(b) - 1))
            b[i:i + draw(st.integers(0, 3))] = draw(st.binary(max_size=3))
    return bytes(b)


TXT_DT = st.sampled_from(["DT_FLOAT", "DT_HALF", "DT_INT32", "DT_INT64", "DT_DOUBLE", "DT_BOOL", "DT_STRING", "DT_BFLOAT16", "DT_FLOAT_REF", "19", "DT_BOGUS"])
TXT_STR = st.sampled_from(['"x"', '"nonlinear_attention/query/0/Tensordot/MatMul"', "'a/b'", '"\\000\\001\\377\\x41"', '"a" "b"', '""', '"^c"'])
TXT_NUM = st.sampled_from(["0", "1", "2", "64", "-1", "1.5", "-2e3", "inf", "-inf", "nan", "1.0f", "0x10", "9223372036854775807", "1e400"])


@st.composite
def text_tensor(draw):
    dims = draw(st.lists(st.sampled_from(["0", "1", "2", "3", "64", "-1", "4294967296", "9223372036854775807"]), max_size=4))
    shape = "tensor_shape { " + " ".join("dim { size: %s }" % d for d in dims) + (" unknown_rank: true" if draw(st.integers(0, 9)) == 0 else "") + " }"
    kind = draw(st.integers(0, 4))
    if kind == 0:
        payload = 'tensor_content: "%s"' % "".join("\\%03o" % b for b in draw(st.binary(max_size=16)))
    elif kind == 1:
        payload = " ".join("float_val: " + draw(TXT_NUM) for _ in range(draw(st.integers(0, 4))))
    elif kind == 2:
        payload = "half_val: [" + ", ".join(str(draw(st.integers(0, 70000))) for _ in range(draw(st.integers(0, 4)))) + "]"
    elif kind == 3:
        payload = " ".join("int_val: " + str(draw(st.integers(-5, 1 << 40))) for _ in range(draw(st.integers(0, 4))))
    else:
        payload = " ".join("string_val: " + draw(TXT_STR) for _ in range(draw(st.integers(0, 2))))
    br = draw(st.sampled_from([("{", "}"), ("<", ">"), (": {", "}")]))
    return "tensor %s dtype: %s %s %s %s" % (br[0], draw(TXT_DT), shape if draw(st.booleans()) else "", payload, br[1])


@st.composite
def text_attr(draw):
    kind = draw(st.integers(0, 6))
    v = [draw(text_tensor()), "type: " + draw(TXT_DT), "shape { dim { size: %s } }" % draw(TXT_NUM), "s: " + draw(TXT_STR), "i: " + draw(TXT_NUM),
         "f: " + draw(TXT_NUM), "list { s: %s i: %s type: %s shape { } }" % (draw(TXT_STR), draw(TXT_NUM), draw(TXT_DT))][kind]
    return 'attr { key: %s value { %s } }' % (draw(TXT_STR), v)


@st.composite
def text_graph(draw):
    nodes = []
    for _ in range(draw(st.integers(0, 4))):
        body = ["name: " + draw(TXT_STR), "op: " + draw(st.sampled_from(['"Const"', '"MatMul"', '"Identity"', '"Placeholder"']))]
        body += ["input: " + draw(TXT_STR) for _ in range(draw(st.integers(0, 2)))]
        body += [draw(text_attr()) for _ in range(draw(st.integers(0, 3)))]
        sep = draw(st.sampled_from([" ", "\n  ", " , ", " ; ", " # c\n "]))
        nodes.append("node {%s%s }" % (sep, sep.join(body)))
    g = "\n".join(nodes) + ("\nversions { producer: 134 }" if draw(st.booleans()) else "")
    if draw(st.integers(0, 3)) == 0:
        g = "meta_graphs { graph_def { %s } }" % g
    b = bytearray(g.encode())
    for _ in range(draw(st.integers(0, 2))):  # light damage
        if b and draw(st.booleans()):
            i = draw(st.integers(0, len(b) - 1))
            b[i:i + draw(st.integers(0, 2))] = draw(st.sampled_from([b"", b"{", b"}", b'"', b"\\", b":", b"\x00", b"<", b"["]))
    return bytes(b)


@seed(2)
@SET
@given(text_graph() | text_soup.map(lambda s: s.encode("utf-8", "replace")) | damaged_pbtxt(), st.sampled_from([0, 3, 3]))
def fuzz_text(data, fmt):
    count("text", _call_graph(data, fmt) == 0)


@st.composite
def blaze_text(draw):
    fields = []
    for _ in range(draw(st.integers(0, 6))):
        fields.append(draw(st.sampled_from(["wait_ms: %s", "run_mode: %s", "xla_compilation: %s", "warmup_batchsize: [%s, 200, 400]", "warmup_batchsize: %s",
                                            "no_warmup_inputs: [\"a\", %s]", "config_proto { graph_options { x: %s } }", "config_proto: { a { b { c: %s } } }",
                                            "auto_mixed_precision: %s", "bogus_field: %s", "unit_flops: %s"]))
                      % draw(st.sampled_from(["5", "0", "-1", "true", "false", "SKIP", "DEFAULT", "BENCHMARK", "7", "2147483648", "ON", '"s"', "1.5", "{", ""])))
    return draw(st.sampled_from([" ", ",\n  ", "; ", " #c\n"])).join(fields).encode()


@seed(3)
@SET
@given(blaze_text() | text_soup.map(lambda s: s.encode("utf-8", "replace")))
def fuzz_blaze_options(data):
    buf = C.create_string_buffer(data, len(data) + 1)
    count("blaze_options", LIB.nann_fuzz_blaze_options(buf, C.c_int64(len(data))) == 0)


# ---------------------------------------------------------------- npy
NP_DT = {0: np.float16, 2: np.float32, 3: np.int32, 4: np.int64, 5: np.float64}
DESCR = st.sampled_from(["<f2", "<f4", "<f8", "<i4", "<i8", "|i1", "=f4", ">f4", "<u8", "<c16", "|S5", "[('a', '<f4')]", "", "<f", "<f44"])
DIM = st.integers(0, 40) | st.sampled_from([1 << 31, 1 << 62, (1 << 63) - 1, 1 << 63, 1 << 64, 10 ** 30])


@st.composite
def npy_image(draw):
    major = draw(st.sampled_from([1, 1, 1, 2, 2, 3, 0, 4, 255]))
    shape = draw(st.lists(DIM, max_size=draw(st.sampled_from([0, 1, 2, 3, 40]))))
    shape_txt = "(" + ", ".join(str(d) for d in shape) + ("," if len(shape) == 1 else "") + ")"
    if draw(st.integers(0, 9)) == 0:
        shape_txt = draw(st.sampled_from(["(", "3", "(3, -1)", "(3L, 4L)", "(3,, 4)", "()", "(a)", "( 3 , 4 )"]))
    fortran = draw(st.sampled_from(["False", "False", "False", "True", "false", "0", ""]))
    descr = draw(DESCR)
    q = draw(st.sampled_from(["'", "'", '"']))
    fields = ["'descr': %s%s%s" % (q, descr, q) if not descr.startswith("[") else "'descr': " + descr,
              "'fortran_order': " + fortran, "'shape': " + shape_txt]
    if draw(st.integers(0, 9)) == 0:
        fields.pop(draw(st.integers(0, 2)))
    draw(st.randoms(use_true_random=False)).shuffle(fields)
    hdr = ("{" + ", ".join(fields) + ", }").encode()
    hdr += b" " * draw(st.integers(0, 40)) + b"\n"
    declared = len(hdr)
    lie = draw(st.integers(0, 9))
    if lie == 0:
        declared = draw(st.sampled_from([0, 1, len(hdr) + 1, len(hdr) + 1000, 65535, (1 << 32) - 1, 1 << 21]))
    count_known = 1
    for d in shape:
        count_known = min(count_known * d, 1 << 20)
    payload = draw(st.binary(min_size=0, max_size=64)) if draw(st.booleans()) else bytes(min(count_known * 8, 4096))
    if major == 1:
        pre = b"\x93NUMPY" + bytes([major, 0]) + struct.pack("<H", declared & 0xffff)
    else:
        pre = b"\x93NUMPY" + bytes([major, 0]) + struct.pack("<I", declared & 0xffffffff)
    img = pre + hdr + payload
    if draw(st.integers(0, 7)) == 0:
        img = img[: draw(st.integers(0, len(img)))]
    if draw(st.integers(0, 15)) == 0:
        img = b"\x93NUMPX" + img[6:]
    return img


@st.composite
def npy_case(draw):
    """(image, expect_dtype, allow_cast, expect_shape): consistent about half of the time (so that the payload paths run), one lie otherwise"""
    if draw(st.booleans()):
        return draw(npy_image()), draw(st.integers(-1, 6)), draw(st.booleans()), draw(st.none() | st.lists(st.integers(0, 40), max_size=3))
    dtype = draw(st.sampled_from(sorted(NP_DT)))
    file_dt = draw(st.sampled_from(sorted(NP_DT))) if draw(st.integers(0, 3)) == 0 else dtype
    shape = draw(st.lists(st.integers(0, 6), max_size=3))
    a = np.zeros(shape, NP_DT[file_dt])
    a.reshape(-1)[:] = np.arange(a.size) % 251
    f = io.BytesIO()
    np.lib.format.write_array(f, a, version=draw(st.sampled_from([(1, 0), (2, 0), (3, 0)])))
    img = bytearray(f.getvalue())
    lie = draw(st.integers(0, 5))
    if lie == 0 and img:
        del img[draw(st.integers(0, len(img) - 1)):]
    elif lie == 1 and img:
        i = draw(st.integers(0, len(img) - 1))
        img[i] ^= 1 << draw(st.integers(0, 7))
    elif lie == 2:
        img += draw(st.binary(max_size=16))
    want_shape = shape if draw(st.booleans()) else draw(st.none() | st.lists(st.integers(0, 6), max_size=3))
    return bytes(img), dtype, draw(st.booleans()), want_shape


def _call_npy(img, dtype, allow_cast, shape=None):
    buf = (C.c_uint8 * max(1, len(img))).from_buffer_copy(img + (b"\0" if not img else b""))
    s = C.c_uint64(0)
    err = C.create_string_buffer(256)
    sh = (C.c_int64 * max(1, len(shape or ())))(*(shape or ()))
    rc = LIB.nann_fuzz_npy(buf, C.c_int64(len(img)), C.c_int32(dtype), C.c_int32(allow_cast), sh if shape is not None else None,
                           C.c_int32(len(shape or ())), C.byref(s), err, C.c_int32(256))
    return rc, s.value, err.value.decode("utf-8", "replace")


@seed(4)
@SET
@given(npy_case())
def fuzz_npy(case):
    img, dtype, allow_cast, shape = case
    rc, _, err = _call_npy(img, dtype, int(allow_cast), shape)
    assert rc in (0, 102, 104, 105, 106), rc
    assert rc == 0 or err, "a rejection without a message"
    count("npy", rc == 0)


@seed(5)
@settings(max_examples=max(50, N // 20), deadline=None, database=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.sampled_from(sorted(NP_DT)), st.lists(st.integers(0, 7), max_size=3), st.integers(0, 2 ** 32 - 1), st.sampled_from([(1, 0), (2, 0), (3, 0)]))
def fuzz_npy_roundtrip(dtype, shape, sd, version):
    """a file numpy wrote (formats 1.0 / 2.0 / 3.0) decodes to exactly its payload; its Fortran twin is refused"""
    a = (np.random.default_rng(sd).integers(-1000, 1000, size=shape)).astype(NP_DT[dtype])
    f = io.BytesIO()
    np.lib.format.write_array(f, a, version=version)
    rc, s, err = _call_npy(f.getvalue(), dtype, 0, list(shape))
    assert rc == 0, err
    exp = 1469598103934665603
    for b in a.tobytes():
        exp = ((exp ^ b) * 1099511628211) & ((1 << 64) - 1)
    for d in shape:
        exp = ((exp ^ d) * 1099511628211) & ((1 << 64) - 1)
    assert s == exp
    if a.ndim >= 2 and a.size:
        f = io.BytesIO()
        np.lib.format.write_array(f, np.asfortranarray(a), version=version)
        if b"'fortran_order': True" in f.getvalue():
            rc, _, err = _call_npy(f.getvalue(), dtype, 0, None)
            assert rc == 102 and err == "Fortran order NOT supported."
    count("npy_roundtrip", True)


@seed(6)
@settings(max_examples=max(20, N // 100), deadline=None, database=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(st.booleans(), st.integers(0, 1 << 30))
def fuzz_valid_graph(folded, sd):
    """the reference-shaped frozen graph (nann_amd/frozen_graph.py) parses and yields its weights; a truncation of it never crashes"""
    import tempfile
    from nann_amd import frozen_graph, synth
    w = synth.make_attn_weights(64, 64, seed=sd % 1000)
    with tempfile.NamedTemporaryFile(suffix=".pb") as t:
        frozen_graph.write_attention_graph(t.name, w, seq_len=50, folded=folded)
        data = open(t.name, "rb").read()
    assert _call_graph(data, 0) == 0 and _call_graph(data, 1) == 0
    rng = np.random.default_rng(sd)
    for _ in range(8):
        _call_graph(data[: int(rng.integers(0, len(data)))], 1)
    count("valid_graph", True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["wire", "text", "blaze_options", "npy", "npy_roundtrip", "valid_graph"]
    table = {"wire": fuzz_wire, "text": fuzz_text, "blaze_options": fuzz_blaze_options, "npy": fuzz_npy,
             "npy_roundtrip": fuzz_npy_roundtrip, "valid_graph": fuzz_valid_graph}
    import time
    for name in which:
        t0 = time.time()
        table[name]()
        c = COUNTS.get(name, [0, 0])
        print("FUZZ %-14s examples %7d accepted %7d rejected %7d  %.1f s" % (name, c[0], c[1], c[0] - c[1], time.time() - t0), flush=True)
    print("FUZZ DONE", flush=True)
