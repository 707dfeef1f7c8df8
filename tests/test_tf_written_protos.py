"""The GraphDef readers behind BlazeXlaOp.graph_def (csrc/host/nann_graphdef.h: protobuf wire format;
nann_graphdef_text.h: protobuf text format, tried first as blaze_xla_kernel.cc:169-175 does) pinned to bytes
TENSORFLOW wrote.  Until round 4 the wire reader had only ever parsed files of this repo's own writer
(nann_amd/frozen_graph.py).  The fork holds TF-written protobufs; copied under tests/golden/tf_written/ as DATA fixtures:

  multi_add.pb                      tensorflow/lite/testdata/multi_add.pb -- a 424-byte binary GraphDef.  What the fork
                                    states about it (tensorflow/lite/testing/tf_driver_test.cc:88-118): inputs a, b, c, d
                                    (float), outputs x, y, and for given inputs the output values -- reproduced here by
                                    evaluating the DECODED graph.
  half_plus_two_saved_model.pb      tensorflow/cc/saved_model/testdata/half_plus_two/00000123/saved_model.pb -- a binary
                                    SavedModel whose graph has Const nodes with TensorProto values, string tensors,
                                    shape / type / list attrs, control inputs.
  half_plus_two_saved_model.pbtxt   .../half_plus_two_pbtxt/00000123/saved_model.pbtxt -- THE SAME GRAPH in text form,
                                    written by TensorFlow: the expected value of every field of the binary decode, and
                                    at the same time a TF-written input for the text reader.

Both decoders run through libnann_host.so (the very parser nann_model_load uses) and dump a canonical JSON."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from nann_amd import index_build

HERE = os.path.dirname(os.path.abspath(__file__))
TFW = os.path.join(HERE, "golden", "tf_written")
ANY, BINARY_GRAPH, BINARY_SAVED_MODEL, TEXT = 0, 1, 2, 3


def dump(path, fmt):
    lib = C.CDLL(index_build.build_host_lib())
    need = C.c_int64(0)
    err = C.create_string_buffer(512)
    rc = lib.nann_graphdef_dump(str(path).encode(), C.c_int32(fmt), None, C.c_int64(0), C.byref(need), err, 512)
    if rc:
        raise ValueError(err.value.decode())
    buf = C.create_string_buffer(need.value)
    assert lib.nann_graphdef_dump(str(path).encode(), C.c_int32(fmt), buf, C.c_int64(need.value), C.byref(need), err, 512) == 0
    return json.loads(buf.value.decode())


def by_name(g):
    return {n["name"]: n for n in g["nodes"]}


def test_multi_add_pb_as_tf_driver_test_states_it():
    """tf_driver_test.cc:88-118: TfDriver({"a","b","c","d"}, float x4, ..., {"x","y"}).LoadModel(multi_add.pb); with
    a = .1,.2,.3,.4  b = .001,...  c reset to zeros  d = .01,...  the outputs read
    x = 0.101000004,0.202000007,0.303000003,0.404000014 and y = 0.0109999999,0.0219999999,0.0329999998,0.0439999998."""
    for fmt in (ANY, BINARY_GRAPH):  # ANY: text is tried first and must give way to the binary reader
        g = dump(os.path.join(TFW, "multi_add.pb"), fmt)
        nodes = by_name(g)
        for name in "abcd":
            assert nodes[name]["op"] == "Placeholder" and nodes[name]["attrs"]["dtype"] == {"kind": "t", "v": 1}  # DT_FLOAT
        assert "x" in nodes and "y" in nodes
        feeds = {"a": np.float32([0.1, 0.2, 0.3, 0.4]), "b": np.float32([0.001, 0.002, 0.003, 0.004]),
                 "c": np.zeros(4, np.float32), "d": np.float32([0.01, 0.02, 0.03, 0.04])}

        def ev(name):
            n = nodes[name.split(":")[0]]
            if n["op"] == "Placeholder":
                return feeds[n["name"]]
            assert n["op"] in ("Add", "AddV2", "Identity"), n["op"]
            vals = [ev(i) for i in n["inputs"] if not i.startswith("^")]
            return vals[0] if n["op"] == "Identity" else np.float32(vals[0] + vals[1])

        fmt9 = lambda v: ",".join("%.9g" % float(t) for t in v)  # noqa: E731  (TfDriver::ReadOutput prints %.9g)
        assert fmt9(ev("x")) == "0.101000004,0.202000007,0.303000003,0.404000014"
        assert fmt9(ev("y")) == "0.0109999999,0.0219999999,0.0329999998,0.0439999998"
        # x = a + b + c and y = b + c + d structurally: each output depends on exactly these placeholders
        def leaves(name):
            n = nodes[name.split(":")[0]]
            return {n["name"]} if n["op"] == "Placeholder" else set().union(*[leaves(i) for i in n["inputs"]])
        assert leaves("x") == {"a", "b", "c"} and leaves("y") == {"b", "c", "d"}
    with pytest.raises(ValueError):
        dump(os.path.join(TFW, "multi_add.pb"), TEXT)  # a binary file is not text


def test_half_plus_two_binary_decode_equals_tensorflows_text_form():
    """Every node / input / attr / TensorProto field the wire reader decodes from saved_model.pb equals what TensorFlow
    itself printed for the same graph in saved_model.pbtxt (decoded by the text reader): 65 nodes."""
    gb = dump(os.path.join(TFW, "half_plus_two_saved_model.pb"), BINARY_SAVED_MODEL)
    gt = dump(os.path.join(TFW, "half_plus_two_saved_model.pbtxt"), TEXT)
    assert len(gb["nodes"]) == len(gt["nodes"]) == 65
    differing = []
    for nb, nt in zip(gb["nodes"], gt["nodes"]):
        if nb != nt:
            differing.append(nb["name"])
            # the ONE field in which the two exports genuinely differ: the saver's temporary file name carries a uuid
            # drawn per export run (saver.py: "_temp_<uuid>/part"); everything else of the node must still agree
            sb, st = nb["attrs"]["value"]["v"]["s"], nt["attrs"]["value"]["v"]["s"]
            assert len(sb) == len(st) == 1 and sb[0].startswith("_temp_") and st[0].startswith("_temp_") and sb[0].endswith("/part")
            nb["attrs"]["value"]["v"]["s"] = nt["attrs"]["value"]["v"]["s"] = ["_temp_/part"]
        assert nb == nt, (nb["name"], nb, nt)
    assert differing == ["save/StringJoin/inputs_1"]
    # and a few literals read straight off the pbtxt, so that the comparison cannot pass on two equally wrong decoders
    txt = open(os.path.join(TFW, "half_plus_two_saved_model.pbtxt")).read()
    n = by_name(gb)
    assert n["a"]["op"] == "VariableV2" and n["a"]["attrs"]["dtype"]["v"] == 1 and n["a"]["attrs"]["shape"]["v"]["dims"] == []
    assert "float_val: 0.5" in txt and n["a/initial_value"]["attrs"]["value"]["v"]["f"] == [0.5]
    assert "float_val: 2.0" in txt and n["b/initial_value"]["attrs"]["value"]["v"]["f"] == [2.0]
    assert n["a/initial_value"]["attrs"]["value"]["v"]["dtype"] == 1 and n["a/initial_value"]["attrs"]["value"]["v"]["shape"] == []
    assert n["a/Assign"]["inputs"] == ["a", "a/initial_value"]
    assert n["a/Assign"]["attrs"]["_class"] == {"kind": "l", "v": {"s": ["loc:@a"], "i": [], "f": [], "b": [], "type": [], "shape": [],
                                                                    "n_tensors": 0, "n_funcs": 0}}
    assert n["a/Assign"]["attrs"]["use_locking"] == {"kind": "b", "v": True}
    strings = [v["attrs"]["value"]["v"]["s"] for v in gb["nodes"] if v["op"] == "Const" and v["attrs"]["value"]["v"]["dtype"] == 7]
    assert ["x"] in strings and ["x2"] in strings  # string_val: "x" / "x2" in the pbtxt
    assert any(i.startswith("^") for v in gb["nodes"] for i in v["inputs"])  # control inputs survive


def test_text_format_frozen_graph_loads_like_the_binary_one(tmp_path):
    """blaze_xla_kernel.cc:169-175 tries ReadTextProto first: the reference's scorer graph written in TEXT format
    (tf.io.write_graph(..., as_text=True)) must give the weight extraction exactly what the binary file gives."""
    from nann_amd import frozen_graph, synth
    from test_frozen_graph import read_attention
    w = synth.make_attn_weights(64, 64)
    pb = tmp_path / "frozen_graph.pb"
    frozen_graph.write_attention_graph(str(pb), w, folded=True)
    g = dump(pb, BINARY_GRAPH)
    names = {1: "DT_FLOAT", 3: "DT_INT32", 9: "DT_INT64", 7: "DT_STRING", 10: "DT_BOOL", 19: "DT_HALF"}

    def esc(b):
        return "".join(chr(c) if 32 <= c < 127 and chr(c) not in '"\\' else "\\%03o" % c for c in b)

    out = []
    for n in g["nodes"]:
        out.append("node {\n  name: \"%s\"\n  op: \"%s\"" % (n["name"], n["op"]))
        out += ["  input: \"%s\"" % i for i in n["inputs"]]
        for k, a in n["attrs"].items():
            v = a["v"]
            if a["kind"] == "t":
                body = "type: %s" % names[v]
            elif a["kind"] == "T":
                dims = "".join(" dim { size: %d }" % d for d in v["shape"])
                if v["f"]:  # tensor_content as TensorFlow prints it: escaped little-endian bytes
                    payload = 'tensor_content: "%s"' % esc(np.asarray(v["f"], np.float32).tobytes())
                else:
                    payload = " ".join("int_val: %d" % x for x in v["i"])
                body = "tensor { dtype: %s tensor_shape {%s } %s }" % (names[v["dtype"]], dims, payload)
            elif a["kind"] == "b":
                body = "b: %s" % ("true" if v else "false")
            elif a["kind"] == "i":
                body = "i: %d" % v
            elif a["kind"] == "f":
                body = "f: %r" % v
            elif a["kind"] == "s":
                body = 's: "%s"' % esc(v.encode("latin1"))
            elif a["kind"] == "h":
                body = "shape {%s }" % "".join(" dim { size: %d }" % d for d in v["dims"])
            else:
                continue
            out.append("  attr {\n    key: \"%s\"\n    value { %s }\n  }" % (k, body))
        out.append("}")
    out.append("versions { producer: 134 }  # trailing fields the reader skips")
    ptxt = tmp_path / "frozen_graph.pbtxt"
    ptxt.write_text("\n".join(out) + "\n")
    assert dump(ptxt, TEXT) == g and dump(ptxt, ANY) == g
    db, eb, wb = read_attention(pb)
    dt, et, wt = read_attention(ptxt)  # the hook parses as BlazeXlaOp does: text first, then binary
    assert (db, eb) == (dt, et)
    for a, b in zip(wb, wt):
        assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_text_reader_rejects_malformed_input(tmp_path):
    for bad in ("node { name: \"a\" op: \"Const\"", "node { name \"a\" }", "node { name: \"a\\q\" }", "}"):
        p = tmp_path / "bad.pbtxt"
        p.write_text(bad)
        with pytest.raises(ValueError):
            dump(p, TEXT)


def test_tensor_content_shorter_than_its_shape_is_rejected(tmp_path):
    """a DT_BOOL Const whose tensor_content is shorter than its shape read past the buffer (every other
    dtype checked content.size() == n * element size).  Both readers must refuse it -- text and wire -- and the same
    for a float tensor, and a bool tensor of the right length must still decode."""
    def pbtxt(dtype, dim, content):
        return ('node { name: "c" op: "Const" attr { key: "dtype" value { type: %s } } attr { key: "value" value { tensor { '
                'dtype: %s tensor_shape { dim { size: %d } } tensor_content: "%s" } } } }' % (dtype, dtype, dim, content))

    def wire(dtype_code, dim, content):
        def vi(v):
            out = b""
            while True:
                b7 = v & 0x7f
                v >>= 7
                out += bytes([b7 | (0x80 if v else 0)])
                if not v:
                    return out
        def ld(field, payload):
            return vi(field << 3 | 2) + vi(len(payload)) + payload
        shape = ld(2, vi(1 << 3) + vi(dim))                              # TensorShapeProto.dim { size }
        tensor = vi(1 << 3) + vi(dtype_code) + ld(2, shape) + ld(4, content)  # dtype, tensor_shape, tensor_content
        attr_v = ld(8, tensor)                                           # AttrValue.tensor
        entry = ld(1, b"value") + ld(2, attr_v)
        node = ld(1, b"c") + ld(2, b"Const") + ld(5, entry)
        return ld(1, node)

    for dtype, code, dim, content, ok in (("DT_BOOL", 10, 200000000, b"\x01", False), ("DT_BOOL", 10, 3, b"\x01\x00\x01", True),
                                          ("DT_BOOL", 10, 2, b"\x01\x00\x01", False), ("DT_FLOAT", 1, 4, b"\x00" * 8, False)):
        pt, pb = tmp_path / "t.pbtxt", tmp_path / "t.pb"
        pt.write_text(pbtxt(dtype, dim, "".join("\\%03o" % b for b in content)))
        pb.write_bytes(wire(code, dim, content))
        for path, fmt in ((pt, TEXT), (pb, BINARY_GRAPH)):
            if ok:
                g = dump(path, fmt)
                assert by_name(g)["c"]["op"] == "Const"
            else:
                with pytest.raises(ValueError):
                    dump(path, fmt)
