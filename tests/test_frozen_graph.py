"""The frozen-GraphDef reader behind BlazeXlaOp.graph_def (csrc/host/nann_graphdef.h) against graphs written by
nann_amd/frozen_graph.py with the reference's node naming (convert_meta.py:361-398; model.py:189-233,
model_util.py:32-97): CPU part (parser + weight extraction through libnann_host.so); the GPU part -- the same
file through nann_model_load / BlazeXlaOp's path, logits equal to the weights-directory form -- is in
tests/test_ops_gpu.py."""
import ctypes as C
import os

import numpy as np
import pytest

from nann_amd import frozen_graph, index_build, synth

ORDER = (["wq1", "bq1", "aq", "wq2", "bq2", "wk1", "bk1", "ak", "wk2", "bk2"] +
         [(k, l) for l in range(3) for k in ("w", "b", "bn_scale", "bn_shift", "alpha")] + [("w", 3)])


def read_attention(path):
    lib = C.CDLL(index_build.build_host_lib())
    counts = (C.c_int64 * 26)()
    d, e = C.c_int32(0), C.c_int32(0)
    err = C.create_string_buffer(512)
    rc = lib.nann_graphdef_attention(str(path).encode(), counts, None, C.byref(d), C.byref(e), err, 512)
    if rc:
        raise ValueError(err.value.decode())
    flat = np.zeros(sum(counts), np.float32)
    rc = lib.nann_graphdef_attention(str(path).encode(), counts, flat.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(e), err, 512)
    assert rc == 0
    out, o = [], 0
    for c in counts:
        out.append(flat[o:o + c]); o += c
    return d.value, e.value, out


def expected(w):
    return [np.asarray(w[k] if isinstance(k, str) else w[k[0]][k[1]], np.float32).reshape(-1) for k in ORDER]


@pytest.mark.parametrize("d", [64, 128])
@pytest.mark.parametrize("folded", [True, False])
def test_reader_recovers_every_tensor_bit_for_bit(tmp_path, d, folded):
    w = synth.make_attn_weights(d, 64)
    p = tmp_path / "frozen_graph.pb"
    frozen_graph.write_attention_graph(str(p), w, folded=folded)
    gd, ge, got = read_attention(p)
    assert (gd, ge) == (d, 64)
    for name, a, b in zip(ORDER, got, expected(w)):
        assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), name


@pytest.mark.parametrize("folded", [True, False])
def test_batch_norm_statistics_are_folded_like_tensorflow_does(tmp_path, folded):
    """real gamma / beta / moving_mean / moving_variance: scale = gamma * rsqrt(var + eps), shift = beta - mean *
    scale (nn.batch_normalization), whether fold_constants already did it or the reader has to"""
    w = synth.make_attn_weights(64, 64)
    rng = np.random.default_rng(5)
    bn = []
    for l, n in enumerate((128, 64, 32)):
        bn.append({"gamma": (1 + 0.1 * rng.standard_normal(n)).astype(np.float32), "beta": (0.1 * rng.standard_normal(n)).astype(np.float32),
                   "mean": (0.3 * rng.standard_normal(n)).astype(np.float32), "var": (0.5 + rng.random(n)).astype(np.float32), "eps": 1e-3})
    p = tmp_path / "g.pb"
    frozen_graph.write_attention_graph(str(p), w, folded=folded, bn=bn, alpha_as_val_list=False)
    _, _, got = read_attention(p)
    for l in range(3):
        scale = (bn[l]["gamma"] * (np.float32(1) / np.sqrt(bn[l]["var"] + np.float32(1e-3)))).astype(np.float32)
        shift = (bn[l]["beta"] - bn[l]["mean"] * scale).astype(np.float32)
        gs, gh = got[10 + 5 * l + 2], got[10 + 5 * l + 3]
        if folded:
            assert (gs.view(np.uint32) == scale.view(np.uint32)).all() and (gh.view(np.uint32) == shift.view(np.uint32)).all()
        else:  # rsqrt * gamma in TensorFlow's operand order; same value up to the commutative product
            np.testing.assert_allclose(gs, scale, rtol=2e-7)
            np.testing.assert_allclose(gh, shift, rtol=1e-6, atol=1e-7)


def test_value_lists_repeat_their_last_value(tmp_path):
    """a constant_initializer'd vector is serialised as ONE float_val with the full shape"""
    w = synth.make_attn_weights(64, 64)
    w["aq"] = np.full(128, 0.25, np.float32)
    p = tmp_path / "g.pb"
    frozen_graph.write_attention_graph(str(p), w, folded=False, alpha_as_val_list=True)
    _, _, got = read_attention(p)
    assert got[2].shape == (128,) and (got[2] == np.float32(0.25)).all()


def test_errors_name_what_is_missing(tmp_path):
    p = tmp_path / "junk.pb"
    p.write_bytes(b"node {\n  name: \"x\"\n}\n")  # a text-format GraphDef is not read
    with pytest.raises(ValueError):
        read_attention(p)
    with pytest.raises(ValueError, match="Fail to open"):
        read_attention(tmp_path / "missing.pb")
    # a graph without the DNN tower
    blob = frozen_graph.const("nonlinear_attention/dense/kernel", np.zeros((64, 128), np.float32))
    (tmp_path / "partial.pb").write_bytes(blob)
    with pytest.raises(ValueError, match="nonlinear_attention/dense"):
        read_attention(tmp_path / "partial.pb")
    # truncated file
    w = synth.make_attn_weights(64, 64)
    blob = frozen_graph.write_attention_graph(str(tmp_path / "ok.pb"), w)
    (tmp_path / "cut.pb").write_bytes(blob[: len(blob) // 2])
    with pytest.raises(ValueError):
        read_attention(tmp_path / "cut.pb")


def test_reader_survives_corrupted_files(tmp_path):
    """Truncations, flipped bytes and inserted garbage: the reader either reports an error or (when only weight bytes
    were hit) returns tensors -- it never crashes or over-reads (this runs inside the test process)."""
    lib = C.CDLL(index_build.build_host_lib())
    blob = frozen_graph.write_attention_graph(str(tmp_path / "ok.pb"), synth.make_attn_weights(64, 64))
    rng = np.random.default_rng(0)
    path = tmp_path / "fuzz.pb"
    outcomes = set()
    for i in range(160):
        b = bytearray(blob)
        kind = i % 4
        if kind == 0:
            b = b[: int(rng.integers(1, len(b)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 2:  # among the first 4 KB: node headers, names, varints
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, 4096))] = int(rng.integers(0, 256))
        else:
            pos = int(rng.integers(0, len(b)))
            b[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 64))).astype(np.uint8))
        path.write_bytes(bytes(b))
        counts = (C.c_int64 * 26)()
        d, e = C.c_int32(0), C.c_int32(0)
        err = C.create_string_buffer(512)
        rc = lib.nann_graphdef_attention(str(path).encode(), counts, None, C.byref(d), C.byref(e), err, 512)
        if rc == 0:
            assert 0 < sum(counts) < 10_000_000
            flat = np.zeros(sum(counts), np.float32)
            rc = lib.nann_graphdef_attention(str(path).encode(), counts, flat.ctypes.data_as(C.c_void_p), C.byref(d), C.byref(e),
                                             err, 512)
        else:
            assert err.value  # an error always says what failed
        outcomes.add(rc)
    assert outcomes <= {0, 1} and 1 in outcomes
