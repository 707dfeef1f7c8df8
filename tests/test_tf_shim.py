"""CPU: the TensorFlow op-kernel shim (nann_amd/tf_ops/nann_tf_ops.cc) as a LOADED shared object.

TensorFlow is not in the image, so the shim is built against tests/tf_mock -- a functional model of the op-kernel API
(ref-counted tensors, REGISTER_OP spec parsing, kernel lookup by device + type constraints, attr type checks, shape
inference) -- and what it registered at static-init time is read back from the registries: same op names, input / output /
attr specs, dtypes and kernels as the reference's REGISTER_OP / REGISTER_KERNEL_BUILDER blocks (cited per op).  No GPU
here: kernels that touch the device only in Compute are constructed; the compute side is tests/test_tf_shim_gpu.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tf_mock import harness as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "nann_amd", "tf_ops", "nann_tf_ops.cc")

# the reference's interface per op, spec strings verbatim (UO = tensorflow/tensorflow/core/user_ops)
REFERENCE_SURFACE = {
    # UO/beam_search_op/GroupGather_kernel.cc:18-26
    "GroupGather": ["Input(params_values: T)", "Input(params_row_splits: int64)", "Input(indices_values: int64)",
                    "Input(indices_row_splits: int64)", "Output(ret_values: T)", "Output(ret_row_splits: int64)",
                    "Attr(T: {int32, int64})", "Attr(unique: bool = false)"],
    # UO/bitmap_op/bitmap_ops.cc:28-32
    "BitmapInit": ["Input(idx: T)", "Input(length: int32)", "Output(bitmap: int32)", "Attr(T: {int32, int64})"],
    # UO/bitmap_op/bitmap_ops.cc:83-88
    "BitmapDifference": ["Input(idx_next: T)", "Input(idx_flag: int32)", "Output(idx_next_new: T)",
                         "Output(idx_flag_new: int32)", "Attr(T: {int32, int64})"],
    # UO/bitmap_op/bitmap_ops.cc:150-157
    "BitmapRefDifference": ["Input(idx_next_values: T)", "Input(idx_next_row_splits: int64)", "Input(idx_flag: Ref (int32))",
                            "Output(c_values: T)", "Output(c_row_splits: int64)", "Output(idx_flag_new: Ref (int32))",
                            "Attr(T: {int32, int64})"],
    # UO/bitmap_op/bitmap_ops.cc:264-273
    "BloomFilterDifference": ["Input(idx_next_values: T)", "Input(idx_next_row_splits: int64)", "Input(idx_flag: Ref (int32))",
                              "Output(c_values: T)", "Output(c_row_splits: int64)", "Output(idx_flag_new: Ref (int32))",
                              "Attr(bucket: int >= 0 = 0)", "Attr(bucket_size: int >= 1)", "Attr(T: {int32, int64})"],
    # UO/topk_op/BlazeTopK_kernel.cc:13-19
    "BlazeTopK": ["Input(input: T)", "Input(k: Tindices)", "Output(value: T)", "Output(index: Tindices)",
                  "Attr(T: {half, float, double})", "Attr(Tindices: {int32})"],
    # UO/topk_op/BatchTopKOnRT_kernel.cc:25-33
    "BatchTopKOnRT": ["Input(values_in: T)", "Input(row_splits_in: int64)", "Input(k: int64)", "Output(values_out: T)",
                      "Output(idx_out: int64)", "Output(row_splits_out: int64)", "Attr(T: {double, float, half})",
                      "Attr(ascending: bool = false)"],
    # UO/huge_const_op/huge_const_op.cc:58-62
    "HugeConst": ["Output(output: dtype)", "Attr(dtype: type)", "Attr(shape: shape)", "Attr(path: string)"],
    # UO/blaze_op/blaze_xla_kernel.cc:24-32
    "BlazeXlaOp": ["Attr(InT: list({int8,int64,float16,float32,int32}))", "Attr(OutT: list({int8,int64,float16,float32,int32}))",
                   "Attr(input_names: list(string) >= 0)", "Attr(output_names: list(string) >= 0)", "Attr(graph_def: string)",
                   "Attr(blaze_option_path: string)", "Input(in_tensor: InT)", "Output(out_tensor: OutT)"],
}
# the kernels the replaced files register for DEVICE_CPU (GroupGather_kernel.cc:177-182, bitmap_ops.cc:428-435,
# BlazeTopK_kernel.cc:108-113 [float], BatchTopKOnRT_kernel.cc:159-165 [float], huge_const_op.cc:228, blaze_xla_kernel.cc:260)
REFERENCE_KERNELS = ([(op, "CPU", f"T={t}") for op in ("GroupGather", "BitmapInit", "BitmapDifference", "BitmapRefDifference",
                                                        "BloomFilterDifference") for t in ("int32", "int64")] +
                     [("BlazeTopK", "CPU", "T=float"), ("BatchTopKOnRT", "CPU", "T=float"), ("HugeConst", "CPU", ""),
                      ("BlazeXlaOp", "CPU", "")])


def test_shim_compiles_warning_free_against_the_op_kernel_api():
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", os.path.join(ROOT, "tests", "tf_mock"),
           "-I", os.path.join(ROOT, "include"), SHIM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr[-3000:]


def test_shim_includes_only_tensorflow_the_abi_header_and_the_standard_library():
    includes = [line.split()[1] for line in open(SHIM) if line.startswith("#include")]
    for inc in includes:
        assert inc in ('"nann_hip.h"', '"tensorflow/core/framework/op_kernel.h"', '"tensorflow/core/framework/shape_inference.h"') \
            or (inc.startswith("<") and "/" not in inc and "." not in inc), inc


def test_registered_op_surface_equals_the_reference():
    ops = H.op_list()
    assert sorted(ops) == sorted(list(REFERENCE_SURFACE) + ["NannHnswSearch"])
    assert H.duplicate_ops() == []
    for op, want in REFERENCE_SURFACE.items():
        got = H.op_def(op)
        assert got[0] == f"Op({op})" and not any(l.startswith("ERROR") for l in got), got
        assert sorted(got[1:]) == sorted(want), (op, got)
        # argument ORDER is part of the interface (positional inputs / outputs of a NodeDef)
        assert [l for l in got if l.startswith(("Input", "Output"))] == [l for l in want if l.startswith(("Input", "Output"))], op


def test_registered_kernels_cover_every_reference_registration():
    kernels = H.kernel_list()
    for k in REFERENCE_KERNELS:
        assert kernels.count(k) == 1, (k, kernels)
    assert ("NannHnswSearch", "CPU", "") in kernels
    assert len(kernels) == len(REFERENCE_KERNELS) + 1


def test_node_attrs_are_validated_like_a_nodedef():
    k = H.Kernel("GroupGather", T=np.int32)  # `unique` takes its default
    assert not k.is_async
    with pytest.raises(H.KernelError) as e:  # T is required
        H.Kernel("GroupGather")
    assert e.value.code == H.INVALID_ARGUMENT and "missing attr 'T'" in e.value.msg
    with pytest.raises(H.KernelError) as e:  # not in {int32, int64}
        H.Kernel("GroupGather", T=np.float32)
    assert "not in the list of allowed values" in e.value.msg
    with pytest.raises(H.KernelError) as e:
        H.Kernel("GroupGather", T=np.int32, uniq=True)
    assert "not in Op" in e.value.msg
    with pytest.raises(H.KernelError) as e:  # int >= 1
        H.Kernel("BloomFilterDifference", T=np.int32, bucket_size=0)
    assert "must be at least minimum 1" in e.value.msg
    with pytest.raises(H.KernelError) as e:  # the reference registers half / double kernels too; this shim: float
        H.Kernel("BlazeTopK", T=np.float64, Tindices=np.int32)
    assert e.value.code == H.NOT_FOUND and "No registered 'BlazeTopK' OpKernel" in e.value.msg
    with pytest.raises(H.KernelError) as e:
        H.Kernel("NoSuchOp")
    assert e.value.code == H.NOT_FOUND


def test_inputs_are_checked_against_the_opdef():
    k = H.Kernel("BitmapRefDifference", T=np.int32)
    flags = np.zeros(4, np.int32)
    r = k(np.array([1], np.int32), np.array([0, 1], np.int64), flags)  # the bitmap must be a Ref
    assert r.code == H.INVALID_ARGUMENT and "must be a Ref" in r.msg
    r = k(np.array([1], np.int64), np.array([0, 1], np.int64), H.Ref(flags))  # T = int32 node fed int64
    assert r.code == H.INVALID_ARGUMENT and "has dtype int64, expected int32" in r.msg
    r = k(np.array([1], np.int32), np.array([0, 1], np.int64))
    assert r.code == H.INVALID_ARGUMENT and "is missing" in r.msg


def test_shape_functions_run():
    assert H.infer_shapes("GroupGather", [[None], [1001], [None], [2]], T=np.int32) == [[None], [2]]
    with pytest.raises(H.KernelError) as e:
        H.infer_shapes("GroupGather", [[3, 3], [4], [2], [2]], T=np.int32)
    assert "Shape must be rank 1 but is rank 2" in e.value.msg
    assert H.infer_shapes("BitmapRefDifference", [[8192], [2], [31250]], T=np.int64) == [[None], [2], [31250]]
    assert H.infer_shapes("BloomFilterDifference", [None, [3], [10]], T=np.int32, bucket_size=10) == [[None], [3], [10]]
    assert H.infer_shapes("BitmapInit", [[5], []], T=np.int32) == [[None]]
    assert H.infer_shapes("BitmapDifference", [[50000], [62500]], T=np.int64) == [[None], [62500]]
    assert H.infer_shapes("BatchTopKOnRT", [[24], [5], [4]], T=np.float32) == [[None], [None], [5]]
    assert H.infer_shapes("BatchTopKOnRT", [[24], [5], []], T=np.float32) == [[None], [None], [5]]
    with pytest.raises(H.KernelError) as e:  # BatchTopKOnRT_kernel.cc:39-43
        H.infer_shapes("BatchTopKOnRT", [[24], [5], [3]], T=np.float32)
    assert "length of k != number of groups: 3 != 5 - 1" in e.value.msg
    assert H.infer_shapes("BlazeTopK", [[4, 100], []], T=np.float32, Tindices=np.int32) == [[4, None], [4, None]]
    assert H.infer_shapes("HugeConst", [], dtype=np.float16, shape=H.Shape(1000, 128), path="x.npy") == [[1000, 128]]
    assert H.infer_shapes("BlazeXlaOp", [None, None], InT=[np.float16, np.float16], OutT=[np.float32], input_names=["a", "b"],
                          output_names=["c"], graph_def="g", blaze_option_path="") == [None]


# ---- BlazeXlaOp's blaze_option_path (blaze_xla_kernel.cc:156-167; config.proto:805-841) -------------------------
def _blaze_options(attr):
    from nann_amd import index_build
    L = C.CDLL(index_build.build_host_lib())
    out = (C.c_int32 * 8)()
    err = C.create_string_buffer(512)
    rc = L.nann_host_blaze_options(attr.encode(), out, err, C.c_int32(512))
    names = ("wait_ms", "run_mode", "xla_compilation", "auto_mixed_precision", "disable_output_padding", "n_warmup_batchsize",
             "max_warmup_batchsize", "from_file")
    return rc, dict(zip(names, list(out))), err.value.decode()


# a file in the shape of NANN_impls/nann/delivery/opt_default.conf (no outer braces, ',' after every field, a list, a
# nested ConfigProto with a '#' comment)
_CONF = """  wait_ms: 7,
  xla_compilation: true,
  auto_mixed_precision: true,
  warmup_batchsize: [
    1,
    200,
    14800
  ],
  disable_output_padding: true,
  no_warmup_inputs: ["inference_feed_inputs/user_seq_emb"],
  config_proto: {
    graph_options: {
      rewrite_options: {
        constant_folding: ON,
        meta_optimizer_timeout_ms: 20000000000,
        #original_delivery_optimization: ON
      }
    },
    force_run_in_caller_thread: true
  },
  use_single_threaded_executor: false,
  gemm_optimization: false
"""


def test_blaze_options_are_read_as_the_reference_reads_them(tmp_path):
    p = tmp_path / "opt.conf"
    p.write_text(_CONF)
    rc, o, err = _blaze_options(str(p))  # 1) the attr as a file
    assert rc == 0, err
    assert o == {"wait_ms": 7, "run_mode": 0, "xla_compilation": 1, "auto_mixed_precision": 1, "disable_output_padding": 1,
                 "n_warmup_batchsize": 3, "max_warmup_batchsize": 14800, "from_file": 1}
    rc, o, err = _blaze_options("wait_ms: 3 run_mode: SKIP")  # 2) the attr string itself
    assert rc == 0 and o["wait_ms"] == 3 and o["run_mode"] == 2 and o["from_file"] == 0
    rc, o, err = _blaze_options("")  # an empty message: all defaults
    assert rc == 0 and o["wait_ms"] == 0 and o["run_mode"] == 0
    rc, o, err = _blaze_options(str(tmp_path / "missing.conf"))  # neither a file nor text format
    assert rc == 1 and err.startswith("parse proto from " + str(tmp_path / "missing.conf") + " failed")
    rc, o, err = _blaze_options("wait_millis: 3")  # protobuf refuses unknown fields
    assert rc == 1 and "no field named 'wait_millis'" in err
    rc, o, err = _blaze_options("wait_ms: soon")
    assert rc == 1 and "not an int32" in err
    q = tmp_path / "bad.conf"  # a file that does not parse falls through to the string, which does not parse either
    q.write_text("wait_ms: {")
    assert _blaze_options(str(q))[0] == 1
