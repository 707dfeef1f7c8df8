"""CPU: the TensorFlow op-kernel shim (nann_amd/tf_ops/nann_tf_ops.cc) keeps the reference's
op surface and compiles against the op-kernel API (syntax check against tests/tf_stub, a
compile-only stub of the few TF classes the shim touches -- TensorFlow is not in the image)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "nann_amd", "tf_ops", "nann_tf_ops.cc")


def test_shim_compiles_against_the_op_kernel_api():
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "tf_stub"),
           "-I", os.path.join(ROOT, "include"), SHIM]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_shim_registers_the_reference_op_surface():
    """Same op / input / output / attr names and dtypes as the reference's REGISTER_OP blocks
    (GroupGather_kernel.cc:18-26, bitmap_ops.cc:150-157)."""
    text = open(SHIM).read()
    gg = text[text.index('REGISTER_OP("GroupGather")'):]
    for frag in ['.Input("params_values: T")', '.Input("params_row_splits: int64")',
                 '.Input("indices_values: int64")', '.Input("indices_row_splits: int64")',
                 '.Output("ret_values: T")', '.Output("ret_row_splits: int64")',
                 '.Attr("T: {int32, int64}")', '.Attr("unique: bool = false")']:
        assert frag in gg[:900], frag
    bm = text[text.index('REGISTER_OP("BitmapRefDifference")'):]
    for frag in ['.Input("idx_next_values: T")', '.Input("idx_next_row_splits: int64")',
                 '.Input("idx_flag: Ref (int32)")', '.Output("c_values: T")',
                 '.Output("c_row_splits: int64")', '.Output("idx_flag_new: Ref (int32)")',
                 '.Attr("T: {int32, int64}")']:
        assert frag in bm[:900], frag
    for op in ("GroupGather", "BitmapRefDifference"):  # both id types the reference registers
        for t in ("int32", "int64"):                     # (GroupGather_kernel.cc:177-182, bitmap_ops.cc:428-435)
            assert re.search(r'Name\("%s"\)\.Device\(DEVICE_CPU\)\.TypeConstraint<%s>\("T"\)' % (op, t), text), (op, t)


def test_shim_registers_blaze_xla_op_and_huge_const():
    """The other two surfaces of SURVEY.md 8(b): BlazeXlaOp (blaze_xla_kernel.cc:24-33) and HugeConst
    (huge_const_op.cc:58-70), attr for attr."""
    text = open(SHIM).read()
    bx = text[text.index('REGISTER_OP("BlazeXlaOp")'):][:700]
    for frag in ['.Attr("InT: list({int8,int64,float16,float32,int32})")',
                 '.Attr("OutT: list({int8,int64,float16,float32,int32})")',
                 '.Attr("input_names: list(string) >= 0")', '.Attr("output_names: list(string) >= 0")',
                 '.Attr("graph_def: string")', '.Attr("blaze_option_path: string")',
                 '.Input("in_tensor: InT")', '.Output("out_tensor: OutT")']:
        assert frag in bx, frag
    hc = text[text.index('REGISTER_OP("HugeConst")'):][:400]
    for frag in ['.Output("output: dtype")', '.Attr("dtype: type")', '.Attr("shape: shape")', '.Attr("path: string")']:
        assert frag in hc, frag
    assert 'Name("BlazeXlaOp").Device(DEVICE_CPU)' in text and "public AsyncOpKernel" in text
    assert 'Name("HugeConst").Device(DEVICE_CPU)' in text
    # the fused node can score with the MLP model as well as L2
    assert '.Attr("scorer_dir: string = \'\'")' in text
