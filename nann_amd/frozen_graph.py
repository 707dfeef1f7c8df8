"""Writes the reference scorer model as a frozen TensorFlow GraphDef, without TensorFlow.

BlazeXlaOp's `graph_def` attr names the file convert_meta.py:361-398 produces (`frozen_graph.pb`: the export
graph of Model.forward(training=False), model.py:189-233, variables frozen to Const nodes, graph transforms,
binary serialisation).  nann_model_load reads that file (csrc/host/nann_graphdef.h); this module is its
counterpart for tests and for hosts that hold the weights as arrays: it hand-encodes the protobuf wire format
(GraphDef -> NodeDef -> AttrValue -> TensorProto) with the node names TensorFlow 1.15 gives the ops the
reference's Python creates:

  nonlinear_attention/dense{,_1,_2,_3}/{kernel,bias}, nonlinear_attention/prelu_{q,k}     (model_util.py:70-97)
  {1,2,3}_dnn/fc/{kernel,bias}, {1,2,3}_dnn/bn/{gamma,beta,moving_mean,moving_variance},
  {1,2,3}_dnn/prelu, 4_dnn/fc/kernel                                                      (model_util.py:32-67)
  consumers: <layer>/Tensordot/MatMul, <layer>/BiasAdd, <scope>/mul (PReLU), <dnn>/bn/batchnorm/{mul_1,add_1}

in either of the two forms such a file takes:
  folded=False  freeze_graph only: Const `V` + Identity `V/read`, Tensordot's transpose/reshape of the kernel and
                batch norm's rsqrt/mul/sub arithmetic still in the graph;
  folded=True   after fold_constants: every constant sub-expression replaced by a Const named
                `<folded op>/_<n>__cf__<n>` (common_runtime/constant_folding.cc), the originals gone.
The arithmetic BETWEEN the weighted ops (einsum, softmax, the concat) is abbreviated to placeholder-typed
stand-ins: the file is a faithful container of weights and consumer names, not a graph TensorFlow could run.
"""
import struct

import numpy as np

DT_FLOAT, DT_INT32, DT_HALF = 1, 3, 19


# ---- protobuf wire format ---------------------------------------------------------------------------------
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):  # length-delimited
    return _key(field, 2) + _varint(len(payload)) + payload


def _shape_proto(dims):
    return b"".join(_ld(2, _key(1, 0) + _varint(int(d))) for d in dims)


def tensor_proto(arr, as_val_list=False):
    """TensorProto of a numpy array: tensor_content (what tf.make_tensor_proto writes for arrays), or, with
    as_val_list, the *_val list holding ONE value that readers repeat (a constant_initializer'd vector)."""
    arr = np.asarray(arr)
    dt = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32, np.dtype(np.float16): DT_HALF}[arr.dtype]
    out = _key(1, 0) + _varint(dt) + _ld(2, _shape_proto(arr.shape))
    if as_val_list:
        assert arr.size and (arr.reshape(-1) == arr.reshape(-1)[0]).all()
        if dt == DT_FLOAT:
            out += _ld(5, struct.pack("<f", float(arr.reshape(-1)[0])))            # float_val, packed
        elif dt == DT_INT32:
            out += _ld(7, _varint(int(arr.reshape(-1)[0])))                        # int_val
        else:
            out += _ld(13, _varint(int(arr.reshape(-1).view(np.uint16)[0])))       # half_val
    else:
        out += _ld(4, np.ascontiguousarray(arr).tobytes())
    return out


def _attr(key, value_payload):
    return _ld(5, _ld(1, key.encode()) + _ld(2, value_payload))


def _attr_type(key, dt):
    return _attr(key, _key(6, 0) + _varint(dt))


def node(name, op, inputs=(), attrs=b""):
    return _ld(1, _ld(1, name.encode()) + _ld(2, op.encode()) + b"".join(_ld(3, i.encode()) for i in inputs) + attrs)


def const(name, arr, as_val_list=False):
    arr = np.asarray(arr)
    dt = {np.dtype(np.float32): DT_FLOAT, np.dtype(np.int32): DT_INT32, np.dtype(np.float16): DT_HALF}[arr.dtype]
    return node(name, "Const", (), _attr_type("dtype", dt) + _attr("value", _ld(8, tensor_proto(arr, as_val_list))))


def placeholder(name, dt, dims):
    return node(name, "Placeholder", (), _attr_type("dtype", dt) + _attr("shape", _ld(7, _shape_proto(dims))))


# ---- the reference's scorer model -------------------------------------------------------------------------
class _Builder:
    def __init__(self, folded):
        self.folded, self.nodes, self.n_cf = folded, [], 0

    def add(self, n):
        self.nodes.append(n)

    def cf(self, op_name):  # the name constant folding gives the Const that replaces `op_name`
        self.n_cf += 1
        return f"{op_name}/_{self.n_cf}__cf__{self.n_cf}"

    def variable(self, name, arr, as_val_list=False):
        """-> the tensor name consumers read the variable through"""
        arr = np.asarray(arr, np.float32)
        if self.folded:
            nm = self.cf(name + "/read")
            self.add(const(nm, arr, as_val_list))
            return nm
        self.add(const(name, arr, as_val_list))
        self.add(node(name + "/read", "Identity", (name,), _attr_type("T", DT_FLOAT) +
                      _attr("_class", _ld(1, _ld(2, ("loc:@" + name).encode())))))
        return name + "/read"

    def dense(self, scope, x, kernel, bias):
        """tf.layers.dense on a rank-3 input (Dense.call -> tensordot): -> output tensor name"""
        kernel = np.asarray(kernel, np.float32)
        t = scope + "/Tensordot"
        self.add(node(t + "/Reshape", "Reshape", (x, t + "/stack")))  # the activations, flattened to [rows, in]
        if self.folded:  # transpose + reshape of the kernel were constants: one folded Const feeds the MatMul
            kname = self.cf(t + "/Reshape_1")
            self.add(const(kname, kernel))
        else:
            kread = self.variable(scope + "/kernel", kernel)
            self.add(const(t + "/transpose_1/perm", np.array([0, 1], np.int32)))
            self.add(node(t + "/transpose_1", "Transpose", (kread, t + "/transpose_1/perm"), _attr_type("T", DT_FLOAT)))
            self.add(const(t + "/Reshape_1/shape", np.array(kernel.shape, np.int32)))
            self.add(node(t + "/Reshape_1", "Reshape", (t + "/transpose_1", t + "/Reshape_1/shape"), _attr_type("T", DT_FLOAT)))
            kname = t + "/Reshape_1"
        self.add(node(t + "/MatMul", "MatMul", (t + "/Reshape", kname), _attr_type("T", DT_FLOAT)))
        self.add(node(t, "Reshape", (t + "/MatMul", t + "/concat_1")))
        if bias is None:
            return t
        b = self.variable(scope + "/bias", bias)
        self.add(node(scope + "/BiasAdd", "BiasAdd", (t, b), _attr_type("T", DT_FLOAT)))
        return scope + "/BiasAdd"

    def prelu(self, scope, suffix, var_name, x, alpha, as_val_list=False):
        """model_util.prelu: tf.maximum(0.0, x) + _alpha * tf.minimum(0.0, x), ops named in the CURRENT scope"""
        a = self.variable(var_name, alpha, as_val_list)
        mx, mn, mul, add = (f"{scope}/{op}{suffix}" for op in ("Maximum", "Minimum", "mul", "add"))
        self.add(node(mx, "Maximum", (scope + "/Maximum/x", x)))
        self.add(node(mn, "Minimum", (scope + "/Minimum/x", x)))
        self.add(node(mul, "Mul", (a, mn), _attr_type("T", DT_FLOAT)))
        self.add(node(add, "Add", (mx, mul), _attr_type("T", DT_FLOAT)))
        return add

    def batch_norm(self, scope, x, bn):
        """tf.layers.batch_normalization(training=False) -> nn.batch_normalization's batchnorm/ ops"""
        p = scope + "/bn/batchnorm"
        if self.folded:
            scale = (bn["gamma"] * (np.float32(1.0) / np.sqrt(bn["var"] + np.float32(bn["eps"])))).astype(np.float32)
            shift = (bn["beta"] - bn["mean"] * scale).astype(np.float32)
            mul, sub = self.cf(p + "/mul"), self.cf(p + "/sub")
            self.add(const(mul, scale))
            self.add(const(sub, shift))
        else:
            g, b = self.variable(scope + "/bn/gamma", bn["gamma"]), self.variable(scope + "/bn/beta", bn["beta"])
            m, v = self.variable(scope + "/bn/moving_mean", bn["mean"]), self.variable(scope + "/bn/moving_variance", bn["var"])
            self.add(const(p + "/add/y", np.array(bn["eps"], np.float32), as_val_list=True))
            self.add(node(p + "/add", "Add", (v, p + "/add/y"), _attr_type("T", DT_FLOAT)))
            self.add(node(p + "/Rsqrt", "Rsqrt", (p + "/add",), _attr_type("T", DT_FLOAT)))
            self.add(node(p + "/mul", "Mul", (p + "/Rsqrt", g), _attr_type("T", DT_FLOAT)))
            self.add(node(p + "/mul_2", "Mul", (m, p + "/mul"), _attr_type("T", DT_FLOAT)))
            self.add(node(p + "/sub", "Sub", (b, p + "/mul_2"), _attr_type("T", DT_FLOAT)))
            mul, sub = p + "/mul", p + "/sub"
        self.add(node(p + "/mul_1", "Mul", (x, mul), _attr_type("T", DT_FLOAT)))
        self.add(node(p + "/add_1", "Add", (p + "/mul_1", sub), _attr_type("T", DT_FLOAT)))
        return p + "/add_1"


def unfold_bn(scale, shift, rng=None, eps=1e-3):
    """gamma / beta / moving_mean / moving_variance whose inference fold reproduces (scale, shift) BIT FOR BIT:
    moving_variance = 0.999 (0.999f + 0.001f rounds to exactly 1.0f, rsqrt(1) = 1), moving_mean = 0."""
    scale, shift = np.asarray(scale, np.float32), np.asarray(shift, np.float32)
    return {"gamma": scale, "beta": shift, "mean": np.zeros_like(scale), "var": np.full_like(scale, np.float32(0.999)),
            "eps": eps}


def write_attention_graph(path, weights, seq_len=50, folded=True, bn=None, alpha_as_val_list=False):
    """weights: dict as synth.make_attn_weights (attention dense layers, DNN layers with batch norm already folded
    to bn_scale / bn_shift).  bn: optional list of 3 dicts {gamma, beta, mean, var, eps} to write real batch-norm
    statistics instead of unfold_bn(scale, shift).  Returns the bytes written."""
    w = weights
    d, e = np.asarray(w["wq1"]).shape[0], np.asarray(w["wk1"]).shape[0]
    B = _Builder(folded)
    feed = "inference_feed_inputs/"  # nann/delivery/constant.py:3; fp16 feeds + casts: convert_meta.py:326-358
    B.add(placeholder(feed + "user_seq_emb", DT_HALF, [1, seq_len, e]))
    B.add(placeholder(feed + "item_emb", DT_HALF, [-1, d]))
    for ph in ("user_seq_emb", "item_emb"):
        B.add(node(feed + ph + "/cast_float2half", "Cast", (feed + ph,), _attr_type("SrcT", DT_HALF) + _attr_type("DstT", DT_FLOAT)))
    user, item = feed + "user_seq_emb/cast_float2half", "strided_slice"
    B.add(node(item, "StridedSlice", (feed + "item_emb/cast_float2half",)))  # item_emb_ph[tf.newaxis, ...], model.py:91
    A = "nonlinear_attention"
    q = B.dense(A + "/dense", item, w["wq1"], w["bq1"])
    q = B.prelu(A, "", A + "/prelu_q", q, w["aq"], alpha_as_val_list)
    q_ = B.dense(A + "/dense_1", q, w["wq2"], w["bq2"])
    k = B.dense(A + "/dense_2", user, w["wk1"], w["bk1"])
    k = B.prelu(A, "_1", A + "/prelu_k", k, w["ak"])
    k_ = B.dense(A + "/dense_3", k, w["wk2"], w["bk2"])
    S = A + "/scale_dot_product"
    B.add(node(S + "/einsum/MatMul", "BatchMatMulV2", (q_, k_)))
    B.add(node(S + "/truediv", "RealDiv", (S + "/einsum/MatMul", S + "/truediv/y")))
    B.add(node(S + "/Softmax", "Softmax", (S + "/truediv",)))
    B.add(node(S + "/mul", "Mul", (S + "/Softmax", user)))
    B.add(node("Sum", "Sum", (S + "/mul", "Sum/reduction_indices")))
    B.add(node("concat", "ConcatV2", ("Sum", item, "concat/axis")))
    x = "concat"
    for l in range(3):
        scope = f"{l + 1}_dnn"
        x = B.dense(scope + "/fc", x, w["w"][l], w["b"][l])
        x = B.batch_norm(scope, x, bn[l] if bn else unfold_bn(w["bn_scale"][l], w["bn_shift"][l]))
        x = B.prelu(scope, "", scope + "/prelu", x, w["alpha"][l])
    x = B.dense("4_dnn/fc", x, np.asarray(w["w"][3], np.float32).reshape(-1, 1), None)
    B.add(node("final_logit", "Reshape", (x, "final_logit/shape")))
    B.add(node("inference_fetch_outputs/logits", "Identity", ("final_logit",), _attr_type("T", DT_FLOAT)))
    versions = _ld(4, _key(1, 0) + _varint(134))  # VersionDef.producer (TF 1.15)
    blob = b"".join(B.nodes) + versions
    with open(path, "wb") as f:
        f.write(blob)
    return blob
