"""ctypes view of tests/tf_mock/_build/libnann_tf_ops_mock.so: OUR op shim (nann_amd/tf_ops/nann_tf_ops.cc) built against
the functional TensorFlow op-kernel model in this directory, plus the executor-side driver (tfm_driver.cc).

    k = Kernel("GroupGather", T=np.int32, unique=False)        # node attrs -> kernel (defaults from the OpDef)
    r = k(pv, prs, iv, irs)                                     # numpy in, Result out (r.ok, r.code, r.msg, r.outputs)
    r = k(values, splits, Ref(bitmap))                          # Ref input: `bitmap` (numpy) is updated IN PLACE
"""
import ctypes as C

import numpy as np

from . import build as _build

# TensorFlow's DataType numbers (core/framework/types.proto)
DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.uint8): 4, np.dtype(np.int16): 5,
      np.dtype(np.int8): 6, np.dtype(np.int64): 9, np.dtype(np.bool_): 10, np.dtype(np.float16): 19}
NP = {v: k for k, v in DT.items()}
# error codes (core/lib/core/error_codes.proto)
OK, INVALID_ARGUMENT, DEADLINE_EXCEEDED, NOT_FOUND, UNIMPLEMENTED, INTERNAL = 0, 3, 4, 5, 12, 13

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_build.build())
        for f in ("tfm_last_error", "tfm_op_list", "tfm_duplicate_ops", "tfm_op_def", "tfm_kernel_list", "tfm_ctx_status_msg"):
            getattr(L, f).restype = C.c_char_p
        L.tfm_kernel_new.restype = C.c_void_p
        L.tfm_ctx_new.restype = C.c_void_p
        L.tfm_ctx_new.argtypes = [C.c_void_p]
        L.tfm_kernel_delete.argtypes = [C.c_void_p]
        L.tfm_ctx_delete.argtypes = [C.c_void_p]
        L.nann_tf_ops_stats.restype = None
        _LIB = L
    return _LIB


def last_error():
    return lib().tfm_last_error().decode("utf-8", "replace"), lib().tfm_last_code()


class KernelError(RuntimeError):
    """kernel lookup / construction failed (what TensorFlow reports when the graph is loaded / the kernel is created)"""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code, self.msg = code, msg


class Ref:
    """marks an input as a Ref(...) input backed by this numpy array (the graph's variable)"""

    def __init__(self, array):
        assert isinstance(array, np.ndarray) and array.flags["C_CONTIGUOUS"] and array.flags["WRITEABLE"]
        self.array = array


class Shape:
    def __init__(self, *dims):
        self.dims = [int(d) for d in dims]


def _is_dtype(v):
    try:
        return isinstance(v, (type, np.dtype)) and np.dtype(v) in DT
    except TypeError:
        return False


def encode_attrs(attrs):
    recs = []
    for name, v in attrs.items():
        if isinstance(v, bool):
            kind, val = "bool", "1" if v else "0"
        elif isinstance(v, (int, np.integer)):
            kind, val = "int", str(int(v))
        elif isinstance(v, float):
            kind, val = "float", repr(v)
        elif isinstance(v, str):
            kind, val = "string", v
        elif isinstance(v, Shape):
            kind, val = "shape", ",".join(str(d) for d in v.dims)
        elif _is_dtype(v):
            kind, val = "type", str(DT[np.dtype(v)])
        elif isinstance(v, (list, tuple)) and v and all(_is_dtype(x) for x in v):
            kind, val = "list(type)", "\x1d".join(str(DT[np.dtype(x)]) for x in v)
        elif isinstance(v, (list, tuple)) and all(isinstance(x, str) for x in v):
            kind, val = "list(string)", "\x1d".join(v)
        else:
            raise TypeError(f"attr {name}: cannot encode {v!r}")
        recs.append("\x1f".join((name, kind, val)))
    return "\x1e".join(recs).encode()


def op_list():
    return lib().tfm_op_list().decode().split()


def duplicate_ops():
    return lib().tfm_duplicate_ops().decode().split()


def op_def(op):
    t = lib().tfm_op_def(op.encode())
    if t is None:
        raise KernelError(lib().tfm_last_code(), lib().tfm_last_error().decode())
    return t.decode().strip().split("\n")


def kernel_list():
    return [tuple(line.split("|")) for line in lib().tfm_kernel_list().decode().strip().split("\n")]


def stats(reset=False):
    out = (C.c_int64 * 6)()
    lib().nann_tf_ops_stats(out, C.c_int(1 if reset else 0))
    return dict(zip(("h2d_bytes", "d2h_bytes", "registry_hits", "registry_misses", "blaze_runs", "blaze_rejected"), list(out)))


def infer_shapes(op, input_shapes, **attrs):
    """runs the op's SetShapeFn; input_shapes: list of None (unknown rank) or lists with None for unknown dims"""
    ranks = (C.c_int * max(1, len(input_shapes)))(*[-1 if s is None else len(s) for s in input_shapes])
    flat = [(-1 if d is None else int(d)) for s in input_shapes if s is not None for d in s]
    dims = (C.c_int64 * max(1, len(flat)))(*flat)
    out = C.create_string_buffer(4096)
    rc = lib().tfm_infer_shapes(op.encode(), encode_attrs(attrs), C.c_int(len(input_shapes)), ranks, dims, out, C.c_int(4096))
    if rc:
        raise KernelError(rc, lib().tfm_last_error().decode())

    def parse(line):
        if line == "?":
            return None
        body = line[1:-1]
        return [] if not body else [None if t == "?" else int(t) for t in body.split(",")]
    return [parse(l) for l in out.value.decode().strip().split("\n")] if out.value else []


class Output:
    def __init__(self, array, address, is_ref, forwarded_from, is_set):
        self.array, self.address, self.is_ref, self.forwarded_from, self.is_set = array, address, is_ref, forwarded_from, is_set


class Result:
    def __init__(self, code, msg, outputs, async_info):
        self.code, self.msg, self.outputs, self.async_info = code, msg, outputs, async_info
        self.ok = code == 0

    def __getitem__(self, i):
        return self.outputs[i].array


class Call:
    """one OpKernelContext: inputs bound, kernel started; .wait() -> Result"""

    def __init__(self, kernel, inputs):
        self.kernel = kernel
        self._keep = []
        self.ctx = C.c_void_p(lib().tfm_ctx_new(kernel.handle))
        for x in inputs:
            is_ref = isinstance(x, Ref)
            a = x.array if is_ref else np.asarray(x, order="C")  # keeps 0-d scalars 0-d
            assert a.dtype in DT, a.dtype
            self._keep.append(a)
            dims = (C.c_int64 * max(1, a.ndim))(*a.shape)
            lib().tfm_ctx_add_input(self.ctx, C.c_int(DT[a.dtype]), C.c_int(a.ndim), dims, C.c_void_p(a.ctypes.data),
                                    C.c_int(1 if is_ref else 0))
        self.start_rc = lib().tfm_ctx_start(self.ctx)
        self.start_err = lib().tfm_last_error().decode() if self.start_rc else ""

    def done(self, timeout_ms=0):
        return bool(lib().tfm_ctx_wait(self.ctx, C.c_int(timeout_ms)))

    def wait(self, timeout_ms=120000):
        if self.start_rc:
            return Result(self.start_rc, self.start_err, [], (0, 0, 0))
        assert self.done(timeout_ms), "the kernel did not finish"
        code = lib().tfm_ctx_status(self.ctx)
        msg = lib().tfm_ctx_status_msg(self.ctx).decode("utf-8", "replace")
        ai = (C.c_int * 3)()
        lib().tfm_ctx_async_info(self.ctx, ai)
        outs = []
        for i in range(lib().tfm_ctx_num_outputs(self.ctx)):
            info, dims, data = (C.c_int * 4)(), (C.c_int64 * 8)(), C.c_void_p()
            assert lib().tfm_ctx_output(self.ctx, C.c_int(i), info, dims, C.byref(data)) == 0
            is_set, dt, nd, fwd = info[0], info[1], info[2], info[3]
            is_ref = dt > 100
            arr = None
            if is_set:
                npdt = NP[dt - 100 if is_ref else dt]
                shape = tuple(dims[d] for d in range(nd))
                n = int(np.prod(shape)) if shape else 1
                if n and data.value:
                    buf = (C.c_char * (n * npdt.itemsize)).from_address(data.value)
                    arr = np.frombuffer(buf, dtype=npdt).reshape(shape).copy()
                else:
                    arr = np.zeros(shape, npdt)
            outs.append(Output(arr, data.value, is_ref, fwd, bool(is_set)))
        return Result(code, msg, outs, tuple(ai))

    def close(self):
        if self.ctx:
            lib().tfm_ctx_delete(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Kernel:
    def __init__(self, op, **attrs):
        self.op = op
        h = lib().tfm_kernel_new(op.encode(), encode_attrs(attrs))
        if not h:
            raise KernelError(lib().tfm_last_code(), lib().tfm_last_error().decode("utf-8", "replace"))
        self.handle = C.c_void_p(h)
        self.is_async = bool(lib().tfm_kernel_is_async(self.handle))

    def start(self, *inputs):
        return Call(self, inputs)

    def __call__(self, *inputs):
        c = Call(self, inputs)
        r = c.wait()
        r._call = c  # outputs that alias kernel memory (HugeConst) stay valid while the result is held
        return r

    def close(self):
        if self.handle:
            lib().tfm_kernel_delete(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
