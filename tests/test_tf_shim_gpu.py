"""-m gpu: EVERY kernel nann_amd/tf_ops/nann_tf_ops.cc registers, instantiated from its REGISTER_OP /
REGISTER_KERNEL_BUILDER records and RUN on the MI355X -- host tensors in, the C ABI underneath, host tensors out --
against the reference-held literals (tests/golden/reference_held.json), the known answers of the reference's own test
scripts (reference_ops.json) and the CPU oracle; then the whole op-by-op graph of build_opt_graph.py:109-149 spelled
with these kernels, bit for bit equal to oracle_search.

The shim is built against tests/tf_mock (a functional model of TensorFlow's op-kernel API: TensorFlow is not in the
image); tests/test_tf_shim.py checks the registered surface on the CPU."""
import json
import os
import time

import numpy as np
import pytest
import torch

from gpu_util import bits, cuda, require_gpu
from tf_mock import harness as H

pytestmark = pytest.mark.gpu

I32, I64, F16, F32 = np.int32, np.int64, np.float16, np.float32


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    require_gpu()
    H.lib()


@pytest.fixture(scope="module")
def ref_ops(golden_dir):
    with open(os.path.join(golden_dir, "reference_ops.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def held(golden_dir):
    with open(os.path.join(golden_dir, "reference_held.json")) as f:
        return json.load(f)


def a(x, dt):
    return np.asarray(x, dtype=dt, order="C")  # (np.ascontiguousarray would turn a scalar into a vector)


# ---------------------------------------------------------------- GroupGather
@pytest.mark.parametrize("T", [I32, I64])
def test_group_gather_kernels_on_the_reference_cases(ref_ops, held, T):
    k = H.Kernel("GroupGather", T=T)
    for case in ref_ops["group_gather"] + held["group_gather"]:
        r = k(a(case["params_values"], T), a(case["params_row_splits"], I64), a(case["indices_values"], I64),
              a(case["indices_row_splits"], I64))
        if case.get("status"):
            assert r.code == H.INVALID_ARGUMENT, case["name"]
            which = "input0 params" if case["status"] == 1 else "input1 indices"  # GroupGather_kernel.cc:62-67
            assert r.msg == f"Invalid RaggedTensor {which}, code: {case['ragged_code']}", (case["name"], r.msg)
            continue
        assert r.ok, (case["name"], r.msg)
        assert r[0].dtype == T and r[0].tolist() == case["ret_values"], case["name"]
        assert r[1].dtype == I64 and r[1].tolist() == case["ret_row_splits"], case["name"]


@pytest.mark.parametrize("T", [I32, I64])
@pytest.mark.parametrize("unique", [False, True])
def test_group_gather_kernels_vs_oracle(oracle, T, unique):
    k = H.Kernel("GroupGather", T=T, unique=unique)
    rng = np.random.default_rng(20)
    for n_rows, n_groups, max_len in [(3000, 7, 200), (20000, 1, 64), (100, 40, 0)]:
        lens = rng.integers(0, max_len + 1, size=n_rows)
        lens[rng.random(n_rows) < 0.3] = 0
        prs = np.concatenate([[0], np.cumsum(lens)]).astype(I64)
        pv = rng.integers(0, 1 << 12 if unique else 1 << 20, size=int(prs[-1])).astype(T)
        glen = rng.integers(0, 300, size=n_groups)
        irs = np.concatenate([[0], np.cumsum(glen)]).astype(I64)
        iv = rng.integers(0, n_rows, size=int(irs[-1])).astype(I64)
        rc, _, ev, ers = oracle.group_gather(pv.astype(I32), prs, iv, irs, unique=unique)
        r = k(pv, prs, iv, irs)
        assert rc == 0 and r.ok, r.msg
        assert (r[0] == ev).all() and (r[1] == ers).all()
    r = k(a([1, 2], T), a([0, 2], I64), a([1], I64), a([0, 1], I64))  # row index beyond the CSR: UB in the reference
    assert r.code == H.INVALID_ARGUMENT
    if T is I64:  # an id that cannot index a shard
        r = k(a([1 << 40, 2], T), a([0, 2], I64), a([0], I64), a([0, 1], I64))
        assert r.code == H.INVALID_ARGUMENT and "does not fit the int32 ids" in r.msg


# ---------------------------------------------------------------- BitmapRefDifference
@pytest.mark.parametrize("T", [I32, I64])
def test_bitmap_ref_difference_chained_calls_on_one_ref_bitmap(ref_ops, T):
    """bitmap_ref_difference.py:16-29: three calls share ONE Ref bitmap; the kernel mutates the caller's buffer in place
    (mutable_input, bitmap_ops.cc:179) and forwards the Ref to output 2 (:238)."""
    k = H.Kernel("BitmapRefDifference", T=T)
    for case in ref_ops["bitmap_ref_difference"]:
        bitmap = np.zeros(case["bitmap_words"], I32)
        for call in case["calls"]:
            r = k(a(call["values"], T), a(call["row_splits"], I64), H.Ref(bitmap))
            assert r.ok, (case["name"], r.msg)
            assert r[0].dtype == T and r[0].tolist() == call["c_values"], case["name"]
            assert r[1].tolist() == call["c_row_splits"], case["name"]
            out2 = r.outputs[2]
            assert out2.is_ref and out2.forwarded_from == 2 and out2.address == bitmap.ctypes.data
        assert bitmap.tolist() == case["final_flags"], case["name"]


def test_bitmap_ref_difference_kernel_vs_oracle_at_serving_size(oracle):
    k = H.Kernel("BitmapRefDifference", T=I32)
    rng = np.random.default_rng(21)
    n_items = 1_000_000
    words = (n_items + 31) // 32
    got_bm, exp_bm = np.zeros(words, I32), np.zeros(words, I32)
    for n in (128, 4096, 8192, 8192):  # a level's calls accumulate in the bitmap
        v = rng.integers(0, n_items, size=n).astype(I32)
        v[rng.random(n) < 0.2] = v[0]
        rs = a([0, n], I64)
        rc, _, ev, ers = oracle.bitmap_ref_difference(v, rs, exp_bm)
        r = k(v, rs, H.Ref(got_bm))
        assert rc == 0 and r.ok
        assert (r[0] == ev).all() and (r[1] == ers).all() and (got_bm == exp_bm).all()
    r = k(a([1, 2, 3], I32), a([0, 2], I64), H.Ref(got_bm))  # bitmap_ops.cc:182-184
    assert r.code == H.INVALID_ARGUMENT and r.msg == "Invalid RaggedTensor input0 a, code: 3"
    before = got_bm.copy()
    r = k(a([5, 32 * words], I32), a([0, 2], I64), H.Ref(got_bm))  # beyond the bitmap: an OOB write in the reference
    assert r.code == H.INVALID_ARGUMENT and (got_bm == before).all()


# ---------------------------------------------------------------- BitmapInit / BitmapDifference
@pytest.mark.parametrize("T", [I32, I64])
def test_bitmap_init_and_bitmap_difference_kernels(oracle, T):
    rng = np.random.default_rng(22)
    init = H.Kernel("BitmapInit", T=T)
    r = init(a([1, 4, 5, 6, 7], T), np.array(62500, I32))  # bitmap_test.py:19
    exp = np.zeros(62500, I32)
    exp[0] = (1 << 1) | (1 << 4) | (1 << 5) | (1 << 6) | (1 << 7)
    assert r.ok and r[0].dtype == I32 and (r[0] == exp).all()
    idx = rng.integers(0, 32 * 2000, size=2000).astype(T)
    rc, bm = oracle.bitmap_init(idx.astype(I32), 2000)
    r = init(idx, np.array(2000, I32))
    assert rc == 0 and r.ok and (r[0] == bm).all()
    r = init(a([], T), np.array(8, I32))
    assert r.ok and (r[0] == 0).all() and r[0].shape == (8,)
    r = init(a([1, 2, 3], T), np.array(2, I32))  # bitmap_ops.cc:56-57, the reference's text
    assert r.code == H.INVALID_ARGUMENT and r.msg == "require: length >= idx.size() and length >=0 butlength:2idx.size():3"
    r = init(a([64], T), np.array(2, I32))  # bit 64 of a 2-word bitmap: OOB write in the reference
    assert r.code == H.INVALID_ARGUMENT

    diff = H.Kernel("BitmapDifference", T=T)
    flags = rng.integers(-2**31, 2**31, size=200, dtype=np.int64).astype(I32)
    keep = flags.copy()
    nxt = rng.integers(0, 6400, size=3000).astype(T)
    rc, out, fnew = oracle.bitmap_difference(nxt.astype(I32), flags)
    r = diff(nxt, flags)
    assert rc == 0 and r.ok
    assert r[0].dtype == T and (r[0] == out).all() and (r[1] == fnew).all()
    assert (flags == keep).all()  # value semantics: the input bitmap is untouched (bitmap_ops.cc:112-116)
    r = diff(a([], T), flags)
    assert r.ok and r[0].shape == (0,) and (r[1] == flags).all()


# ---------------------------------------------------------------- BloomFilterDifference
@pytest.mark.parametrize("T", [I32, I64])
def test_bloom_filter_difference_kernel(oracle, T):
    """bloom_filter_difference.py:9-28: bucket_size = 10, three chained calls on one Ref filter."""
    k = H.Kernel("BloomFilterDifference", T=T, bucket_size=10)  # bucket takes its default 0
    got, exp = np.zeros(10, I32), np.zeros(10, I32)
    av, ars = [1, 1, 2, 2, 3, 4, 5, 11, 12, 13], [0, 7, 10]
    bv, brs = [4, 5, 6, 7, 7, 8, 10, 1000, 13, 14], [0, 7, 10]
    for v, rs in ((av, ars), (bv, brs), (bv, brs)):
        rc, _, ev, ers = oracle.bloom_filter_difference(v, rs, exp, bucket=0, bucket_size=10)
        r = k(a(v, T), a(rs, I64), H.Ref(got))
        assert rc == 0 and r.ok, r.msg
        assert r[0].tolist() == ev.tolist() and r[1].tolist() == ers.tolist() and (got == exp).all()
        assert r.outputs[2].forwarded_from == 2
    rng = np.random.default_rng(23)
    k2 = H.Kernel("BloomFilterDifference", T=T, bucket=1, bucket_size=4096)
    got, exp = np.zeros(3 * 4096, I32), np.zeros(3 * 4096, I32)
    v = rng.integers(0, 1 << 30, size=5000).astype(T)
    rc, _, ev, ers = oracle.bloom_filter_difference(v.astype(I32), [0, 2000, 5000], exp, bucket=1, bucket_size=4096)
    r = k2(v, a([0, 2000, 5000], I64), H.Ref(got))
    assert rc == 0 and r.ok and (r[0] == ev).all() and (r[1] == ers).all() and (got == exp).all()


# ---------------------------------------------------------------- BlazeTopK / BatchTopKOnRT
def test_blaze_topk_kernel(held, oracle):
    k = H.Kernel("BlazeTopK", T=F32, Tindices=I32)
    for case in held["topk"]:
        if case["kind"] != "literal" or "ZeroRows" in case["name"]:
            continue
        x = a(case["inputs"], F32)
        r = k(x, np.array(case["k"], I32))
        assert r.ok, (case["name"], r.msg)
        assert r[0].shape == np.shape(case["values"]) and r[1].shape == np.shape(case["indices"]), case["name"]  # [..., k]
        assert np.allclose(r[0].reshape(-1), np.asarray(case["values"], F32).reshape(-1), equal_nan=True), case["name"]
        assert r[1].reshape(-1).tolist() == np.asarray(case["indices"]).reshape(-1).tolist(), case["name"]
    rng = np.random.default_rng(24)
    x = rng.standard_normal((3, 5000)).astype(F32)
    r = k(x, np.array(200, I32))
    for row in range(3):
        rc, ev, ei = oracle.blaze_topk(x[row], 200)
        assert (bits(r[0][row]) == bits(ev)).all() and (r[1][row] == ei).all()
    r = k(x, np.array(5001, I32))  # BlazeTopK_kernel.cc:47-48
    assert r.code == H.INVALID_ARGUMENT and r.msg == "require: 0 <= k <= input_len, but5001 > 5000"
    r = k(x, np.array(0, I32))
    assert r.ok and r[0].shape == (3, 0)


def test_batch_topk_on_rt_kernel(ref_ops, oracle):
    for case in ref_ops["batch_topk_on_rt"]:
        k = H.Kernel("BatchTopKOnRT", T=F32, ascending=bool(case["ascending"]))
        r = k(a(case["values"], F32), a(case["row_splits"], I64), a(case["k"], I64))
        assert r.ok, (case["name"], r.msg)
        assert r[0].tolist() == case["values_out"] and r[1].tolist() == case["idx_out"], case["name"]
        assert r[1].dtype == I64 and r[2].tolist() == case["row_splits_out"], case["name"]
    rng = np.random.default_rng(25)
    lens = rng.integers(0, 900, size=12)
    rs = np.concatenate([[0], np.cumsum(lens)]).astype(I64)
    v = rng.permutation(int(rs[-1])).astype(F32)  # distinct values: the answer is determined
    ks = rng.integers(0, 300, size=12).astype(I64)
    for asc in (False, True):
        k = H.Kernel("BatchTopKOnRT", T=F32, ascending=asc)
        rc, ev, ei, ers = oracle.batch_topk_on_rt(v, rs, ks, ascending=asc)
        r = k(v, rs, ks)
        assert rc == 0 and r.ok and (r[0] == ev).all() and (r[1] == ei).all() and (r[2] == ers).all()
    k = H.Kernel("BatchTopKOnRT", T=F32)
    r = k(v, rs, a([1, 2], I64))  # BatchTopKOnRT_kernel.cc:104-106
    assert r.code == H.INVALID_ARGUMENT and r.msg == "Size of k vector does NOT match number of groups: 2!=12"
    r = k(v, a([0, 5], I64), np.array(3, I64))  # :76-78
    assert r.code == H.INVALID_ARGUMENT and r.msg == "Invalid RaggedTensor input, code: 3"


# ---------------------------------------------------------------- HugeConst + the registry
def test_huge_const_kernel_and_resident_constants(tmp_path, oracle):
    """HugeConst loads once (constructor), returns the SAME host tensor every step (zero-copy set_output,
    huge_const_op.cc:184-226), and a GroupGather fed that tensor reads the HBM copy: no byte of it crosses PCIe again."""
    rng = np.random.default_rng(26)
    n_rows = 50_000
    lens = rng.integers(0, 65, size=n_rows)
    prs = np.concatenate([[0], np.cumsum(lens)]).astype(I64)
    pv = rng.integers(0, n_rows, size=int(prs[-1])).astype(I32)
    np.save(tmp_path / "v.npy", pv)
    np.save(tmp_path / "rs.npy", prs)
    hv = H.Kernel("HugeConst", dtype=I32, shape=H.Shape(len(pv)), path=str(tmp_path / "v.npy"))
    hrs = H.Kernel("HugeConst", dtype=I64, shape=H.Shape(len(prs)), path=str(tmp_path / "rs.npy"))
    assert not H.lib().tfm_kernel_is_expensive(hv.handle)
    r1, r2 = hv(), hv()
    assert r1.ok and (r1[0] == pv).all() and r1.outputs[0].address == r2.outputs[0].address  # one tensor, shared
    rrs = hrs()
    assert (rrs[0] == prs).all()

    def borrowed(out, like):  # the kernel's own buffer as the next node's input, as the executor hands it on
        import ctypes as C
        buf = (C.c_char * like.nbytes).from_address(out.address)
        return np.frombuffer(buf, dtype=like.dtype)

    gg = H.Kernel("GroupGather", T=I32)
    iv = rng.integers(0, n_rows, size=128).astype(I64)
    irs = a([0, 128], I64)
    rc, _, ev, ers = oracle.group_gather(pv, prs, iv, irs)
    H.stats(reset=True)
    r = gg(borrowed(r1.outputs[0], pv), borrowed(rrs.outputs[0], prs), iv, irs)
    s = H.stats()
    assert r.ok and (r[0] == ev).all() and (r[1] == ers).all()
    assert s["registry_hits"] == 2 and s["registry_misses"] == 0
    assert s["h2d_bytes"] == iv.nbytes + irs.nbytes, s  # the frontier only: neither constant was uploaded
    H.stats(reset=True)
    r = gg(pv, prs, iv, irs)  # the same VALUES in other buffers are not the constants: staged per call
    s = H.stats()
    assert r.ok and (r[0] == ev).all() and s["registry_hits"] == 0 and s["h2d_bytes"] >= pv.nbytes + prs.nbytes
    # validation of the file against the attrs (huge_const_op.cc:96-98, 111-147)
    with pytest.raises(H.KernelError) as e:
        H.Kernel("HugeConst", dtype=I64, shape=H.Shape(len(pv)), path=str(tmp_path / "v.npy"))
    assert e.value.code == H.INVALID_ARGUMENT
    with pytest.raises(H.KernelError) as e:
        H.Kernel("HugeConst", dtype=I32, shape=H.Shape(len(pv) + 1), path=str(tmp_path / "v.npy"))
    assert e.value.code == H.INVALID_ARGUMENT
    with pytest.raises(H.KernelError) as e:
        H.Kernel("HugeConst", dtype=I32, shape=H.Shape(3), path=str(tmp_path / "missing.npy"))
    assert e.value.code == H.NOT_FOUND
    # a destroyed HugeConst leaves the registry: its (freed) buffer address can never be served again
    addr = r1.outputs[0].address
    del r1, r2
    hv.close()
    H.stats(reset=True)
    r = gg(pv, prs, iv, irs)
    assert r.ok and H.stats()["registry_hits"] == 0 and addr


# ---------------------------------------------------------------- BlazeXlaOp
def _blaze(graph_def, options="", **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:  # BLAZE_THREADS_NUM / DENSE_MAX_WAITING_COUNT are read in the constructor (blaze_xla_kernel.cc:87-101,136-140)
        return H.Kernel("BlazeXlaOp", InT=[F16, F16], OutT=[F32],
                        input_names=["inference_feed_inputs/user_seq_emb", "inference_feed_inputs/item_emb"],  # constant.py:9-11
                        output_names=["inference_fetch_outputs/logits"], graph_def=graph_def, blaze_option_path=options)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _user_and_rows(rng, L, E, n, d):
    u = (rng.standard_normal((1, L, E)) / 8).astype(F16)
    u[0, L - 9:] = 0
    rows = (rng.standard_normal((n, d)) / 8).astype(F16)
    return u, rows


@pytest.mark.parametrize("kind", ["l2", "mlp"])
def test_blaze_xla_op_l2_and_mlp_directories(oracle, tmp_path, kind):
    from nann_amd import ops, synth
    d, L, n = 128, 50, 3000
    rng = np.random.default_rng(27)
    u, rows = _user_and_rows(rng, L, d, n, d)
    w = synth.make_mlp_weights(d) if kind == "mlp" else None
    ops.save_scorer_dir(str(tmp_path / kind), kind, w, precision="exact" if kind == "mlp" else None)
    k = _blaze(str(tmp_path / kind))
    assert k.is_async
    r = k(u, rows)
    assert r.ok, r.msg
    assert r[0].shape == (n, 1) and r[0].dtype == F32  # model.py:226-227
    q = oracle.user_seq_mean(u[0])
    rc, exp = oracle.score_rows(oracle.Scorer(kind, d, oracle.EMB_F16, w), q, rows)
    assert rc == 0 and (bits(r[0].ravel()) == bits(exp)).all()  # L2 and the exact-f32 MLP: bit for bit
    done_calls, other_thread, _ = r.async_info
    assert done_calls == 1 and other_thread == 1  # done() came from a worker, not from ComputeAsync's caller


@pytest.mark.parametrize("folded", [True, False])
def test_blaze_xla_op_attention_model_from_directory_and_frozen_graphdef(oracle, tmp_path, folded):
    from nann_amd import frozen_graph, ops, synth
    d, L, n = 128, 50, 900
    rng = np.random.default_rng(28)
    u, rows = _user_and_rows(rng, L, 64, n, d)
    w = synth.make_attn_weights(d, 64)
    ops.save_scorer_dir(str(tmp_path / "dir"), "attention", w, precision="exact")
    pb = tmp_path / "frozen_graph.pb"
    frozen_graph.write_attention_graph(str(pb), w, seq_len=L, folded=folded)
    (tmp_path / "frozen_graph.pb.precision").write_text("exact\n")
    rc, exp = oracle.attn_score_rows(oracle.AttnModel(d, 64, L, oracle.EMB_F16, w), u[0].astype(F32), rows)
    r_dir = _blaze(str(tmp_path / "dir"))(u, rows)
    r_pb = _blaze(str(pb))(u, rows)
    assert rc == 0 and r_dir.ok and r_pb.ok, (r_dir.msg, r_pb.msg)
    assert (bits(r_dir[0]) == bits(r_pb[0])).all()  # the same weights, from arrays and from the GraphDef's Const nodes
    assert np.abs(r_pb[0].ravel() - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max())


def test_blaze_xla_op_errors_of_the_reference(tmp_path):
    from nann_amd import ops
    d, L = 64, 50
    rng = np.random.default_rng(29)
    ops.save_scorer_dir(str(tmp_path / "l2"), "l2")
    k = _blaze(str(tmp_path / "l2"))
    u, rows = _user_and_rows(rng, L, d, 10, d)
    assert k(u, rows).ok
    r = k(u, np.zeros((0, d), F16))  # PadToStatic, blaze_xla_predictor.cc:259-263
    assert r.code == H.INTERNAL and r.msg == "Error when getting input address or size"
    r = k(u, np.zeros((10, 2 * d), F16))  # another d than the model was loaded for
    assert r.code == H.INVALID_ARGUMENT and "item_emb must be [n, 64]" in r.msg
    r = k(u[:, :40], rows)
    assert r.code == H.INVALID_ARGUMENT and "user_seq_emb must be [1, 50, 64]" in r.msg
    r = k(u.astype(F32), rows)  # the node says InT = [float16, float16]
    assert r.code == H.INVALID_ARGUMENT
    with pytest.raises(H.KernelError) as e:  # blaze_xla_kernel.cc:160-163
        _blaze(str(tmp_path / "l2"), options=str(tmp_path / "no_such.conf"))
    assert e.value.code == H.INTERNAL and e.value.msg == f"parse proto from {tmp_path / 'no_such.conf'} failed"
    r = _blaze(str(tmp_path / "nowhere"))(u, rows)  # the model is loaded with the first request
    assert r.code == H.NOT_FOUND
    (tmp_path / "junk.pb").write_bytes(b"\x00\x01\x02 not a graph")
    r = _blaze(str(tmp_path / "junk.pb"))(u, rows)
    assert not r.ok
    skip = _blaze(str(tmp_path / "l2"), options="run_mode: SKIP")  # blaze_xla_kernel.cc:183-188
    r = skip(u, rows)
    assert r.ok and r[0].shape == (10, 2) and H.stats()["blaze_runs"] >= 1


def test_blaze_xla_op_admission_control(tmp_path):
    """blaze_xla_kernel.cc:221-258 with BLAZE_THREADS_NUM = 1: one request runs, DENSE_MAX_WAITING_COUNT wait, the next is
    refused at once with the reference's text; with wait_ms the bound is a deadline."""
    from nann_amd import ops
    d, L, n = 128, 50, 1_500_000  # ~384 MB of rows: tens of ms of upload keep the single worker busy
    rng = np.random.default_rng(30)
    ops.save_scorer_dir(str(tmp_path / "l2"), "l2")
    u, small = _user_and_rows(rng, L, d, 64, d)
    big = np.zeros((n, d), F16)
    big[::1000] = small[0]

    k = _blaze(str(tmp_path / "l2"), BLAZE_THREADS_NUM=1, DENSE_MAX_WAITING_COUNT=2)
    assert k(u, small).ok  # loads the model, warms the slot
    H.stats(reset=True)
    t0 = time.perf_counter()
    calls = [k.start(u, big)] + [k.start(u, small) for _ in range(4)]
    t_submit = time.perf_counter() - t0
    res = [c.wait() for c in calls]
    assert res[0].ok and res[1].ok and res[2].ok  # ran; waited; waited
    for r in res[3:]:  # the waiting pool held 2
        assert r.code == H.INTERNAL and r.msg == "waiting pool is full 2", r.msg
        assert r.async_info[2] == 1  # refused inline: done() before ComputeAsync returned (OP_REQUIRES_ASYNC in Schedule)
    assert res[0].async_info == (1, 1, 0) and res[1].async_info[:2] == (1, 1)
    assert (bits(res[1][0]) == bits(res[2][0])).all()
    assert H.stats()["blaze_rejected"] == 2 and H.stats()["blaze_runs"] == 3
    # ComputeAsync never blocked its caller: five submissions took far less than the big request alone
    assert t_submit < 0.05, t_submit

    kw = _blaze(str(tmp_path / "l2"), options="wait_ms: 2", BLAZE_THREADS_NUM=1, DENSE_MAX_WAITING_COUNT=0)
    assert kw(u, small).ok
    calls = [kw.start(u, big), kw.start(u, small), kw.start(u, small)]
    res = [c.wait() for c in calls]
    assert res[0].ok
    for r in res[1:]:  # still waiting after 2 ms: :227-229 (Internal) or, had a worker picked it up late, :243-248
        assert r.code in (H.INTERNAL, H.DEADLINE_EXCEEDED) and r.msg.startswith("blaze wait too long "), r.msg
        assert int(r.msg.split()[-1]) > 2_000_000  # nanoseconds waited
    assert kw(u, small).ok  # the kernel keeps serving

    k2 = _blaze(str(tmp_path / "l2"), BLAZE_THREADS_NUM=2)  # two workers: two streams, both requests run
    assert k2(u, small).ok
    a_, b_ = k2.start(u, big), k2.start(u, big)
    ra, rb = a_.wait(), b_.wait()
    assert ra.ok and rb.ok and (bits(ra[0]) == bits(rb[0])).all()


# ---------------------------------------------------------------- NannHnswSearch
@pytest.fixture(scope="module")
def small_index(tmp_path_factory):
    from gpu_util import queries_for, synth_index
    from nann_amd import synth
    g, oix, dix = synth_index(20000, 64, 64, seed=77)
    out = tmp_path_factory.mktemp("index")
    synth.save_index(g, str(out))
    return g, oix, str(out)


@pytest.mark.parametrize("batch", [1, 300])
def test_nann_hnsw_search_kernel(oracle, small_index, batch):
    from gpu_util import queries_for
    g, oix, index_dir = small_index
    d, L = 64, 50
    k = H.Kernel("NannHnswSearch", index_dir=index_dir, item_embs_dir=index_dir)  # seq_len = 50, scorer_dir = '' (L2)
    comm_seq = queries_for(g, batch, seed=31).astype(F16)  # [B, L, d]
    topn = a([64, 64, 64, 64, 64, 200], I32)
    r = k(comm_seq.reshape(batch, L * d), topn)
    assert r.ok, r.msg
    assert r[0].shape == (batch, 200) and r[0].dtype == I64
    q = np.stack([oracle.user_seq_mean(comm_seq[b]) for b in range(batch)])
    st, ids, _, _, _ = oracle.search_batch(oix, oracle.Scorer("l2", d, oracle.EMB_F16), q, topn)
    assert (st == 0).all() and (r[0] == ids).all()
    r = k(comm_seq.reshape(batch, L * d), a([64, 64, 64, 64, 64], I32))
    assert r.code == H.INVALID_ARGUMENT and r.msg == "level_topn must have 6 entries"
    r = k(comm_seq.reshape(batch, L * d), a([64, 64, 64, 64, 64, 300], I32))  # 300 of a pool of 256: TopKV2's k > n fails the request
    assert r.code == H.INVALID_ARGUMENT and "failed with nann_status 4" in r.msg


def test_nann_hnsw_search_kernel_with_an_mlp_scorer(oracle, small_index, tmp_path):
    from gpu_util import queries_for
    from nann_amd import ops, synth
    g, oix, index_dir = small_index
    d, L, batch = 64, 50, 40
    w = synth.make_mlp_weights(d)
    ops.save_scorer_dir(str(tmp_path / "mlp"), "mlp", w, precision="exact")
    k = H.Kernel("NannHnswSearch", index_dir=index_dir, item_embs_dir=index_dir, scorer_dir=str(tmp_path / "mlp"))
    comm_seq = queries_for(g, batch, seed=32).astype(F16)
    topn = a([64, 64, 64, 64, 64, 200], I32)
    r = k(comm_seq.reshape(batch, L * d), topn)
    q = np.stack([oracle.user_seq_mean(comm_seq[b]) for b in range(batch)])
    st, ids, _, _, _ = oracle.search_batch(oix, oracle.Scorer("mlp", d, oracle.EMB_F16, w), q, topn)
    assert r.ok and (st == 0).all() and (r[0] == ids).all()


# ---------------------------------------------------------------- the op-by-op serving graph through the shim
def test_build_opt_graph_chain_through_the_shim_kernels(oracle, small_index, tmp_path):
    """build_opt_graph.py:69-149 node for node: HugeConst x 7 -> [GatherV2 -> BlazeXlaOp -> TopKV2] with GroupGather /
    TemporaryVariable + Assign / BitmapRefDifference / ConcatV2 in between.  The four custom ops are the SHIM's kernels,
    fed host tensors exactly as a /CPU:0-pinned graph feeds them (:82,110); the stock ops are nann_gather_rows /
    nann_topk.  ids and scores equal oracle_search's bit for bit."""
    from gpu_util import queries_for
    from nann_amd import ops
    g, oix, index_dir = small_index
    d, L = 64, 50
    n = len(g["item_ids"])
    ops.save_scorer_dir(str(tmp_path / "l2"), "l2")

    def huge(name, dt, arr):  # huge_constant(path, dtype) of model_util.py:107-121 rewrites the file to `dt` first
        p = tmp_path / (name + ".npy")
        np.save(p, np.asarray(arr).astype(dt))
        kk = H.Kernel("HugeConst", dtype=dt, shape=H.Shape(*np.shape(arr)), path=str(p))
        r = kk()
        assert r.ok
        import ctypes as C
        view = np.frombuffer((C.c_char * r[0].nbytes).from_address(r.outputs[0].address), dtype=dt).reshape(np.shape(arr))
        return kk, r, view

    consts = {}
    for name, dt, arr in (("item_embs", F16, g["item_embs"]), ("item_ids", I64, g["item_ids"]),
                          ("nb0_values", I32, g["nb_values"][0]), ("nb0_row_splits", I64, g["nb_row_splits"][0]),
                          ("nb1_values", I32, g["nb_values"][1]), ("nb1_row_splits", I64, g["nb_row_splits"][1])):
        consts[name] = huge(name, dt, arr)
    embs, item_ids = consts["item_embs"][2], consts["item_ids"][2]
    enter_points = np.asarray(g["enter_points"], I32)  # baked as a Const (:70)
    gg = H.Kernel("GroupGather", T=I32)
    diff = H.Kernel("BitmapRefDifference", T=I32)
    blaze = _blaze(str(tmp_path / "l2"))
    embs_dev = cuda(g["item_embs"])

    def ragged_gather(level, idx):  # :39-49
        r = gg(consts[f"nb{level}_values"][2], consts[f"nb{level}_row_splits"][2], idx.astype(I64), a([0, len(idx)], I64))
        assert r.ok, r.msg
        return r[0]

    def set_difference(x, flags):  # :33-36
        r = diff(x, a([0, len(x)], I64), H.Ref(flags))
        assert r.ok, r.msg
        return r[0]

    def forward(user, idx):  # :91-107: GatherV2 -> BlazeXlaOp -> Squeeze
        rows = ops.gather(embs_dev, cuda(idx)).cpu().numpy()
        r = blaze(user, rows)
        assert r.ok, r.msg
        return r[0].reshape(-1)

    def top_k(ids, scores, k):  # :52-66
        v, i = ops.top_k(cuda(scores), k)
        i = i.cpu().numpy()
        return ids[i], v.cpu().numpy()

    topn = [64, 64, 64, 64, 64, 200]
    comm_seq = queries_for(g, 3, seed=33).astype(F16)
    H.stats(reset=True)
    for b in range(3):
        user = comm_seq[b][None]  # [1, L, d]
        s = forward(user, enter_points)                                   # :111
        R, sR = top_k(enter_points, s, topn[0])                           # :112
        C1 = ragged_gather(1, R)                                          # :116
        flags = np.zeros((n + 31) // 32, I32)                             # TemporaryVariable + Assign(zeros), :115-118
        R = set_difference(R, flags)                                      # :119-120
        C1 = set_difference(C1, flags)                                    # :121-122
        sC = forward(user, C1)                                            # :124
        P, sP = top_k(np.concatenate([R, C1]), np.concatenate([sR, sC]), topn[1])   # :125-127
        B = P
        flags[:] = 0                                                      # re-Assign, :131
        B = set_difference(B, flags)                                      # :132-133
        for i in range(3):                                                # :135
            C0 = ragged_gather(0, B)                                      # :136
            C0 = set_difference(C0, flags)                                # :137
            sC = forward(user, C0)                                        # :138
            B, sB = top_k(C0, sC, topn[2 + i])                            # :139
            P, sP = np.concatenate([P, B]), np.concatenate([sP, sB])      # :140-141
        P, sP = top_k(P, sP, topn[5])                                     # :143
        out = item_ids[P]                                                 # :144
        q = oracle.user_seq_mean(comm_seq[b])
        st, ids, sc, idx, _ = oracle.search(oix, oracle.Scorer("l2", d, oracle.EMB_F16), q, topn)
        assert st == 0 and (P == idx).all() and (out == ids).all() and (bits(sP) == bits(sc)).all()
    s = H.stats()
    assert s["registry_hits"] == 3 * 2 * 4 and s["blaze_runs"] == 3 * 5  # every CSR read was resident; 5 scoring calls per query
