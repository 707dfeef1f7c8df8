"""Reference-HELD vectors (tests/golden/reference_held.json): expected values the reference's own
tests assert -- stock TopKV2 / GatherV2 tests shipped in the fork (python/kernel_tests/
topk_op_test.py, gather_op_test.py) and the GroupGather docstring example.  The oracle is pinned to
them on the CPU; the HIP ops must reproduce the same values through the C ABI (-m gpu).
"""
import json
import os

import numpy as np
import pytest


def _load():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_held.json")
    with open(here) as f:
        return json.load(f)


HELD = _load()


def _f(x):
    """json -> float32 array ("nan" strings allowed)."""
    a = np.array(x, dtype=object)
    return np.vectorize(lambda v: np.float32("nan") if v == "nan" else np.float32(v), otypes=[np.float32])(a) \
        if a.size else np.zeros(a.shape, np.float32)


def topk_case_arrays(case):
    """(inputs f32 [rows, n] or [n], expected values, expected indices)"""
    if "linspace" in case:
        lo, hi, total, dt = case["linspace"]
        inp = np.linspace(lo, hi, total, dtype=np.dtype(dt))[np.asarray(case["perm"])].astype(np.float32)
        inp = inp.reshape(case["rows"], -1)
        idx = np.asarray(case["indices"], np.int32)
        vals = np.take_along_axis(inp, idx.astype(np.int64), 1)
        return inp, vals, idx
    inp = _f(case["inputs"])
    if "inputs_shape" in case:
        inp = inp.reshape(case["inputs_shape"])
    vals, idx = _f(case["values"]), np.asarray(case["indices"], np.int32)
    if "out_shape" in case:
        vals, idx = vals.reshape(case["out_shape"]), idx.reshape(case["out_shape"])
    return inp, vals, idx


TOPK_OK = [c for c in HELD["topk"] if c["kind"] != "error"]
TOPK_ERR = [c for c in HELD["topk"] if c["kind"] == "error"]
GATHER_OK = [c for c in HELD["gather"] if c["kind"] != "error"]
GATHER_ERR = [c for c in HELD["gather"] if c["kind"] == "error"]
_id = lambda c: c["name"]


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


# ------------------------------------------------------------------ CPU: the oracle
@pytest.mark.parametrize("case", TOPK_OK, ids=_id)
def test_oracle_topk_reference_held(oracle, case):
    inp, vals, idx = topk_case_arrays(case)
    rows = inp.reshape(-1, inp.shape[-1]) if inp.ndim > 1 else inp[None, :]
    ev, ei = vals.reshape(rows.shape[0], case["k"]), idx.reshape(rows.shape[0], case["k"])
    for r in range(rows.shape[0]):
        rc, v, i = oracle.topk(rows[r], case["k"])
        assert rc == 0, case["name"]
        assert i.tolist() == ei[r].tolist(), case["name"]
        assert _same(v, ev[r]), case["name"]


@pytest.mark.parametrize("case", TOPK_ERR, ids=_id)
def test_oracle_topk_reference_errors(oracle, case):
    rc, _, _ = oracle.topk(_f(case["inputs"])[0], case["k"])
    assert rc == (oracle.ERR_BAD_ARGUMENT if case["k"] < 0 else oracle.ERR_TOPK_K_GT_N)


@pytest.mark.parametrize("case", GATHER_OK, ids=_id)
def test_oracle_gather_reference_held(oracle, case):
    params = np.asarray(case["params"], dtype=case["dtype"])
    idx = np.asarray(case["indices"], np.int32)
    rc, out, _ = oracle.gather_rows(params, idx.reshape(-1))
    assert rc == 0
    exp = np.asarray(case["expected"], dtype=case["dtype"])
    assert out.reshape(exp.shape).tolist() == exp.tolist(), case["name"]


@pytest.mark.parametrize("case", GATHER_ERR, ids=_id)
def test_oracle_gather_reference_errors(oracle, case):
    params = np.asarray(case["params"], dtype=case["dtype"])
    rc, _, bad = oracle.gather_rows(params, np.asarray(case["indices"], np.int32).reshape(-1))
    assert rc == oracle.ERR_INDEX_OUT_OF_RANGE and bad == case["bad_i"]


def test_oracle_group_gather_reference_held(oracle):
    for case in HELD["group_gather"]:
        rc, _, v, rs = oracle.group_gather(case["params_values"], case["params_row_splits"],
                                           case["indices_values"], case["indices_row_splits"])
        assert rc == 0 and v.tolist() == case["ret_values"] and rs.tolist() == case["ret_row_splits"]


def test_provenance_tags():
    """every known-answer case says whether its expected output is held by the reference or was
    produced at survey time ('tag every case')."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_ops.json")
    with open(here) as f:
        ops_json = json.load(f)
    for key, cases in ops_json.items():
        if key.startswith("_"):
            continue
        for c in cases:
            assert c.get("provenance") in ("reference-held", "survey-derived", "repo-derived"), (key, c["name"])
    for key in ("topk", "gather", "group_gather", "fingerprint64"):
        for c in HELD[key]:
            assert c["kind"] in ("literal", "recipe", "error") and c["src"], c["name"]


def test_oracle_fingerprint64_reference_held(oracle):
    """tensorflow::Fingerprint64 = FarmHash (un-vendored dependency): the oracle's restatement against the
    fork's own known answers (core/platform/fingerprint_test.cc:27-28)."""
    for case in HELD["fingerprint64"]:
        assert oracle.fingerprint64(case["input"]) == case["expected"], case["name"]


def _py_fingerprint64(b):
    """independent restatement of farmhashna::Hash64 for len <= 16 (pure Python ints)"""
    M = (1 << 64) - 1
    k0, k2 = 0xc3a5c85c97cb3127, 0x9ae16a3b2f90404f
    rot = lambda v, s: ((v >> s) | (v << (64 - s))) & M
    f = lambda i, n_: int.from_bytes(b[i:i + n_], "little")

    def l16(u, v, mul):
        a = ((u ^ v) * mul) & M; a ^= a >> 47
        c = ((v ^ a) * mul) & M; c ^= c >> 47
        return (c * mul) & M
    n = len(b)
    if n >= 8:
        mul = (k2 + 2 * n) & M; a = (f(0, 8) + k2) & M; c = f(n - 8, 8)
        return l16((rot(c, 37) * mul + a) & M, ((rot(a, 25) + c) * mul) & M, mul)
    if n >= 4:
        return l16((n + (f(0, 4) << 3)) & M, f(n - 4, 4), (k2 + 2 * n) & M)
    y = (b[0] + (b[n >> 1] << 8)) & 0xffffffff; z = (n + (b[n - 1] << 2)) & 0xffffffff
    v = ((y * k2) & M) ^ ((z * k0) & M); v ^= v >> 47
    return (v * k2) & M


def _py_bloom(values, row_splits, flags, bucket, bucket_size):
    """BloomFilterDifference::Compute (bitmap_ops.cc:341-365) in plain Python, for the oracle to agree with"""
    def prime_below(num):
        n_ = num
        while True:
            if all(n_ % i for i in range(2, int(n_ ** 0.5 + 1e-6) + 1)):
                return n_
            n_ -= 1
    primes = [prime_below(m * bucket_size * 32) for m in (29, 47, 67, 83)]
    out, rs = [], [0]
    for g in range(len(row_splits) - 1):
        for j in range(row_splits[g], row_splits[g + 1]):
            raw = _py_fingerprint64(str(int(values[j])).encode())
            if bucket > 0:
                raw %= bucket
            miss = 0
            for l, mult in enumerate((1, 3, 5, 7)):
                pos = (((raw * mult) & ((1 << 64) - 1)) % primes[l]) % (bucket_size * 32)
                if not (int(flags[pos >> 5]) & 0xffffffff) >> (pos & 31) & 1:
                    miss += 1
                    flags[pos >> 5] = np.int32(np.uint32((int(flags[pos >> 5]) & 0xffffffff) | (1 << (pos & 31))))
            if miss:
                out.append(int(values[j]))
        rs.append(len(out))
    return out, rs


BLOOM_CASES = [  # (values, row_splits, bucket, bucket_size); the first two calls are bloom_filter_difference.py:19-20
    ([1, 1, 2, 2, 3, 4, 5, 11, 12, 13], [0, 7, 10], 0, 10),
    ([4, 5, 6, 7, 7, 8, 10, 1000, 13, 14], [0, 7, 10], 0, 10),
    (list(range(0, 4000, 3)) + [7, 7, 123456789, 2147483647, 0], [0, 500, 1339], 1000003, 64),
]


def test_oracle_bloom_filter_difference_matches_python_restatement(oracle):
    """a8: BloomFilterDifference (approximate visited filter).  The reference script
    (UO/bitmap_op/bloom_filter_difference.py) prints and asserts nothing, so the oracle is checked against an
    independent pure-Python restatement of bitmap_ops.cc:341-365 on the script's inputs and a larger case
    (short / long decimal strings, a non-zero first-hash bucket, repeated ids); chained calls share the filter."""
    for bucket, bs in ((0, 10), (1000003, 64)):
        f_o, f_p = np.zeros(bs, np.int32), np.zeros(bs, np.int32)
        for values, rs, b, s_ in BLOOM_CASES:
            if (b, s_) != (bucket, bs):
                continue
            for _ in range(2):  # second call: everything is "visited" now
                rc, _, got, got_rs = oracle.bloom_filter_difference(values, rs, f_o, bucket, bs)
                exp, exp_rs = _py_bloom(values, rs, f_p, bucket, bs)
                assert rc == 0 and got.tolist() == exp and got_rs.tolist() == exp_rs
                assert (f_o == f_p).all()
    rc, _, got, got_rs = oracle.bloom_filter_difference([], [0], np.zeros(4, np.int32), 0, 4)  # void input
    assert rc == 0 and got.tolist() == [] and got_rs.tolist() == [0]
    rc, code, _, _ = oracle.bloom_filter_difference([1, 2], [0, 1], np.zeros(4, np.int32), 0, 4)
    assert (rc, code) == (oracle.ERR_INVALID_RAGGED_INPUT, 3)


def test_oracle_blaze_topk_is_a_valid_answer(oracle):
    """a8: BlazeTopK returns the k largest values sorted by value; ties in unspecified order."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal(3000).astype(np.float32)
    rc, v, i = oracle.blaze_topk(x, 40)
    assert rc == 0 and (i == np.argsort(-x, kind="stable")[:40]).all() and (v == x[i]).all()
    assert oracle.blaze_topk(x, 3001)[0] == oracle.ERR_BAD_ARGUMENT


# ------------------------------------------------------------------ GPU: the HIP ops through the C ABI
@pytest.mark.gpu
def test_hip_bloom_filter_difference(oracle):
    import torch
    from nann_amd import ops
    rng = np.random.default_rng(12)
    cases = list(BLOOM_CASES) + [(rng.integers(0, 1 << 20, 9000).tolist(), [0, 100, 4000, 9000], 0, 40000),
                                 (rng.integers(-50, 50, 300).tolist(), [0, 300], 97, 3)]
    for values, rs, bucket, bs in cases:
        f_o = np.zeros(bs + 2, np.int32)
        f_d = torch.zeros(bs + 2, dtype=torch.int32, device="cuda")
        for _ in range(2):
            rc, _, exp, exp_rs = oracle.bloom_filter_difference(values, rs, f_o, bucket, bs)
            got, got_rs, f = ops.bloom_filter_difference(values, rs, f_d, bucket=bucket, bucket_size=bs)
            assert rc == 0 and f is f_d
            assert got.cpu().tolist() == exp.tolist() and got_rs.cpu().tolist() == exp_rs.tolist()
            assert (f_d.cpu().numpy() == f_o).all()
    got, got_rs, _ = ops.bloom_filter_difference([], [0], torch.zeros(4, dtype=torch.int32, device="cuda"), bucket_size=4)
    assert got.numel() == 0 and got_rs.cpu().tolist() == [0]
    with pytest.raises(ops.InvalidArgumentError):
        ops.bloom_filter_difference([1, 2], [0, 1], torch.zeros(4, dtype=torch.int32, device="cuda"), bucket_size=4)


@pytest.mark.gpu
def test_hip_blaze_top_k():
    import torch
    from nann_amd import ops
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3, 5000)).astype(np.float32)
    v, i = ops.blaze_top_k(torch.as_tensor(x).cuda(), 100)
    order = np.argsort(-x, axis=1, kind="stable")[:, :100]
    assert (i.cpu().numpy() == order).all() and (v.cpu().numpy() == np.take_along_axis(x, order, 1)).all()
    with pytest.raises(ops.InvalidArgumentError):
        ops.blaze_top_k(torch.as_tensor(x).cuda(), 5001)


@pytest.mark.gpu
@pytest.mark.parametrize("case", TOPK_OK, ids=_id)
def test_hip_topk_reference_held(case):
    import torch
    from nann_amd import ops
    inp, vals, idx = topk_case_arrays(case)
    v, i = ops.top_k(torch.as_tensor(inp).cuda(), case["k"])
    assert tuple(v.shape) == vals.shape and tuple(i.shape) == idx.shape, case["name"]
    assert i.cpu().numpy().tolist() == idx.tolist(), case["name"]
    assert _same(v.cpu().numpy(), vals), case["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", TOPK_ERR, ids=_id)
def test_hip_topk_reference_errors(case):
    import torch
    from nann_amd import ops
    with pytest.raises(ops.InvalidArgumentError) as e:
        ops.top_k(torch.as_tensor(_f(case["inputs"])).cuda(), case["k"])
    assert case["error"] in str(e.value), case["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GATHER_OK, ids=_id)
def test_hip_gather_reference_held(case):
    import torch
    from nann_amd import ops
    params = torch.as_tensor(np.asarray(case["params"], dtype=case["dtype"])).cuda()
    idx = np.asarray(case["indices"], np.int32)
    out = ops.gather(params, idx.reshape(-1))
    exp = np.asarray(case["expected"], dtype=case["dtype"])
    assert out.cpu().numpy().reshape(exp.shape).tolist() == exp.tolist(), case["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GATHER_ERR, ids=_id)
def test_hip_gather_reference_errors(case):
    import torch
    from nann_amd import ops
    params = torch.as_tensor(np.asarray(case["params"], dtype=case["dtype"])).cuda()
    with pytest.raises(ops.InvalidArgumentError) as e:
        ops.gather(params, np.asarray(case["indices"], np.int32).reshape(-1))
    # gather_op.cc:170-175 words it "indices[0,0] = 7 is not in [0, 2)": flat position + range here
    assert case["error"] in str(e.value) and "indices[%d]" % case["bad_i"] in str(e.value)


@pytest.mark.gpu
def test_hip_group_gather_reference_held():
    from nann_amd import ops
    for case in HELD["group_gather"]:
        v, rs = ops.group_gather(case["params_values"], case["params_row_splits"], case["indices_values"],
                                 case["indices_row_splits"])
        assert v.cpu().tolist() == case["ret_values"] and rs.cpu().tolist() == case["ret_row_splits"]
