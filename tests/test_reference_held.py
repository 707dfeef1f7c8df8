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
    for key in ("topk", "gather", "group_gather"):
        for c in HELD[key]:
            assert c["kind"] in ("literal", "recipe", "error") and c["src"], c["name"]


# ------------------------------------------------------------------ GPU: the HIP ops through the C ABI
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
