"""Writes tests/golden/reference_held.json: input -> output pairs that the REFERENCE'S OWN
TESTS hold as literals (or as an explicit numpy recipe), transcribed as data.  These pin
the oracle (tests/test_reference_held.py, CPU) and the HIP ops (-m gpu) to values the
reference asserts, independently of anything this repo computed.

Sources (relative to /root/reference/tensorflow/tensorflow/):
  python/kernel_tests/topk_op_test.py    TopKV2 literals :96-102, :168-171, :180-192;
                                         errors :195-209; numpy-recipe cases :104-166
  python/kernel_tests/gather_op_test.py  GatherV2 axis-0 literals :64-76, :92-108,
                                         :257-272 (batch_dims=0 == tf.gather); error :214-220
  core/user_ops/beam_search_op/group_gather_test.py:7-10   GroupGather docstring example
  core/platform/fingerprint_test.cc:27-28                  Fingerprint64 known answers

"recipe" cases: the reference test builds the input with np.random.permutation(np.linspace
(...)) and the expected output with np.argsort / np.sort (mergesort where it says so); the
same recipe is run here with a fixed seed and the result stored, so the vector is
reproducible without TensorFlow.  The int32 inputs of testStableSort are stored as f32
(exactly representable; the HIP/oracle TopKV2 is the f32 kernel the serving graph uses).

    python tests/golden/make_reference_held.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
T = "python/kernel_tests/topk_op_test.py"
G = "python/kernel_tests/gather_op_test.py"


def topk_cases():
    c = []
    c.append({"name": "testTop1", "src": T + ":96-98", "kind": "literal",
              "inputs": [[0.1, 0.3, 0.2, 0.4], [0.1, 0.3, 0.3, 0.2]], "k": 1,
              "values": [[0.4], [0.3]], "indices": [[3], [1]]})
    c.append({"name": "testTop2", "src": T + ":100-102", "kind": "literal",
              "inputs": [[0.1, 0.3, 0.2, 0.4], [0.1, 0.3, 0.4, 0.2]], "k": 2,
              "values": [[0.4, 0.3], [0.4, 0.3]], "indices": [[3, 1], [2, 1]]})
    c.append({"name": "testTopAll (k == n, ties)", "src": T + ":168-171", "kind": "literal",
              "inputs": [[0.1, 0.3, 0.2, 0.4], [0.1, 0.3, 0.3, 0.2]], "k": 4,
              "values": [[0.4, 0.3, 0.2, 0.1], [0.3, 0.3, 0.2, 0.1]],
              "indices": [[3, 1, 2, 0], [1, 2, 3, 0]]})
    c.append({"name": "testTop3Vector / testTensorK (1-D input)", "src": T + ":180-187", "kind": "literal",
              "inputs": [3, 6, 15, 18, 6, 12, 1, 17, 3, 0, 4, 19, 1, 6], "k": 3,
              "values": [19, 18, 17], "indices": [11, 3, 7]})
    c.append({"name": "testTop3ZeroRows", "src": T + ":189-192", "kind": "literal",
              "inputs_shape": [0, 10], "inputs": [], "k": 3, "values": [], "indices": [],
              "out_shape": [0, 3]})
    c.append({"name": "testTop1AllNan", "src": T + ":113-115", "kind": "literal", "nan": True,
              "inputs": [["nan", "nan"], ["nan", "nan"]], "k": 1, "values": [["nan"], ["nan"]],
              "indices": [[0], [0]]})
    c.append({"name": "testKNegative", "src": T + ":195-201", "kind": "error",
              "inputs": [[0.1, 0.2], [0.3, 0.4]], "k": -7, "error": "Need k >= 0, got -7"})
    c.append({"name": "testKTooLarge", "src": T + ":204-209 (kernel check: core/kernels/topk_op.cc:67-71)",
              "kind": "error", "inputs": [[0.1, 0.2], [0.3, 0.4]], "k": 4,
              "error": "input must have at least k columns"})
    rng = np.random.RandomState(20260927)

    def lin(name, src, hi, total, rows, k, dtype, **extra):
        # inputs = np.linspace(0, hi, total, dtype)[perm].reshape(rows, -1) as f32; the expected values
        # are inputs[indices] (what -np.sort(-inputs)[:k] is for distinct or stably ordered values)
        perm = rng.permutation(total)
        inp = np.linspace(0, hi, total, dtype=dtype)[perm].astype(np.float32).reshape(rows, -1)
        idx = np.argsort(-inp, axis=1, kind="mergesort")[:, :k]
        assert (np.take_along_axis(inp, idx, 1) == -np.sort(-inp, axis=1)[:, :k]).all()
        d = {"name": name, "src": T + src, "kind": "recipe", "linspace": [0, hi, total, np.dtype(dtype).name],
             "perm": perm.tolist(), "rows": rows, "k": k, "indices": idx.tolist()}
        d.update(extra)
        c.append(d)

    lin("testTop3 (recipe)", ":104-109", 100, 6140, 1, 5, np.float64)
    lin("testMediumTopK f32 (recipe)", ":148-160", 100, 5 * 500, 5, 50, np.float32)
    # testStableSort :162-173: repeated integers in [0, 3], mergesort (stable) argsort
    for k in (1, 5, 50, 500):
        lin("testStableSort k=%d (recipe)" % k, ":162-173", 3, 5 * 500, 5, k, np.int32)
    # _testLargeSort :117-130 / _testLargeTopK :132-146: k = n and k = n - 1 at n = 5000 (2 of the 10 rows)
    lin("testLargeSort f32 (recipe, 2 rows)", ":117-130", 100, 2 * 5000, 2, 5000, np.float32, large_k=True)
    lin("testLargeTopK f32 (recipe, 2 rows)", ":132-146", 100, 2 * 5000, 2, 4999, np.float32, large_k=True)
    return c


def gather_cases():
    c = []
    data1 = [0, 1, 2, 3, 7, 5]
    for dt in ("int64", "float32"):
        c.append({"name": "testScalar1D %s" % dt, "src": G + ":64-76", "kind": "literal", "dtype": dt,
                  "params": data1, "indices": [1, 2, 2, 4, 5], "expected": [1, 2, 2, 7, 5]})
        c.append({"name": "testScalar1D %s scalar index" % dt, "src": G + ":64-76", "kind": "literal", "dtype": dt,
                  "params": data1, "indices": [4], "expected": [7], "scalar": True})
    data2 = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [12, 13, 14]]
    for dt in ("int64", "float32"):
        c.append({"name": "testSimpleTwoD32 axis 0 %s" % dt, "src": G + ":92-108", "kind": "literal", "dtype": dt,
                  "params": data2, "indices": [0, 1, 0, 2],
                  "expected": [[0, 1, 2], [3, 4, 5], [0, 1, 2], [6, 7, 8]]})
        c.append({"name": "testScalar2D axis 0 %s" % dt, "src": G + ":78-90", "kind": "literal", "dtype": dt,
                  "params": data2, "indices": [2], "expected": [[6, 7, 8]], "scalar": True})
    c.append({"name": "testBatchDims batch_dims=0, 2-D indices", "src": G + ":259-263", "kind": "literal",
              "dtype": "int64", "params": [6, 7, 8, 9], "indices": [[2, 1], [0, 3]], "expected": [[8, 7], [6, 9]]})
    c.append({"name": "testBatchDims batch_dims=0, 3-D indices", "src": G + ":264-268", "kind": "literal",
              "dtype": "int64", "params": [6, 7, 8, 9], "indices": [[[3, 1], [2, 0]], [[0, 3], [2, 2]]],
              "expected": [[[9, 7], [8, 6]], [[6, 9], [8, 8]]]})
    c.append({"name": "testBatchDims batch_dims=0, 4-D indices", "src": G + ":269-275", "kind": "literal",
              "dtype": "int64", "params": [8, 9],
              "indices": [[[[0, 1], [1, 0]], [[0, 0], [1, 1]]], [[[1, 1], [0, 0]], [[0, 1], [1, 0]]]],
              "expected": [[[[8, 9], [9, 8]], [[8, 8], [9, 9]]], [[[9, 9], [8, 8]], [[8, 9], [9, 8]]]]})
    c.append({"name": "testBadIndicesCPU axis 0", "src": G + ":214-220", "kind": "error", "dtype": "int64",
              "params": [[0, 1, 2], [3, 4, 5]], "indices": [[7]], "error": "is not in [0, 2)", "bad_i": 0})
    return c


def main():
    out = {
        "_comment": "REFERENCE-HELD vectors: every expected value below is asserted by the reference's own "
                    "tests (literal, or by an explicit numpy recipe re-run here with a fixed seed). "
                    "Generated by tests/golden/make_reference_held.py; see its docstring for file:line.",
        "topk": topk_cases(),
        "gather": gather_cases(),
        "group_gather": [{
            "name": "group_gather() docstring example", "kind": "literal",
            "src": "core/user_ops/beam_search_op/group_gather_test.py:7-10",
            "params_values": [0, 1, 2, 3, 4, 5, 6, 7, 8, 9], "params_row_splits": [0, 2, 5, 7, 10],
            "indices_values": [0, 1, 3], "indices_row_splits": [0, 2, 3],
            "ret_values": [0, 1, 2, 3, 4, 7, 8, 9], "ret_row_splits": [0, 5, 8]}],
        # tensorflow::Fingerprint64 (FarmHash, un-vendored): what BloomFilterDifference hashes node strings with
        "fingerprint64": [
            {"name": "Fingerprint64(\"Hello\")", "kind": "literal", "src": "core/platform/fingerprint_test.cc:27",
             "input": "Hello", "expected": 15404698994557526151},
            {"name": "Fingerprint64(\"World\")", "kind": "literal", "src": "core/platform/fingerprint_test.cc:28",
             "input": "World", "expected": 18308117990299812472}],
    }
    with open(os.path.join(HERE, "reference_held.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote reference_held.json:", {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
