"""CPU: the metric of the reference's evaluation jobs (NANN_impls/nann/util.py:14-25, 28-59) as nann_amd/evaluate.py
restates it -- the harness around it (test / test_all on the HIP ops) is covered by the -m gpu tests."""
import math

from nann_amd import evaluate


def test_calc_pr_is_set_based_with_one_ground_truth():
    # util.py:14-25: both sides become sets; ONE ground-truth item -> recall is 0 or 1
    assert evaluate.calc_pr(7, [7]) == (1.0, 1.0, 1.0)
    p, r, f1 = evaluate.calc_pr(7, [1, 2, 7, 9])
    assert (p, r) == (0.25, 1.0) and math.isclose(f1, 0.4)
    # duplicates in the retrieved list count once (len(set(retrievals)) in the precision's denominator)
    p, r, f1 = evaluate.calc_pr(5, [1, 5, 9, 9])
    assert math.isclose(p, 1 / 3) and r == 1.0 and math.isclose(f1, 0.5)
    assert evaluate.calc_pr(4, [1, 5]) == (0.0, 0.0, 0.0)          # p + r == 0 -> f1 = 0.0, no division
    assert evaluate.calc_pr(4, range(200)) == (1 / 200, 1.0, 2 * (1 / 200) / (1 + 1 / 200))


def test_average_meter_is_a_weighted_mean():
    m = evaluate.AverageMeter()
    assert m.avg == 0.0                                            # nothing recorded yet
    m.update(1.0)
    m.update(0.0, n=3)
    assert m.count == 4 and m.sum == 1.0 and m.avg == 0.25
