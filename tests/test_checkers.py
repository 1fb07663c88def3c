"""The test suite's own checkers, checked (CPU)."""
import numpy as np
import pytest

from conftest import assert_same_results_tol


def test_tolerance_comparison_is_not_vacuous():
    """the checker itself (CPU): permutations inside a tied run and swaps at the boundary pass, everything else fails"""
    ids = np.array([5, 9, 2, 7, 4], dtype=np.uint32)
    d = np.array([0.1, 0.2, 0.2 + 1e-9, 0.3, 0.4])
    assert assert_same_results_tol(ids, d, ids, d) == 0
    assert assert_same_results_tol(np.array([5, 2, 9, 7, 4]), d, ids, d) == 2                 # tied run permuted
    assert assert_same_results_tol(np.array([5, 9, 2, 7, 11]), d, ids, d) == 1                # another id AT the boundary distance
    with pytest.raises(AssertionError):
        assert_same_results_tol(np.array([5, 9, 2, 11, 4]), d, ids, d)                        # foreign id in the middle
    with pytest.raises(AssertionError):
        assert_same_results_tol(np.array([9, 5, 2, 7, 4]), d, ids, d)                         # swap across distinct distances
    d2 = np.array([0.1, 0.2, 0.25, 0.3, 0.4])
    with pytest.raises(AssertionError):
        assert_same_results_tol(np.array([5, 9, 7, 2, 4]), d2, ids, d2)                       # rank-wise equal distances would pass, ids not
    with pytest.raises(AssertionError):
        assert_same_results_tol(np.array([5, 9, 2, 7, 4]), d + 1e-2, ids, d)                  # distances off


