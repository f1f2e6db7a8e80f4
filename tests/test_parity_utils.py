"""CPU: the parity helper itself -- an analytically zero reference gradient must not let a large HIP result pass."""
import numpy as np
import pytest

from tests.parity_utils import assert_close


def test_floor_branch_checks_the_result_too():
    want = np.zeros((4, 4))
    assert_close(np.full((4, 4), 1e-5), want, 4e-2, "tiny", floor=1e-6, ref_scale=1.0)        # noise at 4e-5 of the neighbours' scale
    with pytest.raises(AssertionError):
        assert_close(np.full((4, 4), 0.5), want, 4e-2, "garbage", floor=1e-6, ref_scale=1.0)
    with pytest.raises(AssertionError):
        assert_close(np.full((4, 4), 1e-3), want, 4e-2, "no scale given", floor=1e-6)
