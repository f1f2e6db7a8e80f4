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


def test_attention_executed_ratio_counts_whole_tiles_and_skipped_padding():
    """bench.py prints the attention fractions on executed MFMA work too: whole 64-key tiles per 64-query wave (three quarters of the
    diagonal one) plus the forward's row-sum products, minus the leading all-padding key tiles the kernels skip; the backward
    executes S and dP in both of its kernels."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    S = 2048
    full = bench.attn_executed_ratio(torch.ones(2, S), S)
    # 32 waves of 64 queries: wave w sweeps w + 1 tiles of 64 keys, the last one at 3/4
    pq = sum(w + 1 - 0.25 for w in range(32)) * 64 * 64
    pk = sum((S - 128 * kb) // 64 for kb in range(16)) * 64 * 128
    assert abs(full["fwd"] - 2.25 * pq / (2 * S * S / 2)) < 1e-12
    assert abs(full["bwd"] - (3 * pq + 4 * pk) / (5 * S * S / 2)) < 1e-12
    am = torch.ones(1, S)
    am[0, :640] = 0                      # ten whole key tiles of left padding
    pad = bench.attn_executed_ratio(am, S)
    assert pad["fwd"] < full["fwd"] and pad["bwd"] < full["bwd"]
    pq = sum(max(0, w + 1 - 10) - 0.25 for w in range(32) if w + 1 > 10) * 64 * 64
    assert abs(pad["fwd"] - 2.25 * pq / (2 * S * S / 2)) < 1e-12
    none = bench.attn_executed_ratio(torch.zeros(1, S), S)
    assert none["fwd"] == 0.0 and none["bwd"] == 0.0
