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
    """bench.py prints the attention fractions on executed FLOPs too: whole 64-key tiles (the masked half of the diagonal ones
    included) minus the leading all-padding key tiles the kernels skip; the backward executes 7 contractions for the 5
    algorithmic ones."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    S = 2048
    full = bench.attn_executed_ratio(torch.ones(2, S), S)
    # 16 query blocks of 128: block qb sweeps 2 (qb + 1) tiles of 64 keys -> 128 * 64 * 272 pairs against S^2 / 2
    assert abs(full["fwd"] - (128 * 64 * 272) / (S * S / 2)) < 1e-12 and abs(full["bwd"] - 7.0 / 5.0 * full["fwd"]) < 1e-12
    am = torch.ones(1, S)
    am[0, :640] = 0                      # ten whole key tiles of left padding
    pad = bench.attn_executed_ratio(am, S)
    assert pad["fwd"] < full["fwd"] and pad["bwd"] < full["bwd"]
    tiles = sum(max(0, 2 * (qb + 1) - 10) for qb in range(16))
    assert abs(pad["fwd"] - tiles * 64 * 128 / (S * S / 2)) < 1e-12
    none = bench.attn_executed_ratio(torch.zeros(1, S), S)
    assert none["fwd"] == 0.0 and none["bwd"] == 0.0
