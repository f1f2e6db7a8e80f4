"""CPU: the generated backward dQ loop (tools/asmgen/attn_dq.py -> unirec_amd/csrc/gen/attn_dq_c128_asm.h) in the instruction
emulator against a float64 reference of dQ = scale * (P o (dO V^T - delta)) K (the q half of SDPA's backward, transformers
modeling_qwen3.py:185-208 under autograd).  Counted waits, the LDS-DMA ring protocol and the hazards are enforced by the emulator."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))

import attn_dq as G  # noqa: E402
import dq_host as H  # noqa: E402
import emit  # noqa: E402
from fwd_host import f32_to_bf16  # noqa: E402


@pytest.fixture(scope="module")
def prog():
    return G.build_program()[0]


def test_committed_header_is_the_generator_output():
    with open(os.path.join(ROOT, "unirec_amd", "csrc", "gen", "attn_dq_c128_asm.h")) as f:
        assert f.read() == emit.dq_header(), "run python tools/asmgen/emit.py"


def _case(prog, S, x, pad=0, holes=False, seed=0, nq=2, nkv=1, hq=1, scale=128 ** -0.5):
    rng = np.random.default_rng(seed)
    q, k, v, do = [f32_to_bf16(rng.standard_normal((S, n * 128)).astype(np.float32)) for n in (nq, nkv, nkv, nq)]
    km = None
    if pad or holes:
        km = np.ones(S, bool)
        km[:pad] = False
        if holes:
            km[rng.integers(0, S, S // 5)] = False
    dQ, counts = H.run_block(q, k, v, do, km, x, hq, nq // nkv, scale, prog)
    ref = H.reference(q, k, v, do, km, x, hq, nq // nkv, scale)
    assert np.isfinite(dQ).all()
    assert np.abs(dQ - ref).max() < 0.02 * np.abs(ref).max() + 1e-3
    return counts


def test_first_block_all_four_waves(prog):
    c = _case(prog, 256, 0)
    # first tile 64 MFMAs (48 on the diagonal), steady 96, diagonal 80, then 24 for the dQ products that follow it
    assert [d["mfma"] for d in c] == [72, 168, 264, 360]


def test_second_block_full_pipeline(prog):
    _case(prog, 512, 1)


@pytest.mark.parametrize("pad", [40, 100, 300])
def test_left_padding(prog, pad):
    _case(prog, 512, 1, pad=pad)
    _case(prog, 256, 0, pad=min(pad, 200))


def test_random_key_holes(prog):
    _case(prog, 512, 1, holes=True, seed=3)


def test_block_without_a_valid_key_is_zero(prog):
    c = _case(prog, 512, 0, pad=256)
    assert all(d.get("mfma", 0) == 0 for d in c)


def test_sequence_not_a_multiple_of_the_block(prog):
    c = _case(prog, 320, 1)
    assert [d.get("mfma", 0) for d in c][1:] == [0, 0, 0]
