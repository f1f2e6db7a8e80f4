"""CPU: the generated backward dK / dV loop (tools/asmgen/attn_dkv.py -> unirec_amd/csrc/gen/attn_dkv_c128_asm.h) in the instruction
emulator against a float64 reference (the k / v half of SDPA's backward, transformers modeling_qwen3.py:185-208 under autograd; sum
over the query heads of the GQA group).  Counted waits, the 4-slot LDS-DMA ring and the hazards are enforced by the emulator."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))

import attn_dkv as G  # noqa: E402
import dkv_host as H  # noqa: E402
import emit  # noqa: E402
from fwd_host import f32_to_bf16  # noqa: E402


@pytest.fixture(scope="module")
def prog():
    return G.build_program()[0]


def test_committed_header_is_the_generator_output():
    with open(os.path.join(ROOT, "unirec_amd", "csrc", "gen", "attn_dkv_c128_asm.h")) as f:
        assert f.read() == emit.dkv_header(), "run python tools/asmgen/emit.py"


def _case(prog, S, xk, pad=0, holes=False, seed=0, rep=2, scale=128 ** -0.5):
    rng = np.random.default_rng(seed)
    q, k, v, do = [f32_to_bf16(rng.standard_normal((S, n * 128)).astype(np.float32)) for n in (rep, 1, 1, rep)]
    km = None
    if pad or holes:
        km = np.ones(S, bool)
        km[:pad] = False
        if holes:
            km[rng.integers(0, S, S // 5)] = False
    rK, rV, ws = H.reference(q, k, v, do, km, xk, 0, rep, scale)
    dK, dV, counts = H.run_block(q, k, v, do, km, xk, 0, rep, scale, ws, prog)
    assert np.isfinite(dK).all() and np.isfinite(dV).all()
    assert np.abs(dK - rK).max() < 0.02 * np.abs(rK).max() + 1e-3
    assert np.abs(dV - rV).max() < 0.02 * np.abs(rV).max() + 1e-3
    return counts


def test_first_key_block_sweeps_every_tile_of_both_heads(prog):
    c = _case(prog, 256, 0)
    assert [d["mfma"] for d in c] == [4 * 2 * 64] * 4          # 4 query tiles x 2 heads x 64 MFMAs, every wave


def test_last_key_block(prog):
    _case(prog, 256, 1)


def test_longer_sequence(prog):
    _case(prog, 512, 1)


@pytest.mark.parametrize("pad,xk", [(40, 0), (200, 1)])
def test_left_padding(prog, pad, xk):
    _case(prog, 512, xk, pad=pad)


def test_random_key_holes(prog):
    _case(prog, 512, 1, holes=True, seed=3)


def test_one_query_head_per_kv_head(prog):
    _case(prog, 256, 0, rep=1)
