"""CPU: the generated forward attention loop (tools/asmgen/attn_fwd.py -> unirec_amd/csrc/gen/attn_fwd_c128_asm.h) run in the
instruction emulator of tools/asmgen/isa.py against a float64 attention reference (SDPA semantics: causal + key padding,
transformers modeling_qwen3.py:185-208).  The emulator also enforces every counted wait (LDS reads, LDS-DMA + barrier) and the
software-visible hazards, so a schedule edit that breaks one of them fails here, before it reaches an MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "asmgen"))

import attn_fwd as G  # noqa: E402
import emit  # noqa: E402
import fwd_host as H  # noqa: E402


@pytest.fixture(scope="module")
def prog():
    return G.build_program()[0]


def test_committed_header_is_the_generator_output():
    with open(os.path.join(ROOT, "unirec_amd", "csrc", "gen", "attn_fwd_c128_asm.h")) as f:
        assert f.read() == emit.fwd_header(), "run python tools/asmgen/emit.py"


def _case(prog, S, x, pad=0, holes=False, spike=None, amp=1.0, seed=0, nq=2, nkv=1, hq=1, scale=128 ** -0.5):
    rng = np.random.default_rng(seed)
    q = (rng.standard_normal((S, nq * 128)) * amp).astype(np.float32)
    k = (rng.standard_normal((S, nkv * 128)) * amp).astype(np.float32)
    v = rng.standard_normal((S, nkv * 128)).astype(np.float32)
    if spike is not None:      # one key far above everything before it: forces the deferred-maximum path mid-stream
        k[spike] = q[min(300, S - 1), hq * 128:(hq + 1) * 128] * 3
    q, k, v = H.f32_to_bf16(q), H.f32_to_bf16(k), H.f32_to_bf16(v)
    km = None
    if pad or holes:
        km = np.ones(S, bool)
        km[:pad] = False
        if holes:
            km[rng.integers(0, S, S // 5)] = False
    O, m, l, counts = H.run_block(q, k, v, km, x, hq, nq // nkv, scale, prog)
    ref = H.reference(q, k, v, km, x, hq, nq // nkv, scale)
    err = np.abs(O - ref).max()
    assert np.isfinite(O).all()
    # q is rounded to bf16 once more after the scale * log2(e) pre-multiplication: the error grows with the score magnitude
    assert err < 0.02 * max(1.0, np.abs(ref).max()) * amp * amp, err
    return counts


def test_first_block_all_four_waves(prog):
    c = _case(prog, 256, 0)
    # wave w sweeps w + 1 tiles: 32 MFMAs for its first tile's S (24 when it is the diagonal tile), 72 per steady tile, 64 for the
    # diagonal tile and 30 for the P V + row sums that follow it
    assert [d["mfma"] for d in c] == [54, 126, 198, 270]


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


def test_deferred_maximum_paths(prog):
    c = _case(prog, 512, 1, spike=130, amp=2.0)
    assert sum(d.get("@RESC0", 0) + d.get("@RESC1", 0) for d in c) > 4 and sum(d.get("@ORESC0", 0) + d.get("@ORESC1", 0) for d in c) > 0
    c = _case(prog, 512, 1, amp=4.0)
    assert sum(d.get("@ORESC0", 0) for d in c) > 0 and sum(d.get("@ORESC1", 0) for d in c) > 0


def test_ordinary_data_takes_the_rare_path_once(prog):
    c = _case(prog, 512, 1, seed=5)
    assert all(d.get("@RESC0", 0) == 1 and d.get("@RESC1", 0) == 0 and d.get("@ORESC0", 0) == 0 for d in c)
