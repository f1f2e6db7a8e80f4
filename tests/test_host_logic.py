"""CPU: host-side logic that needs no kernel -- optimizer-state resume across a pack layout change, the C-ABI workspace sizes."""
import pytest
import torch
import torch.nn as nn

from unirec_amd import _lib
from unirec_amd.optim import FusedAdamW
from unirec_amd.packing import ParamPack


def _pack(order):
    g = torch.Generator().manual_seed(3)
    ps = {"a.weight": (8, 16), "b.weight": (24,), "c.weight": (4, 8), "d.bias": (7,)}
    return ParamPack([(n, nn.Parameter(torch.randn(ps[n], generator=g))) for n in order], "cpu")


def test_optimizer_state_survives_a_pack_layout_change():
    """A checkpoint written before a reordering of the pack (the hoisted cross-attention K|V weights moved every later offset)
    loads into the new layout tensor by tensor; a different SET of tensors is still refused."""
    old = _pack(["a.weight", "b.weight", "c.weight", "d.bias"])
    o_old = FusedAdamW([old])
    g = torch.Generator().manual_seed(5)
    o_old.state[0][0].copy_(torch.randn(old.numel, generator=g))
    o_old.state[0][1].copy_(torch.rand(old.numel, generator=g))
    o_old.steps[0].update({"a.weight": 3, "b.weight": 3, "c.weight": 2, "d.bias": 0})
    o_old.step_count = 3
    sd = o_old.state_dict()

    new = _pack(["c.weight", "a.weight", "d.bias", "b.weight"])
    assert new.offsets != old.offsets
    o_new = FusedAdamW([new])
    o_new.load_state_dict(sd)
    for n in new.names:
        num = new.params[n].numel()
        for k in (0, 1):
            got = o_new.state[0][k][new.offsets[n]:new.offsets[n] + num]
            want = o_old.state[0][k][old.offsets[n]:old.offsets[n] + num]
            assert torch.equal(got, want), n
    assert o_new.steps[0] == o_old.steps[0] and o_new.step_count == 3

    same = FusedAdamW([_pack(["a.weight", "b.weight", "c.weight", "d.bias"])])
    same.load_state_dict(sd)                                  # identical layout: the flat copy
    assert torch.equal(same.state[0][0], o_old.state[0][0])

    g2 = torch.Generator().manual_seed(3)
    other = ParamPack([("a.weight", nn.Parameter(torch.randn(8, 16, generator=g2))), ("zz", nn.Parameter(torch.randn(3, generator=g2)))], "cpu")
    with pytest.raises(ValueError):
        FusedAdamW([other]).load_state_dict({**sd, "packs": sd["packs"]})


def test_optimizer_state_refuses_the_same_names_at_another_shape():
    """the reorder branch must not load moments recorded for another shape of the same tensor names (a different LoRA
    rank): recorded extents are derived from the recorded offsets and compared with the current sizes."""
    def pk(order, rank):
        g = torch.Generator().manual_seed(3)
        ps = {"l.lora_A": (rank, 32), "l.lora_B": (32, rank), "m.lora_A": (rank, 32)}
        return ParamPack([(n, nn.Parameter(torch.randn(ps[n], generator=g))) for n in order], "cpu")
    sd = FusedAdamW([pk(["l.lora_A", "l.lora_B", "m.lora_A"], 16)]).state_dict()
    with pytest.raises(ValueError, match="recorded with"):
        FusedAdamW([pk(["m.lora_A", "l.lora_A", "l.lora_B"], 8)]).load_state_dict(sd)
    FusedAdamW([pk(["m.lora_A", "l.lora_A", "l.lora_B"], 16)]).load_state_dict(sd)     # same shapes, another order: accepted


def test_attention_backward_workspace_holds_the_queue_words():
    """ur_attn_bwd's workspace = two row-constant planes + the call's own eight work-queue words (no library-owned device state)."""
    lib = _lib.load()
    for B, nq, Sq in ((1, 1, 1), (2, 4, 128), (64, 16, 2048)):
        n = lib.ur_attn_bwd_workspace_floats(B, nq, Sq)
        assert n >= 2 * B * nq * Sq + 8
