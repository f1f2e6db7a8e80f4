"""GPU: the hand-scheduled causal head_dim-128 forward (attn_fwd_c128_kernel: generated main loop, tools/asmgen) against
(a) a plain PyTorch fp32 reference of the same op (SDPA semantics, transformers modeling_qwen3.py:185-208) and
(b) the compiler-scheduled attn_fwd_kernel it replaces (ur_attn_mode(UR_ATTN_MODE_C128, 0)), including the (m, 1/l) row statistics the backward reads."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from unirec_amd import hip  # noqa: E402

DEV = "cuda"


def _ref(q, k, v, km):
    B, S, nq, hd = q.shape
    rep = nq // k.shape[2]
    qh, kh, vh = q.float().permute(0, 2, 1, 3), k.float().permute(0, 2, 1, 3).repeat_interleave(rep, 1), v.float().permute(0, 2, 1, 3).repeat_interleave(rep, 1)
    s = qh @ kh.transpose(-1, -2) * hd ** -0.5
    ok = torch.tril(torch.ones(S, S, dtype=torch.bool, device=q.device))[None, None]
    if km is not None:
        ok = ok & km.bool()[:, None, None, :]
    w = torch.softmax(s.masked_fill(~ok, float("-inf")), dim=-1)
    w = torch.where(ok.any(-1, keepdim=True), w, torch.zeros_like(w))
    return (w @ vh).permute(0, 2, 1, 3)


def _inputs(B, S, nq, nkv, seed, amp=1.0, spike=False):
    g = torch.Generator(device="cpu").manual_seed(seed)
    buf = torch.randn(B, S, (nq + 2 * nkv) * 128, generator=g) * amp
    if spike:      # a key that lines up with later queries: the running maximum jumps mid-sequence
        buf[:, S // 3, nq * 128:(nq + 1) * 128] = buf[:, S - 5, :128] * 3
    buf = buf.to(DEV).to(torch.bfloat16)
    q = buf[..., :nq * 128].view(B, S, nq, 128)
    k = buf[..., nq * 128:(nq + nkv) * 128].view(B, S, nkv, 128)
    v = buf[..., (nq + nkv) * 128:].view(B, S, nkv, 128)
    return q, k, v


def _both(q, k, v, km):
    o_new, ctx_new = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    with hip.attn_mode_set(hip.ATTN_MODE_C128, 0):
        o_old, ctx_old = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    torch.cuda.synchronize()
    return o_new, ctx_new, o_old, ctx_old


def _stats(ctx):
    st = ctx["stats"] if isinstance(ctx, dict) else ctx.stats
    return st.float()


@pytest.mark.parametrize("S", [128, 256, 320, 1024, 2048])
@pytest.mark.parametrize("mask", ["none", "left", "holes"])
def test_against_fp32_and_the_compiler_scheduled_kernel(S, mask):
    B, nq, nkv = 3, 4, 2
    q, k, v = _inputs(B, S, nq, nkv, seed=S)
    km = None
    if mask != "none":
        g = torch.Generator(device="cpu").manual_seed(S + 1)
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "left":
            for b in range(B):
                km[b, : int(torch.randint(1, min(S - 1, 300), (1,), generator=g))] = 0
        else:
            km = (torch.rand(B, S, generator=g) < 0.8).to(torch.uint8)
            km[:, 0] = 1
        km = km.to(DEV)
    o_new, ctx_new, o_old, ctx_old = _both(q, k, v, km)
    ref = _ref(q, k, v, km)
    assert torch.isfinite(o_new.float()).all()
    assert torch.allclose(o_new.float(), ref, rtol=2e-2, atol=2e-2), (o_new.float() - ref).abs().max().item()
    assert torch.allclose(o_new.float(), o_old.float(), rtol=2e-2, atol=2e-2), (o_new.float() - o_old.float()).abs().max().item()


def test_row_statistics_are_a_valid_pair_for_the_backward():
    # (m, 1/l) may differ from the other kernel's (the maximum is deferred), but m + ln(l) = LSE must agree
    q, k, v = _inputs(2, 1024, 4, 2, seed=11)
    o_new, ctx_new, o_old, ctx_old = _both(q, k, v, None)
    dout = torch.randn_like(o_new.float()).to(torch.bfloat16)
    g_new = hip.attn_bwd(ctx_new, dout)
    g_old = hip.attn_bwd(ctx_old, dout)
    torch.cuda.synchronize()
    for a, b in zip(g_new, g_old):
        scale = b.float().abs().max().item()
        assert (a.float() - b.float()).abs().max().item() <= 2e-2 * scale + 1e-3


@pytest.mark.parametrize("amp,spike", [(2.0, True), (4.0, False)])
def test_deferred_maximum_under_large_scores(amp, spike):
    q, k, v = _inputs(2, 1024, 4, 2, seed=5, amp=amp, spike=spike)
    o_new, _, o_old, _ = _both(q, k, v, None)
    ref = _ref(q, k, v, None)
    tol = 2e-2 * amp * amp
    assert torch.isfinite(o_new.float()).all()
    assert (o_new.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert (o_new.float() - o_old.float()).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_bitwise_reproducible():
    q, k, v = _inputs(2, 2048, 4, 2, seed=3)
    a, _ = hip.attn_fwd(q, k, v, causal=True)
    b, _ = hip.attn_fwd(q, k, v, causal=True)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("S,B,mask", [(2048, 2, "none"), (1024, 4, "left"), (512, 8, "none")])
def test_persistent_dkv_walk_equals_one_workgroup_per_key_block(S, B, mask, monkeypatch):
    """The dK/dV kernel's persistent walk (8 x U workgroups, the key block rotating with the step) computes every (batch, kv head,
    key block) exactly as the one-workgroup-per-block launch does: dk, dv bit for bit (ur_attn_mode(UR_ATTN_MODE_DKV_PERSIST, 0) switches it off).  Shapes
    whose sweep divides evenly over 256 CUs: 16 (batch, kv head) groups x 16 / 8 / 4 key blocks."""
    nq, nkv = 16, 8 if S != 512 else 8
    q, k, v = _inputs(B, S, nq, nkv, 50 + S)
    km = None
    if mask == "left":
        km = torch.ones((B, S), dtype=torch.uint8)
        km[1, :200] = 0
        km = km.to(DEV)
    o, ctx = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    dout = torch.randn(B, S, nq, 128, generator=torch.Generator().manual_seed(3)).to(DEV).to(torch.bfloat16)
    dq1, dk1, dv1 = hip.attn_bwd(ctx, dout)
    with hip.attn_mode_set(hip.ATTN_MODE_DKV_PERSIST, 0):
        dq0, dk0, dv0 = hip.attn_bwd(ctx, dout)
    torch.cuda.synchronize()
    assert torch.equal(dk1, dk0) and torch.equal(dv1, dv0) and torch.equal(dq1, dq0)
    assert dk1.float().abs().max() > 0
