"""GPU: fused attention fwd/bwd against a plain PyTorch fp32 reference of the same op, for both mask
semantics (Q-Former additive finfo.min incl. fully masked rows; Qwen3 causal + padding, SDPA zeros)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from unirec_amd import hip  # noqa: E402

DEV = "cuda"
F32_MIN = torch.finfo(torch.float32).min


def _randn(shape, seed, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).to(DEV).to(torch.bfloat16)


def _ref(q, k, v, key_mask, causal):
    """q [B,Sq,nq,hd] fp32 etc.  Returns o [B,Sq,nq,hd]."""
    B, Sq, nq, hd = q.shape
    Sk, nkv = k.shape[1], k.shape[2]
    rep = nq // nkv
    qh = q.permute(0, 2, 1, 3)
    kh = k.permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    vh = v.permute(0, 2, 1, 3).repeat_interleave(rep, dim=1)
    s = qh @ kh.transpose(-1, -2) * hd ** -0.5
    if causal:
        ok = torch.tril(torch.ones(Sq, Sk, dtype=torch.bool, device=q.device))[None, None]
        if key_mask is not None:
            ok = ok & key_mask.bool()[:, None, None, :]
        w = torch.softmax(s.masked_fill(~ok, float("-inf")), dim=-1)
        w = torch.where(ok.any(-1, keepdim=True), w, torch.zeros_like(w))
    else:
        if key_mask is not None:
            s = s + (1.0 - key_mask.float())[:, None, None, :] * F32_MIN
        w = torch.softmax(s, dim=-1)
    return (w @ vh).permute(0, 2, 1, 3)


def _run(B, Sq, Sk, nq, nkv, hd, causal, mask_kind, seed=0, fused_qkv=False):
    if fused_qkv:   # q,k,v as strided slices of one projection buffer (how the model stack calls it)
        buf = _randn((B, Sq, (nq + 2 * nkv) * hd), seed)
        q = buf[..., :nq * hd].view(B, Sq, nq, hd)
        k = buf[..., nq * hd:(nq + nkv) * hd].view(B, Sq, nkv, hd)
        v = buf[..., (nq + nkv) * hd:].view(B, Sq, nkv, hd)
    else:
        q, k, v = _randn((B, Sq, nq, hd), seed), _randn((B, Sk, nkv, hd), seed + 1), _randn((B, Sk, nkv, hd), seed + 2)
    km = None
    if mask_kind != "none":
        g = torch.Generator(device="cpu").manual_seed(seed + 7)
        km = (torch.rand((B, Sk), generator=g) < 0.7).to(torch.uint8)
        km[:, 0] = 1
        if mask_kind == "left":       # left padding: first keys masked -> fully masked causal rows
            km[:] = 1
            for b in range(1, B):
                km[b, : min(Sk - 1, 3 * b + 1)] = 0
        if mask_kind == "full" and B > 1:
            km[1] = 0                 # fully masked sample (padded history slot)
        km = km.to(DEV)
    dout = _randn((B, Sq, nq, hd), seed + 3)
    o, ctx = hip.attn_fwd(q, k, v, causal=causal, key_mask=km)
    dq, dk, dv = hip.attn_bwd(ctx, dout)
    qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref(qf, kf, vf, km, causal)
    ref.backward(dout.float())
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    tol = dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(o.float(), ref, **tol), f"o max err {(o.float() - ref).abs().max().item()}"
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        err = (got.float() - want).abs().max().item()
        scale = want.abs().max().item()
        assert err <= 2e-2 * scale + 2e-2, f"{name}: max err {err} (scale {scale})"
    return o


@pytest.mark.parametrize("Sq,Sk", [(4, 8), (2, 14), (32, 32), (32, 14), (64, 64), (64, 100), (64, 1600), (40, 70)])
@pytest.mark.parametrize("mask_kind", ["none", "rand", "full"])
def test_qformer_attention(Sq, Sk, mask_kind):
    _run(B=3, Sq=Sq, Sk=Sk, nq=2, nkv=2, hd=64, causal=False, mask_kind=mask_kind, seed=Sq * 1000 + Sk)


def test_qformer_fully_masked_row_is_uniform():
    B, Sq, Sk, nh, hd = 2, 4, 10, 2, 64
    q, k, v = _randn((B, Sq, nh, hd), 1), _randn((B, Sk, nh, hd), 2), _randn((B, Sk, nh, hd), 3)
    km = torch.zeros((B, Sk), dtype=torch.uint8, device=DEV)
    o, _ = hip.attn_fwd(q, k, v, causal=False, key_mask=km)
    want = v.float().mean(dim=1, keepdim=True).expand(B, Sq, nh, hd)
    assert torch.allclose(o.float(), want, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("S", [5, 48, 64, 130, 512, 1024])
@pytest.mark.parametrize("mask_kind", ["none", "rand", "left"])
def test_qwen3_causal_gqa(S, mask_kind):
    _run(B=2, Sq=S, Sk=S, nq=4, nkv=2, hd=128, causal=True, mask_kind=mask_kind, seed=S, fused_qkv=True)


def test_causal_hd64_and_noncausal_hd128():
    _run(B=2, Sq=96, Sk=96, nq=2, nkv=1, hd=64, causal=True, mask_kind="rand", seed=5)
    _run(B=2, Sq=64, Sk=200, nq=2, nkv=2, hd=128, causal=False, mask_kind="rand", seed=6)


def test_attention_dropout_is_deterministic_and_unbiased():
    B, Sq, Sk, nh, hd = 4, 64, 256, 4, 64
    q, k, v = _randn((B, Sq, nh, hd), 1, 0.3), _randn((B, Sk, nh, hd), 2, 0.3), _randn((B, Sk, nh, hd), 3)
    o0, _ = hip.attn_fwd(q, k, v, causal=False)
    o1, c1 = hip.attn_fwd(q, k, v, causal=False, dropout_p=0.2, seed=11)
    o2, _ = hip.attn_fwd(q, k, v, causal=False, dropout_p=0.2, seed=11)
    o3, _ = hip.attn_fwd(q, k, v, causal=False, dropout_p=0.2, seed=12)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    assert (o1.float() - o0.float()).abs().mean() < 0.2 and not torch.equal(o1, o0)
    # backward with dropout: finite-difference-free check via linearity in V: dV = (P.drop)^T dO
    dout = _randn((B, Sq, nh, hd), 4)
    dq, dk, dv = hip.attn_bwd(c1, dout)
    assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all()
    # o is linear in v: <dout, o(v)> == <dv, v>
    lhs = (dout.float() * o1.float()).sum().item()
    rhs = (dv.float() * v.float()).sum().item()
    assert abs(lhs - rhs) <= 2e-2 * abs(lhs) + 1.0


@pytest.mark.parametrize("B,nh,Sq,Sk", [(7, 3, 2, 2), (50, 16, 2, 14), (5, 4, 4, 4), (9, 2, 4, 8), (3, 1, 1, 16), (6, 5, 3, 5)])
@pytest.mark.parametrize("p_drop", [0.0, 0.2])
def test_tiny_attention_kernels(B, nh, Sq, Sk, p_drop, monkeypatch):
    """<= 4 queries x <= 16 keys (the item Q-Former of the joint step: 2 x 2 and 2 x 14) runs on the DPP-row kernels
    (4 pairs per wave, one backward kernel): against the fp32 reference without dropout, and against the MFMA kernels
    (ur_attn_mode(UR_ATTN_MODE_TINY, 0); same dropout counters, so the same masks) with it.  Pair counts that do not fill a wave, ragged
    masks and a fully masked sample (uniform softmax) included."""
    q, k, v = _randn((B, Sq, nh, 64), 1, 0.7), _randn((B, Sk, nh, 64), 2, 0.7), _randn((B, Sk, nh, 64), 3)
    g = torch.Generator(device="cpu").manual_seed(Sk * 10 + Sq)
    km = (torch.rand((B, Sk), generator=g) < 0.7).to(torch.uint8)
    km[:, 0] = 1
    km[1] = 0
    km = km.to(DEV)
    dout = _randn((B, Sq, nh, 64), 4)
    o1, c1 = hip.attn_fwd(q, k, v, causal=False, key_mask=km, dropout_p=p_drop, seed=9)
    dq1, dk1, dv1 = hip.attn_bwd(c1, dout)
    with hip.attn_mode_set(hip.ATTN_MODE_TINY, 0):
        o0, c0 = hip.attn_fwd(q, k, v, causal=False, key_mask=km, dropout_p=p_drop, seed=9)
        dq0, dk0, dv0 = hip.attn_bwd(c0, dout)
        # mixed: MFMA backward from the tiny forward's context (its row statistics must be the same quantities)
        dqm, dkm, dvm = hip.attn_bwd(c1, dout)
        torch.cuda.synchronize()
    if p_drop == 0.0:
        qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
        ref = _ref(qf, kf, vf, km, False)
        ref.backward(dout.float())
        want = (ref, qf.grad, kf.grad, vf.grad)
    else:
        want = (o0.float(), dq0.float(), dk0.float(), dv0.float())
    for name, got, w in zip(("o", "dq", "dk", "dv"), (o1, dq1, dk1, dv1), want):
        err = (got.float() - w).abs().max().item()
        assert err <= 2e-2 * w.abs().max().item() + 2e-2, f"{name}: max err {err} (scale {w.abs().max().item()})"
    for name, got, w in zip(("dq", "dk", "dv"), (dqm, dkm, dvm), (dq0, dk0, dv0)):
        err = (got.float() - w.float()).abs().max().item()
        assert err <= 2e-2 * w.float().abs().max().item() + 2e-2, f"mixed {name}: max err {err}"
    assert o1[1].float().sub(v[1].float().mean(dim=0, keepdim=True)).abs().max() < (3e-2 if p_drop == 0.0 else 10.0)   # fully masked sample: uniform


@pytest.mark.parametrize("B,nh,Sq,Sk,p_drop", [(3, 2, 64, 1600, 0.1), (2, 4, 64, 1600, 0.0), (5, 2, 40, 333, 0.2), (1, 1, 20, 257, 0.1),
                                               (2, 16, 64, 800, 0.1), (40, 16, 33, 256, 0.0)])
def test_few_query_dkv_kernel_matches_generic_kernel_bitwise(B, nh, Sq, Sk, p_drop, monkeypatch):
    """The user Q-Former's cross-attention backward (<= 64 queries, >= 256 keys) takes attn_bwd_dkv_fewq_kernel: the
    same arithmetic in the same order as attn_bwd_dkv_kernel, so dK / dV must agree bit for bit -- with ragged key
    masks (incl. a fully masked sample: uniform softmax), probability dropout, key counts that are not multiples of
    32 and both workgroup-per-pair and chunked grids."""
    q, k, v = _randn((B, Sq, nh, 64), 1, 0.5), _randn((B, Sk, nh, 64), 2, 0.5), _randn((B, Sk, nh, 64), 3)
    g = torch.Generator(device="cpu").manual_seed(Sk)
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    km = (torch.arange(Sk)[None, :] < lens[:, None]).to(torch.uint8)
    if B > 1:
        km[1] = 0
    km = km.to(DEV)
    dout = _randn((B, Sq, nh, 64), 4)
    o, ctx = hip.attn_fwd(q, k, v, causal=False, key_mask=km, dropout_p=p_drop, seed=5)
    with hip.attn_mode_set(hip.ATTN_MODE_FEWQ, 0):
        dq0, dk0, dv0 = hip.attn_bwd(ctx, dout)
    dq1, dk1, dv1 = hip.attn_bwd(ctx, dout)
    torch.cuda.synchronize()
    assert torch.equal(dq0, dq1)
    assert torch.equal(dk0, dk1), f"dk differs: {(dk0.float() - dk1.float()).abs().max().item()}"
    assert torch.equal(dv0, dv1), f"dv differs: {(dv0.float() - dv1.float()).abs().max().item()}"
    assert dk1.float().abs().max() > 0
    if p_drop == 0.0:      # and against the fp32 reference
        qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
        _ref(qf, kf, vf, km, False).backward(dout.float())
        for name, got, want in (("dk", dk1, kf.grad), ("dv", dv1, vf.grad)):
            err = (got.float() - want).abs().max().item()
            assert err <= 2e-2 * want.abs().max().item() + 2e-2, f"{name}: max err {err}"


@pytest.mark.parametrize("S,mask_kind", [(130, "none"), (512, "left"), (200, "rand")])
def test_dq_kernel_carries_qnorm_rope_backward(S, mask_kind):
    """ur_attn_bwd with rope_q_raw: the dQ kernel un-rotates, applies the norm weight and the RMS-norm backward in its store
    and writes the gradient of the RAW q projection; against the two-kernel path (dQ, then ur_qknorm_rope_bwd) on the same
    inputs -- same arithmetic from the same bf16-rounded dq, so equal up to the order of two 128-term row sums."""
    B, nq, nkv, hd, eps = 2, 4, 2, 128, 1e-6
    NQ, NKV = nq * hd, nkv * hd
    M = B * S
    qkv = _randn((M, NQ + 2 * NKV), S).contiguous()
    g = torch.Generator(device="cpu").manual_seed(S)
    qw = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(DEV)
    kw = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(DEV)
    cos, sin = hip.rope_table(S, hd, 1e6, DEV)
    q_r, k_r = hip.qknorm_rope_fwd(qkv, qw, kw, cos, sin, S, nq, nkv, hd, eps)
    km = None
    if mask_kind != "none":
        km = torch.ones((B, S), dtype=torch.uint8)
        if mask_kind == "left":
            km[1, :70] = 0
        else:
            km = (torch.rand((B, S), generator=g) < 0.8).to(torch.uint8); km[:, 0] = 1
        km = km.to(DEV)
    v4 = qkv[:, NQ + NKV:].view(B, S, nkv, hd)
    o, ctx = hip.attn_fwd(q_r.view(B, S, nq, hd), k_r.view(B, S, nkv, hd), v4, causal=True, key_mask=km)
    dout = _randn((B, S, nq, hd), S + 3)
    # two kernels
    d0 = torch.zeros_like(qkv)
    dq_r = torch.empty((M, NQ), dtype=torch.bfloat16, device=DEV); dk_r = torch.empty((M, NKV), dtype=torch.bfloat16, device=DEV)
    hip.attn_bwd(ctx, dout, dq=dq_r.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd), dv=d0[:, NQ + NKV:].view(B, S, nkv, hd))
    hip.qknorm_rope_bwd(dq_r, dk_r, qkv, qw, kw, cos, sin, d0, S, nq, nkv, hd, eps)
    # fused q part + k-only stand-alone kernel
    d1 = torch.zeros_like(qkv)
    dk2 = torch.empty_like(dk_r)
    hip.attn_bwd(ctx, dout, dk=dk2.view(B, S, nkv, hd), dv=d1[:, NQ + NKV:].view(B, S, nkv, hd), rope_q=(qkv[:, :NQ], qw, cos, sin, eps, d1[:, :NQ]))
    hip.qknorm_rope_bwd(dk2, dk2, qkv[:, NQ:], qw, kw, cos, sin, d1[:, NQ:], S, 0, nkv, hd, eps)
    torch.cuda.synchronize()
    assert torch.equal(dk_r, dk2) and torch.equal(d0[:, NQ:], d1[:, NQ:])          # k and v parts: identical kernels / inputs
    a, bq = d0[:, :NQ].float(), d1[:, :NQ].float()
    assert torch.isfinite(bq).all() and a.abs().max() > 0
    assert (a - bq).abs().max() <= 2e-2 * a.abs().max()                            # elementwise: a bf16 ulp or two
    assert (a - bq).norm() / a.norm() < 3e-3


@pytest.mark.parametrize("S,mask_kind", [(130, "none"), (512, "left"), (200, "rand"), (1024, "left")])
def test_dq_kernel_carries_qnorm_rope_backward_from_roped_q(S, mask_kind):
    """ur_attn_bwd with rope_rstd: the forward ran q/k-norm + RoPE as the q|k|v GEMM's epilogue, so only the ROPED q and 1 / rms exist.
    The dQ kernel's store recovers the normalised row from its own q operand and writes the gradient of the raw projection; against
    the two-kernel path (dQ, then ur_qknorm_rope_bwd_roped) on the same inputs, and the k-only stand-alone kernel against the k
    columns of the full one (same kernel, same inputs: bit for bit).  S 512 / 1024 run the generated kernels, 130 / 200 the generic."""
    B, nq, nkv, hd, eps = 2, 4, 2, 128, 1e-6
    NQ, NKV = nq * hd, nkv * hd
    M = B * S
    qkv = _randn((M, NQ + 2 * NKV), S).contiguous()
    g = torch.Generator(device="cpu").manual_seed(S)
    qw = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(DEV)
    kw = (1.0 + 0.1 * torch.randn(hd, generator=g)).to(DEV)
    cos, sin = hip.rope_table(S, hd, 1e6, DEV)
    q_r, k_r = hip.qknorm_rope_fwd(qkv, qw, kw, cos, sin, S, nq, nkv, hd, eps)
    rstd = torch.rsqrt(qkv[:, :NQ + NKV].float().view(M, nq + nkv, hd).pow(2).mean(-1) + eps).contiguous()
    km = None
    if mask_kind != "none":
        km = torch.ones((B, S), dtype=torch.uint8)
        if mask_kind == "left":
            km[1, :70] = 0
        else:
            km = (torch.rand((B, S), generator=g) < 0.8).to(torch.uint8); km[:, 0] = 1
        km = km.to(DEV)
    v4 = qkv[:, NQ + NKV:].view(B, S, nkv, hd)
    o, ctx = hip.attn_fwd(q_r.view(B, S, nq, hd), k_r.view(B, S, nkv, hd), v4, causal=True, key_mask=km)
    dout = _randn((B, S, nq, hd), S + 3)
    d0 = torch.zeros_like(qkv)
    dq_r = torch.empty((M, NQ), dtype=torch.bfloat16, device=DEV); dk_r = torch.empty((M, NKV), dtype=torch.bfloat16, device=DEV)
    hip.attn_bwd(ctx, dout, dq=dq_r.view(B, S, nq, hd), dk=dk_r.view(B, S, nkv, hd), dv=d0[:, NQ + NKV:].view(B, S, nkv, hd))
    hip.qknorm_rope_bwd_roped(dq_r, dk_r, q_r, k_r, rstd, qw, kw, cos, sin, d0, S, nq, nkv, hd)
    d1 = torch.zeros_like(qkv)
    dk2 = torch.empty_like(dk_r)
    hip.attn_bwd(ctx, dout, dk=dk2.view(B, S, nkv, hd), dv=d1[:, NQ + NKV:].view(B, S, nkv, hd), rope_q=(q_r, qw, cos, sin, eps, d1[:, :NQ]),
                 rope_rstd=(rstd, 0))
    hip.qknorm_rope_bwd_roped_k(dk2, k_r, rstd, nq, kw, cos, sin, d1[:, NQ:NQ + NKV], S, nkv, hd)
    torch.cuda.synchronize()
    assert torch.equal(dk_r, dk2) and torch.equal(d0[:, NQ:], d1[:, NQ:])
    a, bq = d0[:, :NQ].float(), d1[:, :NQ].float()
    assert torch.isfinite(bq).all() and a.abs().max() > 0
    assert (a - bq).abs().max() <= 2e-2 * a.abs().max()
    assert (a - bq).norm() / a.norm() < 3e-3
    # the k heads in the same call (rope_k): the dK/dV kernel's store where the generated kernel runs (S 512 / 1024), the stand-alone
    # kernel inside ur_attn_bwd elsewhere (bit for bit then)
    d3 = torch.zeros_like(qkv)
    dk3 = torch.empty_like(dk_r)
    hip.attn_bwd(ctx, dout, dk=dk3.view(B, S, nkv, hd), dv=d3[:, NQ + NKV:].view(B, S, nkv, hd), rope_q=(q_r, qw, cos, sin, eps, d3[:, :NQ]),
                 rope_rstd=(rstd, 0), rope_k=(k_r, kw, nq, d3[:, NQ:NQ + NKV]))
    torch.cuda.synchronize()
    assert torch.equal(d3[:, :NQ], d1[:, :NQ]) and torch.equal(d3[:, NQ + NKV:], d1[:, NQ + NKV:])
    ak, bk = d0[:, NQ:NQ + NKV].float(), d3[:, NQ:NQ + NKV].float()
    if S % 128:
        assert torch.equal(ak, bk)
    assert torch.isfinite(bk).all() and ak.abs().max() > 0
    assert (ak - bk).abs().max() <= 2e-2 * ak.abs().max() and (ak - bk).norm() / ak.norm() < 3e-3
    # and against the raw-projection backward (the reference chain): to the tolerance of the roped recovery
    d2 = torch.zeros_like(qkv)
    hip.qknorm_rope_bwd(dq_r, dk_r, qkv, qw, kw, cos, sin, d2, S, nq, nkv, hd, eps)
    assert (d2[:, :NQ].float() - bq).norm() / d2[:, :NQ].float().norm() < 1e-2


@pytest.mark.parametrize("spike_key,gain", [(200, 6.0), (31, 3.0), (449, 12.0), (64, 1.5)])
def test_deferred_max_rescale_is_exact_when_forced(spike_key, gain):
    """The forward defers the running-maximum update while the maximum grows by less than 2^6: a rare, data-dependent
    branch.  Force it both ways: one key row aligned with the query rows makes the row maximum jump at a chosen tile
    (by more than the threshold for large gains, by less for small ones), against a full-tensor f32 reference,
    forward and backward (the backward re-derives P from the saved log-sum-exp, so a wrong l or m shows there too)."""
    B, S, nq, nkv, hd = 1, 512, 2, 1, 128
    q = _randn((B, S, nq, hd), 11, std=0.5)
    k = _randn((B, S, nkv, hd), 12, std=0.5)
    v = _randn((B, S, nkv, hd), 13)
    direction = torch.nn.functional.normalize(torch.ones(hd), dim=0).to(DEV)
    q = (q.float() + 2.0 * direction).to(torch.bfloat16)          # every query has a component along `direction`
    k[0, spike_key, 0] = (gain * 4.0 * direction).to(torch.bfloat16)      # ... and one key is far along it
    dout = _randn((B, S, nq, hd), 14)
    o, ctx = hip.attn_fwd(q, k, v, causal=True)
    dq, dk, dv = hip.attn_bwd(ctx, dout)
    qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref(qf, kf, vf, None, True)
    ref.backward(dout.float())
    assert torch.allclose(o.float(), ref, rtol=2e-2, atol=2e-2), (o.float() - ref).abs().max().item()
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        err, scale = (got.float() - want).abs().max().item(), want.abs().max().item()
        assert err <= 2e-2 * scale + 2e-2, f"{name}: max err {err} (scale {scale})"


@pytest.mark.parametrize("S,pads", [(512, (200, 130)), (384, (383, 64)), (1024, (0, 700))])
def test_causal_left_padding_skips_whole_tiles(S, pads):
    """Left padding long enough to cover whole 64-key tiles and whole 128-key blocks: the causal kernels skip those
    (forward / dQ: leading key tiles and their K/V loads; dK/dV: whole key blocks) -- outputs and every gradient must
    still equal the dense masked reference, zero rows included (padded queries: o = 0; padded keys: dK = dV = 0)."""
    B, nq, nkv, hd = 2, 4, 2, 128
    q, k, v = _randn((B, S, nq, hd), 21), _randn((B, S, nkv, hd), 22), _randn((B, S, nkv, hd), 23)
    km = torch.ones((B, S), dtype=torch.uint8)
    for b, npad in enumerate(pads):
        km[b, :npad] = 0
    km[1, min(S - 1, pads[1] + 70)] = 0              # plus one masked key in the middle of the valid range
    km = km.to(DEV)
    dout = _randn((B, S, nq, hd), 24)
    o, ctx = hip.attn_fwd(q, k, v, causal=True, key_mask=km)
    dq, dk, dv = hip.attn_bwd(ctx, dout)
    qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    ref = _ref(qf, kf, vf, km, True)
    ref.backward(dout.float())
    assert torch.isfinite(o.float()).all()
    assert torch.allclose(o.float(), ref, rtol=2e-2, atol=2e-2), (o.float() - ref).abs().max().item()
    for b, npad in enumerate(pads):
        assert o[b, :npad].float().abs().max().item() == 0 if npad else True
        if npad:
            assert dk[b, :npad].float().abs().max().item() == 0 and dv[b, :npad].float().abs().max().item() == 0
            assert dq[b, :npad].float().abs().max().item() == 0
    for name, got, want in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        err, scale = (got.float() - want).abs().max().item(), want.abs().max().item()
        assert err <= 2e-2 * scale + 2e-2, f"{name}: max err {err} (scale {scale})"
