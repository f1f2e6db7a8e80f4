"""GPU: every primitive of the C ABI against a plain PyTorch fp32 reference of the same op.
Integer-valued operands make the MFMA layout checks EXACT (a wrong fragment map cannot hide)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from unirec_amd import hip  # noqa: E402

DEV = "cuda"


def _ints(shape, lo=-3, hi=4, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(torch.float32)


def _randn(shape, seed=0, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g) * std


def _bf(x):
    return x.to(DEV).to(torch.bfloat16).contiguous()


def _ref_gemm(Rm, Sm):  # Rm [M,K], Sm [N,K] (logical), fp32
    return Rm.double() @ Sm.double().t()


LAYOUTS = [(True, True), (True, False), (False, False), (False, True)]
SHAPES = [(128, 128, 64), (256, 384, 128), (200, 136, 72), (64, 16, 1024), (1000, 48, 256), (8, 8, 8),
          (384, 1024, 4096), (130, 260, 200)]


@pytest.mark.parametrize("rk,sk", LAYOUTS)
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_exact_integer(M, N, K, rk, sk):
    if (not rk and M % 8) or (not sk and N % 8):
        pytest.skip("K-strided operand needs the contiguous dim % 8 == 0")
    Rm, Sm = _ints((M, K), seed=1), _ints((N, K), seed=2)
    R = _bf(Rm if rk else Rm.t())
    S = _bf(Sm if sk else Sm.t())
    ref = _ref_gemm(Rm, Sm)
    out32 = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk, out_f32=True)
    torch.cuda.synchronize()
    assert torch.equal(out32.double().cpu(), ref), f"max err {(out32.double().cpu() - ref).abs().max()}"
    if ref.abs().max() < 256:   # exactly representable in bf16
        out16 = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk)
        assert torch.equal(out16.double().cpu(), ref)


# the 256x256 tile with K-strided operands (two 128-column LDS images per operand tile): the token reductions dW = dY^T X of the
# Q-Formers (reference: autograd of nn.Linear, models/qformer.py:126-130) run the 8-phase loop on interior tiles from 224 workgroups
# on; edge tiles, K tails and the mixed layouts keep the 2-slot loop on the same LDS image
@pytest.mark.parametrize("M,N,K,split,rk,sk", [
    (768, 3072, 8192, 7, False, False), (2304, 768, 8192, 9, False, False), (1024, 4096, 4096, 4, False, False),
    (776, 3080, 1600, 5, False, False), (768, 3072, 8200, 7, False, False), (1024, 2048, 328, 8, False, False),
    (4096, 4096, 512, 1, True, False), (4096, 4096, 520, 1, False, True), (4104, 4096, 512, 1, False, False)])
def test_gemm_big_tile_k_strided_exact(M, N, K, split, rk, sk):
    Rm, Sm = _ints((M, K), seed=11), _ints((N, K), seed=12)
    R = _bf(Rm if rk else Rm.t())
    S = _bf(Sm if sk else Sm.t())
    ref = _ref_gemm(Rm, Sm)
    out32 = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk, out_f32=True, split_k=split)
    torch.cuda.synchronize()
    assert torch.equal(out32.double().cpu(), ref), f"max err {(out32.double().cpu() - ref).abs().max()}"


def test_gemm_big_tile_k_strided_stress_exact():
    """The K-strided 8-phase loop keeps three half tiles of LDS-DMA in flight behind counted waits: exact integer results, repeated
    under memory load from a second stream (a scheduling hazard shows up as rare wrong tiles), over 128 K tiles per workgroup."""
    M, N, K, split = 1024, 4096, 32768, 4
    Rm, Sm = _ints((M, K), seed=21), _ints((N, K), seed=22)
    R, S = _bf(Rm.t()), _bf(Sm.t())
    ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    for it in range(6):
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
        out = hip.gemm(R, S, r_kcontig=False, s_kcontig=False, out_f32=True, split_k=split)
        assert torch.equal(out, ref), f"iteration {it}: {(out - ref).abs().max().item()}"
    torch.cuda.synchronize()


@pytest.mark.parametrize("rk,sk", LAYOUTS)
def test_gemm_random_tolerance(rk, sk):
    M, N, K = 512, 768, 1024
    Rm, Sm = _randn((M, K), 3), _randn((N, K), 4, std=0.05)
    R = _bf(Rm if rk else Rm.t())
    S = _bf(Sm if sk else Sm.t())
    Rq = (R if rk else R.t()).float().cpu()
    Sq = (S if sk else S.t()).float().cpu()
    ref = _ref_gemm(Rq, Sq)
    out = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk, out_f32=True).double().cpu()
    assert (out - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-4


@pytest.mark.parametrize("M,N,K", [(200, 136, 256), (200, 132, 256), (8192, 4096, 192), (8200, 4104, 128), (4352, 4100, 64)])
def test_gemm_epilogues(M, N, K):
    """bias / residual / GELU second output / gelu' epilogues on both tile configurations (128x128 and, from 256
    workgroups on, 256x256), 16-byte pieces (N % 8 == 0) and 8-byte pieces (N % 8 == 4), interior and edge tiles."""
    Rm, Sm = _randn((M, K), 5), _randn((N, K), 6, std=0.1)
    R, S = _bf(Rm), _bf(Sm)
    bias = _randn((N,), 7).to(DEV)
    res = _bf(_randn((M, N), 8))
    aux = _bf(_randn((M, N), 9))
    base = 0.5 * (R.float() @ S.float().t()) + bias
    # bias + residual + gelu second output
    g = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    out = hip.gemm(R, S, alpha=0.5, bias=bias, residual=res, gelu_out=g)
    want = base + res.float()
    assert torch.allclose(out.float(), want, rtol=1e-2, atol=1e-2)
    assert torch.allclose(g.float(), torch.nn.functional.gelu(out.float()), rtol=1e-2, atol=1e-2)
    # multiply by gelu'(aux)
    out2 = hip.gemm(R, S, alpha=0.5, bias=bias, gelu_grad_aux=aux)
    a = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(a).sum().backward()
    assert torch.allclose(out2.float(), base * a.grad, rtol=2e-2, atol=2e-2)
    # f32 out with bias
    out3 = hip.gemm(R, S, alpha=0.5, bias=bias, out_f32=True)
    assert torch.allclose(out3, base, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rk,sk", [(True, True), (True, False)])
def test_gemm_second_pair_lora(rk, sk):
    M, N, K, r = 300, 264, 192, 16
    Rm, Sm = _ints((M, K), seed=1), _ints((N, K), seed=2)
    R2m, S2m = _ints((M, r), seed=3), _ints((N, r), seed=4)
    R, R2 = _bf(Rm), _bf(R2m)
    S = _bf(Sm if sk else Sm.t())
    S2 = _bf(S2m if sk else S2m.t())
    ref = _ref_gemm(Rm, Sm) + _ref_gemm(R2m, S2m)
    out = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk, R2=R2, S2=S2, out_f32=True)
    assert torch.equal(out.double().cpu(), ref)


@pytest.mark.parametrize("split", [2, 5, 16])
def test_gemm_splitk_tn(split):
    # dW = dY^T X : reduction over the token dim, both operands K-strided
    Mred, N, Kp = 1000, 136, 264
    dY, X = _ints((Mred, N), seed=1), _ints((Mred, Kp), seed=2)
    ref = dY.double().t() @ X.double()
    out = hip.gemm(_bf(dY), _bf(X), r_kcontig=False, s_kcontig=False, out_f32=True, split_k=split)
    assert torch.equal(out.double().cpu(), ref)
    out2 = hip.gemm(_bf(dY), _bf(X), r_kcontig=False, s_kcontig=False, out_f32=True, split_k=split)
    assert torch.equal(out, out2), "split-K reduction must be deterministic"


@pytest.mark.parametrize("rk,sk", [(True, True), (True, False), (False, False)])
def test_gemm_ring_stress_exact(rk, sk):
    """The LDS-DMA ring is ordered only by counted vmcnt waits + barriers: a scheduling bug shows up as
    rare wrong tiles under load.  Long K (128 ring tiles), many workgroups, repeated, two streams, exact."""
    M, N, K = 4096, 1024, 4096
    Rm, Sm = _ints((M, K), lo=-2, hi=3, seed=11), _ints((N, K), lo=-2, hi=3, seed=12)
    R = _bf(Rm if rk else Rm.t())
    S = _bf(Sm if sk else Sm.t())
    ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
    big = torch.randn(64 * 1024 * 1024, device=DEV)          # memory traffic beside the GEMMs
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()      # `big` may reuse memory the reference's temporaries just freed: s2 must not run ahead of them
    for it in range(6):
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
        out = hip.gemm(R, S, r_kcontig=rk, s_kcontig=sk, out_f32=True)
        assert torch.equal(out, ref), f"iteration {it}: {(out - ref).abs().max().item()}"
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(8192, 8192, 1024), (16384, 4096, 4096), (8192 + 256, 8192, 192 + 64), (12288, 6144, 1024 + 48)])
def test_gemm_8phase_stress_exact(M, N, K):
    """The 256x256 8-phase ping-pong loop (K-contiguous operands, >= 256 workgroups) keeps 7 half tiles of LDS-DMA in
    flight under counted vmcnt waits and two staggered wave groups: exact integer results, repeated under memory
    load from a second stream, f32 and bf16 outputs, K tails and the second (LoRA) K range handed to the generic tiles."""
    Rm, Sm = _ints((M, K), lo=-2, hi=3, seed=21), _ints((N, K), lo=-2, hi=3, seed=22)
    R, S = _bf(Rm), _bf(Sm)
    ref = (Rm.to(DEV).double() @ Sm.to(DEV).double().t()).float()
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()      # `big` may reuse memory the reference's temporaries just freed: s2 must not run ahead of them
    for it in range(4):
        with torch.cuda.stream(s2):
            big.mul_(1.0001)
        out = hip.gemm(R, S, out_f32=True)
        assert torch.equal(out, ref), f"iteration {it}: {(out - ref).abs().max().item()}"
    torch.cuda.synchronize()
    r = 16
    R2m, S2m = _ints((M, r), lo=-1, hi=2, seed=23), _ints((N, r), lo=-1, hi=2, seed=24)
    out16 = hip.gemm(R, S, R2=_bf(R2m), S2=_bf(S2m), alpha=1.0 / 64)
    ref2 = ((ref.double() + R2m.to(DEV).double() @ S2m.to(DEV).double().t()) / 64).to(torch.bfloat16)
    assert torch.equal(out16, ref2)


def test_gemm_rejects_bad_arguments():
    from unirec_amd._lib import UniRecHipError
    R, S = _bf(_randn((16, 12))), _bf(_randn((8, 12)))
    with pytest.raises(UniRecHipError):
        hip.gemm(R, S)   # K = 12 not a multiple of 8


@pytest.mark.parametrize("M,H", [(5, 128), (64, 256), (300, 768), (1000, 1024), (17, 2048)])
def test_layernorm_fwd_bwd(M, H):
    y, res = _bf(_randn((M, H), 1)), _bf(_randn((M, H), 2))
    gamma = (1 + 0.1 * _randn((H,), 3)).to(DEV)
    beta = (0.1 * _randn((H,), 4)).to(DEV)
    out, z, mean, rstd = hip.layernorm_fwd(y, gamma, beta, 1e-12, residual=res)
    zf = (y.float() + res.float()).to(torch.bfloat16).float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zf, (H,), gr, br, 1e-12)
    assert torch.equal(z.float(), zf.detach())
    assert torch.allclose(out.float(), ref, rtol=1e-2, atol=2e-2)
    dout = _bf(_randn((M, H), 5))
    ref.backward(dout.float())
    dg, db, dbias = (torch.empty(H, device=DEV) for _ in range(3))
    dz, dy = hip.layernorm_bwd(dout, z, mean, rstd, gamma, dg, db, dbias)
    assert dy.data_ptr() == dz.data_ptr()
    assert torch.allclose(dz.float(), zf.grad, rtol=2e-2, atol=2e-2)
    scale = math.sqrt(M)
    assert torch.allclose(dg, gr.grad, rtol=1e-2, atol=2e-2 * scale)
    assert torch.allclose(db, br.grad, rtol=1e-2, atol=2e-2 * scale)
    assert torch.allclose(dbias, dz.float().sum(0), rtol=1e-3, atol=1e-3 * scale)


def test_layernorm_broadcast_rows_and_dropout():
    Q, H, B = 8, 256, 5
    qe = _bf(_randn((Q, H), 1))
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    out, z, mean, rstd = hip.layernorm_fwd(qe, gamma, beta, 1e-12, M=B * Q)
    ref = torch.nn.functional.layer_norm(qe.float(), (H,))
    assert torch.allclose(out.float().view(B, Q, H), ref.expand(B, Q, H), rtol=1e-2, atol=2e-2)
    # dropout: mask is deterministic in (seed, index); kept elements are scaled by 1/(1-p)
    p = 0.25
    o1, _, _, _ = hip.layernorm_fwd(qe, gamma, beta, 1e-12, M=B * Q, p_post=p, seed_post=123)
    o2, _, _, _ = hip.layernorm_fwd(qe, gamma, beta, 1e-12, M=B * Q, p_post=p, seed_post=123)
    o3, _, _, _ = hip.layernorm_fwd(qe, gamma, beta, 1e-12, M=B * Q, p_post=p, seed_post=124)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    kept = o1.float() != 0
    frac = 1.0 - kept.float().mean().item()
    assert abs(frac - p) < 0.03
    assert torch.allclose(o1.float()[kept], (out.float() / (1 - p))[kept], rtol=2e-2, atol=2e-2)
    # pre-dropout + backward consistency: dy == dz * mask / (1-p)
    y, res = _bf(_randn((B * Q, H), 2)), _bf(_randn((B * Q, H), 3))
    _, z0, _, _ = hip.layernorm_fwd(y, gamma, beta, 1e-12, p_pre=p, seed_pre=77)     # no residual: z0 = dropout(y)
    dropped = (z0.float() == 0) & (y.float() != 0)
    assert abs(dropped.float().mean().item() - p) < 0.03
    o, z, mean, rstd = hip.layernorm_fwd(y, gamma, beta, 1e-12, residual=res, p_pre=p, seed_pre=77)
    assert torch.equal(z.float()[dropped], res.float()[dropped])
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    dz, dy = hip.layernorm_bwd(_bf(_randn((B * Q, H), 4)), z, mean, rstd, gamma, dg, db, p_pre=p, seed_pre=77)
    assert (dy.float()[dropped] == 0).all()
    assert torch.allclose(dy.float()[~dropped], dz.float()[~dropped] / (1 - p), rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("M,I,K", [(300, 256, 128), (16384, 1024, 64), (1000, 516, 72)])
def test_gemm_swiglu_backward_epilogue(M, I, K):
    """ur_gemm with swiglu_gu: the down-projection's dX GEMM hands d(act) to the SwiGLU backward in its epilogue (dgate | dup
    leave, d(act) is never stored) -- against fp32 torch and against the two-kernel path (GEMM, then ur_swiglu_bwd); both
    tile configurations, interior and edge tiles, 16- and 8-byte pieces."""
    dy, w = _bf(_randn((M, K), 1)), _bf(_randn((I, K), 2, 0.2))
    gu = _bf(_randn((M, 2 * I), 3))
    dgu = torch.full((M, 2 * I), float("nan"), device=DEV, dtype=torch.bfloat16)
    out = hip.gemm(dy, w, swiglu_bwd=(gu, dgu))
    assert out.data_ptr() == dgu.data_ptr()
    dact = dy.float() @ w.float().t()
    g, u = gu.float()[:, :I], gu.float()[:, I:]
    sg = torch.sigmoid(g)
    want = torch.cat([dact * u * (sg * (1 + g * (1 - sg))), dact * g * sg], dim=1)
    assert torch.isfinite(dgu.float()).all()
    err = (dgu.float() - want).norm() / want.norm()
    assert err < 6e-3, err
    if I % 8 == 0:       # (ur_swiglu_bwd's own constraint)
        two = hip.swiglu_bwd(hip.gemm(dy, w), gu, I)
        assert (two.float() - want).norm() / want.norm() < 1e-2
        assert (dgu.float() - two.float()).norm() / want.norm() < 1e-2


@pytest.mark.parametrize("M,I,K", [(300, 256, 128), (16384, 1024, 64), (1000, 516, 72)])
def test_gemm_swiglu_forward_epilogue(M, I, K):
    """ur_gemm with swiglu_gate: the up projection's epilogue reads the gate tile and writes act = silu(gate) * up beside up;
    identical to the two-kernel path bit for bit (the product uses the ROUNDED up in both), gate | up as column halves of
    one buffer (strided views) like the decoder calls it."""
    x, w = _bf(_randn((M, K), 1)), _bf(_randn((I, K), 2, 0.2))
    act = torch.full((M, I), float("nan"), device=DEV, dtype=torch.bfloat16)
    if I % 8 == 0:
        gu = torch.empty((M, 2 * I), device=DEV, dtype=torch.bfloat16)
        gu[:, :I] = _bf(_randn((M, I), 3))
        gate, upv = gu[:, :I], gu[:, I:]
    else:               # (column halves of one buffer would leave the up half 8-byte aligned only: separate tensors)
        gate, upv = _bf(_randn((M, I), 3)), torch.empty((M, I), device=DEV, dtype=torch.bfloat16)
    hip.gemm(x, w, out=upv, swiglu_fwd=(gate, act))
    up = hip.gemm(x, w)
    assert torch.equal(upv, up)
    want = torch.nn.functional.silu(gate.float()) * up.float()
    assert torch.isfinite(act.float()).all()
    assert (act.float() - want).norm() / want.norm() < 4e-3
    if I % 8 == 0:
        assert torch.equal(act, hip.swiglu_fwd(gu.contiguous(), I))


@pytest.mark.parametrize("p", [0.1, 0.2, 0.5])
def test_dropout_counter_hash_statistics(p):
    """The counter-based dropout generator (common.hip.h ur_hash2: keyed 32-bit murmur finaliser): drop rate, independence
    of neighbouring elements (along a row and across rows), and independence of the streams of neighbouring seeds and of
    seeds that differ only in their high word -- all within 5 sigma on 2^21 decisions."""
    M, H = 2048, 1024
    ones = torch.ones((M, H), device=DEV, dtype=torch.bfloat16)
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)

    def mask(seed):
        _, z0, _, _ = hip.layernorm_fwd(ones, gamma, beta, 1e-12, p_pre=p, seed_pre=seed)      # no residual: z0 = dropout(ones)
        return (z0.float() == 0)
    n = M * H
    sig1 = (p * (1 - p) / n) ** 0.5
    m0 = mask(1000)
    assert abs(m0.float().mean().item() - p) < 5 * sig1
    # joint drop probability of two independent decisions is p^2
    sig2 = (p * p * (1 - p * p) / n) ** 0.5
    for other in (torch.roll(m0, 1, dims=1), torch.roll(m0, 1, dims=0), torch.roll(m0, 7, dims=1), mask(1001), mask(1002), mask(1000 + (1 << 32)),
                  mask(1000 ^ 0x55555555)):
        both = (m0 & other).float().mean().item()
        assert abs(both - p * p) < 5 * sig2 + 1e-4, (both, p * p)
    # per-row and per-column drop rates show no structure
    assert (m0.float().mean(dim=1) - p).abs().max().item() < 6 * (p * (1 - p) / H) ** 0.5
    assert (m0.float().mean(dim=0) - p).abs().max().item() < 6 * (p * (1 - p) / M) ** 0.5


def test_batch_reduce_and_colsum():
    nb, rows, H = 37, 4, 64
    x = _bf(_randn((nb * rows, H), 1))
    out = hip.batch_reduce(x, nb, rows, H)
    assert torch.allclose(out, x.float().view(nb, rows, H).sum(0), rtol=1e-5, atol=1e-4)
    y = _bf(_randn((1000, 136), 2))
    assert torch.allclose(hip.colsum(y), y.float().sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("M,D", [(7, 128), (513, 1024), (64, 256)])
def test_rmsnorm_fwd_bwd(M, D):
    x = _bf(_randn((M, D), 1))
    w = (1 + 0.1 * _randn((D,), 2)).to(DEV)
    out, rstd = hip.rmsnorm_fwd(x, w, 1e-6)
    xf = x.float().requires_grad_(True)
    ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert torch.allclose(out.float(), ref, rtol=1e-2, atol=1e-2)
    dout, add = _bf(_randn((M, D), 3)), _bf(_randn((M, D), 4))
    ref.backward(dout.float())
    dx = hip.rmsnorm_bwd(dout, x, w, rstd, add=add)
    assert torch.allclose(dx.float(), xf.grad + add.float(), rtol=2e-2, atol=2e-2)


def test_elementwise_and_adamw():
    a32 = _randn((1000003,), 1).to(DEV)
    b = hip.cast_f32_to_bf16(a32)
    assert torch.equal(b, a32.to(torch.bfloat16))
    assert torch.equal(hip.cast_bf16_to_f32(b), b.float())
    x, y = _bf(_randn((4096,), 2)), _bf(_randn((4096,), 3))
    assert torch.equal(hip.add_bf16(x, y), (x.float() + y.float()).to(torch.bfloat16))
    M, I = 33, 96
    gu = _bf(_randn((M, 2 * I), 4))
    act = hip.swiglu_fwd(gu, I)
    guf = gu.float().requires_grad_(True)
    ref = torch.nn.functional.silu(guf[:, :I]) * guf[:, I:]
    assert torch.allclose(act.float(), ref, rtol=1e-2, atol=1e-2)
    d = _bf(_randn((M, I), 5))
    ref.backward(d.float())
    assert torch.allclose(hip.swiglu_bwd(d, gu, I).float(), guf.grad, rtol=2e-2, atol=2e-2)
    # AdamW vs torch.optim.AdamW, 3 steps
    n = 10007
    p0, gs = _randn((n,), 6).to(DEV), [_randn((n,), 10 + i).to(DEV) for i in range(3)]
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for i, g in enumerate(gs):
        pt.grad = g.clone()
        opt.step()
        hip.adamw_step(p, g * 4.0, m, v, 1e-2, 0.9, 0.999, 1e-8, 0.01, i + 1, grad_scale=0.25)
    assert torch.allclose(p, pt.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K", [(300, 128), (520, 200), (1111, 1024)])
def test_lora_kernels_with_dropout_bits(M, K):
    """The dedicated rank-16 LoRA kernels against torch, with the dropout bit planes the library generated:
    ur_lora_project (t = dropout_a(x) A_a^T, shared input), ur_lora_reduce (dA_a = tb_a^T dropout_a(x)) and
    ur_gemm's masked rank-r epilogue (dx = dy W + sum_a mask_a * (tb_a A_a))."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(3)
    p, seed, r = 0.3, 1234, 16
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    for nad in (1, 3):
        A = (torch.randn(nad * r, K, generator=g) * 0.2).to(DEV).to(torch.bfloat16)
        bits = hip.lora_dropout_bits(seed + nad, p, M, K, nad, DEV)
        keep = hip.lora_bits_to_keep(bits, K).float()                       # [nad, M, K]
        assert abs(keep.mean().item() - (1 - p)) < 0.02
        if nad > 1:
            assert not torch.equal(keep[0], keep[1])                        # one mask per adapter
        again = hip.lora_dropout_bits(seed + nad, p, M, K, nad, DEV)
        assert torch.equal(bits, again)                                     # a pure function of (seed, p, shape)
        got = hip.lora_project(x, [A[a * r:(a + 1) * r] for a in range(nad)], alpha=1.0 / (1 - p), bits=bits).float()
        want = torch.cat([(x.float() * keep[a] / (1 - p)) @ A[a * r:(a + 1) * r].float().t() for a in range(nad)], 1)
        assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
        tb = torch.randn(M, nad * r, generator=g).to(DEV).to(torch.bfloat16)
        gA = torch.empty(nad * r, K, device=DEV)
        hip.lora_reduce(x, tb, gA, nad=nad, alpha=1.0 / (1 - p), bits=bits)
        want = torch.cat([tb[:, a * r:(a + 1) * r].float().t() @ (x.float() * keep[a] / (1 - p)) for a in range(nad)], 0)
        assert (gA - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
        # masked epilogue of the dx GEMM
        Nout = 136
        dy = torch.randn(M, Nout, generator=g).to(DEV).to(torch.bfloat16)
        WT = (torch.randn(K, Nout, generator=g) * 0.1).to(DEV).to(torch.bfloat16)          # [in, out] = transposed weight
        got = hip.gemm(dy, WT, R2=tb, S2=hip.transpose_bf16(A), drop=(bits, p, r)).float()
        want = dy.float() @ WT.float().t()
        for a in range(nad):
            want = want + keep[a] / (1 - p) * (tb[:, a * r:(a + 1) * r].float() @ A[a * r:(a + 1) * r].float())
        assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 2e-2
    # p = 0: nothing is dropped
    assert hip.lora_bits_to_keep(hip.lora_dropout_bits(1, 0.0, 64, K, 2, DEV), K).min().item() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,nad,p", [(1024, 128, 1, 0.3), (4096, 1024, 3, 0.1), (2048, 192, 2, 0.5), (8192, 2048, 1, 0.1), (1024, 256, 4, 0.2), (1024, 128, 2, 0.0)])
def test_lora_reduce_ring_kernel(M, K, nad, p):
    """Token counts / widths that are multiples of 128 / 64 take the LDS-DMA ring kernel (csrc/lora.hip: lora_reduce_ring_kernel): dA_a =
    tb_a^T dropout_a(x) with the TOKEN-packed flags (ur_lora_bits_transpose) masking the transposed fragments, against an f32 product of
    the same bf16 inputs and the same flags, and against the register-staged kernel (no bits_t: same flags, other summation order);
    the unpacked transposed flags equal the row planes bit for bit; deterministic.  p = 0: the unmasked instantiation."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(M + K + nad)
    r = 16
    x = torch.randn(M, K + 8, generator=g).to(DEV).to(torch.bfloat16)[:, :K]
    tb = torch.randn(M, nad * r, generator=g).to(DEV).to(torch.bfloat16)
    bits = hip.lora_dropout_bits(77 + nad, p, M, K, nad, DEV) if p > 0 else None
    keep = hip.lora_bits_to_keep(bits, K).float() if p > 0 else torch.ones((nad, M, K), device=DEV)
    bt = hip.lora_bits_transpose(bits, K) if p > 0 else None
    if p > 0:
        # word (tg, c): byte g = tokens 32 tg + 8 g .. + 7, bit i (i < 4) = token 8g + 2i, bit 4 + i = token 8g + 2i + 1
        w = bt[:, :, :K].to(torch.int64) & 0xffffffff
        pos = torch.tensor([8 * (rr >> 3) + ((rr & 7) >> 1) + 4 * (rr & 1) for rr in range(32)], device=DEV)
        dropped_t = ((w.unsqueeze(-1) >> pos) & 1).permute(0, 1, 3, 2).reshape(nad, M, K)
        assert torch.equal(dropped_t.float(), 1.0 - keep)
    gA = torch.full((nad * r, K), float("nan"), device=DEV)
    hip.lora_reduce(x, tb, gA, nad=nad, alpha=1.0 / (1 - p), bits=bits, bits_t=bt)
    want = torch.cat([tb[:, a * r:(a + 1) * r].float().t() @ (x.float() * keep[a] / (1 - p)) for a in range(nad)], 0)
    assert (gA - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 2e-3
    old = torch.empty_like(gA)
    hip.lora_reduce(x, tb, old, nad=nad, alpha=1.0 / (1 - p), bits=bits)
    assert (gA - old).abs().max().item() <= 1e-4 * want.abs().max().item() + 2e-3
    again = torch.empty_like(gA)
    hip.lora_reduce(x, tb, again, nad=nad, alpha=1.0 / (1 - p), bits=bits, bits_t=bt)
    assert torch.equal(gA, again)


@pytest.mark.gpu
def test_lora_row_products_do_not_depend_on_the_row_position():
    """Per-token outputs of the ring kernels (t of ur_lora_project, tb of ur_lora_bgrad) are bit-identical wherever the row sits in
    the launch: the shard invariance of the forward and the per-sample invariance of the backward (full-size additivity test) rest on
    it -- a chunk order that varies per workgroup (tried: rotated orders against memory-channel camping) breaks it."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(21)
    M, K, r = 4096, 2048, 16
    x = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    A = (torch.randn(r, K, generator=g) * 0.2).to(DEV).to(torch.bfloat16)
    bits = hip.lora_dropout_bits(5, 0.1, M, K, 1, DEV)
    full = hip.lora_project(x, [A], alpha=1.1, bits=bits)
    for lo, hi in ((1024, 2048), (2304, 4096)):
        part = hip.lora_project(x[lo:hi], [A], alpha=1.1, bits=bits[:, lo:hi])
        assert torch.equal(part, full[lo:hi])
    cols = [(0, 1024), (1024, 512), (1536, 512)]
    Bt = [(torch.randn(r, n, generator=g) * 0.2).to(DEV).to(torch.bfloat16) for _, n in cols]
    t = torch.randn(M, 3 * r, generator=g).to(DEV).to(torch.bfloat16)
    gB = torch.empty(K, r, device=DEV)
    tb = hip.lora_bgrad(x, t, Bt, cols, gB)
    for lo, hi in ((1024, 2048), (2304, 4096)):
        tbp = hip.lora_bgrad(x[lo:hi], t[lo:hi], Bt, cols, torch.empty_like(gB))
        assert torch.equal(tbp, tb[lo:hi])


@pytest.mark.gpu
def test_lora_reduce_ring_kernel_on_column_ranges():
    """dB_a = dy_a^T t_a over per-adapter column ranges (shared = 0, transposed output, no dropout) through the ring kernel."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(9)
    M, r = 2048, 16
    cols = [(0, 256), (256, 128), (384, 64)]
    Wt = 448
    dy = torch.randn(M, Wt + 8, generator=g).to(DEV).to(torch.bfloat16)[:, :Wt]
    t = torch.randn(M, 3 * r, generator=g).to(DEV).to(torch.bfloat16)
    gB = torch.full((Wt, r), float("nan"), device=DEV)
    hip.lora_reduce(dy, t, gB, cols=cols, transposed=True)
    want = torch.cat([dy[:, c0:c0 + n].float().t() @ t[:, a * r:(a + 1) * r].float() for a, (c0, n) in enumerate(cols)], 0)
    assert (gB - want).abs().max().item() <= 1e-4 * want.abs().max().item() + 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("M,cols", [(200, [(0, 64)]), (1024, [(0, 128), (128, 64), (256, 192)]), (4100, [(64, 256), (320, 128)]),
                                    (3000, [(0, 2048), (2048, 1024), (3072, 1024)]), (88000 + 200, [(0, 256), (256, 128), (384, 64)])])
def test_lora_bgrad_ring_kernel(M, cols):
    """Widths that are multiples of 64 take the LDS-DMA ring kernel (csrc/lora.hip: lora_bgrad_ring_kernel): tb = s * dy_a B_a and
    dB_a = dy_a^T t_a against fp32 products of the same bf16 inputs; ragged token counts (partial 512-token block, partial
    256-token tile), column ranges that do not start at 0, ld != width; deterministic.  The last case has enough tokens for the
    1024-token blocks (one block per CU at least)."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(M)
    r = 16
    Wt = max(c0 + n for c0, n in cols)
    dy = torch.randn(M, Wt + 8, generator=g).to(DEV).to(torch.bfloat16)[:, :Wt]
    B = [(torch.randn(n, r, generator=g) * 0.2).to(DEV).to(torch.bfloat16) for _, n in cols]
    t = torch.randn(M, len(cols) * r, generator=g).to(DEV).to(torch.bfloat16)
    ntot = sum(n for _, n in cols)
    gB = torch.full((ntot, r), float("nan"), device=DEV)
    tb = hip.lora_bgrad(dy, t, [hip.transpose_bf16(b) for b in B], cols, gB, alpha=0.5).float()
    want_tb = torch.cat([0.5 * dy[:, c0:c0 + n].float() @ B[a].float() for a, (c0, n) in enumerate(cols)], 1)
    want_gB = torch.cat([dy[:, c0:c0 + n].float().t() @ t[:, a * r:(a + 1) * r].float() for a, (c0, n) in enumerate(cols)], 0)
    # tb is rounded to bf16 once (2^-9 relative); dB is an f32 sum of exact bf16 x bf16 products
    assert (tb - want_tb).abs().max().item() <= 6e-3 * want_tb.abs().max().item() + 1e-3
    assert (gB - want_gB).abs().max().item() <= 1e-4 * want_gB.abs().max().item() + 1e-3
    gB2 = torch.empty_like(gB)
    tb2 = hip.lora_bgrad(dy, t, [hip.transpose_bf16(b) for b in B], cols, gB2, alpha=0.5).float()
    assert torch.equal(gB, gB2) and torch.equal(tb, tb2)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [200, 4100])
def test_lora_kernels_on_column_ranges(M):
    """Adapters that own column ranges of one activation (backward): tb_a = s * dy_a B_a (ur_lora_project, no
    dropout) and dB_a = dy_a^T t_a (ur_lora_reduce, transposed dense output), plus the no-dropout shared forms."""
    from unirec_amd import hip
    g = torch.Generator().manual_seed(5)
    r = 16
    cols = [(0, 256), (256, 128), (384, 72)]
    Wt = 456
    dy = torch.randn(M, Wt + 8, generator=g).to(DEV).to(torch.bfloat16)[:, :Wt]             # ld != width
    B = [(torch.randn(n, r, generator=g) * 0.2).to(DEV).to(torch.bfloat16) for _, n in cols]
    t = torch.randn(M, 3 * r, generator=g).to(DEV).to(torch.bfloat16)
    tb = hip.lora_project(dy, [hip.transpose_bf16(b) for b in B], cols=cols, alpha=2.0).float()
    want = torch.cat([2.0 * dy[:, c0:c0 + n].float() @ B[a].float() for a, (c0, n) in enumerate(cols)], 1)
    assert (tb - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
    gB = torch.empty(Wt, r, device=DEV)
    hip.lora_reduce(dy, t, gB, cols=cols, transposed=True)
    want = torch.cat([dy[:, c0:c0 + n].float().t() @ t[:, a * r:(a + 1) * r].float() for a, (c0, n) in enumerate(cols)], 0)
    assert (gB - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
    again = torch.empty_like(gB)
    hip.lora_reduce(dy, t, again, cols=cols, transposed=True)
    assert torch.equal(gB, again)                                                          # deterministic token split
    # the fused B-side backward (one pass over dy) gives the same two results
    gB2 = torch.empty_like(gB)
    tb2 = hip.lora_bgrad(dy, t, [hip.transpose_bf16(b) for b in B], cols, gB2, alpha=2.0).float()
    want_tb = torch.cat([2.0 * dy[:, c0:c0 + n].float() @ B[a].float() for a, (c0, n) in enumerate(cols)], 1)
    assert (tb2 - want_tb).abs().max().item() <= 2e-2 * want_tb.abs().max().item() + 1e-2
    assert (gB2 - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
    gB3 = torch.empty_like(gB)
    hip.lora_bgrad(dy, t, [hip.transpose_bf16(b) for b in B], cols, gB3, alpha=2.0)
    assert torch.equal(gB2, gB3)
    # shared input, no dropout
    x = dy[:, :200]
    A = (torch.randn(2 * r, 200, generator=g) * 0.2).to(DEV).to(torch.bfloat16)
    got = hip.lora_project(x, [A[:r], A[r:]]).float()
    want = x.float() @ A.float().t()
    assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2
    gA = torch.empty(2 * r, 200, device=DEV)
    hip.lora_reduce(x, t[:, :2 * r], gA, nad=2)
    want = t[:, :2 * r].float().t() @ x.float()
    assert (gA - want).abs().max().item() <= 2e-2 * want.abs().max().item() + 1e-2


def test_fused_adamw_skips_untouched_tensors_like_torch_and_resumes():
    """torch.optim.AdamW leaves parameters with grad=None alone (no decay, no moment update, own step count);
    FusedAdamW must do the same for tensors no backward published, and its state must round-trip for resume."""
    from unirec_amd.optim import FusedAdamW
    from unirec_amd.packing import ParamPack
    torch.manual_seed(0)
    names = ["a.weight", "a.bias", "head.weight", "head.bias", "b.weight"]
    shapes = [(64, 32), (64,), (16, 64), (16,), (8, 8)]
    ps = [(n, torch.nn.Parameter(torch.randn(s))) for n, s in zip(names, shapes)]
    ref = [torch.nn.Parameter(p.detach().clone().to(DEV)) for _, p in ps]
    pack = ParamPack(ps, DEV)
    opt = FusedAdamW([pack], lr=1e-2, weight_decay=0.1)
    topt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.1)
    plan = [["a.weight", "a.bias", "b.weight"], names, ["head.weight", "head.bias"], ["a.weight", "b.weight"], names]
    for it, live in enumerate(plan):
        opt.zero_grad(); topt.zero_grad(set_to_none=True)
        for n, r in zip(names, ref):
            if n in live:
                g = torch.randn_like(r)
                pack.g32(n).copy_(g)
                r.grad = g.clone()
        pack.publish_grads(live)
        assert all((pack.params[n].grad is None) == (n not in live) for n in names)
        opt.step(); topt.step()
        for n, r in zip(names, ref):
            torch.testing.assert_close(pack.w32(n), r.detach(), rtol=2e-6, atol=2e-7, msg=f"step {it}: {n}")
        if it == 2:            # resume: a fresh optimizer loaded from the state continues identically
            sd = opt.state_dict()
            opt = FusedAdamW([pack], lr=1.0, weight_decay=0.0)
            opt.load_state_dict(sd)
    assert opt.steps[0]["head.weight"] == 3 and opt.steps[0]["a.weight"] == 4 and opt.steps[0]["a.bias"] == 3


def test_load_base_weights_keeps_the_added_special_rows():
    """a real Qwen3 checkpoint has the BASE vocabulary; loading it after the table was extended must fill the
    first rows, keep the special rows and refresh the frozen bf16 copies."""
    from unirec_amd.joint import MultiModalQwenEmbedding
    from unirec_amd.qformer_model import QFormerForItemRepresentation
    from unirec_amd.qwen3 import Qwen3Config, Qwen3LoRAModel
    cfg = Qwen3Config(vocab_size=100, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=128)
    qf = QFormerForItemRepresentation(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                                      num_query_tokens=2, field_embedding_dim=64, num_fields=5, dropout=0.0)
    m = MultiModalQwenEmbedding(qformer_model=qf, use_lora=True, qwen_config=cfg, num_history_items=3, num_query_tokens_per_item=2).to(DEV).eval()
    assert m.base_model.embed_tokens.weight.shape[0] == 106
    ids = torch.randint(0, 100, (2, 16), device=DEV)
    with torch.no_grad():
        before = m.base_model.forward_pooled(ids)                  # builds the frozen bf16 copies
    torch.manual_seed(1)
    donor = Qwen3LoRAModel(Qwen3Config(vocab_size=100, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                       num_key_value_heads=2, head_dim=128), use_lora=False)
    sd = {"model." + k: v for k, v in donor.state_dict().items()}
    with pytest.raises(RuntimeError):
        m.base_model.load_state_dict({k[6:]: v for k, v in sd.items()}, strict=False)       # the documented failure mode
    special = m.base_model.embed_tokens.weight[100:].detach().clone()
    missing, unexpected = m.load_base_weights(sd)
    assert not missing and not unexpected
    assert torch.equal(m.base_model.embed_tokens.weight[:100].cpu(), donor.embed_tokens.weight.detach())
    assert torch.equal(m.base_model.embed_tokens.weight[100:], special)
    with torch.no_grad():
        after = m.base_model.forward_pooled(ids)
    assert not torch.equal(before, after)                          # stale frozen copies would give `before` again
    donor = donor.to(DEV).eval()
    with torch.no_grad():
        want = donor.forward_pooled(ids)
    # LoRA B is zero-initialised: the adapter contributes nothing, so the loaded model reproduces the donor
    torch.testing.assert_close(after, want, rtol=2e-2, atol=2e-3)


def test_lora_dropout_planes_of_different_sites_are_unrelated():
    """the masks of different (layer, adapter group, step) must be independent draws -- not XOR-permuted
    copies of one stream (v_proj's plane used to equal gate_proj's, and layer i+1 was layer i with its 32-column groups
    swapped).  Two independent Bernoulli(0.9) keep masks agree on 0.9^2 + 0.1^2 = 0.82 of the positions."""
    from unirec_amd.qwen3 import Qwen3Config, Qwen3LoRAModel
    m = Qwen3LoRAModel(Qwen3Config(num_hidden_layers=2, vocab_size=64), use_lora=True)
    M, W, p = 512, 1024, 0.1

    def keep(step, layer, group, nad):
        bits = hip.lora_dropout_bits(m.lora_dropout_seed(step, layer, group), p, M, W, nad, DEV)
        return hip.lora_bits_to_keep(bits, W).float()
    a = keep(0, 0, 0, 3)            # q|k|v planes of layer 0
    g = keep(0, 0, 2, 2)            # gate|up planes of layer 0
    b = keep(0, 1, 0, 3)            # q|k|v planes of layer 1
    c = keep(1, 0, 0, 3)            # next step
    assert abs(a.mean().item() - 0.9) < 5e-3

    def agree(x, y):
        return (x == y).float().mean().item()
    ind = 0.82
    assert abs(agree(a[2], g[0]) - ind) < 0.01            # v_proj vs gate_proj (same M, same W): were identical
    assert abs(agree(a[0], b[0]) - ind) < 0.01 and abs(agree(a[0], c[0]) - ind) < 0.01 and abs(agree(a[0], a[1]) - ind) < 0.01
    # layer 1 against layer 0 with neighbouring 32-column groups swapped, and with any single XOR of the group index
    a0, b0 = a[0].view(M, W // 32, 32), b[0].view(M, W // 32, 32)
    for x in (1, 2, 3, 4, 8, 16, 31):
        perm = torch.arange(W // 32, device=DEV) ^ x
        assert abs(agree(a0[:, perm], b0) - ind) < 0.01, x
    assert torch.equal(keep(0, 0, 0, 3), a)               # and a pure function of (seed, step, layer, group)


def test_lora_dropout_plane_is_pairwise_independent_inside_a_word():
    """The 32 decisions of one (row, 32-column group) word come from 16 consecutive states of one small generator: every pair
    of columns inside the group, neighbouring groups and neighbouring rows must still agree like independent draws."""
    M, W, p = 4096, 1024, 0.1
    keep = hip.lora_bits_to_keep(hip.lora_dropout_bits(987654321, p, M, W, 1, DEV), W)[0].float()      # [M, W]
    assert abs(keep.mean().item() - 0.9) < 2e-3
    per_col = keep.mean(0)
    assert float((per_col - 0.9).abs().max()) < 0.025                     # 4096 draws per column: sigma 0.0047
    k3 = keep.view(M, W // 32, 32)
    d = (1.0 - k3).reshape(-1, 32)                                         # dropped flags, one row per word
    n = d.shape[0]
    co = (d.t() @ d) / n                                                   # P(both dropped) for every column pair
    off = co - torch.diag(torch.diag(co))
    # independent: p^2 = 0.01; sigma of the estimate over 131072 words = sqrt(0.01 * 0.99 / n) = 2.7e-4
    assert float((off + torch.eye(32, device=DEV) * 0.01 - 0.01).abs().max()) < 2e-3
    ind = 0.82
    assert abs((k3[:, 1:] == k3[:, :-1]).float().mean().item() - ind) < 5e-3        # same bit of neighbouring groups
    assert abs((keep[1:] == keep[:-1]).float().mean().item() - ind) < 5e-3          # neighbouring rows
    runs = (1.0 - keep)[:, 1:] * (1.0 - keep)[:, :-1]                               # adjacent columns both dropped
    assert abs(runs.mean().item() - 0.01) < 1e-3


@pytest.mark.parametrize("p", [0.1, 0.25])
def test_lora_dropout_word_popcounts_and_triples_follow_the_binomial(p):
    """The 32 decisions of a word are bit-sliced over consecutive states of one small generator (csrc/lora.hip): beyond
    pairs, the NUMBER of dropped columns per word must follow Binomial(32, p) and column triples / quadruples inside a word
    must be dropped together with probability p^3 / p^4 (peft's nn.Dropout draws are fully independent)."""
    M, W = 16384, 1024
    keep = hip.lora_bits_to_keep(hip.lora_dropout_bits(24681357, p, M, W, 1, DEV), W)[0]
    d = (1 - keep.view(-1, 32).to(torch.float64))                           # dropped flags, one row per word
    n = d.shape[0]
    cnt = d.sum(1)
    # popcount histogram against the binomial pmf: every bin with >= 200 expected words within 6 sigma
    hist = torch.bincount(cnt.long(), minlength=33).double().cpu()
    for k in range(33):
        pm = math.comb(32, k) * p ** k * (1 - p) ** (32 - k)
        if n * pm >= 200:
            assert abs(hist[k].item() - n * pm) <= 6 * math.sqrt(n * pm * (1 - pm)), (k, hist[k].item(), n * pm)
    assert abs(cnt.mean().item() - 32 * p) < 0.02 and abs(cnt.var().item() - 32 * p * (1 - p)) < 0.05 * 32 * p * (1 - p)
    # triples and quadruples of columns inside a word (neighbours, strided, scattered)
    g = torch.Generator().manual_seed(5)
    sets = [(0, 1, 2), (3, 4, 5), (0, 8, 16), (1, 9, 17), (5, 13, 30), (29, 30, 31), (0, 15, 31), (2, 11, 23)]
    sets += [tuple(sorted(torch.randperm(32, generator=g)[:3].tolist())) for _ in range(24)]
    for cs in sets:
        pr = p ** len(cs)
        obs = d[:, list(cs)].prod(1).sum().item()
        assert abs(obs - n * pr) <= 6 * math.sqrt(n * pr) + 1, (cs, obs, n * pr)
    for cs in [(0, 1, 2, 3), (4, 12, 20, 28), (7, 8, 9, 31), (1, 6, 18, 27)]:
        pr = p ** 4
        obs = d[:, list(cs)].prod(1).sum().item()
        assert abs(obs - n * pr) <= 6 * math.sqrt(n * pr) + 3, (cs, obs, n * pr)


def test_batched_transpose_matches_per_matrix_transposes():
    g = torch.Generator().manual_seed(3)
    shapes = [(16, 1024), (48, 1024), (2048, 16), (33, 70), (1, 5), (3072, 16), (32, 32)]
    srcs = [torch.randn(s, generator=g).to(DEV).to(torch.bfloat16) for s in shapes]
    bt = hip.BatchedTranspose(srcs)
    outs = bt.run()
    for s_, o in zip(srcs, outs):
        assert tuple(o.shape) == (s_.shape[1], s_.shape[0]) and torch.equal(o, s_.t().contiguous())
    srcs[3].mul_(2)                      # sources are live views: a second run picks the new values up
    assert torch.equal(bt.run()[3], srcs[3].t().contiguous())


@pytest.mark.parametrize("nad,p", [(3, 0.1), (2, 0.1), (3, 0.0), (2, 0.0)])
@pytest.mark.parametrize("M", [64, 1000, 4133])
def test_fused_rmsnorm_lora_projection_matches_the_two_kernels(nad, p, M):
    """ur_rmsnorm_lora_fwd == ur_rmsnorm_fwd followed by ur_lora_project over its output (same masks), ragged M included."""
    g = torch.Generator().manual_seed(M + nad)
    D = 1024
    x = (torch.randn(M, D, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
    w = (1.0 + 0.1 * torch.randn(D, generator=g)).to(DEV)
    U = [(torch.randn(16, D, generator=g) * 0.05).to(DEV).to(torch.bfloat16) for _ in range(nad)]
    bits = hip.lora_dropout_bits(1234, p, M, D, nad, DEV) if p > 0 else None
    alpha = 2.0 / (1.0 - p)
    h0, r0 = hip.rmsnorm_fwd(x, w, 1e-6)
    t0 = hip.lora_project(h0, U, alpha=alpha, bits=bits)
    h1, r1, t1 = hip.rmsnorm_lora_fwd(x, w, 1e-6, U, alpha=alpha, bits=bits)
    torch.testing.assert_close(r1, r0, rtol=2e-6, atol=0)
    # h: identical up to one bf16 ulp where the two reduction orders of the sum of squares round rstd differently
    dh = (h1.float() - h0.float()).abs()
    assert float((dh > 0).float().mean()) < 0.02 and float((dh / (h0.float().abs() + 1e-6)).max()) < 1e-2
    ref = torch.stack([((h1.float() * (hip.lora_bits_to_keep(bits, D)[a].float() if bits is not None else 1.0)) @ U[a].float().t()) * alpha
                       for a in range(nad)], 1).reshape(M, 16 * nad)
    assert float((t1.float() - ref).norm() / ref.norm()) < 1e-2
    assert float((t1.float() - t0.float()).norm() / t0.float().norm()) < 1e-2


@pytest.mark.parametrize("p", [0.1, 0.0])
@pytest.mark.parametrize("M,I", [(64, 3072), (1000, 3072), (4133, 512), (37, 128)])
def test_fused_swiglu_lora_projection_matches_the_two_kernels(p, M, I):
    """ur_swiglu_lora_fwd == ur_swiglu_fwd followed by ur_lora_project over act (same masks): act bit-identical, t to bf16 rounding."""
    g = torch.Generator().manual_seed(M + I)
    gu = (torch.randn(M, 2 * I, generator=g) * 1.5).to(DEV).to(torch.bfloat16)
    U = (torch.randn(16, I, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    bits = hip.lora_dropout_bits(77, p, M, I, 1, DEV) if p > 0 else None
    alpha = 2.0 / (1.0 - p)
    act0 = hip.swiglu_fwd(gu, I)
    t0 = hip.lora_project(act0, [U], alpha=alpha, bits=bits)
    act1, t1 = hip.swiglu_lora_fwd(gu, I, U, alpha=alpha, bits=bits)
    assert torch.equal(act0, act1)
    keep = hip.lora_bits_to_keep(bits, I)[0].float() if bits is not None else 1.0
    ref = ((act0.float() * keep) @ U.float().t()) * alpha
    assert float((t1.float() - ref).norm() / ref.norm()) < 1e-2
    assert float((t1.float() - t0.float()).norm() / t0.float().norm()) < 1e-2


@pytest.mark.parametrize("B,Q,F,E", [(5, 32, 14, 256), (300, 32, 14, 1024), (7, 32, 16, 384), (6, 4, 8, 256), (3, 32, 14, 100)])
def test_field_projection_fwd_bwd_match_torch(B, Q, F, E):
    """ur_field_projection_fwd/bwd (models/qformer_utils.py:54: Linear(Q -> F) over the query axis) against torch fp32 -- the
    Q = 32 fast paths (MFMA weight gradient) and the generic kernels (Q = 4; E = 100 is not a multiple of 4)."""
    g = torch.Generator().manual_seed(B * 1000 + E)
    rec = (torch.randn(B, Q, E, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    W = (torch.randn(F, Q, generator=g) * 0.2).to(DEV)
    b = (torch.randn(F, generator=g) * 0.1).to(DEV)
    out = hip.field_projection_fwd(rec, W, b)
    want = torch.einsum("fq,bqe->bfe", W, rec.float()) + b[None, :, None]
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    dout = (torch.randn(B, F, E, generator=g) * 0.3).to(DEV)
    dW, db = torch.empty(F, Q, device=DEV), torch.empty(F, device=DEV)
    drec = hip.field_projection_bwd(dout, rec, W, dW, db)
    want_drec = torch.einsum("fq,bfe->bqe", W, dout)
    assert float((drec.float() - want_drec).norm() / want_drec.norm()) < 4e-3              # bf16 output
    want_dW = torch.einsum("bfe,bqe->fq", dout, rec.float())
    assert float((dW - want_dW).norm() / want_dW.norm()) < 4e-3                              # fast path: dout rounded to bf16
    torch.testing.assert_close(db, dout.sum((0, 2)), rtol=1e-4, atol=1e-3)
    dW2, db2 = torch.empty_like(dW), torch.empty_like(db)
    drec2 = hip.field_projection_bwd(dout, rec, W, dW2, db2)
    assert torch.equal(dW, dW2) and torch.equal(db, db2) and torch.equal(drec, drec2)        # bitwise reproducible


def test_dropout_counters_follow_the_global_sample_index():
    """every dropout site keys its mask on (seed, GLOBAL element index).  A shard that starts at global sample b1
    and passes that offset draws bit for bit the masks (and results) of the same samples inside the whole batch."""
    g = torch.Generator().manual_seed(9)
    # LoRA bit planes: rows
    full = hip.lora_dropout_bits(4242, 0.1, 12, 1024, 3, DEV)
    part = hip.lora_dropout_bits(4242, 0.1, 8, 1024, 3, DEV, row0=4)
    assert torch.equal(part, full[:, 4:12]) and not torch.equal(hip.lora_dropout_bits(4242, 0.1, 8, 1024, 3, DEV), full[:, 4:12])
    # LayerNorm with hidden dropout before and after: rows
    M, H, r1 = 96, 256, 40
    y = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
    res = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
    gam, bet = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    o_full, z_full, _, _ = hip.layernorm_fwd(y, gam, bet, 1e-12, residual=res, p_pre=0.2, seed_pre=7, p_post=0.2, seed_post=8)
    o_part, z_part, mu, rs = hip.layernorm_fwd(y[r1:].contiguous(), gam, bet, 1e-12, residual=res[r1:].contiguous(), p_pre=0.2, seed_pre=7,
                                              p_post=0.2, seed_post=8, drop_row0=r1)
    assert torch.equal(o_part, o_full[r1:]) and torch.equal(z_part, z_full[r1:])
    dout = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
    _, zf, muf, rsf = hip.layernorm_fwd(y, gam, bet, 1e-12, residual=res, p_pre=0.2, seed_pre=7, p_post=0.2, seed_post=8)
    dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    dz_full, dy_full = hip.layernorm_bwd(dout, zf, muf, rsf, gam, dg, db, p_pre=0.2, seed_pre=7, p_post=0.2, seed_post=8)
    dz_part, dy_part = hip.layernorm_bwd(dout[r1:].contiguous(), z_part, mu, rs, gam, dg, db, p_pre=0.2, seed_pre=7, p_post=0.2, seed_post=8,
                                         drop_row0=r1)
    assert torch.equal(dz_part, dz_full[r1:]) and torch.equal(dy_part, dy_full[r1:])
    # attention-probability dropout (non-causal Q-Former attention): batch rows
    B, Sq, Sk, nh, hd, b1 = 6, 32, 50, 4, 64, 2
    q = torch.randn(B, Sq, nh, hd, generator=g).to(DEV).to(torch.bfloat16)
    k = torch.randn(B, Sk, nh, hd, generator=g).to(DEV).to(torch.bfloat16)
    v = torch.randn(B, Sk, nh, hd, generator=g).to(DEV).to(torch.bfloat16)
    of, cf = hip.attn_fwd(q, k, v, causal=False, dropout_p=0.3, seed=99)
    op, cp = hip.attn_fwd(q[b1:].contiguous(), k[b1:].contiguous(), v[b1:].contiguous(), causal=False, dropout_p=0.3, seed=99, drop_batch0=b1)
    assert torch.equal(op, of[b1:])
    do = torch.randn(B, Sq, nh, hd, generator=g).to(DEV).to(torch.bfloat16)
    dqf, dkf, dvf = hip.attn_bwd(cf, do)
    dqp, dkp, dvp = hip.attn_bwd(cp, do[b1:].contiguous())
    assert torch.equal(dqp, dqf[b1:]) and torch.equal(dkp, dkf[b1:]) and torch.equal(dvp, dvf[b1:])
    # user-sequence assembly (positional-encoding dropout): users
    Bu, L, Qi, Hh = 5, 6, 2, 128
    tok = torch.randn(Bu, L, Qi, Hh, generator=g).to(DEV).to(torch.bfloat16)
    ctx = torch.randn(Bu, L, Hh, generator=g).to(DEV).to(torch.bfloat16)
    lens = torch.tensor([6, 3, 5, 1, 6], dtype=torch.int32, device=DEV)
    uf, _ = hip.user_sequence_assemble(tok, ctx, lens, 0.1, 31)
    up, _ = hip.user_sequence_assemble(tok[2:].contiguous(), ctx[2:].contiguous(), lens[2:].contiguous(), 0.1, 31, drop_batch0=2)
    assert torch.equal(up, uf[2:])
