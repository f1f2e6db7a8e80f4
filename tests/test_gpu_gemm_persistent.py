"""GPU: the persistent 256x256 projection GEMM (csrc/gemm_pers.hip) against the generic kernel (bit for bit: same
accumulation order, same roundings) and against exact integer references.  Replaces nn.Linear at
/root/reference/models/qformer.py:126-130 and installed modeling_qwen3.py:81-83,227-238 on the decoder's big launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from unirec_amd import hip  # noqa: E402

DEV = "cuda"


def _ints(shape, lo=-2, hi=3, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(torch.float32)


def _randn(shape, seed=0, std=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * std).to(DEV).to(torch.bfloat16).contiguous()


def _bf(x):
    return x.to(DEV).to(torch.bfloat16).contiguous()


def _both(fn):
    """fn() once on the generic kernel, once with the persistent kernel enabled."""
    prev = hip.gemm_persistent_mode(0)
    try:
        ref = fn()
        torch.cuda.synchronize()
        hip.gemm_persistent_mode(1)
        got = fn()
        torch.cuda.synchronize()
    finally:
        hip.gemm_persistent_mode(prev)
    return ref, got


# 512 tiles is the smallest eligible launch; 33 x 16 = 528 tiles leaves workgroups with 2 and 3 tiles (and no column chunks);
# K = 256 is the shortest K stream (4 K tiles), 320 an odd number of K tiles (the ring slot parity flips from tile to tile)
@pytest.mark.parametrize("M,N,K", [(8192, 4096, 256), (8448, 4096, 320), (8192, 4096, 1024), (4096, 8192, 576), (16384, 2048, 2048)])
def test_persistent_plain_is_exact_and_identical(M, N, K):
    Rm, Sm = _ints((M, K), seed=31), _ints((N, K), seed=32)
    R, S = _bf(Rm), _bf(Sm)
    ref = ((Rm.to(DEV).double() @ Sm.to(DEV).double().t()) / 8).to(torch.bfloat16)
    a, b = _both(lambda: hip.gemm(R, S, alpha=0.125))
    assert torch.equal(a, ref)
    assert torch.equal(b, ref), f"{(b.float() - ref.float()).abs().max().item()}"


@pytest.mark.parametrize("K2", [16, 32, 48, 64])
def test_persistent_second_k_range(K2):
    """The LoRA term t B^T as one more K tile of the stream; k chunks beyond K2 come from the zero word, so the row
    strides of t / B stay K2 (no padded copies)."""
    M, N, K = 8192, 4096, 512
    Rm, Sm = _ints((M, K), seed=41), _ints((N, K), seed=42)
    R2m, S2m = _ints((M, K2), lo=-1, hi=2, seed=43), _ints((N, K2), lo=-1, hi=2, seed=44)
    R, S, R2, S2 = _bf(Rm), _bf(Sm), _bf(R2m), _bf(S2m)
    ref = ((Rm.to(DEV).double() @ Sm.to(DEV).double().t() + R2m.to(DEV).double() @ S2m.to(DEV).double().t()) / 16).to(torch.bfloat16)
    a, b = _both(lambda: hip.gemm(R, S, R2=R2, S2=S2, alpha=1.0 / 16))
    assert torch.equal(a, ref)
    assert torch.equal(b, ref)
    # views with a wider row stride (the merged q|k|v launch reads t [M, 48] and a block-diagonal B [N, 48])
    R2w, S2w = _randn((M, 80), 45), _randn((N, 80), 46, 0.1)
    Rr, Sr = _randn((M, K), 47), _randn((N, K), 48, 0.05)
    a, b = _both(lambda: hip.gemm(Rr, Sr, R2=R2w[:, 8:8 + K2], S2=S2w[:, 16:16 + K2]))
    assert torch.equal(a, b)


def test_persistent_bias_and_residual_identical():
    M, N, K = 8192, 4096, 384
    R, S = _randn((M, K), 51), _randn((N, K), 52, 0.05)
    res = _randn((M, N), 53)
    bias = torch.randn(N, device=DEV)
    t, Bm = _randn((M, 16), 54), _randn((N, 16), 55, 0.1)
    for kw in (dict(residual=res), dict(bias=bias), dict(bias=bias, residual=res, alpha=0.5), dict(residual=res, R2=t, S2=Bm)):
        a, b = _both(lambda: hip.gemm(R, S, **kw))
        assert torch.equal(a, b), str(list(kw))
    ref = (R.float() @ S.float().t() + res.float())
    a, b = _both(lambda: hip.gemm(R, S, residual=res))
    assert (b.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(8192, 3072, 768), (16384, 3072, 768), (8192, 4096, 320)])
def test_persistent_gelu_epilogues_identical(M, N, K):
    """The Q-Former FFN's two launches (reference models/qformer.py:386-395 forward: dense + GELU; backward: dX times gelu'(u)):
    bias + GELU second output of the bf16-rounded pre-activation, and the product with gelu'(saved pre-activation)."""
    R, S = _randn((M, K), 56), _randn((N, K), 57, 0.05)
    bias = torch.randn(N, device=DEV)
    aux = _randn((M, N), 58, 1.5)

    def fwd():
        g = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        u = hip.gemm(R, S, bias=bias, gelu_out=g)
        return u, g
    (ua, ga), (ub, gb) = _both(fwd)
    assert torch.equal(ua, ub) and torch.equal(ga, gb)
    assert torch.allclose(gb.float(), torch.nn.functional.gelu(ub.float()), rtol=1e-2, atol=1e-2)
    for kw in (dict(), dict(bias=bias, alpha=0.5)):
        a, b = _both(lambda: hip.gemm(R, S, gelu_grad_aux=aux, **kw))
        assert torch.equal(a, b), str(list(kw))
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = (R.float() @ S.float().t()) * x.grad
    assert (b.float() - (0.5 * ref + bias * x.grad)).abs().max().item() <= 2e-2 * ref.abs().max().item()
    # column views of wider buffers (row strides that differ from the widths)
    wide_u, wide_g, wide_a = (torch.zeros(M, N + 256, device=DEV, dtype=torch.bfloat16) for _ in range(3))
    wide_a[:, 128:128 + N] = aux

    def fwd_views():
        hip.gemm(R, S, bias=bias, out=wide_u[:, 64:64 + N], gelu_out=wide_g[:, 192:192 + N])
        return wide_u.clone(), wide_g.clone()
    (ua, ga), (ub, gb) = _both(fwd_views)
    assert torch.equal(ua, ub) and torch.equal(ga, gb)
    a, b = _both(lambda: hip.gemm(R, S, gelu_grad_aux=wide_a[:, 128:128 + N]))
    assert torch.equal(a, b)


@pytest.mark.parametrize("nad,K", [(1, 512), (2, 512), (3, 512), (2, 6144)])      # (K = 6144: the gate|up dX launch of the C4 step, on the persistent kernel since round 5)
def test_persistent_masked_lora_epilogue_identical(nad, K):
    """dX under LoRA dropout: dy W + sum_a keep_a / (1 - p) * (tb_a A_a)."""
    M, N, r, p = 8192, 4096, 16, 0.1
    dy, Wt = _randn((M, K), 61), _randn((N, K), 62, 0.05)
    tb, At = _randn((M, nad * r), 63), _randn((N, nad * r), 64, 0.1)
    bits = hip.lora_dropout_bits(7, p, M, N, nad, DEV)
    a, b = _both(lambda: hip.gemm(dy, Wt, R2=tb, S2=At, drop=(bits, p, r)))
    assert torch.equal(a, b)
    plain = hip.gemm(dy, Wt)
    assert not torch.equal(a, plain)


@pytest.mark.parametrize("drop", [False, True])
def test_persistent_swiglu_backward_epilogue_identical(drop):
    M, I, K, r, p = 8192, 4096, 256, 16, 0.1
    dy, Wt = _randn((M, K), 71), _randn((I, K), 72, 0.05)
    gu = _randn((M, 2 * I), 73)
    tb, At = _randn((M, r), 74), _randn((I, r), 75, 0.1)
    bits = hip.lora_dropout_bits(9, p, M, I, 1, DEV) if drop else None

    def run():
        dgu = torch.empty_like(gu)
        hip.gemm(dy, Wt, R2=tb, S2=At, drop=(bits, p, r) if drop else None, swiglu_bwd=(gu, dgu))
        return dgu
    a, b = _both(run)
    assert torch.equal(a, b)


def test_persistent_stream_stress_exact():
    """The K stream crosses output tiles under counted vmcnt waits with epilogue stores in the queue: exact integer
    results, repeated under memory load from a second stream (a scheduling bug shows up as rare wrong tiles)."""
    M, N, K = 16384, 8192, 1024
    Rm, Sm = _ints((M, K), seed=81), _ints((N, K), seed=82)
    R, S = _bf(Rm), _bf(Sm)
    ref = ((Rm.to(DEV).double() @ Sm.to(DEV).double().t()) / 32).to(torch.bfloat16)
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    prev = hip.gemm_persistent_mode(1)
    try:
        for it in range(6):
            with torch.cuda.stream(s2):
                big.mul_(1.0001)
            out = hip.gemm(R, S, alpha=1.0 / 32)
            assert torch.equal(out, ref), f"iteration {it}: {(out.float() - ref.float()).abs().max().item()}"
        torch.cuda.synchronize()
    finally:
        hip.gemm_persistent_mode(prev)


def _paired_rows(W, nqk_heads):
    perm = hip.qkrope_perm(128)
    rows = torch.cat([(torch.arange(nqk_heads)[:, None] * 128 + perm[None, :]).reshape(-1), torch.arange(nqk_heads * 128, W.shape[0])])
    return W[rows.to(W.device)].contiguous()


@pytest.mark.parametrize("lora", [False, True])
def test_qkv_projection_with_qknorm_rope_epilogue(lora):
    """q/k-norm + RoPE as the epilogue of the merged q|k|v launch (ur_gemm_args.qkr_*; Qwen3Attention, modeling_qwen3.py:227-245)
    against the projection followed by ur_qknorm_rope_fwd: v bit for bit, q_r / k_r to bf16 rounding (the epilogue normalises
    the f32 accumulators, the separate pass the bf16-rounded projection), and its backward from the roped outputs."""
    B, S, D, nq, nkv, hd = 4, 2048, 1024, 16, 8, 128
    M, NQ, NKV = B * S, nq * hd, nkv * hd
    x, W = _randn((M, D), 91), _randn((NQ + 2 * NKV, D), 92, 0.05)
    qw = (1.0 + 0.1 * torch.randn(hd, generator=torch.Generator().manual_seed(93))).to(DEV)
    kw = (1.0 + 0.1 * torch.randn(hd, generator=torch.Generator().manual_seed(94))).to(DEV)
    cos, sin = hip.rope_table(S, hd, 1e6, DEV)
    t, Bm = (_randn((M, 48), 95), _randn((NQ + 2 * NKV, 48), 96, 0.1)) if lora else (None, None)
    assert hip.gemm_qkrope_supported(M, NQ + 2 * NKV, D, 48 if lora else 0, S, NQ, NKV, DEV)
    assert not hip.gemm_qkrope_supported(1024, NQ + 2 * NKV, D, 0, 256, NQ, NKV, DEV)        # 64 tiles, too few for the persistent kernel: the separate pass
    raw = hip.gemm(x, W, R2=t, S2=Bm)
    q0, k0 = hip.qknorm_rope_fwd(raw, qw, kw, cos, sin, S, nq, nkv, hd, 1e-6)
    q1, k1, v1, rstd = hip.gemm_qkv_rope(x, _paired_rows(W, nq + nkv), qw, kw, cos, sin, S, NQ, NKV, 1e-6, R2=t,
                                         S2=None if Bm is None else _paired_rows(Bm, nq + nkv))
    torch.cuda.synchronize()
    assert torch.equal(v1, raw[:, NQ + NKV:])
    for a, b, n in ((q1, q0, "q_r"), (k1, k0, "k_r")):
        rel = float((a.float() - b.float()).norm() / b.float().norm())
        assert rel <= 6e-3, (n, rel)
    ref_rs = torch.rsqrt(raw[:, :NQ + NKV].float().view(M, nq + nkv, hd).pow(2).mean(-1) + 1e-6)
    assert float((rstd - ref_rs).abs().max() / ref_rs.abs().max()) <= 1e-2
    # backward: gradients of the raw q, k from the roped outputs + rstd against the kernel that reads the raw projection
    dq, dk = _randn((M, NQ), 97), _randn((M, NKV), 98)
    d0 = torch.zeros((M, NQ + 2 * NKV), dtype=torch.bfloat16, device=DEV)
    d1 = torch.zeros_like(d0)
    hip.qknorm_rope_bwd(dq, dk, raw, qw, kw, cos, sin, d0, S, nq, nkv, hd, 1e-6)
    hip.qknorm_rope_bwd_roped(dq, dk, q1, k1, rstd, qw, kw, cos, sin, d1, S, nq, nkv, hd)
    rel = float((d1.float() - d0.float()).norm() / d0.float().norm())
    assert rel <= 1e-2, rel


@pytest.mark.parametrize("lora", [False, True])
def test_gate_up_projection_with_paired_swiglu_epilogue(lora):
    """SwiGLU forward as the epilogue of the merged gate|up launch (ur_gemm_args.swp_*; Qwen3MLP, modeling_qwen3.py:81-91):
    gate | up in the standard order and act = silu(gate) * up, bit for bit what the projection + ur_swiglu_fwd give."""
    M, D, I = 8192, 1024, 3072
    x, W = _randn((M, D), 101), _randn((2 * I, D), 102, 0.05)
    t, Bm = (_randn((M, 32), 103), _randn((2 * I, 32), 104, 0.1)) if lora else (None, None)
    assert hip.gemm_swiglu_paired_supported(M, I, D, 32 if lora else 0, DEV)
    assert not hip.gemm_swiglu_paired_supported(1024, I, D, 0, DEV)
    rows = hip.swiglu_pair_rows(I).to(DEV)
    gu0 = hip.gemm(x, W, R2=t, S2=Bm)
    act0 = hip.swiglu_fwd(gu0, I)
    gu1 = torch.empty_like(gu0)
    act1 = torch.empty((M, I), dtype=torch.bfloat16, device=DEV)
    hip.gemm(x, W[rows].contiguous(), out=gu1, R2=t, S2=None if Bm is None else Bm[rows].contiguous(), swiglu_paired=act1)
    torch.cuda.synchronize()
    assert torch.equal(gu1, gu0)
    assert torch.equal(act1, act0)


# ---- seeded sweep over shapes, row strides and epilogue combinations: persistent == generic, bit for bit ----
# (round 5 re-mapped which output columns a wave owns and how its epilogue addresses every operand; the cases above pin each epilogue at
# one or two shapes, this one walks odd tile counts, K tile counts of both parities, strided views of every operand and the combinations)
@pytest.mark.parametrize("seed", list(range(12)))
def test_persistent_equals_generic_on_random_configurations(seed):
    import random
    rng = random.Random(1000 + seed)
    N = rng.choice([2048, 2304, 3072, 4096])
    M = 256 * rng.randint(max(4, (128 * 65536) // (256 * N) + 1), 40)
    K = 64 * rng.randint(4, 20)
    kind = ["plain", "bias", "residual", "bias_residual", "lora", "lora_residual", "drop", "gelu_out", "gelu_grad", "swiglu_bwd", "swiglu_bwd_drop", "drop3"][seed]
    pad = lambda cols: rng.choice([0, 8, 64])

    def strided(shape, sd, std=1.0):            # a column view of a wider buffer: row stride != width, 16-byte aligned start
        extra = pad(shape[1])
        off = rng.choice([0, 8, 16]) if extra else 0
        buf = _randn((shape[0], shape[1] + extra + (16 if extra else 0)), sd, std)
        return buf[:, off:off + shape[1]]
    R, S = strided((M, K), 11 * seed + 1), strided((N, K), 11 * seed + 2, 0.05)
    kw = {}
    r, p = 16, 0.1
    if kind in ("bias", "bias_residual", "gelu_out"):
        kw["bias"] = torch.randn(N, device=DEV)
    if kind in ("residual", "bias_residual", "lora_residual"):
        kw["residual"] = strided((M, N), 11 * seed + 3)
    if kind in ("lora", "lora_residual"):
        k2 = rng.choice([16, 32, 48])
        kw["R2"], kw["S2"] = strided((M, k2), 11 * seed + 4), strided((N, k2), 11 * seed + 5, 0.1)
    if kind in ("drop", "drop3", "swiglu_bwd_drop"):
        nad = 3 if kind == "drop3" else (1 if kind == "swiglu_bwd_drop" else rng.choice([1, 2]))
        kw["R2"], kw["S2"] = _randn((M, nad * r), 11 * seed + 4), _randn((N, nad * r), 11 * seed + 5, 0.1)
        kw["drop"] = (hip.lora_dropout_bits(seed + 3, p, M, N, nad, DEV), p, r)
    if kind == "gelu_grad":
        kw["gelu_grad_aux"] = strided((M, N), 11 * seed + 6, 1.5)
    if rng.random() < 0.5 and kind not in ("swiglu_bwd", "swiglu_bwd_drop"):
        kw["alpha"] = 0.5

    def run():
        if kind == "gelu_out":
            g = torch.zeros(M, N + 64, device=DEV, dtype=torch.bfloat16)[:, 32:32 + N]
            u = hip.gemm(R, S, gelu_out=g, **kw)
            return torch.cat([u, g], 1)
        if kind in ("swiglu_bwd", "swiglu_bwd_drop"):
            gu = _randn((M, 2 * N), 11 * seed + 7)
            dgu = torch.zeros_like(gu)
            hip.gemm(R, S, swiglu_bwd=(gu, dgu), **kw)
            return dgu
        out = torch.zeros(M, N + 64, device=DEV, dtype=torch.bfloat16)[:, 16:16 + N] if rng.random() < 0.5 else None
        return hip.gemm(R, S, out=out, **kw).clone()
    state = rng.getstate()
    prev = hip.gemm_persistent_mode(0)
    try:
        a = run()
        torch.cuda.synchronize()
        rng.setstate(state)
        hip.gemm_persistent_mode(1)
        b = run()
        torch.cuda.synchronize()
    finally:
        hip.gemm_persistent_mode(prev)
    assert a.shape == b.shape and torch.equal(a, b), f"{kind} M={M} N={N} K={K}: {(a.float() - b.float()).abs().max().item()}"
    assert torch.isfinite(b.float()).all() and b.float().abs().max().item() > 0
