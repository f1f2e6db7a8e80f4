#!/usr/bin/env python3
"""Micro-benchmarks of the individual hot kernels (attention fwd/bwd, projection GEMMs) at the C4
shapes; used under rocprofv3 (--kernel-trace / --pmc) when tuning.  Usage:
    python tools/kernel_bench.py attn|gemm|lora [--B 8] [--iters 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unirec_amd import hip


def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def attn(args):
    B, S, nq, nkv, hd = args.B, args.S, 16, 8, 128
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g)).cuda().to(torch.bfloat16)
    q = qkv[..., :nq * hd].view(B, S, nq, hd); k = qkv[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = qkv[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
    dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
    fl = 4 * B * nq * S * S * hd / 2
    o, ctx = hip.attn_fwd(q, k, v, causal=True)
    t = timeit(lambda: hip.attn_fwd(q, k, v, causal=True), args.iters)
    print(f"attn fwd  B={B} S={S}: {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s")
    t = timeit(lambda: hip.attn_bwd(ctx, dout), args.iters)
    print(f"attn bwd  B={B} S={S}: {t:.3f} ms  {2.5 * fl / t / 1e9:.1f} TFLOP/s (2.5x fwd flops; 3.5x executed)")


def gemm(args):
    M = args.B * args.S
    g = torch.Generator().manual_seed(0)
    for (N, K, rk, sk, name) in [(2048, 1024, True, True, "q_proj fwd"), (3072, 1024, True, True, "gate fwd"), (1024, 3072, True, True, "down fwd"),
                                 (1024, 4096, True, False, "dX qkv"), (3072, 1024, True, False, "dX down"), (1024, 6144, True, False, "dX gate|up")]:
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        S_ = (torch.randn(N, K, generator=g) if sk else torch.randn(K, N, generator=g)).cuda().to(torch.bfloat16) * 0.05
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        t = timeit(lambda: hip.gemm(R, S_, r_kcontig=rk, s_kcontig=sk, out=out), args.iters)
        line = f"gemm {name:12s} M={M} N={N} K={K}: {t:.3f} ms  {2 * M * N * K / t / 1e9:.1f} TFLOP/s"
        if args.lib:       # reference point only (the product never calls it): the vendor library GEMM torch dispatches to
            Sm = S_ if sk else S_.t()
            tl = timeit(lambda: torch.matmul(R, Sm.t(), out=out), args.iters)
            line += f"   | torch.matmul (hipBLASLt / rocBLAS): {tl:.3f} ms  {2 * M * N * K / tl / 1e9:.1f} TFLOP/s"
        print(line)


def dw(args):
    """Token-reduction (dW) GEMM of the user Q-Former's K|V projection at C3: out[2048,1024] = dKV^T X over 819200
    tokens; token-major operands (K-strided, as the backward has them) against pre-transposed K-contiguous ones."""
    Kt, Mo, No = 512 * 1600, 2048, 1024
    g = torch.Generator().manual_seed(0)
    dy = torch.randn(Kt, Mo, generator=g).cuda().to(torch.bfloat16)
    x = torch.randn(Kt, No, generator=g).cuda().to(torch.bfloat16)
    out = torch.empty(Mo, No, device="cuda")
    fl = 2.0 * Kt * Mo * No
    for sp in (4, 8, 16):
        t = timeit(lambda: hip.gemm(dy, x, r_kcontig=False, s_kcontig=False, out=out, split_k=sp), args.iters)
        print(f"dW token-major  split {sp:2d}: {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s")
    ref = out.clone()
    dyT, xT = dy.t().contiguous(), x.t().contiguous()
    for sp in (8, 16):
        t = timeit(lambda: hip.gemm(dyT, xT, r_kcontig=True, s_kcontig=True, out=out, split_k=sp), args.iters)
        print(f"dW transposed   split {sp:2d}: {t:.3f} ms  {fl / t / 1e9:.1f} TFLOP/s   max|diff| vs token-major {float((out - ref).abs().max()):.3e}")
    t = timeit(lambda: hip.transpose_bf16(dy), args.iters)
    print(f"transpose [819200,2048] bf16: {t:.3f} ms")


def xattn(args):
    """User Q-Former cross-attention at C3: 64 queries x 1600 keys, B*heads = 512*16, ragged key mask, p = 0.1."""
    B, nh, Sq, Sk = 512, 16, 64, 1600
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, Sq, nh, 64, generator=g) * 0.5).cuda().to(torch.bfloat16)
    kv = (torch.randn(B, Sk, 2, nh, 64, generator=g) * 0.5).cuda().to(torch.bfloat16)
    lens = torch.randint(Sk // 2, Sk + 1, (B,), generator=g)
    km = (torch.arange(Sk)[None, :] < lens[:, None]).to(torch.uint8).cuda()
    dout = torch.randn(B, Sq, nh, 64, generator=g).cuda().to(torch.bfloat16)
    dkv = torch.empty_like(kv)
    for pd in (0.1, 0.0):
        o, ctx = hip.attn_fwd(q, kv[:, :, 0], kv[:, :, 1], causal=False, key_mask=km, dropout_p=pd, seed=3)
        t = timeit(lambda: hip.attn_fwd(q, kv[:, :, 0], kv[:, :, 1], causal=False, key_mask=km, dropout_p=pd, seed=3), args.iters)
        print(f"xattn fwd p={pd}: {t:.3f} ms")
        for sw in ("0", "1"):
            with hip.attn_mode_set(hip.ATTN_MODE_FEWQ, int(sw)):
                t = timeit(lambda: hip.attn_bwd(ctx, dout, dk=dkv[:, :, 0], dv=dkv[:, :, 1]), args.iters)
            print(f"xattn bwd (dq + dkv) p={pd} ur_attn_mode(FEWQ, {sw}): {t:.3f} ms")


def gemm_merge(args):
    """One projection launch over q|k|v (N = 4096) or gate|up (N = 6144) against the separate launches the decoder issues
    today (one per LoRA adapter): what merging them behind a block-diagonal second K range would buy."""
    M, K = args.B * args.S, 1024
    g = torch.Generator().manual_seed(0)
    R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    for name, parts in (("q|k|v", (2048, 1024, 1024)), ("gate|up", (3072, 3072))):
        N = sum(parts)
        W = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        def separate():
            c = 0
            for n in parts:
                hip.gemm(R, W[c:c + n], out=out[:, c:c + n]); c += n
        t1 = timeit(separate, args.iters)
        t2 = timeit(lambda: hip.gemm(R, W, out=out), args.iters)
        print(f"gemm {name:8s} N={N}: separate launches {t1:.3f} ms | one launch {t2:.3f} ms  ({2 * M * N * K / t2 / 1e9:.0f} TFLOP/s)")


STEP_SHAPES = [(4096, 1024, True, "q|k|v fwd"), (1024, 2048, True, "o_proj fwd"), (6144, 1024, True, "gate|up fwd"), (1024, 3072, True, "down fwd"),
               (1024, 4096, False, "dX q|k|v"), (2048, 1024, False, "dX o_proj"), (1024, 6144, False, "dX gate|up")]


def gemm_step(args):
    """The K-contiguous 256x256 projection launches the C4 step actually issues (merged q|k|v and gate|up), ONE launch each
    in a fixed order after one warm-up round -- for the rocprofv3 --pmc passes (tools/gemm_pmc.py maps dispatch order to shape).
    dX launches read the frozen weights' transposed copies, i.e. they are K-contiguous [N,K] operands as well."""
    M = args.B * args.S
    g = torch.Generator().manual_seed(0)
    ops = []
    for (N, K, fwd, name) in STEP_SHAPES:
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        ops.append((R, W, torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), name, N, K))
    for rnd in range(2):          # round 0 = warm-up, round 1 = the measured dispatches (the last 7 gemm_kernel dispatches)
        for (R, W, out, name, N, K) in ops:
            hip.gemm(R, W, out=out)
        torch.cuda.synchronize()
    for (R, W, out, name, N, K) in ops:
        t = timeit(lambda: hip.gemm(R, W, out=out), args.iters)
        print(f"gemm {name:12s} M={M} N={N} K={K}: {t:.3f} ms  {2 * M * N * K / t / 1e9:.1f} TFLOP/s")


def rmslora(args):
    """RMSNorm forward + lora_project as two kernels against the fused ur_rmsnorm_lora_fwd (q|k|v: 3 adapters, gate|up: 2)."""
    M, D = args.B * args.S, 1024
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, D, generator=g).cuda().to(torch.bfloat16)
    w = torch.ones(D).cuda()
    for nad in (3, 2):
        U = [(torch.randn(16, D, generator=g) * 0.05).cuda().to(torch.bfloat16) for _ in range(nad)]
        bits = hip.lora_dropout_bits(1, 0.1, M, D, nad, "cuda")
        t_n = timeit(lambda: hip.rmsnorm_fwd(x, w, 1e-6), args.iters)
        h, _ = hip.rmsnorm_fwd(x, w, 1e-6)
        t_p = timeit(lambda: hip.lora_project(h, U, alpha=2.2, bits=bits), args.iters)
        t_f = timeit(lambda: hip.rmsnorm_lora_fwd(x, w, 1e-6, U, alpha=2.2, bits=bits), args.iters)
        gb = 2.0 * M * D * 2 / 1e9
        print(f"nad={nad}: rmsnorm {t_n * 1e3:.1f} us + project {t_p * 1e3:.1f} us = {(t_n + t_p) * 1e3:.1f} us | fused {t_f * 1e3:.1f} us "
              f"({gb / t_f:.2f} TB/s on x + h)")


def swilora(args):
    """SwiGLU forward + the down adapter's lora_project as two kernels against the fused ur_swiglu_lora_fwd."""
    M, I = args.B * args.S, 3072
    g = torch.Generator().manual_seed(0)
    gu = torch.randn(M, 2 * I, generator=g).cuda().to(torch.bfloat16)
    U = (torch.randn(16, I, generator=g) * 0.05).cuda().to(torch.bfloat16)
    bits = hip.lora_dropout_bits(1, 0.1, M, I, 1, "cuda")
    t_s = timeit(lambda: hip.swiglu_fwd(gu, I), args.iters)
    act = hip.swiglu_fwd(gu, I)
    t_p = timeit(lambda: hip.lora_project(act, [U], alpha=2.2, bits=bits), args.iters)
    t_f = timeit(lambda: hip.swiglu_lora_fwd(gu, I, U, alpha=2.2, bits=bits), args.iters)
    gb = 3.0 * M * I * 2 / 1e9
    print(f"swiglu {t_s * 1e3:.1f} us + project {t_p * 1e3:.1f} us = {(t_s + t_p) * 1e3:.1f} us | fused {t_f * 1e3:.1f} us ({gb / t_f:.2f} TB/s on gate|up + act)")


def gemm_lora(args):
    """What the LoRA term costs inside the projection GEMMs: plain / + second K range (forward: t B^T, K2 = 16) /
    + masked rank-16 epilogue (dX under LoRA dropout)."""
    M, r = args.B * args.S, 16
    g = torch.Generator().manual_seed(0)
    for (N, K, nad, name) in [(2048, 1024, 1, "q fwd"), (3072, 1024, 1, "gate fwd"), (1024, 3072, 1, "down fwd"),
                              (1024, 4096, 3, "dX qkv"), (3072, 1024, 1, "dX down"), (1024, 6144, 2, "dX gate|up")]:
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        S_ = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        t = torch.randn(M, nad * r, generator=g).cuda().to(torch.bfloat16)
        Bm = (torch.randn(N, nad * r, generator=g) * 0.05).cuda().to(torch.bfloat16)
        bits = hip.lora_dropout_bits(1, 0.1, M, N, nad, "cuda")
        out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        t0 = timeit(lambda: hip.gemm(R, S_, out=out), args.iters)
        t1 = timeit(lambda: hip.gemm(R, S_, out=out, R2=t, S2=Bm), args.iters)
        t2 = timeit(lambda: hip.gemm(R, S_, out=out, R2=t, S2=Bm, drop=(bits, 0.1, r)), args.iters)
        print(f"gemm {name:12s} N={N} K={K}: plain {t0:.3f} ms | + K2={nad * r} range {t1:.3f} ms | + masked epilogue ({nad} adapters) {t2:.3f} ms")


def rope(args):
    """q/k-norm + RoPE forward / backward over the raw q|k|v projection (16 q + 8 kv heads of 128): TB/s of raw q|k in + q_r|k_r out."""
    M, S, nq, nkv, hd = args.B * args.S, args.S, 16, 8, 128
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(M, (nq + 2 * nkv) * hd, generator=g).cuda().to(torch.bfloat16)
    qn, kn = torch.ones(hd).cuda(), torch.ones(hd).cuda()
    cos, sin = hip.rope_table(S, hd, 1e6, "cuda")
    t = timeit(lambda: hip.qknorm_rope_fwd(qkv, qn, kn, cos, sin, S, nq, nkv, hd, 1e-6), args.iters)
    nb = 2.0 * M * (nq + nkv) * hd * 2
    print(f"qknorm_rope fwd: {t * 1e3:.1f} us  {nb / t / 1e9:.2f} TB/s (raw q|k read + q_r|k_r written)")
    q_r, k_r = hip.qknorm_rope_fwd(qkv, qn, kn, cos, sin, S, nq, nkv, hd, 1e-6)
    dqkv = torch.empty_like(qkv)
    t = timeit(lambda: hip.qknorm_rope_bwd(q_r, k_r, qkv, qn, kn, cos, sin, dqkv, S, nq, nkv, hd, 1e-6), args.iters)
    print(f"qknorm_rope bwd: {t * 1e3:.1f} us  {1.5 * nb / t / 1e9:.2f} TB/s (dq_r|dk_r + raw q|k read, dq|dk raw written)")
    rstd = torch.rsqrt(qkv[:, :(nq + nkv) * hd].float().view(M, nq + nkv, hd).pow(2).mean(-1) + 1e-6)
    t = timeit(lambda: hip.qknorm_rope_bwd_roped(q_r, k_r, q_r, k_r, rstd, qn, kn, cos, sin, dqkv, S, nq, nkv, hd), args.iters)
    print(f"qknorm_rope bwd from the roped outputs: {t * 1e3:.1f} us  {1.5 * nb / t / 1e9:.2f} TB/s")


def lora(args):
    """The rank-16 LoRA side kernels at the C4 shapes (M = B*S tokens): GB/s of the activation they stream."""
    M, r = args.B * args.S, 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, 1024, generator=g).cuda().to(torch.bfloat16)
    dy = torch.randn(M, 4096, generator=g).cuda().to(torch.bfloat16)
    A = [(torch.randn(r, 1024, generator=g) * 0.1).cuda().to(torch.bfloat16) for _ in range(3)]
    cols = [(0, 2048), (2048, 1024), (3072, 1024)]
    Bt = [(torch.randn(r, n, generator=g) * 0.1).cuda().to(torch.bfloat16) for _, n in cols]
    t = torch.randn(M, 3 * r, generator=g).cuda().to(torch.bfloat16)
    bits = hip.lora_dropout_bits(1, 0.1, M, 1024, 3, "cuda")
    gA, gB = torch.empty(3 * r, 1024, device="cuda"), torch.empty(4096, r, device="cuda")
    for name, fn, nbytes in (
            ("bits   3 planes of [M,1024]", lambda: hip.lora_dropout_bits(1, 0.1, M, 1024, 3, "cuda"), 3 * M * 1024 / 8),
            ("project t  = drop(x) A^T (3)", lambda: hip.lora_project(x, A, bits=bits), x.numel() * 2),
            ("project tb = dy B        (3)", lambda: hip.lora_project(dy, Bt, cols=cols), dy.numel() * 2),
            ("reduce  dB = dy^T t      (3)", lambda: hip.lora_reduce(dy, t, gB, cols=cols, transposed=True), dy.numel() * 2),
            ("bgrad   tb and dB, one pass(3)", lambda: hip.lora_bgrad(dy, t, Bt, cols, gB), dy.numel() * 2),
            ("reduce  dA = tb^T drop(x)(3)", lambda: hip.lora_reduce(x, t, gA, nad=3, bits=bits), x.numel() * 2)):
        ms = timeit(fn, args.iters)
        print(f"lora {name}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e6:7.1f} GB/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "gemm", "lora", "rope", "dw", "xattn", "gemm_lora", "gemm_merge", "gemm_step", "rmslora", "swilora"])
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--lib", action="store_true", help="gemm: also time torch.matmul on the same operands (reference point)")
    a = ap.parse_args()
    {"attn": attn, "gemm": gemm, "lora": lora, "rope": rope, "dw": dw, "xattn": xattn, "gemm_lora": gemm_lora, "gemm_merge": gemm_merge, "gemm_step": gemm_step, "rmslora": rmslora, "swilora": swilora}[a.what](a)
