#!/usr/bin/env python3
"""A/B of the persistent projection GEMM (csrc/gemm_pers.hip) against the generic 256x256 kernel on the C4 step's launches,
interleaved rounds in ONE process (cdna_hip_programming.md rule 24).  Usage: python tools/lab/gemm_pers_ab.py [--B 64] [--rounds 5]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip

# (N, K, kind, name): kind f = forward (+ LoRA second K range), r = forward + residual, d = dX with masked LoRA epilogue,
# s = dX with the SwiGLU backward epilogue (+ masked LoRA)
SHAPES = [(4096, 1024, "f", "q|k|v fwd", 3), (1024, 2048, "r", "o_proj fwd", 1), (6144, 1024, "f", "gate|up fwd", 2), (1024, 3072, "r", "down fwd", 1),
          (3072, 1024, "s", "dX down (swiglu)", 1), (1024, 6144, "d", "dX gate|up", 2), (2048, 1024, "d", "dX o_proj", 1), (1024, 4096, "d", "dX q|k|v", 3)]


def ksweep(a, M):
    N = 4096
    g = torch.Generator().manual_seed(0)
    Ks = [256, 512, 1024, 2048, 4096]
    Rf = torch.randn(M, max(Ks), generator=g).cuda().to(torch.bfloat16)
    Wf = (torch.randn(N, max(Ks), generator=g) * 0.05).cuda().to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    res = {}
    for rnd in range(a.rounds + 1):
        for mode in (0, 1):
            hip.gemm_persistent_mode(mode)
            for K in Ks:
                R, W = Rf[:, :K], Wf[:, :K]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                hip.gemm(R, W, out=out)
                e0.record()
                for _ in range(a.iters):
                    hip.gemm(R, W, out=out)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    res.setdefault((K, mode), []).append(e0.elapsed_time(e1) / a.iters)
    hip.gemm_persistent_mode(-1)
    rounds = (M // 256) * (N // 256) / 256.0          # output tiles per CU
    for mode in (0, 1):
        t = {K: statistics.median(res[(K, mode)]) for K in Ks}
        slope = (t[4096] - t[1024]) / ((4096 - 1024) / 64) / rounds * 1e3        # us per K tile per output tile
        fixed = t[1024] / rounds * 1e3 - slope * 16
        print(("persistent" if mode else "generic   ") + "  " + "  ".join(f"K={K}: {t[K]:.3f} ms ({2.0 * M * N * K / t[K] / 1e9:5.0f} TF/s)" for K in Ks))
        print(f"            {slope:.3f} us per K tile, {fixed:.2f} us fixed per output tile (rows of {max(Ks)} elements: the K < 4096 operands are strided views)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="", help="comma list of kinds (f, r, d, s) to run")
    ap.add_argument("--plain", action="store_true", help="no LoRA terms / epilogues: the bare GEMMs")
    ap.add_argument("--ksweep", action="store_true", help="plain GEMM, N = 4096, K = 256 .. 4096: per-K-tile and fixed cost per output tile of both kernels")
    a = ap.parse_args()
    M = a.B * a.S
    if a.ksweep:
        return ksweep(a, M)
    g = torch.Generator().manual_seed(0)
    ops = []
    for (N, K, kind, name, nad) in SHAPES:
        if a.only and kind not in a.only.split(","):
            continue
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        kw = {}
        if not a.plain:
            t = torch.randn(M, 16 * nad, generator=g).cuda().to(torch.bfloat16)
            Bm = (torch.randn(N, 16 * nad, generator=g) * 0.05).cuda().to(torch.bfloat16)
            kw = dict(R2=t, S2=Bm)
            if kind == "r":
                kw["residual"] = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
            if kind in "ds":
                kw["drop"] = (hip.lora_dropout_bits(1, 0.1, M, N, nad, "cuda"), 0.1, 16)
        if kind == "s":
            gu = torch.randn(M, 2 * N, generator=g).cuda().to(torch.bfloat16)
            kw["swiglu_bwd"] = (gu, torch.empty_like(gu))
            out = None
        else:
            out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.append((name, N, K, R, W, out, kw))
    times = {(name, m): [] for name, *_ in ops for m in (0, 1)}
    for rnd in range(a.rounds + 1):
        for mode in (0, 1):
            hip.gemm_persistent_mode(mode)
            for (name, N, K, R, W, out, kw) in ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                hip.gemm(R, W, out=out, **kw)
                e0.record()
                for _ in range(a.iters):
                    hip.gemm(R, W, out=out, **kw)
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    times[(name, mode)].append(e0.elapsed_time(e1) / a.iters)
    hip.gemm_persistent_mode(-1)
    tot = [0.0, 0.0]
    for (name, N, K, *_rest) in ops:
        t0, t1 = statistics.median(times[(name, 0)]), statistics.median(times[(name, 1)])
        tot[0] += t0; tot[1] += t1
        fl = 2.0 * M * N * K
        print(f"{name:18s} N={N:5d} K={K:5d}: generic {t0:.3f} ms ({fl / t0 / 1e9:6.0f} TF/s) | persistent {t1:.3f} ms ({fl / t1 / 1e9:6.0f} TF/s)  "
              f"{(t0 / t1 - 1) * 100:+.1f} %   min {min(times[(name, 0)]):.3f} / {min(times[(name, 1)]):.3f}")
    print(f"sum per layer: generic {tot[0]:.3f} ms | persistent {tot[1]:.3f} ms  ({(tot[0] / tot[1] - 1) * 100:+.1f} %)")


if __name__ == "__main__":
    main()
