#!/usr/bin/env python3
"""A/B of the wave-specialised projection GEMM (csrc/gemm_ws.hip, ur_gemm_persistent_mode(2)) against the persistent 256x256 kernel
(mode 1) on the C4 step's shapes WITHOUT LoRA terms, interleaved rounds in ONE process; checks the outputs against each other first.
Usage: python tools/lab/gemm_ws_ab.py [--B 64] [--rounds 5]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip

SHAPES = [(4096, 1024, "p", "q|k|v"), (1024, 2048, "p", "o_proj"), (6144, 1024, "p", "gate|up"), (1024, 3072, "p", "down"),
          (3072, 1024, "s", "dX down (swiglu)"), (1024, 6144, "p", "dX gate|up"), (2048, 1024, "p", "dX o_proj"), (1024, 4096, "p", "dX q|k|v")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    M = a.B * a.S
    g = torch.Generator().manual_seed(0)
    ops = []
    R_of, W_of, gu_of = {}, {}, {}
    for (N, K, kind, name) in SHAPES:
        if a.only and kind not in a.only.split(","):
            continue
        R = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
        kw, outs = {}, []
        R_of[name], W_of[name] = R, W
        if kind == "s":
            gu = torch.randn(M, 2 * N, generator=g).cuda().to(torch.bfloat16)
            outs = [torch.empty_like(gu), torch.empty_like(gu)]
            gu_of[name] = gu
            mk = lambda o: dict(swiglu_bwd=(gu, o))
            call = lambda o, R=R, W=W, mk=mk: hip.gemm(R, W, out=None, **mk(o))
        else:
            outs = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
            call = lambda o, R=R, W=W: hip.gemm(R, W, out=o)
        ops.append((name, N, K, call, outs, kind))
    # correctness: mode 2 against mode 1
    for (name, N, K, call, outs, kind) in ops:
        for o in outs:
            o.fill_(float("nan"))
        hip.gemm_persistent_mode(1); call(outs[0])
        hip.gemm_persistent_mode(2); call(outs[1])
        torch.cuda.synchronize()
        if kind == "s":      # the wave-specialised epilogue rounds d(act) to bf16 first: bit-identical to the UNFUSED pair plain GEMM + ur_swiglu_bwd
            hip.gemm_persistent_mode(1)
            ref = hip.swiglu_bwd(hip.gemm(R_of[name], W_of[name]), gu_of[name], N)
            torch.cuda.synchronize()
            print(f"{name:18s} vs the unfused pair: differing elements {int((ref != outs[1]).sum())}", flush=True)
            del ref
        a0, a1 = outs[0].float(), outs[1].float()
        bad = int((outs[0] != outs[1]).sum())
        err = float((a0 - a1).abs().max())
        ref = float(a0.abs().max())
        print(f"{name:18s} N={N:5d} K={K:5d}: differing elements {bad} of {a0.numel()}, max |diff| {err:.4g} (max |ref| {ref:.4g}), nan {int(torch.isnan(a1).sum())}", flush=True)
    times = {(name, m): [] for name, *_ in ops for m in (1, 2)}
    for rnd in range(a.rounds + 1):
        for mode in (1, 2):
            hip.gemm_persistent_mode(mode)
            for (name, N, K, call, outs, kind) in ops:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                call(outs[0])
                e0.record()
                for _ in range(a.iters):
                    call(outs[0])
                e1.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    times[(name, mode)].append(e0.elapsed_time(e1) / a.iters)
    hip.gemm_persistent_mode(-1)
    tot = [0.0, 0.0]
    for (name, N, K, *_rest) in ops:
        t0, t1 = statistics.median(times[(name, 1)]), statistics.median(times[(name, 2)])
        tot[0] += t0; tot[1] += t1
        fl = 2.0 * M * N * K
        print(f"{name:18s} N={N:5d} K={K:5d}: persistent {t0:.3f} ms ({fl / t0 / 1e9:6.0f} TF/s) | wave-specialised {t1:.3f} ms ({fl / t1 / 1e9:6.0f} TF/s)  "
              f"{(t0 / t1 - 1) * 100:+.1f} %   min {min(times[(name, 1)]):.3f} / {min(times[(name, 2)]):.3f}")
    print(f"sum: persistent {tot[0]:.3f} ms | wave-specialised {tot[1]:.3f} ms  ({(tot[0] / tot[1] - 1) * 100:+.1f} %)")


if __name__ == "__main__":
    main()
