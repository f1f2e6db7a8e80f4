"""Lab: every LoRA ring launch of ONE decoder layer of the C4 step at its real shape (M = B * S tokens), each timed alone:
   forward   project_ring<masked>   o (x = attention output [M, 2048]) and down (x = act [M, 3072])
   backward  bgrad_ring<4>          q|k|v (dy [M, 4096], 3 entries) and gate|up (d [M, 6144], 2 entries)
             bgrad_ring<2>          o, down (dy [M, 1024], one entry)
             reduce_ring<1, masked> dA of o ([M, 2048]) and down ([M, 3072])
             reduce_ring<3 / 2, masked> dA of q|k|v and gate|up (x = h [M, 1024], shared input)
usage: [UNIREC_HIP_LIB=...] python tools/lab/lora_step_bench.py [--B 64] [--iters 20]; prints us per launch and the layer's sum."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from unirec_amd import hip  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--S", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    M, r = args.B * args.S, 16
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).cuda().to(torch.bfloat16)
    W16 = lambda n: (torch.randn(r, n, generator=g) * 0.1).cuda().to(torch.bfloat16)
    rows = []

    def add(name, fn, nbytes):
        ms = timeit(fn, args.iters)
        rows.append((name, ms))
        print(f"{name:46s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.1f} GB/s", flush=True)

    h = rnd(M, 1024)
    for W, tag in ((2048, "o"), (3072, "down")):
        x = rnd(M, W)
        bits = hip.lora_dropout_bits(1, 0.1, M, W, 1, "cuda")
        bt = hip.lora_bits_transpose(bits, W)
        A = [W16(W)]
        t = rnd(M, r)
        gA = torch.empty(r, W, device="cuda")
        add(f"project_ring<masked> {tag} [M,{W}]", lambda x=x, A=A, bits=bits: hip.lora_project(x, A, bits=bits), x.numel() * 2)
        add(f"reduce_ring<1,masked> dA {tag} [M,{W}]", lambda x=x, t=t, gA=gA, bits=bits, bt=bt: hip.lora_reduce(x, t, gA, nad=1, bits=bits, bits_t=bt), x.numel() * 2)
        del x, bits, bt
    for nad, tag in ((3, "q|k|v"), (2, "gate|up")):
        bits = hip.lora_dropout_bits(1, 0.1, M, 1024, nad, "cuda")
        bt = hip.lora_bits_transpose(bits, 1024)
        t = rnd(M, nad * r)
        gA = torch.empty(nad * r, 1024, device="cuda")
        add(f"reduce_ring<{nad},masked> dA {tag} [M,1024]", lambda t=t, gA=gA, bits=bits, bt=bt, nad=nad: hip.lora_reduce(h, t, gA, nad=nad, bits=bits, bits_t=bt), h.numel() * 2)
        del bits, bt
    for cols, tag in (([(0, 2048), (2048, 1024), (3072, 1024)], "q|k|v"), ([(0, 3072), (3072, 3072)], "gate|up"), ([(0, 1024)], "o"), ([(0, 1024)], "down")):
        Wt = sum(w for _, w in cols)
        dy = rnd(M, Wt)
        Bt = [W16(w) for _, w in cols]
        t = rnd(M, len(cols) * r)
        gB = torch.empty(Wt, r, device="cuda")
        add(f"bgrad_ring {tag} [M,{Wt}] ({len(cols)})", lambda dy=dy, t=t, Bt=Bt, cols=cols, gB=gB: hip.lora_bgrad(dy, t, Bt, cols, gB), dy.numel() * 2)
        del dy
    print(f"layer sum {sum(ms for _, ms in rows) * 1e3:8.1f} us  (x 28 layers = {sum(ms for _, ms in rows) * 28:6.2f} ms per step)")


if __name__ == "__main__":
    main()
