#!/usr/bin/env python3
"""Relative Frobenius error of the tiny (DPP-row) and MFMA attention kernels against an fp64 reference computed from the
SAME bf16 inputs (2 x 2 self-attention-like and 2 x 14 cross-attention-like shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unirec_amd import hip

def ref(q, k, v, dout):
    q, k, v = (t.double().detach().clone().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * 64 ** -0.5
    o = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), v)
    o.backward(dout.double())
    return o, q.grad, k.grad, v.grad

def rel(a, b):
    return float((a.double() - b).norm() / b.norm())

for (Sq, Sk, selfattn) in ((2, 2, True), (2, 14, False), (4, 4, True)):
    g = torch.Generator().manual_seed(Sq * 100 + Sk)
    B, nh = 400, 16
    x = torch.randn(B, Sq, nh, 64, generator=g)
    q = (x * 0.8 + 0.1 * torch.randn(B, Sq, nh, 64, generator=g)).cuda().bfloat16()
    if selfattn:
        k = (x * 0.7 + 0.1 * torch.randn(B, Sk, nh, 64, generator=g)).cuda().bfloat16()
        v = (x * 0.9 + 0.1 * torch.randn(B, Sk, nh, 64, generator=g)).cuda().bfloat16()
    else:
        k = torch.randn(B, Sk, nh, 64, generator=g).cuda().bfloat16()
        v = torch.randn(B, Sk, nh, 64, generator=g).cuda().bfloat16()
    dout = torch.randn(B, Sq, nh, 64, generator=g).cuda().bfloat16()
    want = ref(q, k, v, dout)
    for sw in ("0", "1"):
        hip.attn_mode(hip.ATTN_MODE_TINY, 3 if sw == "1" else 0)
        o, ctx = hip.attn_fwd(q, k, v, causal=False)
        dq, dk, dv = hip.attn_bwd(ctx, dout)
        torch.cuda.synchronize()
        print(f"Sq={Sq} Sk={Sk} UR_ATTN_TINY={sw}: o {rel(o, want[0]):.2e}  dq {rel(dq, want[1]):.2e}  dk {rel(dk, want[2]):.2e}  dv {rel(dv, want[3]):.2e}")
