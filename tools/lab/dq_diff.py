"""lab: where does a variant library's causal backward differ from the fp32 reference?  usage: UNIREC_HIP_LIB=... python tools/lab/dq_diff.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from unirec_amd import hip
B, S, nq, nkv, hd = 2, 1024, 4, 2, 128
g = torch.Generator().manual_seed(11)
buf = (torch.randn(B, S, (nq + 2 * nkv) * hd, generator=g) * 1.0).cuda().to(torch.bfloat16)
q = buf[..., :nq * hd].view(B, S, nq, hd); k = buf[..., nq * hd:(nq + nkv) * hd].view(B, S, nkv, hd); v = buf[..., (nq + nkv) * hd:].view(B, S, nkv, hd)
dout = torch.randn(B, S, nq, hd, generator=g).cuda().to(torch.bfloat16)
o, ctx = hip.attn_fwd(q, k, v, causal=True)
dq, dk, dv = hip.attn_bwd(ctx, dout)
qf, kf, vf = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
kk = kf.repeat_interleave(nq // nkv, dim=2); vv = vf.repeat_interleave(nq // nkv, dim=2)
s = torch.einsum("bqhd,bkhd->bhqk", qf, kk) * hd ** -0.5
s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device="cuda"), 1), float("-inf"))
ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vv)
ref.backward(dout.float())
for name, a, r in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
    e = (a.float() - r).abs()
    bad = e > 0.05 * r.abs().max()
    print(name, "max err", e.max().item(), "scale", r.abs().max().item(), "bad elements", int(bad.sum()))
    if bad.any():
        idx = bad.nonzero()
        rows = sorted(set((idx[:, 1] % 256).tolist()))
        print("   rows mod 256:", rows[:40], "... n", len(rows))
        print("   rows mod 64:", sorted(set((idx[:, 1] % 64).tolist())))
        print("   d:", sorted(set(idx[:, 3].tolist()))[:40])
        print("   blocks (row // 256):", sorted(set((idx[:, 1] // 256).tolist())), "heads", sorted(set(idx[:, 2].tolist())))
